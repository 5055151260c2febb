// Generation 7 of om_gemm_nt for 16-bit inputs and outputs (bf16 inference epilogues): the K loop of
// gemm_core7.h inside a PERSISTENT kernel -- one workgroup per CU walks 256 x 256 tiles, and the first K
// step of the next tile (64 KiB of operands, the ~5 k-cycle cold start of every tile in generation 6)
// is fetched while the current tile's epilogue runs.
//
// Why this works now and did not in round 1 (gemm_wide6.h, "persistent form ... dropped twice"): vmcnt
// retires in order, loads and stores alike.  The next tile's fetch is issued BEFORE the epilogue's
// stores and every later wait is an exact count of the younger operations ("32 stores + 16 DMA may
// still be outstanding"), so nothing ever waits for a store acknowledgement.  Exact counts need a
// fixed instruction stream: this kernel only takes problems made of whole tiles (M, N multiples of
// 256, K of 64 -- the encoder pads its token count) and issues every DMA unconditionally (a workgroup
// without a next tile re-fetches its own).
//
// LDS (160 KiB, all dynamic; gemm_core7.h's five 32 KiB units U0..U4):
//   K loop        U0-U4 rotate
//   epilogue      U0, U1   step 0 of the next tile (in flight)
//                 U2, U3   per wave 16 slices of 1 KiB (its own DMA targets): residual ring 3 x 4 KiB, bf16 staging 4 KiB
//                 U4       [0, 8 KiB)   accumulator-init tables of the NEXT tile, 2 KiB per wave:
//                                         s_n | b'_n (128 + 128 f32), LayerNorm statistics of the wave's 128 rows
//                          [8, 16 KiB)  epilogue tables, 2 KiB per wave: gamma | beta, statistics of the residual rows
//   The tables are consumed (accumulator initialisation) before the K loop's first barrier; the K loop's own DMA into U4
//   (the spare unit, A of step 2) is issued after that barrier, so it cannot overwrite a table that is still needed.
// One patch = 32 rows x 64 columns (8 per wave tile): staged as packed bf16 (row = 128 B, 16-byte chunk XOR
// (row & 7)), read back row-major, stored as whole 128-byte lines.
//
// LNF == 3 (round 3): the LNF == 2 epilogue on a TWO-PLANE residual stream.  The pre-LayerNorm sums y are stored as
// y_hi = round16(y) (the plane the next GEMM reads as its A operand) and y_lo = round16(y - y_hi) (ep.out_lo); the
// residual is read as r_hi + r_lo (ep.resid_lo).  The reference's autocast keeps that stream in f32
// (HF:models/bert/modeling_bert.py:289-293,347-351 under torch.autocast); one 16-bit plane was the single reason the
// fused path sat at 1 - cos 4.8e-5 against the reference's own 1.8e-5 (tools/emulate_16bit_dataflow.py: 1.1e-5 with the
// second plane).  LDS of this variant: residual ring of TWO entries, each 4 KiB hi + 4 KiB lo, in the wave's 16 slices
// of U2 / U3; the staging patch moves to the upper half of U4 (4 KiB per wave at 16 KiB + wave * 4 KiB, plain rows of
// 128 B) and is used twice per patch (hi, then lo -- a wave's LDS operations execute in order).
#pragma once
#include <type_traits>
#include "gemm_core7.h"
#include "gemm_epilogue6.h"

#define G7E_RES_DEPTH 3
// Epilogue buffers of a wave live in the 1 KiB slices of U2 / U3 that the SAME wave's K-loop DMA instructions fill
// (instruction i of wave w -> bytes [(4i + w) KiB, +1 KiB) of a unit): slice s = 0..15 of wave w.  A 4 KiB buffer
// (32 rows x 128 B) takes four consecutive slices, row r at slice (r >> 3), byte (r & 7) * 128.  Slices 0-11: the
// residual ring (3 patches); 12-15: the bf16 staging patch.  Because no wave ever touches another wave's slices
// here, K step 1 of the next tile can be issued by each wave as soon as ITS epilogue is done -- no barrier.
#define G7E_SLICE(S, WAVE) (2 * G7_UNIT_BYTES + ((S) >> 3) * G7_UNIT_BYTES + ((((S) & 7) * 4 + (WAVE)) * 1024))
#define G7E_ROW(R) (((R) >> 3) * 4096 + ((R) & 7) * 128)
#define G7_TAB_OFF (4 * G7_UNIT_BYTES)
#define G7_ETAB_OFF (G7_TAB_OFF + 8192)

// cache policy of the output stores: 0 plain, 1 non-temporal (lines stay in the XCD's L2 either way), 2 sc1, 3 sc0 sc1
// (write-through: the line is NOT kept in L2 -- MI355X_MICROARCH.md, stores of each flavour)
#ifndef G7_ST_POLICY
#define G7_ST_POLICY 1
#endif
typedef unsigned int g7_u32x4 __attribute__((ext_vector_type(4)));
#if G7_ST_POLICY == 1
#define G7E_STORE16(P, V) __builtin_nontemporal_store(g7_u32x4{(V).x, (V).y, (V).z, (V).w}, (g7_u32x4*)(P))
#elif G7_ST_POLICY == 2
#define G7E_STORE16U(UB, VO, V) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(VO), "v"(g7_u32x4{(V).x, (V).y, (V).z, (V).w}), "s"(UB) : "memory")
#elif G7_ST_POLICY == 3
#define G7E_STORE16U(UB, VO, V) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(VO), "v"(g7_u32x4{(V).x, (V).y, (V).z, (V).w}), "s"(UB) : "memory")
#else
#define G7E_STORE16(P, V) (*(uint4*)(P) = (V))
#endif
#define G7_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define G7_FENCE_() __builtin_amdgcn_sched_barrier(0)

// work id -> tile, XCD aware: the 32 workgroups of an XCD (block b runs on XCD b % 8) take 32 CONSECUTIVE tiles of the
// grouped order (group_m row tiles sweeping the column tiles) in every round, so the panels they share stay in that L2.
__device__ __forceinline__ bool g7_tile(int it, int64_t ntm, int64_t ntn, int group_m, int64_t& m0, int64_t& n0) {
  // 32-bit arithmetic throughout (the launchers refuse ntm * ntn >= 2^31): the 64-bit divisions of the first version
  // were ~700 dependent scalar instructions per tile on the only wave of each SIMD
  const uint32_t tm = (uint32_t)ntm, tn = (uint32_t)ntn, gm = (uint32_t)group_m & 0xffffu;   // bit 16: reversed walk
  const uint32_t ntiles = tm * tn;
  uint32_t w;
  if ((gridDim.x & 7) == 0) {
    const uint32_t nslots = gridDim.x >> 3;
    w = ((uint32_t)it * 8 + (blockIdx.x & 7)) * nslots + (blockIdx.x >> 3);
  } else {
    w = (uint32_t)it * gridDim.x + blockIdx.x;
  }
  if (w >= ntiles) return false;
  if ((group_m >> 16) & 1) w = ntiles - 1 - w;
  const uint32_t per_group = gm * tn;
  const uint32_t g = w / per_group;
  const uint32_t first_m = g * gm;
  const uint32_t gsz = (tm - first_m) < gm ? (tm - first_m) : gm;
  const uint32_t in_g = w - g * per_group;
  const uint32_t col = in_g / gsz;
  m0 = (int64_t)(first_m + (in_g - col * gsz)) * 256;
  n0 = (int64_t)col * 256;
  return true;
}

// Probe (round 6, OM_GEMM_STAGGER): workgroups on alternate CUs of an XCD start late by a fraction of a tile period, so that the chip's
// epilogues (all HBM traffic, no MFMA) and K loops (the reverse) stop coinciding across CUs.  group_m bits 17-27: delay in units of
// 1 024 cycles per phase; bits 28-29: log2 of the number of phases.
__device__ __forceinline__ void g7_stagger(int group_m) {
  const int unit = (group_m >> 17) & 0x7ff;
  if (unit == 0) return;
  const int nph = 1 << ((group_m >> 28) & 3);
  const int n = ((blockIdx.x >> 3) & (nph - 1)) * unit;
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
}
static int g7_stagger_bits() {
  static const int v = [] {
    const char* e = getenv("OM_GEMM_STAGGER");
    const char* p = getenv("OM_GEMM_STAGGER_PH");
    const int unit = e ? atoi(e) & 0x7ff : 0, ph = p ? atoi(p) : 2;
    return (unit << 17) | ((ph >= 8 ? 3 : ph >= 4 ? 2 : ph >= 2 ? 1 : 0) << 28);
  }();
  return v;
}

// v summed over the four lanes {l, l ^ 16, l ^ 32, l ^ 48}: (own + lane ^ 16) + the same of the other half, in every lane -- v_permlane16_swap
// (odd 16-lane rows of one operand <-> even rows of the other) and v_permlane32_swap (upper half <-> lower half) with both operands = v
__device__ __forceinline__ float g7_quad_row_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}
// one 1 KiB table by LDS-DMA: lanes 0-31 fetch 512 B from `lo`, lanes 32-63 from `hi` (16 bytes per lane)
__device__ __forceinline__ void g7_table2(const float* lo, const float* hi, char* dst, int lane) {
  g7_dma_v((lane < 32 ? lo : hi) + (lane & 31) * 4, g7_lds_addr(dst));
}
// 1 KiB of contiguous floats
__device__ __forceinline__ void g7_table1(const float* src, char* dst, int lane) {
  g7_dma((const char*)src, lane * 16, g7_lds_addr(dst));
}

// Accumulator-init tables of tile (mc, nc) for this wave (TI DMA instructions: 2 with LNF == 1, else 1).  Absent
// vectors are fetched from a valid dummy and ignored by the reader.
template <int LNF>
__device__ __forceinline__ void g7_init_tables(const GemmEpilogue& ep, const void* dummy, char* smem, int64_t mc, int64_t nc,
                                               int wave, int lane) {
  char* tab = smem + G7_TAB_OFF + wave * 2048;
  const float* d = (const float*)dummy;
  const float* cs = (LNF == 1 && ep.ln_colsum) ? ep.ln_colsum + nc : d;
  const float* bs = ep.bias ? ep.bias + nc : d;
  g7_table2(cs, bs, tab, lane);
  if (LNF == 1) g7_table1(ep.ln_stats + mc * 2, tab + 1024, lane);
}

template <typename T, int ACT, bool RESID, int LNF, bool PERSIST>
__global__ __launch_bounds__(G6_THREADS) void gemm_nt_kernel7(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, T* C, int64_t ldc,
    int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  typedef T OutT;
  static_assert(sizeof(T) == 2, "16-bit in, 16-bit out");
  static_assert(LNF < 2 || RESID, "the output-side LayerNorm variants add a residual");
  constexpr bool LNO = LNF >= 2;               // output side: normalised residual + row statistics of the output
  constexpr bool TWO = LNF == 3;               // ... on a two-plane (hi + lo) residual stream
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntm = M / 256, ntn = N / 256;
  const int nk = (int)((K * 2) / G7_ROW_BYTES);
  const EpiScalars es(ep);
  constexpr int R = RESID ? (TWO ? 8 : 4) : 0; // DMA instructions per residual patch
  constexpr int TI = LNF == 1 ? 2 : 1;         // ... of the next tile's init tables
  constexpr int A2 = LNO ? 1 : 0;              // one row-statistics store behind every second patch
  constexpr int PF = 16;                       // ... of the next tile's first K step

  int it = 0;
  int64_t m0, n0;
  if (!g7_tile(0, ntm, ntn, group_m, m0, n0)) return;
  G7SrcU src;                                  // per-lane offsets once; only the two tile bases change
  g7_offsets_u<T>(src, lda, ldb, wave, lane0);
  src.a = (const char*)(A + m0 * lda);
  src.b = (const char*)(B + n0 * ldb);
  g7_init_tables<LNF>(ep, A, smem, m0 + wm * 128, n0 + wn * 128, wave, lane0);
  g7_fill_a(src, src.a, smem, wave);
  g7_fill_b(src, src.b, smem + G7_UNIT_BYTES, wave);
  bool pending = false;                        // the previous epilogue's 32 stores may still be in flight

  for (;;) {
    const int64_t mc = m0 + wm * 128, nc = n0 + wn * 128;
    unsigned long long* tr = nullptr;
    if (ep.trace) {
      const int64_t tile_id = (m0 / 256) * ntn + n0 / 256;
      if (tile_id < 8192) tr = ep.trace + tile_id * 32;
    }
    if (tr && threadIdx.x == 0) { tr[0] = clock64(); tr[30] = wall_clock64(); }
    // ---- tile start: K step 1 into U2, U3 (free once every wave has left the previous epilogue) --------------------
    // (U2 / U3: this wave's epilogue buffers were exactly the slices its own DMA instructions fill -- no barrier)
    if (pending) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nk > 1) {
      g7_fill_a(src, src.a + G7_ROW_BYTES, smem + 2 * G7_UNIT_BYTES, wave);
      g7_fill_b(src, src.b + G7_ROW_BYTES, smem + 3 * G7_UNIT_BYTES, wave);
    }
    // my tables (and K step 0) have landed: only K step 1 and the previous tile's stores are younger
    if (pending) { if (nk > 1) G7_WAIT_VM(48); else G7_WAIT_VM(32); }
    else { if (nk > 1) G7_WAIT_VM(16); else G7_WAIT_VM(0); }
    if (tr && threadIdx.x == 0) tr[1] = clock64();
    // ---- accumulators start at the bias, or at b'_n / rstd_m - mu_m s_n for a raw pre-LayerNorm A operand
    // (gemm_wide6.h) -- written by ONE extra MFMA sub-step instead of 256 vector moves: the rank-2 product
    //     u_m b_n + v_m s_n      (u = 1 or 1 / rstd_m, v = 0 or -mu_m)
    // with every factor split into bf16 hi + lo, k slots (u_hi b_hi, u_lo b_hi, u_hi b_lo, v_hi s_hi, v_lo s_hi,
    // v_hi s_lo): relative error 2^-16 of each term, far below the bf16 rounding of the output.  (Per-lane values of
    // this section and of the epilogue derive from an OPAQUE copy of the lane id: otherwise the compiler hoists
    // dozens of loop-invariant addresses out of the tile loop and spills them around the K loop.)
    f32x16_t acc[4][4];
    float rs[4] = {1.f, 1.f, 1.f, 1.f};
    {
      typedef typename MmaOps<T>::frag_t frag_t;
      int lane_i = lane0;
      asm volatile("" : "+v"(lane_i));
      const int l31 = lane_i & 31, half = lane_i >> 5;
      const char* tab = smem + G7_TAB_OFF + wave * 2048;
      const bool ln_in = LNF == 1 && ep.ln_stats != nullptr;
      const bool has_cs = LNF == 1 && ep.ln_colsum != nullptr, has_b = ep.bias != nullptr;
      auto split = [](float x, uint32_t& hi, uint32_t& lo) {
        hi = Half16<T>::bits(x); lo = Half16<T>::bits(x - Half16<T>::value(hi));
      };
      frag_t fa[4], fb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        float u = 1.f, v = 0.f;
        if (ln_in) {
          const float2 st = *(const float2*)(tab + 1024 + (mi * 32 + l31) * 8);
          const float mu = ep.ln_rms ? 0.f : st.x * ep.ln_inv_h;
          const float var = fmaxf(st.y * ep.ln_inv_h - mu * mu, 0.f) + ep.ln_eps;
          rs[mi] = rsqrtf(var);
          u = sqrtf(var); v = -mu;
        }
        uint32_t uh, ul, vh, vl;
        split(u, uh, ul); split(v, vh, vl);
        uint4 w = make_uint4(uh | (ul << 16), uh | (vh << 16), vl | (vh << 16), 0u);     // k: u_hi u_lo u_hi v_hi v_lo v_hi 0 0
        if (half) w = make_uint4(0u, 0u, 0u, 0u);
        asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z));       // opaque per row block: 16 MFMAs, not 4 + 192 accumulator moves (as kernels 7c / 7r)
        fa[mi] = __builtin_bit_cast(frag_t, w);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        float b = *(const float*)(tab + 512 + (ni * 32 + l31) * 4), sc = *(const float*)(tab + (ni * 32 + l31) * 4);
        if (!has_b) b = 0.f;
        if (!(ln_in && has_cs)) sc = 0.f;
        uint32_t bh, bl, sh, sl;
        split(b, bh, bl); split(sc, sh, sl);
        uint4 w = make_uint4(bh | (bh << 16), bl | (sh << 16), sh | (sl << 16), 0u);     // k: b_hi b_hi b_lo s_hi s_hi s_lo 0 0
        if (half) w = make_uint4(0u, 0u, 0u, 0u);
        fb[ni] = __builtin_bit_cast(frag_t, w);
      }
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[q >> 2][q & 3] = zero; MmaOps<T>::mma(fb[q & 3], fa[q >> 2], acc[q >> 2][q & 3]); }
    }
#ifdef G7_M16_PROBE      // tools/gemm7_probe.hip with -DG7_ABL=4: the K loop alone on 16 x 16 x 32 MFMAs
    f32x4_t acc16[8][8];
    {
      typedef typename MmaOps<T>::frag_t frag_t;
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
      const frag_t zf = __builtin_bit_cast(frag_t, z4);
      const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 64; ++q) { acc16[q >> 3][q & 7] = zero4; Mma16<T>::mma(zf, zf, acc16[q >> 3][q & 7]); }
    }
    gemm_mainloop7_run16<T, G7SrcU>(src, nk, smem, acc16, tr, pending);
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("" : "+a"(acc16[q >> 3][q & 7]));
#else
    gemm_mainloop7_run<T, G7SrcU>(src, nk, smem, acc, tr, pending);     // waits again (a no-op now), barrier, K loop, barrier
#endif
    if (tr && threadIdx.x == 0) tr[15] = clock64();

    // ---- next tile (a workgroup that has none re-fetches its own: the instruction stream stays fixed) -----------------
    ++it;
    int64_t m1 = m0, n1 = n0;
    const bool has_next = PERSIST && g7_tile(it, ntm, ntn, group_m, m1, n1);
    const char* const next_a = (const char*)(A + m1 * lda);
    const char* const next_b = (const char*)(B + n1 * ldb);

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, half = lane >> 5;
    size_t ldc2 = (size_t)ldc * sizeof(OutT), ldr2 = (size_t)ep.ldr * sizeof(OutT);
    asm volatile("" : "+s"(ldc2), "+s"(ldr2));
    const char* const rbase = RESID ? (const char*)((const OutT*)ep.resid + mc * ep.ldr + nc) : nullptr;   // wave-uniform
    // second plane of the residual; absent (layer 0 adds the one-plane embedding output): fetched from the first and scaled by 0
    const char* const rlo_base = (TWO && ep.resid_lo) ? (const char*)((const OutT*)ep.resid_lo + mc * ep.ldr + nc) : rbase;
    const float rlo_scale = (TWO && ep.resid_lo) ? 1.f : 0.f;
    uint32_t roff[4];                                                    // row (lane >> 3) of an 8-row group, swizzled source chunk
#pragma unroll
    for (int k = 0; k < 4; ++k) roff[k] = (uint32_t)((lane >> 3) * ldr2) + (((lane & 7) ^ ((4 * k + (lane >> 4)) & 7)) << 4);
    // staging patch: the wave's slices 12-15 of U2 / U3 (8 rows per 1 KiB slice, slices 4 KiB apart) -- or, two planes, a
    // contiguous 4 KiB in the upper half of U4
    char* const stage = TWO ? smem + G7_TAB_OFF + 16384 + wave * 4096 : smem + G7E_SLICE(12, wave);
#define G7E_SROW(R_) (TWO ? (R_) * 128 : G7E_ROW(R_))
#define G7E_SPASS (TWO ? 1024 : 4096)
    const char* const etab = smem + G7_ETAB_OFF + wave * 2048;
    const bool res_ln = LNO && ep.rln_stats != nullptr;
    // residual patch p = (mi, nh): 32 rows x 128 B, chunk position XOR ((row >> 1) & 7); 4 instructions of 8 rows per plane.
    // One plane: ring of three patches (slices 0-11).  Two planes: ring of two entries of 8 slices (hi 0-3, lo 4-7).
#define G7E_RES_SLICE(P_) (TWO ? ((P_) & 1) * 8 : ((P_) % G7E_RES_DEPTH) * 4)
#define G7E_RES_DMA(P_)                                                                                        \
  do {                                                                                                         \
    const size_t poff = (size_t)(((P_) >> 1) * 32) * ldr2 + ((P_) & 1) * 128;                                  \
    const uint32_t buf = g7_lds_addr(smem) + G7E_SLICE(G7E_RES_SLICE(P_), wave);                               \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) g7_dma(rbase + poff + (size_t)(8 * k) * ldr2, roff[k], buf + k * 4096); \
    if (TWO) {                                                                                                 \
      const uint32_t buf2 = g7_lds_addr(smem) + G7E_SLICE(G7E_RES_SLICE(P_) + 4, wave);                        \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) g7_dma(rlo_base + poff + (size_t)(8 * k) * ldr2, roff[k], buf2 + k * 4096); \
    }                                                                                                          \
  } while (0)
    if (LNO) {             // gamma | beta of my 128 columns, statistics of my 128 residual rows (dummies when not normalised)
      g7_table2(res_ln ? ep.rln_g + nc : (const float*)A, res_ln ? ep.rln_b + nc : (const float*)A, (char*)etab, lane);
      g7_table1(res_ln ? ep.rln_stats + mc * 2 : (const float*)A, (char*)etab + 1024, lane);
    }
    if (RESID && !(G7_ABL & 4)) { G7E_RES_DMA(0); G7E_RES_DMA(1); if (!TWO) G7E_RES_DMA(2); }
    g7_fill_a(src, next_a, smem, wave);                                 // K step 0 of the next tile: U0, U1
    g7_fill_b(src, next_b, smem + G7_UNIT_BYTES, wave);
    g7_init_tables<LNF>(ep, A, smem, m1 + wm * 128, n1 + wn * 128, wave, lane);

    G7_FENCE_();
    if (!(G7_ABL & 4)) {
    float ra[4] = {1.f, 1.f, 1.f, 1.f}, rc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x2_t ssum = {0.f, 0.f}, ssq = {0.f, 0.f};
    const int skey = l31 & 7;
    char* const st_wr = stage + G7E_SROW(l31) + 8 * half;
    const char* const st_rd = stage + (lane >> 3) * 128;               // read-back pass i4 covers rows 8 i4 .. 8 i4 + 7
    char* const cbase = (char*)(C + mc * ldc + nc);                     // wave-uniform; the per-lane part is 32 bits
    char* const cbase_lo = TWO ? (char*)((OutT*)ep.out_lo + mc * ldc + nc) : nullptr;
    const uint32_t coff = (uint32_t)((lane >> 3) * ldc2) + (lane & 7) * 16;
    // row statistics of this launch's output: slot (column tile, wave column) of ep.stats_out, [slots][M] pairs (sum, sum
    // of squares) written with PLAIN stores -- one writer per (slot, row); omk_ln_stats_reduce adds the slots in a fixed
    // order.  (Round 2 added into one pair per row with f32 atomics: the last bits depended on the arrival order.)
    float2* const stat_slot = LNO ? (float2*)ep.stats_out + ((n0 >> 8) * 2 + wn) * M : nullptr;
    uint2 plo[8];                                                       // two planes: the patch's remainder words until the hi plane is read back
    // WRITE(p): residual + conversion of patch p, staged as packed 16-bit (two planes: the hi plane; the lo words wait in plo)
#define G7E_WRITE(P_)                                                                                          \
  do {                                                                                                         \
    constexpr int MI = (P_) >> 1, NH = (P_) & 1;                                                               \
    /* the patch's two accumulator tiles stay in their AGPRs up to here: without the pin the compiler copies all   \
       256 accumulators into VGPRs behind the K loop and spills everything that lives across the epilogue */      \
    asm volatile("" : "+a"(acc[MI][NH * 2]), "+a"(acc[MI][NH * 2 + 1]));                                       \
    const int64_t m = mc + MI * 32 + l31;                                                                      \
    uint2 rpatch[2][4], rplo[2][4];                                                                            \
    if (RESID) {                                                                                               \
      const char* buf = smem + G7E_SLICE(G7E_RES_SLICE(P_), wave) + G7E_ROW(l31) + 8 * half;                   \
      _Pragma("unroll") for (int nl = 0; nl < 2; ++nl)                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
          rpatch[nl][j] = *(const uint2*)(buf + (((nl * 4 + j) ^ ((l31 >> 1) & 7)) << 4));                     \
      if (TWO) {                                                                                               \
        const char* buf2 = smem + G7E_SLICE(G7E_RES_SLICE(P_) + 4, wave) + G7E_ROW(l31) + 8 * half;            \
        _Pragma("unroll") for (int nl = 0; nl < 2; ++nl)                                                       \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
            rplo[nl][j] = *(const uint2*)(buf2 + (((nl * 4 + j) ^ ((l31 >> 1) & 7)) << 4));                    \
      }                                                                                                        \
    }                                                                                                          \
    if (!LNO) {                                                                                                \
    f32x8_t gq[4];        /* erf-GELU: the patch's 32 values, eight per polynomial evaluation (four chains in flight) */ \
    if (ACT == OM_ACT_GELU_ERF) {                                                                              \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                       \
        f32x8_t v8;                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) v8[e] = acc[MI][NH * 2 + (g_ >> 1)][8 * (g_ & 1) + e];    \
        if (LNF == 1) v8 *= rs[MI];                                                                            \
        gq[g_] = gelu_erf_poly8(v8);                                                                           \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int nl = 0; nl < 2; ++nl)                                                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
        const int ni = NH * 2 + nl;                                                                            \
        const int64_t n = nc + ni * 32 + 8 * j + 4 * half;                                                     \
        f32x2_t a_lo = {acc[MI][ni][4 * j], acc[MI][ni][4 * j + 1]}, a_hi = {acc[MI][ni][4 * j + 2], acc[MI][ni][4 * j + 3]}; \
        if (LNF == 1) { a_lo *= rs[MI]; a_hi *= rs[MI]; }                                                      \
        f32x2_t lo_, hi_;                                                                                      \
        if (ACT == OM_ACT_GELU_ERF) {                                                                          \
          const f32x8_t g8 = gq[nl * 2 + (j >> 1)];                                                            \
          lo_ = (f32x2_t){g8[4 * (j & 1)], g8[4 * (j & 1) + 1]}; hi_ = (f32x2_t){g8[4 * (j & 1) + 2], g8[4 * (j & 1) + 3]}; \
        } else {                                                                                               \
          lo_ = epi_pair<ACT, false, OutT>(a_lo, m, n, M, N, ep, es, 0, 0);                                    \
          hi_ = epi_pair<ACT, false, OutT>(a_hi, m, n + 2, M, N, ep, es, 0, 0);                                \
        }                                                                                                      \
        if (RESID) {                                                                                           \
          const uint2 rr = rpatch[nl][j];                                                                      \
          float r0 = Half16<OutT>::lo(rr.x), r1 = Half16<OutT>::hi(rr.x);                                      \
          float r2 = Half16<OutT>::lo(rr.y), r3 = Half16<OutT>::hi(rr.y);                                      \
          if (TWO) {                                                                                           \
            const uint2 rl = rplo[nl][j];                                                                      \
            r0 = fmaf(Half16<OutT>::lo(rl.x), rlo_scale, r0); r1 = fmaf(Half16<OutT>::hi(rl.x), rlo_scale, r1); \
            r2 = fmaf(Half16<OutT>::lo(rl.y), rlo_scale, r2); r3 = fmaf(Half16<OutT>::hi(rl.y), rlo_scale, r3); \
          }                                                                                                    \
          if (res_ln) {                                                                                        \
            const f32x4_t g4 = *(const f32x4_t*)(etab + (ni * 32 + 8 * j + 4 * half) * 4);                     \
            const f32x4_t b4 = *(const f32x4_t*)(etab + 512 + (ni * 32 + 8 * j + 4 * half) * 4);               \
            r0 = fmaf(fmaf(r0, ra[MI], rc[MI]), g4[0], b4[0]); r1 = fmaf(fmaf(r1, ra[MI], rc[MI]), g4[1], b4[1]); \
            r2 = fmaf(fmaf(r2, ra[MI], rc[MI]), g4[2], b4[2]); r3 = fmaf(fmaf(r3, ra[MI], rc[MI]), g4[3], b4[3]); \
          }                                                                                                    \
          if (!LNO && es.mul) { lo_[0] *= r0; lo_[1] *= r1; hi_[0] *= r2; hi_[1] *= r3; }   /* T5 gated act(.) * gate: never with LNO */ \
          else {                                                                                               \
            lo_[0] = epi_resid<ACT, sizeof(OutT) == 2>(lo_[0], r0, false); lo_[1] = epi_resid<ACT, sizeof(OutT) == 2>(lo_[1], r1, false);            \
            hi_[0] = epi_resid<ACT, sizeof(OutT) == 2>(hi_[0], r2, false); hi_[1] = epi_resid<ACT, sizeof(OutT) == 2>(hi_[1], r3, false);            \
          }                                                                                                    \
        }                                                                                                      \
        if (LNO) {                                                                                             \
          ssum += lo_ + hi_;                                                                                   \
          ssq = __builtin_elementwise_fma(lo_, lo_, __builtin_elementwise_fma(hi_, hi_, ssq));                 \
        }                                                                                                      \
        { const uint2 pk_ = make_uint2(Half16<OutT>::pack2(lo_[0], lo_[1]), Half16<OutT>::pack2(hi_[0], hi_[1]));       \
          if (TWO)          /* what the 16-bit word dropped, rounded once more: y = hi + lo to ~2^-17 */        \
            plo[nl * 4 + j] = make_uint2(Half16<OutT>::pack2(lo_[0] - Half16<OutT>::lo(pk_.x), lo_[1] - Half16<OutT>::hi(pk_.x)), \
                                         Half16<OutT>::pack2(hi_[0] - Half16<OutT>::lo(pk_.y), hi_[1] - Half16<OutT>::hi(pk_.y))); \
          if (G7_ABL & 2) asm volatile("" ::"v"(pk_)); else *(uint2*)(st_wr + (((nl * 4 + j) ^ skey) << 4)) = pk_; }  \
      }                                                                                                        \
    } else {                                                                                                   \
    /* Output-side LayerNorm variants: the same arithmetic on EIGHT values per step (columns 8 j .. of two adjacent j): \
       every step expands into four independent packed instructions, where the per-(nl, j) form left the compiler one \
       or two dependent chains at ~9 cycles per instruction (tools/gemm7_probe.hip: 12 us of a 25 us two-plane epilogue) */ \
    f32x8_t s8 = 0.f, q8 = 0.f;                                                                                \
    _Pragma("unroll") for (int nl = 0; nl < 2; ++nl)                                                           \
      _Pragma("unroll") for (int jp = 0; jp < 2; ++jp) {                                                       \
        const int ni = NH * 2 + nl;                                                                            \
        f32x8_t v8;                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) v8[e] = acc[MI][ni][8 * jp + e];                         \
        const uint2 ra_ = rpatch[nl][2 * jp], rb_ = rpatch[nl][2 * jp + 1];                                    \
        f32x8_t r8 = {Half16<OutT>::lo(ra_.x), Half16<OutT>::hi(ra_.x), Half16<OutT>::lo(ra_.y), Half16<OutT>::hi(ra_.y), \
                      Half16<OutT>::lo(rb_.x), Half16<OutT>::hi(rb_.x), Half16<OutT>::lo(rb_.y), Half16<OutT>::hi(rb_.y)}; \
        if (TWO) {                                                                                             \
          const uint2 la_ = rplo[nl][2 * jp], lb_ = rplo[nl][2 * jp + 1];                                      \
          const f32x8_t l8 = {Half16<OutT>::lo(la_.x), Half16<OutT>::hi(la_.x), Half16<OutT>::lo(la_.y), Half16<OutT>::hi(la_.y), \
                              Half16<OutT>::lo(lb_.x), Half16<OutT>::hi(lb_.x), Half16<OutT>::lo(lb_.y), Half16<OutT>::hi(lb_.y)}; \
          r8 = __builtin_elementwise_fma(l8, (f32x8_t)(rlo_scale), r8);                                        \
        }                                                                                                      \
        if (res_ln) {                                                                                          \
          const int c0 = ni * 32 + 16 * jp + 4 * half;              /* columns c0 .. c0 + 3 and c0 + 8 .. c0 + 11 */ \
          const f32x4_t ga = *(const f32x4_t*)(etab + c0 * 4), gb = *(const f32x4_t*)(etab + (c0 + 8) * 4);     \
          const f32x4_t ba = *(const f32x4_t*)(etab + 512 + c0 * 4), bb = *(const f32x4_t*)(etab + 512 + (c0 + 8) * 4); \
          const f32x8_t g8 = __builtin_shufflevector(ga, gb, 0, 1, 2, 3, 4, 5, 6, 7);                          \
          const f32x8_t b8 = __builtin_shufflevector(ba, bb, 0, 1, 2, 3, 4, 5, 6, 7);                          \
          r8 = __builtin_elementwise_fma(__builtin_elementwise_fma(r8, (f32x8_t)(ra[MI]), (f32x8_t)(rc[MI])), g8, b8); \
        }                                                                                                      \
        v8 += r8;                                                                                              \
        s8 += v8;                                                                                              \
        q8 = __builtin_elementwise_fma(v8, v8, q8);                                                            \
        const uint2 pa_ = make_uint2(Half16<OutT>::pack2(v8[0], v8[1]), Half16<OutT>::pack2(v8[2], v8[3]));    \
        const uint2 pb_ = make_uint2(Half16<OutT>::pack2(v8[4], v8[5]), Half16<OutT>::pack2(v8[6], v8[7]));    \
        if (TWO) {          /* what the 16-bit words dropped, rounded once more: y = hi + lo to ~2^-17 */       \
          const f32x8_t h8 = {Half16<OutT>::lo(pa_.x), Half16<OutT>::hi(pa_.x), Half16<OutT>::lo(pa_.y), Half16<OutT>::hi(pa_.y), \
                              Half16<OutT>::lo(pb_.x), Half16<OutT>::hi(pb_.x), Half16<OutT>::lo(pb_.y), Half16<OutT>::hi(pb_.y)}; \
          const f32x8_t d8 = v8 - h8;                                                                          \
          plo[nl * 4 + 2 * jp] = make_uint2(Half16<OutT>::pack2(d8[0], d8[1]), Half16<OutT>::pack2(d8[2], d8[3])); \
          plo[nl * 4 + 2 * jp + 1] = make_uint2(Half16<OutT>::pack2(d8[4], d8[5]), Half16<OutT>::pack2(d8[6], d8[7])); \
        }                                                                                                      \
        if (G7_ABL & 2) asm volatile("" ::"v"(pa_), "v"(pb_));                                                 \
        else {                                                                                                 \
          *(uint2*)(st_wr + (((nl * 4 + 2 * jp) ^ skey) << 4)) = pa_;                                          \
          *(uint2*)(st_wr + (((nl * 4 + 2 * jp + 1) ^ skey) << 4)) = pb_;                                      \
        }                                                                                                      \
      }                                                                                                        \
    ssum[0] += (s8[0] + s8[1]) + (s8[2] + s8[3]); ssum[1] += (s8[4] + s8[5]) + (s8[6] + s8[7]);                \
    ssq[0] += (q8[0] + q8[1]) + (q8[2] + q8[3]); ssq[1] += (q8[4] + q8[5]) + (q8[6] + q8[7]);                  \
    }                                                                                                          \
    if (LNO && NH == 1) {           /* both column halves of the row block done: this wave's partial sums of the row */ \
      float s1 = ssum[0] + ssum[1], s2 = ssq[0] + ssq[1];                                                      \
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);                                              \
      if (half == 0) stat_slot[m] = make_float2(s1, s2);                                                       \
      ssum = (f32x2_t){0.f, 0.f}; ssq = (f32x2_t){0.f, 0.f};                                                   \
    }                                                                                                          \
  } while (0)
    // two planes: the remainder words of the patch just converted go through the same staging patch, behind the read-back
    // of its hi plane (a wave's LDS operations execute in order)
#define G7E_WRITE_LO()                                                                                         \
  do {                                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) *(uint2*)(st_wr + ((c ^ skey) << 4)) = plo[c];                \
  } while (0)

    // vmcnt retires in order: each wait names (at most) the operations issued after the one it needs
    if (RESID || LNO) G7_WAIT_VM((TWO ? R : 2 * R) + PF + TI);          // residual patch 0 (and the epilogue tables)
    if (res_ln) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const float2 st = *(const float2*)(etab + 1024 + (mi * 32 + l31) * 8);
        const float mu = st.x * ep.ln_inv_h;
        const float rstd = rsqrtf(fmaxf(st.y * ep.ln_inv_h - mu * mu, 0.f) + ep.ln_eps);
        ra[mi] = rstd; rc[mi] = -mu * rstd;
      }
    }
    if (tr && threadIdx.x == 0) tr[16] = clock64();
#define G7E_RB(I4) ((G7_ABL & 2) ? make_uint4(0u, 0u, 0u, 0u) : *(const uint4*)(st_rd + (I4) * G7E_SPASS + (((lane & 7) ^ (((lane >> 3) + (I4) * 8) & 7)) << 4)))
#ifdef G7E_STORE16U      // wave-uniform base in scalar registers + 32-bit lane offset (what the compiler selects for the builtin forms)
#define G7E_ST_(BASE, PP, I4, V) do { if (G7_ABL & 1) asm volatile("" ::"v"((V).x), "v"((V).y), "v"((V).z), "v"((V).w)); else G7E_STORE16U((BASE) + (size_t)(((PP) >> 1) * 32 + (I4) * 8) * ldc2 + ((PP) & 1) * 128, coff, V); } while (0)
#else
#define G7E_ST_(BASE, PP, I4, V) do { if (G7_ABL & 1) asm volatile("" ::"v"((V).x), "v"((V).y), "v"((V).z), "v"((V).w)); else G7E_STORE16((BASE) + (size_t)(((PP) >> 1) * 32 + (I4) * 8) * ldc2 + ((PP) & 1) * 128 + coff, V); } while (0)
#endif
#define G7E_ST(PP, I4, V) G7E_ST_(cbase, PP, I4, V)
    if (!TWO) {
    // Software pipeline over the 8 patches: WRITE(p+1) -> read-back of p+1 ISSUED at once (its data is only needed one
    // iteration later, behind the next patch's conversion work) -> stores of patch p from the registers read one
    // iteration ago.  A wave's LDS operations execute in order, so the single staging buffer needs no waits.
    G7E_WRITE(0);
    G7_FENCE_();
    uint4 sa0 = G7E_RB(0), sa1 = G7E_RB(1), sa2 = G7E_RB(2), sa3 = G7E_RB(3), sb0, sb1, sb2, sb3;
    G7_FENCE_();
    // CUR / NXT: the register sets holding patch P_ (read last iteration) and patch P_ + 1 (read now)
#define G7E_ITER(P_, YWAIT, C0, C1, C2, C3, N0, N1, N2, N3)                                                    \
  do {                                                                                                         \
    if (tr && threadIdx.x == 0) tr[17 + (P_)] = clock64();                                                     \
    if ((P_) + 1 < 8) {                                                                                        \
      if (RESID) {                                                                                             \
        if ((P_) + 3 < 8) G7E_RES_DMA((P_) + 3);                                                               \
        G7_WAIT_VM(YWAIT);                                                                                     \
      }                                                                                                        \
      G7E_WRITE((P_) + 1);                                                                                     \
      G7_FENCE_();                                                                                             \
      N0 = G7E_RB(0); N1 = G7E_RB(1); N2 = G7E_RB(2); N3 = G7E_RB(3);                                          \
    }                                                                                                          \
    G7_FENCE_();                                                                                               \
    G7E_ST(P_, 0, C0); G7E_ST(P_, 1, C1); G7E_ST(P_, 2, C2); G7E_ST(P_, 3, C3);                                 \
    G7_FENCE_();                                                                                               \
  } while (0)
    G7E_ITER(0, 2 * R + PF + TI, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
    G7E_ITER(1, 2 * R + PF + TI + A2 + 4, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
    G7E_ITER(2, 2 * R + 8 + A2, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
    G7E_ITER(3, 2 * R + 8 + A2, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
    G7E_ITER(4, 2 * R + 8 + A2, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
    G7E_ITER(5, R + 8 + A2, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
    G7E_ITER(6, 8 + A2, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
    G7E_ITER(7, 0, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
#undef G7E_ITER
    } else {
    // Two planes: the same pipeline with a ring of two residual entries (patch p + 2 is fetched into the entry patch p
    // was read from one iteration ago) and eight stores per patch.  Per iteration: WRITE(p + 1) (hi staged, lo words in
    // registers) -> read back hi -> stage lo -> read back lo -> stores of patch p.  Operations issued after
    // RES_DMA(q) when iteration p = q - 1 waits for it:  [p = 0] K step 0 of the next tile, its tables, RES_DMA(2);
    // [p >= 1] the statistics store behind an odd patch, 8 stores, RES_DMA(q + 1) if any.
    G7E_WRITE(0);
    G7_FENCE_();
    uint4 ha0 = G7E_RB(0), ha1 = G7E_RB(1), ha2 = G7E_RB(2), ha3 = G7E_RB(3), hb0, hb1, hb2, hb3;
    G7_FENCE_();
    G7E_WRITE_LO();
    G7_FENCE_();
    uint4 la0 = G7E_RB(0), la1 = G7E_RB(1), la2 = G7E_RB(2), la3 = G7E_RB(3), lb0, lb1, lb2, lb3;
    G7_FENCE_();
#define G7E_ITER2(P_, YWAIT, CH0, CH1, CH2, CH3, CL0, CL1, CL2, CL3, NH0, NH1, NH2, NH3, NL0, NL1, NL2, NL3)      \
  do {                                                                                                         \
    if ((P_) + 1 < 8) {                                                                                        \
      if ((P_) + 2 < 8) G7E_RES_DMA((P_) + 2);                                                                 \
      G7_WAIT_VM(YWAIT);                                                                                       \
      G7E_WRITE((P_) + 1);                                                                                     \
      G7_FENCE_();                                                                                             \
      NH0 = G7E_RB(0); NH1 = G7E_RB(1); NH2 = G7E_RB(2); NH3 = G7E_RB(3);                                      \
      G7_FENCE_();                                                                                             \
      G7E_WRITE_LO();                                                                                          \
      G7_FENCE_();                                                                                             \
      NL0 = G7E_RB(0); NL1 = G7E_RB(1); NL2 = G7E_RB(2); NL3 = G7E_RB(3);                                      \
    }                                                                                                          \
    G7_FENCE_();                                                                                               \
    G7E_ST(P_, 0, CH0); G7E_ST(P_, 1, CH1); G7E_ST(P_, 2, CH2); G7E_ST(P_, 3, CH3);                             \
    G7E_ST_(cbase_lo, P_, 0, CL0); G7E_ST_(cbase_lo, P_, 1, CL1); G7E_ST_(cbase_lo, P_, 2, CL2); G7E_ST_(cbase_lo, P_, 3, CL3); \
    G7_FENCE_();                                                                                               \
  } while (0)
#define G7E_ITER2_(...) G7E_ITER2(__VA_ARGS__)
#define G7E_A ha0, ha1, ha2, ha3, la0, la1, la2, la3
#define G7E_B hb0, hb1, hb2, hb3, lb0, lb1, lb2, lb3
    G7E_ITER2_(0, PF + TI + R, G7E_A, G7E_B);
    G7E_ITER2_(1, 16 + A2, G7E_B, G7E_A);
    G7E_ITER2_(2, 16, G7E_A, G7E_B);
    G7E_ITER2_(3, 16 + A2, G7E_B, G7E_A);
    G7E_ITER2_(4, 16, G7E_A, G7E_B);
    G7E_ITER2_(5, 16 + A2, G7E_B, G7E_A);
    G7E_ITER2_(6, 8, G7E_A, G7E_B);
    G7E_ITER2_(7, 0, G7E_B, G7E_A);
#undef G7E_A
#undef G7E_B
#undef G7E_ITER2
#undef G7E_ITER2_
    }
#undef G7E_RB
#undef G7E_ST
#undef G7E_ST_
#undef G7E_WRITE
#undef G7E_WRITE_LO
    } else {
#ifndef G7_M16_PROBE
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+a"(acc[q >> 2][q & 3]));
#endif
    }
#undef G7E_RES_DMA
#undef G7E_RES_SLICE
#undef G7E_SROW
#undef G7E_SPASS
    if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; tr[31] = wall_clock64(); }
    if (!has_next) break;
    src.a = next_a; src.b = next_b; m0 = m1; n0 = n1;
    pending = !(G7_ABL & 5);        // (probe builds without stores: the tile-start waits count no stores)
  }
  G7_WAIT_VM(0);      // the last (dummy) prefetch must not outlive the workgroup's LDS allocation
}

// =========================================================================================================================
// Generation 7 on the CONTINUOUS ring (round 4; gemm_core7.h: gemm_mainloop7_cont) -- the variants without a residual:
// the encoder's QKV and FFN1 contractions, 42 of the 54 tiles a CU computes per layer.
//
// What the restart-per-tile kernel above pays between two K loops (tile traces, profiles/r04_probe1_*): the epilogue
// (7.8 k cycles plain, 14.9 k with GELU), then 1.3-1.9 k cycles issuing the 16 DMA instructions of K step 1 while the
// stores drain, then 2.5 k cycles building the accumulator-initialising fragments -- 12-19 k cycles per 30 k-cycle K loop,
// all of it at the same cycle cost whether 8 or 256 CUs run.  Here:
//   * the ring never stops: when the K loop of a tile ends, steps 0 and 1 of the next tile are landed / in flight, so the
//     tile boundary issues no operand DMA at all;
//   * the epilogue lives in the spare unit (this wave's own 1 KiB slices: staging patch 0-3, tables 4-5) -- no barrier on
//     either side of it;
//   * the next tile's tables are fetched at the start of the epilogue, its initialising fragments are built between the
//     sixth and the seventh patch (VALU work under the store-bound part) and the 16 initialising MFMAs are issued behind the
//     last conversion, under the last patch's stores: the next K loop starts at the epilogue's last store;
//   * the output stores of patch p are spread over the four quarters of patch p + 1's conversion (one store behind each
//     quarter) instead of four back to back: with GELU the polynomial and the store path now overlap.
// One loop body  [epilogue of the previous tile + initialisation of this one][K loop]  with two wave-uniform flags: `live`
// (false on the first pass only: nothing to store) and `have` (false on the last: nothing to initialise).
template <int LNF>
__device__ __forceinline__ void g7c_tables(const GemmEpilogue& ep, const void* dummy, char* tab0, char* tab1, int64_t mc, int64_t nc,
                                           int lane) {
  const float* d = (const float*)dummy;
  const float* cs = (LNF == 1 && ep.ln_colsum) ? ep.ln_colsum + nc : d;
  const float* bs = ep.bias ? ep.bias + nc : d;
  g7_table2(cs, bs, tab0, lane);                                   // s_n | b_n of my 128 columns
  if (LNF == 1) g7_table1(ep.ln_stats + mc * 2, tab1, lane);       // (sum, sum of squares) of my 128 rows
}

// =========================================================================================================================
// Kernel 7c on 16 x 16 x 32 MFMAs (round 4, late): the same tile boundary, the K loop of gemm_mainloop7_cont16.
//   acc[ti][fj][r] = C[m0 + wm*128 + ti*16 + (lane&15)][n0 + wn*128 + fj*16 + 4*(lane>>4) + r]
// A lane owns output row (lane & 15) of each 16-row block and four consecutive columns of each 16-column block.  Patch
// P = (mi, nh) is still 32 rows x 64 columns (two row blocks x four column blocks: 32 values per lane); quarter G of it is
// row block ti2 = G >> 1 and the column-block pair 2 (G & 1) + {0, 1}.  Staging and read-back as above: row = 128 B, 16-byte
// chunk XOR (row & 7); the lane's four columns are 8 bytes at chunk 2 fjl + (lane >> 5), half (lane >> 4) & 1.
// The accumulator-initialising rank-2 product needs one MFMA per 16 x 16 tile (64, under the last patch's stores).
//
// TRAIN (round 5): the training forward's FFN1 (train.hip) -- erf-GELU whose backward factor gelu'(v) goes to the tape
// (OM_ACT_PRE_GRAD) -- writes TWO tensors per tile: C = gelu(v) and ep.pre_act = gelu'(v) (ldp == ldc).  Generation 6 wrote the
// second one as 4-byte scattered stores from the accumulator layout (32 rows per instruction: 120 us per launch at 9 216 x 3072 x 768
// against ~60 for the contraction itself, profiles/r04_train_kernel_stats_v0.csv).  Here one evaluation of the polynomial yields both
// values; the staging patch is used twice per patch -- "virtual patch" 2 P is gelu of patch P, 2 P + 1 its gelu' (kept packed in 16
// registers from the conversion until its turn; a wave's LDS operations execute in order) -- and both leave as whole 128-byte lines.
template <typename T, int ACT, int LNF, bool TRAIN = false>
__global__ __launch_bounds__(G6_THREADS) void gemm_nt_kernel7c16(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, T* C, int64_t ldc,
    int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  typedef T OutT;
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "16-bit in, 16-bit out");
  static_assert(LNF == 0 || LNF == 1, "no residual: plain or LayerNorm-folded A operand");
  static_assert(!TRAIN || (ACT == OM_ACT_GELU_ERF && LNF == 0), "two-output form: erf-GELU + gelu' only");
  constexpr int NVP = TRAIN ? 16 : 8;          // virtual patches per wave tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntm = M / 256, ntn = N / 256;
  const int nk = (int)((K * 2) / G7_ROW_BYTES);
  const EpiScalars es(ep);

  int it = 0;
  int64_t m0, n0;
  if (!g7_tile(0, ntm, ntn, group_m, m0, n0)) return;
  G7SrcU src;
  g7_offsets_u<T>(src, lda, ldb, wave, lane0);
  G7Ring ring;
  g7_ring_reset(ring);
  const char* cur_a = (const char*)(A + m0 * lda);
  const char* cur_b = (const char*)(B + n0 * ldb);
  g7_fill_a(src, cur_a, smem + ring.ac, wave);
  g7_fill_b(src, cur_b, smem + ring.bc, wave);
  g7_fill_a(src, cur_a + G7_ROW_BYTES, smem + ring.an, wave);
  g7_fill_b(src, cur_b + G7_ROW_BYTES, smem + ring.bn, wave);
  g7_stagger(group_m);

  bool live = false, have = true;
  int64_t pm = m0, pn = n0;
  f32x4_t acc[8][8];
  float rs[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  unsigned long long* tr_prev = nullptr;

  for (;;) {
    char* const sp = smem + ring.sp + wave * 1024;         // this wave's slices of the spare unit: slice i at sp + i * 4096
    char* const tab0 = sp + 4 * 4096;
    char* const tab1 = sp + 5 * 4096;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    if (have) g7c_tables<LNF>(ep, A, tab0, tab1, m0 + wm * 128, n0 + wn * 128, lane);
    const int l15 = lane & 15, q4 = lane >> 4;
    float rsn[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    frag_t fa[8], fb[8];
    size_t ldc2 = (size_t)ldc * sizeof(OutT);
    asm volatile("" : "+s"(ldc2));
    const int64_t pmc = pm + wm * 128, pnc = pn + wn * 128;
    const char* const st_rd = sp + (lane >> 3) * 128;
    char* const cbase = (char*)(C + pmc * ldc + pnc);
    char* const pbase = TRAIN ? (char*)((OutT*)ep.pre_act + pmc * ldc + pnc) : cbase;      // second output (ldp == ldc: the launcher checks)
    const uint32_t coff = (uint32_t)((lane >> 3) * ldc2) + (lane & 7) * 16;
    uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    uint32_t gk[16];                                       // TRAIN: the patch's packed gelu' values, quarter G at gk[4 G .. 4 G + 3]

    // quarter G_ of patch P_: convert (and, TRAIN, keep gelu') and stage the first output
#define G7C_WRITE_Q(P_, G_)                                                                                    \
  do {                                                                                                         \
    constexpr int MI = (P_) >> 1, NH = (P_) & 1, TI = 2 * MI + ((G_) >> 1), FJ0 = 4 * NH + 2 * ((G_) & 1);     \
    if ((G_) == 0) { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) asm volatile("" : "+a"(acc[2 * MI + (u_ >> 2)][4 * NH + (u_ & 3)])); } \
    const int rr = ((G_) >> 1) * 16 + l15;                           /* row of the 32-row patch */             \
    const int64_t m = pmc + MI * 32 + rr;                                                                      \
    f32x8_t v8, g8;                                                                                            \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) v8[e] = acc[TI][FJ0 + (e >> 2)][e & 3];                      \
    if (LNF == 1) v8 *= rs[TI];                                                                                \
    if (ACT == OM_ACT_GELU_ERF) { if (TRAIN) v8 = gelu_erf_poly8_both(v8, g8); else v8 = gelu_erf_poly8(v8); } \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                         \
      f32x2_t lo_ = {v8[4 * jj], v8[4 * jj + 1]}, hi_ = {v8[4 * jj + 2], v8[4 * jj + 3]};                      \
      if (ACT != OM_ACT_GELU_ERF) {                                                                            \
        const int64_t n = pnc + (FJ0 + jj) * 16 + 4 * q4;                                                      \
        lo_ = epi_pair<ACT, false, OutT>(lo_, m, n, M, N, ep, es, 0, 0);                                       \
        hi_ = epi_pair<ACT, false, OutT>(hi_, m, n + 2, M, N, ep, es, 0, 0);                                   \
      }                                                                                                        \
      const uint2 pk_ = make_uint2(Half16<OutT>::pack2(lo_[0], lo_[1]), Half16<OutT>::pack2(hi_[0], hi_[1]));  \
      if (TRAIN) {                                                                                             \
        gk[4 * (G_) + 2 * jj] = Half16<OutT>::pack2(g8[4 * jj], g8[4 * jj + 1]);                               \
        gk[4 * (G_) + 2 * jj + 1] = Half16<OutT>::pack2(g8[4 * jj + 2], g8[4 * jj + 3]);                       \
      }                                                                                                        \
      const int c_ = (2 * ((G_) & 1) + jj) * 2 + (q4 >> 1);          /* 16-byte chunk of the patch row */       \
      *(uint2*)(sp + G7E_ROW(rr) + ((c_ ^ (rr & 7)) << 4) + 8 * (q4 & 1)) = pk_;                               \
    }                                                                                                          \
  } while (0)
    // TRAIN: quarter G_ of the kept gelu' values into the staging patch (second use of it for this patch)
#define G7C_STAGE_GK(G_)                                                                                       \
  do {                                                                                                         \
    const int rr = ((G_) >> 1) * 16 + l15;                                                                     \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                         \
      const int c_ = (2 * ((G_) & 1) + jj) * 2 + (q4 >> 1);                                                    \
      *(uint2*)(sp + G7E_ROW(rr) + ((c_ ^ (rr & 7)) << 4) + 8 * (q4 & 1)) = make_uint2(gk[4 * (G_) + 2 * jj], gk[4 * (G_) + 2 * jj + 1]); \
    }                                                                                                          \
  } while (0)
    // virtual patch VP_: patch VP_ itself, or (TRAIN) output VP_ & 1 of patch VP_ >> 1
#define G7C_WRITE_V(VP_, G_)                                                                                   \
  do {                                                                                                         \
    if constexpr (!TRAIN) G7C_WRITE_Q(((VP_) & 7), G_);                                                        \
    else if constexpr (((VP_) & 1) == 0) G7C_WRITE_Q((((VP_) >> 1) & 7), G_);                                  \
    else G7C_STAGE_GK(G_);                                                                                     \
  } while (0)
#define G7C_RB(I4) (*(const uint4*)(st_rd + (I4) * 4096 + (((lane & 7) ^ (((lane >> 3) + (I4) * 8) & 7)) << 4)))
#ifdef G7E_STORE16U
#define G7C_STB(BASE, PP, I4, V) G7E_STORE16U((BASE) + (size_t)(((PP) >> 1) * 32 + (I4) * 8) * ldc2 + ((PP) & 1) * 128, coff, V)
#else
#define G7C_STB(BASE, PP, I4, V) G7E_STORE16((BASE) + (size_t)(((PP) >> 1) * 32 + (I4) * 8) * ldc2 + ((PP) & 1) * 128 + coff, V)
#endif
#define G7C_ST(VP_, I4, V)                                                                                     \
  do {                                                                                                         \
    if constexpr (!TRAIN) G7C_STB(cbase, VP_, I4, V);                                                          \
    else if constexpr (((VP_) & 1) == 0) G7C_STB(cbase, (VP_) >> 1, I4, V);                                    \
    else G7C_STB(pbase, (VP_) >> 1, I4, V);                                                                    \
  } while (0)
#define G7C_ITER(P_, C0, C1, C2, C3, N0, N1, N2, N3)                                                           \
  do {                                                                                                         \
    if constexpr ((P_) < 8) { if (tr_prev && threadIdx.x == 0) tr_prev[17 + (P_)] = clock64(); }               \
    if constexpr ((P_) + 1 < NVP) {                                                                            \
      G7C_WRITE_V((P_) + 1, 0); G7_FENCE_(); G7C_ST(P_, 0, C0); G7_FENCE_();                                   \
      G7C_WRITE_V((P_) + 1, 1); G7_FENCE_(); G7C_ST(P_, 1, C1); G7_FENCE_();                                   \
      G7C_WRITE_V((P_) + 1, 2); G7_FENCE_(); G7C_ST(P_, 2, C2); G7_FENCE_();                                   \
      G7C_WRITE_V((P_) + 1, 3); G7_FENCE_();                                                                   \
      N0 = G7C_RB(0); N1 = G7C_RB(1); N2 = G7C_RB(2); N3 = G7C_RB(3);                                          \
      G7_FENCE_(); G7C_ST(P_, 3, C3); G7_FENCE_();                                                             \
    } else {                                                                                                   \
      G7C_ST(P_, 0, C0); G7C_ST(P_, 1, C1); G7C_ST(P_, 2, C2); G7C_ST(P_, 3, C3); G7_FENCE_();                 \
    }                                                                                                          \
  } while (0)
#define G7C_SA sa0, sa1, sa2, sa3
#define G7C_SB sb0, sb1, sb2, sb3
#define G7C_ITER_(...) G7C_ITER(__VA_ARGS__)

    if (live) {
      if (tr_prev && threadIdx.x == 0) tr_prev[16] = clock64();
      G7C_WRITE_V(0, 0); G7C_WRITE_V(0, 1); G7C_WRITE_V(0, 2); G7C_WRITE_V(0, 3);
      G7_FENCE_();
      sa0 = G7C_RB(0); sa1 = G7C_RB(1); sa2 = G7C_RB(2); sa3 = G7C_RB(3);
      G7_FENCE_();
      G7C_ITER_(0, G7C_SA, G7C_SB);
      G7C_ITER_(1, G7C_SB, G7C_SA);
      G7C_ITER_(2, G7C_SA, G7C_SB);
      G7C_ITER_(3, G7C_SB, G7C_SA);
      G7C_ITER_(4, G7C_SA, G7C_SB);
      G7C_ITER_(5, G7C_SB, G7C_SA);
      if constexpr (TRAIN) {
        G7C_ITER_(6, G7C_SA, G7C_SB);
        G7C_ITER_(7, G7C_SB, G7C_SA);
        G7C_ITER_(8, G7C_SA, G7C_SB);
        G7C_ITER_(9, G7C_SB, G7C_SA);
        G7C_ITER_(10, G7C_SA, G7C_SB);
        G7C_ITER_(11, G7C_SB, G7C_SA);
        G7C_ITER_(12, G7C_SA, G7C_SB);
      }
    }
    // ---- the next tile's initialising fragments: u_m b_n + v_m s_n as ONE 16 x 16 x 32 MFMA per tile, factors split into
    // 16-bit hi + lo in k slots 0-5 of k block 0 (lanes 0-15); the other k blocks are zero
    if (have) {
      // The next tile's tables (fetched at the top of this epilogue) have landed once nothing OLDER than the stores issued since is
      // outstanding: 6 (TRAIN: 13) patch iterations of four stores.  Rounds 4-5 waited vmcnt(8) here -- for the ACKNOWLEDGEMENT of all but
      // the last two patches' stores, 2.5 k of the 47.6 k cycles of an FFN1 tile (tools/epilogue_trace.py: the iteration in front of this
      // block took 5.0 k cycles where its neighbours take 1.5-1.6 k).
      if (live) { if constexpr (TRAIN) G7_WAIT_VM(52); else G7_WAIT_VM(24); } else G7_WAIT_VM(0);
      const bool ln_in = LNF == 1 && ep.ln_stats != nullptr;
      const bool has_cs = LNF == 1 && ep.ln_colsum != nullptr, has_b = ep.bias != nullptr;
      auto split = [](float x, uint32_t& hi, uint32_t& lo) {
        hi = Half16<T>::bits(x); lo = Half16<T>::bits(x - Half16<T>::value(hi));
      };
      // Round 6: the eight row blocks' statistics are read first and their chains kept free of branches and of libm's correctly rounded
      // sqrtf (u = sqrt(var) only feeds a 16-bit hi + lo split: var * rsq(var) is exact enough, and var >= eps is never subnormal, so the
      // bare v_rsq_f32 is what rsqrtf would return) -- the form of rounds 4-5 was eight DEPENDENT chains of ~40 instructions behind one
      // LDS read each, a branch between them: 3.4 k cycles per tile on the only wave of the SIMD (tools/epilogue_trace.py, FFN1: the
      // iteration in front of this block 4.9-5.0 k cycles, its neighbours 1.5-1.6 k).
      float uu[8], vv[8];
      {
        float2 st[8];
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {       // unconditional reads (eight in flight, one wait): without statistics the values are never selected
          if constexpr (LNF == 1) st[ti] = *(const float2*)(tab1 + (ti * 16 + l15) * 8); else st[ti] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
          const float mu = ep.ln_rms ? 0.f : st[ti].x * ep.ln_inv_h;
          const float var = fmaxf(st[ti].y * ep.ln_inv_h - mu * mu, 0.f) + ep.ln_eps;
          const float rstd = __builtin_amdgcn_rsqf(var);
          rsn[ti] = ln_in ? rstd : 1.f;
          uu[ti] = ln_in ? var * rstd : 1.f;
          vv[ti] = ln_in ? -mu : 0.f;
        }
      }
#pragma unroll
      for (int ti = 0; ti < 8; ++ti) {
        uint32_t uh, ul, vh, vl;
        split(uu[ti], uh, ul); split(vv[ti], vh, vl);
        uint4 w = make_uint4(uh | (ul << 16), uh | (vh << 16), vl | (vh << 16), 0u);     // k: u_hi u_lo u_hi v_hi v_lo v_hi 0 0
        if (q4) w = make_uint4(0u, 0u, 0u, 0u);
        asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z));       // opaque per row block: one MFMA per tile, no accumulator copies
        fa[ti] = __builtin_bit_cast(frag_t, w);
      }
      float bb[8], ss[8];
#pragma unroll
      for (int fj = 0; fj < 8; ++fj) { bb[fj] = *(const float*)(tab0 + 512 + (fj * 16 + l15) * 4); ss[fj] = *(const float*)(tab0 + (fj * 16 + l15) * 4); }
#pragma unroll
      for (int fj = 0; fj < 8; ++fj) {
        float b = bb[fj], sc = ss[fj];
        if (!has_b) b = 0.f;
        if (!(ln_in && has_cs)) sc = 0.f;
        uint32_t bh, bl, sh, sl;
        split(b, bh, bl); split(sc, sh, sl);
        uint4 w = make_uint4(bh | (bh << 16), bl | (sh << 16), sh | (sl << 16), 0u);     // k: b_hi b_hi b_lo s_hi s_hi s_lo 0 0
        if (q4) w = make_uint4(0u, 0u, 0u, 0u);
        fb[fj] = __builtin_bit_cast(frag_t, w);
      }
    }
    if (live) {                                  // converts the last patch: the last reader of the accumulators
      if constexpr (TRAIN) G7C_ITER_(13, G7C_SB, G7C_SA); else G7C_ITER_(6, G7C_SA, G7C_SB);
    }
    if (have) {
      const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 64; ++q) { acc[q >> 3][q & 7] = zero; Mma16c<T>::mma(fb[q & 7], fa[q >> 3], acc[q >> 3][q & 7]); }
    }
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("" : "+a"(acc[q >> 3][q & 7]));
    if (live) {
      if constexpr (TRAIN) { G7C_ITER_(14, G7C_SA, G7C_SB); G7C_ITER_(15, G7C_SB, G7C_SA); }
      else G7C_ITER_(7, G7C_SB, G7C_SA);
      if (tr_prev && threadIdx.x == 0) { tr_prev[28] = clock64(); tr_prev[29] = blockIdx.x; tr_prev[31] = wall_clock64(); }
    }
#undef G7C_ITER_
#undef G7C_SB
#undef G7C_SA
#undef G7C_ITER
#undef G7C_ST
#undef G7C_STB
#undef G7C_RB
#undef G7C_WRITE_V
#undef G7C_STAGE_GK
#undef G7C_WRITE_Q
    if (!have) break;
    if (!live) __builtin_amdgcn_s_barrier();     // first pass: K step 0 (waited for above) is published to the other waves
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) rs[ti] = rsn[ti];

    ++it;
    int64_t m1 = m0, n1 = n0;
    const bool has_next = g7_tile(it, ntm, ntn, group_m, m1, n1);
    const char* const next_a = (const char*)(A + m1 * lda);
    const char* const next_b = (const char*)(B + n1 * ldb);
    unsigned long long* tr = nullptr;
    if (ep.trace) {
      const int64_t tile_id = (m0 / 256) * ntn + n0 / 256;
      if (tile_id < 8192) tr = ep.trace + tile_id * 32;
    }
    if (tr && threadIdx.x == 0) { tr[0] = tr[1] = clock64(); tr[30] = wall_clock64(); }
    gemm_mainloop7_cont16<T>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc, tr);
    if (tr && threadIdx.x == 0) tr[15] = clock64();
    tr_prev = tr;
    pm = m0; pn = n0; live = true;
    cur_a = next_a; cur_b = next_b; m0 = m1; n0 = n1; have = has_next;
  }
  G7_WAIT_VM(0);
}

// =========================================================================================================================
// Generation 7 on the continuous ring, the variants WITH a residual (round 4): out-proj and FFN2 of the encoder -- one-plane
// residual stream (LNF 0: plain residual; LNF 2: normalised residual + row statistics of the output).
//
// Per-patch stamps of the restart kernel's residual epilogue (profiles/r04_probe3_residual_epilogue_per_patch.log): 7.6 k of
// its 21 k cycles pass before the first patch is converted -- 29 DMA issues (residual ring, next tile's K step 0, tables)
// and the HBM latency of the first residual patch behind them.  Here the LAST K step of a tile issues the epilogue's
// tables and its first two residual patches in the twelve slots that would fetch A(nk + 1) / B(nk + 1) (gemm_core7.h TAIL):
// they land under that step's MFMAs, in the units the step frees.  Across the boundary the ring keeps step 0 of the next
// tile only; A(1) and the first half of B(1) are issued behind the last conversion of the epilogue, and the accumulator
// initialisation of the next tile runs under the last patch's stores as in kernel 7c.
//   after the K loop (ring rotated):   ring.an (the spare of the last step)   slices 0-3 residual slot 0, 4 gamma | beta,
//                                                                              5 statistics of the residual rows, 6 next tile's s_n | b_n
//                                      ring.bn (the unit A(nk - 1) left)      slices 0-3 residual slot 1, 4-7 slot 2
//                                      ring.sp (the unit B(nk - 1) left)      slices 0-3 staging patch
// (slice i of a unit = this wave's i-th own KiB, at unit + (4 i + wave) KiB: no wave touches another wave's slices.)
// =========================================================================================================================
// LNF == 4, float16 only (round 6, second step; OPT-IN: OM_ENCODER_TWO_PLANE bit 2): the second plane in EIGHT bits.  The first version (two 16-bit planes, LNF == 3)
// paid 44 k cycles per tile epilogue against 24 k with one plane (tools/epilogue_trace.py, profiles/r06_epilogue_trace_*.json): the
// epilogue's time is its count of 1 KiB vector-memory operations (~300 cycles each per wave with four waves issuing), and two planes
// double them.  The remainder y - hi is at most half an ulp of hi, so e5m2 of (remainder * 2^10) carries y to ~2^-14 |y| -- eight times
// finer than one float16 plane, where the reference's autocast keeps f32 -- in ONE byte.  Nobody but this kernel (as producer and, two
// launches later, as the consumer of the residual) and the final LayerNorm reads that plane, so it is not a row-major matrix but a
// wave-native blob: tile (m0, n0) of an [M, N] output -> 4 waves x 8 patches x (64 lanes x 32 bytes), lane l holding quarter G's eight
// bytes at (G >> 1) KiB + 16 l + 8 (G & 1): the producer stores two 16-byte registers per patch straight from the conversion (no LDS
// transposition), the consumer fetches them by two lane-linear LDS-DMA instructions and reads its own bytes back.  Per patch 6 fetches
// + 6 stores (16 with two 16-bit planes, 8 with one).  LDS: entry e = hi 4 KiB + lo 2 KiB (slices 0-5 of ring.an / ring.bn), tables in
// slices 6-7 of ring.an and slice 6 of ring.bn, staging in ring.sp; all sixteen tail slots of the last K step are fetches of patches 0 / 1
// and the tables.  kernels.h: omk_lo8_offset is the blob's index function (the final LayerNorm kernels decode through it).
// Measured (profiles/r06_*): epilogue 33.5 k cycles per tile against 44.0 k (two 16-bit planes) and 23.8 k (one plane); encode 42.6 k
// passages/s against 41.5 k / 44.2 k; 1 - cos and max |ddot| on the config-1 fixture 0.73 x / 0.74 x of the reference's float16 autocast,
// the same as with two 16-bit planes -- but the fixture's tie-broken MRR@10 moves by one swapped pair (0.0048; 0.0000 with 16 bits, the
// reference's own float16 run 0.0007), so the default stays LNF == 3.
// LNF == 3 (round 6): the same on the TWO-PLANE residual stream of gemm_nt_kernel7 above (y = y_hi + y_lo; the residual is read as
// r_hi + r_lo, the output written as C = round16(y), ep.out_lo = round16(y - C)) -- both 16-bit formats; float16 is the headline
// format since round 4 and kept its stream in ONE plane until now, which put it 2.6 x (1 - cos) / 1.8 x (max |ddot|) outside the
// reference's own float16 autocast on the config-1 fixture (autocast keeps LayerNorm and the residual in fp32:
// HF:models/bert/modeling_bert.py:289-293,347-351 under retriever/dense_retriever.py:76).  LDS after the K loop, per wave:
//      ring.an (the spare of the last step)   slices 0-3 / 4-7 residual entry 0, hi / lo plane
//      ring.bn (the unit A(nk - 1) left)      slices 0-3 / 4-7 residual entry 1, hi / lo plane
//      ring.sp (the unit B(nk - 1) left)      slices 0-3 staging patch (used twice per patch: hi, then lo), 4 gamma | beta, 5 statistics
//                                              of the residual rows, 6 next tile's s_n | b_n
// The last K step's sixteen tail slots: both planes of patch 0 behind sub-step 0 (spare), tables (into the unit B(nk - 1) leaves: free
// behind the step's barrier, the fragment reads of sub-step 1 are the next step's) and the first plane of patch 1 behind sub-step 1;
// the second plane of patch 1 is issued when the K loop returns.  Ring of two entries: patch p + 2 is fetched into the entry patch p
// was read from one iteration earlier.
// Kernel 7r on 16 x 16 x 32 MFMAs (round 4, late): layout as kernel 7c16.  The last K step has sixteen tail slots: tables and
// residual patch 0 behind sub-step 0 (into the spare unit), patches 1 and 2 behind sub-step 1 (into the unit A(nk - 1) leaves) --
// the epilogue starts with all three ring slots in flight.  Behind the last conversion the next tile's A(1) and ALL of B(1) are
// issued (gemm_mainloop7_cont16 enters with both).
template <typename T, int ACT, int LNF>
__global__ __launch_bounds__(G6_THREADS) void gemm_nt_kernel7r16(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, T* C, int64_t ldc,
    int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  typedef T OutT;
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "16-bit in, 16-bit out");
  static_assert(LNF == 0 || LNF == 2 || LNF == 3 || LNF == 4, "residual variants: plain, or output-side LayerNorm on a one- / two-plane residual stream");
  static_assert(LNF != 4 || std::is_same<T, f16_t>::value, "the eight-bit second plane is a float16 format (half an ulp of float16 * 2^10 fits e5m2)");
  constexpr bool LNO = LNF >= 2;
  constexpr bool TWO = LNF >= 3;               // two-plane residual stream (round 6): see the layout notes above the kernel
  constexpr bool LO8 = LNF == 4;               // ... whose second plane is EIGHT bits per element (e5m2 of remainder * 2^10) in a wave-native blob
  constexpr bool TWO16 = LNF == 3;             // ... or a second 16-bit row-major plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntm = M / 256, ntn = N / 256;
  const int nk = (int)((K * 2) / G7_ROW_BYTES);
  const EpiScalars es(ep);
  const uint32_t lds_base = g7_lds_addr(smem);

  int it = 0;
  int64_t m0, n0;
  if (!g7_tile(0, ntm, ntn, group_m, m0, n0)) return;
  G7SrcU src;
  g7_offsets_u<T>(src, lda, ldb, wave, lane0);
  G7Ring ring;
  g7_ring_reset(ring);
  const char* cur_a = (const char*)(A + m0 * lda);
  const char* cur_b = (const char*)(B + n0 * ldb);
  g7_fill_a(src, cur_a, smem + ring.ac, wave);
  g7_fill_b(src, cur_b, smem + ring.bc, wave);
  g7_fill_a(src, cur_a + G7_ROW_BYTES, smem + ring.an, wave);
  g7_fill_b(src, cur_b + G7_ROW_BYTES, smem + ring.bn, wave);
  g7_table2((const float*)A, ep.bias ? ep.bias + n0 + wn * 128 : (const float*)A, smem + ring.sp + (6 * 4 + wave) * 1024, lane0);
  g7_stagger(group_m);

  bool live = false, have = true;
  int64_t pm = m0, pn = n0;
  f32x4_t acc[8][8];
  unsigned long long* tr_prev = nullptr;
  size_t ldc2 = (size_t)ldc * sizeof(OutT), ldr2 = (size_t)ep.ldr * sizeof(OutT);
  asm volatile("" : "+s"(ldc2), "+s"(ldr2));
  // second plane of the residual; absent (layer 0 adds the one-plane embedding output): fetched from the first plane and scaled by 0
  const float rlo_scale = (TWO16 && ep.resid_lo) ? 1.f : 0.f;
  const bool has_lo8 = LO8 && ep.resid_lo != nullptr;

  for (;;) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l15 = lane & 15, q4 = lane >> 4;
    const int64_t pmc = pm + wm * 128, pnc = pn + wn * 128;
    char* const an0 = smem + ring.an + wave * 1024;          // slice i at + i * 4096
    char* const bn0 = smem + ring.bn + wave * 1024;
    char* const sp0 = smem + ring.sp + wave * 1024;
    // tables: one plane -- behind the first residual slot in the spare of the last step; two planes -- in the unit B(nk - 1) left
    // tables: one plane -- behind the first residual slot in the spare of the last step; two 16-bit planes -- in the unit B(nk - 1) left;
    // 16 + 8 bits -- gamma | beta and the statistics behind entry 0 (slices 6, 7), the next tile's bias behind entry 1 (slice 6)
    const char* const etab0 = LO8 ? an0 + 6 * 4096 : (TWO ? sp0 : an0) + 4 * 4096;   // gamma | beta
    const char* const etab1 = LO8 ? an0 + 7 * 4096 : (TWO ? sp0 : an0) + 5 * 4096;   // (sum, sum of squares) of my 128 residual rows
    const char* const tab0 = (live ? (LO8 ? bn0 : (TWO ? sp0 : an0)) : sp0) + 6 * 4096;   // s_n | b_n of the tile about to start
    const bool res_ln = LNO && ep.rln_stats != nullptr;
    const char* const rbase = (const char*)((const OutT*)ep.resid + pmc * ep.ldr + pnc);
    const char* const rlo_base = (TWO16 && ep.resid_lo) ? (const char*)((const OutT*)ep.resid_lo + pmc * ep.ldr + pnc) : rbase;
    // the 8-bit plane of the finished tile (pm, pn): 16 KiB per wave tile = 8 patches of 64 lanes x 32 bytes, wave-uniform base
    const size_t blob8 = LO8 ? ((size_t)(((pm >> 8) * ntn + (pn >> 8)) * 4 + wave) * 8) * 2048 : 0;
    const char* const rl8_base = LO8 ? (const char*)(ep.resid_lo ? ep.resid_lo : ep.out_lo) + blob8 : nullptr;
    char* const cl8_base = LO8 ? (char*)ep.out_lo + blob8 : nullptr;
    const uint32_t lane16 = (uint32_t)lane * 16;
    uint32_t roff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) roff[k] = (uint32_t)((lane >> 3) * ldr2) + (((lane & 7) ^ ((4 * k + (lane >> 4)) & 7)) << 4);
    float ra[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, rc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};          // per row block of the current 32-row pair
    frag_t fb[8];
    const char* const st_rd = sp0 + (lane >> 3) * 128;
    char* const cbase = (char*)(C + pmc * ldc + pnc);
    char* const cbase_lo = TWO16 ? (char*)((OutT*)ep.out_lo + pmc * ldc + pnc) : cbase;
    const uint32_t coff = (uint32_t)((lane >> 3) * ldc2) + (lane & 7) * 16;
    float2* const stat_slot = LNO ? (float2*)ep.stats_out + ((pn >> 8) * 2 + wn) * M : nullptr;
    uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    uint4 la0, la1, la2, la3, lb0, lb1, lb2, lb3;            // two planes: the read-back of the remainder plane
    uint2 plo[8];                                             // two 16-bit planes: the patch's remainder words until the hi plane is read back
    uint2 plo8[4];                                            // 16 + 8 bits: the patch's remainder bytes, quarter G at plo8[G]

    // residual slot of patch P_: one plane -- a ring of three 4 KiB slots; two planes -- a ring of two 8 KiB entries (hi 0-3, lo 4-7)
#define G7R_SLOT(P_) (TWO ? (((P_) & 1) == 0 ? an0 : bn0) : (((P_) % 3) == 0 ? an0 : (((P_) % 3) == 1 ? bn0 : bn0 + 4 * 4096)))
#define G7R_RES_DMA(P_)                                                                                        \
  do {                                                                                                         \
    const size_t poff = (size_t)(((P_) >> 1) * 32) * ldr2 + ((P_) & 1) * 128;                                  \
    const uint32_t buf = g7_lds_addr(G7R_SLOT(P_));                                                            \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) g7_dma(rbase + poff + (size_t)(8 * k) * ldr2, roff[k], buf + k * 4096); \
    if (TWO16) { _Pragma("unroll") for (int k = 0; k < 4; ++k) g7_dma(rlo_base + poff + (size_t)(8 * k) * ldr2, roff[k], buf + (4 + k) * 4096); } \
    if (LO8) { _Pragma("unroll") for (int k = 0; k < 2; ++k) g7_dma(rl8_base + (P_) * 2048 + k * 1024, lane16, buf + (4 + k) * 4096); } \
  } while (0)
    // quarter G_ of patch P_ = (mi, nh): row block ti2 = G_ >> 1, column blocks 2 (G_ & 1) + {0, 1} -- eight values per lane
#define G7R_WRITE_Q(P_, G_)                                                                                    \
  do {                                                                                                         \
    constexpr int MI = (P_) >> 1, NH = (P_) & 1, T2 = (G_) >> 1, TI = 2 * MI + T2, FJ0 = 4 * NH + 2 * ((G_) & 1); \
    if ((G_) == 0) { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) asm volatile("" : "+a"(acc[2 * MI + (u_ >> 2)][4 * NH + (u_ & 3)])); } \
    const int rr = T2 * 16 + l15;                                                                              \
    const int64_t m = pmc + MI * 32 + rr;                                                                      \
    const int c0_ = (2 * ((G_) & 1)) * 2 + (q4 >> 1), c1_ = c0_ + 2;          /* 16-byte chunks of the patch row */ \
    const char* buf = G7R_SLOT(P_) + G7E_ROW(rr) + 8 * (q4 & 1);                                               \
    const uint2 ra_ = *(const uint2*)(buf + ((c0_ ^ ((rr >> 1) & 7)) << 4));                                   \
    const uint2 rb_ = *(const uint2*)(buf + ((c1_ ^ ((rr >> 1) & 7)) << 4));                                   \
    f32x8_t v8;                                                                                                \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) v8[e] = acc[TI][FJ0 + (e >> 2)][e & 3];                      \
    f32x8_t r8 = {Half16<OutT>::lo(ra_.x), Half16<OutT>::hi(ra_.x), Half16<OutT>::lo(ra_.y), Half16<OutT>::hi(ra_.y), \
                  Half16<OutT>::lo(rb_.x), Half16<OutT>::hi(rb_.x), Half16<OutT>::lo(rb_.y), Half16<OutT>::hi(rb_.y)}; \
    if (LO8) {           /* my own eight bytes of the quarter, as the producing lane stored them (same tile / wave / patch / lane map) */ \
      const uint2 lw_ = *(const uint2*)(G7R_SLOT(P_) + (4 + ((G_) >> 1)) * 4096 + lane * 16 + ((G_) & 1) * 8);   \
      const f32x2_t l01 = __builtin_amdgcn_cvt_pk_f32_bf8((int)lw_.x, false), l23 = __builtin_amdgcn_cvt_pk_f32_bf8((int)lw_.x, true); \
      const f32x2_t l45 = __builtin_amdgcn_cvt_pk_f32_bf8((int)lw_.y, false), l67 = __builtin_amdgcn_cvt_pk_f32_bf8((int)lw_.y, true); \
      const f32x8_t l8 = {l01[0], l01[1], l23[0], l23[1], l45[0], l45[1], l67[0], l67[1]};                     \
      if (has_lo8) r8 = __builtin_elementwise_fma(l8, (f32x8_t)(0.0009765625f), r8);       /* (select, not a product by 0: the dummy bytes may decode to inf) */ \
    }                                                                                                          \
    if (TWO16) {                                                                                               \
      const uint2 la_ = *(const uint2*)(buf + 4 * 4096 + ((c0_ ^ ((rr >> 1) & 7)) << 4));                      \
      const uint2 lb_ = *(const uint2*)(buf + 4 * 4096 + ((c1_ ^ ((rr >> 1) & 7)) << 4));                      \
      const f32x8_t l8 = {Half16<OutT>::lo(la_.x), Half16<OutT>::hi(la_.x), Half16<OutT>::lo(la_.y), Half16<OutT>::hi(la_.y), \
                          Half16<OutT>::lo(lb_.x), Half16<OutT>::hi(lb_.x), Half16<OutT>::lo(lb_.y), Half16<OutT>::hi(lb_.y)}; \
      r8 = __builtin_elementwise_fma(l8, (f32x8_t)(rlo_scale), r8);                                            \
    }                                                                                                          \
    if (LNO) {                                                                                                 \
      if (res_ln) {                                                                                            \
        const int n0_ = FJ0 * 16 + 4 * q4;                          /* columns n0_ .. + 3 and n0_ + 16 .. + 19 of my 128 */ \
        const f32x4_t ga = *(const f32x4_t*)(etab0 + n0_ * 4), gb = *(const f32x4_t*)(etab0 + (n0_ + 16) * 4);  \
        const f32x4_t ba = *(const f32x4_t*)(etab0 + 512 + n0_ * 4), bb = *(const f32x4_t*)(etab0 + 512 + (n0_ + 16) * 4); \
        const f32x8_t g8 = __builtin_shufflevector(ga, gb, 0, 1, 2, 3, 4, 5, 6, 7);                            \
        const f32x8_t b8 = __builtin_shufflevector(ba, bb, 0, 1, 2, 3, 4, 5, 6, 7);                            \
        r8 = __builtin_elementwise_fma(__builtin_elementwise_fma(r8, (f32x8_t)(ra[TI]), (f32x8_t)(rc[TI])), g8, b8); \
      }                                                                                                        \
      v8 += r8;                                                                                                \
      ssum[T2] += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));                   \
      const f32x8_t q8 = v8 * v8;                                                                              \
      ssq[T2] += ((q8[0] + q8[1]) + (q8[2] + q8[3])) + ((q8[4] + q8[5]) + (q8[6] + q8[7]));                    \
    } else {                                                                                                   \
      _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                       \
        const int64_t n = pnc + (FJ0 + jj) * 16 + 4 * q4;                                                      \
        f32x2_t lo_ = {v8[4 * jj], v8[4 * jj + 1]}, hi_ = {v8[4 * jj + 2], v8[4 * jj + 3]};                    \
        lo_ = epi_pair<ACT, false, OutT>(lo_, m, n, M, N, ep, es, 0, 0);                                       \
        hi_ = epi_pair<ACT, false, OutT>(hi_, m, n + 2, M, N, ep, es, 0, 0);                                   \
        if (es.mul) { lo_[0] *= r8[4 * jj]; lo_[1] *= r8[4 * jj + 1]; hi_[0] *= r8[4 * jj + 2]; hi_[1] *= r8[4 * jj + 3]; } \
        else {                                                                                                 \
          lo_[0] = epi_resid<ACT, true>(lo_[0], r8[4 * jj], false); lo_[1] = epi_resid<ACT, true>(lo_[1], r8[4 * jj + 1], false);       \
          hi_[0] = epi_resid<ACT, true>(hi_[0], r8[4 * jj + 2], false); hi_[1] = epi_resid<ACT, true>(hi_[1], r8[4 * jj + 3], false);   \
        }                                                                                                      \
        v8[4 * jj] = lo_[0]; v8[4 * jj + 1] = lo_[1]; v8[4 * jj + 2] = hi_[0]; v8[4 * jj + 3] = hi_[1];        \
      }                                                                                                        \
    }                                                                                                          \
    const uint2 pa_ = make_uint2(Half16<OutT>::pack2(v8[0], v8[1]), Half16<OutT>::pack2(v8[2], v8[3]));        \
    const uint2 pb_ = make_uint2(Half16<OutT>::pack2(v8[4], v8[5]), Half16<OutT>::pack2(v8[6], v8[7]));        \
    if (TWO) {          /* what the 16-bit words dropped, rounded once more: y = hi + lo to ~2^-22 (float16) / 2^-17 (bfloat16) */ \
      const f32x8_t h8 = {Half16<OutT>::lo(pa_.x), Half16<OutT>::hi(pa_.x), Half16<OutT>::lo(pa_.y), Half16<OutT>::hi(pa_.y), \
                          Half16<OutT>::lo(pb_.x), Half16<OutT>::hi(pb_.x), Half16<OutT>::lo(pb_.y), Half16<OutT>::hi(pb_.y)}; \
      const f32x8_t d8 = v8 - h8;                                                                              \
      if (LO8) {       /* e5m2 of the remainder * 2^10 (|remainder| <= half an ulp of the float16 word: <= 2^-11 |y|): y = hi + lo to ~2^-14 |y| */ \
        const f32x8_t s8_ = d8 * (f32x8_t)(1024.f);                                                            \
        int w0_ = __builtin_amdgcn_cvt_pk_bf8_f32(s8_[0], s8_[1], 0, false); w0_ = __builtin_amdgcn_cvt_pk_bf8_f32(s8_[2], s8_[3], w0_, true); \
        int w1_ = __builtin_amdgcn_cvt_pk_bf8_f32(s8_[4], s8_[5], 0, false); w1_ = __builtin_amdgcn_cvt_pk_bf8_f32(s8_[6], s8_[7], w1_, true); \
        plo8[(G_)] = make_uint2((uint32_t)w0_, (uint32_t)w1_);                                                 \
      } else {                                                                                                 \
      plo[2 * (G_)] = make_uint2(Half16<OutT>::pack2(d8[0], d8[1]), Half16<OutT>::pack2(d8[2], d8[3]));        \
      plo[2 * (G_) + 1] = make_uint2(Half16<OutT>::pack2(d8[4], d8[5]), Half16<OutT>::pack2(d8[6], d8[7]));    \
      }                                                                                                        \
    }                                                                                                          \
    *(uint2*)(sp0 + G7E_ROW(rr) + ((c0_ ^ (rr & 7)) << 4) + 8 * (q4 & 1)) = pa_;                               \
    *(uint2*)(sp0 + G7E_ROW(rr) + ((c1_ ^ (rr & 7)) << 4) + 8 * (q4 & 1)) = pb_;                               \
    if (LNO && NH == 1 && ((G_) & 1) == 1) {   /* row block T2 of this 32-row pair has all its 128 columns: partial sums of the row */ \
      /* the four lanes l15 + 16 {0, 1, 2, 3} hold the row's partial sums: the same tree as own + lane ^ 16, then + lane ^ 32 (bit-identical to \
         the shuffles of rounds 4-5) on the two VALU row / half swaps of gfx950 instead of four ds_bpermute round trips through the LDS */ \
      const float s1 = g7_quad_row_sum(ssum[T2]), s2 = g7_quad_row_sum(ssq[T2]);                               \
      if (q4 == 0) stat_slot[m] = make_float2(s1, s2);                                                         \
      ssum[T2] = 0.f; ssq[T2] = 0.f;                                                                           \
    }                                                                                                          \
  } while (0)
    // two planes: the remainder words of the patch just converted go through the same staging patch, behind the read-back of its
    // hi plane (a wave's LDS operations execute in order)
#define G7R_WRITE_LO()                                                                                         \
  do {                                                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                         \
      const int rr = (g_ >> 1) * 16 + l15;                                                                     \
      const int c0_ = (2 * (g_ & 1)) * 2 + (q4 >> 1), c1_ = c0_ + 2;                                           \
      *(uint2*)(sp0 + G7E_ROW(rr) + ((c0_ ^ (rr & 7)) << 4) + 8 * (q4 & 1)) = plo[2 * g_];                     \
      *(uint2*)(sp0 + G7E_ROW(rr) + ((c1_ ^ (rr & 7)) << 4) + 8 * (q4 & 1)) = plo[2 * g_ + 1];                 \
    }                                                                                                          \
  } while (0)
#define G7R_RB(I4) (*(const uint4*)(st_rd + (I4) * 4096 + (((lane & 7) ^ (((lane >> 3) + (I4) * 8) & 7)) << 4)))
#ifdef G7E_STORE16U
#define G7R_ST_(BASE, PP, I4, V) G7E_STORE16U((BASE) + (size_t)(((PP) >> 1) * 32 + (I4) * 8) * ldc2 + ((PP) & 1) * 128, coff, V)
#else
#define G7R_ST_(BASE, PP, I4, V) G7E_STORE16((BASE) + (size_t)(((PP) >> 1) * 32 + (I4) * 8) * ldc2 + ((PP) & 1) * 128 + coff, V)
#endif
#define G7R_ST(PP, I4, V) G7R_ST_(cbase, PP, I4, V)
    // (YWAIT: operations issued after the awaited patch's fetch, the row-statistics stores -- two per odd patch here -- NOT counted)
#define G7R_ITER(P_, YWAIT, C0, C1, C2, C3, N0, N1, N2, N3)                                                    \
  do {                                                                                                         \
    if (tr_prev && threadIdx.x == 0) tr_prev[17 + (P_)] = clock64();                                           \
    if ((P_) + 1 < 8) {                                                                                        \
      if ((P_) + 3 < 8) G7R_RES_DMA((P_) + 3);                                                                 \
      G7_WAIT_VM(YWAIT);                                                                                       \
      G7R_WRITE_Q((P_) + 1, 0); G7_FENCE_(); G7R_ST(P_, 0, C0); G7_FENCE_();                                   \
      G7R_WRITE_Q((P_) + 1, 1); G7_FENCE_(); G7R_ST(P_, 1, C1); G7_FENCE_();                                   \
      G7R_WRITE_Q((P_) + 1, 2); G7_FENCE_(); G7R_ST(P_, 2, C2); G7_FENCE_();                                   \
      G7R_WRITE_Q((P_) + 1, 3); G7_FENCE_();                                                                   \
      N0 = G7R_RB(0); N1 = G7R_RB(1); N2 = G7R_RB(2); N3 = G7R_RB(3);                                          \
      G7_FENCE_(); G7R_ST(P_, 3, C3); G7_FENCE_();                                                             \
    } else {                                                                                                   \
      G7R_ST(P_, 0, C0); G7R_ST(P_, 1, C1); G7R_ST(P_, 2, C2); G7R_ST(P_, 3, C3); G7_FENCE_();                 \
    }                                                                                                          \
  } while (0)
    // 16 + 8 bits: half K_ (quarters 2 K_, 2 K_ + 1) of patch PP's remainder bytes straight from the registers -- one 16-byte store per lane,
    // 1 KiB per wave, no LDS round trip (the consumer reads them back with the same lane map)
#ifdef G7E_STORE16U
#define G7R_ST_LO8(PP, K_) do { const uint4 v_ = make_uint4(plo8[2 * (K_)].x, plo8[2 * (K_)].y, plo8[2 * (K_) + 1].x, plo8[2 * (K_) + 1].y); \
                                G7E_STORE16U(cl8_base + (PP) * 2048 + (K_) * 1024, lane16, v_); } while (0)
#else
#define G7R_ST_LO8(PP, K_) do { const uint4 v_ = make_uint4(plo8[2 * (K_)].x, plo8[2 * (K_)].y, plo8[2 * (K_) + 1].x, plo8[2 * (K_) + 1].y); \
                                G7E_STORE16(cl8_base + (PP) * 2048 + (K_) * 1024 + lane16, v_); } while (0)
#endif
    // ... and its pipeline: the one-plane iteration with a ring of TWO entries (hi 4 KiB + lo 2 KiB), six fetches and 4 + 2 stores per patch
#define G7R_ITER8(P_, YWAIT, C0, C1, C2, C3, N0, N1, N2, N3)                                                   \
  do {                                                                                                         \
    if (tr_prev && threadIdx.x == 0) tr_prev[17 + (P_)] = clock64();                                           \
    if ((P_) + 1 < 8) {                                                                                        \
      if ((P_) + 2 < 8) G7R_RES_DMA((P_) + 2);                                                                 \
      G7_WAIT_VM(YWAIT);                                                                                       \
      G7R_WRITE_Q((P_) + 1, 0); G7_FENCE_(); G7R_ST(P_, 0, C0); G7_FENCE_();                                   \
      G7R_WRITE_Q((P_) + 1, 1); G7_FENCE_(); G7R_ST_LO8((P_) + 1, 0); G7R_ST(P_, 1, C1); G7_FENCE_();          \
      G7R_WRITE_Q((P_) + 1, 2); G7_FENCE_(); G7R_ST(P_, 2, C2); G7_FENCE_();                                   \
      G7R_WRITE_Q((P_) + 1, 3); G7_FENCE_();                                                                   \
      N0 = G7R_RB(0); N1 = G7R_RB(1); N2 = G7R_RB(2); N3 = G7R_RB(3);                                          \
      G7_FENCE_(); G7R_ST_LO8((P_) + 1, 1); G7R_ST(P_, 3, C3); G7_FENCE_();                                    \
    } else {                                                                                                   \
      G7R_ST(P_, 0, C0); G7R_ST(P_, 1, C1); G7R_ST(P_, 2, C2); G7R_ST(P_, 3, C3); G7_FENCE_();                 \
    }                                                                                                          \
  } while (0)
    // Two planes: the same pipeline with a ring of TWO residual entries (patch p + 2 is fetched into the entry patch p was read
    // from one iteration ago) and eight stores per patch -- the hi and the lo line of a row quarter behind each quarter of the next
    // patch's conversion.  Per iteration: fetch of p + 2 -> wait for p + 1 -> [convert a quarter of p + 1 (hi staged, lo words kept)
    // -> hi and lo store of p] x 4 -> read back hi of p + 1 -> stage its lo words -> read back lo.
#define G7R_ITER2(P_, YWAIT, CH0, CH1, CH2, CH3, CL0, CL1, CL2, CL3, NH0, NH1, NH2, NH3, NL0, NL1, NL2, NL3)    \
  do {                                                                                                         \
    if (tr_prev && threadIdx.x == 0) tr_prev[17 + (P_)] = clock64();                                           \
    if ((P_) + 1 < 8) {                                                                                        \
      if ((P_) + 2 < 8) G7R_RES_DMA((P_) + 2);                                                                 \
      G7_WAIT_VM(YWAIT);                                                                                       \
      G7R_WRITE_Q((P_) + 1, 0); G7_FENCE_(); G7R_ST(P_, 0, CH0); G7R_ST_(cbase_lo, P_, 0, CL0); G7_FENCE_();    \
      G7R_WRITE_Q((P_) + 1, 1); G7_FENCE_(); G7R_ST(P_, 1, CH1); G7R_ST_(cbase_lo, P_, 1, CL1); G7_FENCE_();    \
      G7R_WRITE_Q((P_) + 1, 2); G7_FENCE_(); G7R_ST(P_, 2, CH2); G7R_ST_(cbase_lo, P_, 2, CL2); G7_FENCE_();    \
      G7R_WRITE_Q((P_) + 1, 3); G7_FENCE_();                                                                   \
      NH0 = G7R_RB(0); NH1 = G7R_RB(1); NH2 = G7R_RB(2); NH3 = G7R_RB(3);                                      \
      G7_FENCE_(); G7R_ST(P_, 3, CH3); G7R_ST_(cbase_lo, P_, 3, CL3); G7_FENCE_();                             \
      G7R_WRITE_LO();                                                                                          \
      G7_FENCE_();                                                                                             \
      NL0 = G7R_RB(0); NL1 = G7R_RB(1); NL2 = G7R_RB(2); NL3 = G7R_RB(3);                                      \
      G7_FENCE_();                                                                                             \
    } else {                                                                                                   \
      G7R_ST(P_, 0, CH0); G7R_ST(P_, 1, CH1); G7R_ST(P_, 2, CH2); G7R_ST(P_, 3, CH3);                           \
      G7R_ST_(cbase_lo, P_, 0, CL0); G7R_ST_(cbase_lo, P_, 1, CL1); G7R_ST_(cbase_lo, P_, 2, CL2); G7R_ST_(cbase_lo, P_, 3, CL3); G7_FENCE_(); \
    }                                                                                                          \
  } while (0)
#define G7R_ITER2_(...) G7R_ITER2(__VA_ARGS__)
#define G7R_A sa0, sa1, sa2, sa3, la0, la1, la2, la3
#define G7R_B sb0, sb1, sb2, sb3, lb0, lb1, lb2, lb3

    if (live) {
      if (TWO16) {
        // the second plane of patch 1 (the tail's sixteen slots hold tables, both planes of patch 0 and the first of patch 1)
        const uint32_t buf = g7_lds_addr(bn0);
#pragma unroll
        for (int k = 0; k < 4; ++k) g7_dma(rlo_base + 128 + (size_t)(8 * k) * ldr2, roff[k], buf + (4 + k) * 4096);
      }
      G7_WAIT_VM(8);                           // tables and patch 0 have landed: one plane -- only patches 1 and 2 are younger; two -- both planes of patch 1
      if (res_ln) {
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
          const float2 st = *(const float2*)(etab1 + (ti * 16 + l15) * 8);
          const float mu = st.x * ep.ln_inv_h;
          const float rstd = rsqrtf(fmaxf(st.y * ep.ln_inv_h - mu * mu, 0.f) + ep.ln_eps);
          ra[ti] = rstd; rc[ti] = -mu * rstd;
        }
      }
      if (tr_prev && threadIdx.x == 0) tr_prev[16] = clock64();
      G7R_WRITE_Q(0, 0); G7R_WRITE_Q(0, 1);
      if (LO8) G7R_ST_LO8(0, 0);
      G7R_WRITE_Q(0, 2); G7R_WRITE_Q(0, 3);
      G7_FENCE_();
      sa0 = G7R_RB(0); sa1 = G7R_RB(1); sa2 = G7R_RB(2); sa3 = G7R_RB(3);
      G7_FENCE_();
      if (LO8) {
        G7R_ST_LO8(0, 1);
        G7_FENCE_();
        // younger than the awaited patch's fetch: [p = 0] the next tile's bias table (2), patch 0's two lo stores, the fetch of patch 2 (6);
        // [p = 1 .. 5] 4 + 2 stores + the next fetch (6); [p = 6] 4 + 2 stores
        G7R_ITER8(0, 10, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
        G7R_ITER8(1, 12, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
        G7R_ITER8(2, 12, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
        G7R_ITER8(3, 12, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
        G7R_ITER8(4, 12, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
        G7R_ITER8(5, 12, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
      } else if (TWO) {
        G7R_WRITE_LO();
        G7_FENCE_();
        la0 = G7R_RB(0); la1 = G7R_RB(1); la2 = G7R_RB(2); la3 = G7R_RB(3);
        G7_FENCE_();
        // operations younger than the awaited patch's fetch: [p = 0] the fetch of patch 2; [p = 1 .. 5] eight stores + the next fetch; [p = 6] eight stores
        G7R_ITER2_(0, 8, G7R_A, G7R_B);
        G7R_ITER2_(1, 16, G7R_B, G7R_A);
        G7R_ITER2_(2, 16, G7R_A, G7R_B);
        G7R_ITER2_(3, 16, G7R_B, G7R_A);
        G7R_ITER2_(4, 16, G7R_A, G7R_B);
        G7R_ITER2_(5, 16, G7R_B, G7R_A);
      } else {
        G7R_ITER(0, 8, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);        // younger than patch 1: patch 2, patch 3
        G7R_ITER(1, 12, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);       // patch 3, stores of 0, patch 4
        G7R_ITER(2, 16, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);       // stores of 0, patch 4, stores of 1, patch 5
        G7R_ITER(3, 16, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
        G7R_ITER(4, 16, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
        G7R_ITER(5, 12, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);       // stores of 3, patch 7, stores of 4
      }
    }
    if (have) {
      if (!live) G7_WAIT_VM(0);
      const bool has_b = ep.bias != nullptr;
      auto split = [](float x, uint32_t& hi, uint32_t& lo) {
        hi = Half16<T>::bits(x); lo = Half16<T>::bits(x - Half16<T>::value(hi));
      };
      float bb[8];
#pragma unroll
      for (int fj = 0; fj < 8; ++fj) bb[fj] = *(const float*)(tab0 + 512 + (fj * 16 + l15) * 4);      // (eight reads in flight, one wait)
#pragma unroll
      for (int fj = 0; fj < 8; ++fj) {
        float b = bb[fj];
        if (!has_b) b = 0.f;
        uint32_t bh, bl;
        split(b, bh, bl);
        uint4 w = make_uint4(bh | (bh << 16), bl, 0u, 0u);              // k: b_hi b_hi b_lo 0 ...
        if (q4) w = make_uint4(0u, 0u, 0u, 0u);
        fb[fj] = __builtin_bit_cast(frag_t, w);
      }
    }
    if (live) {
      if (LO8) G7R_ITER8(6, 6, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
      else if (TWO) G7R_ITER2_(6, 8, G7R_A, G7R_B);
      else G7R_ITER(6, 8, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3);
      if (have) {                              // every ring slot and table of this epilogue has been read: A(1) and B(1) of the next tile
        g7_fill_a(src, cur_a + G7_ROW_BYTES, smem + ring.an, wave);
        g7_fill_b(src, cur_b + G7_ROW_BYTES, smem + ring.bn, wave);
      }
    }
    if (have) {
      // the row-side factor is the same for every row block (u = 1: k slots u_hi u_lo u_hi); ONE fragment, made opaque per row block so
      // that the sixty-four products stay sixty-four MFMAs (equal operands would be eight MFMAs + 56 accumulator copies) -- eight
      // separate fragments cost 28 more registers across the seventh patch (the two-plane variant has none to spare)
      uint32_t uh = Half16<T>::bits(1.f), ul = Half16<T>::bits(1.f - Half16<T>::value(Half16<T>::bits(1.f)));
      uint4 w = make_uint4(uh | (ul << 16), uh, 0u, 0u);              // k: u_hi u_lo u_hi 0 ...
      if (q4) w = make_uint4(0u, 0u, 0u, 0u);
      const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ti = 0; ti < 8; ++ti) {
        asm volatile("" : "+v"(w.x), "+v"(w.y));
        const frag_t fa = __builtin_bit_cast(frag_t, w);
#pragma unroll
        for (int fj = 0; fj < 8; ++fj) { acc[ti][fj] = zero; Mma16c<T>::mma(fb[fj], fa, acc[ti][fj]); }
      }
    }
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("" : "+a"(acc[q >> 3][q & 7]));
    if (live) {
      if (LO8) G7R_ITER8(7, 0, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
      else if (TWO) G7R_ITER2_(7, 0, G7R_B, G7R_A);
      else G7R_ITER(7, 0, sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
      if (tr_prev && threadIdx.x == 0) { tr_prev[28] = clock64(); tr_prev[29] = blockIdx.x; tr_prev[31] = wall_clock64(); }
    }
#undef G7R_A
#undef G7R_B
#undef G7R_ITER2_
#undef G7R_ITER2
#undef G7R_ITER8
#undef G7R_ST_LO8
#undef G7R_ITER
#undef G7R_ST
#undef G7R_ST_
#undef G7R_RB
#undef G7R_WRITE_LO
#undef G7R_WRITE_Q
#undef G7R_RES_DMA
#undef G7R_SLOT
    if (!have) break;
    if (!live) __builtin_amdgcn_s_barrier();

    ++it;
    int64_t m1 = m0, n1 = n0;
    const bool has_next = g7_tile(it, ntm, ntn, group_m, m1, n1);
    const char* const next_a = (const char*)(A + m1 * lda);
    const char* const next_b = (const char*)(B + n1 * ldb);
    unsigned long long* tr = nullptr;
    if (ep.trace) {
      const int64_t tile_id = (m0 / 256) * ntn + n0 / 256;
      if (tile_id < 8192) tr = ep.trace + tile_id * 32;
    }
    if (tr && threadIdx.x == 0) { tr[0] = tr[1] = clock64(); tr[30] = wall_clock64(); }
    {
      const int64_t mc = m0 + wm * 128, nc = n0 + wn * 128;
      const char* const rb_ = (const char*)((const OutT*)ep.resid + mc * ep.ldr + nc);
      const char* const rl_ = (TWO16 && ep.resid_lo) ? (const char*)((const OutT*)ep.resid_lo + mc * ep.ldr + nc) : rb_;
      const char* const rl8_ = LO8 ? (const char*)(ep.resid_lo ? ep.resid_lo : ep.out_lo) + ((size_t)(((m0 >> 8) * ntn + (n0 >> 8)) * 4 + wave) * 8) * 2048 : nullptr;
      const uint32_t l16_ = (uint32_t)lane0 * 16;
      const bool rln = LNO && ep.rln_stats != nullptr;
      const float* const dummy = (const float*)A;
      const float* const t_g = rln ? ep.rln_g + nc : dummy;
      const float* const t_b = rln ? ep.rln_b + nc : dummy;
      const float* const t_s = rln ? ep.rln_stats + mc * 2 : dummy;
      const float* const t_bias = ep.bias ? ep.bias + n1 + wn * 128 : dummy;
      uint32_t ro[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) ro[k] = (uint32_t)((lane0 >> 3) * ldr2) + (((lane0 & 7) ^ ((4 * k + (lane0 >> 4)) & 7)) << 4);
      const int w1k = wave * 1024;
      auto tail = [&](int slot, int u_spare, int u_olda, int u_oldb) __attribute__((always_inline)) {
        char* const s0 = smem + u_spare + w1k;
        const uint32_t o0 = lds_base + u_olda + w1k;
        if (LO8) {
          // slots 0-7 (behind sub-step 0: the spare only): patch 0 (hi 0-3, lo 4-5), gamma | beta (6), statistics (7); 8-15 (behind the step's
          // barrier, into the unit A(nk - 1) leaves): patch 1 (hi 0-3, lo 4-5), the next tile's bias table (6; twice: the count is fixed)
          (void)u_oldb;
          if (slot < 4) g7_dma(rb_ + (size_t)(8 * slot) * ldr2, ro[slot], g7_lds_addr(s0) + slot * 4096);
          else if (slot < 6) g7_dma(rl8_ + (slot - 4) * 1024, l16_, g7_lds_addr(s0) + slot * 4096);
          else if (slot == 6) g7_table2(t_g, t_b, s0 + 6 * 4096, lane0);
          else if (slot == 7) g7_table1(t_s, s0 + 7 * 4096, lane0);
          else if (slot < 12) g7_dma(rb_ + 128 + (size_t)(8 * (slot - 8)) * ldr2, ro[slot - 8], o0 + (slot - 8) * 4096);
          else if (slot < 14) g7_dma(rl8_ + 2048 + (slot - 12) * 1024, l16_, o0 + (slot - 8) * 4096);
          else g7_table2(dummy, t_bias, smem + u_olda + w1k + 6 * 4096, lane0);
        } else if (TWO) {
          // slots 0-7 (behind sub-step 0: the spare only) both planes of patch 0; 8-11 (behind the step's barrier) the tables into the
          // unit B(nk - 1) left; 12-15 the first plane of patch 1 into the unit A(nk - 1) left
          char* const b0 = smem + u_oldb + w1k;
          if (slot < 4) g7_dma(rb_ + (size_t)(8 * slot) * ldr2, ro[slot], g7_lds_addr(s0) + slot * 4096);
          else if (slot < 8) g7_dma(rl_ + (size_t)(8 * (slot - 4)) * ldr2, ro[slot - 4], g7_lds_addr(s0) + slot * 4096);
          else if (slot == 8) g7_table2(t_g, t_b, b0 + 4 * 4096, lane0);
          else if (slot == 9) g7_table1(t_s, b0 + 5 * 4096, lane0);
          else if (slot < 12) g7_table2(dummy, t_bias, b0 + 6 * 4096, lane0);
          else g7_dma(rb_ + 128 + (size_t)(8 * (slot - 12)) * ldr2, ro[slot - 12], o0 + (slot - 12) * 4096);
        } else {
          // spare: slices 0-3 patch 0, 4 gamma | beta, 5 statistics, 6 next table; the unit A(nk - 1) leaves: slices 0-3 patch 1, 4-7 patch 2
          (void)u_oldb;
          if (slot == 0) g7_table2(t_g, t_b, s0 + 4 * 4096, lane0);
          else if (slot == 1) g7_table1(t_s, s0 + 5 * 4096, lane0);
          else if (slot == 2 || slot == 3) g7_table2(dummy, t_bias, s0 + 6 * 4096, lane0);
          else if (slot < 8) g7_dma(rb_ + (size_t)(8 * (slot - 4)) * ldr2, ro[slot - 4], g7_lds_addr(s0) + (slot - 4) * 4096);     // patch 0 = (mi 0, nh 0)
          else if (slot < 12) g7_dma(rb_ + 128 + (size_t)(8 * (slot - 8)) * ldr2, ro[slot - 8], o0 + (slot - 8) * 4096);          // patch 1 = (mi 0, nh 1)
          else g7_dma(rb_ + (size_t)32 * ldr2 + (size_t)(8 * (slot - 12)) * ldr2, ro[slot - 12], o0 + (slot - 8) * 4096);         // patch 2 = (mi 1, nh 0)
        }
      };
      gemm_mainloop7_cont16<T, true>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc, tr, tail);
    }
    if (tr && threadIdx.x == 0) tr[15] = clock64();
    tr_prev = tr;
    pm = m0; pn = n0; live = true;
    cur_a = next_a; cur_b = next_b; m0 = m1; n0 = n1; have = has_next;
  }
  G7_WAIT_VM(0);
}

static int g7_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

template <typename T, int ACT, int LNF, bool TRAIN = false>
static int launch7c(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t ntiles = (M / 256) * (N / 256);
  if (ntiles >= 0x7fff0000LL) OM_FAIL("gemm: more than 2^31 output tiles");
  if (TRAIN && (!ep.pre_act || ep.ldp != ldc || ep.drop_p > 0.f || !(ep.act & OM_ACT_PRE_GRAD) || ((uintptr_t)ep.pre_act & 15)))
    OM_FAIL("two-output GELU epilogue: pre_act with ldp == ldc, OM_ACT_PRE_GRAD, no dropout");
  int grid = g7_num_cus();
  const int cap = om_option(OM_OPT_GEMM_MAX_GRID);
  if (cap > 0 && cap < grid) grid = cap;
  if (ntiles < grid) grid = (int)ntiles;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel7c16<T, ACT, LNF, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    attr_set = true;
  }
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(OM_TIMING_GEMM_BF16, s);
  const int gm_arg = (std::max(1, om_option(OM_OPT_GEMM_GROUP_M)) & 0xffff) | (ep.reverse ? 1 << 16 : 0) | g7_stagger_bits();
  hipLaunchKernelGGL((gemm_nt_kernel7c16<T, ACT, LNF, TRAIN>), dim3((unsigned)grid), dim3(G6_THREADS), G7_LDS_BYTES, s,
                     (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M, N, K, ep, gm_arg);
  if (timing) om_timing_end(OM_TIMING_GEMM_BF16, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T, int ACT, int LNF>
static int launch7r(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t ntiles = (M / 256) * (N / 256);
  if (ntiles >= 0x7fff0000LL) OM_FAIL("gemm: more than 2^31 output tiles");
  if (LNF >= 3 && (!ep.out_lo || ((uintptr_t)ep.out_lo & 15) || ((uintptr_t)ep.resid_lo & 15))) OM_FAIL("two-plane residual epilogue: out_lo (and resid_lo) 16-byte aligned planes");
  int grid = g7_num_cus();
  const int cap = om_option(OM_OPT_GEMM_MAX_GRID);
  if (cap > 0 && cap < grid) grid = cap;
  if (ntiles < grid) grid = (int)ntiles;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel7r16<T, ACT, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    attr_set = true;
  }
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(OM_TIMING_GEMM_BF16, s);
  const int gm_arg = (std::max(1, om_option(OM_OPT_GEMM_GROUP_M)) & 0xffff) | (ep.reverse ? 1 << 16 : 0) | g7_stagger_bits();
  hipLaunchKernelGGL((gemm_nt_kernel7r16<T, ACT, LNF>), dim3((unsigned)grid), dim3(G6_THREADS), G7_LDS_BYTES, s,
                     (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M, N, K, ep, gm_arg);
  if (timing) om_timing_end(OM_TIMING_GEMM_BF16, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T, int ACT, bool RESID, int LNF>
static int launch7(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  if constexpr (!RESID && LNF <= 1) {      // the continuous ring: needs three K steps (its prefetch reaches at most one tile ahead)
    if (K * 2 >= 3 * G7_ROW_BYTES && (om_option(OM_OPT_GEMM_CONT) & 1) != 0) return launch7c<T, ACT, LNF>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  }
  if constexpr (RESID && (LNF == 0 || LNF == 2)) {      // one-plane residual variants on the continuous ring (bit 1 of the option)
    if (K * 2 >= 3 * G7_ROW_BYTES && (om_option(OM_OPT_GEMM_CONT) & 2) != 0) return launch7r<T, ACT, LNF>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  }
  if constexpr (LNF == 4) {                              // float16 with the eight-bit second plane: the continuous kernel only
    if (K * 2 < 3 * G7_ROW_BYTES) OM_FAIL("two-plane residual epilogue with the eight-bit plane: K >= 192");
    return launch7r<T, ACT, LNF>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  } else {
  if constexpr (RESID && LNF == 3) {                     // two-plane residual variant on the continuous ring (round 6): bit 8 float16, bit 9 bfloat16
    // (bfloat16 stays on the restart-per-tile kernel by default: the continuous kernel is 1.3 % faster end to end, but its 16 x 16 x 32 summation
    // order moves the config-1 fixture's tie-broken MRR@10 from 0.0025 to 0.0037 against the reference's own 0.0035 -- one swapped pair)
    constexpr int bit = sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? 512 : 256;
    if (K * 2 >= 3 * G7_ROW_BYTES && (om_option(OM_OPT_GEMM_CONT) & bit) != 0) return launch7r<T, ACT, LNF>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  }
  const int64_t ntiles = (M / 256) * (N / 256);
  if (ntiles >= 0x7fff0000LL) OM_FAIL("gemm: more than 2^31 output tiles");      // g7_tile works in 32 bits
  int grid = g7_num_cus();
  const int cap = om_option(OM_OPT_GEMM_MAX_GRID);       // > 0: at most this many workgroups (multiples of 8 keep the XCD-aware walk)
  if (cap > 0 && cap < grid) grid = cap;
  if (ntiles < grid) grid = (int)ntiles;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel7<T, ACT, RESID, LNF, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    attr_set = true;
  }
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(OM_TIMING_GEMM_BF16, s);
  hipLaunchKernelGGL((gemm_nt_kernel7<T, ACT, RESID, LNF, true>), dim3((unsigned)grid), dim3(G6_THREADS), G7_LDS_BYTES, s,
                     (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M, N, K, ep, (std::max(1, om_option(OM_OPT_GEMM_GROUP_M)) & 0xffff) | (ep.reverse ? 1 << 16 : 0));
  if (timing) om_timing_end(OM_TIMING_GEMM_BF16, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
  }
}
