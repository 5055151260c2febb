// Seventh-generation NT GEMM main loop for gfx950 (16-bit inputs): the 256 x 256 tile / four waves of
// 128 x 128 of gemm_core6.h, fed in 128-BYTE K steps (64 bf16) so that every LDS-DMA request is a
// whole 128-byte cache line.
//
// Why (profiles/r02_gemm_v6_pmc_memory_path.txt, profiles/r02_vendor_gemm_reference.jsonl): with
// 64-byte K steps a wave's DMA instruction covers 16 rows x 64 B, i.e. 16 HALF lines; the vector
// L1 (TCP) sends one 64-byte request per half line (TCP_TCC_READ_REQ = bytes / 64), sits in
// TCP_PENDING_STALL 41 % of the kernel with an average L2 latency of only 265-305 cycles, and the K
// step takes 1320-1550 cycles on streamed operands against 1123 cache-hot: the per-CU miss queue,
// not L2 bandwidth or latency, bounds the stream.  Whole-line requests carry twice the bytes per
// queue entry.  (The vendor GEMM runs these K = 768 shapes at 1.13-1.20 PFLOP/s, v6 at 0.72-0.92.)
//
// LDS: five 32 KiB units, each one operand tile of one K step, [256 rows][128 B]; a row holds four
// MFMA k sub-steps of two 16-byte chunks (k halves), chunk position XOR ((row >> 1) & 7) -- every
// ds_read_b128 lane group then covers 16 distinct 16-byte slots of the 256-byte bank row, and the
// eight lanes that fetch a row read one whole line (permuted).  Units rotate:
//     step t computes from (A_cur, B_cur); (A_nxt, B_nxt) hold step t+1;
//     A(t+2) is fetched into the spare unit during sub-steps 0-1 of step t,
//     B(t+2) into A_cur once every wave has read its last A(t) fragment (the step's single barrier,
//     between sub-steps 2 and 3, which also publishes step t+1), during sub-step 3.
// Per wave and step: 64 MFMA (2048 cycles), 32 ds_read_b128, 16 global_load_lds_dwordx4, one
// s_waitcnt vmcnt(8) + s_barrier.  Issue order is pinned by hand as in gemm_core6.h.
#pragma once
#include <type_traits>
#include "gemm_core6.h"

#define G7_ROW_BYTES 128
// Developer probe (tools/gemm7_probe.hip only; 0 in the product): switches parts of the tile off to attribute its cycles.
//   1 no output stores   2 no epilogue LDS staging   4 no epilogue at all   8 no K-loop DMA   16 no fragment reads
//   32 no MFMA   64 no K-loop barrier
#ifndef G7_ABL
#define G7_ABL 0
#endif
#define G7_UNIT_BYTES (256 * G7_ROW_BYTES)     // 32 KiB
#define G7_UNITS 5
#define G7_LDS_BYTES (G7_UNITS * G7_UNIT_BYTES)   // 160 KiB: the whole LDS

// DMA sources of tile (m0, n0): a wave-uniform 64-bit base per operand (the tile's first row, advanced by the
// K offset in scalar registers) plus a 32-bit per-lane byte offset per instruction -- the saddr form of
// global_load_lds, so the steady state spends no vector ALU on addresses.  Instruction i of this wave moves
// tile rows (i*4 + wave)*8 .. +7: lane -> row (lane >> 3), physical chunk (lane & 7).  Rows past M / N are
// clamped to the last valid row (their results are never stored).
struct G7Src {
  const char* a;        // A + m0 * lda   (wave-uniform)
  const char* b;        // B + n0 * ldb
  uint32_t oa[8], ob[8];
};
// offsets of a tile whose rows all exist (every tile when M and N are multiples of 256): the same for every tile
template <typename T>
__device__ __forceinline__ void g7_offsets(G7Src& src, int64_t lda, int64_t ldb, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (i * 4 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    src.oa[i] = (uint32_t)(r * lda * (int64_t)sizeof(T)) + c * 16;
    src.ob[i] = (uint32_t)(r * ldb * (int64_t)sizeof(T)) + c * 16;
  }
}
template <typename T>
__device__ __forceinline__ void g7_point(G7Src& src, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                         int64_t ldb, int64_t M, int64_t N, int64_t m0, int64_t n0, int wave, int lane) {
  src.a = (const char*)(A + m0 * lda);
  src.b = (const char*)(B + n0 * ldb);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (i * 4 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t ra = r; if (m0 + ra > M - 1) ra = M - 1 - m0;
    int64_t rb = r; if (n0 + rb > N - 1) rb = N - 1 - n0;
    src.oa[i] = (uint32_t)(ra * lda * (int64_t)sizeof(T)) + c * 16;
    src.ob[i] = (uint32_t)(rb * ldb * (int64_t)sizeof(T)) + c * 16;
  }
}

// One LDS-DMA instruction in its scalar-base form: 64 lanes x 16 bytes from (wave-uniform base + 32-bit lane offset)
// to LDS bytes [lds, lds + 1024).  Inline assembly because the builtin only selects the 64-bit-VGPR-address form
// (a 64-bit vector add per instruction and a register pair per pointer).  hipcc does not count these in its own
// s_waitcnt bookkeeping: every wait for them in this generation is written by hand.  M0 carries the LDS address
// (nothing else in these kernels uses M0); `s_nop 4` covers both the M0 write -> DMA wait state and a base that the
// compiler produced with v_readfirstlane (VALU-written SGPR -> VMEM address: 5 wait states).
// G7_DMA_FORM: 2 (default since late round 4) = the base is copied to a scratch SGPR pair by s_mov_b64 -- a SALU-written address
// needs no wait states before the VMEM instruction whatever produced `base`, and the copy IS the one wait state M0 needs: no
// s_nop.  0 = `s_nop 4` as rounds 2-4 had it: five wait states of four cycles each on the only wave of the SIMD, sixteen times per K
// step -- 260 of a step's 2 620 cycles (tools/gemm7h_probe.hip, profiles/r04_probe17_*: K loop 2 622 -> 2 425 cycles per step, 566 ->
// 551 us at K = 3072).  1 = `s_nop 0` (probe only: unsafe if the compiler hands over a v_readfirstlane result; 2 362 cycles).
#ifndef G7_DMA_FORM
#define G7_DMA_FORM 2
#endif
__device__ __forceinline__ void g7_dma(const char* base, uint32_t lane_off, uint32_t lds) {
#if G7_DMA_FORM == 2
  const char* t_;
  asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, %3\n\tglobal_load_lds_dwordx4 %2, %0" : "=&s"(t_) : "s"(lds), "v"(lane_off), "s"(base) : "memory");
#elif G7_DMA_FORM == 1
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane_off), "s"(base) : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane_off), "s"(base) : "memory");
#endif
}
// the same with the non-temporal cache policy: for streams that ONE CU reads ONCE (the index pass of a small query batch);
// MI355X_MICROARCH.md "nt-weights": issued -> landed -18 %, chip 6.5-6.8 instead of 6.4 TB/s
__device__ __forceinline__ void g7_dma_nt(const char* base, uint32_t lane_off, uint32_t lds) {
#if G7_DMA_FORM == 2
  const char* t_;
  asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, %3\n\tglobal_load_lds_dwordx4 %2, %0 nt" : "=&s"(t_) : "s"(lds), "v"(lane_off), "s"(base) : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds), "v"(lane_off), "s"(base) : "memory");
#endif
}
// the same with a full 64-bit address per lane (unrelated sources in one instruction)
__device__ __forceinline__ void g7_dma_v(const void* lane_ptr, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(lane_ptr) : "memory");
}
__device__ __forceinline__ uint32_t g7_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lptr_t)p; }

// all eight DMA instructions of one operand unit (prologue only; the steady state spreads them)
__device__ __forceinline__ void g7_fill(const char* base, const uint32_t (&off)[8], char* unit, int wave) {
  const uint32_t lds = g7_lds_addr(unit) + wave * 1024;
#pragma unroll
  for (int i = 0; i < 8; ++i) g7_dma(base, off[i], lds + i * 4096);
}

// Whole tiles only (the persistent GEMM: M, N multiples of 256): instruction i's lane offset is instruction 0's plus
// i * 32 rows -- the swizzle term ((r >> 1) & 7) does not depend on i -- so ONE VGPR per operand and a scalar stride
// replace the eight offsets of G7Src (14 VGPRs that the deferred output vectors of gemm_wide7.h need).
struct G7SrcU {
  const char* a;
  const char* b;
  uint32_t oa0, ob0;    // per lane: row ((wave * 8) + (lane >> 3)), swizzled chunk
  uint32_t sa, sb;      // 32 rows of A / B in bytes (wave-uniform)
};
template <typename T>
__device__ __forceinline__ void g7_offsets_u(G7SrcU& src, int64_t lda, int64_t ldb, int wave, int lane) {
  const int r = wave * 8 + (lane >> 3);
  const int c = (lane & 7) ^ ((r >> 1) & 7);
  src.oa0 = (uint32_t)(r * lda * (int64_t)sizeof(T)) + c * 16;
  src.ob0 = (uint32_t)(r * ldb * (int64_t)sizeof(T)) + c * 16;
  src.sa = (uint32_t)(32 * lda * (int64_t)sizeof(T));
  src.sb = (uint32_t)(32 * ldb * (int64_t)sizeof(T));
}
__device__ __forceinline__ void g7_issue_a(const G7Src& s, const char* base, int i, uint32_t lds) { g7_dma(base, s.oa[i], lds); }
__device__ __forceinline__ void g7_issue_b(const G7Src& s, const char* base, int i, uint32_t lds) { g7_dma(base, s.ob[i], lds); }
__device__ __forceinline__ void g7_issue_a(const G7SrcU& s, const char* base, int i, uint32_t lds) { g7_dma(base + (size_t)((uint32_t)i * s.sa), s.oa0, lds); }
__device__ __forceinline__ void g7_issue_b(const G7SrcU& s, const char* base, int i, uint32_t lds) { g7_dma(base + (size_t)((uint32_t)i * s.sb), s.ob0, lds); }
// all eight DMA instructions of one operand unit from a compact descriptor
__device__ __forceinline__ void g7_fill_a(const G7SrcU& s, const char* base, char* unit, int wave) {
  const uint32_t lds = g7_lds_addr(unit) + wave * 1024;
#pragma unroll
  for (int i = 0; i < 8; ++i) g7_issue_a(s, base, i, lds + i * 4096);
}
__device__ __forceinline__ void g7_fill_b(const G7SrcU& s, const char* base, char* unit, int wave) {
  const uint32_t lds = g7_lds_addr(unit) + wave * 1024;
#pragma unroll
  for (int i = 0; i < 8; ++i) g7_issue_b(s, base, i, lds + i * 4096);
}

// K steps 0 and 1 of a tile into units 0-3 (32 DMA instructions per wave).
__device__ __forceinline__ void g7_begin(const G7Src& src, int nk, char* smem, int wave) {
  g7_fill(src.a, src.oa, smem, wave);
  g7_fill(src.b, src.ob, smem + G7_UNIT_BYTES, wave);
  if (nk > 1) {
    g7_fill(src.a + G7_ROW_BYTES, src.oa, smem + 2 * G7_UNIT_BYTES, wave);
    g7_fill(src.b + G7_ROW_BYTES, src.ob, smem + 3 * G7_UNIT_BYTES, wave);
  }
}

// The K loop of one tile whose steps 0 and 1 are in flight (g7_begin, nk >= 2).  acc as in gemm_core6.h:
//   acc[mi][ni][r] = C[m0 + wm*128 + mi*32 + (lane&31)][n0 + wn*128 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
// Per-step trace stamps (tr[3 + t]) are compiled in only for the probes (-DG7_TRACE_STEPS; tools/gemm7_probe.hip): with a run-time
// trace pointer the test is a wave-uniform branch in EVERY K step of every production launch -- on the only wave of a SIMD.  The
// library stamps tr[3] once, in front of the loop.
#ifdef G7_TRACE_STEPS
#define G7_STEP_STAMP() do { if (tr && tid == 0 && t < 12) tr[3 + t] = clock64(); } while (0)
#define G7_LOOP_STAMP() do {} while (0)
#else
#define G7_STEP_STAMP() do {} while (0)
#define G7_LOOP_STAMP() do { if (tr && tid == 0) tr[3] = clock64(); } while (0)
#endif
template <typename T, typename SRC = G7Src>
__device__ __forceinline__ void gemm_mainloop7_run(const SRC& src, int nk, char* smem,
                                          f32x16_t (&acc)[4][4], unsigned long long* tr = nullptr,
                                          bool stores_pending = false) {
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "128-byte K steps: 16-bit operands only");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 1) & 7;           // == ((row >> 1) & 7) for row = 32*x + (lane & 31)
  const int half = lane >> 5;
  int slot[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) slot[kk] = (((kk << 1) | half) ^ key) << 4;
  const int rowa = (wm * 128 + (lane & 31)) * G7_ROW_BYTES;
  const int rowb = (wn * 128 + (lane & 31)) * G7_ROW_BYTES;

  // steps 0 and 1 are in flight: step 0 has landed once only step 1's 16 instructions are outstanding -- plus, in the
  // persistent kernel, the 32 output stores of the previous tile's epilogue, issued between the two (vmcnt retires in
  // order; the step-0 fetch is OLDER than those stores, so it must not wait for their acknowledgement)
  if (stores_pending) { if (nk > 1) __builtin_amdgcn_s_waitcnt(0xC070); else __builtin_amdgcn_s_waitcnt(0x8070); }   // vmcnt(48) / vmcnt(32)
  else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x4070);      // vmcnt(16) lgkmcnt(0)
  else __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0)
  __builtin_amdgcn_s_barrier();

  const uint32_t lds0 = g7_lds_addr(smem);
  const char* ka = src.a + 2 * G7_ROW_BYTES;      // K offset of step t + 2 (scalar registers)
  const char* kb = src.b + 2 * G7_ROW_BYTES;
  int u_ac = 0, u_bc = G7_UNIT_BYTES, u_an = 2 * G7_UNIT_BYTES, u_bn = 3 * G7_UNIT_BYTES, u_sp = 4 * G7_UNIT_BYTES;
  frag_t a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const frag_t*)(smem + u_ac + rowa + i * 32 * G7_ROW_BYTES + slot[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b0[i] = *(const frag_t*)(smem + u_bc + rowb + i * 32 * G7_ROW_BYTES + slot[0]);

#define G7_FENCE() __builtin_amdgcn_sched_barrier(0)
#define G7_DMA(P, I, UNIT) do { if (!(G7_ABL & 8)) g7_issue_##P(src, k##P, I, lds0 + (UNIT) + ((I) * 4 + wave) * 1024); } while (0)
  // one k sub-step: 16 MFMAs from (AF, BF); the first eight each cover one fragment read into (AN, BN) from
  // (UA, UB) chunk SLOT; MFMAs 8, 10, 12, 14 each cover one DMA issue of operand P into UNIT (instructions
  // DBASE .. DBASE + 3) when the wave-uniform COND holds
#define G7_SUB(AF, BF, AN, BN, UA, UB, SLOT, DO_READ, DPOS, P, UNIT, DBASE, COND)                        \
  _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                       \
    if (!(G7_ABL & 32)) MmaOps<T>::mma(BF[q & 3], AF[q >> 2], acc[q >> 2][q & 3]);                       \
    if (q < 8 && (DO_READ) && !(G7_ABL & 16)) {                                                          \
      if (q < 4) AN[q] = *(const frag_t*)(smem + (UA) + rowa + q * 32 * G7_ROW_BYTES + (SLOT));          \
      else BN[q - 4] = *(const frag_t*)(smem + (UB) + rowb + (q - 4) * 32 * G7_ROW_BYTES + (SLOT));      \
    }                                                                                                    \
    if ((DPOS) == 1 && q >= 8 && !(q & 1)) { if (COND) G7_DMA(P, (DBASE) + ((q - 8) >> 1), UNIT); }      \
    G7_FENCE();                                                                                          \
  }
  // Even schedule (round 3): FOUR issues behind every sub-step (round 2: eight back-to-back B issues in sub-step 3):
  //     sub-step 0   B(t+1) instructions 4-7 -> B_nxt   (the half the previous step's sub-step 3 left out)
  //     sub-step 1   A(t+2) instructions 0-3 -> spare
  //     sub-step 2   A(t+2) instructions 4-7 -> spare
  //     barrier      vmcnt(8): everything but A(t+2) has landed -- B(t+1)'s second half is >= 2 sub-steps old
  //     sub-step 3   B(t+2) instructions 0-3 -> A_cur
  // ONE step body for the whole loop: whether a step still issues (t + 2 < nk) and whether the previous step issued the
  // first half of B(t+1) are wave-uniform run-time flags (a scalar branch around each DMA), and the last step reads
  // "next" fragments nobody uses.  Peeled first / last steps cost 7 KiB of code each, and kernels past ~40 KiB ran three to
  // four times slower (instruction cache; profiles/r03_gemm7_ablation_v0.log vs the peeled build).
#define G7_STEP(ISSUE, B2H)                                                                              \
  do {                                                                                                   \
    G7_STEP_STAMP();                                                                                     \
    { const char* const kb_cur = kb; kb -= G7_ROW_BYTES;                                                 \
      G7_SUB(a0, b0, a1, b1, u_ac, u_bc, slot[1], true, 1, b, u_bn, 4, B2H)                              \
      kb = kb_cur; }                                                                                     \
    G7_SUB(a1, b1, a0, b0, u_ac, u_bc, slot[2], true, 1, a, u_sp, 0, ISSUE)                              \
    G7_SUB(a0, b0, a1, b1, u_ac, u_bc, slot[3], true, 1, a, u_sp, 4, ISSUE)                              \
    if (ISSUE) __builtin_amdgcn_s_waitcnt(0x0078); else __builtin_amdgcn_s_waitcnt(0x0070);             \
    if (!(G7_ABL & 64)) __builtin_amdgcn_s_barrier();                                                    \
    G7_FENCE();                                                                                          \
    G7_SUB(a1, b1, a0, b0, u_an, u_bn, slot[0], true, 1, b, u_ac, 0, ISSUE)                              \
    { const int o_ac = u_ac, o_bc = u_bc; u_ac = u_an; u_bc = u_bn; u_an = u_sp; u_bn = o_ac; u_sp = o_bc; } \
    ka += G7_ROW_BYTES; kb += G7_ROW_BYTES;                                                              \
  } while (0)

  G7_LOOP_STAMP();
  for (int t = 0; t < nk; ++t) {
    const bool issue = t + 2 < nk;
    const bool b2h = t > 0 && t + 1 < nk;
    G7_STEP(issue, b2h);
  }
#undef G7_STEP
#undef G7_SUB
#undef G7_DMA
#undef G7_FENCE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                             // everyone is done with the ring
}

// ---- the continuous ring (round 4) --------------------------------------------------------------------------------------
// The K loop above restarts the ring for every tile: the epilogue issues K step 0 of the next tile (16 DMA instructions
// among its stores), the tile start issues K step 1 (16 more, ~1.3-1.9 k cycles of issue time at ~100 cycles each while
// the stores drain) and the first steps run on a cold pipeline.  The tile traces of round 4 (profiles/r04_probe1_*) show
// that every phase of a tile costs the SAME number of shader cycles whatever else the chip does (8 ... 256 active CUs:
// K loop 32.9 k, plain epilogue 7.8 k, tile start 3.8-4.3 k) -- the whole tile is a per-CU instruction stream, and the
// 12 k cycles outside the K loop are 28 % of an encoder layer.  Here the ring NEVER stops: step t of a tile always issues
// steps t + 1 (second half of B) and t + 2, and once t + 2 reaches the tile's step count the sources are the NEXT tile's
// first steps.  When a tile's last step ends, steps 0 and 1 of the next tile are already landed / in flight in four of
// the five units; the epilogue lives in the fifth (the spare: this wave's own 1 KiB slices of it, so no barrier), and
// the next tile's first MFMA follows the epilogue's last store without a wait, a barrier or a DMA issue in between.
//
// On entry (steady state and first tile alike): step 0 of the tile is landed and published (ring.ac / ring.bc), A(1) is
// issued into ring.an, the FIRST half of B(1) (instructions 0-3) into ring.bn; acc holds the tile's initial values.
// Needs nk >= 3 (the sources of step t + 2 are at most one tile ahead).
struct G7Ring { int ac, bc, an, bn, sp; };      // byte offsets of the five units: A / B of the current step, of the next, spare
__device__ __forceinline__ void g7_ring_reset(G7Ring& r) {
  r.ac = 0; r.bc = G7_UNIT_BYTES; r.an = 2 * G7_UNIT_BYTES; r.bn = 3 * G7_UNIT_BYTES; r.sp = 4 * G7_UNIT_BYTES;
}
// TAIL (residual variants, gemm_wide7.h kernel 7r): the LAST step of a tile gives its twelve issue slots behind sub-steps
// 1, 2, 3 -- A(nk + 1) and the first half of B(nk + 1) in the plain ring -- to tail(slot 0..11, spare unit, the unit A(nk - 1) leaves, the unit
// B(nk - 1) leaves -- free behind the step's barrier like A's: the fragment reads behind it are the NEXT step's): the epilogue's tables and
// its first residual patches go into the units that step frees (the spare during sub-steps 1-2, the unit A(nk - 1) leaves
// behind the barrier), so that the epilogue starts on landed data.  The tile after then starts with step 0 resident only and
// issues A(1) / the first half of B(1) itself.  The eight issues of sub-steps 1-2 are the step's youngest at its barrier
// either way (vmcnt(8)).  TAIL_EMPTY: a tail that issues nothing (the index scan: its filter needs the three units for staging).
struct G7NoTail { __device__ __forceinline__ void operator()(int, int, int, int) const {} };
// ZERO_FIRST (the index scan): the tile starts from zero -- the first sixteen MFMAs of its first step take the constant 0 as
// their C operand and `acc` need not be initialised (one more copy of the step body instead of sixteen initialising MFMAs).
template <typename T, bool TAIL = false, typename TailFn = G7NoTail, bool TAIL_EMPTY = false, bool ZERO_FIRST = false>
__device__ __forceinline__ void gemm_mainloop7_cont(const G7SrcU& src, const char* cur_a, const char* cur_b,
                                                    const char* next_a, const char* next_b, int nk, char* smem, G7Ring& ring,
                                                    f32x16_t (&acc)[4][4], unsigned long long* tr = nullptr, TailFn tail = TailFn()) {
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "128-byte K steps: 16-bit operands only");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 1) & 7;
  const int half = lane >> 5;
  int slot[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) slot[kk] = (((kk << 1) | half) ^ key) << 4;
  const int rowa = (wm * 128 + (lane & 31)) * G7_ROW_BYTES;
  const int rowb = (wn * 128 + (lane & 31)) * G7_ROW_BYTES;
  const uint32_t lds0 = g7_lds_addr(smem);
  const char* ka = cur_a + 2 * G7_ROW_BYTES;      // source of A(t + 2)
  const char* kb = cur_b + 2 * G7_ROW_BYTES;      // source of B(t + 2)
  const char* kbp = cur_b + G7_ROW_BYTES;         // source of B(t + 1): its second half is still to be issued
  int u_ac = ring.ac, u_bc = ring.bc, u_an = ring.an, u_bn = ring.bn, u_sp = ring.sp;
  frag_t a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const frag_t*)(smem + u_ac + rowa + i * 32 * G7_ROW_BYTES + slot[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b0[i] = *(const frag_t*)(smem + u_bc + rowb + i * 32 * G7_ROW_BYTES + slot[0]);

#define G7_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one k sub-step: 16 MFMAs from (AF, BF); the first eight each cover one fragment read into (AN, BN) from (UA, UB) chunk
  // SLOT; MFMAs 8, 10, 12, 14 each cover one DMA issue: operand P (source base PTR, instructions DBASE .. DBASE + 3) into
  // UNIT, or -- LASTSTEP with a tail slot -- tail(TSLOT .. TSLOT + 3).  No branch anywhere in a step: the restart-per-tile loop
  // above guards every issue with a wave-uniform flag, and those 16 scalar branches per step cost it ~800 of its ~3200 cycles
  // (the first version of the TAIL loop had 12 and ran 3209 cycles per step against 2404: profiles/r04_probe4_*).
#define G7C_SUB(AF, BF, AN, BN, UA, UB, SLOT, P, PTR, UNIT, DBASE, TSLOT, LASTSTEP, ZERO)                \
  _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                       \
    if (ZERO) { f32x16_t z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  \
                MmaOps<T>::mma(BF[q & 3], AF[q >> 2], z_); acc[q >> 2][q & 3] = z_; }                     \
    else MmaOps<T>::mma(BF[q & 3], AF[q >> 2], acc[q >> 2][q & 3]);                                      \
    if (q < 8) {                                                                                         \
      if (q < 4) AN[q] = *(const frag_t*)(smem + (UA) + rowa + q * 32 * G7_ROW_BYTES + (SLOT));          \
      else BN[q - 4] = *(const frag_t*)(smem + (UB) + rowb + (q - 4) * 32 * G7_ROW_BYTES + (SLOT));      \
    } else if (!(q & 1)) {                                                                               \
      if ((LASTSTEP) && (TSLOT) >= 0) tail((TSLOT) + ((q - 8) >> 1), u_sp, u_ac, u_bc);                        \
      else g7_issue_##P(src, PTR, (DBASE) + ((q - 8) >> 1), lds0 + (UNIT) + (((DBASE) + ((q - 8) >> 1)) * 4 + wave) * 1024); \
    }                                                                                                    \
    G7_FENCE();                                                                                          \
  }
#define G7C_STEP(LASTSTEP, ZERO)                                                                         \
  do {                                                                                                   \
    G7_STEP_STAMP();                                                                                     \
    G7C_SUB(a0, b0, a1, b1, u_ac, u_bc, slot[1], b, kbp, u_bn, 4, -1, LASTSTEP, ZERO)   /* second half of B(t+1) */ \
    G7C_SUB(a1, b1, a0, b0, u_ac, u_bc, slot[2], a, ka, u_sp, 0, 0, LASTSTEP, false)     /* A(t+2) */            \
    G7C_SUB(a0, b0, a1, b1, u_ac, u_bc, slot[3], a, ka, u_sp, 4, 4, LASTSTEP, false)                            \
    /* vmcnt(8) lgkmcnt(0): everything but the last eight issues has landed (an EMPTY tail issues nothing behind sub-steps 1-2: vmcnt(0)) */ \
    if ((LASTSTEP) && TAIL_EMPTY) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0x0078); \
    __builtin_amdgcn_s_barrier();                                                                        \
    G7_FENCE();                                                                                          \
    G7C_SUB(a1, b1, a0, b0, u_an, u_bn, slot[0], b, kb, u_ac, 0, 8, LASTSTEP, false)     /* first half of B(t+2) into the unit A(t) leaves */ \
    { const int o_ac = u_ac, o_bc = u_bc; u_ac = u_an; u_bc = u_bn; u_an = u_sp; u_bn = o_ac; u_sp = o_bc; } \
    kbp = kb;                                                                                            \
    if (t + 3 == nk) { ka = next_a; kb = next_b; } else { ka += G7_ROW_BYTES; kb += G7_ROW_BYTES; }      \
  } while (0)
  int t = 0;
  G7_LOOP_STAMP();
  const int nplain = TAIL ? nk - 1 : nk;
  if (ZERO_FIRST) { G7C_STEP(false, true); ++t; }
  for (; t < nplain; ++t) G7C_STEP(false, false);
  if (TAIL) G7C_STEP(true, false);                   // (one more copy of the step body: ~7 KiB of code, no branch inside either)
#undef G7C_STEP
#undef G7C_SUB
#undef G7_FENCE
  ring.ac = u_ac; ring.bc = u_bc; ring.an = u_an; ring.bn = u_bn; ring.sp = u_sp;
}

// ---- the continuous ring on 16 x 16 x 32 MFMAs (round 4) -----------------------------------------------------------------
// The chip is power-bound under these kernels (tile traces: the same cycle counts at 8 and at 256 active CUs, 2.4 vs 1.7 GHz),
// and MFMA-only loops sustain 2.15 PFLOP/s with this shape against 1.87 with 32 x 32 x 16 on random data (round 3:
// profiles/r03_mfma_shape_power_probe.log).  The branch-free K loop alone runs 0.5-9 % faster with it
// (profiles/r04_probe13_kloop_32x32x16_vs_16x16x32_continuous_ring.log).
//   acc[ti][fj][r] = C[m0 + wm*128 + ti*16 + (lane&15)][n0 + wn*128 + fj*16 + 4*(lane>>4) + r]
// Per step two sub-steps of 64 MFMAs; 16 fragment reads and 8 DMA issues each (one behind every eighth MFMA):
//     sub-step 0   k 0-31 of step t;  reads k 32-63;  A(t+2) -> spare
//     barrier      vmcnt(8): everything but A(t+2) has landed (B(t+1) is a step and a half old)
//     sub-step 1   k 32-63;  reads step t+1, k 0-31;  B(t+2) -> the unit A(t) leaves
// On entry: step 0 landed and published, A(1) and ALL of B(1) issued (the 32 x 32 x 16 loop above enters with half of B(1)).
// TAIL: the last step's sixteen issue slots go to tail(slot 0..15, spare unit, the unit A(nk - 1) leaves, the unit B(nk - 1) leaves) -- slots 0-7 behind
// sub-step 0 (the step's youngest at its barrier), 8-15 behind sub-step 1.
template <typename T> struct Mma16c;
template <> struct Mma16c<bf16_t> {
  __device__ static inline void mma(const bf16x8_t& a, const bf16x8_t& b, f32x4_t& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16c<f16_t> {
  __device__ static inline void mma(const f16x8_t& a, const f16x8_t& b, f32x4_t& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
// ZERO_FIRST (the index scan, round 6): the tile starts from zero -- the 64 MFMAs of the first sub-step take the constant 0 as their C operand
// and `acc` need not be initialised (one more copy of the step body instead of 256 accumulator writes).
template <typename T, bool TAIL = false, typename TailFn = G7NoTail, bool ZERO_FIRST = false>
__device__ __forceinline__ void gemm_mainloop7_cont16(const G7SrcU& src, const char* cur_a, const char* cur_b,
                                                      const char* next_a, const char* next_b, int nk, char* smem, G7Ring& ring,
                                                      f32x4_t (&acc)[8][8], unsigned long long* tr = nullptr, TailFn tail = TailFn()) {
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "128-byte K steps: 16-bit operands only");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 1) & 7;           // == ((row >> 1) & 7) for row = 16*x + (lane & 15)
  const int kb4 = lane >> 4;
  int slot[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) slot[kk] = (((kk << 2) | kb4) ^ key) << 4;
  const int rowa = (wm * 128 + (lane & 15)) * G7_ROW_BYTES;
  const int rowb = (wn * 128 + (lane & 15)) * G7_ROW_BYTES;
  const uint32_t lds0 = g7_lds_addr(smem);
  const char* ka = cur_a + 2 * G7_ROW_BYTES;
  const char* kb = cur_b + 2 * G7_ROW_BYTES;
  int u_ac = ring.ac, u_bc = ring.bc, u_an = ring.an, u_bn = ring.bn, u_sp = ring.sp;
  frag_t a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b0[i] = *(const frag_t*)(smem + u_bc + rowb + i * 16 * G7_ROW_BYTES + slot[0]);
#pragma unroll
  for (int i = 0; i < 8; ++i) a0[i] = *(const frag_t*)(smem + u_ac + rowa + i * 16 * G7_ROW_BYTES + slot[0]);
#define G7_FENCE() __builtin_amdgcn_sched_barrier(0)
  // 64 MFMAs from (AF, BF); every fourth covers one fragment read into (BN, then AN) from (UA, UB) chunk SLOT; MFMAs 5, 13, ..
  // each cover one DMA issue of operand P (PTR, instruction q >> 3) into UNIT -- or, in the last step of a TAIL loop, tail(TBASE + (q >> 3))
#define G7C_SUB16(AF, BF, AN, BN, UA, UB, SLOT, P, PTR, UNIT, TBASE, LASTSTEP, ZERO)                     \
  _Pragma("unroll") for (int q = 0; q < 64; ++q) {                                                       \
    if (ZERO) { f32x4_t z_ = {0.f, 0.f, 0.f, 0.f}; Mma16c<T>::mma(BF[q & 7], AF[q >> 3], z_); acc[q >> 3][q & 7] = z_; } \
    else Mma16c<T>::mma(BF[q & 7], AF[q >> 3], acc[q >> 3][q & 7]);                                      \
    if ((q & 3) == 0) {                                                                                  \
      if (q < 32) BN[q >> 2] = *(const frag_t*)(smem + (UB) + rowb + (q >> 2) * 16 * G7_ROW_BYTES + (SLOT)); \
      else AN[(q >> 2) - 8] = *(const frag_t*)(smem + (UA) + rowa + ((q >> 2) - 8) * 16 * G7_ROW_BYTES + (SLOT)); \
    }                                                                                                    \
    if ((q & 7) == 5) {                                                                                  \
      if (LASTSTEP) tail((TBASE) + (q >> 3), u_sp, u_ac, u_bc);                                                \
      else g7_issue_##P(src, PTR, q >> 3, lds0 + (UNIT) + ((q >> 3) * 4 + wave) * 1024);                 \
    }                                                                                                    \
    G7_FENCE();                                                                                          \
  }
#define G7C_STEP16(LASTSTEP, ZERO)                                                                       \
  do {                                                                                                   \
    G7_STEP_STAMP();                                                                                     \
    G7C_SUB16(a0, b0, a1, b1, u_ac, u_bc, slot[1], a, ka, u_sp, 0, LASTSTEP, ZERO)                       \
    /* vmcnt(8) lgkmcnt(0); a tail that issues nothing (the index scan) leaves the next tile's B(0) as the youngest issues: vmcnt(0) */ \
    if ((LASTSTEP) && std::is_same<TailFn, G7NoTail>::value) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0x0078); \
    __builtin_amdgcn_s_barrier();                                                                        \
    G7_FENCE();                                                                                          \
    G7C_SUB16(a1, b1, a0, b0, u_an, u_bn, slot[0], b, kb, u_ac, 8, LASTSTEP, false)                      \
    { const int o_ac = u_ac, o_bc = u_bc; u_ac = u_an; u_bc = u_bn; u_an = u_sp; u_bn = o_ac; u_sp = o_bc; } \
    if (t + 3 == nk) { ka = next_a; kb = next_b; } else { ka += G7_ROW_BYTES; kb += G7_ROW_BYTES; }      \
  } while (0)
  int t = 0;
  G7_LOOP_STAMP();
  const int nplain = TAIL ? nk - 1 : nk;
  if (ZERO_FIRST) { G7C_STEP16(false, true); ++t; }
  for (; t < nplain; ++t) G7C_STEP16(false, false);
  if (TAIL) G7C_STEP16(true, false);
#undef G7C_STEP16
#undef G7C_SUB16
#undef G7_FENCE
  ring.ac = u_ac; ring.bc = u_bc; ring.an = u_an; ring.bn = u_bn; ring.sp = u_sp;
}


#ifdef G7_M16_PROBE
// ---- the same K loop on 16 x 16 x 32 MFMAs (round 3) ------------------------------------------------------------------
// PROBE ONLY (tools/gemm7_probe.hip -DG7_ABL=4 -DG7_M16_PROBE; not compiled into the library).  Why it was tried: under
// the chip's power budget the matrix core sustains more with the small shape on random operands -- MFMA-only loops on
// register operands (tools/mfma_power_probe.hip, profiles/r03_mfma_shape_power_probe.log): 32x32x16 1.87 PFLOP/s,
// 16x16x32 2.15 (zeros: 2.49 / 2.44).  What it gave: the K loop alone 2.5-5 % faster (19.2 vs 20.0 us per K = 768 tile,
// 80.9 vs 83.1 at K = 3072; profiles/r03_gemm7_kloop_16x16x32_probe.log) -- the loop's limit is not the matrix core's
// share of the power.  Not worth re-deriving every epilogue for the other accumulator layout; kept for the record.
//   acc[ti][fj][r] = C[m0 + wm*128 + ti*16 + (lane&15)][n0 + wn*128 + fj*16 + 4*(lane>>4) + r]
// A fragment = 16 rows x 32 k: lane -> row (lane & 15), 16-byte k block (lane >> 4) of the sub-step's four; a 16-lane
// group reads 16 consecutive rows at one k block: with chunk position XOR ((row >> 1) & 7) again 16 distinct slots.
// Per wave and step: two sub-steps of 64 MFMAs; 16 ds_read_b128 and 8 DMA instructions per sub-step:
//     sub-step 0   computes k 0-31 of step t; reads the fragments of k 32-63; issues A(t+2) -> spare
//     barrier      vmcnt(8): step t+1 has landed (only A(t+2) is younger); every wave has read its last fragment of step t
//     sub-step 1   computes k 32-63; reads the fragments of step t+1, k 0-31; issues B(t+2) -> A_cur
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
  __device__ static inline void mma(const bf16x8_t& a, const bf16x8_t& b, f32x4_t& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16<f16_t> {
  __device__ static inline void mma(const f16x8_t& a, const f16x8_t& b, f32x4_t& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename T, typename SRC = G7Src>
__device__ __forceinline__ void gemm_mainloop7_run16(const SRC& src, int nk, char* smem,
                                            f32x4_t (&acc)[8][8], unsigned long long* tr = nullptr,
                                            bool stores_pending = false) {
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "128-byte K steps: 16-bit operands only");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 1) & 7;           // == ((row >> 1) & 7) for row = 16*x + (lane & 15)
  const int kb4 = lane >> 4;
  int slot[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) slot[kk] = (((kk << 2) | kb4) ^ key) << 4;
  const int rowa = (wm * 128 + (lane & 15)) * G7_ROW_BYTES;
  const int rowb = (wn * 128 + (lane & 15)) * G7_ROW_BYTES;

  if (stores_pending) { if (nk > 1) __builtin_amdgcn_s_waitcnt(0xC070); else __builtin_amdgcn_s_waitcnt(0x8070); }   // vmcnt(48) / vmcnt(32)
  else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x4070);      // vmcnt(16) lgkmcnt(0)
  else __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0)
  __builtin_amdgcn_s_barrier();

  const uint32_t lds0 = g7_lds_addr(smem);
  const char* ka = src.a + 2 * G7_ROW_BYTES;      // K offset of step t + 2 (scalar registers)
  const char* kb = src.b + 2 * G7_ROW_BYTES;
  int u_ac = 0, u_bc = G7_UNIT_BYTES, u_an = 2 * G7_UNIT_BYTES, u_bn = 3 * G7_UNIT_BYTES, u_sp = 4 * G7_UNIT_BYTES;
  frag_t a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b0[i] = *(const frag_t*)(smem + u_bc + rowb + i * 16 * G7_ROW_BYTES + slot[0]);
#pragma unroll
  for (int i = 0; i < 8; ++i) a0[i] = *(const frag_t*)(smem + u_ac + rowa + i * 16 * G7_ROW_BYTES + slot[0]);

#define G7_FENCE() __builtin_amdgcn_sched_barrier(0)
#define G7_DMA(P, I, UNIT) do { if (!(G7_ABL & 8)) g7_issue_##P(src, k##P, I, lds0 + (UNIT) + ((I) * 4 + wave) * 1024); } while (0)
  // one k sub-step: 64 MFMAs from (AF, BF); every fourth covers one fragment read into (BN, then AN) from (UA, UB)
  // chunk SLOT; MFMAs 5, 13, .. each cover one DMA issue of operand P into UNIT when the wave-uniform COND holds
#define G7_SUB16(AF, BF, AN, BN, UA, UB, SLOT, P, UNIT, COND)                                            \
  _Pragma("unroll") for (int q = 0; q < 64; ++q) {                                                       \
    if (!(G7_ABL & 32)) Mma16<T>::mma(BF[q & 7], AF[q >> 3], acc[q >> 3][q & 7]);                        \
    if ((q & 3) == 0 && !(G7_ABL & 16)) {                                                                \
      if (q < 32) BN[q >> 2] = *(const frag_t*)(smem + (UB) + rowb + (q >> 2) * 16 * G7_ROW_BYTES + (SLOT)); \
      else AN[(q >> 2) - 8] = *(const frag_t*)(smem + (UA) + rowa + ((q >> 2) - 8) * 16 * G7_ROW_BYTES + (SLOT)); \
    }                                                                                                    \
    if ((q & 7) == 5) { if (COND) G7_DMA(P, q >> 3, UNIT); }                                             \
    G7_FENCE();                                                                                          \
  }
  for (int t = 0; t < nk; ++t) {
    const bool issue = t + 2 < nk;
    if (tr && tid == 0 && t < 12) tr[3 + t] = clock64();
    G7_SUB16(a0, b0, a1, b1, u_ac, u_bc, slot[1], a, u_sp, issue)
    if (issue) __builtin_amdgcn_s_waitcnt(0x0078); else __builtin_amdgcn_s_waitcnt(0x0070);
    if (!(G7_ABL & 64)) __builtin_amdgcn_s_barrier();
    G7_FENCE();
    G7_SUB16(a1, b1, a0, b0, u_an, u_bn, slot[0], b, u_ac, issue)
    { const int o_ac = u_ac, o_bc = u_bc; u_ac = u_an; u_bc = u_bn; u_an = u_sp; u_bn = o_ac; u_sp = o_bc; }
    ka += G7_ROW_BYTES; kb += G7_ROW_BYTES;
  }
#undef G7_SUB16
#undef G7_DMA
#undef G7_FENCE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                             // everyone is done with the ring
}
#endif  // G7_M16_PROBE
