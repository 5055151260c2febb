// Sixth-generation NT GEMM main loop for gfx950: 256 x 256 tile, FOUR waves (one per SIMD), each
// owning a 128 x 128 accumulator block (16 MFMA tiles = 256 accumulator registers).
//
// Why (profiles/r01_gemm_variants_trace.log): with eight waves of 128 x 64 (the retired generations 3 and 4) a
// 64-byte K step reads 8 x 12 KiB of fragments out of LDS and the DMA writes 32 KiB into it:
// 128 KiB per step against an LDS port of 128 B/clk is 1024 cycles -- exactly the MFMA time of the
// step, so the LDS is a co-bottleneck and measured steps take ~1750 cycles.  A 128 x 128 wave tile
// reads (128 + 128) rows for twice the outputs: 4 x 16 KiB + 32 KiB = 96 KiB per step (75 % of the
// port).  With one wave per SIMD nothing else hides latency, so the loop is software pipelined:
// the fragments of sub-step kk+1 are read while the 16 MFMAs of sub-step kk issue.
//
//   stage = A tile [256 rows][64 B] + B tile [256 rows][64 B], ring of G6_STAGES
//   per wave and stage: 4 + 4 global_load_lds_dwordx4 (vmcnt counts 8 per stage)
#pragma once
#include "gemm_core.h"

// stage = A tile [256 rows][64 B] (16 KiB) + B tile [256 rows][64 B] (16 KiB); one global_load_lds_dwordx4 wave
// instruction = 1 KiB = 16 rows.  Swizzle for 64-byte rows: physical 16-B slot = logical ^ ((row >> 2) & 3) (a 256-B
// bank row holds 4 tile rows; every ds_read_b128 lane group then covers 16 distinct slots).
#define G6_BM 256
#define G6_BN 256
#define G6_ROW_BYTES 64
#define G6_OPERAND_BYTES (256 * 64)
#define G6_STAGE_BYTES (2 * G6_OPERAND_BYTES)

#define G6_THREADS 256
// Ring depth: 4 x 32 KiB, three K steps in flight (2.5 steps ~ 2.8 k cycles of lead).  5 slots (the
// whole 160 KiB LDS, 3.5 steps of lead) were measured and are SLOWER (K step 1489 -> 1520 cycles at
// M = 131072, N = K = 768; encoder shapes -3..-15 %): the K loop's stalls on streamed operands are a
// throughput limit of the L2 / fabric path, not a latency one, and more requests in flight only
// queue.  -DG6_STAGES=5 rebuilds that variant.
#ifndef G6_STAGES
#define G6_STAGES 4
#endif
#define G6_AHEAD (G6_STAGES - 1)
#define G6_LDS_BYTES (G6_STAGES * G6_STAGE_BYTES)

__device__ inline void g6_stage(const char* const (&pa)[4], const char* const (&pb)[4], size_t kbyte,
                                char* slot, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + kbyte), (lptr_t)(slot + (i * 4 + wave) * 1024),
                                     16, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + kbyte),
                                     (lptr_t)(slot + G6_OPERAND_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ void g6_read(typename MmaOps<T>::frag_t (&a)[4], typename MmaOps<T>::frag_t (&b)[4],
                                        const char* cur, int rowa, int rowb, int slot) {
  typedef typename MmaOps<T>::frag_t frag_t;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = *(const frag_t*)(cur + rowa + i * 32 * G6_ROW_BYTES + slot);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = *(const frag_t*)(cur + rowb + i * 32 * G6_ROW_BYTES + slot);
}

// The MFMA operands are SWAPPED (B tile rows as the A operand), so a lane owns one output ROW and
// four consecutive accumulator registers are four consecutive output COLUMNS (wm = wave>>1, wn = wave&1):
//   acc[mi][ni][r] = C[m0 + wm*128 + mi*32 + (lane&31)][n0 + wn*128 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
// -- the layout gemm_epilogue6.h streams out.  acc[.][ni] starts at init[ni] (the bias in that order, or 0).
// PROBE (tools/gemm_loop_probe.hip only; 0 in the product): bit 0 drops the steady-state DMA issue,
// bit 1 the fragment reads, bit 2 the per-step barrier -- to attribute the cycles of a K step.
// Per-lane DMA source pointers of tile (m0, n0): 4 + 4 rows, swizzled chunk (swizzle above).
template <typename T>
__device__ __forceinline__ void g6_point(const char* (&pa)[4], const char* (&pb)[4], const T* __restrict__ A,
                                         int64_t lda, const T* __restrict__ B, int64_t ldb, int64_t M, int64_t N,
                                         int64_t m0, int64_t n0, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    int64_t ra = m0 + r; if (ra > M - 1) ra = M - 1;
    int64_t rb = n0 + r; if (rb > N - 1) rb = N - 1;
    pa[i] = (const char*)(A + ra * lda) + c * 16;
    pb[i] = (const char*)(B + rb * ldb) + c * 16;
  }
}

// Start a tile: K steps 0 and 1 go into ring slots 0 and 1.  The persistent kernel calls this for the
// NEXT tile before it runs the epilogue of the current one.
__device__ __forceinline__ void g6_begin(const char* const (&pa)[4], const char* const (&pb)[4], int nk,
                                         char* smem, int wave) {
  g6_stage(pa, pb, 0, smem, wave);
  if (nk > 1) g6_stage(pa, pb, G6_ROW_BYTES, smem + G6_STAGE_BYTES, wave);
}

// The K loop of one tile whose steps 0 and 1 are already in flight (g6_begin).  Issues step 2, then
// waits for everything older (steps 0, 1 -- and, in the persistent kernel, the previous tile's
// epilogue stores: vmcnt retires in order).  acc must be initialised by the caller.
template <typename T, int PROBE = 0>
__device__ inline void gemm_mainloop6_run(const char* (&pa)[4], const char* (&pb)[4], int nk, char* smem,
                                          f32x16_t (&acc)[4][4], unsigned long long* tr = nullptr) {
  typedef typename MmaOps<T>::frag_t frag_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 2) & 3;
  const int half = lane >> 5;
  const int slot0 = ((half ^ key) << 4), slot1 = (((2 | half) ^ key) << 4);
  const int rowa = (wm * 128 + (lane & 31)) * G6_ROW_BYTES;
  const int rowb = G6_OPERAND_BYTES + (wn * 128 + (lane & 31)) * G6_ROW_BYTES;

  // steps 2 .. G6_AHEAD-1 join steps 0, 1; then wait until only those newer than step 0 are outstanding
  const int pre = nk < G6_AHEAD ? nk : G6_AHEAD;
  for (int sidx = 2; sidx < pre; ++sidx) g6_stage(pa, pb, (size_t)sidx * G6_ROW_BYTES, smem + sidx * G6_STAGE_BYTES, wave);
  // s_waitcnt immediates (gfx9 encoding: vmcnt = [15:14][3:0], expcnt [6:4] = 7 "no wait", lgkmcnt [11:8] = 0)
  if (pre >= 4) __builtin_amdgcn_s_waitcnt(0x4078);        // vmcnt(24)
  else if (pre == 3) __builtin_amdgcn_s_waitcnt(0x4070);   // vmcnt(16)
  else if (pre == 2) __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8)
  else __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();

  frag_t a0[4], b0[4], a1[4], b1[4];
  g6_read<T>(a0, b0, smem, rowa, rowb, slot0);
  if (PROBE & 2) g6_read<T>(a1, b1, smem, rowa, rowb, slot1);

  // One K step.  ISSUE: start the DMA of tile t+3; VMW: s_waitcnt immediate that proves tile t+1
  // has landed; NEXT: read tile t+1's first fragments.  All compile-time, so the steady-state loop
  // is straight-line code.  Issue order is pinned by hand (sched_barrier fences): every MFMA covers
  // at most one LDS read or one DMA issue -- eight back-to-back global_load_lds cost ~320 cycles of
  // a 1024-cycle step when they sit in front of the MFMAs (tools/gemm_loop_probe.hip).
#define G6_FENCE() __builtin_amdgcn_sched_barrier(0)
#define G6_DMA(P, I, OFF)                                                                                \
  __builtin_amdgcn_global_load_lds((gptr_t)(P[I] + (size_t)(t + G6_AHEAD) * G6_ROW_BYTES),               \
                                   (lptr_t)(smem + o_far + (OFF) + ((I) * 4 + wave) * 1024), 16, 0, 0)
#define G6_HALF(AF, BF, AN, BN, SRC, SLOT, DO_READ, DO_DMA, P, OFF)                                      \
  _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                       \
    MmaOps<T>::mma(BF[q & 3], AF[q >> 2], acc[q >> 2][q & 3]);                                           \
    if (q < 8) {                                                                                         \
      if (DO_READ) {                                                                                     \
        if (q < 4) AN[q] = *(const frag_t*)((SRC) + rowa + q * 32 * G6_ROW_BYTES + (SLOT));              \
        else BN[q - 4] = *(const frag_t*)((SRC) + rowb + (q - 4) * 32 * G6_ROW_BYTES + (SLOT));          \
      }                                                                                                  \
    } else if (!(q & 1)) {                                                                               \
      if (DO_DMA) G6_DMA(P, (q - 8) >> 1, OFF);                                                          \
    }                                                                                                    \
    G6_FENCE();                                                                                          \
  }
#define G6_PIN() _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) asm volatile("" : "+a"(acc[q_ >> 2][q_ & 3]))
#define G6_STEP(ISSUE, VMW, NEXT)                                                                        \
  do {                                                                                                   \
    const char* cur = smem + o_cur;                                                                      \
    const char* nxt = smem + o_nxt;                                                                      \
    if (tr && tid == 0 && t < 12) tr[3 + t] = clock64();                                                 \
    G6_PIN();                                                                                            \
    /* slot o_far held tile t-1: every wave finished reading it before the barrier of step t-1 */        \
    G6_HALF(a0, b0, a1, b1, cur, slot1, !(PROBE & 2), (ISSUE) && !(PROBE & 1), pa, 0)                    \
    G6_PIN();                                                                                            \
    __builtin_amdgcn_s_waitcnt(VMW);     /* vmcnt(n) lgkmcnt(0): tile t+1 landed, my reads done */       \
    if (!(PROBE & 4)) __builtin_amdgcn_s_barrier();                                                      \
    G6_FENCE();                                                                                          \
    G6_HALF(a1, b1, a0, b0, nxt, slot0, (NEXT) && !(PROBE & 2), (ISSUE) && !(PROBE & 1), pb, G6_OPERAND_BYTES) \
    /* the accumulators' home is the AGPR file: left alone, the register allocator carries one of the 16 tiles in     \
       VGPRs across the loop's back edge and copies it in and out around its two MFMAs of every step (32-48 v_accvgpr    \
       moves per K step that also wait for the matrix core -- found by scanning every MFMA loop of the library) */     \
    G6_PIN();                                                                                            \
    o_cur = o_nxt;                                                                                       \
    o_nxt = o_nxt + G6_STAGE_BYTES == G6_LDS_BYTES ? 0 : o_nxt + G6_STAGE_BYTES;                         \
    o_far = o_far + G6_STAGE_BYTES == G6_LDS_BYTES ? 0 : o_far + G6_STAGE_BYTES;                         \
  } while (0)

  int t = 0;
  int o_cur = 0, o_nxt = G6_STAGE_BYTES, o_far = G6_AHEAD * G6_STAGE_BYTES;   // ring offsets of steps t, t+1, t+G6_AHEAD
#if G6_STAGES == 5
  for (; t + 4 < nk; ++t) G6_STEP(true, 0x4074, true);      // vmcnt(20): t+2, t+3 and the A half of t+4 may be in flight
  if (t + 3 < nk) { G6_STEP(false, 0x4070, true); ++t; }    // vmcnt(16): t+2, t+3
#else
  for (; t + 3 < nk; ++t) G6_STEP(true, 0x007C, true);      // vmcnt(12): t+2 and the A half of t+3
#endif
  if (t + 2 < nk) { G6_STEP(false, 0x0078, true); ++t; }    // vmcnt(8):  t+2
  if (t + 1 < nk) { G6_STEP(false, 0x0070, true); ++t; }    // vmcnt(0)
  G6_STEP(false, 0x0070, false);                            // last tile
#undef G6_STEP
#undef G6_PIN
#undef G6_HALF
#undef G6_DMA
#undef G6_FENCE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                             // everyone is done with the ring
}

// One tile start to finish (the non-persistent kernels and the probes).
template <typename T, int PROBE = 0>
__device__ inline void gemm_mainloop6(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                      int64_t ldb, int64_t M, int64_t N, int64_t K, int64_t m0,
                                      int64_t n0, char* smem, f32x16_t (&acc)[4][4],
                                      const f32x16_t (&init)[4], unsigned long long* tr = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* pa[4];
  const char* pb[4];
  g6_point<T>(pa, pb, A, lda, B, ldb, M, N, m0, n0, wave, lane);
  const int nk = (int)((K * (int64_t)sizeof(T)) / G6_ROW_BYTES);
  if (tr && threadIdx.x == 0) tr[1] = clock64();
  g6_begin(pa, pb, nk, smem, wave);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = init[ni][r];   // the columns' bias (or 0)
  gemm_mainloop6_run<T, PROBE>(pa, pb, nk, smem, acc, tr);
}


__device__ inline void g6_tile_coords(int64_t M, int64_t N, int group_m, int64_t& m0, int64_t& n0) {
  const int64_t ntm = (M + G6_BM - 1) / G6_BM, ntn = (N + G6_BN - 1) / G6_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  m0 = tm * G6_BM;
  n0 = tn * G6_BN;
}
