// Second-generation NT GEMM main loop for gfx950: 256 x 128 output tile, 8 waves (4 x 2, each
// 64 x 64 = 2x2 MFMA 32x32 tiles), K walked in 128-byte steps through a 3-deep LDS ring.
//
// Why (measured on MI355X, profiles/r01_*): the 128x128 / 2-stage loop of gemm_core.h drains
// its LDS-DMA (vmcnt(0)) at every K step, so each step costs one full memory latency (~2 us
// under load) for 0.2 us of MFMA work; at the encoder's K = 768 (12 steps) it reaches 17 % of
// the bf16 MFMA peak.  Here the DMA of tile t+2 is issued right after the barrier of step t and
// only `vmcnt(6)` (= "everything but the newest tile") is waited for, so one tile is always in
// flight ACROSS the barrier; one workgroup per CU owns 144 KiB of LDS and keeps 2 waves per
// SIMD, and the epilogue goes through LDS so global stores are whole 16-byte row segments.
//
//   ring slot = A tile [256 rows][128 B] (32 KiB) + B tile [128 rows][128 B] (16 KiB)
//   per wave and K step: 4 + 2 global_load_lds_dwordx4 (1 KiB each)  -> vmcnt quantum = 6
//   16-byte slot swizzle as in gemm_core.h: physical = logical ^ ((row >> 1) & 7)
#pragma once
#include "gemm_core.h"

#define G2_BM 256
#define G2_BN 128
#define G2_THREADS 512
#define G2_A_BYTES (256 * 128)
#define G2_B_BYTES (128 * 128)
#define G2_STAGE_BYTES (G2_A_BYTES + G2_B_BYTES)
#define G2_STAGES 3
#define G2_LDS_BYTES (G2_STAGES * G2_STAGE_BYTES)

__device__ inline void g2_stage(const char* const (&pa)[4], const char* const (&pb)[2], size_t kbyte,
                                char* slot, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + kbyte), (lptr_t)(slot + (i * 8 + wave) * 1024),
                                     16, 0, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + kbyte),
                                     (lptr_t)(slot + G2_A_BYTES + (i * 8 + wave) * 1024), 16, 0, 0);
}

template <typename T>
__device__ inline void gemm_mainloop2(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                      int64_t ldb, int64_t M, int64_t N, int64_t K, int64_t m0,
                                      int64_t n0, char* smem, f32x16_t (&acc)[2][2],
                                      unsigned long long* tr = nullptr) {
  typedef typename MmaOps<T>::frag_t frag_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 1, wn = wave & 1;

  const char* pa[4];
  const char* pb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t ra = m0 + r; if (ra > M - 1) ra = M - 1;
    pa[i] = (const char*)(A + ra * lda) + c * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t rb = n0 + r; if (rb > N - 1) rb = N - 1;
    pb[i] = (const char*)(B + rb * ldb) + c * 16;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = (int)((K * (int64_t)sizeof(T)) / GEMM_ROW_BYTES);
  const int key = (lane >> 1) & 7;
  const int half = lane >> 5;
  const int rowa = (wm * 64 + (lane & 31)) * GEMM_ROW_BYTES;
  const int rowb = G2_A_BYTES + (wn * 64 + (lane & 31)) * GEMM_ROW_BYTES;

  char* s_cur = smem;                        // slot of tile t
  char* s_nxt = smem + G2_STAGE_BYTES;       // slot of tile t+1
  char* s_far = smem + 2 * G2_STAGE_BYTES;   // slot of tile t+2 (== slot of tile t-1)
  if (tr && tid == 0) tr[1] = clock64();
  g2_stage(pa, pb, 0, s_cur, wave);
  if (nk > 1) g2_stage(pa, pb, GEMM_ROW_BYTES, s_nxt, wave);

  for (int t = 0; t < nk; ++t) {
    // tile t has landed for THIS wave once at most the newest tile's 6 DMAs are outstanding
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS reads of tile t-1 are done
    __builtin_amdgcn_s_barrier();                         // -> true for every wave
    if (tr && tid == 0 && t < 13) tr[2 + t] = clock64();
    if (t + 2 < nk) g2_stage(pa, pb, (size_t)(t + 2) * GEMM_ROW_BYTES, s_far, wave);
    // register double-buffering of the fragments: the LDS reads of sub-step kk+1 are in flight
    // while the MFMAs of sub-step kk issue (LDS latency under load is several hundred cycles)
    frag_t a0[2], a1[2], b0[2], b1[2];
    {
      const int slot = ((0 | half) ^ key) << 4;
      a0[0] = *(const frag_t*)(s_cur + rowa + slot);
      a1[0] = *(const frag_t*)(s_cur + rowa + 32 * GEMM_ROW_BYTES + slot);
      b0[0] = *(const frag_t*)(s_cur + rowb + slot);
      b1[0] = *(const frag_t*)(s_cur + rowb + 32 * GEMM_ROW_BYTES + slot);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);        // pin: 4 LDS reads (sub-step 0) first
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cb = kk & 1, nb = cb ^ 1;
      if (kk < 3) {
        const int slot = ((((kk + 1) << 1) | half) ^ key) << 4;
        a0[nb] = *(const frag_t*)(s_cur + rowa + slot);
        a1[nb] = *(const frag_t*)(s_cur + rowa + 32 * GEMM_ROW_BYTES + slot);
        b0[nb] = *(const frag_t*)(s_cur + rowb + slot);
        b1[nb] = *(const frag_t*)(s_cur + rowb + 32 * GEMM_ROW_BYTES + slot);
      }
      MmaOps<T>::mma(a0[cb], b0[cb], acc[0][0]);
      MmaOps<T>::mma(a0[cb], b1[cb], acc[0][1]);
      MmaOps<T>::mma(a1[cb], b0[cb], acc[1][0]);
      MmaOps<T>::mma(a1[cb], b1[cb], acc[1][1]);
      // pin the issue order: the next sub-step's 4 LDS reads go out BEFORE this sub-step's MFMAs
      // (hipcc otherwise re-serialises read -> wait -> MFMA to save registers)
      if (kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MmaOps<T>::kMfmaPerMma * 4, 0);
    }
    char* tmp = s_cur; s_cur = s_nxt; s_nxt = s_far; s_far = tmp;
  }
}

// Same work-id -> tile mapping as gemm_tile_coords, for 256-row tiles.
__device__ inline void g2_tile_coords(int64_t M, int64_t N, int group_m, int64_t& m0, int64_t& n0) {
  const int64_t ntm = (M + G2_BM - 1) / G2_BM, ntn = (N + G2_BN - 1) / G2_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  m0 = tm * G2_BM;
  n0 = tn * G2_BN;
}
