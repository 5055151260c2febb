// Third-generation NT GEMM main loop for gfx950: 256 x 256 output tile, 8 waves (2 x 4, each
// 128 x 64 = 4x2 MFMA 32x32 tiles, 128 accumulator VGPRs), K walked in 128-byte steps through a
// double-buffered 2 x 64 KiB LDS image.
//
// Why (profiles/r01_pmc_gemm_v2): with 1 KiB LDS-DMA pieces and ~100 KiB of them in flight per
// CU the L2 -> LDS path delivers ~20 B/clk/CU, and a tile needs (BM+BN)/(BM*BN) bytes per FLOP:
//   128x128 -> 64 B/clk/CU at MFMA peak,  256x128 -> 48,  256x256 -> 32.
// The square 256 tile is the first whose operand traffic fits under what the memory system
// sustains, it halves the LDS fragment reads per MFMA (6 reads : 8 MFMAs per sub-step instead of
// 4 : 4), and at the encoder's K = 768 it doubles the work that amortises a tile's prologue and
// epilogue.  Same 16-byte slot swizzle as gemm_core.h: physical = logical ^ ((row >> 1) & 7).
#pragma once
#include "gemm_core.h"

#define G3_BM 256
#define G3_BN 256
#define G3_THREADS 512
#define G3_OPERAND_BYTES (256 * 128)
#define G3_STAGE_BYTES (2 * G3_OPERAND_BYTES)
#define G3_LDS_BYTES (2 * G3_STAGE_BYTES)

__device__ inline void g3_stage(const char* const (&pa)[4], const char* const (&pb)[4], size_t kbyte,
                                char* slot, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + kbyte), (lptr_t)(slot + (i * 8 + wave) * 1024),
                                     16, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + kbyte),
                                     (lptr_t)(slot + G3_OPERAND_BYTES + (i * 8 + wave) * 1024), 16, 0, 0);
}

// acc[mi][ni]: rows m0 + wm*128 + mi*32 + .., cols n0 + wn*64 + ni*32 + ..  (wm = wave>>2, wn = wave&3)
template <typename T>
__device__ inline void gemm_mainloop3(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                      int64_t ldb, int64_t M, int64_t N, int64_t K, int64_t m0,
                                      int64_t n0, char* smem, f32x16_t (&acc)[4][2],
                                      unsigned long long* tr = nullptr) {
  typedef typename MmaOps<T>::frag_t frag_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 2, wn = wave & 3;

  const char* pa[4];
  const char* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t ra = m0 + r; if (ra > M - 1) ra = M - 1;
    int64_t rb = n0 + r; if (rb > N - 1) rb = N - 1;
    pa[i] = (const char*)(A + ra * lda) + c * 16;
    pb[i] = (const char*)(B + rb * ldb) + c * 16;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = (int)((K * (int64_t)sizeof(T)) / GEMM_ROW_BYTES);
  const int key = (lane >> 1) & 7;
  const int half = lane >> 5;
  const int rowa = (wm * 128 + (lane & 31)) * GEMM_ROW_BYTES;
  const int rowb = G3_OPERAND_BYTES + (wn * 64 + (lane & 31)) * GEMM_ROW_BYTES;

  if (tr && tid == 0) tr[1] = clock64();
  g3_stage(pa, pb, 0, smem, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (tr && tid == 0) tr[2] = clock64();

  for (int t = 0; t < nk; ++t) {
    const char* cur = smem + (t & 1) * G3_STAGE_BYTES;
    if (t + 1 < nk)
      g3_stage(pa, pb, (size_t)(t + 1) * GEMM_ROW_BYTES, smem + ((t + 1) & 1) * G3_STAGE_BYTES, wave);
    // fragments are double-buffered in registers: sub-step kk+1's LDS reads fly under kk's MFMAs
    frag_t a[2][4], b[2][2];
    {
      const int slot = ((0 | half) ^ key) << 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[0][mi] = *(const frag_t*)(cur + rowa + mi * 32 * GEMM_ROW_BYTES + slot);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[0][ni] = *(const frag_t*)(cur + rowb + ni * 32 * GEMM_ROW_BYTES + slot);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cb = kk & 1, nb = cb ^ 1;
      if (kk < 3) {
        const int slot = ((((kk + 1) << 1) | half) ^ key) << 4;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[nb][mi] = *(const frag_t*)(cur + rowa + mi * 32 * GEMM_ROW_BYTES + slot);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[nb][ni] = *(const frag_t*)(cur + rowb + ni * 32 * GEMM_ROW_BYTES + slot);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) MmaOps<T>::mma(a[cb][mi], b[cb][ni], acc[mi][ni]);
      if (kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MmaOps<T>::kMfmaPerMma * 8, 0);
    }
    // next stage landed (all of this wave's DMA) and this wave's reads of `cur` are done
    if (tr && tid == 0 && t < 12) tr[16 + t] = clock64();      // MFMAs issued, before the drain
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tr && tid == 0 && t < 12) tr[3 + t] = clock64();
  }
}

__device__ inline void g3_tile_coords(int64_t M, int64_t N, int group_m, int64_t& m0, int64_t& n0) {
  const int64_t ntm = (M + G3_BM - 1) / G3_BM, ntn = (N + G3_BN - 1) / G3_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  m0 = tm * G3_BM;
  n0 = tn * G3_BN;
}
