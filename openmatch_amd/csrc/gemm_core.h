// NT GEMM main loop for gfx950:  acc[128x128] = A[m0:m0+128, :K] · B[n0:n0+128, :K]^T
//
// One workgroup = 4 waves (2x2), each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32
// tiles (64 accumulator VGPRs).  K is walked in 128-BYTE steps (64 bf16 / 32 f32)
// so both element types share one LDS geometry:
//
//   stage (2x, double buffered) = A tile [128 rows][128 B] + B tile [128 rows][128 B] = 32 KiB
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4): one wave instruction moves
//   1 KiB = 8 tile rows; the LDS image is lane-linear, so the bank-conflict swizzle
//   is applied to the per-lane SOURCE address and again on the fragment read:
//       physical 16-B slot = logical slot ^ ((row >> 1) & 7)
//   which makes every ds_read_b128 lane group hit 16 distinct slots of the 256-B bank row.
//
// MFMA operand maps (gfx950):
//   v_mfma_f32_32x32x16_bf16 : A lane l holds A[i=l&31][k=(l>>5)*8 .. +7], B likewise with j=l&31
//   v_mfma_f32_32x32x2_f32   : A lane l holds A[i=l&31][k=l>>5]
//   C/D                      : col j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
// The assignment of k-slots to lanes only has to be the SAME for A and B (a dot product
// is order-free), which is what lets f32 fragments be read as 16-byte chunks too.
#pragma once
#include "common.h"

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_ROW_BYTES 128
#define GEMM_OPERAND_BYTES (128 * 128)
#define GEMM_STAGE_BYTES (2 * GEMM_OPERAND_BYTES)
#define GEMM_LDS_BYTES (2 * GEMM_STAGE_BYTES)
#define GEMM_THREADS 256

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct MmaOps;

template <> struct MmaOps<bf16_t> {
  typedef bf16x8_t frag_t;
  static constexpr int kMfmaPerMma = 1;
  __device__ static inline void mma(const frag_t& a, const frag_t& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaOps<f16_t> {
  typedef f16x8_t frag_t;
  static constexpr int kMfmaPerMma = 1;
  __device__ static inline void mma(const frag_t& a, const frag_t& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaOps<float> {
  typedef f32x4_t frag_t;
  static constexpr int kMfmaPerMma = 4;
  __device__ static inline void mma(const frag_t& a, const frag_t& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
  }
};

// Issue the LDS-DMA for one 128-byte K step of both operand tiles.
__device__ inline void gemm_stage(const char* const (&pa)[4], const char* const (&pb)[4],
                                  size_t kbyte, char* stage_base, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = i * 4 + wave;
    __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + kbyte), (lptr_t)(stage_base + ci * 1024), 16,
                                     0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = i * 4 + wave;
    __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + kbyte),
                                     (lptr_t)(stage_base + GEMM_OPERAND_BYTES + ci * 1024), 16, 0,
                                     0);
  }
}

// acc[mi][ni] accumulates the wave's 64x64 sub-tile.  Rows past M / N are clamped to
// the last valid row when loading (their results must be discarded by the caller).
template <typename T>
__device__ inline void gemm_mainloop(const T* __restrict__ A, int64_t lda,
                                     const T* __restrict__ B, int64_t ldb, int64_t M, int64_t N,
                                     int64_t K, int64_t m0, int64_t n0, char* smem,
                                     f32x16_t (&acc)[2][2], int64_t kbyte0 = 0, int nk_limit = -1) {
  typedef typename MmaOps<T>::frag_t frag_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // per-lane source pointers for the 4 DMA rounds of each operand
  const char* pa[4];
  const char* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = i * 4 + wave;
    const int r = ci * 8 + (lane >> 3);              // tile row this lane feeds
    const int c = (lane & 7) ^ ((r >> 1) & 7);       // logical 16-B chunk stored at slot lane&7
    int64_t ra = m0 + r; if (ra > M - 1) ra = M - 1;
    int64_t rb = n0 + r; if (rb > N - 1) rb = N - 1;
    pa[i] = (const char*)(A + ra * lda) + c * 16;
    pb[i] = (const char*)(B + rb * ldb) + c * 16;
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  int nk = (int)((K * (int64_t)sizeof(T) - kbyte0) / GEMM_ROW_BYTES);   // K range [kbyte0, kbyte0 + nk*128)
  if (nk_limit >= 0 && nk > nk_limit) nk = nk_limit;
  const int key = (lane >> 1) & 7;   // == ((row>>1)&7) for row = 32*x + (lane&31)
  const int half = lane >> 5;
  const int rowa = (wm * 64 + (lane & 31)) * GEMM_ROW_BYTES;
  const int rowb = (wn * 64 + (lane & 31)) * GEMM_ROW_BYTES;

  gemm_stage(pa, pb, (size_t)kbyte0, smem, wave);
  __syncthreads();  // drains the DMA (vmcnt(0)) and publishes stage 0

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * GEMM_STAGE_BYTES;
    if (t + 1 < nk)
      gemm_stage(pa, pb, (size_t)kbyte0 + (size_t)(t + 1) * GEMM_ROW_BYTES,
                 smem + ((t + 1) & 1) * GEMM_STAGE_BYTES, wave);
    const char* sA = cur;
    const char* sB = cur + GEMM_OPERAND_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int slot = (((kk << 1) | half) ^ key) << 4;
      frag_t a0 = *(const frag_t*)(sA + rowa + slot);
      frag_t a1 = *(const frag_t*)(sA + rowa + 32 * GEMM_ROW_BYTES + slot);
      frag_t b0 = *(const frag_t*)(sB + rowb + slot);
      frag_t b1 = *(const frag_t*)(sB + rowb + 32 * GEMM_ROW_BYTES + slot);
      MmaOps<T>::mma(a0, b0, acc[0][0]);
      MmaOps<T>::mma(a0, b1, acc[0][1]);
      MmaOps<T>::mma(a1, b0, acc[1][0]);
      MmaOps<T>::mma(a1, b1, acc[1][1]);
    }
    __syncthreads();  // next stage landed (vmcnt(0)) and this stage is free to overwrite
  }
}

// Work id -> (m tile, n tile): XCD-contiguous, grouped so that `group_m` row tiles sweep
// all column tiles together (operand panels stay in the XCD's L2).
__device__ inline void gemm_tile_coords(int64_t ntm, int64_t ntn, int group_m, int64_t& tm,
                                        int64_t& tn) {
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t per_group = (int64_t)group_m * ntn;
  const int64_t g = w / per_group;
  const int64_t first_m = g * group_m;
  const int64_t gsz = (ntm - first_m) < group_m ? (ntm - first_m) : group_m;
  const int64_t in_g = w % per_group;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}
