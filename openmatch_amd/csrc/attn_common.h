// Shared pieces of the attention forward / backward kernels (gfx950).
#pragma once
#include "gemm_core.h"
#include "kernels.h"

template <typename T> struct AttnGeom;
template <> struct AttnGeom<bf16_t> {
  static constexpr int ROWB = 128, CPR = 8, EPC = 8, NKK = 4;
  __device__ static inline int key(int row) { return (row >> 1) & 7; }
  __device__ static inline float exp_(float x) { return __expf(x); }
};
template <> struct AttnGeom<float> {
  static constexpr int ROWB = 256, CPR = 16, EPC = 4, NKK = 8;
  __device__ static inline int key(int row) { return row & 15; }
  __device__ static inline float exp_(float x) { return expf(x); }
};

// One 32-wide slab of a contraction whose left operand sits in a 32x32 accumulator layout:
//   o[dt][i][j] += sum_c a[i][c] * Bt[dt*32 + j][c]      (i = lane&31 of the A operand)
// `a` holds, for lane (i, half), the 16 values c = (r&3) + 8(r>>2) + 4*half, r in [0,16);
// `bt` points at Bt[(lane&31)][slab*32 + 4*half] of a TRANSPOSED LDS image with row pitch LP,
// so each lane reads its k-slots as contiguous 8/16-byte pieces.
template <typename T> struct SlabMma;
template <> struct SlabMma<bf16_t> {
  // k-slot j of half h  <->  c = 16u + 8(j>>2) + (j&3) + 4h
  __device__ static inline void run(const f32x16_t& a, const bf16_t* bt, int LP, f32x16_t (&o)[2]) {
    bf16x8_t pa[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[u][j] = (short)f32_to_bf16(a[8 * u + j]);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16_t* p = bt + dt * 32 * LP + 16 * u;
        const bf16x4_t v0 = *(const bf16x4_t*)p;
        const bf16x4_t v1 = *(const bf16x4_t*)(p + 8);
        bf16x8_t vb;
        vb[0] = v0[0]; vb[1] = v0[1]; vb[2] = v0[2]; vb[3] = v0[3];
        vb[4] = v1[0]; vb[5] = v1[1]; vb[6] = v1[2]; vb[7] = v1[3];
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[u], vb, o[dt], 0, 0, 0);
      }
  }
};
template <> struct SlabMma<float> {
  // register r of half h  <->  c = (r&3) + 8(r>>2) + 4h  (one register per k = 2 MFMA)
  __device__ static inline void run(const f32x16_t& a, const float* bt, int LP, f32x16_t (&o)[2]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t vb = *(const f32x4_t*)(bt + dt * 32 * LP + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + e], vb[e], o[dt], 0, 0, 0);
      }
  }
};

// dropout element index of attention probability (b, h, q, key)
__device__ inline uint64_t attn_drop_idx(int64_t b, int h, int heads, int L, int q, int key) {
  return (((uint64_t)b * heads + h) * (uint64_t)L + q) * (uint64_t)L + key;
}
