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
template <> struct AttnGeom<f16_t> {              // IEEE half: the same geometry as bfloat16
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
template <> struct SlabMma<f16_t> {
  __device__ static inline void run(const f32x16_t& a, const f16_t* bt, int LP, f32x16_t (&o)[2]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    f16x8_t pa[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[u][j] = (f16_t)a[8 * u + j];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16_t* p = bt + dt * 32 * LP + 16 * u;
        const f16x4_t v0 = *(const f16x4_t*)p;
        const f16x4_t v1 = *(const f16x4_t*)(p + 8);
        f16x8_t vb;
        vb[0] = v0[0]; vb[1] = v0[1]; vb[2] = v0[2]; vb[3] = v0[3];
        vb[4] = v1[0]; vb[5] = v1[1]; vb[6] = v1[2]; vb[7] = v1[3];
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[u], vb, o[dt], 0, 0, 0);
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

// Dropout of the attention probabilities.  One 64-bit hash serves the FOUR probabilities (b, h, q, 4 kg .. 4 kg + 3): bits
// 16 e .. 16 e + 15 decide element e (kept iff >= p * 2^16).  Four consecutive keys are exactly what one lane holds per
// accumulator group in the lane-per-query orientation of the forward and of the backward's first phase, so those
// kernels pay ~10 instead of ~35 integer instructions per probability (the three 64-bit multiplies of the hash were the
// largest single cost of the attention backward).  p is resolved to 2^-16; keep_scale uses the same rounded value.
using AttnDrop = DropCfg;   // kernels.h: 16-bit fields of one 64-bit hash per four elements
__device__ inline uint64_t attn_drop_bits(uint64_t seed, int64_t b, int h, int heads, int L, int q, int kg) {
  return om_hash64(seed, (((uint64_t)b * heads + h) * (uint64_t)L + q) * (uint64_t)((L + 3) >> 2) + kg);
}
__device__ inline bool attn_drop_keep(uint64_t bits, int e, uint32_t thresh) { return dropout_field(bits, e, thresh); }
// one probability (the lane-per-key orientation): the group's hash, this element's field
__device__ inline bool attn_drop_keep1(uint64_t seed, int64_t b, int h, int heads, int L, int q, int key, uint32_t thresh) {
  return attn_drop_keep(attn_drop_bits(seed, b, h, heads, L, q, key >> 2), key & 3, thresh);
}
