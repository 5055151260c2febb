// One wave, one row: the arithmetic of LayerNorm / RMSNorm shared by the normalisation kernels (elementwise.hip) and by the
// contraction that applies a PENDING LayerNorm to its own operand and residual (gemm_skinny.hip, round 6) -- one definition, so that
// a row normalised by either gives the same bits.  torch.nn.LayerNorm: two-pass mean / biased variance in f32;
// T5LayerNorm (HF:models/t5/modeling_t5.py:59-72): x * rsqrt(mean(x^2) + eps) * g.
#pragma once
#include "common.h"

// four consecutive elements of a row <-> four floats (T = float, bf16_t, f16_t); the stores round to nearest even
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  __device__ static inline void load(const float* p, float (&v)[4]) {
    const float4 t = *(const float4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static inline void store(float* p, const float (&v)[4]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec4<bf16_t> {
  __device__ static inline void load(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *(const uint2*)p;
    v[0] = bf16_to_f32((bf16_t)(t.x & 0xffff)); v[1] = bf16_to_f32((bf16_t)(t.x >> 16));
    v[2] = bf16_to_f32((bf16_t)(t.y & 0xffff)); v[3] = bf16_to_f32((bf16_t)(t.y >> 16));
  }
  __device__ static inline void store(bf16_t* p, const float (&v)[4]) {
    // v_cvt_pk_bf16_f32 (gfx950): round to nearest even like f32_to_bf16, which it equals on every non-NaN input (the two give different NaNs)
    *(uint2*)p = make_uint2(Half16<bf16_t>::pack2(v[0], v[1]), Half16<bf16_t>::pack2(v[2], v[3]));
  }
};

template <> struct Vec4<f16_t> {
  __device__ static inline void load(const f16_t* p, float (&v)[4]) {
    const uint2 t = *(const uint2*)p;
    v[0] = Half16<f16_t>::lo(t.x); v[1] = Half16<f16_t>::hi(t.x);
    v[2] = Half16<f16_t>::lo(t.y); v[3] = Half16<f16_t>::hi(t.y);
  }
  __device__ static inline void store(f16_t* p, const float (&v)[4]) {
    *(uint2*)p = make_uint2(Half16<f16_t>::pack2(v[0], v[1]), Half16<f16_t>::pack2(v[2], v[3]));
  }
};


// v + (lane ^ O's v), O = 32 ... 1: the pairing of the __shfl_xor butterfly (common.h wave_sum; float addition commutes, so the bits are
// the same) without its LDS-crossbar trip where the hardware has a cheaper route: v_permlane32_swap / v_permlane16_swap (gfx950), DPP row
// rotation by 8, DPP quad permutations for 2 and 1; 4 is a ds_swizzle (xor mask, no address register)
template <int O>
__device__ __forceinline__ float ln_xor_add(float v) {
  const int vi = (int)__float_as_uint(v);
  if constexpr (O == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if constexpr (O == 8) {
    return v + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, vi, 0x128, 0xf, 0xf, false));      // row_ror:8 == lane ^ 8 within a row of 16
  } else if constexpr (O == 4) {
    return v + __uint_as_float((uint32_t)__builtin_amdgcn_ds_swizzle(vi, 0x101f));                         // bit mode: and 0x1f, or 0, xor 4
  } else if constexpr (O == 2) {
    return v + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, vi, 0x4e, 0xf, 0xf, false));       // quad_perm [2, 3, 0, 1]
  } else {
    return v + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, vi, 0xb1, 0xf, 0xf, false));       // quad_perm [1, 0, 3, 2]
  }
}
template <int R>
__device__ __forceinline__ void ln_wave_sums(float (&s)[R]) {
#pragma unroll
  for (int q = 0; q < R; ++q) s[q] = ln_xor_add<32>(s[q]);
#pragma unroll
  for (int q = 0; q < R; ++q) s[q] = ln_xor_add<16>(s[q]);
#pragma unroll
  for (int q = 0; q < R; ++q) s[q] = ln_xor_add<8>(s[q]);
#pragma unroll
  for (int q = 0; q < R; ++q) s[q] = ln_xor_add<4>(s[q]);
#pragma unroll
  for (int q = 0; q < R; ++q) s[q] = ln_xor_add<2>(s[q]);
#pragma unroll
  for (int q = 0; q < R; ++q) s[q] = ln_xor_add<1>(s[q]);
}

// R rows at once, one wave: lane l holds elements (l + 64 j) * 4 .. + 3 of every row, j < nvec.  Per row the arithmetic is fixed -- lane
// partials in (j, e) order, a 32-16-8-4-2-1 butterfly, mean = sum / H, then the same for the squared deviations -- and the rows' butterflies
// are interleaved step by step: one row's is a chain of twelve dependent cross-lane steps, R of them in flight cost little more than
// one (the few-rows contraction normalises 4-8 rows per wave under its weight loads).
template <int R, int MAX_VEC>
__device__ inline void ln_rows_stats(const float (&x)[R][MAX_VEC][4], int nvec, int lane, int H, float eps, int rms, float (&mean)[R], float (&rstd)[R]) {
  float s[R];
#pragma unroll
  for (int q = 0; q < R; ++q) { s[q] = 0.f; mean[q] = 0.f; }
  if (!rms) {
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int j = 0; j < MAX_VEC; ++j)
        if (j < nvec && (lane + 64 * j) * 4 < H) s[q] += (x[q][j][0] + x[q][j][1]) + (x[q][j][2] + x[q][j][3]);
    ln_wave_sums<R>(s);
#pragma unroll
    for (int q = 0; q < R; ++q) mean[q] = s[q] / (float)H;
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_VEC; ++j)
      if (j < nvec && (lane + 64 * j) * 4 < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[q][j][e] - mean[q]; ss += d * d; }
      }
    s[q] = ss;
  }
  ln_wave_sums<R>(s);
#pragma unroll
  for (int q = 0; q < R; ++q) rstd[q] = rsqrtf(s[q] / (float)H + eps);
}

// one row
template <int MAX_VEC>
__device__ inline void ln_row_stats(const float (&x)[MAX_VEC][4], int nvec, int lane, int H, float eps, int rms, float& mean, float& rstd) {
  float m1[1], r1[1];
  ln_rows_stats<1, MAX_VEC>(*(const float (*)[1][MAX_VEC][4])&x, nvec, lane, H, eps, rms, m1, r1);
  mean = m1[0]; rstd = r1[0];
}

// one element of the normalised row (with a shift): the product and the sum as ONE fused multiply-add, written out so that every
// translation unit rounds alike
__device__ __forceinline__ float ln_affine(float x, float mean, float rstd, float g, float b) {
  return __builtin_fmaf((x - mean) * rstd, g, b);
}
