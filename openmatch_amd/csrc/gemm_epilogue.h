// Shared GEMM epilogue (bias, activation, residual, dropout, pre-activation copy) and the launcher
// template of the om_gemm_nt kernel generations; included by gemm.hip and gemm_wide*.hip.
//
// The epilogue is specialised at COMPILE time on the activation and on the "training extras"
// (pre-activation copy, dropout): a run-time `switch (act)` per output element costs ~200
// instructions per element once erff/tanhf are inlined (measured: the epilogue of a K = 768
// tile then takes longer than its whole main loop -- profiles/r01_gemm_trace_v2.log).
#pragma once
#include <stdlib.h>

#include "gemm_core.h"
#include "kernels.h"

// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): ~15 instructions instead of erff's ~60.
// Used where the result is rounded to 16 bits anyway; f32 outputs keep the libm erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-ax * ax);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

template <int ACT, bool FAST>
__device__ __forceinline__ float act_apply(float x) {
  if (ACT == OM_ACT_GELU_ERF)
    return FAST ? gelu_erf_fast(x) : 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  if (ACT == OM_ACT_RELU) return fmaxf(x, 0.0f);
  if (ACT == OM_ACT_GELU_TANH) {
    // HF NewGELUActivation: 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3)))
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
  }
  return x;
}

// d/dx of the erf GELU:  Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
// The same for 16-bit outputs, ~20 instructions instead of ~90 (libm erff + expf made the epilogue of the GELU-backward
// contraction longer than its K = 768 main loop: 138 us per launch, profiles/r02_train_kernel_stats_v2.csv):
// Phi(x) = 0.5 + xc Q(xc^2) with the degree-8 fit of gemm_epilogue6.h's forward (|error| <= 7.4e-6, xc = clamp(x, +-4.2)),
// phi by the hardware exp2 (1 ulp).  |error| < 1e-5 against an output rounded to 2^-9.
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f);
  const float t = xc * xc;
  float q = 5.998145036e-11f;
  q = fmaf(q, t, -5.633389311e-09f); q = fmaf(q, t, 2.343703613e-07f); q = fmaf(q, t, -5.760840850e-06f);
  q = fmaf(q, t, 9.457556007e-05f); q = fmaf(q, t, -1.114161685e-03f); q = fmaf(q, t, 9.830250405e-03f);
  q = fmaf(q, t, -6.636118144e-02f); q = fmaf(q, t, 3.989123106e-01f);
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);
  return fmaf(x, pdf, fmaf(xc, q, 0.5f));
}

// value of one output element before the residual: v = dropout(act(acc + bias))
template <int ACT, bool TRAIN, typename OutT>
__device__ __forceinline__ float epi_value(float v, int64_t m, int64_t n, int64_t M, int64_t N,
                                  const GemmEpilogue& ep, uint32_t drop_thresh, float drop_scale) {
  if (ACT == OM_ACT_GELU_ERF_GRAD) return v;          // multiplied by gelu'(resid) at store time
  if (TRAIN) {
    if (ep.pre_act && m < M && n < N) {
      float pv = v;          // OM_ACT_PRE_GRAD: the tape keeps gelu'(v) instead of v (every generation's epilogue honours the flag)
      if (ACT == OM_ACT_GELU_ERF && (ep.act & OM_ACT_PRE_GRAD)) pv = sizeof(OutT) == 2 ? gelu_erf_grad_fast(v) : gelu_erf_grad(v);
      ElemOps<OutT>::store((OutT*)ep.pre_act + m * ep.ldp + n, pv);
    }
  }
  v = act_apply<ACT, sizeof(OutT) == 2>(v);
  if (TRAIN) {
    if (drop_thresh) {
      const uint64_t mr = (ep.drop_rows && m < M) ? (uint64_t)(int64_t)ep.drop_rows[m] : (uint64_t)m;      // the row's token (packed rows)
      v = dropout_keep(ep.seed, mr * (uint64_t)N + (uint64_t)n, drop_thresh) ? v * drop_scale : 0.f;
    }
  }
  return v;
}

template <int ACT, bool FAST = false>      // FAST: the output is rounded to 16 bits
__device__ __forceinline__ float epi_resid(float v, float r, bool mul) {
  if (ACT == OM_ACT_GELU_ERF_GRAD) return v * (FAST ? gelu_erf_grad_fast(r) : gelu_erf_grad(r));
  return mul ? v * r : v + r;
}

// run-time (act, train) -> compile-time dispatch.  A plain macro on purpose: routing the
// accumulators through a lambda capture (or any reference) makes hipcc spill them to scratch.
#ifdef OM_EPI_PROBE_ONE   /* developer builds: a single epilogue variant (compile time, ISA reading) */
#define OM_EPI_SWITCH(ACTV, TRAINV, CALL) { CALL(OM_EPI_PROBE_ONE, false); }
#else
#define OM_EPI_SWITCH(ACTV, TRAINV, CALL)                              \
  switch (ACTV) {                                                      \
    case OM_ACT_GELU_ERF:      if (TRAINV) { CALL(OM_ACT_GELU_ERF, true); } else { CALL(OM_ACT_GELU_ERF, false); } break;           \
    case OM_ACT_RELU:          if (TRAINV) { CALL(OM_ACT_RELU, true); } else { CALL(OM_ACT_RELU, false); } break;                   \
    case OM_ACT_GELU_TANH:     if (TRAINV) { CALL(OM_ACT_GELU_TANH, true); } else { CALL(OM_ACT_GELU_TANH, false); } break;         \
    case OM_ACT_GELU_ERF_GRAD: if (TRAINV) { CALL(OM_ACT_GELU_ERF_GRAD, true); } else { CALL(OM_ACT_GELU_ERF_GRAD, false); } break; \
    default:                   if (TRAINV) { CALL(OM_ACT_NONE, true); } else { CALL(OM_ACT_NONE, false); }                          \
  }
#endif

typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32: (lo, hi) -> packed dword, round-to-nearest-even
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

// 16-byte output vectors
template <typename OutT> struct OutVec;
template <> struct OutVec<float> {
  static constexpr int VEC = 4;
  __device__ static inline void unpack(const uint4& u, float (&v)[4]) {
    v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
  }
  __device__ static inline uint4 pack(const float (&v)[4]) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
  }
};
template <> struct OutVec<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static inline void unpack(const uint4& u, float (&v)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffff)); v[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
  }
  __device__ static inline uint4 pack(const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <> struct OutVec<f16_t> {
  static constexpr int VEC = 8;
  __device__ static inline void unpack(const uint4& u, float (&v)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = Half16<f16_t>::lo(w[i]); v[2 * i + 1] = Half16<f16_t>::hi(w[i]); }
  }
  __device__ static inline uint4 pack(const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = Half16<f16_t>::pack2(v[2 * i], v[2 * i + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

struct EpiScalars {
  uint32_t drop_thresh; float drop_scale; bool train, mul; int act;
  __device__ __forceinline__ explicit EpiScalars(const GemmEpilogue& ep) {
    const DropCfg dc(ep.drop_p);
    drop_thresh = dc.thresh; drop_scale = dc.keep_scale;
    train = ep.pre_act != nullptr || ep.drop_p > 0.f;
    mul = (ep.act & OM_ACT_MUL_RESID) != 0;
    act = ep.act & 0xff;
  }
};

#define PATCH_STRIDE (64 * 4 + 16)   // f32 staging row of 64 columns, +16 B against bank conflicts

// Stage one [32 x 64] f32 patch (one `mi` row block of a wave's sub-tile: tiles acc0 | acc1) into
// the wave's LDS region, then stream it out as whole 16-byte row segments with the residual
// applied in f32 (a single rounding).  Used by the 512-thread kernels.
template <typename OutT, int ACT, bool TRAIN>
__device__ __forceinline__ void store_patch(const f32x16_t acc0, const f32x16_t acc1, float bias0, float bias1,
                                   int64_t mrow0, int64_t ncol0, OutT* C, int64_t ldc, int64_t M,
                                   int64_t N, const GemmEpilogue& ep, const EpiScalars& es,
                                   char* region) {
  constexpr int VEC = OutVec<OutT>::VEC;
  constexpr int CPR = 64 / VEC;
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const OutT* resid = (const OutT*)ep.resid;   // may alias C (in-place +=)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    float* dst = (float*)(region + row * PATCH_STRIDE);
    dst[l31] = epi_value<ACT, TRAIN, OutT>(acc0[r] + bias0, mrow0 + row, ncol0 + l31, M, N, ep,
                                           es.drop_thresh, es.drop_scale);
    dst[32 + l31] = epi_value<ACT, TRAIN, OutT>(acc1[r] + bias1, mrow0 + row, ncol0 + 32 + l31, M, N,
                                                ep, es.drop_thresh, es.drop_scale);
  }
  // the region is private to this wave and a wave's LDS operations execute in order: no block
  // barrier, just keep the compiler from hoisting the reads above the writes
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < CPR / 2; ++it) {
    const int id = it * 64 + lane;
    const int row = id / CPR, c = id % CPR;
    const int64_t m = mrow0 + row, n = ncol0 + c * VEC;
    if (m < M && n < N) {
      float xv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const f32x4_t t4 = *(const f32x4_t*)(region + row * PATCH_STRIDE + (c * VEC + e) * 4);
        xv[e] = t4[0]; xv[e + 1] = t4[1]; xv[e + 2] = t4[2]; xv[e + 3] = t4[3];
      }
      if (resid) {
        float rv[VEC];
        OutVec<OutT>::unpack(*(const uint4*)(resid + m * ep.ldr + n), rv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) xv[e] = epi_resid<ACT, sizeof(OutT) == 2>(xv[e], rv[e], es.mul);
      }
      *(uint4*)(C + m * ldc + n) = OutVec<OutT>::pack(xv);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next patch overwrites
}

// v1 epilogue: the wave's 64 x 64 sub-tile goes straight from the accumulators to memory.
template <typename OutT, int ACT, bool TRAIN>
__device__ __forceinline__ void store_direct(const f32x16_t a00, const f32x16_t a01, const f32x16_t a10,
                                             const f32x16_t a11, int64_t mrow0, int64_t ncol0, OutT* C,
                                             int64_t ldc, int64_t M, int64_t N, const GemmEpilogue& ep,
                                             const EpiScalars& es) {
  const int lane = threadIdx.x & 63;
  const OutT* resid = (const OutT*)ep.resid;   // may alias C (in-place +=)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = ncol0 + ni * 32 + (lane & 31);
    if (n >= N) continue;
    const float bv = ep.bias ? ep.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int64_t mbase = mrow0 + mi * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
        if (m >= M) continue;
        const float a = mi == 0 ? (ni == 0 ? a00[r] : a01[r]) : (ni == 0 ? a10[r] : a11[r]);
        float v = epi_value<ACT, TRAIN, OutT>(a + bv, m, n, M, N, ep, es.drop_thresh, es.drop_scale);
        if (resid) v = epi_resid<ACT, sizeof(OutT) == 2>(v, ElemOps<OutT>::load(resid + m * ep.ldr + n), es.mul);
        ElemOps<OutT>::store(C + m * ldc + n, v);
      }
    }
  }
}

// ---- launchers -------------------------------------------------------------------------------------
#define OM_DEFINE_LAUNCHER(NAME, KERNEL, THREADS, LDS, BMV, BNV)                                        \
  template <typename T, typename OutT>                                                                  \
  static int NAME(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,         \
                  int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {             \
    const int64_t nwg = ((M + BMV - 1) / BMV) * ((N + BNV - 1) / BNV);                                  \
    if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");                                                  \
    static std::atomic<bool> attr_set{false};                                                                       \
    if (!attr_set) {                                                                                    \
      OM_HIP(hipFuncSetAttribute((const void*)KERNEL<T, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
      attr_set = true;                                                                                  \
    }                                                                                                   \
    const int tclass = sizeof(T) == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32; /* f16 counts as 16-bit */ \
    const bool timing = om_timing_on();                                                                 \
    if (timing) om_timing_begin(tclass, s);                                                             \
    /* sweep order: 8 row tiles stay resident while the column tiles are walked (L2 reuse per XCD) */    \
    hipLaunchKernelGGL((KERNEL<T, OutT>), dim3((unsigned)nwg), dim3(THREADS), LDS, s, (const T*)A, lda,  \
                       (const T*)B, ldb, (OutT*)C, ldc, M, N, K, ep, 8);                                \
    if (timing) om_timing_end(tclass, s, 2.0 * (double)M * (double)N * (double)K);                      \
    OM_LAUNCH_CHECK();                                                                                  \
    return 0;                                                                                           \
  }
