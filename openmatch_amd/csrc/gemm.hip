// om_gemm_nt: C = act(A · B^T + bias) (+|*) resid  on MFMA (bf16 / f16 or exact-f32), gfx950.
// Stands in for the ATen/BLAS GEMM under every nn.Linear on the hot path
// (HF:models/bert/modeling_bert.py:175-177,289-293,334-351; linear.py:22-23).
//
// Three kernel generations share the epilogue below:
//   v1  128x128 tile, 2-stage LDS, direct stores          (small / ragged problems)
//   v2  256x128 tile, 3-deep LDS ring (gemm_core2.h)       (N < 256)
//   v3  256x256 tile, 2 x 64 KiB LDS    (gemm_core3.h)       (everything large)
// The epilogue is specialised at COMPILE time on the activation and on the "training extras"
// (pre-activation copy, dropout): a run-time `switch (act)` per output element costs ~200
// instructions per element once erff/tanhf are inlined (measured: the epilogue of a K = 768
// tile then takes longer than its whole main loop -- profiles/r01_gemm_trace_v2.log).
#include <stdlib.h>

#include "gemm_core2.h"
#include "gemm_core3.h"
#include "gemm_core4.h"
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

// d/dx of the erf GELU
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// value of one output element before the residual: v = dropout(act(acc + bias))
template <int ACT, bool TRAIN, typename OutT>
__device__ __forceinline__ float epi_value(float v, int64_t m, int64_t n, int64_t M, int64_t N,
                                  const GemmEpilogue& ep, uint32_t drop_thresh, float drop_scale) {
  if (ACT == OM_ACT_GELU_ERF_GRAD) return v;          // multiplied by gelu'(resid) at store time
  if (TRAIN) {
    if (ep.pre_act && m < M && n < N) ElemOps<OutT>::store((OutT*)ep.pre_act + m * ep.ldp + n, v);
  }
  v = act_apply<ACT, sizeof(OutT) == 2>(v);
  if (TRAIN) {
    if (drop_thresh)
      v = dropout_keep(ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)n, drop_thresh) ? v * drop_scale : 0.f;
  }
  return v;
}

template <int ACT>
__device__ __forceinline__ float epi_resid(float v, float r, bool mul) {
  if (ACT == OM_ACT_GELU_ERF_GRAD) return v * gelu_erf_grad(r);
  return mul ? v * r : v + r;
}

// run-time (act, train) -> compile-time dispatch.  A plain macro on purpose: routing the
// accumulators through a lambda capture (or any reference) makes hipcc spill them to scratch.
#define OM_EPI_SWITCH(ACTV, TRAINV, CALL)                              \
  switch (ACTV) {                                                      \
    case OM_ACT_GELU_ERF:      if (TRAINV) { CALL(OM_ACT_GELU_ERF, true); } else { CALL(OM_ACT_GELU_ERF, false); } break;           \
    case OM_ACT_RELU:          if (TRAINV) { CALL(OM_ACT_RELU, true); } else { CALL(OM_ACT_RELU, false); } break;                   \
    case OM_ACT_GELU_TANH:     if (TRAINV) { CALL(OM_ACT_GELU_TANH, true); } else { CALL(OM_ACT_GELU_TANH, false); } break;         \
    case OM_ACT_GELU_ERF_GRAD: if (TRAINV) { CALL(OM_ACT_GELU_ERF_GRAD, true); } else { CALL(OM_ACT_GELU_ERF_GRAD, false); } break; \
    default:                   if (TRAINV) { CALL(OM_ACT_NONE, true); } else { CALL(OM_ACT_NONE, false); }                          \
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
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

struct EpiScalars {
  uint32_t drop_thresh; float drop_scale; bool train, mul; int act;
  __device__ __forceinline__ explicit EpiScalars(const GemmEpilogue& ep) {
    drop_thresh = ep.drop_p > 0.f ? (uint32_t)(ep.drop_p * 4294967296.0) : 0u;
    drop_scale = ep.drop_p > 0.f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
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
        for (int e = 0; e < VEC; ++e) xv[e] = epi_resid<ACT>(xv[e], rv[e], es.mul);
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
        if (resid) v = epi_resid<ACT>(v, ElemOps<OutT>::load(resid + m * ep.ldr + n), es.mul);
        ElemOps<OutT>::store(C + m * ldc + n, v);
      }
    }
  }
}

// ---- v1: 128x128 tile, direct (scattered) stores; any M, N --------------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t ntm = (M + GEMM_BM - 1) / GEMM_BM, ntn = (N + GEMM_BN - 1) / GEMM_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  const int64_t m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
  f32x16_t acc[2][2];
  gemm_mainloop<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const EpiScalars es(ep);
#define OM_V1_CALL(A, TR) store_direct<OutT, A, TR>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], m0 + wm * 64, n0 + wn * 64, C, ldc, M, N, ep, es)
  OM_EPI_SWITCH(es.act, es.train, OM_V1_CALL)
#undef OM_V1_CALL
}

// ---- split-K: C (f32, zero-initialised by the caller) += A[:, ks] · B[:, ks]^T per K slice --------
// Weight-gradient contractions have a long K (the token count) and a small output (768 x 768 is
// 9 tiles of 256^2): slicing K over blockIdx.y fills the chip; partial sums meet in f32 atomics.
template <typename T>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_splitk_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, int steps_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t ntn = (N + GEMM_BN - 1) / GEMM_BN;
  const int64_t m0 = (blockIdx.x / ntn) * GEMM_BM, n0 = (blockIdx.x % ntn) * GEMM_BN;
  f32x16_t acc[2][2];
  gemm_mainloop<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc,
                   (int64_t)blockIdx.y * steps_per_slice * GEMM_ROW_BYTES, steps_per_slice);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + (lane & 31);
    if (n >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int64_t mbase = m0 + wm * 64 + mi * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
        if (m < M) atomicAdd(C + m * ldc + n, acc[mi][ni][r]);
      }
    }
  }
}

// ---- v2: 256x128 tile, 3-deep LDS ring -----------------------------------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(G2_THREADS) void gemm_nt_kernel2(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g2_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[2][2];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = clock64();
  gemm_mainloop2<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, tr);
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const EpiScalars es(ep);
  const int64_t nc = n0 + wn * 64;
  const float b0 = (ep.bias && nc + (lane & 31) < N) ? ep.bias[nc + (lane & 31)] : 0.f;
  const float b1 = (ep.bias && nc + 32 + (lane & 31) < N) ? ep.bias[nc + 32 + (lane & 31)] : 0.f;
  char* region = smem + wave * (32 * PATCH_STRIDE);
  __syncthreads();                                  // every wave is done reading the ring
#define OM_V2_CALL(A, TR)                                                                                   \
  store_patch<OutT, A, TR>(acc[0][0], acc[0][1], b0, b1, m0 + wm * 64, nc, C, ldc, M, N, ep, es, region);    \
  store_patch<OutT, A, TR>(acc[1][0], acc[1][1], b0, b1, m0 + wm * 64 + 32, nc, C, ldc, M, N, ep, es, region)
  OM_EPI_SWITCH(es.act, es.train, OM_V2_CALL)
#undef OM_V2_CALL
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; }
}

// ---- v3: 256x256 tile ------------------------------------------------------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(G3_THREADS) void gemm_nt_kernel3(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g3_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[4][2];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = clock64();
  gemm_mainloop3<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, tr);   // ends on a barrier
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const EpiScalars es(ep);
  const int64_t nc = n0 + wn * 64;
  const float b0 = (ep.bias && nc + (lane & 31) < N) ? ep.bias[nc + (lane & 31)] : 0.f;
  const float b1 = (ep.bias && nc + 32 + (lane & 31) < N) ? ep.bias[nc + 32 + (lane & 31)] : 0.f;
  char* region = smem + wave * (32 * PATCH_STRIDE);
#define OM_V3_CALL(A, TR)                                                                                        \
  store_patch<OutT, A, TR>(acc[0][0], acc[0][1], b0, b1, m0 + wm * 128, nc, C, ldc, M, N, ep, es, region);        \
  store_patch<OutT, A, TR>(acc[1][0], acc[1][1], b0, b1, m0 + wm * 128 + 32, nc, C, ldc, M, N, ep, es, region);   \
  store_patch<OutT, A, TR>(acc[2][0], acc[2][1], b0, b1, m0 + wm * 128 + 64, nc, C, ldc, M, N, ep, es, region);   \
  store_patch<OutT, A, TR>(acc[3][0], acc[3][1], b0, b1, m0 + wm * 128 + 96, nc, C, ldc, M, N, ep, es, region)
  OM_EPI_SWITCH(es.act, es.train, OM_V3_CALL)
#undef OM_V3_CALL
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; }
}

// ---- v4: 256x256 tile, 4-deep ring of 64-byte K steps (gemm_core4.h) -------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(G4_THREADS) void gemm_nt_kernel4(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g4_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[4][2];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = clock64();
  gemm_mainloop4<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, tr);   // ends on a barrier
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const EpiScalars es(ep);
  const int64_t nc = n0 + wn * 64;
  const float b0 = (ep.bias && nc + (lane & 31) < N) ? ep.bias[nc + (lane & 31)] : 0.f;
  const float b1 = (ep.bias && nc + 32 + (lane & 31) < N) ? ep.bias[nc + 32 + (lane & 31)] : 0.f;
  char* region = smem + wave * (32 * PATCH_STRIDE);
#define OM_V4_CALL(A, TR)                                                                                        \
  store_patch<OutT, A, TR>(acc[0][0], acc[0][1], b0, b1, m0 + wm * 128, nc, C, ldc, M, N, ep, es, region);        \
  store_patch<OutT, A, TR>(acc[1][0], acc[1][1], b0, b1, m0 + wm * 128 + 32, nc, C, ldc, M, N, ep, es, region);   \
  store_patch<OutT, A, TR>(acc[2][0], acc[2][1], b0, b1, m0 + wm * 128 + 64, nc, C, ldc, M, N, ep, es, region);   \
  store_patch<OutT, A, TR>(acc[3][0], acc[3][1], b0, b1, m0 + wm * 128 + 96, nc, C, ldc, M, N, ep, es, region)
  OM_EPI_SWITCH(es.act, es.train, OM_V4_CALL)
#undef OM_V4_CALL
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; }
}

// ---- launchers -------------------------------------------------------------------------------------
#define OM_DEFINE_LAUNCHER(NAME, KERNEL, THREADS, LDS, BMV, BNV)                                        \
  template <typename T, typename OutT>                                                                  \
  static int NAME(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,         \
                  int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {             \
    const int64_t nwg = ((M + BMV - 1) / BMV) * ((N + BNV - 1) / BNV);                                  \
    if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");                                                  \
    static bool attr_set = false;                                                                       \
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
OM_DEFINE_LAUNCHER(launch_gemm, gemm_nt_kernel, GEMM_THREADS, GEMM_LDS_BYTES, GEMM_BM, GEMM_BN)
OM_DEFINE_LAUNCHER(launch_gemm2, gemm_nt_kernel2, G2_THREADS, G2_LDS_BYTES, G2_BM, G2_BN)
OM_DEFINE_LAUNCHER(launch_gemm3, gemm_nt_kernel3, G3_THREADS, G3_LDS_BYTES, G3_BM, G3_BN)
OM_DEFINE_LAUNCHER(launch_gemm4, gemm_nt_kernel4, G4_THREADS, G4_LDS_BYTES, G4_BM, G4_BN)

static int gemm_variant() {     // OM_GEMM_VARIANT=1|2|3 pins a kernel generation (A/B measurements)
  static const int v = getenv("OM_GEMM_VARIANT") ? atoi(getenv("OM_GEMM_VARIANT")) : 0;
  return v;
}

// The 256-row kernels write whole 16-byte output segments; small or ragged problems use v1.
static bool wide_ok(int out_dtype, const void* C, int64_t ldc, int64_t M, int64_t N, const GemmEpilogue& ep) {
  if (gemm_variant() == 1) return false;
  const int64_t vec = out_dtype == OM_F32 ? 4 : 8;
  if (M < 512 || N % vec || ldc % vec || ((uintptr_t)C & 15)) return false;
  if (ep.resid && (ep.ldr % vec || ((uintptr_t)ep.resid & 15))) return false;
  return true;
}

static unsigned long long* g_trace = nullptr;
extern "C" void om_debug_gemm_trace(unsigned long long* buf) { g_trace = buf; }

int omk_gemm(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep_in,
             hipStream_t s) {
  GemmEpilogue ep = ep_in;
  ep.trace = g_trace;
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0) OM_FAIL("K must be positive");
  const int64_t es = in_dtype == OM_F32 ? 4 : 2;
  if ((K * es) % GEMM_ROW_BYTES != 0) OM_FAIL("K*sizeof(elem) must be a multiple of 128 bytes");
  if ((lda * es) % 16 != 0 || (ldb * es) % 16 != 0) OM_FAIL("lda/ldb must keep rows 16-byte aligned");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) OM_FAIL("A/B must be 16-byte aligned");
  if ((ep.act & 0xff) == OM_ACT_GELU_ERF_GRAD && !ep.resid) OM_FAIL("gelu-grad epilogue needs resid");
  const bool wide = wide_ok(out_dtype, C, ldc, M, N, ep);
  // Pick the tile generation that finishes first: whole rounds of (256 CUs x resident workgroups)
  // times the tile's work over its measured relative efficiency (profiles/r01_selftest_gemm_v4.log).
  int gen = 1;
  if (wide) {
    // cost = rounds x (work a CU has in flight per round) / efficiency; v1 keeps 2 workgroups per CU
    auto rounds = [&](int64_t bm, int64_t bn, int64_t slots) {
      const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
      return (double)((tiles + slots - 1) / slots);
    };
    const double c4 = N >= 256 ? rounds(256, 256, 256) * (256.0 * 256.0) / 1.00 : 1e30;
    const double c2 = rounds(256, 128, 256) * (256.0 * 128.0) / 0.92;
    const double c1 = rounds(128, 128, 512) * (2 * 128.0 * 128.0) / 0.70;
    gen = c4 <= c2 && c4 <= c1 ? 4 : (c2 <= c1 ? 2 : 1);
    if (gemm_variant() == 2) gen = 2;
    if (gemm_variant() == 3 || gemm_variant() == 4) gen = N >= 256 ? gemm_variant() : 2;
  }
  if (gen == 4 && (K * es) % 128 != 0) gen = 3;
#define OM_GEMM_GO(TI, TO)                                                                            \
  do {                                                                                                \
    if (gen == 4) return launch_gemm4<TI, TO>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                 \
    if (gen == 3) return launch_gemm3<TI, TO>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                 \
    if (gen == 2) return launch_gemm2<TI, TO>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                 \
    return launch_gemm<TI, TO>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                                \
  } while (0)
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16) OM_GEMM_GO(bf16_t, bf16_t);
  if (in_dtype == OM_BF16 && out_dtype == OM_F32) OM_GEMM_GO(bf16_t, float);
  if (in_dtype == OM_F32 && out_dtype == OM_F32) OM_GEMM_GO(float, float);
  if (in_dtype == OM_F16 && out_dtype == OM_F32) OM_GEMM_GO(f16_t, float);
  if (in_dtype == OM_F32 && out_dtype == OM_BF16) return launch_gemm<float, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
#undef OM_GEMM_GO
  OM_FAIL("unsupported dtype combination");
}

int omk_gemm_splitk(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int64_t es = in_dtype == OM_F32 ? 4 : 2;
  if ((K * es) % GEMM_ROW_BYTES != 0) OM_FAIL("K*sizeof(elem) must be a multiple of 128 bytes");
  const int64_t tiles = ((M + GEMM_BM - 1) / GEMM_BM) * ((N + GEMM_BN - 1) / GEMM_BN);
  const int nk = (int)(K * es / GEMM_ROW_BYTES);
  int slices = (int)((1024 + tiles - 1) / tiles);          // aim at ~4 workgroups per CU
  if (slices > nk / 4) slices = nk / 4 > 0 ? nk / 4 : 1;   // but at least 4 K steps per slice
  const int per = (nk + slices - 1) / slices;
  slices = (nk + per - 1) / per;
  static bool attr_bf16 = false, attr_f32 = false;
  const bool timing = om_timing_on();
  const int tclass = es == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32;
  if (timing) om_timing_begin(tclass, s);
  if (in_dtype == OM_BF16) {
    if (!attr_bf16) { OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_splitk_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)); attr_bf16 = true; }
    hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16_t>), dim3((unsigned)tiles, slices), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s,
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, per);
  } else if (in_dtype == OM_F32) {
    if (!attr_f32) { OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_splitk_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)); attr_f32 = true; }
    hipLaunchKernelGGL((gemm_nt_splitk_kernel<float>), dim3((unsigned)tiles, slices), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s,
                       (const float*)A, lda, (const float*)B, ldb, C, ldc, M, N, K, per);
  } else {
    OM_FAIL("unsupported dtype");
  }
  if (timing) om_timing_end(tclass, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

extern "C" int om_gemm_nt(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                          int out_dtype, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const float* bias, const void* resid, int64_t ldr, int act,
                          void* stream) {
  GemmEpilogue ep = {};
  ep.bias = bias; ep.resid = resid; ep.ldr = ldr; ep.act = act;
  return omk_gemm(in_dtype, A, lda, B, ldb, out_dtype, C, ldc, M, N, K, ep, (hipStream_t)stream);
}
