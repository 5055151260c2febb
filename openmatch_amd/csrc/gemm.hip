// om_gemm_nt: C = act(A · B^T + bias) + resid  on MFMA (bf16 or exact-f32), gfx950.
// Stands in for the ATen/BLAS GEMM under every nn.Linear on the hot path
// (HF:models/bert/modeling_bert.py:175-177,289-293,334-351; linear.py:22-23).
#include <stdlib.h>

#include "gemm_core2.h"
#include "kernels.h"

__device__ inline float act_apply(float x, int act) {
  if (act == OM_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  if (act == OM_ACT_RELU) return fmaxf(x, 0.0f);
  if (act == OM_ACT_GELU_TANH) {
    // HF NewGELUActivation: 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3)))
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
  }
  return x;
}

// d/dx of the erf GELU
__device__ inline float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

template <typename T, typename OutT>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  // ep.resid may alias C (in-place +=)
  const float* __restrict__ bias = ep.bias;
  const OutT* resid = (const OutT*)ep.resid;
  const int64_t ldr = ep.ldr;
  const int act = ep.act;
  OutT* pre_act = (OutT*)ep.pre_act;
  const uint32_t drop_thresh = ep.drop_p > 0.f ? (uint32_t)(ep.drop_p * 4294967296.0) : 0u;
  const float drop_scale = ep.drop_p > 0.f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t ntm = (M + GEMM_BM - 1) / GEMM_BM, ntn = (N + GEMM_BN - 1) / GEMM_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  const int64_t m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  f32x16_t acc[2][2];
  gemm_mainloop<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + (lane & 31);
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int64_t mbase = m0 + wm * 64 + mi * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
        if (m >= M) continue;
        float v = acc[mi][ni][r] + bv;
        if ((act & 0xff) == OM_ACT_GELU_ERF_GRAD) {          // v = acc * gelu'(resid)
          v *= gelu_erf_grad(ElemOps<OutT>::load(resid + m * ldr + n));
        } else {
          if (pre_act) ElemOps<OutT>::store(pre_act + m * ep.ldp + n, v);
          v = act_apply(v, act & 0xff);
          if (drop_thresh)
            v = dropout_keep(ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)n, drop_thresh) ? v * drop_scale : 0.f;
          if (resid) {
            const float rv = ElemOps<OutT>::load(resid + m * ldr + n);
            v = (act & OM_ACT_MUL_RESID) ? v * rv : v + rv;
          }
        }
        ElemOps<OutT>::store(C + m * ldc + n, v);
      }
    }
  }
}

// ---- v2: 256x128 tile, 3-deep LDS ring, LDS-staged epilogue (whole 16-byte row segments) ----
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

template <typename T, typename OutT>
__global__ __launch_bounds__(G2_THREADS) void gemm_nt_kernel2(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g2_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[2][2];
  gemm_mainloop2<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc);

  // The patch is staged in f32 whatever the output type, so bias/activation/dropout/residual are
  // all applied in f32 and the result is rounded ONCE (staging in bf16 would double-round).
  constexpr int VEC = OutVec<OutT>::VEC;
  constexpr int STRIDE = 64 * 4 + 16;           // +16 B: rows of a lane group land on different banks
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const float* __restrict__ bias = ep.bias;
  const OutT* resid = (const OutT*)ep.resid;
  OutT* pre_act = (OutT*)ep.pre_act;
  const int act = ep.act & 0xff;
  const bool mul_resid = (ep.act & OM_ACT_MUL_RESID) != 0;
  const uint32_t drop_thresh = ep.drop_p > 0.f ? (uint32_t)(ep.drop_p * 4294967296.0) : 0u;
  const float drop_scale = ep.drop_p > 0.f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;

  __syncthreads();                                // every wave is done reading the ring
  char* region = smem + wave * (64 * STRIDE);     // this wave's 64x64 output patch
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + l31;
    const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[mi][ni][r] + bv;
        if (act != OM_ACT_GELU_ERF_GRAD) {
          const int64_t m = m0 + wm * 64 + row;
          if (pre_act && m < M && n < N) ElemOps<OutT>::store(pre_act + m * ep.ldp + n, v);
          v = act_apply(v, act);
          if (drop_thresh)
            v = dropout_keep(ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)n, drop_thresh) ? v * drop_scale : 0.f;
        }
        ((float*)(region + row * STRIDE))[ni * 32 + l31] = v;
      }
  }
  __syncthreads();
  constexpr int CPR = 64 / VEC;                   // 16-byte chunks per patch row
#pragma unroll
  for (int it = 0; it < CPR; ++it) {
    const int id = it * 64 + lane;
    const int row = id / CPR, c = id % CPR;
    const int64_t m = m0 + wm * 64 + row;
    const int64_t n = n0 + wn * 64 + c * VEC;
    if (m < M && n < N) {
      float xv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const f32x4_t t4 = *(const f32x4_t*)(region + row * STRIDE + (c * VEC + e) * 4);
        xv[e] = t4[0]; xv[e + 1] = t4[1]; xv[e + 2] = t4[2]; xv[e + 3] = t4[3];
      }
      if (resid) {
        float rv[VEC];
        OutVec<OutT>::unpack(*(const uint4*)(resid + m * ep.ldr + n), rv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (act == OM_ACT_GELU_ERF_GRAD) xv[e] *= gelu_erf_grad(rv[e]);
          else xv[e] = mul_resid ? xv[e] * rv[e] : xv[e] + rv[e];
        }
      }
      *(uint4*)(C + m * ldc + n) = OutVec<OutT>::pack(xv);
    }
  }
}

template <typename T, typename OutT>
static int launch_gemm2(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                        int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t ntm = (M + G2_BM - 1) / G2_BM, ntn = (N + G2_BN - 1) / G2_BN;
  const int64_t nwg = ntm * ntn;
  if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel2<T, OutT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    attr_set = true;
  }
  const int group_m = 8;
  const int tclass = sizeof(T) == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32;
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(tclass, s);
  hipLaunchKernelGGL((gemm_nt_kernel2<T, OutT>), dim3((unsigned)nwg), dim3(G2_THREADS), G2_LDS_BYTES, s,
                     (const T*)A, lda, (const T*)B, ldb, (OutT*)C, ldc, M, N, K, ep, group_m);
  if (timing) om_timing_end(tclass, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

// The 256x128 kernel needs whole 16-byte output segments; small or ragged problems use v1.
static bool use_v2(int out_dtype, const void* C, int64_t ldc, int64_t M, int64_t N, const GemmEpilogue& ep) {
  static const bool off = getenv("OM_GEMM_V1") != nullptr;
  if (off) return false;
  const int64_t vec = out_dtype == OM_F32 ? 4 : 8;
  if (M < 512 || N % vec || ldc % vec || ((uintptr_t)C & 15)) return false;
  if (ep.resid && (ep.ldr % vec || ((uintptr_t)ep.resid & 15))) return false;
  return true;
}

template <typename T, typename OutT>
static int launch_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t ntm = (M + GEMM_BM - 1) / GEMM_BM, ntn = (N + GEMM_BN - 1) / GEMM_BN;
  const int64_t nwg = ntm * ntn;
  if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    attr_set = true;
  }
  // sweep order: keep `group_m` activation row-tiles resident while walking the weight tiles
  const int group_m = 8;
  const int tclass = sizeof(T) == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32;  // (f16 counts as 16-bit)
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(tclass, s);
  hipLaunchKernelGGL((gemm_nt_kernel<T, OutT>), dim3((unsigned)nwg), dim3(GEMM_THREADS),
                     GEMM_LDS_BYTES, s, (const T*)A, lda, (const T*)B, ldb, (OutT*)C, ldc, M, N, K,
                     ep, group_m);
  if (timing) om_timing_end(tclass, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_gemm(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep,
             hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0) OM_FAIL("K must be positive");
  const int64_t es = in_dtype == OM_F32 ? 4 : 2;
  if ((K * es) % GEMM_ROW_BYTES != 0) OM_FAIL("K*sizeof(elem) must be a multiple of 128 bytes");
  if ((lda * es) % 16 != 0 || (ldb * es) % 16 != 0) OM_FAIL("lda/ldb must keep rows 16-byte aligned");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) OM_FAIL("A/B must be 16-byte aligned");
  if ((ep.act & 0xff) == OM_ACT_GELU_ERF_GRAD && !ep.resid) OM_FAIL("gelu-grad epilogue needs resid");
  if (use_v2(out_dtype, C, ldc, M, N, ep)) {
    if (in_dtype == OM_BF16 && out_dtype == OM_BF16)
      return launch_gemm2<bf16_t, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    if (in_dtype == OM_BF16 && out_dtype == OM_F32)
      return launch_gemm2<bf16_t, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    if (in_dtype == OM_F32 && out_dtype == OM_F32)
      return launch_gemm2<float, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    if (in_dtype == OM_F16 && out_dtype == OM_F32)
      return launch_gemm2<f16_t, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  }
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16)
    return launch_gemm<bf16_t, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_BF16 && out_dtype == OM_F32)
    return launch_gemm<bf16_t, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F32 && out_dtype == OM_F32)
    return launch_gemm<float, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F32 && out_dtype == OM_BF16)
    return launch_gemm<float, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F16 && out_dtype == OM_F32)
    return launch_gemm<f16_t, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  OM_FAIL("unsupported dtype combination");
}

extern "C" int om_gemm_nt(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                          int out_dtype, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const float* bias, const void* resid, int64_t ldr, int act,
                          void* stream) {
  GemmEpilogue ep = {};
  ep.bias = bias; ep.resid = resid; ep.ldr = ldr; ep.act = act;
  return omk_gemm(in_dtype, A, lda, B, ldb, out_dtype, C, ldc, M, N, K, ep, (hipStream_t)stream);
}
