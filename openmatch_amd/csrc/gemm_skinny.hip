// C[M,N] = epilogue(A[M,K] W[N,K]^T) for FEW rows (M <= a few hundred): the nn.Linear of a forward over one query or a handful of
// sequences -- what the reference runs per search request (retriever/dense_retriever.py:151 encodes the query batch; a served
// query is ONE 32-token row block).  Round 5.
//
// Why a kernel of its own: the tile generations of gemm.hip stream W through 128- or 256-column tiles, so at M = 32 a
// [768 x 3072] weight is read by SIX workgroups, 786 KB each, one K step after another: ~20 us per launch whatever the batch,
// x 84 launches = the 1.9 ms floor under every small forward (docs/history.md: "a captured graph changes nothing: the floor is
// the GPU-side chain").  The operation is a weight STREAM: 2 K N bytes once from HBM, M x that in flops, nothing to reuse.
// So: one workgroup per 16 output columns (N / 16 workgroups: 48 ... 192 on 256 CUs), its four waves split K four ways, every
// lane loads its MFMA fragment straight from global memory (16 bytes per lane: W row n0 + lane % 16, 8 consecutive k) -- no LDS
// staging, up to SK_UNR steps of loads in flight per wave before the first MFMA -- and the four partial tiles meet in LDS in a
// fixed order.  A row's result does not depend on M or on the rows beside it.
//
// Roofline: HBM; algorithmic bytes per launch = 2 N K (the weight) + 2 M (K + N) (activations in, out).
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {
constexpr int SK_THREADS = 256;
constexpr int SK_UNR = 8;          // K steps (32 elements each) a wave has in flight
typedef uint32_t sk_u32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct SkMma;
template <> struct SkMma<bf16_t> {
  __device__ static inline f32x4_t mma(const sk_u32x4_t& a, const sk_u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct SkMma<f16_t> {
  __device__ static inline f32x4_t mma(const sk_u32x4_t& a, const sk_u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ float sk_act(float v, int act) {
  switch (act) {
    case OM_ACT_GELU_ERF: return act_apply<OM_ACT_GELU_ERF, true>(v);
    case OM_ACT_RELU: return act_apply<OM_ACT_RELU, true>(v);
    case OM_ACT_GELU_TANH: return act_apply<OM_ACT_GELU_TANH, true>(v);
  }
  return v;
}

// grid (N / 16, ceil(M / (16 MT))); MFMA operand 1 = 16 rows of A, operand 2 = 16 rows of W (the output columns):
// lane l holds k = 8 (l >> 4) .. + 7 of row l & 15 of either; D[row 4 (l >> 4) + i][column l & 15].
template <typename T, int MT>
__global__ __launch_bounds__(SK_THREADS) void gemm_nt_skinny_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw,
                                                                   T* C, int64_t ldc, int M, int N, int K, const float* __restrict__ bias,
                                                                   const T* resid, int64_t ldr, int act, int mul) {
  __shared__ float red[4][MT][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
  const int ksl = K >> 2;                                   // this wave's K slice
  const T* wp = W + (int64_t)(n0 + r) * ldw + wave * ksl + kg * 8;
  const T* ap[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    int m = m0 + 16 * j + r;
    m = m < M ? m : M - 1;                                  // rows past M: computed from a valid row, never stored
    ap[j] = A + (int64_t)m * lda + wave * ksl + kg * 8;
  }
  f32x4_t acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int steps = ksl >> 5;
  for (int s0 = 0; s0 < steps; s0 += SK_UNR) {
    sk_u32x4_t wq[SK_UNR], aq[MT][SK_UNR];
#pragma unroll
    for (int u = 0; u < SK_UNR; ++u)
      if (s0 + u < steps) {
        wq[u] = __builtin_nontemporal_load((const sk_u32x4_t*)(wp + (s0 + u) * 32));      // the weight passes once
#pragma unroll
        for (int j = 0; j < MT; ++j) aq[j][u] = *(const sk_u32x4_t*)(ap[j] + (s0 + u) * 32);
      }
#pragma unroll
    for (int u = 0; u < SK_UNR; ++u)
      if (s0 + u < steps) {
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[j] = SkMma<T>::mma(aq[j][u], wq[u], acc[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][j][i][lane] = acc[j][i];
  __syncthreads();
  const int i = wave;                                        // each wave finishes one accumulator register of every tile
  const int n = n0 + r;
  const float b = bias ? bias[n] : 0.f;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + 16 * j + 4 * kg + i;
    if (m >= M) continue;
    float v = ((red[0][j][i][lane] + red[1][j][i][lane]) + red[2][j][i][lane]) + red[3][j][i][lane];
    v = sk_act(v + b, act);
    if (resid) {
      const float rv = ElemOps<T>::load(resid + (int64_t)m * ldr + n);      // may alias C: read and written by this lane only
      v = mul ? v * rv : v + rv;
    }
    ElemOps<T>::store(C + (int64_t)m * ldc + n, v);
  }
}

template <typename T>
int launch_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                  const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff, mul = (ep.act & OM_ACT_MUL_RESID) ? 1 : 0;
#define SK_GO(MT_)                                                                                                          \
  hipLaunchKernelGGL((gemm_nt_skinny_kernel<T, MT_>), dim3((unsigned)(N / 16), (unsigned)((M + 16 * MT_ - 1) / (16 * MT_))), \
                     dim3(SK_THREADS), 0, s, (const T*)A, lda, (const T*)W, ldw, (T*)C, ldc, (int)M, (int)N, (int)K, ep.bias, \
                     (const T*)ep.resid, ep.ldr, act, mul)
  if (M <= 16) SK_GO(1);
  else if (M <= 32) SK_GO(2);
  else SK_GO(4);
#undef SK_GO
  OM_LAUNCH_CHECK();
  return 0;
}
}  // namespace

// shapes and epilogues the kernel takes (omk_gemm asks before it picks a tile generation)
bool omk_gemm_skinny_ok(int in_dtype, int out_dtype, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep) {
  const int max_m = om_option(OM_OPT_GEMM_SKINNY_M);
  if (max_m <= 0 || M > max_m || M < 1) return false;
  if (in_dtype != out_dtype || (in_dtype != OM_BF16 && in_dtype != OM_F16)) return false;
  if (N % 16 || K % 128 || N > (int64_t)65535 * 16) return false;
  const int act = ep.act & 0xff;
  if (act != OM_ACT_NONE && act != OM_ACT_GELU_ERF && act != OM_ACT_RELU && act != OM_ACT_GELU_TANH) return false;
  if (ep.pre_act || ep.drop_p > 0.f || ep.ln_stats || ep.rln_stats || ep.stats_out || ep.resid_lo || ep.out_lo) return false;
  if ((ep.act & OM_ACT_MUL_RESID) && !ep.resid) return false;
  return true;
}

int omk_gemm_skinny(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  if (dtype == OM_F16) return launch_skinny<f16_t>(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
  return launch_skinny<bf16_t>(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
}
