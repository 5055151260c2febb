// om_gemm_nt, tile generation 7 (gemm_wide7.h): bf16 -> bf16 inference epilogues without fused LayerNorm.
#include "gemm_wide7.h"

bool omk_gemm_wide7_has(int act, bool resid, int lnf) {
  if (lnf == 2 || lnf == 3) return act == OM_ACT_NONE && resid;
  switch (act) {
    case OM_ACT_NONE: return lnf == 0 || !resid;
    case OM_ACT_GELU_TANH: return true;
    case OM_ACT_GELU_ERF: case OM_ACT_RELU: return !resid;
  }
  return false;
}

int omk_gemm_wide7_ln(int act, bool resid, int lnf, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);

// persist = false: one tile per workgroup (A/B measurements of the cross-tile prefetch; two variants only)
int omk_gemm_wide7(bool persist, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff;
  const bool resid = ep.resid != nullptr;
  const int lnf = ep.ln_stats ? 1 : ((ep.rln_stats || ep.stats_out) ? (ep.out_lo ? 3 : 2) : 0);
  if (M % 256 || N % 256 || (K * 2) % G7_ROW_BYTES) OM_FAIL("generation 7 takes whole 256 x 256 tiles and 128-byte K steps");
  if (lnf >= 2 && !ep.stats_out) OM_FAIL("the output-side LayerNorm variant writes row statistics: stats_out is null");
  if ((ep.out_lo || ep.resid_lo) && lnf != 3) OM_FAIL("two-plane residual stream: only with the output-side LayerNorm epilogue");
  if (ep.ln_stats && (ep.rln_stats || ep.stats_out)) OM_FAIL("fused LayerNorm: either the A side or the output side");
  if (lnf) return omk_gemm_wide7_ln(act, resid, lnf, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (!persist) {
    const int64_t ntiles = (M / 256) * (N / 256);
#define OM_L7NP(A_)                                                                                                   \
  do {                                                                                                                \
    static std::atomic<bool> attr_set{false};                                                                                     \
    if (!attr_set) {                                                                                                  \
      OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel7<bf16_t, A_, false, 0, false>,                           \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));                          \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((gemm_nt_kernel7<bf16_t, A_, false, 0, false>), dim3((unsigned)ntiles), dim3(G6_THREADS), G7_LDS_BYTES, s, \
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, M, N, K, ep, 8);                \
    OM_LAUNCH_CHECK();                                                                                                \
    return 0;                                                                                                         \
  } while (0)
    if (act == OM_ACT_NONE && !resid) OM_L7NP(OM_ACT_NONE);
    if (act == OM_ACT_GELU_ERF && !resid) OM_L7NP(OM_ACT_GELU_ERF);
#undef OM_L7NP
  }
#define OM_L7(A_, R_) return launch7<bf16_t, A_, R_, 0>(A, lda, B, ldb, C, ldc, M, N, K, ep, s)
  switch (act) {
    case OM_ACT_NONE:      if (resid) OM_L7(OM_ACT_NONE, true); else OM_L7(OM_ACT_NONE, false);
    case OM_ACT_GELU_TANH: if (resid) OM_L7(OM_ACT_GELU_TANH, true); else OM_L7(OM_ACT_GELU_TANH, false);
    case OM_ACT_GELU_ERF:  if (!resid) OM_L7(OM_ACT_GELU_ERF, false); break;
    case OM_ACT_RELU:      if (!resid) OM_L7(OM_ACT_RELU, false); break;
  }
#undef OM_L7
  OM_FAIL("no generation-7 kernel for this epilogue");
}

// The training forward's FFN1: C = gelu(A B^T + bias) and ep.pre_act = gelu'(A B^T + bias), both as whole lines (kernel 7c16, TRAIN)
bool omk_gemm_wide7_train_ok(int64_t M, int64_t N, int64_t K, int64_t ldc, const GemmEpilogue& ep) {
  return (om_option(OM_OPT_GEMM_CONT) & 32) && M % 256 == 0 && N % 256 == 0 && (K * 2) % G7_ROW_BYTES == 0 && K * 2 >= 3 * G7_ROW_BYTES &&
         (ep.act & 0xff) == OM_ACT_GELU_ERF && (ep.act & OM_ACT_PRE_GRAD) && !(ep.act & OM_ACT_MUL_RESID) && ep.pre_act && ep.ldp == ldc &&
         !ep.resid && ep.drop_p == 0.f && (((uintptr_t)ep.pre_act | (uintptr_t)ep.bias) & 15) == 0 && !ep.ln_stats && !ep.rln_stats && !ep.stats_out;
}
int omk_gemm_wide7_train_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                             int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);      // gemm_wide7_f16.hip
int omk_gemm_wide7_train(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                         int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  if (dtype == OM_F16) return omk_gemm_wide7_train_f16(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  return launch7c<bf16_t, OM_ACT_GELU_ERF, 0, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
}
