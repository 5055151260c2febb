// om_gemm_nt, tile generation 7 (gemm_wide7.h): the variants with LayerNorm / RMSNorm fused across the GEMMs
// (LNF 1: the A operand is a raw pre-norm tensor; LNF 2: normalised residual + row statistics of the output).
#include "gemm_wide7.h"

int omk_gemm_wide7_ln(int act, bool resid, int lnf, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
#define OM_L7(A_, R_, F_) return launch7<bf16_t, A_, R_, F_>(A, lda, B, ldb, C, ldc, M, N, K, ep, s)
  if (lnf == 1) {
    if (act == OM_ACT_NONE && !resid) OM_L7(OM_ACT_NONE, false, 1);
    if (act == OM_ACT_GELU_ERF && !resid) OM_L7(OM_ACT_GELU_ERF, false, 1);
    if (act == OM_ACT_RELU && !resid) OM_L7(OM_ACT_RELU, false, 1);
    if (act == OM_ACT_GELU_TANH && !resid) OM_L7(OM_ACT_GELU_TANH, false, 1);
    if (act == OM_ACT_GELU_TANH && resid) OM_L7(OM_ACT_GELU_TANH, true, 1);         // T5 gated: act(.) * gate
  } else if (lnf == 2) {
    if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true, 2);
  } else if (lnf == 3) {                     // two-plane residual stream (GemmEpilogue::out_lo / resid_lo)
    if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true, 3);
  }
#undef OM_L7
  OM_FAIL("no generation-7 kernel for this fused-LayerNorm epilogue");
}
