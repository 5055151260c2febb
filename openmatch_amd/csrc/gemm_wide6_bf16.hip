// om_gemm_nt, tile generation 6 (gemm_core6.h), 16-bit (bf16 / f16) inputs.
#include "gemm_wide6.h"

// Variants with a dedicated wide kernel; anything else stays on the older generations.
bool omk_gemm_wide6_b16_has(int in_dtype, int out_dtype, int act, bool train, bool resid) {
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16) return launch6_has(act, resid);
  if ((in_dtype == OM_BF16 || in_dtype == OM_F16) && out_dtype == OM_F32) return act == OM_ACT_NONE && !train && !resid;
  return false;
}

int omk_gemm_wide6_b16(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
                       void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff;
  const bool train = ep.pre_act != nullptr || ep.drop_p > 0.f;
  const bool resid = ep.resid != nullptr;
  if (ep.ln_stats || ep.rln_stats || ep.stats_out) {     // normalisation fused across GEMMs: dedicated variants
    if (in_dtype != OM_BF16 || out_dtype != OM_BF16 || train) OM_FAIL("fused LayerNorm epilogue: bf16 inference only");
    const bool a_side = ep.ln_stats != nullptr, out_side = ep.rln_stats != nullptr || ep.stats_out != nullptr;
    if (a_side && out_side) OM_FAIL("fused LayerNorm epilogue: either the A side or the output side");
#define OM_LNF(A_, R_, F_) return launch6<bf16_t, bf16_t, A_, false, R_, F_>(A, lda, B, ldb, C, ldc, M, N, K, ep, s)
    if (a_side) {                                         // consumes a raw pre-norm tensor
      if (act == OM_ACT_NONE && !resid) OM_LNF(OM_ACT_NONE, false, 1);
      if (act == OM_ACT_GELU_ERF && !resid) OM_LNF(OM_ACT_GELU_ERF, false, 1);
      if (act == OM_ACT_RELU && !resid) OM_LNF(OM_ACT_RELU, false, 1);
      if (act == OM_ACT_GELU_TANH && !resid) OM_LNF(OM_ACT_GELU_TANH, false, 1);
      if (act == OM_ACT_GELU_TANH && resid) OM_LNF(OM_ACT_GELU_TANH, true, 1);         // T5 gated: act(.) * gate
    } else {                                              // produces one: row statistics, normalised residual
      if (act == OM_ACT_NONE && resid) OM_LNF(OM_ACT_NONE, true, 2);
    }
#undef OM_LNF
    OM_FAIL("no fused-LayerNorm kernel for this activation");
  }
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16)
    return launch6_any<bf16_t, bf16_t>(act, train, resid, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_BF16 && out_dtype == OM_F32 && act == OM_ACT_NONE && !train && !resid)
    return launch6<bf16_t, float, OM_ACT_NONE, false, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F16 && out_dtype == OM_F32 && act == OM_ACT_NONE && !train && !resid)
    return launch6<f16_t, float, OM_ACT_NONE, false, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  OM_FAIL("no wide kernel for this dtype / epilogue combination");
}
