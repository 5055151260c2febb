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
  if (ep.ln_stats || ep.rln_stats || ep.stats_out) OM_FAIL("the fused-LayerNorm epilogues live in generation 7 (whole 256 x 256 tiles)");
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16)
    return launch6_any<bf16_t, bf16_t>(act, train, resid, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_BF16 && out_dtype == OM_F32 && act == OM_ACT_NONE && !train && !resid)
    return launch6<bf16_t, float, OM_ACT_NONE, false, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F16 && out_dtype == OM_F32 && act == OM_ACT_NONE && !train && !resid)
    return launch6<f16_t, float, OM_ACT_NONE, false, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  OM_FAIL("no wide kernel for this dtype / epilogue combination");
}
