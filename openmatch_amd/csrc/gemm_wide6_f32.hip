// om_gemm_nt, tile generation 6 (gemm_core6.h), f32 (exact 32x32x2 MFMA) inputs.
#include "gemm_wide6.h"

// Variants with a dedicated wide kernel; anything else stays on the older generations.
bool omk_gemm_wide6_f32_has(int in_dtype, int out_dtype, int act, bool train, bool resid) {
  return in_dtype == OM_F32 && out_dtype == OM_F32 && launch6_has(act, resid);
}

int omk_gemm_wide6_f32(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
                       void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff;
  const bool train = ep.pre_act != nullptr || ep.drop_p > 0.f;
  const bool resid = ep.resid != nullptr;
  if (in_dtype == OM_F32 && out_dtype == OM_F32)
    return launch6_any<float, float>(act, train, resid, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  OM_FAIL("no wide kernel for this dtype / epilogue combination");
}
