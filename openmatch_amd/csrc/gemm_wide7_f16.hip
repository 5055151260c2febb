// om_gemm_nt, tile generation 7 (gemm_wide7.h) for IEEE half: the epilogues of the BERT-family inference encoder in its
// float16 mode (the reference's `--fp16` is torch.cuda.amp float16; retriever/dense_retriever.py:76,151).  Same kernels as
// the bfloat16 build -- the matrix core runs both formats at the same rate -- with three more mantissa bits in every
// stored activation: 1 - cos against the fp32 chain drops from 4e-5 to 6e-7 at bert-base.
#include "gemm_wide7.h"

bool omk_gemm_wide7_f16_has(int act, bool resid, int lnf) {
  if (lnf == 2 || lnf == 3 || lnf == 4) return act == OM_ACT_NONE && resid;
  if (lnf == 1) return !resid && (act == OM_ACT_NONE || act == OM_ACT_GELU_ERF || act == OM_ACT_RELU);
  return (act == OM_ACT_NONE) || ((act == OM_ACT_GELU_ERF || act == OM_ACT_RELU) && !resid);      // (T5's gated tanh-GELU layers: the generic tiles)
}

int omk_gemm_wide7_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff;
  const bool resid = ep.resid != nullptr;
  const int lnf = ep.ln_stats ? 1 : ((ep.rln_stats || ep.stats_out) ? (ep.out_lo ? (ep.lo8 ? 4 : 3) : 2) : 0);
  if (M % 256 || N % 256 || (K * 2) % G7_ROW_BYTES) OM_FAIL("generation 7 takes whole 256 x 256 tiles and 128-byte K steps");
  if (lnf >= 2 && !ep.stats_out) OM_FAIL("the output-side LayerNorm variant accumulates row statistics: stats_out is null");
  if ((ep.out_lo || ep.resid_lo) && lnf < 3) OM_FAIL("two-plane residual stream: only with the output-side LayerNorm epilogue");
  if (ep.ln_stats && (ep.rln_stats || ep.stats_out)) OM_FAIL("fused LayerNorm: either the A side or the output side");
#define OM_L7(A_, R_, F_) return launch7<f16_t, A_, R_, F_>(A, lda, B, ldb, C, ldc, M, N, K, ep, s)
  if (lnf == 2) {
    if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true, 2);
  } else if (lnf == 3) {                     // two-plane residual stream (round 6: GemmEpilogue::out_lo / resid_lo)
    if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true, 3);
  } else if (lnf == 4) {                     // ... with the eight-bit second plane (GemmEpilogue::lo8)
    if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true, 4);
  } else if (lnf == 1) {
    if (act == OM_ACT_NONE && !resid) OM_L7(OM_ACT_NONE, false, 1);
    if (act == OM_ACT_GELU_ERF && !resid) OM_L7(OM_ACT_GELU_ERF, false, 1);
    if (act == OM_ACT_RELU && !resid) OM_L7(OM_ACT_RELU, false, 1);                  // T5 (round 5): RMSNorm-folded wi + ReLU
  } else {
    if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true, 0);
    if (act == OM_ACT_NONE && !resid) OM_L7(OM_ACT_NONE, false, 0);
    if (act == OM_ACT_GELU_ERF && !resid) OM_L7(OM_ACT_GELU_ERF, false, 0);
    if (act == OM_ACT_RELU && !resid) OM_L7(OM_ACT_RELU, false, 0);
  }
#undef OM_L7
  OM_FAIL("no generation-7 float16 kernel for this epilogue");
}

// float16 training (round 5): FFN1 with gelu and gelu' in one epilogue (kernel 7c16, TRAIN)
int omk_gemm_wide7_train_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                             int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  return launch7c<f16_t, OM_ACT_GELU_ERF, 0, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
}
