// Internal launchers shared between translation units (not part of the C ABI).
#pragma once
#include "common.h"

// LayerNorm folded into the consuming bf16 weight (elementwise.hip)
int omk_ln_fold(int dtype /* OM_BF16 | OM_F16 */, const void* W, const float* gamma, const float* beta, const float* b, void* Wf,
                float* colsum, float* bf, int N, int K, hipStream_t s);
int omk_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* g,
                  const float* b, int64_t M, int H, float eps, int rms, hipStream_t s,
                  const void* x_lo = nullptr /* second plane of a two-plane input: normalises x + x_lo */,
                  int lo8 = 0 /* 1: x_lo is the eight-bit plane of omk_lo8_offset (float16; ldx == H, H % 256 == 0) */);
int omk_layernorm_dual(int dtype, const float* x, int64_t ldx, void* y, float* y32, int64_t ldy, const float* g, const float* b,
                       int64_t M, int H, float eps, hipStream_t s);
int omk_layernorm_f32out(int dtype, const void* x, int64_t ldx, float* y, int64_t ldy, const float* g, const float* b,
                         int64_t M, int H, float eps, int rms, hipStream_t s, const void* x_lo = nullptr,
                         const int* rows = nullptr /* gather: output row r normalises input row rows[r] */,
                         int lo8 = 0 /* 1: x_lo is the eight-bit plane of omk_lo8_offset; ldx = H or a multiple of it (the CLS-row gather) */);
int omk_embed(int dtype, const int64_t* ids, const int64_t* type_ids, const float* word,
              const float* pos, const float* type, const float* g, const float* b, void* out,
              int64_t M, int L, int H, int vocab, int type_vocab, float eps, int bert,
              hipStream_t s, const int* row_map = nullptr /* packed rows: output row t embeds token row_map[t] of ids (-1: a zero row) */,
              float* out32 = nullptr /* BERT: the unrounded f32 copy of the normalised rows as well (few-rows path) */);
int omk_pool(int dtype, const void* x, const int64_t* mask, float* out, int64_t B, int L, int H,
             int mode, hipStream_t s, const int* cu = nullptr /* packed rows: sequence b is rows cu[b] .. of x */);
int omk_l2norm(const float* x, float* y, int64_t M, int D, hipStream_t s);
int omk_t5_bias(const float* table, const int* lut, float* out, int L, int heads, hipStream_t s);

// softmax(scale * Q K^T + mask [+ pos_bias]) V for every (batch, head); qkv is the fused
// projection output [B*L, 3H] (q | k | v), ctx is [B*L, H].  L <= 256, head_dim == 64.
int omk_attention(int dtype, const void* qkv, void* ctx, const int64_t* mask,
                  const float* pos_bias, int64_t B, int L, int H, int heads, float scale,
                  float drop_p, uint64_t seed, hipStream_t s, int reverse = 0 /* batch rows last to first */,
                  const int* kmax = nullptr /* omk_mask_extent: per batch row, 1 + its last unmasked key (16-bit kernels skip the key tiles past it) */,
                  const int* cu = nullptr /* packed rows: sequence b occupies rows cu[b] .. cu[b + 1] - 1 of qkv / ctx (L stays the mask's row pitch) */);
// packed rows (om_encoder_forward_packed): cu[0..B] = offsets of the sequences (kmax[b] rows each) clamped to `rows`, cu[B + 1] = the
// unclamped token count; cls_rows[b] = min(cu[b], rows - 1); row_map[t] = b * L + position of packed row t, -1 for the pad rows
int omk_pack_rows(const int* kmax, int64_t B, int L, int64_t rows, int* cu, int* cls_rows, int* row_map, hipStream_t s);
int omk_pack_overflow_poison(const int* cu, int64_t B, int64_t rows, float* out, int64_t n, hipStream_t s);
int omk_mask_extent(const int64_t* mask, int64_t B, int L, int* kmax, hipStream_t s);

// ---- extended GEMM epilogue (training) ---------------------------------------------------
// order: v = acc + bias ; [pre_act <- v] ; v = act(v) ; v = dropout(v) ; v = v (+|*) resid
// act == OM_ACT_GELU_ERF_GRAD:  v = (acc + bias) * gelu'(resid)   (backward through GELU)
#define OM_ACT_GELU_ERF_GRAD 4
struct GemmEpilogue {
  const float* bias;
  const void* resid;   // out dtype
  int64_t ldr;
  int act;
  void* pre_act;       // out dtype, optional: value before the activation
  int64_t ldp;
  float drop_p;        // 0 = no dropout
  uint64_t seed;
  const int* drop_rows;  // packed rows (training): drop_rows[m] = the token (b * L + position) output row m holds -- the dropout mask is keyed on
                         //   (token, column), so a packed step and the padded step of the same batch draw the SAME mask; NULL: the row itself
  unsigned long long* trace;  // debug: per-block phase timestamps (om_debug_gemm_trace), else NULL
  // ---- LayerNorm fused across GEMMs (16-bit inference path of the BERT encoder, v6 kernel only) ----
  // stats = [M][2] f32 (sum, sum of squares) of a row over `1 / ln_inv_h` columns.
  const float* ln_stats;     // the A operand is a RAW pre-LayerNorm tensor: C = LN(A) W^T + b computed as
  const float* ln_colsum;    //   rstd_m (A W'^T - mu_m s_n) + b'_n  with W' = W*gamma, s_n = sum_k W'_nk, bias = b'
  const float* rln_stats;    // the residual is a RAW pre-LayerNorm tensor: add LN(resid) with these statistics
  const float* rln_g;        //   and this affine (per output column)
  const float* rln_b;
  float* stats_out;          // row statistics of the output (the next LayerNorm's input): PARTIAL (sum, sum of squares) pairs,
                             //   [2 * N / 256 slots][M] float2 -- slot = (column tile, wave column), one plain store per (slot, row);
                             //   omk_ln_stats_reduce adds the slots in a fixed order into the [M][2] array the consumers read
  const void* resid_lo;      // two-plane residual stream (bf16 BERT inference): the residual is resid + resid_lo (NULL: one plane)
  void* out_lo;              //   and the output is written as C = round16(y), out_lo = round16(y - C); selects the LNF == 3 kernel
  const float* resid32;      // few-rows path (gemm_skinny.hip, round 6): the residual as f32 [M, ldr] instead of `resid` -- the reference's autocast keeps the
  float* out32;              //   LayerNorm outputs it adds in fp32 -- and the sum written as f32 [M, ldc] here instead of into C
  // few rows with PENDING LayerNorms (gemm_skinny.hip, round 6): the normalisations of a forward over a handful of rows are not launches of
  // their own -- the contraction that consumes a LayerNorm's output normalises its operand rows itself, the one that adds it as a residual
  // re-derives the element from the row's (mean, rstd)
  const float* a_ln32;       // the A operand is LN(a_ln32 [M, K] f32, row pitch K) with a_ln_g / a_ln_b and ln_eps (A itself is ignored); K = hidden size
  const float* a_ln_g;
  const float* a_ln_b;
  float* a_ln_stats_out;     //   (mean, rstd) per row [M][2], written by the first column block of every row block: what a later rln32 reads
  const float* rln32;        // the residual is LN(rln32 [M, ldr] f32) with rln32_stats [M][2] = (mean, rstd) and rln_g / rln_b (per output column)
  const float* rln32_stats;
  int lo8;                   //   1 (float16): both second planes are EIGHT-bit blobs (omk_lo8_offset below; the LNF == 4 kernel) instead of 16-bit matrices
  float ln_inv_h, ln_eps;
  int ln_rms;                // 1: the statistics describe a T5 RMSNorm (no mean, no shift): only sum of squares is used
  int reverse;               // 1: walk the output tiles from the last row block to the first (persistent 16-bit kernel only).
                             //   A consumer that starts where its producer finished finds those rows in the memory-side cache
                             //   (256 MB; the encoder's activations are 200-800 MB per tensor) -- encoder.hip alternates.
};
// Byte offset of element (m, n) of an [M, N] tensor's EIGHT-BIT second plane (float16 two-plane residual stream, round 6: gemm_wide7.h kernel
// 7r16 with LNF == 3 writes and reads it in its own lane order -- tile (256 x 256) -> wave (128 x 128) -> patch (32 rows x 64 columns) ->
// 64 lanes x 32 bytes; value = e5m2 of (y - hi) * 2^10).  M, N multiples of 256.
__host__ __device__ inline size_t omk_lo8_offset(int64_t m, int64_t n, int64_t N) {
  const int64_t tile = (m >> 8) * (N >> 8) + (n >> 8);
  const int wave = (int)(((m >> 7) & 1) * 2 + ((n >> 7) & 1));
  const int r = (int)(m & 127), c = (int)(n & 127);
  const int P = 2 * (r >> 5) + (c >> 6);
  const int G = 2 * ((r >> 4) & 1) + ((c >> 5) & 1);
  const int lane = ((c >> 2) & 3) * 16 + (r & 15);
  return ((size_t)((tile * 4 + wave) * 8 + P)) * 2048 + (size_t)((G >> 1) * 1024 + lane * 16 + (G & 1) * 8 + ((c >> 4) & 1) * 4 + (c & 3));
}
// (sum, sum of squares) per row from the slot partials a GEMM with stats_out left: out[m] = sum over slots, in slot order
int omk_ln_stats_reduce(const float* slots, int nslots, int64_t M, float* out, hipStream_t s);
// true when omk_gemm will run [M,N] x K (16-bit) on the kernel that implements the ln_* / rln_* / stats_out fields
bool omk_gemm_ln_fusable(int dtype, int64_t M, int64_t N, int64_t K);
unsigned long long* omk_debug_trace();
// C[N,K] += A[M,N]^T B[M,K] (+ bias[N] += colsum A) on row-major bf16 operands (gemm_tn.hip); _ok: shapes it takes
bool omk_gemm_tn_ok(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb);
int omk_gemm_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* bias,
                int64_t M, int64_t N, int64_t K, hipStream_t s);      // om_debug_gemm_trace buffer (NULL: off)
// the same for many problems over one token count in one launch (256 x 256 tiles, whole token axis, no atomics)
bool omk_gemm_tn_batch_ok(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb);
int omk_gemm_tn_batch(int dtype, const OmTnProblem* probs, int n, int64_t M, hipStream_t s);
// wide-tile generations, one translation unit each (gemm_wide6_*.hip, gemm_wide7*.hip); omk_gemm dispatches
bool omk_gemm_wide6_b16_has(int in_dtype, int out_dtype, int act, bool train, bool resid);
int omk_gemm_wide6_b16(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
                       void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);
bool omk_gemm_wide6_f32_has(int in_dtype, int out_dtype, int act, bool train, bool resid);
int omk_gemm_wide6_f32(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
                       void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);
int omk_gemm(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep,
             hipStream_t s);

// C (f32, pre-zeroed) += A · B^T with K sliced across workgroups (weight gradients)
int omk_gemm_splitk(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t s);

// Stateless counter-based dropout.  Forward and backward regenerate the same mask from (seed, idx); nothing is stored.
// One 64-bit hash serves FOUR consecutive elements (idx >> 2 names the group): bits 16 e .. 16 e + 15 decide element
// idx & 3 == e, kept iff the field >= p * 2^16.  The three 64-bit multiplies of the hash cost ~30 integer instructions;
// a lane that holds four consecutive elements (every vectorised kernel here) pays them once, not four times.  p is
// resolved to 2^-16 and the keep scale uses the same rounded value, so E[mask * scale] is exactly 1.
__host__ __device__ inline uint64_t om_hash64(uint64_t seed, uint64_t idx) {
  uint64_t x = idx * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return x;
}
struct DropCfg {
  uint32_t thresh;        // 0: no dropout
  float keep_scale;
  __host__ __device__ explicit DropCfg(float p) {
    thresh = p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
    if (thresh > 65535u) thresh = 65535u;
    keep_scale = 65536.0f / (float)(65536u - thresh);
  }
};
__host__ __device__ inline uint64_t dropout_bits(uint64_t seed, uint64_t group) { return om_hash64(seed, group); }
__host__ __device__ inline bool dropout_field(uint64_t bits, int e, uint32_t thresh) {
  return ((uint32_t)(bits >> (16 * e)) & 0xffffu) >= thresh;
}
__host__ __device__ inline bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
  return dropout_field(dropout_bits(seed, idx >> 2), (int)(idx & 3), thresh);
}
