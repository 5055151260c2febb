// om_encoder_forward: the whole eval-mode encoder (BERT post-LN or T5 pre-RMSNorm stack)
// + pooling + LinearHead + normalise, as a fixed sequence of launches on ONE stream.
// Stands in for DRModel.encode (modeling/dense_retrieval_model.py:133-155) and the HF model
// it calls (HF:models/bert/modeling_bert.py:623-684 / HF:models/t5/modeling_t5.py T5Stack).
//
// Activation buffers live in the caller's workspace ([M = B*L tokens] x width, compute dtype):
//   x   [M,H]   hidden state entering a layer (LN output; T5: residual stream)
//   y   [M,H]   pre-LayerNorm sum (BERT) / normed input (T5)
//   x1  [M,H]   post-attention hidden (BERT)
//   qkv [M,3H]  fused projection      ctx [M,H] attention output
//   ff  [M,F]   FFN inner activation  ff2 [M,F] gate (T5 v1.1 only)
#include <math.h>

#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "kernels.h"

// Folded LayerNorm weights (W * gamma, its row sums, the folded bias) depend on the weights alone.  A caller that
// keeps weights across forwards folds them ONCE into a buffer it owns (om_encoder_fold_bytes / om_encoder_fold_weights,
// OmEncoderWeights::folded) -- 23 launches of ln_fold_kernel per forward gone; without one they are folded per forward
// into workspace scratch.  (Round 2 kept a library-owned cache keyed by weight pointers, which allocated inside a forward
// and could serve stale folds after an in-place weight update: gone.)
//
// Blob layout, per layer l: [qkv: W' (3H x H, 16-bit) | column sums (3H f32) | folded bias (3H f32)]
//                           [ffn1: W' (F x H)         | column sums (F f32)  | folded bias (F f32)], every part 256-byte aligned.
// BERT: qkv of layer l is folded with LN2 of layer l-1 (layer 0's slot is unused), ffn1 with LN1 of layer l.
// T5 (not gated): qkv with the layer's first RMSNorm weight (l >= 1), ffn1 with its second.
namespace {
struct FoldSlot { size_t w, cs, bf; };
struct FoldLayout { size_t per_layer, total; FoldSlot qkv, ffn1; };
FoldLayout fold_layout(const OmEncoderConfig* c) {
  const size_t H = c->hidden, F = c->ffn;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  FoldLayout L;
  L.qkv.w = take(3 * H * H * 2); L.qkv.cs = take(3 * H * 4); L.qkv.bf = take(3 * H * 4);
  L.ffn1.w = take(F * H * 2); L.ffn1.cs = take(F * 4); L.ffn1.bf = take(F * 4);
  L.per_layer = off;
  L.total = off * (size_t)c->n_layers;
  return L;
}
bool fold_applies(const OmEncoderConfig* c) {
  return (c->dtype == OM_BF16 || c->dtype == OM_F16) && c->n_layers > 0 && c->hidden % 256 == 0 && c->ffn % 256 == 0;
}
}  // namespace

extern "C" size_t om_encoder_fold_bytes(const OmEncoderConfig* cfg) {
  if (!cfg || !fold_applies(cfg)) return 0;
  return fold_layout(cfg).total;
}

extern "C" int om_encoder_fold_weights(const OmEncoderConfig* c, const OmEncoderWeights* w, void* blob, size_t bytes, void* stream) {
  if (!c || !w || !blob) OM_FAIL("null argument");
  if (!fold_applies(c)) OM_FAIL("nothing to fold for this configuration (om_encoder_fold_bytes is 0)");
  const FoldLayout L = fold_layout(c);
  if (bytes < L.total || ((uintptr_t)blob & 255)) OM_FAIL("fold buffer too small or not 256-byte aligned");
  const OmLayerWeights* Ls = w->layers_host;
  if (!Ls) OM_FAIL("layers_host is null");
  hipStream_t s = (hipStream_t)stream;
  const int H = c->hidden, F = c->ffn;
  const bool bert = c->arch == OM_ARCH_BERT;
  for (int l = 0; l < c->n_layers; ++l) {
    char* base = (char*)blob + (size_t)l * L.per_layer;
    const OmLayerWeights& lw = Ls[l];
    if (l > 0) {
      const float* g = bert ? Ls[l - 1].ln2_g : lw.ln1_g;
      const float* beta = bert ? Ls[l - 1].ln2_b : nullptr;
      if (omk_ln_fold(c->dtype, lw.qkv_w, g, beta, bert ? lw.qkv_b : nullptr, base + L.qkv.w, (float*)(base + L.qkv.cs),
                      (float*)(base + L.qkv.bf), 3 * H, H, s)) return 1;
    }
    if (!lw.ffn1g_w) {
      const float* g = bert ? lw.ln1_g : lw.ln2_g;
      const float* beta = bert ? lw.ln1_b : nullptr;
      if (omk_ln_fold(c->dtype, lw.ffn1_w, g, beta, bert ? lw.ffn1_b : nullptr, base + L.ffn1.w, (float*)(base + L.ffn1.cs),
                      (float*)(base + L.ffn1.bf), F, H, s)) return 1;
    }
  }
  return 0;
}

// Wf / colsum / bf of one site: from the caller's fold buffer, or computed into workspace scratch
static int folded_weights(const OmEncoderConfig* c, const OmEncoderWeights* w, int layer, bool ffn1, const void* W, const float* g,
                          const float* beta, const float* b, int N, int K, void* scratch_w, float* scratch_cs, float* scratch_bf,
                          hipStream_t s, const void** Wf, const float** cs, const float** bf) {
  if (w->folded) {
    const FoldLayout L = fold_layout(c);
    const char* base = (const char*)w->folded + (size_t)layer * L.per_layer;
    const FoldSlot& sl = ffn1 ? L.ffn1 : L.qkv;
    *Wf = base + sl.w; *cs = (const float*)(base + sl.cs); *bf = (const float*)(base + sl.bf);
    return 0;
  }
  if (omk_ln_fold(c->dtype, W, g, beta, b, scratch_w, scratch_cs, scratch_bf, N, K, s)) return 1;
  *Wf = scratch_w; *cs = scratch_cs; *bf = scratch_bf;
  return 0;
}

struct EncWs {
  char *x, *y, *x1, *qkv, *ctx, *ff, *ff2;
  float *pooled, *headout, *posbias;
  float* final32;   // 16-bit runs: the last normalisation's output in f32 for the pooling tail (CLS rows, or all rows for mean pooling)
  int* lut;
  int* kmax;        // per batch row: 1 + its last unmasked key (omk_mask_extent), read by every layer's attention launch
  int *cu, *cls_rows, *row_map;   // packed rows (om_encoder_forward_packed): sequence offsets [B + 2], [CLS] row of each sequence [B], token of each row
  // fused-LayerNorm path (16-bit): folded weight, its column sums and bias, two statistics buffers per layer, the slot
  // partials one GEMM leaves (kernels.h: GemmEpilogue::stats_out), and the second plane of the two residual tensors
  char* wfold;
  float *colsum, *bfold, *stats1, *stats2, *slots;
  char *y_lo, *x1_lo;
  float *r32a, *r32b, *y32;   // few-rows 16-bit BERT (round 6): the f32 residual stream -- LayerNorm outputs and pre-LayerNorm sums as autocast keeps them
  int64_t Mp;       // row count the GEMMs run on: M rounded up to whole 256-row tiles (the buffers are that tall)
  size_t total;
};

static EncWs carve(const OmEncoderConfig* c, int64_t B, int64_t L, char* base, int64_t packed_rows = 0) {
  const bool half = c->dtype == OM_BF16 || c->dtype == OM_F16;
  const size_t es = half ? 2 : 4;
  // 16-bit batches of >= 512 tokens are padded to whole 256-row tiles: the persistent GEMM generation
  // (gemm_wide7.h) takes whole tiles only.  Rows are independent in every contraction, so whatever the pad rows
  // hold stays in the pad rows; every other kernel (embedding, attention, normalisation, pooling) sees B*L rows.
  const size_t Mreal = packed_rows > 0 ? (size_t)packed_rows : (size_t)B * L, H = c->hidden, F = c->ffn;
  const size_t M = (half && Mreal >= 512) ? (Mreal + 255) / 256 * 256 : Mreal;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  EncWs w;
  w.x = take(M * H * es);
  w.y = take(M * H * es);
  w.x1 = take(M * H * es);
  w.qkv = take(M * 3 * H * es);
  w.ctx = take(M * H * es);
  w.ff = take(M * F * es);
  w.ff2 = take(c->arch == OM_ARCH_T5 ? M * F * es : 0);
  w.pooled = (float*)take((size_t)B * H * 4);
  w.headout = (float*)take((size_t)B * (c->head_out > 0 ? c->head_out : 1) * 4);
  w.posbias = (float*)take(c->arch == OM_ARCH_T5 ? (size_t)c->n_heads * L * L * 4 : 0);
  w.lut = (int*)take(c->arch == OM_ARCH_T5 ? (size_t)(2 * L) * 4 : 0);
  w.kmax = (int*)take((size_t)B * 4);
  w.cu = (int*)take(packed_rows > 0 ? (size_t)(B + 2) * 4 : 0);
  w.cls_rows = (int*)take(packed_rows > 0 ? (size_t)B * 4 : 0);
  w.row_map = (int*)take(packed_rows > 0 ? (size_t)packed_rows * 4 : 0);
  w.final32 = (float*)take(half && c->pooling != OM_POOL_NONE ? (c->pooling == OM_POOL_FIRST ? (size_t)B : Mreal) * H * 4 : 0);
  const bool fuse = half;                             // fused-norm path (BERT LayerNorm / T5 RMSNorm)
  const size_t wide = std::max((size_t)3 * H, F);
  w.wfold = take(fuse ? wide * H * es : 0);
  w.colsum = (float*)take(fuse ? wide * 4 : 0);
  w.bfold = (float*)take(fuse ? wide * 4 : 0);
  // one (sum, sum of squares) buffer per LayerNorm site (written whole by omk_ln_stats_reduce: nothing to zero)
  w.stats1 = (float*)take(fuse ? (size_t)2 * c->n_layers * M * 8 : 0);
  w.stats2 = w.stats1 ? w.stats1 + (size_t)c->n_layers * M * 2 : nullptr;
  w.slots = (float*)take(fuse ? (size_t)2 * ((H + 255) / 256) * M * 8 : 0);
  const bool two = fuse && c->arch == OM_ARCH_BERT;      // both 16-bit formats (round 6): the planes exist whether or not the switch uses them
  w.y_lo = take(two ? M * H * es : 0);
  w.x1_lo = take(two ? M * H * es : 0);
  const bool few32 = half && c->arch == OM_ARCH_BERT && packed_rows == 0 && Mreal <= (size_t)std::max(0, om_option(OM_OPT_GEMM_SKINNY_M));
  w.r32a = (float*)take(few32 ? Mreal * H * 4 : 0);
  w.r32b = (float*)take(few32 ? Mreal * H * 4 : 0);
  w.y32 = (float*)take(few32 ? Mreal * H * 4 : 0);
  w.Mp = (int64_t)M;
  w.total = off;
  return w;
}

extern "C" size_t om_encoder_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L) {
  if (!cfg || B <= 0 || L <= 0) return 0;
  return carve(cfg, B, L, nullptr).total;
}

// Whether om_encoder_forward_packed would take (cfg, B, L, packed_rows) under the CURRENT run-time switches: the same tests the
// forward makes (fused 16-bit path for these widths and this row count, the 16-bit inference attention kernel).  The host layer
// asks before it chooses the packed entry, so that an A/B switch (OM_GEMM_VARIANT, OM_ENCODER_FUSED_LN = 0, OM_ATTENTION_FAST = 0,
// om_debug_gemm_gen) degrades a compact batch to the padded entry instead of failing the call.
extern "C" int om_encoder_packed_supported(const OmEncoderConfig* c, int gated_ffn, int64_t B, int64_t L, int64_t packed_rows) {
  if (!c || B <= 0 || L <= 0 || L > 1024 || packed_rows <= 0) return 0;      // (beyond 256 tokens: round 6, the online-softmax attention kernel takes packed rows)
  if (packed_rows % 256 || packed_rows < 512 || packed_rows > B * L + 255) return 0;
  // few rows: the padded entry's contractions take the weight-streaming kernel -- decided there on ITS row count B * L, so only a
  // batch whose PADDED form is that small is sent back (B = 64, L = 128 with 1 024 real tokens would otherwise run the tile kernels
  // over all 8 192 padded rows: ADVICE r5)
  if (B * L <= (int64_t)om_option(OM_OPT_GEMM_SKINNY_M)) return 0;
  const int dt = c->dtype;
  if (dt != OM_BF16 && dt != OM_F16) return 0;
  if (dt == OM_BF16 && !om_option(OM_OPT_ATTENTION_FAST)) return 0;
  if (om_option(OM_OPT_ENCODER_FUSED_LN) == 0 || c->n_layers <= 0 || c->hidden % 8) return 0;
  const int H = c->hidden, F = c->ffn;
  if (c->arch == OM_ARCH_BERT) { if (c->act != OM_ACT_GELU_ERF) return 0; }
  else if (c->arch == OM_ARCH_T5) { if (gated_ffn || (dt == OM_F16 && c->act != OM_ACT_RELU)) return 0; }
  else return 0;
  return omk_gemm_ln_fusable(dt, packed_rows, H, H) && omk_gemm_ln_fusable(dt, packed_rows, F, H) &&
         omk_gemm_ln_fusable(dt, packed_rows, 3 * H, H) && omk_gemm_ln_fusable(dt, packed_rows, H, F) ? 1 : 0;
}

extern "C" size_t om_encoder_workspace_bytes_packed(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows) {
  if (!cfg || B <= 0 || L <= 0 || packed_rows <= 0) return 0;
  return carve(cfg, B, L, nullptr, packed_rows).total;
}

extern "C" int om_t5_relative_bucket(int relative_position, int num_buckets, int max_distance) {
  // bidirectional: half the buckets for each sign; exact below max_exact, log-spaced above
  int nb = num_buckets / 2;
  int bucket = relative_position > 0 ? nb : 0;
  const int n = relative_position < 0 ? -relative_position : relative_position;
  const int max_exact = nb / 2;
  if (n < max_exact) return bucket + n;
  const double v = log((double)n / max_exact) / log((double)max_distance / max_exact) * (nb - max_exact);
  int large = max_exact + (int)v;
  if (large > nb - 1) large = nb - 1;
  return bucket + large;
}

static int check_cfg(const OmEncoderConfig* c) {
  if (c->dtype != OM_F32 && c->dtype != OM_BF16 && c->dtype != OM_F16) OM_FAIL("dtype must be OM_F32, OM_BF16 or OM_F16");
  // float16 (the reference's --fp16 = torch.cuda.amp float16, retriever/dense_retriever.py:76): BERT-family erf-GELU encoders, and
  // (round 5) T5 encoder stacks with ReLU / tanh-GELU feed-forwards.  As under the reference's autocast, nothing clamps: a checkpoint
  // whose feed-forward activations leave the float16 range overflows here as it does there (the caller picks OM_BF16 for those).
  if (c->dtype == OM_F16 && c->arch == OM_ARCH_BERT && c->act != OM_ACT_GELU_ERF) OM_FAIL("float16 mode: erf-GELU BERT-family encoders only");
  if (c->dtype == OM_F16 && c->arch == OM_ARCH_T5 && c->act != OM_ACT_RELU && c->act != OM_ACT_GELU_TANH)
    OM_FAIL("float16 mode: T5 feed-forwards with ReLU or tanh-GELU only");
  if (c->arch != OM_ARCH_BERT && c->arch != OM_ARCH_T5) OM_FAIL("unknown arch");
  if (c->head_dim != 64 || c->n_heads * 64 != c->hidden)
    OM_FAIL("only head_dim 64 with n_heads*64 == hidden is supported");
  const int es = c->dtype == OM_F32 ? 4 : 2;
  if ((c->hidden * es) % 128 || (c->ffn * es) % 128) OM_FAIL("hidden/ffn rows must be multiples of 128 bytes");
  if (c->head_in > 0 && ((c->head_in * 4) % 128 || c->head_in != c->hidden)) OM_FAIL("head_in must equal hidden");
  return 0;
}

// packed_rows > 0: the token axis holds only the rows up to each sequence's last unmasked token, back to back
// (om_encoder_forward_packed); every per-token kernel and contraction then runs over packed_rows rows instead of B * L.
bool omk_gemm_skinny_ok(int in_dtype, int out_dtype, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep);      // gemm_skinny.hip
// whether the four contractions of a BERT layer take their pending-LayerNorm forms on the few-rows kernel at this shape
static bool pending_ln_ok(int dt, int64_t M, int H, int F, int act) {
  static const float one = 1.f;
  GemmEpilogue a = {}, r = {};
  a.a_ln32 = &one; a.a_ln_g = &one; a.a_ln_b = &one;
  r.rln32 = &one; r.rln32_stats = &one; r.rln_g = &one; r.rln_b = &one; r.out32 = const_cast<float*>(&one);
  GemmEpilogue f = a;
  f.act = act;
  return omk_gemm_skinny_ok(dt, dt, M, 3 * (int64_t)H, H, a) && omk_gemm_skinny_ok(dt, dt, M, F, H, f) &&
         omk_gemm_skinny_ok(dt, dt, M, H, H, r) && omk_gemm_skinny_ok(dt, dt, M, H, F, r);
}

static int encoder_forward_impl(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                const int64_t* input_ids, const int64_t* attention_mask,
                                const int64_t* token_type_ids, int64_t B, int64_t L,
                                void* out_hidden, float* out_reps, void* workspace,
                                size_t workspace_bytes, void* stream, int64_t packed_rows) {
  if (!c || !w || !input_ids || !attention_mask) OM_FAIL("null argument");
  if (check_cfg(c)) return 1;
  if (B <= 0) return 0;
  if (L < 1 || L > 1024) OM_FAIL("sequence length must be in [1,1024]");
  if (!workspace || ((uintptr_t)workspace & 255)) OM_FAIL("workspace must be 256-byte aligned");
  const bool packed = packed_rows > 0;
  if (packed) {
    if (c->dtype == OM_F32 || out_hidden || c->pooling == OM_POOL_NONE || c->n_layers < 1)
      OM_FAIL("packed rows: 16-bit inference that returns representations only");
    if (packed_rows % 256 || packed_rows < 512 || packed_rows > B * L + 255) OM_FAIL("packed_rows: a multiple of 256 in [512, B * L + 255]");
  }
  EncWs ws = carve(c, B, L, (char*)workspace, packed_rows);
  if (ws.total > workspace_bytes) OM_FAIL("workspace too small");
  if (c->pooling != OM_POOL_NONE && !out_reps) OM_FAIL("out_reps required when pooling is set");
  hipStream_t s = (hipStream_t)stream;
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads;
  const int64_t M = packed ? packed_rows : B * L;
  const int* const cu = packed ? ws.cu : nullptr;
  // Few rows (a served query, a handful of sequences; round 5): up to OM_OPT_GEMM_SKINNY_M rows the contractions run on the
  // weight-streaming kernel (gemm_skinny.hip) with the normalisations as kernels -- the persistent 256 x 256 tiles of the fused path
  // put 6 ... 24 workgroups on 256 CUs there and take ~150 us per layer whatever the batch (profiles/r05_small_forward_*.txt).
  // (bfloat16 BERT from 512 rows on stays on the fused path: its two-plane residual stream is what holds bfloat16 inside the
  // reference's own autocast deviation, and the unfused path keeps one plane)
  const bool few_rows = !packed && dt != OM_F32 && M <= (int64_t)om_option(OM_OPT_GEMM_SKINNY_M) &&
                        !(dt == OM_BF16 && c->arch == OM_ARCH_BERT && M >= 512 && (om_option(OM_OPT_ENCODER_TWO_PLANE) & 1) != 0);
  const int64_t Mg = few_rows ? M : ws.Mp;          // rows of the contractions (M padded to whole tiles for large 16-bit batches)
  const bool bert = c->arch == OM_ARCH_BERT;
  const OmLayerWeights* Ls = w->layers_host;
  if (!Ls) OM_FAIL("layers_host is null");

#define GEMM(A_, lda_, W_, ldw_, C_, ldc_, N_, K_, bias_, res_, ldr_, act_)                      \
  do {                                                                                           \
    if (om_gemm_nt(dt, A_, lda_, W_, ldw_, dt, C_, ldc_, Mg, N_, K_, bias_, res_, ldr_, act_, s)) \
      return 1;                                                                                  \
  } while (0)
#define RUN(expr) do { if (expr) return 1; } while (0)

  RUN(omk_mask_extent(attention_mask, B, (int)L, ws.kmax, s));
  if (packed) RUN(omk_pack_rows(ws.kmax, B, (int)L, packed_rows, ws.cu, ws.cls_rows, ws.row_map, s));
  char* final_hidden = nullptr;
  // 16-bit runs that only return representations: the LAST normalisation writes f32 (the reference's autocast runs
  // layer_norm in fp32), into ws.final32 -- B CLS rows (pooling "first": already the pooled vectors) or all M rows
  int64_t final32_rows = 0;
  const float* f32_rows = ws.final32;       // where those rows are (the few-rows path leaves them in its f32 residual buffer)
  if (bert) {
    if (L > c->max_pos) OM_FAIL("sequence longer than the position table");
    // few rows, 16-bit (round 6): the residual stream in f32 -- the embedding leaves its LayerNorm output in both forms
    const bool few32 = few_rows && ws.y32 && c->n_layers > 0 && (om_option(OM_OPT_ENCODER_TWO_PLANE) & (dt == OM_BF16 ? 1 : 2)) != 0;
    RUN(omk_embed(dt, input_ids, token_type_ids, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g,
                  w->emb_ln_b, ws.x, M, (int)L, H, c->vocab, c->type_vocab, c->ln_eps, 1, s, packed ? ws.row_map : nullptr, few32 ? ws.r32a : nullptr));
    const float scale = 1.0f / sqrtf((float)c->head_dim);
    // LayerNorm fused across the GEMMs (bf16, large batches): the LayerNorm outputs are never
    // written.  The GEMM that produces a pre-LayerNorm sum y also accumulates its row statistics; the
    // GEMM that consumes LN(y) reads y itself against the folded weight W*gamma and rescales its
    // rows in the epilogue; the GEMM that adds LN(y) as a residual normalises it on the fly
    // (kernels.h: GemmEpilogue::ln_*, rln_*, stats_out).  23 of the 25 LayerNorm passes over
    // [M,H] disappear (the embedding LayerNorm and the last one stay).
    const bool no_fuse = om_option(OM_OPT_ENCODER_FUSED_LN) == 0;   // A/B switch (om_debug_option)
    const bool fuse = !no_fuse && !few_rows && c->act == OM_ACT_GELU_ERF && c->n_layers > 0 && H % 8 == 0 &&
                      omk_gemm_ln_fusable(dt, Mg, H, H) && omk_gemm_ln_fusable(dt, Mg, F, H) &&
                      omk_gemm_ln_fusable(dt, Mg, 3 * H, H) && omk_gemm_ln_fusable(dt, Mg, H, F);
    if (om_option(OM_OPT_ENCODER_DEBUG)) fprintf(stderr, "om_encoder_forward: M=%ld fused_ln=%d packed=%d\n", (long)M, (int)fuse, (int)packed);
    if (packed && !fuse) OM_FAIL("packed rows need the fused 16-bit path (hidden, ffn multiples of 256; erf-GELU)");
    if (fuse) {
      const float inv_h = 1.0f / (float)H;
      // Two-plane residual stream (bfloat16): y1 = ws.y + ws.y_lo, y2 = ws.x1 + ws.x1_lo; the GEMMs that consume LN(y)
      // as their A operand read the first plane, the residual adds and the final LayerNorm read both.
      // (OM_OPT_ENCODER_TWO_PLANE: bit 0 bfloat16 (round 3), bit 1 float16 (round 6) -- the headline format, whose single plane was
      // what kept it outside the reference's own float16 autocast on small-weight models: DESIGN.md section 2)
      const bool two = (om_option(OM_OPT_ENCODER_TWO_PLANE) & (dt == OM_BF16 ? 1 : 2)) != 0;
      // bit 2 (float16, opt-in): the second plane in eight bits -- + 2.6 % passages/s, the same cosine / dot ratios, one more swapped tie on
      // the config-1 fixture's MRR@10 (gemm_wide7.h kernel 7r16, LNF == 4)
      const int lo8 = two && dt == OM_F16 && (om_option(OM_OPT_ENCODER_TWO_PLANE) & 4) ? 1 : 0;
      const int nslots = 2 * (H / 256);
      // y1 lives in ws.y, y2 in ws.x1; ws.x is the embedding output (layer 0's input)
      // Ping-pong walk: every kernel of the chain starts on the rows its producer wrote last (reverse = 1 on every second
      // launch), so the head of each activation tensor (200-800 MB, far beyond the 32 MB of L2) is found in the 256 MB
      // memory-side cache instead of HBM.  OM_OPT_ENCODER_PINGPONG = 0 walks every kernel first row to last.
      const bool pingpong = om_option(OM_OPT_ENCODER_PINGPONG) != 0;
      int walk = 1;               // the embedding kernel wrote first row to last
#define OM_WALK() (pingpong ? (walk ^= 1, walk ^ 1) : 0)
      for (int l = 0; l < c->n_layers; ++l) {
        const OmLayerWeights& lw = Ls[l];
        float* st1 = ws.stats1 + (size_t)l * Mg * 2;                     // LN1 of this layer
        float* st2 = ws.stats2 + (size_t)l * Mg * 2;                     // LN2 of this layer
        const float* st2p = l ? ws.stats2 + (size_t)(l - 1) * Mg * 2 : nullptr;   // LN2 of the previous one
        GemmEpilogue e = {};
        // ---- QKV: x0 for the first layer, LN2_{l-1}(y2) folded afterwards
        if (l == 0) {
          e.bias = lw.qkv_b; e.reverse = OM_WALK();
          RUN(omk_gemm(dt, ws.x, H, lw.qkv_w, H, dt, ws.qkv, 3 * H, Mg, 3 * H, H, e, s));
        } else {
          const OmLayerWeights& pw = Ls[l - 1];
          const void* wf; const float *cs, *bfp;
          RUN(folded_weights(c, w, l, false, lw.qkv_w, pw.ln2_g, pw.ln2_b, lw.qkv_b, 3 * H, H, ws.wfold, ws.colsum, ws.bfold, s, &wf, &cs, &bfp));
          e.bias = bfp; e.ln_stats = st2p; e.ln_colsum = cs; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps; e.reverse = OM_WALK();
          RUN(omk_gemm(dt, ws.x1, H, wf, H, dt, ws.qkv, 3 * H, Mg, 3 * H, H, e, s));
        }
        RUN(omk_attention(dt, ws.qkv, ws.ctx, attention_mask, nullptr, B, (int)L, H, nh, scale, 0.f, 0, s, OM_WALK(), ws.kmax, cu));
        // ---- attention output + residual -> y1, statistics of LN1
        e = GemmEpilogue{};
        e.bias = lw.o_b; e.ldr = H; e.stats_out = ws.slots; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps;
        if (two) { e.out_lo = ws.y_lo; e.lo8 = lo8; }
        if (l == 0) {
          e.resid = ws.x;                    // the embedding output: one plane
        } else {
          const OmLayerWeights& pw = Ls[l - 1];
          e.resid = ws.x1; e.rln_stats = st2p; e.rln_g = pw.ln2_g; e.rln_b = pw.ln2_b;
          if (two) e.resid_lo = ws.x1_lo;
        }
        e.reverse = OM_WALK();
        RUN(omk_gemm(dt, ws.ctx, H, lw.o_w, H, dt, ws.y, H, Mg, H, H, e, s));
        RUN(omk_ln_stats_reduce(ws.slots, nslots, Mg, st1, s));
        // ---- FFN1 on LN1(y1), folded
        const void* wf1; const float *cs1, *bf1;
        RUN(folded_weights(c, w, l, true, lw.ffn1_w, lw.ln1_g, lw.ln1_b, lw.ffn1_b, F, H, ws.wfold, ws.colsum, ws.bfold, s, &wf1, &cs1, &bf1));
        e = GemmEpilogue{};
        e.bias = bf1; e.act = c->act; e.ln_stats = st1; e.ln_colsum = cs1; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps;
        e.reverse = OM_WALK();
        RUN(omk_gemm(dt, ws.y, H, wf1, H, dt, ws.ff, F, Mg, F, H, e, s));
        // ---- FFN2 + LN1(y1) as the residual -> y2, statistics of LN2
        e = GemmEpilogue{};
        e.bias = lw.ffn2_b; e.resid = ws.y; e.ldr = H; e.rln_stats = st1; e.rln_g = lw.ln1_g; e.rln_b = lw.ln1_b;
        e.stats_out = ws.slots; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps;
        if (two) { e.resid_lo = ws.y_lo; e.out_lo = ws.x1_lo; e.lo8 = lo8; }
        e.reverse = OM_WALK();
        RUN(omk_gemm(dt, ws.ff, F, lw.ffn2_w, F, dt, ws.x1, H, Mg, H, F, e, s));
        RUN(omk_ln_stats_reduce(ws.slots, nslots, Mg, st2, s));
      }
#undef OM_WALK
      const OmLayerWeights& last = Ls[c->n_layers - 1];
      const void* lo = two ? ws.x1_lo : nullptr;
      if (packed && c->pooling == OM_POOL_FIRST) {         // the [CLS] row of sequence b is packed row cu[b]
        RUN(omk_layernorm_f32out(dt, ws.x1, H, ws.final32, H, last.ln2_g, last.ln2_b, B, H, c->ln_eps, 0, s, lo, ws.cls_rows, lo8));
        final32_rows = B;
      } else if (!out_hidden && c->pooling == OM_POOL_FIRST) {    // only the [CLS] rows are ever read: normalised straight into f32
        RUN(omk_layernorm_f32out(dt, ws.x1, L * H, ws.final32, H, last.ln2_g, last.ln2_b, B, H, c->ln_eps, 0, s, lo, nullptr, lo8));
        final32_rows = B;
      } else if (!out_hidden && c->pooling != OM_POOL_NONE) {
        RUN(omk_layernorm_f32out(dt, ws.x1, H, ws.final32, H, last.ln2_g, last.ln2_b, M, H, c->ln_eps, 0, s, lo, nullptr, lo8));
        final32_rows = M;
      } else {
        void* dst = out_hidden ? out_hidden : (void*)ws.x;
        RUN(omk_layernorm(dt, ws.x1, H, dst, H, last.ln2_g, last.ln2_b, M, H, c->ln_eps, 0, s, lo, lo8));
        final_hidden = (char*)dst;
        // hidden states AND representations: the pooled rows still come from an f32 normalisation (the reference's autocast
        // returns fp32 from layer_norm), so encode() and forward() give the same representations
        if (c->pooling == OM_POOL_FIRST) {
          RUN(omk_layernorm_f32out(dt, ws.x1, L * H, ws.final32, H, last.ln2_g, last.ln2_b, B, H, c->ln_eps, 0, s, lo, nullptr, lo8));
          final32_rows = B;
        } else if (c->pooling != OM_POOL_NONE) {
          RUN(omk_layernorm_f32out(dt, ws.x1, H, ws.final32, H, last.ln2_g, last.ln2_b, M, H, c->ln_eps, 0, s, lo, nullptr, lo8));
          final32_rows = M;
        }
      }
    } else if (few32 && M <= (int64_t)om_option(OM_OPT_FEW_ROWS_LN_FUSE) && pending_ln_ok(dt, Mg, H, F, c->act)) {
      // A handful of rows (<= 64: a served query), 16-bit (round 6): the f32 residual stream of the branch below with its LayerNorms PENDING --
      // no normalisation launches between the embedding and the last layer (86 -> 63 dependent launches for bert-base; a launch costs
      // 5.3 us here whatever it does, profiles/r06_few_rows_graph_probe.json).  y_a / y_b hold the pre-LayerNorm sums in f32; the
      // contraction that consumes LN(y) normalises its operand rows itself and leaves (mean, rstd) per row, the one that adds LN(y)
      // re-derives the element from them (gemm_skinny.hip: a_ln32 / rln32; ln_row.h is the one definition of the arithmetic, so
      // the bits are those of the branch below -- test_few_rows_forward_*).
      float* const y_a = ws.y32;      // attention block's sum: ctx Wo^T + b + x
      float* const y_b = ws.r32b;     // feed-forward block's sum: ff W2^T + b + x1
      for (int l = 0; l < c->n_layers; ++l) {
        const OmLayerWeights& lw = Ls[l];
        float* const st1 = ws.stats1 + (size_t)l * M * 2;                          // (mean, rstd) of LN1 of this layer
        float* const st2p = l ? ws.stats2 + (size_t)(l - 1) * M * 2 : nullptr;     // ... of LN2 of the previous one
        GemmEpilogue e = {};
        e.bias = lw.qkv_b; e.ln_eps = c->ln_eps;
        if (l) { e.a_ln32 = y_b; e.a_ln_g = Ls[l - 1].ln2_g; e.a_ln_b = Ls[l - 1].ln2_b; e.a_ln_stats_out = st2p; }
        RUN(omk_gemm(dt, ws.x, H, lw.qkv_w, H, dt, ws.qkv, 3 * H, Mg, 3 * H, H, e, s));
        RUN(omk_attention(dt, ws.qkv, ws.ctx, attention_mask, nullptr, B, (int)L, H, nh, scale, 0.f, 0, s, 0, ws.kmax));
        e = GemmEpilogue{};
        e.bias = lw.o_b; e.ldr = H; e.out32 = y_a; e.ln_eps = c->ln_eps;
        if (l) { e.rln32 = y_b; e.rln32_stats = st2p; e.rln_g = Ls[l - 1].ln2_g; e.rln_b = Ls[l - 1].ln2_b; }
        else e.resid32 = ws.r32a;
        RUN(omk_gemm(dt, ws.ctx, H, lw.o_w, H, dt, ws.y, H, Mg, H, H, e, s));
        e = GemmEpilogue{};
        e.bias = lw.ffn1_b; e.act = c->act; e.ln_eps = c->ln_eps;
        e.a_ln32 = y_a; e.a_ln_g = lw.ln1_g; e.a_ln_b = lw.ln1_b; e.a_ln_stats_out = st1;
        RUN(omk_gemm(dt, ws.x1, H, lw.ffn1_w, H, dt, ws.ff, F, Mg, F, H, e, s));
        e = GemmEpilogue{};
        e.bias = lw.ffn2_b; e.ldr = H; e.out32 = y_b; e.ln_eps = c->ln_eps;
        e.rln32 = y_a; e.rln32_stats = st1; e.rln_g = lw.ln1_g; e.rln_b = lw.ln1_b;
        RUN(omk_gemm(dt, ws.ff, F, lw.ffn2_w, F, dt, ws.y, H, Mg, H, F, e, s));
      }
      const OmLayerWeights& last = Ls[c->n_layers - 1];
      void* dst = out_hidden ? out_hidden : (void*)ws.x;
      RUN(omk_layernorm_dual(dt, y_b, H, dst, ws.r32a, H, last.ln2_g, last.ln2_b, M, H, c->ln_eps, s));
      final_hidden = (char*)dst;
      if (c->pooling != OM_POOL_NONE) { f32_rows = ws.r32a; final32_rows = M; }
    } else if (few32) {
      // Few rows (a served query, a handful of sequences), 16-bit (round 6): the residual stream in f32, as the reference's autocast keeps
      // it (layer_norm runs and returns fp32; a 16-bit dense output + an fp32 LayerNorm output is an fp32 sum: HF:models/bert/modeling_bert.py
      // :289-293,347-351 under retriever/dense_retriever.py:76) -- what the two-plane stream is to the fused path.  Every LayerNorm writes
      // its output twice (16-bit: the next contraction's operand; f32: what the next residual add reads), the residual contractions add
      // the f32 copy and leave their sum in f32 (gemm_skinny.hip: resid32 / out32).  Rounds 4-5 kept one 16-bit plane here.
      for (int l = 0; l < c->n_layers; ++l) {
        const OmLayerWeights& lw = Ls[l];
        const bool last = l == c->n_layers - 1;
        GEMM(ws.x, H, lw.qkv_w, H, ws.qkv, 3 * H, 3 * H, H, lw.qkv_b, nullptr, 0, OM_ACT_NONE);
        RUN(omk_attention(dt, ws.qkv, ws.ctx, attention_mask, nullptr, B, (int)L, H, nh, scale, 0.f, 0, s, 0, ws.kmax));
        GemmEpilogue e = {};
        e.bias = lw.o_b; e.resid32 = ws.r32a; e.ldr = H; e.out32 = ws.y32;
        RUN(omk_gemm(dt, ws.ctx, H, lw.o_w, H, dt, ws.y, H, Mg, H, H, e, s));                  // y32 = ctx Wo^T + b + x (f32)
        RUN(omk_layernorm_dual(dt, ws.y32, H, ws.x1, ws.r32b, H, lw.ln1_g, lw.ln1_b, M, H, c->ln_eps, s));
        GEMM(ws.x1, H, lw.ffn1_w, H, ws.ff, F, F, H, lw.ffn1_b, nullptr, 0, c->act);
        e = GemmEpilogue{};
        e.bias = lw.ffn2_b; e.resid32 = ws.r32b; e.ldr = H; e.out32 = ws.y32;
        RUN(omk_gemm(dt, ws.ff, F, lw.ffn2_w, F, dt, ws.y, H, Mg, H, F, e, s));                // y32 = ff W2^T + b + x1 (f32)
        void* dst = (last && out_hidden) ? out_hidden : (void*)ws.x;
        RUN(omk_layernorm_dual(dt, ws.y32, H, dst, ws.r32a, H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, s));
        final_hidden = (char*)dst;
      }
      if (c->pooling != OM_POOL_NONE) {       // the pooled rows come from the f32 copy of the last normalisation (the reference's autocast returns fp32)
        f32_rows = ws.r32a;
        final32_rows = M;
      }
    } else
    for (int l = 0; l < c->n_layers; ++l) {
      const OmLayerWeights& lw = Ls[l];
      GEMM(ws.x, H, lw.qkv_w, H, ws.qkv, 3 * H, 3 * H, H, lw.qkv_b, nullptr, 0, OM_ACT_NONE);
      RUN(omk_attention(dt, ws.qkv, ws.ctx, attention_mask, nullptr, B, (int)L, H, nh, scale, 0.f, 0, s, 0, ws.kmax));
      GEMM(ws.ctx, H, lw.o_w, H, ws.y, H, H, H, lw.o_b, ws.x, H, OM_ACT_NONE);
      RUN(omk_layernorm(dt, ws.y, H, ws.x1, H, lw.ln1_g, lw.ln1_b, M, H, c->ln_eps, 0, s));
      GEMM(ws.x1, H, lw.ffn1_w, H, ws.ff, F, F, H, lw.ffn1_b, nullptr, 0, c->act);
      GEMM(ws.ff, F, lw.ffn2_w, F, ws.y, H, H, F, lw.ffn2_b, ws.x1, H, OM_ACT_NONE);
      const bool half16 = dt != OM_F32;
      if (l == c->n_layers - 1 && half16 && !out_hidden && c->pooling == OM_POOL_FIRST) {
        RUN(omk_layernorm_f32out(dt, ws.y, L * H, ws.final32, H, lw.ln2_g, lw.ln2_b, B, H, c->ln_eps, 0, s));
        final32_rows = B;
      } else if (l == c->n_layers - 1 && half16 && !out_hidden && c->pooling != OM_POOL_NONE) {
        RUN(omk_layernorm_f32out(dt, ws.y, H, ws.final32, H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, 0, s));
        final32_rows = M;
      } else {
        void* dst = (l == c->n_layers - 1 && out_hidden) ? out_hidden : (void*)ws.x;
        RUN(omk_layernorm(dt, ws.y, H, dst, H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, 0, s));
        final_hidden = (char*)dst;
        if (l == c->n_layers - 1 && half16 && c->pooling == OM_POOL_FIRST) {          // (with out_hidden: pooled rows from f32 all the same)
          RUN(omk_layernorm_f32out(dt, ws.y, L * H, ws.final32, H, lw.ln2_g, lw.ln2_b, B, H, c->ln_eps, 0, s));
          final32_rows = B;
        } else if (l == c->n_layers - 1 && half16 && c->pooling != OM_POOL_NONE) {
          RUN(omk_layernorm_f32out(dt, ws.y, H, ws.final32, H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, 0, s));
          final32_rows = M;
        }
      }
    }
    if (c->n_layers == 0) final_hidden = ws.x;
  } else {
    // relative-position bias, shared by all layers (table lives in block 0)
    if (!w->rel_bias || !w->final_ln_g) OM_FAIL("T5 needs rel_bias and final_ln_g");
    const int* lut = nullptr;                    // device-resident, cached per (L, buckets, max distance): no copy, no sync
    RUN(om_t5_lut_device((int)L, c->rel_buckets, c->rel_max_dist, &lut));
    RUN(omk_t5_bias(w->rel_bias, lut, ws.posbias, (int)L, nh, s));
    RUN(omk_embed(dt, input_ids, nullptr, w->word_emb, nullptr, nullptr, nullptr, nullptr, ws.x, M,
                  (int)L, H, c->vocab, 1, c->ln_eps, 0, s, packed ? ws.row_map : nullptr));
    // RMSNorm fused across the GEMMs (bf16, >= 512 tokens), the pre-norm counterpart of the BERT path
    // above: the GEMM that updates the residual stream x accumulates sum(x^2) per row; the GEMMs that
    // consume rms(x) * g read x itself against the folded weight W * g and scale their rows by
    // rsqrt(mean(x^2) + eps) in the epilogue.  Only the first and the final norm run as kernels.
    const bool no_fuse_t5 = om_option(OM_OPT_ENCODER_FUSED_LN) == 0;
    // (gated feed-forward layers keep the kernels: two folded GEMMs per norm measured 1 % slower, tools/gtr_bench.py)
    const bool fuse_t5 = !no_fuse_t5 && !few_rows && c->n_layers > 0 && !Ls[0].ffn1g_w && H % 8 == 0 && omk_gemm_ln_fusable(dt, Mg, H, H) &&
                         omk_gemm_ln_fusable(dt, Mg, F, H) && omk_gemm_ln_fusable(dt, Mg, 3 * H, H) &&
                         omk_gemm_ln_fusable(dt, Mg, H, F);
    if (om_option(OM_OPT_ENCODER_DEBUG)) fprintf(stderr, "om_encoder_forward (t5): M=%ld fused_norm=%d packed=%d\n", (long)M, (int)fuse_t5, (int)packed);
    if (packed && !fuse_t5) OM_FAIL("packed rows need the fused 16-bit path (T5: widths of 256, no gated feed-forward)");
    if (fuse_t5) {
      const float inv_h = 1.0f / (float)H;
      const int nslots = 2 * (H / 256);
      auto folded = [&](int layer_, bool ffn1_, const void* A_, const void* W_, const float* g_, const float* stats_, void* C_, int N_, int act_,
                        const void* res_, int64_t ldr_) -> int {
        const void* wf; const float *cs, *bfp;
        // (the gated feed-forward's two weights are not in the caller's fold buffer -- and gated layers are not fused at all)
        if (folded_weights(c, w, layer_, ffn1_, W_, g_, nullptr, nullptr, N_, H, ws.wfold, ws.colsum, ws.bfold, s, &wf, &cs, &bfp)) return 1;
        (void)cs; (void)bfp;            // RMSNorm: no mean, no shift -- only the folded weight is used
        GemmEpilogue e = {};
        e.act = act_; e.resid = res_; e.ldr = ldr_;
        e.ln_stats = stats_; e.ln_rms = 1; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps;
        return omk_gemm(dt, A_, H, wf, H, dt, C_, N_, Mg, N_, H, e, s);
      };
      for (int l = 0; l < c->n_layers; ++l) {
        const OmLayerWeights& lw = Ls[l];
        float* st1 = ws.stats1 + (size_t)l * Mg * 2;                     // sum(x1^2): input of the FFN norm
        float* st2 = ws.stats2 + (size_t)l * Mg * 2;                     // sum(x'^2): input of the next layer's first norm
        if (l == 0) {
          RUN(omk_layernorm(dt, ws.x, H, ws.y, H, lw.ln1_g, nullptr, M, H, c->ln_eps, 1, s));
          GEMM(ws.y, H, lw.qkv_w, H, ws.qkv, 3 * H, 3 * H, H, nullptr, nullptr, 0, OM_ACT_NONE);
        } else {
          RUN(folded(l, false, ws.x, lw.qkv_w, lw.ln1_g, ws.stats2 + (size_t)(l - 1) * Mg * 2, ws.qkv, 3 * H, OM_ACT_NONE, nullptr, 0));
        }
        RUN(omk_attention(dt, ws.qkv, ws.ctx, attention_mask, ws.posbias, B, (int)L, H, nh, 1.0f, 0.f, 0, s, 0, ws.kmax, cu));
        GemmEpilogue e = {};
        e.resid = ws.x; e.ldr = H; e.stats_out = ws.slots; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps;
        RUN(omk_gemm(dt, ws.ctx, H, lw.o_w, H, dt, ws.x, H, Mg, H, H, e, s));           // x += o(ctx), sum(x^2)
        RUN(omk_ln_stats_reduce(ws.slots, nslots, Mg, st1, s));
        RUN(folded(l, true, ws.x, lw.ffn1_w, lw.ln2_g, st1, ws.ff, F, c->act, nullptr, 0));      // (fuse_t5 excludes gated layers)
        e = GemmEpilogue{};
        e.resid = ws.x; e.ldr = H; e.stats_out = ws.slots; e.ln_inv_h = inv_h; e.ln_eps = c->ln_eps;
        RUN(omk_gemm(dt, ws.ff, F, lw.ffn2_w, F, dt, ws.x, H, Mg, H, F, e, s));         // x += wo(ff), sum(x^2)
        RUN(omk_ln_stats_reduce(ws.slots, nslots, Mg, st2, s));
      }
    } else
    for (int l = 0; l < c->n_layers; ++l) {
      const OmLayerWeights& lw = Ls[l];
      RUN(omk_layernorm(dt, ws.x, H, ws.y, H, lw.ln1_g, nullptr, M, H, c->ln_eps, 1, s));
      GEMM(ws.y, H, lw.qkv_w, H, ws.qkv, 3 * H, 3 * H, H, nullptr, nullptr, 0, OM_ACT_NONE);
      RUN(omk_attention(dt, ws.qkv, ws.ctx, attention_mask, ws.posbias, B, (int)L, H, nh, 1.0f, 0.f, 0, s, 0, ws.kmax));
      GEMM(ws.ctx, H, lw.o_w, H, ws.x, H, H, H, nullptr, ws.x, H, OM_ACT_NONE);  // x += o(ctx)
      RUN(omk_layernorm(dt, ws.x, H, ws.y, H, lw.ln2_g, nullptr, M, H, c->ln_eps, 1, s));
      if (lw.ffn1g_w) {
        GEMM(ws.y, H, lw.ffn1g_w, H, ws.ff2, F, F, H, nullptr, nullptr, 0, OM_ACT_NONE);
        GEMM(ws.y, H, lw.ffn1_w, H, ws.ff, F, F, H, nullptr, ws.ff2, F, c->act | OM_ACT_MUL_RESID);
      } else {
        GEMM(ws.y, H, lw.ffn1_w, H, ws.ff, F, F, H, nullptr, nullptr, 0, c->act);
      }
      GEMM(ws.ff, F, lw.ffn2_w, F, ws.x, H, H, F, nullptr, ws.x, H, OM_ACT_NONE);  // x += wo(ff)
    }
    if (packed && c->pooling == OM_POOL_FIRST) {
      RUN(omk_layernorm_f32out(dt, ws.x, H, ws.final32, H, w->final_ln_g, nullptr, B, H, c->ln_eps, 1, s, nullptr, ws.cls_rows));
      final32_rows = B;
    } else if (dt != OM_F32 && !out_hidden && c->pooling == OM_POOL_FIRST) {
      RUN(omk_layernorm_f32out(dt, ws.x, L * H, ws.final32, H, w->final_ln_g, nullptr, B, H, c->ln_eps, 1, s));
      final32_rows = B;
    } else if (dt != OM_F32 && !out_hidden && c->pooling != OM_POOL_NONE) {
      RUN(omk_layernorm_f32out(dt, ws.x, H, ws.final32, H, w->final_ln_g, nullptr, M, H, c->ln_eps, 1, s));
      final32_rows = M;
    } else {
      void* dst = out_hidden ? out_hidden : (void*)ws.y;
      RUN(omk_layernorm(dt, ws.x, H, dst, H, w->final_ln_g, nullptr, M, H, c->ln_eps, 1, s));
      final_hidden = (char*)dst;
      if (dt != OM_F32 && c->pooling == OM_POOL_FIRST) {
        RUN(omk_layernorm_f32out(dt, ws.x, L * H, ws.final32, H, w->final_ln_g, nullptr, B, H, c->ln_eps, 1, s));
        final32_rows = B;
      } else if (dt != OM_F32 && c->pooling != OM_POOL_NONE) {
        RUN(omk_layernorm_f32out(dt, ws.x, H, ws.final32, H, w->final_ln_g, nullptr, M, H, c->ln_eps, 1, s));
        final32_rows = M;
      }
    }
  }

  if (c->pooling != OM_POOL_NONE) {
    const bool head = c->head_in > 0 && w->head_w;
    float* pooled = head ? ws.pooled : out_reps;
    if (final32_rows == B && c->pooling == OM_POOL_FIRST && (f32_rows == ws.final32 || L == 1))
      OM_HIP(hipMemcpyAsync(pooled, f32_rows, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    else if (final32_rows == M)
      RUN(omk_pool(OM_F32, f32_rows, attention_mask, pooled, B, (int)L, H, c->pooling, s, cu));
    else
      RUN(omk_pool(dt, final_hidden, attention_mask, pooled, B, (int)L, H, c->pooling, s));
    int D = H;
    if (head) {
      D = c->head_out;
      if (om_gemm_nt(OM_F32, pooled, H, w->head_w, c->head_in, OM_F32, out_reps, D, B, D, c->head_in,
                     nullptr, nullptr, 0, OM_ACT_NONE, s))
        return 1;
    }
    if (c->normalize) RUN(omk_l2norm(out_reps, out_reps, B, D, s));
    if (packed) RUN(omk_pack_overflow_poison(ws.cu, B, packed_rows, out_reps, B * (int64_t)D, s));
  }
#undef GEMM
#undef RUN
  return 0;
}

extern "C" int om_encoder_forward(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                  const int64_t* input_ids, const int64_t* attention_mask,
                                  const int64_t* token_type_ids, int64_t B, int64_t L,
                                  void* out_hidden, float* out_reps, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return encoder_forward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, out_hidden, out_reps, workspace, workspace_bytes, stream, 0);
}

extern "C" int om_encoder_forward_packed(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                         const int64_t* input_ids, const int64_t* attention_mask,
                                         const int64_t* token_type_ids, int64_t B, int64_t L, int64_t packed_rows,
                                         float* out_reps, void* workspace, size_t workspace_bytes, void* stream) {
  if (packed_rows <= 0) OM_FAIL("packed_rows must be positive");
  return encoder_forward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, nullptr, out_reps, workspace, workspace_bytes, stream, packed_rows);
}
