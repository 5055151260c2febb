// om_encoder_train_forward / om_encoder_train_backward: the BERT encoder with its backward
// pass as fixed launch sequences on one stream.  Stands in for autograd through HF BertModel
// under DRModel.forward (modeling/dense_retrieval_model.py:89-131) + loss.backward().
//
// Forward = the inference sequence of encoder.hip with (a) every tensor the backward needs
// kept on a caller-owned "tape", (b) dropout at HF's four sites (embeddings, attention
// probabilities, attention output, FFN output) from a stateless counter hash.
// Backward = per layer, in reverse:  LayerNorm bwd -> dropout bwd -> {bias col-sum, wgrad GEMM on
// transposed operands, dgrad GEMM against the transposed weight (GELU' fused in its epilogue)}
// -> attention bwd -> QKV wgrad/dgrad; then the embedding LayerNorm bwd + table scatter.
// Every dense contraction runs on the NT MFMA GEMM (gemm.hip); wgrad outputs are f32.
#include <math.h>

#include "train_kernels.h"

namespace {
struct Dims { int64_t M, Mp, B, L; int H, F, nl, nh, D; size_t es; };

Dims dims_of(const OmEncoderConfig* c, int64_t B, int64_t L) {
  Dims d;
  d.B = B; d.L = L; d.M = B * L; d.Mp = (d.M + 63) / 64 * 64;
  d.H = c->hidden; d.F = c->ffn; d.nl = c->n_layers; d.nh = c->n_heads;
  d.D = c->head_in > 0 ? c->head_out : c->hidden;
  d.es = c->dtype == OM_BF16 ? 2 : 4;
  return d;
}

// ---- tape: activations saved by the forward ------------------------------------------------
struct Tape {
  char* x0pre;             // [M,H] embedding LayerNorm output BEFORE dropout (only if dropout)
  char* x;                 // [(nl+1)][M,H] layer inputs / final output
  char *qkv, *ctx, *y1, *x1, *f, *y2;   // per layer, strided by their per-layer size
  float *pooled, *headout;             // [B,H], [B,D] (pre-normalise)
  size_t total;
  size_t sx, sqkv, sf;     // per-layer strides in bytes
};
Tape carve_tape(const Dims& d, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  Tape t;
  t.sx = align_up((size_t)d.M * d.H * d.es, 256);
  t.sqkv = align_up((size_t)d.M * 3 * d.H * d.es, 256);
  t.sf = align_up((size_t)d.M * d.F * d.es, 256);
  t.x = take(t.sx * (d.nl + 1));
  t.qkv = take(t.sqkv * d.nl);
  t.ctx = take(t.sx * d.nl);
  t.y1 = take(t.sx * d.nl);
  t.x1 = take(t.sx * d.nl);
  t.f = take(t.sf * d.nl);
  t.y2 = take(t.sx * d.nl);
  t.pooled = (float*)take((size_t)d.B * d.H * 4);
  t.headout = (float*)take((size_t)d.B * d.D * 4);
  t.total = off;
  return t;
}

// ---- scratch of forward (g) and backward -----------------------------------------------------
struct Ws {
  char *g;                          // fwd: GELU output [M,F]
  char *dxa, *dxb, *dy, *dd, *df, *dqkv, *dctx;   // bwd activation gradients
  char *tl, *tr, *wt;               // transposed operands / transposed weight
  float *dhead, *dpooled;
  size_t total;
};
Ws carve_ws(const Dims& d, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  Ws w;
  const size_t mh = (size_t)d.M * d.H * d.es, mf = (size_t)d.M * d.F * d.es;
  const size_t wide = (size_t)std::max(3 * d.H, d.F);
  w.g = take(mf);
  w.dxa = take(mh); w.dxb = take(mh); w.dy = take(mh); w.dd = take(mh);
  w.df = take(mf); w.dqkv = take(3 * mh); w.dctx = take(mh);
  w.tl = take(wide * d.Mp * d.es);
  w.tr = take(wide * d.Mp * d.es);
  w.wt = take(wide * (size_t)std::max(d.H, d.F) * d.es);
  w.dhead = (float*)take((size_t)d.B * d.D * 4);
  w.dpooled = (float*)take((size_t)d.B * d.H * 4);
  w.total = off;
  return w;
}

int check_train_cfg(const OmEncoderConfig* c, int64_t L) {
  if (c->arch != OM_ARCH_BERT) OM_FAIL("training is implemented for the BERT encoder");
  if (c->dtype != OM_F32 && c->dtype != OM_BF16) OM_FAIL("dtype must be OM_F32 or OM_BF16");
  if (c->head_dim != 64 || c->n_heads * 64 != c->hidden) OM_FAIL("head_dim must be 64");
  if (L < 1 || L > 128) OM_FAIL("training supports sequence lengths up to 128");
  if (c->act != OM_ACT_GELU_ERF) OM_FAIL("training supports the erf-GELU FFN");
  const int es = c->dtype == OM_BF16 ? 2 : 4;
  if ((c->hidden * es) % 128 || (c->ffn * es) % 128) OM_FAIL("hidden/ffn rows must be multiples of 128 bytes");
  if (c->pooling != OM_POOL_FIRST && c->pooling != OM_POOL_MEAN) OM_FAIL("pooling must be first or mean");
  return 0;
}

inline uint64_t site_seed(uint64_t seed, int layer, int site) {
  return seed + 0x9E3779B97F4A7C15ull * (uint64_t)(16 * layer + site + 1);
}
}  // namespace

extern "C" size_t om_encoder_tape_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L) {
  if (!cfg || B <= 0 || L <= 0) return 0;
  return carve_tape(dims_of(cfg, B, L), nullptr).total;
}
extern "C" size_t om_encoder_train_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L) {
  if (!cfg || B <= 0 || L <= 0) return 0;
  return carve_ws(dims_of(cfg, B, L), nullptr).total;
}

#define RUN(expr) do { if (expr) return 1; } while (0)

extern "C" int om_encoder_train_forward(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                        const int64_t* input_ids, const int64_t* attention_mask,
                                        const int64_t* token_type_ids, int64_t B, int64_t L,
                                        float hidden_dropout, float attn_dropout, uint64_t seed,
                                        void* tape_mem, size_t tape_bytes, float* out_reps,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !w || !input_ids || !attention_mask || !tape_mem || !workspace || !out_reps) OM_FAIL("null argument");
  if (check_train_cfg(c, L)) return 1;
  if (B <= 0) return 0;
  if (((uintptr_t)tape_mem & 255) || ((uintptr_t)workspace & 255)) OM_FAIL("tape/workspace must be 256-byte aligned");
  const Dims d = dims_of(c, B, L);
  Tape t = carve_tape(d, (char*)tape_mem);
  Ws ws = carve_ws(d, (char*)workspace);
  if (t.total > tape_bytes || ws.total > workspace_bytes) OM_FAIL("tape or workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int dt = c->dtype, H = d.H, F = d.F;
  const int64_t M = d.M;
  const OmLayerWeights* Ls = w->layers_host;
  if (!Ls) OM_FAIL("layers_host is null");
  if (L > c->max_pos) OM_FAIL("sequence longer than the position table");

  RUN(omk_embed(dt, input_ids, token_type_ids, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g,
                w->emb_ln_b, t.x, M, (int)L, H, c->vocab, c->type_vocab, c->ln_eps, 1, s));
  if (hidden_dropout > 0.f) RUN(omk_dropout(dt, t.x, t.x, M * H, hidden_dropout, site_seed(seed, 0, 0), s));
  const float scale = 1.0f / sqrtf((float)c->head_dim);
  for (int l = 0; l < d.nl; ++l) {
    const OmLayerWeights& lw = Ls[l];
    char* x = t.x + t.sx * l;
    char* qkv = t.qkv + t.sqkv * l;
    char* ctx = t.ctx + t.sx * l;
    char* y1 = t.y1 + t.sx * l;
    char* x1 = t.x1 + t.sx * l;
    char* f = t.f + t.sf * l;
    char* y2 = t.y2 + t.sx * l;
    GemmEpilogue ep = {};
    ep.bias = lw.qkv_b;
    RUN(omk_gemm(dt, x, H, lw.qkv_w, H, dt, qkv, 3 * H, M, 3 * H, H, ep, s));
    RUN(omk_attention(dt, qkv, ctx, attention_mask, nullptr, B, (int)L, H, d.nh, scale, attn_dropout,
                      site_seed(seed, l, 2), s));
    ep = GemmEpilogue{};
    ep.bias = lw.o_b; ep.resid = x; ep.ldr = H; ep.drop_p = hidden_dropout; ep.seed = site_seed(seed, l, 3);
    RUN(omk_gemm(dt, ctx, H, lw.o_w, H, dt, y1, H, M, H, H, ep, s));
    RUN(omk_layernorm(dt, y1, H, x1, H, lw.ln1_g, lw.ln1_b, M, H, c->ln_eps, 0, s));
    ep = GemmEpilogue{};
    ep.bias = lw.ffn1_b; ep.act = OM_ACT_GELU_ERF; ep.pre_act = f; ep.ldp = F;
    RUN(omk_gemm(dt, x1, H, lw.ffn1_w, H, dt, ws.g, F, M, F, H, ep, s));
    ep = GemmEpilogue{};
    ep.bias = lw.ffn2_b; ep.resid = x1; ep.ldr = H; ep.drop_p = hidden_dropout; ep.seed = site_seed(seed, l, 4);
    RUN(omk_gemm(dt, ws.g, F, lw.ffn2_w, F, dt, y2, H, M, H, F, ep, s));
    RUN(omk_layernorm(dt, y2, H, t.x + t.sx * (l + 1), H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, 0, s));
  }
  const char* xf = t.x + t.sx * d.nl;
  const bool head = c->head_in > 0 && w->head_w;
  RUN(omk_pool(dt, xf, attention_mask, t.pooled, B, (int)L, H, c->pooling, s));
  float* pre = c->normalize ? t.headout : out_reps;      // value before F.normalize
  if (head) {
    if (om_gemm_nt(OM_F32, t.pooled, H, w->head_w, c->head_in, OM_F32, pre, d.D, B, d.D, c->head_in,
                   nullptr, nullptr, 0, OM_ACT_NONE, s)) return 1;
  } else {
    OM_HIP(hipMemcpyAsync(pre, t.pooled, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
  }
  if (c->normalize) RUN(omk_l2norm(t.headout, out_reps, B, d.D, s));
  return 0;
}

extern "C" int om_encoder_train_backward(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                         const int64_t* input_ids, const int64_t* attention_mask,
                                         const int64_t* token_type_ids, int64_t B, int64_t L,
                                         float hidden_dropout, float attn_dropout, uint64_t seed,
                                         const void* tape_mem, const float* d_reps,
                                         const OmEncoderGrads* g, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!c || !w || !g || !tape_mem || !d_reps || !workspace) OM_FAIL("null argument");
  if (check_train_cfg(c, L)) return 1;
  if (B <= 0) return 0;
  const Dims d = dims_of(c, B, L);
  Tape t = carve_tape(d, (char*)tape_mem);
  Ws ws = carve_ws(d, (char*)workspace);
  if (ws.total > workspace_bytes) OM_FAIL("workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int dt = c->dtype, H = d.H, F = d.F;
  const int64_t M = d.M, Mp = d.Mp;
  const OmLayerWeights* Ls = w->layers_host;
  const OmLayerGrads* Gs = g->layers_host;
  if (!Ls || !Gs) OM_FAIL("layers_host is null");
  const float scale = 1.0f / sqrtf((float)c->head_dim);

  // ---- tail: normalise -> head -> pooling ----------------------------------------------------
  const bool head = c->head_in > 0 && w->head_w;
  const float* dhead = d_reps;
  if (c->normalize) {
    RUN(omk_l2norm_bwd(t.headout, d_reps, ws.dhead, B, d.D, s));
    dhead = ws.dhead;
  }
  const float* dpooled = dhead;
  if (head) {
    if (g->head_w) RUN(omk_small_tn(dhead, t.pooled, g->head_w, (int)B, d.D, c->head_in, s));   // dW = dY^T X
    RUN(omk_small_nn(dhead, w->head_w, ws.dpooled, (int)B, d.D, c->head_in, s));                 // dX = dY W
    dpooled = ws.dpooled;
  }
  char* dx = ws.dxa;       // gradient w.r.t. the current layer's OUTPUT
  char* dx_prev = ws.dxb;  // gradient w.r.t. its input (next iteration's dx)
  RUN(omk_pool_bwd(dt, dpooled, attention_mask, dx, B, (int)L, H, c->pooling, s));

  // weight gradients: split-K over the token axis, f32 atomics into the caller-zeroed buffers
#define WGRAD(left, right, rows_out, cols_in, dst) \
  RUN(omk_gemm_splitk(dt, left, Mp, right, Mp, dst, cols_in, rows_out, cols_in, Mp, s))

  for (int l = d.nl - 1; l >= 0; --l) {
    const OmLayerWeights& lw = Ls[l];
    const OmLayerGrads& lg = Gs[l];
    const char* x = t.x + t.sx * l;
    const char* qkv = t.qkv + t.sqkv * l;
    const char* ctx = t.ctx + t.sx * l;
    const char* y1 = t.y1 + t.sx * l;
    const char* x1 = t.x1 + t.sx * l;
    const char* f = t.f + t.sf * l;
    const char* y2 = t.y2 + t.sx * l;

    // LN2 backward: dy2 = d(loss)/d(y2)
    RUN(omk_ln_bwd(dt, dx, y2, lw.ln2_g, ws.dy, lg.ln2_g, lg.ln2_b, M, H, c->ln_eps, s));
    // FFN output branch (dropout after the dense, before the residual add)
    const char* dO = ws.dy;
    if (hidden_dropout > 0.f) { RUN(omk_dropout(dt, ws.dy, ws.dd, M * H, hidden_dropout, site_seed(seed, l, 4), s)); dO = ws.dd; }
    RUN(omk_colsum(dt, dO, H, M, H, lg.ffn2_b, s));
    RUN(omk_transpose(dt, dO, H, M, H, ws.tl, Mp, Mp, 0, s));       // dO^T   [H, Mp]
    RUN(omk_transpose(dt, f, F, M, F, ws.tr, Mp, Mp, 1, s));        // gelu(f)^T [F, Mp]
    WGRAD(ws.tl, ws.tr, H, F, lg.ffn2_w);                           // dW2 [H,F]
    RUN(omk_transpose(dt, lw.ffn2_w, F, H, F, ws.wt, H, H, 0, s));  // W2^T [F,H]
    {
      GemmEpilogue e1 = {};
      e1.act = OM_ACT_GELU_ERF_GRAD; e1.resid = f; e1.ldr = F;      // df = (dO W2) * gelu'(f)
      RUN(omk_gemm(dt, dO, H, ws.wt, H, dt, ws.df, F, M, F, H, e1, s));
    }
    RUN(omk_colsum(dt, ws.df, F, M, F, lg.ffn1_b, s));
    RUN(omk_transpose(dt, ws.df, F, M, F, ws.tl, Mp, Mp, 0, s));    // df^T [F, Mp]
    RUN(omk_transpose(dt, x1, H, M, H, ws.tr, Mp, Mp, 0, s));       // x1^T [H, Mp]
    WGRAD(ws.tl, ws.tr, F, H, lg.ffn1_w);                           // dW1 [F,H]
    RUN(omk_transpose(dt, lw.ffn1_w, H, F, H, ws.wt, F, F, 0, s));  // W1^T [H,F]
    {
      GemmEpilogue e2 = {};
      e2.resid = ws.dy; e2.ldr = H;                                 // dx1 = df W1 + dy2 (residual path)
      RUN(omk_gemm(dt, ws.df, F, ws.wt, F, dt, ws.dctx, H, M, H, F, e2, s));
    }
    // LN1 backward (ws.dctx holds d/d(x1) for now)
    RUN(omk_ln_bwd(dt, ws.dctx, y1, lw.ln1_g, ws.dy, lg.ln1_g, lg.ln1_b, M, H, c->ln_eps, s));
    const char* dA = ws.dy;
    if (hidden_dropout > 0.f) { RUN(omk_dropout(dt, ws.dy, ws.dd, M * H, hidden_dropout, site_seed(seed, l, 3), s)); dA = ws.dd; }
    RUN(omk_colsum(dt, dA, H, M, H, lg.o_b, s));
    RUN(omk_transpose(dt, dA, H, M, H, ws.tl, Mp, Mp, 0, s));
    RUN(omk_transpose(dt, ctx, H, M, H, ws.tr, Mp, Mp, 0, s));
    WGRAD(ws.tl, ws.tr, H, H, lg.o_w);                              // dWo [H,H]
    RUN(omk_transpose(dt, lw.o_w, H, H, H, ws.wt, H, H, 0, s));     // Wo^T
    {
      GemmEpilogue e3 = {};
      RUN(omk_gemm(dt, dA, H, ws.wt, H, dt, ws.dctx, H, M, H, H, e3, s));   // dctx = dA Wo
    }
    RUN(omk_attention_bwd(dt, qkv, ws.dctx, ws.dqkv, attention_mask, B, (int)L, H, d.nh, scale,
                          attn_dropout, site_seed(seed, l, 2), s));
    RUN(omk_colsum(dt, ws.dqkv, 3 * H, M, 3 * H, lg.qkv_b, s));
    RUN(omk_transpose(dt, ws.dqkv, 3 * H, M, 3 * H, ws.tl, Mp, Mp, 0, s));
    RUN(omk_transpose(dt, x, H, M, H, ws.tr, Mp, Mp, 0, s));
    WGRAD(ws.tl, ws.tr, 3 * H, H, lg.qkv_w);                        // dWqkv [3H,H]
    RUN(omk_transpose(dt, lw.qkv_w, H, 3 * H, H, ws.wt, 3 * H, 3 * H, 0, s));   // Wqkv^T [H,3H]
    {
      GemmEpilogue e4 = {};
      e4.resid = ws.dy; e4.ldr = H;                                 // dx = dqkv Wqkv + dy1
      RUN(omk_gemm(dt, ws.dqkv, 3 * H, ws.wt, 3 * H, dt, dx_prev, H, M, H, 3 * H, e4, s));
    }
    char* tmp = dx; dx = dx_prev; dx_prev = tmp;
  }
  // ---- embeddings: dropout bwd -> LayerNorm bwd -> scatter into the three tables -------------
  const char* de = dx;
  if (hidden_dropout > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, M * H, hidden_dropout, site_seed(seed, 0, 0), s)); de = ws.dd; }
  RUN(omk_embed_bwd(dt, de, input_ids, token_type_ids, w->word_emb, w->pos_emb, w->type_emb,
                    w->emb_ln_g, g->word_emb, g->pos_emb, g->type_emb, g->emb_ln_g, g->emb_ln_b, M,
                    (int)L, H, c->vocab, c->type_vocab, c->ln_eps, s));
#undef WGRAD
  return 0;
}
