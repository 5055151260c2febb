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

#include <mutex>
#include <unordered_map>
#include <vector>

#include "train_kernels.h"

namespace {
struct Dims {
  int64_t M, Mp, B, L; int H, F, nl, nh, D; size_t es; bool t5, gated;
  bool packed;       // packed rows (round 5): M = the caller's row bound; sequence b lives in rows cu[b] .. cu[b + 1] - 1 of every [M, .] tensor
  // what a 16-bit BERT tape holds (fixed by the FORWARD, remembered per tape: tape_flags below):
  bool bert16;       // 16-bit BERT: the configurations the two flags below apply to
  bool res32;        // the pre-LayerNorm sums y1 / y2 in f32 (the forward's residual stream stays in f32: OM_OPT_TRAIN_RES32)
  bool pre_grad;     // gelu'(f) in place of f (OM_OPT_TRAIN_TAPE_GRAD)
};
constexpr int TAPE_RES32 = 1, TAPE_PRE_GRAD = 2;

Dims dims_of(const OmEncoderConfig* c, int64_t B, int64_t L, int64_t packed_rows = 0) {
  Dims d;
  d.packed = packed_rows > 0;
  d.B = B; d.L = L; d.M = d.packed ? packed_rows : B * L; d.Mp = (d.M + 63) / 64 * 64;
  d.H = c->hidden; d.F = c->ffn; d.nl = c->n_layers; d.nh = c->n_heads;
  d.D = c->head_in > 0 ? c->head_out : c->hidden;
  d.es = (c->dtype == OM_BF16 || c->dtype == OM_F16) ? 2 : 4;
  d.t5 = c->arch == OM_ARCH_T5;
  d.gated = d.t5 && (c->act & 0xff) == OM_ACT_GELU_TANH;      // T5 v1.1: gated gelu_new (wi_0, wi_1)
  d.bert16 = !d.t5 && d.es == 2 && (size_t)d.F * d.es >= (size_t)d.H * 4;
  d.res32 = d.bert16 && om_option(OM_OPT_TRAIN_RES32) != 0;
  d.pre_grad = d.es == 2 && om_option(OM_OPT_TRAIN_TAPE_GRAD) != 0;
  return d;
}
Dims dims_with_flags(Dims d, int flags) { d.res32 = (flags & TAPE_RES32) != 0; d.pre_grad = (flags & TAPE_PRE_GRAD) != 0; return d; }
int flags_of(const Dims& d) { return (d.res32 ? TAPE_RES32 : 0) | (d.pre_grad ? TAPE_PRE_GRAD : 0); }

// ---- tape: activations saved by the forward ------------------------------------------------
struct Tape {
  char* x0pre;             // [M,H] embedding LayerNorm output BEFORE dropout (only if dropout)
  char* x;                 // [(nl+1)][M,H] layer inputs / final output
  char *qkv, *ctx, *y1, *x1, *f, *y2;   // per layer, strided by their per-layer size
  char* f2;                // T5 gated FFN: the gate projection (per layer, stride sf)
  char* g;                 // BERT: gelu(f), the FFN2 input (per layer, stride sf) -- the weight gradient reads it as is
  float *pooled, *headout;             // [B,H], [B,D] (pre-normalise)
  size_t total;
  size_t sx, sqkv, sf;     // per-layer strides in bytes
  size_t sy;               // stride of y1 / y2 (f32 when Dims::res32)
};
Tape carve_tape(const Dims& d, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  Tape t;
  t.sx = align_up((size_t)d.M * d.H * d.es, 256);
  t.sqkv = align_up((size_t)d.M * 3 * d.H * d.es, 256);
  t.sf = align_up((size_t)d.M * d.F * d.es, 256);
  t.sy = d.res32 ? align_up((size_t)d.M * d.H * 4, 256) : t.sx;
  t.x = take(t.sx * (d.nl + 1));
  t.qkv = take(t.sqkv * d.nl);
  t.ctx = take(t.sx * d.nl);
  t.y1 = take(d.t5 ? 0 : t.sy * d.nl);           // BERT: pre-LayerNorm sums; T5 (pre-norm) keeps none
  t.x1 = take(t.sx * d.nl);
  t.f = take(t.sf * d.nl);
  t.y2 = take(d.t5 ? 0 : t.sy * d.nl);
  t.f2 = take(d.gated ? t.sf * d.nl : 0);
  t.g = take(d.t5 ? 0 : t.sf * d.nl);
  t.pooled = (float*)take((size_t)d.B * d.H * 4);
  t.headout = (float*)take((size_t)d.B * d.D * 4);
  t.total = off;
  return t;
}

// ---- scratch of forward (g) and backward -----------------------------------------------------
struct Ws {
  char *g;                          // fwd: GELU output [M,F]
  char *dxa, *dxb, *dy, *dd, *df, *dqkv, *dctx;   // bwd activation gradients
  float* astats;                    // attention backward beyond 256 tokens: (max, 1 / sum, delta) per (sequence, head, query)
  char *tl, *tr, *wt;               // transposed operands (f32 / odd widths only) / every layer's transposed weights
  size_t swt;                       // bytes of one layer's transposed weights
  float *dhead, *dpooled;
  // T5 extras: normed-input scratch, gate gradient, bias [nh,L,L], its LUT and per-offset gradient
  char *nbuf, *df2;
  float *x32a, *x32b;               // res32: the unrounded LayerNorm outputs the residual adds read (layer input / after attention)
  float *posbias, *drel;
  int* lut;
  // deferred weight gradients (bf16 BERT, widths of 256): every layer's dY of the four sites is KEPT until the layer
  // group's batched weight-gradient launch has read it -- per layer [M,H] (FFN2 site) | [M,F] (FFN1) | [M,H] (out-proj)
  // | [M,3H] (QKV): 127 MB per bert-base layer at 9 216 tokens, 1.5 GB per backward of 288 GB
  char* keep;
  size_t skeep;                     // bytes per layer (0: not available for this configuration)
  // BERT: the LayerNorm backward's per-block column sums of d_gamma / d_beta, one area per site (2 per layer), added up by
  // omk_ln_param_reduce when the layer group's gradients are published (round 5: no same-address atomics, a fixed order)
  float* lnpart;
  size_t slnpart;                   // floats per site
  int *kmax, *cu, *cls_rows, *row_map;      // packed rows: per-sequence extents, row offsets (+ total, true total), CLS rows, token of every row
  size_t total;
};
inline bool keep_ok(const Dims& d) { return !d.t5 && d.es == 2 && d.H % 256 == 0 && d.F % 256 == 0 && d.M >= 32; }
Ws carve_ws(const Dims& d, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  Ws w;
  const size_t mh = (size_t)d.M * d.H * d.es, mf = (size_t)d.M * d.F * d.es;
  const size_t wide = (size_t)std::max(3 * d.H, d.F);
  w.g = take(mf);
  w.dxa = take(mh); w.dxb = take(mh); w.dy = take(mh); w.dd = take(mh);
  w.df = take(mf); w.dqkv = take(3 * mh); w.dctx = take(mh);
  w.astats = (float*)take(d.es == 2 ? omk_attention_bwd_long_stats_bytes(d.B, d.nh) : 0);
  w.tl = take(wide * d.Mp * d.es);
  w.tr = take(wide * d.Mp * d.es);
  w.swt = align_up(((size_t)4 * d.H * d.H + (size_t)(d.gated ? 3 : 2) * d.F * d.H) * d.es, 256);
  w.wt = take(w.swt * d.nl);
  w.dhead = (float*)take((size_t)d.B * d.D * 4);
  w.dpooled = (float*)take((size_t)d.B * d.H * 4);
  w.x32a = (float*)take(d.bert16 ? (size_t)d.M * d.H * 4 : 0);      // (reserved whatever the option says now: a tape written
  w.x32b = (float*)take(d.bert16 ? (size_t)d.M * d.H * 4 : 0);      //  under the other setting must find the same workspace)
  w.nbuf = take(d.t5 ? mh : 0);
  w.df2 = take(d.gated ? mf : 0);
  w.posbias = (float*)take(d.t5 ? (size_t)d.nh * d.L * d.L * 4 : 0);
  w.drel = (float*)take(d.t5 ? (size_t)d.nh * (2 * d.L) * 4 : 0);
  w.lut = (int*)take(d.t5 ? (size_t)(2 * d.L) * 4 : 0);
  w.skeep = keep_ok(d) ? align_up(5 * mh + mf, 256) : 0;
  w.keep = take(w.skeep * d.nl);
  w.slnpart = d.t5 ? 0 : (size_t)OM_LNB_MAX_BLOCKS * 2 * d.H;
  w.lnpart = (float*)take(w.slnpart * 4 * 2 * d.nl);
  w.kmax = (int*)take(d.packed ? (size_t)d.B * 4 : 0);
  w.cu = (int*)take(d.packed ? (size_t)(d.B + 2) * 4 : 0);
  w.cls_rows = (int*)take(d.packed ? (size_t)d.B * 4 : 0);
  w.row_map = (int*)take(d.packed ? (size_t)d.M * 4 : 0);
  w.total = off;
  return w;
}

// W^T of every dense weight of the encoder, all layers in one launch: layer l at ws.wt + l * ws.swt holds
// [Wqkv^T (H x 3H) | Wo^T (H x H) | W1^T (H x F) | W2^T (F x H) | gated T5: W1g^T (H x F)]
struct WtView { const char *qkv, *o, *f1, *f2, *f1g; };
WtView wt_of(const Dims& d, const Ws& ws, int l) {
  const char* p = ws.wt + ws.swt * l;
  WtView v;
  v.qkv = p; p += (size_t)3 * d.H * d.H * d.es;
  v.o = p; p += (size_t)d.H * d.H * d.es;
  v.f1 = p; p += (size_t)d.F * d.H * d.es;
  v.f2 = p; p += (size_t)d.F * d.H * d.es;
  v.f1g = p;
  return v;
}
int transpose_weights(int dt, const OmLayerWeights* Ls, const Dims& d, Ws& ws, hipStream_t s) {
  std::vector<const void*> in; std::vector<void*> out; std::vector<int> R, C;
  for (int l = 0; l < d.nl; ++l) {
    const WtView v = wt_of(d, ws, l);
    in.push_back(Ls[l].qkv_w); out.push_back((void*)v.qkv); R.push_back(3 * d.H); C.push_back(d.H);
    in.push_back(Ls[l].o_w); out.push_back((void*)v.o); R.push_back(d.H); C.push_back(d.H);
    in.push_back(Ls[l].ffn1_w); out.push_back((void*)v.f1); R.push_back(d.F); C.push_back(d.H);
    in.push_back(Ls[l].ffn2_w); out.push_back((void*)v.f2); R.push_back(d.H); C.push_back(d.F);
    if (d.gated) { in.push_back(Ls[l].ffn1g_w); out.push_back((void*)v.f1g); R.push_back(d.F); C.push_back(d.H); }
  }
  return omk_transpose_batch(dt, in.data(), out.data(), R.data(), C.data(), (int)in.size(), s);
}

// Weight and bias gradients of one nn.Linear: dW[N,K] += dY^T X, db[N] += column sums of dY (db may be NULL), f32
// atomics into the caller-zeroed buffers.  16-bit runs read dY and X as they lie (gemm_tn.hip: transposing LDS reads,
// the bias sums on the matrix core); f32 runs -- and widths that are not multiples of 128 -- transpose both operands
// and use the NT split-K kernel.
int wgrad(int dt, const void* dY, int N, const void* X, int K, float* dW, float* db, const Dims& d, Ws& ws, hipStream_t s) {
  if (omk_gemm_tn_ok(dt, d.M, N, K, N, K)) return omk_gemm_tn(dt, dY, N, X, K, dW, K, db, d.M, N, K, s);
  if (db && omk_colsum(dt, dY, N, d.M, N, db, s)) return 1;
  if (omk_transpose(dt, dY, N, d.M, N, ws.tl, d.Mp, d.Mp, 0, s)) return 1;
  if (omk_transpose(dt, X, K, d.M, K, ws.tr, d.Mp, d.Mp, 0, s)) return 1;
  return omk_gemm_splitk(dt, ws.tl, d.Mp, ws.tr, d.Mp, dW, K, N, K, d.Mp, s);
}

int check_train_cfg(const OmEncoderConfig* c, int64_t L) {
  if (c->arch != OM_ARCH_BERT && c->arch != OM_ARCH_T5) OM_FAIL("unknown arch");
  if (c->dtype != OM_F32 && c->dtype != OM_BF16 && c->dtype != OM_F16) OM_FAIL("dtype must be OM_F32, OM_BF16 or OM_F16");
  // float16 training (the reference's --fp16 = torch.cuda.amp float16 + GradScaler for every backbone): BERT-family since round 5, T5 stacks
  // since round 6 (ReLU / gated tanh-GELU feed-forwards; as under the reference's autocast nothing clamps -- a checkpoint whose feed-forward
  // activations leave the float16 range overflows there and here alike, the loss scaler skips such steps)
  if (c->dtype == OM_F16 && c->arch != OM_ARCH_BERT && c->arch != OM_ARCH_T5) OM_FAIL("float16 training: BERT-family and T5 encoders");
  if (c->head_dim != 64 || c->n_heads * 64 != c->hidden) OM_FAIL("head_dim must be 64");
  if (L < 1 || L > 512) OM_FAIL("training supports sequence lengths up to 512");      // (257 .. 512: round 6, the tile-at-a-time attention kernels)
  if (c->dtype == OM_F32 && L > 192) OM_FAIL("float32 training supports sequence lengths up to 192 (16-bit formats: 512)");
  if (c->arch == OM_ARCH_BERT && c->act != OM_ACT_GELU_ERF) OM_FAIL("BERT training supports the erf-GELU FFN");
  if (c->arch == OM_ARCH_T5 && (c->act & 0xff) != OM_ACT_RELU && (c->act & 0xff) != OM_ACT_GELU_TANH)
    OM_FAIL("T5 training supports relu and gated gelu_new feed-forward layers");
  const int es = c->dtype == OM_F32 ? 4 : 2;
  if ((c->hidden * es) % 128 || (c->ffn * es) % 128) OM_FAIL("hidden/ffn rows must be multiples of 128 bytes");
  if (c->pooling != OM_POOL_FIRST && c->pooling != OM_POOL_MEAN) OM_FAIL("pooling must be first or mean");
  return 0;
}

inline uint64_t site_seed(uint64_t seed, int layer, int site) {
  return seed + 0x9E3779B97F4A7C15ull * (uint64_t)(16 * layer + site + 1);
}
#define RUN(expr) do { if (expr) return 1; } while (0)

// relative-position bias [nh, L, L] from the bucket table (and the LUT the backward needs again)
int t5_bias_setup(const OmEncoderConfig* c, const OmEncoderWeights* w, const Dims& d, Ws& ws, hipStream_t s) {
  if (!w->rel_bias || !w->final_ln_g) OM_FAIL("T5 needs rel_bias and final_ln_g");
  const int L = (int)d.L;
  const int* lut = nullptr;                      // device-resident, cached per (L, buckets, max distance)
  if (om_t5_lut_device(L, c->rel_buckets, c->rel_max_dist, &lut)) return 1;
  ws.lut = const_cast<int*>(lut);                // the backward reads it again (never written)
  return omk_t5_bias(w->rel_bias, ws.lut, ws.posbias, L, d.nh, s);
}

// T5 encoder stack (pre-RMSNorm residual blocks, HF:models/t5/modeling_t5.py T5Stack / T5Block):
//   x0 = drop(E[ids]);  per layer:  x1 = x + drop(Attn(rms(x)) Wo^T),  x' = x1 + drop(drop(act(rms(x1) Wi^T)) Wo2^T)
//   out = drop(rms(x_last))
int t5_train_forward(const OmEncoderConfig* c, const OmEncoderWeights* w, const int64_t* input_ids,
                     const int64_t* attention_mask, const Dims& d, Tape& t, Ws& ws, float hd, float ad,
                     uint64_t seed, char** final_hidden, hipStream_t s) {
  const int dt = c->dtype, H = d.H, F = d.F;
  const int64_t M = d.M;
  const OmLayerWeights* Ls = w->layers_host;
  RUN(t5_bias_setup(c, w, d, ws, s));
  // packed rows (round 6): the tables are in ws (train_forward_impl); every dropout mask is keyed on the token through row_map
  const int* const cu = d.packed ? ws.cu : nullptr;
  const int* const row_map = d.packed ? ws.row_map : nullptr;
  RUN(omk_embed(dt, input_ids, nullptr, w->word_emb, nullptr, nullptr, nullptr, nullptr, t.x, M, (int)d.L, H, c->vocab, 1, c->ln_eps, 0, s, row_map));
  if (hd > 0.f) RUN(omk_dropout(dt, t.x, t.x, M * H, hd, site_seed(seed, 0, 0), s, row_map, H));
  for (int l = 0; l < d.nl; ++l) {
    const OmLayerWeights& lw = Ls[l];
    char* x = t.x + t.sx * l;
    char* qkv = t.qkv + t.sqkv * l;
    char* ctx = t.ctx + t.sx * l;
    char* x1 = t.x1 + t.sx * l;
    char* f = t.f + t.sf * l;
    RUN(omk_layernorm(dt, x, H, ws.nbuf, H, lw.ln1_g, nullptr, M, H, c->ln_eps, 1, s));
    GemmEpilogue ep = {};
    RUN(omk_gemm(dt, ws.nbuf, H, lw.qkv_w, H, dt, qkv, 3 * H, M, 3 * H, H, ep, s));
    RUN(omk_attention(dt, qkv, ctx, attention_mask, ws.posbias, d.B, (int)d.L, H, d.nh, 1.0f, ad, site_seed(seed, l, 2), s, 0, d.packed ? ws.kmax : nullptr, cu));
    if (d.packed) RUN(omk_zero_rows_from(ctx, (int64_t)H * d.es, ws.cu + d.B, M, s));      // the rows no sequence owns
    ep = GemmEpilogue{};
    ep.resid = x; ep.ldr = H; ep.drop_p = hd; ep.seed = site_seed(seed, l, 3); ep.drop_rows = row_map;
    RUN(omk_gemm(dt, ctx, H, lw.o_w, H, dt, x1, H, M, H, H, ep, s));
    RUN(omk_layernorm(dt, x1, H, ws.nbuf, H, lw.ln2_g, nullptr, M, H, c->ln_eps, 1, s));
    ep = GemmEpilogue{};
    ep.pre_act = f; ep.ldp = F; ep.drop_p = hd; ep.seed = site_seed(seed, l, 5); ep.drop_rows = row_map;
    if (d.gated) {
      if (!lw.ffn1g_w) OM_FAIL("gated T5 feed-forward needs ffn1g_w");
      char* f2 = t.f2 + t.sf * l;
      GemmEpilogue eg = {};
      RUN(omk_gemm(dt, ws.nbuf, H, lw.ffn1g_w, H, dt, f2, F, M, F, H, eg, s));
      ep.act = OM_ACT_GELU_TANH | OM_ACT_MUL_RESID; ep.resid = f2; ep.ldr = F;
    } else {
      ep.act = OM_ACT_RELU;
    }
    RUN(omk_gemm(dt, ws.nbuf, H, lw.ffn1_w, H, dt, ws.g, F, M, F, H, ep, s));
    ep = GemmEpilogue{};
    ep.resid = x1; ep.ldr = H; ep.drop_p = hd; ep.seed = site_seed(seed, l, 4); ep.drop_rows = row_map;
    RUN(omk_gemm(dt, ws.g, F, lw.ffn2_w, F, dt, t.x + t.sx * (l + 1), H, M, H, F, ep, s));
  }
  RUN(omk_layernorm(dt, t.x + t.sx * d.nl, H, ws.dxa, H, w->final_ln_g, nullptr, M, H, c->ln_eps, 1, s));
  if (hd > 0.f) RUN(omk_dropout(dt, ws.dxa, ws.dxa, M * H, hd, site_seed(seed, d.nl, 1), s, row_map, H));
  *final_hidden = ws.dxa;
  return 0;
}

// Progress events of the NEXT backward on this thread (gradient all-reduce overlapped with the backward: the host waits
// for event l on a side stream and reduces layer l's slice of the gradient arena while layers l-1 .. 0 are still being
// differentiated).  events[l], l = n_layers-1 .. 0: recorded on the backward's stream once every kernel writing layer l's
// gradients has been enqueued; events[n_layers]: everything (embeddings, T5 position table) enqueued.  Consumed by one call.
static thread_local void* const* g_bwd_events = nullptr;
static thread_local int g_bwd_nevents = 0;
extern "C" int om_encoder_train_set_layer_events(void* const* events, int n) {
  if (n < 0 || (n > 0 && !events)) OM_FAIL("bad event array");
  g_bwd_events = n > 0 ? events : nullptr;
  g_bwd_nevents = n;
  return 0;
}
// "Consumed by one call" on EVERY exit path: the array is owned by the caller (a ctypes buffer that may be freed right after the
// call), so an early return or a failing launch must not leave the pointer behind for the next backward on this thread.
struct BwdEventsScope { ~BwdEventsScope() { g_bwd_events = nullptr; g_bwd_nevents = 0; } };
// What a tape holds is decided by the run-time options AT THE FORWARD (gelu'(f) in place of f: OM_OPT_TRAIN_TAPE_GRAD; f32
// pre-LayerNorm sums: OM_OPT_TRAIN_RES32) and remembered per tape address: a backward that runs after an option was toggled
// (A/B scripts, GradCache replays over several tapes) reads the tape as it was written, not as the options say now.
// Entries are never erased when a tape is freed (the library does not see frees); past 4 096 entries the OLDEST half goes (by the
// order of the forwards that wrote them), never the whole map: a backward whose forward ran recently -- the A/B-toggle and GradCache
// replay cases the map exists for -- still finds its entry (the round-5 map was cleared whole).
static std::mutex g_tape_mu;
static std::unordered_map<const void*, std::pair<int, uint64_t>> g_tape_flags;      // tape -> (flags, sequence number of the forward)
static uint64_t g_tape_seq = 0;
static void tape_flags_set(const void* tape, int flags) {
  std::lock_guard<std::mutex> lk(g_tape_mu);
  if (g_tape_flags.size() > 4096) {
    const uint64_t keep_from = g_tape_seq > 2048 ? g_tape_seq - 2048 : 0;
    for (auto it = g_tape_flags.begin(); it != g_tape_flags.end();) it = it->second.second < keep_from ? g_tape_flags.erase(it) : std::next(it);
  }
  g_tape_flags[tape] = std::make_pair(flags, ++g_tape_seq);
}
static int tape_flags_get(const void* tape, int fallback) {
  std::lock_guard<std::mutex> lk(g_tape_mu);
  auto it = g_tape_flags.find(tape);
  return it == g_tape_flags.end() ? fallback : it->second.first;
}
static int record_layer_event(int l, hipStream_t s) {
  if (g_bwd_events && l < g_bwd_nevents && g_bwd_events[l]) OM_HIP(hipEventRecord((hipEvent_t)g_bwd_events[l], s));
  return 0;
}

// Weight-gradient lane of the BERT backward (OM_OPT_TRAIN_WGRAD_STREAM).  A layer's four weight-gradient GEMMs read
// the same dY as the data-gradient GEMM that follows each of them and feed nothing inside the backward: at the training
// batch (9 216 token rows) neither kind of launch fills 256 CUs on its own (108 - 432 tiles), so they run side by side
// -- weight gradients on a second stream of this thread, ordered against the main stream by events:
//   ready[l][i]  main -> side : dY (and the zeroed gradient arena) of weight gradient i of layer l is complete
//   done[l][i]   side -> main : weight gradient i of layer l has finished reading its dY (the buffer may be rewritten)
//   mark[l]      main -> side : every main-stream kernel of layer l is enqueued (the layer's progress event is then
//                               recorded on the side stream, after both)
// One event per (layer, site): no event is re-recorded while a wait on it may be pending.  Created on first use per
// thread and device and kept (a training process owns one device for its lifetime).
struct WgradLane {
  int device = -1;
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ready, done, mark;
  // Early weight transposes (round 5, OM_OPT_TRAIN_WGRAD_STREAM bit 1, OFF by default: measured 102.4-103.0 steps/s with them against
  // 105.4-105.8 without on one box, profiles/r05_train_ab_v2_*.jsonl): the backward's data gradients need every W^T (170 us of
  // memory-bound transposes at the head of each backward, profiles/r05_train_timeline_v1.txt).  They depend on the weights only,
  // so the training FORWARD launches them on this side stream, under its own contractions; the backward waits for wt_done instead
  // of transposing -- when the workspace and the weights are still the ones the transposes were made from.
  hipEvent_t wt_start = nullptr, wt_done = nullptr;
  const void *wt_ws = nullptr /* the ws.wt they were written to */, *wt_w0 = nullptr, *wt_w1 = nullptr;
  int wt_nl = 0;
};
static thread_local WgradLane g_lane;
static int lane_get(int n_layers, WgradLane** out) {
  int dev = 0;
  OM_HIP(hipGetDevice(&dev));
  if (g_lane.device != dev) {
    g_lane = WgradLane{};
    OM_HIP(hipStreamCreateWithFlags(&g_lane.side, hipStreamNonBlocking));
    g_lane.device = dev;
  }
  auto grow = [](std::vector<hipEvent_t>& v, size_t n) -> int {
    while (v.size() < n) {
      hipEvent_t e = nullptr;
      OM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      v.push_back(e);
    }
    return 0;
  };
  RUN(grow(g_lane.ready, (size_t)n_layers * 4));
  RUN(grow(g_lane.done, (size_t)n_layers * 4));
  RUN(grow(g_lane.mark, (size_t)n_layers + 1));
  if (!g_lane.wt_start) {
    OM_HIP(hipEventCreateWithFlags(&g_lane.wt_start, hipEventDisableTiming));
    OM_HIP(hipEventCreateWithFlags(&g_lane.wt_done, hipEventDisableTiming));
  }
  *out = &g_lane;
  return 0;
}

// dx: gradient w.r.t. the stack's output (after the final dropout); returns through the grads struct
int t5_train_backward(const OmEncoderConfig* c, const OmEncoderWeights* w, const int64_t* input_ids,
                      const int64_t* attention_mask, const Dims& d, const Tape& t, Ws& ws, float hd, float ad,
                      uint64_t seed, char* dx, char* dx_other, const OmEncoderGrads* g, hipStream_t s) {
  const int dt = c->dtype, H = d.H, F = d.F;
  const int64_t M = d.M, Mp = d.Mp;
  const OmLayerWeights* Ls = w->layers_host;
  const OmLayerGrads* Gs = g->layers_host;
  if (!g->final_ln_g || !g->rel_bias) OM_FAIL("T5 gradients need final_ln_g and rel_bias buffers");
  RUN(t5_bias_setup(c, w, d, ws, s));
  OM_HIP(hipMemsetAsync(ws.drel, 0, (size_t)d.nh * (2 * d.L) * 4, s));
#define WGRAD(dY_, N_, X_, K_, dW_) RUN(wgrad(dt, dY_, N_, X_, K_, dW_, nullptr, d, ws, s))
  RUN(transpose_weights(dt, Ls, d, ws, s));
  // final dropout + RMSNorm
  const int* const cu = d.packed ? ws.cu : nullptr;                 // packed rows (round 6): the tables train_backward_impl rebuilt
  const int* const row_map = d.packed ? ws.row_map : nullptr;
  if (hd > 0.f) RUN(omk_dropout(dt, dx, dx, M * H, hd, site_seed(seed, d.nl, 1), s, row_map, H));
  RUN(omk_norm_bwd(dt, dx, t.x + t.sx * d.nl, w->final_ln_g, dx_other, g->final_ln_g, nullptr, M, H, c->ln_eps, 1, nullptr, s));
  { char* tmp = dx; dx = dx_other; dx_other = tmp; }
  const int kind = d.gated ? 1 : 0;
  for (int l = d.nl - 1; l >= 0; --l) {
    const OmLayerWeights& lw = Ls[l];
    const OmLayerGrads& lg = Gs[l];
    const char* x = t.x + t.sx * l;
    const char* qkv = t.qkv + t.sqkv * l;
    const char* ctx = t.ctx + t.sx * l;
    const char* x1 = t.x1 + t.sx * l;
    const char* f = t.f + t.sf * l;
    const char* f2 = d.gated ? t.f2 + t.sf * l : nullptr;
    const WtView wt = wt_of(d, ws, l);
    // ---- feed-forward branch
    const char* dO = dx;
    if (hd > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, M * H, hd, site_seed(seed, l, 4), s, row_map, H)); dO = ws.dd; }
    RUN(omk_t5_act_fwd(dt, f, f2, ws.g, M * F, kind, s));                     // g = act(f) [* f2]
    if (hd > 0.f) RUN(omk_dropout(dt, ws.g, ws.g, M * F, hd, site_seed(seed, l, 5), s, row_map, F));
    WGRAD(dO, H, ws.g, F, lg.ffn2_w);                                         // dWo2 [H,F]
    GemmEpilogue e = {};
    RUN(omk_gemm(dt, dO, H, wt.f2, H, dt, ws.df, F, M, F, H, e, s));          // dg = dO Wo2;  Wo2^T [F,H]
    if (hd > 0.f) RUN(omk_dropout(dt, ws.df, ws.df, M * F, hd, site_seed(seed, l, 5), s, row_map, F));
    RUN(omk_t5_act_bwd(dt, ws.df, f, f2, ws.df, ws.df2, M * F, kind, s));     // df (in place), df2
    RUN(omk_layernorm(dt, x1, H, ws.nbuf, H, lw.ln2_g, nullptr, M, H, c->ln_eps, 1, s));   // n2 again
    WGRAD(ws.df, F, ws.nbuf, H, lg.ffn1_w);                                   // dWi (wi / wi_0) [F,H]
    if (d.gated) {
      if (!lg.ffn1g_w) OM_FAIL("gated T5 gradients need ffn1g_w");
      WGRAD(ws.df2, F, ws.nbuf, H, lg.ffn1g_w);                               // dWi_1 [F,H]
    }
    e = GemmEpilogue{};
    RUN(omk_gemm(dt, ws.df, F, wt.f1, F, dt, ws.dy, H, M, H, F, e, s));       // dn2 = df Wi;  Wi^T [H,F]
    if (d.gated) {
      e = GemmEpilogue{};
      e.resid = ws.dy; e.ldr = H;
      RUN(omk_gemm(dt, ws.df2, F, wt.f1g, F, dt, ws.dy, H, M, H, F, e, s));   // += df2 Wi_1
    }
    RUN(omk_norm_bwd(dt, ws.dy, x1, lw.ln2_g, dx_other, lg.ln2_g, nullptr, M, H, c->ln_eps, 1, dx, s));   // dx1
    // ---- attention branch (dx_other now holds d/d x1)
    const char* dA = dx_other;
    if (hd > 0.f) { RUN(omk_dropout(dt, dx_other, ws.dd, M * H, hd, site_seed(seed, l, 3), s, row_map, H)); dA = ws.dd; }
    WGRAD(dA, H, ctx, H, lg.o_w);
    e = GemmEpilogue{};
    RUN(omk_gemm(dt, dA, H, wt.o, H, dt, ws.dctx, H, M, H, H, e, s));         // dctx = dA Wo
    // (round 6) beyond 256 tokens -- and from 193 on: the generic kernel keeps a whole score row in registers, which is fine up to six key tiles
    // (it wins by 4-7 % of a step at 144 ... 192 tokens) and 20 % of a step slower with eight (profiles/r06_train_long_sequences.txt) --
    if (d.L > 256 || (!d.packed && dt != OM_F32 && ((om_option(OM_OPT_ATTENTION_FAST) & 2) || ((om_option(OM_OPT_ATTENTION_FAST) & 1) && d.L > 192))))      // one score tile in registers at a time, delta from the tape's attention output
      RUN(omk_attention_bwd_long(dt, qkv, ctx, ws.dctx, ws.dqkv, attention_mask, d.B, (int)d.L, H, d.nh, 1.0f, ad,
                                 site_seed(seed, l, 2), ws.posbias, ws.drel, ws.astats, s));
    else
    RUN(omk_attention_bwd_bias(dt, qkv, ws.dctx, ws.dqkv, attention_mask, d.B, (int)d.L, H, d.nh, 1.0f, ad,
                               site_seed(seed, l, 2), ws.posbias, ws.drel, s, cu));
    if (d.packed) RUN(omk_zero_rows_from(ws.dqkv, (int64_t)3 * H * d.es, ws.cu + d.B, M, s));      // rows the kernel does not own: zero, not stale
    RUN(omk_layernorm(dt, x, H, ws.nbuf, H, lw.ln1_g, nullptr, M, H, c->ln_eps, 1, s));    // n1 again
    WGRAD(ws.dqkv, 3 * H, ws.nbuf, H, lg.qkv_w);
    e = GemmEpilogue{};
    RUN(omk_gemm(dt, ws.dqkv, 3 * H, wt.qkv, 3 * H, dt, ws.dy, H, M, H, 3 * H, e, s));  // dn1
    RUN(omk_norm_bwd(dt, ws.dy, x, lw.ln1_g, dx, lg.ln1_g, nullptr, M, H, c->ln_eps, 1, dx_other, s));   // dx of layer input
    RUN(record_layer_event(l, s));
  }
#undef WGRAD
  const char* de = dx;
  if (hd > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, M * H, hd, site_seed(seed, 0, 0), s, row_map, H)); de = ws.dd; }
  RUN(omk_t5_embed_bwd(dt, de, input_ids, g->word_emb, M, H, c->vocab, s, row_map));
  RUN(omk_t5_bias_bwd(ws.drel, ws.lut, g->rel_bias, (int)d.L, d.nh, s));
  RUN(record_layer_event(d.nl, s));
  g_bwd_events = nullptr; g_bwd_nevents = 0;
  return 0;
}
#undef RUN
}  // namespace

extern "C" size_t om_encoder_tape_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L) {
  if (!cfg || B <= 0 || L <= 0) return 0;
  return carve_tape(dims_of(cfg, B, L), nullptr).total;
}
extern "C" size_t om_encoder_train_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L) {
  if (!cfg || B <= 0 || L <= 0) return 0;
  return carve_ws(dims_of(cfg, B, L), nullptr).total;
}
// Packed rows in training (round 5): the contractions, normalisations and the tape run over `packed_rows` rows -- the tokens up to
// each sequence's last unmasked one, back to back -- instead of B * L (the reference pads every sequence of a batch to one length,
// dataset/data_collator.py:13-24, and computes over the padding).  16-bit BERT-family and (round 6) T5 configurations with widths of 256, L <= 256,
// packed_rows a multiple of 256 that is >= the token count (a bound that is too small turns the representations into NaN).
extern "C" int om_encoder_train_packed_supported(const OmEncoderConfig* c, int64_t B, int64_t L, int64_t packed_rows) {
  if (!c || B <= 0 || L <= 0 || packed_rows <= 0) return 0;
  if ((c->arch != OM_ARCH_BERT && c->arch != OM_ARCH_T5) || (c->dtype != OM_BF16 && c->dtype != OM_F16) || c->n_layers <= 0) return 0;      // (T5: round 6)
  if (packed_rows % 256 || packed_rows < 512 || packed_rows > B * L + 255 || packed_rows >= B * L) return 0;
  if (c->hidden % 256 || c->ffn % 256 || c->n_heads * 64 != c->hidden) return 0;
  if (c->pooling != OM_POOL_FIRST && c->pooling != OM_POOL_MEAN) return 0;
  if (!om_option(OM_OPT_ATTENTION_FAST) || L > 256) return 0;
  if (c->arch == OM_ARCH_BERT && (size_t)c->ffn < (size_t)2 * c->hidden) return 0;      // (the f32 pooled tail borrows the [M, F] scratch)
  return 1;
}
extern "C" size_t om_encoder_tape_bytes_packed(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows) {
  if (!om_encoder_train_packed_supported(cfg, B, L, packed_rows)) return 0;
  return carve_tape(dims_of(cfg, B, L, packed_rows), nullptr).total;
}
extern "C" size_t om_encoder_train_workspace_bytes_packed(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows) {
  if (!om_encoder_train_packed_supported(cfg, B, L, packed_rows)) return 0;
  return carve_ws(dims_of(cfg, B, L, packed_rows), nullptr).total;
}

#define RUN(expr) do { if (expr) return 1; } while (0)

// out_hidden != NULL: the stack's output [B,L,H] in the compute dtype (T5: after the final RMSNorm and its dropout) is
// copied out and the pooling / head / normalise tail is skipped (the T5 decoder position consumes it, decoder.hip)
static int train_forward_impl(const OmEncoderConfig* c, const OmEncoderWeights* w,
                              const int64_t* input_ids, const int64_t* attention_mask,
                              const int64_t* token_type_ids, int64_t B, int64_t L,
                              float hidden_dropout, float attn_dropout, uint64_t seed,
                              void* tape_mem, size_t tape_bytes, float* out_reps, void* out_hidden,
                              void* workspace, size_t workspace_bytes, void* stream, int64_t packed_rows = 0) {
  if (!c || !w || !input_ids || !attention_mask || !tape_mem || !workspace || (!out_reps && !out_hidden)) OM_FAIL("null argument");
  if (check_train_cfg(c, L)) return 1;
  if (B <= 0) return 0;
  if (((uintptr_t)tape_mem & 255) || ((uintptr_t)workspace & 255)) OM_FAIL("tape/workspace must be 256-byte aligned");
  if (packed_rows > 0 && (out_hidden || !om_encoder_train_packed_supported(c, B, L, packed_rows)))
    OM_FAIL("packed rows in training: 16-bit BERT-family or T5 encoder, widths of 256, L <= 256, rows a multiple of 256 in [512, B * L) (om_encoder_train_packed_supported)");
  const Dims d = dims_of(c, B, L, packed_rows);
  const bool packed = d.packed;
  Tape t = carve_tape(d, (char*)tape_mem);
  Ws ws = carve_ws(d, (char*)workspace);
  if (t.total > tape_bytes || ws.total > workspace_bytes) OM_FAIL("tape or workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int dt = c->dtype, H = d.H, F = d.F;
  const int64_t M = d.M;
  const OmLayerWeights* Ls = w->layers_host;
  if (!Ls) OM_FAIL("layers_host is null");
  if (d.t5) {
    char* xf_t5 = nullptr;
    if (packed) {      // (round 6) the row tables of the BERT branch below; a T5 block has no bias anywhere, so the rows no sequence owns stay exact zeros
      RUN(omk_mask_extent(attention_mask, B, (int)L, ws.kmax, s));
      RUN(omk_pack_rows(ws.kmax, B, (int)L, M, ws.cu, ws.cls_rows, ws.row_map, s));
    }
    if (t5_train_forward(c, w, input_ids, attention_mask, d, t, ws, hidden_dropout, attn_dropout, seed, &xf_t5, s)) return 1;
    if (out_hidden) {
      OM_HIP(hipMemcpyAsync(out_hidden, xf_t5, (size_t)M * H * d.es, hipMemcpyDeviceToDevice, s));
      return 0;
    }
    const bool head_t5 = c->head_in > 0 && w->head_w;
    RUN(omk_pool(dt, xf_t5, attention_mask, t.pooled, B, (int)L, H, c->pooling, s, packed ? ws.cu : nullptr));
    float* pre_t5 = c->normalize ? t.headout : out_reps;
    if (head_t5) {
      if (om_gemm_nt(OM_F32, t.pooled, H, w->head_w, c->head_in, OM_F32, pre_t5, d.D, B, d.D, c->head_in,
                     nullptr, nullptr, 0, OM_ACT_NONE, s)) return 1;
    } else {
      OM_HIP(hipMemcpyAsync(pre_t5, t.pooled, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    }
    if (c->normalize) RUN(omk_l2norm(t.headout, out_reps, B, d.D, s));
    if (packed) RUN(omk_pack_overflow_poison(ws.cu, B, M, out_reps, B * (int64_t)d.D, s));
    return 0;
  }
  if (L > c->max_pos) OM_FAIL("sequence longer than the position table");

  tape_flags_set(tape_mem, flags_of(d));
  if ((om_option(OM_OPT_TRAIN_WGRAD_STREAM) & 2) && d.es == 2 && d.nl > 0) {
    WgradLane* lane = nullptr;
    RUN(lane_get(d.nl, &lane));
    OM_HIP(hipEventRecord(lane->wt_start, s));                       // everything queued so far (the optimizer's update of the
    OM_HIP(hipStreamWaitEvent(lane->side, lane->wt_start, 0));       // weights, the last reader of ws.wt) comes first
    lane->wt_ws = nullptr;
    RUN(transpose_weights(dt, Ls, d, ws, lane->side));
    OM_HIP(hipEventRecord(lane->wt_done, lane->side));
    lane->wt_ws = ws.wt; lane->wt_w0 = Ls[0].qkv_w; lane->wt_w1 = Ls[d.nl - 1].ffn2_w; lane->wt_nl = d.nl;
  }
  const int* cu = nullptr;
  const int* row_map = nullptr;
  if (packed) {
    // rows of sequence b: cu[b] .. cu[b + 1] - 1 (up to its last unmasked token); row_map names the token of every row (-1: a zero
    // row behind the last sequence).  The pad rows stay finite through the forward (a zero embedding row, normalised, projected ...)
    // except where no kernel writes them -- the attention output -- which is zeroed: the weight gradients sum over all M rows, and
    // 0 (their dY) x anything finite is 0.
    RUN(omk_mask_extent(attention_mask, B, (int)L, ws.kmax, s));
    RUN(omk_pack_rows(ws.kmax, B, (int)L, M, ws.cu, ws.cls_rows, ws.row_map, s));
    cu = ws.cu; row_map = ws.row_map;
  }
  RUN(omk_embed(dt, input_ids, token_type_ids, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g,
                w->emb_ln_b, t.x, M, (int)L, H, c->vocab, c->type_vocab, c->ln_eps, 1, s, row_map));
  // (every hidden-dropout mask is keyed on (token, column): with packed rows the kernels get the row -> token map, so the packed and the
  // padded step of one batch draw the same masks -- VERDICT r5 item 5; the attention mask was keyed on (sequence, head, query, key) already)
  if (hidden_dropout > 0.f) RUN(omk_dropout(dt, t.x, t.x, M * H, hidden_dropout, site_seed(seed, 0, 0), s, row_map, H));
  // res32 (16-bit BERT, default): the residual stream of the FORWARD stays in f32, as the reference's autocast keeps it (layer_norm
  // runs and returns fp32; the residual add of a 16-bit dense output and an fp32 LayerNorm output is fp32).  The pre-LayerNorm sums
  // y1 / y2 are f32 on the tape; every LayerNorm writes its output twice -- in the compute format (the next contraction's operand,
  // and the weight gradient's) and unrounded into x32a / x32b (what the next residual add reads).  On tests/golden/train_base.npz the
  // 16-bit residual stream was the reason the step sat 2.4 x further from the reference's fp32 gradients than the reference's own
  // encoder-only bf16 autocast; with it in f32: 1.0 x (tools/emulate_train_dataflow.py).  The gradient stream stays 16-bit.
  float* xres = nullptr;        // f32 copy of the current layer input
  if (d.res32) {
    RUN(omk_embed(OM_F32, input_ids, token_type_ids, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g,
                  w->emb_ln_b, ws.x32a, M, (int)L, H, c->vocab, c->type_vocab, c->ln_eps, 1, s, row_map));
    if (hidden_dropout > 0.f) RUN(omk_dropout(OM_F32, ws.x32a, ws.x32a, M * H, hidden_dropout, site_seed(seed, 0, 0), s, row_map, H));
    xres = ws.x32a;
  }
  const int ydt = d.res32 ? OM_F32 : dt;      // format of y1 / y2 and of the residual operands
  const float scale = 1.0f / sqrtf((float)c->head_dim);
  for (int l = 0; l < d.nl; ++l) {
    const OmLayerWeights& lw = Ls[l];
    char* x = t.x + t.sx * l;
    char* qkv = t.qkv + t.sqkv * l;
    char* ctx = t.ctx + t.sx * l;
    char* y1 = t.y1 + t.sy * l;
    char* x1 = t.x1 + t.sx * l;
    char* f = t.f + t.sf * l;
    char* y2 = t.y2 + t.sy * l;
    GemmEpilogue ep = {};
    ep.bias = lw.qkv_b;
    RUN(omk_gemm(dt, x, H, lw.qkv_w, H, dt, qkv, 3 * H, M, 3 * H, H, ep, s));
    RUN(omk_attention(dt, qkv, ctx, attention_mask, nullptr, B, (int)L, H, d.nh, scale, attn_dropout,
                      site_seed(seed, l, 2), s, 0, packed ? ws.kmax : nullptr, cu));
    if (packed) RUN(omk_zero_rows_from(ctx, (int64_t)H * d.es, ws.cu + B, M, s));      // the rows no sequence owns
    ep = GemmEpilogue{};
    ep.bias = lw.o_b; ep.resid = d.res32 ? (const void*)xres : (const void*)x; ep.ldr = H; ep.drop_p = hidden_dropout; ep.seed = site_seed(seed, l, 3);
    ep.drop_rows = row_map;
    RUN(omk_gemm(dt, ctx, H, lw.o_w, H, ydt, y1, H, M, H, H, ep, s));
    if (d.res32) RUN(omk_layernorm_dual(dt, (const float*)y1, H, x1, ws.x32b, H, lw.ln1_g, lw.ln1_b, M, H, c->ln_eps, s));
    else RUN(omk_layernorm(dt, y1, H, x1, H, lw.ln1_g, lw.ln1_b, M, H, c->ln_eps, 0, s));
    ep = GemmEpilogue{};
    char* gl = t.g + t.sf * l;
    // 16-bit runs keep gelu'(f) on the tape instead of f (the forward has Phi(f) in hand; the backward multiplies): OM_ACT_PRE_GRAD
    ep.bias = lw.ffn1_b; ep.act = OM_ACT_GELU_ERF | (d.pre_grad ? OM_ACT_PRE_GRAD : 0); ep.pre_act = f; ep.ldp = F;
    RUN(omk_gemm(dt, x1, H, lw.ffn1_w, H, dt, gl, F, M, F, H, ep, s));
    ep = GemmEpilogue{};
    ep.bias = lw.ffn2_b; ep.resid = d.res32 ? (const void*)ws.x32b : (const void*)x1; ep.ldr = H; ep.drop_p = hidden_dropout; ep.seed = site_seed(seed, l, 4);
    ep.drop_rows = row_map;
    RUN(omk_gemm(dt, gl, F, lw.ffn2_w, F, ydt, y2, H, M, H, F, ep, s));
    if (d.res32) RUN(omk_layernorm_dual(dt, (const float*)y2, H, t.x + t.sx * (l + 1), ws.x32a, H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, s));
    else RUN(omk_layernorm(dt, y2, H, t.x + t.sx * (l + 1), H, lw.ln2_g, lw.ln2_b, M, H, c->ln_eps, 0, s));
  }
  const char* xf = t.x + t.sx * d.nl;
  if (out_hidden) {
    OM_HIP(hipMemcpyAsync(out_hidden, xf, (size_t)M * H * d.es, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  const bool head = c->head_in > 0 && w->head_w;
  if (d.es == 2 && d.nl > 0 && (c->pooling == OM_POOL_FIRST || d.bert16)) {
    // 16-bit runs: the pooled rows come from an f32 evaluation of the LAST LayerNorm (its input y2 is on the tape) -- the reference's
    // autocast runs layer_norm in fp32, so its representations are not rounded to 16 bits on the way to the loss, and the loss of a
    // contrastive batch lives in the DIFFERENCES between near-equal dot products (tests/golden/train_base.npz: the bf16 rounding
    // of the final hidden state alone moved the gradients by tens of percent on a random-init model).  The backward is unchanged:
    // the same function of y2; t.x[n_layers] stays on the tape in the compute format for callers that want the hidden state.
    const OmLayerWeights& last = Ls[d.nl - 1];
    const char* y2_last = t.y2 + t.sy * (d.nl - 1);
    const int ydt_last = d.res32 ? OM_F32 : dt;
    if (c->pooling == OM_POOL_FIRST) {
      if (packed) RUN(omk_layernorm_f32out(ydt_last, y2_last, H, t.pooled, H, last.ln2_g, last.ln2_b, B, H, c->ln_eps, 0, s, nullptr, ws.cls_rows));
      else RUN(omk_layernorm_f32out(ydt_last, y2_last, L * H, t.pooled, H, last.ln2_g, last.ln2_b, B, H, c->ln_eps, 0, s));
    } else {
      float* x32 = (float*)ws.df;                        // [M, H] f32 fits the [M, F] 16-bit scratch of the backward (bert16: F >= 2 H)
      RUN(omk_layernorm_f32out(ydt_last, y2_last, H, x32, H, last.ln2_g, last.ln2_b, M, H, c->ln_eps, 0, s));
      RUN(omk_pool(OM_F32, x32, attention_mask, t.pooled, B, (int)L, H, c->pooling, s, cu));
    }
  } else {
    if (packed) OM_FAIL("packed rows in training: the 16-bit pooled tail only");
    RUN(omk_pool(dt, xf, attention_mask, t.pooled, B, (int)L, H, c->pooling, s));
  }
  float* pre = c->normalize ? t.headout : out_reps;      // value before F.normalize
  if (head) {
    if (om_gemm_nt(OM_F32, t.pooled, H, w->head_w, c->head_in, OM_F32, pre, d.D, B, d.D, c->head_in,
                   nullptr, nullptr, 0, OM_ACT_NONE, s)) return 1;
  } else {
    OM_HIP(hipMemcpyAsync(pre, t.pooled, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
  }
  if (c->normalize) RUN(omk_l2norm(t.headout, out_reps, B, d.D, s));
  if (packed) RUN(omk_pack_overflow_poison(ws.cu, B, M, out_reps, B * (int64_t)d.D, s));      // a row bound below the token count: NaN, never a truncated batch
  return 0;
}

extern "C" int om_encoder_train_forward_packed(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                               const int64_t* input_ids, const int64_t* attention_mask,
                                               const int64_t* token_type_ids, int64_t B, int64_t L, int64_t packed_rows,
                                               float hidden_dropout, float attn_dropout, uint64_t seed,
                                               void* tape_mem, size_t tape_bytes, float* out_reps,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!out_reps || packed_rows <= 0) OM_FAIL("null argument / packed_rows must be positive");
  return train_forward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, hidden_dropout, attn_dropout, seed,
                            tape_mem, tape_bytes, out_reps, nullptr, workspace, workspace_bytes, stream, packed_rows);
}

extern "C" int om_encoder_train_forward(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                        const int64_t* input_ids, const int64_t* attention_mask,
                                        const int64_t* token_type_ids, int64_t B, int64_t L,
                                        float hidden_dropout, float attn_dropout, uint64_t seed,
                                        void* tape_mem, size_t tape_bytes, float* out_reps,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!out_reps) OM_FAIL("null argument");
  return train_forward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, hidden_dropout, attn_dropout, seed,
                            tape_mem, tape_bytes, out_reps, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int om_encoder_train_forward_hidden(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                               const int64_t* input_ids, const int64_t* attention_mask,
                                               const int64_t* token_type_ids, int64_t B, int64_t L,
                                               float hidden_dropout, float attn_dropout, uint64_t seed,
                                               void* tape_mem, size_t tape_bytes, void* out_hidden,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!out_hidden) OM_FAIL("null argument");
  return train_forward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, hidden_dropout, attn_dropout, seed,
                            tape_mem, tape_bytes, nullptr, out_hidden, workspace, workspace_bytes, stream);
}

// d_hidden != NULL: the gradient w.r.t. the stack's output [B,L,H] (compute dtype) replaces the tail's backward
static int train_backward_impl(const OmEncoderConfig* c, const OmEncoderWeights* w,
                               const int64_t* input_ids, const int64_t* attention_mask,
                               const int64_t* token_type_ids, int64_t B, int64_t L,
                               float hidden_dropout, float attn_dropout, uint64_t seed,
                               const void* tape_mem, const float* d_reps, const void* d_hidden,
                               const OmEncoderGrads* g, void* workspace,
                               size_t workspace_bytes, void* stream, int64_t packed_rows = 0) {
  BwdEventsScope events_scope;
  if (!c || !w || !g || !tape_mem || (!d_reps && !d_hidden) || !workspace) OM_FAIL("null argument");
  if (check_train_cfg(c, L)) return 1;
  if (B <= 0) return 0;
  if (packed_rows > 0 && (d_hidden || !om_encoder_train_packed_supported(c, B, L, packed_rows)))
    OM_FAIL("packed rows in training: not for this configuration (om_encoder_train_packed_supported)");
  const Dims d0 = dims_of(c, B, L, packed_rows);
  const Dims d = dims_with_flags(d0, tape_flags_get(tape_mem, flags_of(d0)));      // the tape as its forward wrote it
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

  const int* cu = nullptr;
  if (d.packed) {            // the forward's row tables again (the workspace is not the tape: nothing of it survives between the calls)
    RUN(omk_mask_extent(attention_mask, B, (int)L, ws.kmax, s));
    RUN(omk_pack_rows(ws.kmax, B, (int)L, M, ws.cu, ws.cls_rows, ws.row_map, s));
    cu = ws.cu;
  }
  char* dx = ws.dxa;       // gradient w.r.t. the current layer's OUTPUT
  char* dx_prev = ws.dxb;  // gradient w.r.t. its input (next iteration's dx)
  const float* dpool32 = nullptr;      // the same for the top layer as f32 (16-bit BERT, pooled tail)
  if (d_hidden) {
    OM_HIP(hipMemcpyAsync(dx, d_hidden, (size_t)M * H * d.es, hipMemcpyDeviceToDevice, s));
  } else {
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
    // 16-bit BERT: the gradient of the pooled rows enters the last LayerNorm's backward in f32 (ws.df is free until the first
    // GELU' contraction writes it).  That backward projects out most of a contrastive gradient -- the pooled vectors of a batch are
    // nearly parallel -- and what survives is of the size of a bf16 rounding of what went in: rounding here alone moved the
    // step from 2.4 x to 3.9 x (worst tensor 8.9 x) the reference's encoder-only bf16 autocast (tools/emulate_train_dataflow.py).
    if (d.bert16) {
      dpool32 = (const float*)ws.df;
      RUN(omk_pool_bwd(OM_F32, dpooled, attention_mask, ws.df, B, (int)L, H, c->pooling, s, cu));
      // packed rows: the rows behind the last sequence carry a ZERO gradient from here down (LayerNorm backward, dropout and the
      // contractions map zero rows to zero rows), so they add nothing to the weight gradients that sum over all M rows
      if (d.packed) RUN(omk_zero_rows_from(ws.df, (int64_t)H * 4, ws.cu + B, M, s));
    } else {
      if (d.packed && !d.t5) OM_FAIL("packed rows in training: the 16-bit pooled tail only");
      RUN(omk_pool_bwd(dt, dpooled, attention_mask, dx, B, (int)L, H, c->pooling, s, cu));
      if (d.packed) RUN(omk_zero_rows_from(dx, (int64_t)H * d.es, ws.cu + B, M, s));      // zero gradient rows stay zero all the way down (no bias in a T5 block)
    }
  }
  if (d.t5)
    return t5_train_backward(c, w, input_ids, attention_mask, d, t, ws, hidden_dropout, attn_dropout, seed, dx,
                             dx_prev, g, s);

  // weight gradients on the second stream when every one of them takes the direct kernel (16-bit, widths of 128)
  WgradLane* lane = nullptr;
  if (om_option(OM_OPT_TRAIN_WGRAD_STREAM) && omk_gemm_tn_ok(dt, M, H, F, H, F) && omk_gemm_tn_ok(dt, M, F, H, F, H) &&
      omk_gemm_tn_ok(dt, M, H, H, H, H) && omk_gemm_tn_ok(dt, M, 3 * H, H, 3 * H, H))
    RUN(lane_get(d.nl, &lane));
  // Deferred, batched weight gradients (OM_OPT_TRAIN_WGRAD_BATCH = layers per launch): the sites only RECORD their
  // contraction; the dY they name live in the layer's keep slice (never rewritten inside this backward), and after the
  // lowest layer of a group one om_gemm_tn_acc_batch launch computes the group's 4 x layers weight gradients -- on the
  // side stream when the lane is on (one event per group each way instead of one per site), else in line.
  int wbatch = om_option(OM_OPT_TRAIN_WGRAD_BATCH);
  if (wbatch < 0) wbatch = 0;
  if (wbatch > 12) wbatch = 12;
  if (wbatch && !(ws.skeep && omk_gemm_tn_batch_ok(dt, M, H, F, H, F) && omk_gemm_tn_batch_ok(dt, M, F, H, F, H) &&
                  omk_gemm_tn_batch_ok(dt, M, H, H, H, H) && omk_gemm_tn_batch_ok(dt, M, 3 * H, H, 3 * H, H)))
    wbatch = 0;
  std::vector<OmTnProblem> pend;
  const bool ln_atomics = (om_option(OM_OPT_TRAIN_WGRAD_STREAM) & 4) != 0;      // A/B: the LayerNorm parameter sums by atomics, as before round 5
  std::vector<OmLnSite> ln_pend;                                    // LayerNorm sites whose parameter sums wait for the group's reduce
  int group_top = d.nl - 1;                                         // highest layer of the group being collected
  const size_t mh_b = (size_t)M * H * d.es, mf_b = (size_t)M * F * d.es;
#define WGRAD(I_, dY_, N_, X_, K_, dW_, db_)                                                        \
  do {                                                                                             \
    if (wbatch) {                                                                                  \
      OmTnProblem q_;                                                                              \
      q_.A = dY_; q_.B = X_; q_.C = dW_; q_.bias = db_; q_.lda = N_; q_.ldb = K_; q_.ldc = K_; q_.N = N_; q_.K = K_; \
      pend.push_back(q_);                                                                          \
    } else if (lane) {                                                                             \
      OM_HIP(hipEventRecord(lane->ready[l * 4 + (I_)], s));                                        \
      OM_HIP(hipStreamWaitEvent(lane->side, lane->ready[l * 4 + (I_)], 0));                        \
      RUN(wgrad(dt, dY_, N_, X_, K_, dW_, db_, d, ws, lane->side));                                \
      OM_HIP(hipEventRecord(lane->done[l * 4 + (I_)], lane->side));                                \
    } else {                                                                                       \
      RUN(wgrad(dt, dY_, N_, X_, K_, dW_, db_, d, ws, s));                                         \
    }                                                                                              \
  } while (0)
  // before the main stream rewrites a buffer that weight gradient I_ of layer L_ reads
#define WGRAD_DONE(L_, I_) do { if (lane && !wbatch && (L_) < d.nl) OM_HIP(hipStreamWaitEvent(s, lane->done[(L_) * 4 + (I_)], 0)); } while (0)
  if (g_lane.wt_ws == ws.wt && g_lane.wt_w0 == Ls[0].qkv_w && g_lane.wt_w1 == Ls[d.nl - 1].ffn2_w && g_lane.wt_nl == d.nl &&
      g_lane.wt_done && (om_option(OM_OPT_TRAIN_WGRAD_STREAM) & 2)) {
    OM_HIP(hipStreamWaitEvent(s, g_lane.wt_done, 0));               // the forward's side-stream transposes of these very weights
  } else {
    // (a forward's transposes of OTHER weights / another shape may still be writing an overlapping part of the workspace)
    if (g_lane.wt_done && g_lane.wt_ws) OM_HIP(hipStreamWaitEvent(s, g_lane.wt_done, 0));
    RUN(transpose_weights(dt, Ls, d, ws, s));
    g_lane.wt_ws = nullptr;                                          // ws.wt now holds what THIS call made: not the forward's
  }

  for (int l = d.nl - 1; l >= 0; --l) {
    const OmLayerWeights& lw = Ls[l];
    const OmLayerGrads& lg = Gs[l];
    const char* x = t.x + t.sx * l;
    const char* qkv = t.qkv + t.sqkv * l;
    const char* ctx = t.ctx + t.sx * l;
    const char* y1 = t.y1 + t.sy * l;
    const char* x1 = t.x1 + t.sx * l;
    const char* f = t.f + t.sf * l;
    const char* gl = t.g + t.sf * l;
    const char* y2 = t.y2 + t.sy * l;
    const WtView wt = wt_of(d, ws, l);
    // where this layer's dY go: shared scratch, or (deferred weight gradients) the layer's keep slice.  The operand of
    // the FFN2 / out-proj sites is the dropout-masked gradient when there is dropout, else the LayerNorm backward's output.
    char* const kp = wbatch ? ws.keep + ws.skeep * l : nullptr;
    char* const dy2 = (wbatch && hidden_dropout <= 0.f) ? kp : ws.dy;
    char* const dd2 = (wbatch && hidden_dropout > 0.f) ? kp : ws.dd;
    char* const dfl = wbatch ? kp + mh_b : ws.df;
    char* const dy1 = (wbatch && hidden_dropout <= 0.f) ? kp + mh_b + mf_b : ws.dy;
    char* const dd1 = (wbatch && hidden_dropout > 0.f) ? kp + mh_b + mf_b : ws.dd;
    char* const dqkvl = wbatch ? kp + 2 * mh_b + mf_b : ws.dqkv;

    // LN2 backward: dy2 = d(loss)/d(y2)
    // (+ the FFN output branch's dropout, which sits after the dense and before the residual add, in the same pass)
    WGRAD_DONE(l + 1, 2);                                           // ws.dy / ws.dd: last read by dWo of the layer above
    {
      OmLnSite st = {ln_atomics ? nullptr : ws.lnpart + ws.slnpart * (2 * l + 1), lg.ln2_g, lg.ln2_b, 0};
      RUN(omk_ln_bwd_drop(dt, dx, y2, lw.ln2_g, dy2, dd2, hidden_dropout, site_seed(seed, l, 4), lg.ln2_g, lg.ln2_b, M, H, c->ln_eps, s,
                          l == d.nl - 1 ? dpool32 : nullptr, d.res32 ? (const float*)y2 : nullptr, (float*)st.partial, &st.blocks,
                          d.packed ? ws.row_map : nullptr));
      if (st.partial) ln_pend.push_back(st);
    }
    const char* dO = hidden_dropout > 0.f ? dd2 : dy2;
    WGRAD(0, dO, H, gl, F, lg.ffn2_w, lg.ffn2_b);                   // dW2 [H,F], db2
    {
      GemmEpilogue e1 = {};
      // df = (dO W2) * gelu'(f);  W2^T [F,H].  The tape holds gelu'(f) itself in 16-bit runs (see the forward)
      e1.act = d.pre_grad ? OM_ACT_MUL_RESID : OM_ACT_GELU_ERF_GRAD; e1.resid = f; e1.ldr = F;
      WGRAD_DONE(l + 1, 1);                                         // ws.df: last read by dW1 of the layer above
      RUN(omk_gemm(dt, dO, H, wt.f2, H, dt, dfl, F, M, F, H, e1, s));
    }
    WGRAD(1, dfl, F, x1, H, lg.ffn1_w, lg.ffn1_b);                  // dW1 [F,H], db1
    {
      GemmEpilogue e2 = {};
      e2.resid = dy2; e2.ldr = H;                                   // dx1 = df W1 + dy2 (residual path);  W1^T [H,F]
      RUN(omk_gemm(dt, dfl, F, wt.f1, F, dt, ws.dctx, H, M, H, F, e2, s));
    }
    // LN1 backward (ws.dctx holds d/d(x1) for now)
    WGRAD_DONE(l, 0);                                               // ws.dy / ws.dd: read by dW2 of this layer
    {
      OmLnSite st = {ln_atomics ? nullptr : ws.lnpart + ws.slnpart * (2 * l), lg.ln1_g, lg.ln1_b, 0};
      RUN(omk_ln_bwd_drop(dt, ws.dctx, y1, lw.ln1_g, dy1, dd1, hidden_dropout, site_seed(seed, l, 3), lg.ln1_g, lg.ln1_b, M, H, c->ln_eps, s,
                          nullptr, d.res32 ? (const float*)y1 : nullptr, (float*)st.partial, &st.blocks, d.packed ? ws.row_map : nullptr));
      if (st.partial) ln_pend.push_back(st);
    }
    const char* dA = hidden_dropout > 0.f ? dd1 : dy1;
    WGRAD(2, dA, H, ctx, H, lg.o_w, lg.o_b);                        // dWo [H,H], dbo
    {
      GemmEpilogue e3 = {};
      RUN(omk_gemm(dt, dA, H, wt.o, H, dt, ws.dctx, H, M, H, H, e3, s));    // dctx = dA Wo
    }
    WGRAD_DONE(l + 1, 3);                                           // ws.dqkv: last read by dWqkv of the layer above
    // (round 6) beyond 256 tokens -- and from 193 on (see t5_train_backward) --
    if (L > 256 || (!d.packed && dt != OM_F32 && ((om_option(OM_OPT_ATTENTION_FAST) & 2) || ((om_option(OM_OPT_ATTENTION_FAST) & 1) && L > 192))))      // one score tile in registers at a time, delta from the tape's attention output
      RUN(omk_attention_bwd_long(dt, qkv, ctx, ws.dctx, dqkvl, attention_mask, B, (int)L, H, d.nh, scale,
                                 attn_dropout, site_seed(seed, l, 2), nullptr, nullptr, ws.astats, s));
    else
    RUN(omk_attention_bwd(dt, qkv, ws.dctx, dqkvl, attention_mask, B, (int)L, H, d.nh, scale,
                          attn_dropout, site_seed(seed, l, 2), s, cu));
    if (d.packed) RUN(omk_zero_rows_from(dqkvl, (int64_t)3 * H * d.es, ws.cu + B, M, s));      // rows the kernel does not own: zero, not stale
    WGRAD(3, dqkvl, 3 * H, x, H, lg.qkv_w, lg.qkv_b);               // dWqkv [3H,H], dbqkv
    {
      GemmEpilogue e4 = {};
      e4.resid = dy1; e4.ldr = H;                                   // dx = dqkv Wqkv + dy1;  Wqkv^T [H,3H]
      RUN(omk_gemm(dt, dqkvl, 3 * H, wt.qkv, 3 * H, dt, dx_prev, H, M, H, 3 * H, e4, s));
    }
    char* tmp = dx; dx = dx_prev; dx_prev = tmp;
    if (wbatch) {
      // the group is complete at its lowest layer: one launch for all of its weight gradients, then the progress events
      // of its layers (their gradients exist once that launch is done)
      if (l == 0 || group_top - l + 1 >= wbatch) {
        hipStream_t ws_stream = s;
        if (lane) {
          OM_HIP(hipEventRecord(lane->mark[l], s));
          OM_HIP(hipStreamWaitEvent(lane->side, lane->mark[l], 0));
          ws_stream = lane->side;
        }
        RUN(omk_gemm_tn_batch(dt, pend.data(), (int)pend.size(), M, ws_stream));
        pend.clear();
        RUN(omk_ln_param_reduce(ln_pend.data(), (int)ln_pend.size(), H, ws_stream));      // the group's LayerNorm parameter gradients
        ln_pend.clear();
        for (int ll = group_top; ll >= l; --ll) RUN(record_layer_event(ll, ws_stream));
        if (lane && l == 0) OM_HIP(hipEventRecord(lane->done[3], lane->side));       // the join below waits for it
        group_top = l - 1;
      }
    } else if (lane && g_bwd_events) {                              // the layer's gradients are complete when BOTH streams got here
      RUN(omk_ln_param_reduce(ln_pend.data(), (int)ln_pend.size(), H, s));
      ln_pend.clear();
      OM_HIP(hipEventRecord(lane->mark[l], s));
      OM_HIP(hipStreamWaitEvent(lane->side, lane->mark[l], 0));
      RUN(record_layer_event(l, lane->side));
    } else {
      RUN(omk_ln_param_reduce(ln_pend.data(), (int)ln_pend.size(), H, s));
      ln_pend.clear();
      RUN(record_layer_event(l, s));
    }
  }
  if (!ln_pend.empty()) OM_FAIL("internal: LayerNorm parameter sums left unreduced");
  if (lane) OM_HIP(hipStreamWaitEvent(s, lane->done[3], 0));        // join: the side stream is in order, layer 0's dWqkv is its last launch
  // ---- embeddings: dropout bwd -> LayerNorm bwd -> scatter into the three tables -------------
  const char* de = dx;
  if (hidden_dropout > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, M * H, hidden_dropout, site_seed(seed, 0, 0), s, d.packed ? ws.row_map : nullptr, H)); de = ws.dd; }
  RUN(omk_embed_bwd(dt, de, input_ids, token_type_ids, w->word_emb, w->pos_emb, w->type_emb,
                    w->emb_ln_g, g->word_emb, g->pos_emb, g->type_emb, g->emb_ln_g, g->emb_ln_b, B * L,
                    (int)L, H, c->vocab, c->type_vocab, c->ln_eps, s, cu));
#undef WGRAD
#undef WGRAD_DONE
  RUN(record_layer_event(d.nl, s));
  g_bwd_events = nullptr; g_bwd_nevents = 0;
  return 0;
}

extern "C" int om_encoder_train_backward(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                         const int64_t* input_ids, const int64_t* attention_mask,
                                         const int64_t* token_type_ids, int64_t B, int64_t L,
                                         float hidden_dropout, float attn_dropout, uint64_t seed,
                                         const void* tape_mem, const float* d_reps,
                                         const OmEncoderGrads* g, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!d_reps) OM_FAIL("null argument");
  return train_backward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, hidden_dropout, attn_dropout, seed,
                             tape_mem, d_reps, nullptr, g, workspace, workspace_bytes, stream);
}
extern "C" int om_encoder_train_backward_packed(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                                const int64_t* input_ids, const int64_t* attention_mask,
                                                const int64_t* token_type_ids, int64_t B, int64_t L, int64_t packed_rows,
                                                float hidden_dropout, float attn_dropout, uint64_t seed,
                                                const void* tape_mem, const float* d_reps,
                                                const OmEncoderGrads* g, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  if (!d_reps || packed_rows <= 0) OM_FAIL("null argument / packed_rows must be positive");
  return train_backward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, hidden_dropout, attn_dropout, seed,
                             tape_mem, d_reps, nullptr, g, workspace, workspace_bytes, stream, packed_rows);
}
extern "C" int om_encoder_train_backward_hidden(const OmEncoderConfig* c, const OmEncoderWeights* w,
                                                const int64_t* input_ids, const int64_t* attention_mask,
                                                const int64_t* token_type_ids, int64_t B, int64_t L,
                                                float hidden_dropout, float attn_dropout, uint64_t seed,
                                                const void* tape_mem, const void* d_hidden,
                                                const OmEncoderGrads* g, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  if (!d_hidden) OM_FAIL("null argument");
  return train_backward_impl(c, w, input_ids, attention_mask, token_type_ids, B, L, hidden_dropout, attn_dropout, seed,
                             tape_mem, nullptr, d_hidden, g, workspace, workspace_bytes, stream);
}
