// om_t5_decoder_step: ONE decoder position of a T5 encoder-decoder over the encoder's output -- what the reference runs
// when a T5 backbone is not `--encoder_only`:
//   DRModel.encode      modeling/dense_retrieval_model.py:137-141   decoder_input_ids = zeros([B,1]); reps = decoder hidden[:, 0]
//   RRModel.encode      modeling/reranking_model.py:110-114         logits[:, 0, [neg_token, pos_token]] of T5ForConditionalGeneration
//                       (+ log_softmax(...)[:, 1] in retriever/reranker.py:114-115)
// HF:models/t5/modeling_t5.py T5Stack (decoder) with a single query token:
//   x = shared[decoder_start]                                                  (no scaling, dropout off in eval)
//   per layer:  x += Wo Wv n            n = RMSNorm(x): self-attention over ONE position -- softmax over a single key is
//                                       1 whatever the score and its relative-position bias, so q, k are never needed
//               x += Wo_c ctx           q = Wq_c RMSNorm(x);  K | V = E (Wk_c | Wv_c)^T over the encoder output E [B,L,H];
//                                       p = softmax_l(q_h . K_h[l] + mask)  (T5: no 1/sqrt(d), no bias in cross attention)
//               x += Wo act(Wi n)       relu, or gelu_new(Wi_0 n) * (Wi_1 n) for v1.1
//   out = RMSNorm_final(x)  [B,H] f32
// The only token-count-sized work is the K | V projection (one [B L, 2H] GEMM per layer on the encoder's GEMM kernels,
// +17 % of an encoder forward) and the single-query attention over it; everything else is [B, H] sized.
#include <vector>

#include "attn_common.h"
#include "train_kernels.h"

namespace {
#define RUN(expr) do { if (expr) return 1; } while (0)

// x[b, :] = emb[:]  (f32 table row -> compute dtype)
template <typename T>
__global__ void dec_start_kernel(const float* __restrict__ emb, T* __restrict__ x, int64_t B, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * H) ElemOps<T>::store(x + i, emb[i % H]);
}

#define DEC_MAX_L 1024
// Single-query cross attention.  Block = (batch b, head h), 256 threads: thread l scores keys l, l + 256, ... (L <= 1 024), block softmax,
// then 64 threads x 4 row groups accumulate ctx[d] = sum_l p[l] V[l][d].  kv: [B, L, 2H] (K | V), q, ctx: [B, H].
// Training: drop_p > 0 zeroes probabilities by the hash of (seed, (b heads + h) L + l) -- HF drops attention weights after
// the softmax (modeling_t5.py T5Attention); the backward regenerates the same mask.
template <typename T>
__global__ __launch_bounds__(256) void dec_cross_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                        const int64_t* __restrict__ mask, T* __restrict__ ctx, int L,
                                                        int H, int heads, float drop_p, uint64_t seed) {
  constexpr int NJ = DEC_MAX_L / 256;          // keys per thread: l = tid + 256 j (round 6: up to 1 024 encoder positions; was 256)
  __shared__ float sq[64];
  __shared__ float sp[DEC_MAX_L];
  __shared__ float red[8];
  __shared__ float part[4][64];
  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int tid = threadIdx.x;
  if (tid < 64) sq[tid] = ElemOps<T>::load(q + b * H + h * 64 + tid);
  __syncthreads();
  float sc[NJ];
  float mxl = -INFINITY;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int l = tid + 256 * j;
    sc[j] = -INFINITY;
    if (l < L) {
      const T* kr = kv + (b * L + l) * 2 * (int64_t)H + h * 64;
      float a = 0.f;
#pragma unroll 8
      for (int d = 0; d < 64; ++d) a = fmaf(sq[d], ElemOps<T>::load(kr + d), a);
      sc[j] = a + (mask[b * L + l] != 0 ? 0.f : -3.4028235e38f);        // HF: (1 - mask) * finfo.min
    }
    mxl = fmaxf(mxl, sc[j]);
  }
  float mx = wave_max(mxl);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const DropCfg dc(drop_p);
  float suml = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int l = tid + 256 * j;
    const float e = l < L ? expf(sc[j] - mx) : 0.f;
    suml += e;
    sp[l] = (dc.thresh && l < L && !dropout_keep(seed, (uint64_t)blockIdx.x * (uint64_t)L + (uint64_t)l, dc.thresh)) ? 0.f : e * (dc.thresh ? dc.keep_scale : 1.f);
  }
  float sum = wave_sum(suml);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  const int d = tid & 63, grp = tid >> 6;
  float acc = 0.f;
  for (int l = grp; l < L; l += 4) acc = fmaf(sp[l], ElemOps<T>::load(kv + (b * L + l) * 2 * (int64_t)H + H + h * 64 + d), acc);
  part[grp][d] = acc;
  __syncthreads();
  if (tid < 64) ElemOps<T>::store(ctx + b * H + h * 64 + tid, ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) * inv);
}

// out[b, :] = RMSNorm(x[b, :]) * g   in f32
template <typename T>
__global__ __launch_bounds__(256) void dec_final_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                        float* __restrict__ out, int H, float eps) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  float ss = 0.f;
  for (int c = threadIdx.x; c < H; c += 256) { const float v = ElemOps<T>::load(x + b * H + c); ss += v * v; }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)H + eps);
  for (int c = threadIdx.x; c < H; c += 256) out[b * H + c] = ElemOps<T>::load(x + b * H + c) * rstd * g[c];
}

struct DecWs { char *x, *n, *t, *ctx, *ff, *ff2, *kv; size_t total; };
DecWs carve(const OmEncoderConfig* c, int64_t B, int64_t L, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  const size_t es = (c->dtype == OM_BF16 || c->dtype == OM_F16) ? 2 : 4;
  DecWs w;
  w.x = take((size_t)B * c->hidden * es); w.n = take((size_t)B * c->hidden * es);
  w.t = take((size_t)B * c->hidden * es); w.ctx = take((size_t)B * c->hidden * es);
  w.ff = take((size_t)B * c->ffn * es); w.ff2 = take((size_t)B * c->ffn * es);
  w.kv = take((size_t)B * L * 2 * c->hidden * es);
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t om_t5_decoder_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L) {
  if (!cfg || B <= 0 || L <= 0) return 0;
  return carve(cfg, B, L, nullptr).total;
}

extern "C" int om_t5_decoder_step(const OmEncoderConfig* c, const OmT5DecoderWeights* w, const void* enc_hidden,
                                  const int64_t* attention_mask, int64_t B, int64_t L, float* out_hidden, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!c || !w || !enc_hidden || !attention_mask || !out_hidden) OM_FAIL("null argument");
  if (B <= 0) return 0;
  if (c->arch != OM_ARCH_T5) OM_FAIL("decoder step: T5 only");
  if (c->dtype != OM_F32 && c->dtype != OM_BF16 && c->dtype != OM_F16) OM_FAIL("dtype must be OM_F32, OM_BF16 or OM_F16");
  if (c->head_dim != 64 || c->n_heads * 64 != c->hidden) OM_FAIL("head_dim must be 64 (inner dim == d_model)");
  if (L < 1 || L > DEC_MAX_L) OM_FAIL("sequence length must be in [1,1024]");
  if (!w->layers_host || w->n_layers < 1 || !w->start_emb || !w->final_ln_g) OM_FAIL("incomplete decoder weights");
  const int akind = c->act & 0xff;
  if (akind != OM_ACT_RELU && akind != OM_ACT_GELU_TANH) OM_FAIL("T5 decoder supports relu and gated gelu_new feed-forward layers");
  if (!workspace || ((uintptr_t)workspace & 255)) OM_FAIL("workspace must be 256-byte aligned");
  DecWs ws = carve(c, B, L, (char*)workspace);
  if (ws.total > workspace_bytes) OM_FAIL("workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads;
  const int64_t M = B * L;
#define GEMM(A_, lda_, W_, ldw_, C_, ldc_, M_, N_, K_, res_, ldr_, act_)                            \
  do {                                                                                              \
    if (om_gemm_nt(dt, A_, lda_, W_, ldw_, dt, C_, ldc_, M_, N_, K_, nullptr, res_, ldr_, act_, s)) return 1; \
  } while (0)
  if (dt == OM_BF16) hipLaunchKernelGGL((dec_start_kernel<bf16_t>), dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, w->start_emb, (bf16_t*)ws.x, B, H);
  else if (dt == OM_F16) hipLaunchKernelGGL((dec_start_kernel<f16_t>), dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, w->start_emb, (f16_t*)ws.x, B, H);
  else hipLaunchKernelGGL((dec_start_kernel<float>), dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, w->start_emb, (float*)ws.x, B, H);
  OM_LAUNCH_CHECK();
  for (int l = 0; l < w->n_layers; ++l) {
    const OmT5DecoderLayer& lw = w->layers_host[l];
    if (!lw.sa_v_w || !lw.sa_o_w || !lw.sa_ln_g || !lw.ca_q_w || !lw.ca_kv_w || !lw.ca_o_w || !lw.ca_ln_g || !lw.ffn1_w ||
        !lw.ffn2_w || !lw.ffn_ln_g) OM_FAIL("incomplete decoder layer weights");
    // self-attention over the single position: x += Wo (Wv n)
    RUN(omk_layernorm(dt, ws.x, H, ws.n, H, lw.sa_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    GEMM(ws.n, H, lw.sa_v_w, H, ws.t, H, B, H, H, nullptr, 0, OM_ACT_NONE);
    GEMM(ws.t, H, lw.sa_o_w, H, ws.x, H, B, H, H, ws.x, H, OM_ACT_NONE);
    // cross-attention: q from the decoder token, K | V from the encoder output
    RUN(omk_layernorm(dt, ws.x, H, ws.n, H, lw.ca_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    GEMM(ws.n, H, lw.ca_q_w, H, ws.t, H, B, H, H, nullptr, 0, OM_ACT_NONE);
    GEMM(enc_hidden, H, lw.ca_kv_w, H, ws.kv, 2 * H, M, 2 * H, H, nullptr, 0, OM_ACT_NONE);
    if (dt == OM_BF16) hipLaunchKernelGGL((dec_cross_kernel<bf16_t>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const bf16_t*)ws.t, (const bf16_t*)ws.kv, attention_mask, (bf16_t*)ws.ctx, (int)L, H, nh, 0.f, 0ull);
    else if (dt == OM_F16) hipLaunchKernelGGL((dec_cross_kernel<f16_t>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const f16_t*)ws.t, (const f16_t*)ws.kv, attention_mask, (f16_t*)ws.ctx, (int)L, H, nh, 0.f, 0ull);
    else hipLaunchKernelGGL((dec_cross_kernel<float>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const float*)ws.t, (const float*)ws.kv, attention_mask, (float*)ws.ctx, (int)L, H, nh, 0.f, 0ull);
    OM_LAUNCH_CHECK();
    GEMM(ws.ctx, H, lw.ca_o_w, H, ws.x, H, B, H, H, ws.x, H, OM_ACT_NONE);
    // feed-forward
    RUN(omk_layernorm(dt, ws.x, H, ws.n, H, lw.ffn_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    if (lw.ffn1g_w) {
      GEMM(ws.n, H, lw.ffn1g_w, H, ws.ff2, F, B, F, H, nullptr, 0, OM_ACT_NONE);
      GEMM(ws.n, H, lw.ffn1_w, H, ws.ff, F, B, F, H, ws.ff2, F, c->act | OM_ACT_MUL_RESID);
    } else {
      GEMM(ws.n, H, lw.ffn1_w, H, ws.ff, F, B, F, H, nullptr, 0, c->act);
    }
    GEMM(ws.ff, F, lw.ffn2_w, F, ws.x, H, B, H, F, ws.x, H, OM_ACT_NONE);
  }
  if (dt == OM_BF16) hipLaunchKernelGGL((dec_final_kernel<bf16_t>), dim3((unsigned)B), dim3(256), 0, s, (const bf16_t*)ws.x, w->final_ln_g, out_hidden, H, c->ln_eps);
  else if (dt == OM_F16) hipLaunchKernelGGL((dec_final_kernel<f16_t>), dim3((unsigned)B), dim3(256), 0, s, (const f16_t*)ws.x, w->final_ln_g, out_hidden, H, c->ln_eps);
  else hipLaunchKernelGGL((dec_final_kernel<float>), dim3((unsigned)B), dim3(256), 0, s, (const float*)ws.x, w->final_ln_g, out_hidden, H, c->ln_eps);
  OM_LAUNCH_CHECK();
#undef GEMM
  return 0;
}

// =====================================================================================================================
// Training through the decoder position (round 3): what autograd does under the reference's
//   DRModel.encode  (modeling/dense_retrieval_model.py:137-141, T5 backbones that are not --encoder_only)
//   RRModel.encode  (modeling/reranking_model.py:110-114, monoT5)
// in train mode -- HF T5Stack(decoder) with dropout_rate at its sites: the start embedding, the self-attention weight of
// the single key (one draw per (row, head): the softmax over one key is 1, and HF drops it like any other attention
// weight), the cross-attention probabilities, the three branch outputs before their residual adds, the feed-forward
// inner activation, and the output of the final RMSNorm.  Masks are hashes of (seed, site, element index), regenerated in
// the backward.  The forward keeps a tape (per layer: the three residual-stream states, v, q, K | V, ctx, the
// feed-forward pre-activations); the backward ADDS weight gradients into caller-zeroed f32 buffers and writes the
// gradient w.r.t. the encoder output, which om_encoder_train_backward_hidden takes from there.
// =====================================================================================================================
namespace {
// dropout sites of layer l: 1 self-attention weight, 2 cross-attention probabilities, 3 cross output, 4 feed-forward output,
// 5 feed-forward inner, 6 self-attention output; (0, 0) the start embedding, (n_layers, 1) the final output
inline uint64_t dec_site(uint64_t seed, int layer, int site) {
  return seed + 0x9E3779B97F4A7C15ull * (uint64_t)(16 * layer + site + 1) + 0x5DEECE66Dull;
}

// v[b, h*64 .. h*64+63] *= keep(seed, b heads + h) / (1 - p): the dropped single-key attention weight (forward and backward)
template <typename T>
__global__ void dec_head_drop_kernel(T* __restrict__ v, int64_t B, int H, float p, uint64_t seed) {
  const DropCfg dc(p);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int64_t b = i / H;
  const int h = (int)(i % H) >> 6;
  const bool keep = dropout_keep(seed, (uint64_t)(b * (H >> 6) + h), dc.thresh);
  ElemOps<T>::store(v + i, keep ? ElemOps<T>::load(v + i) * dc.keep_scale : 0.f);
}
template <typename T>
__global__ void dec_cast_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    ElemOps<T>::store(y + i, x[i]);
}

// Backward of the single-query cross attention.  Block = (b, h), thread l <-> key l:
//   p = softmax(q . K + mask),  pd = dropout(p),  ctx = pd V        (recomputed)
//   dV[l] = pd_l dctx,  dpd_l = dctx . V[l],  dp = dropout'(dpd),  ds_l = p_l (dp_l - sum_j p_j dp_j)
//   dK[l] = ds_l q,  dq = sum_l ds_l K[l]
// dkv rows of masked keys come out as exact zeros (p_l = 0); every row l < L of the block's sequence is written.
template <typename T>
__global__ __launch_bounds__(256) void dec_cross_bwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                            const int64_t* __restrict__ mask, const T* __restrict__ dctx,
                                                            T* __restrict__ dq, T* __restrict__ dkv, int L, int H, int heads,
                                                            float drop_p, uint64_t seed) {
  constexpr int NJ = DEC_MAX_L / 256;          // keys per thread: l = tid + 256 j
  __shared__ float sq[64], sd[64];
  __shared__ float sp[DEC_MAX_L];
  __shared__ float red[12];
  __shared__ float part[4][64];
  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int tid = threadIdx.x;
  if (tid < 64) { sq[tid] = ElemOps<T>::load(q + b * H + h * 64 + tid); sd[tid] = ElemOps<T>::load(dctx + b * H + h * 64 + tid); }
  __syncthreads();
  float sc[NJ], dpd[NJ];
  float mxl = -INFINITY;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int l = tid + 256 * j;
    sc[j] = -INFINITY; dpd[j] = 0.f;
    if (l < L) {
      const T* kr = kv + (b * L + l) * 2 * (int64_t)H + h * 64;
      float a = 0.f, dd = 0.f;
#pragma unroll 8
      for (int d = 0; d < 64; ++d) { a = fmaf(sq[d], ElemOps<T>::load(kr + d), a); dd = fmaf(sd[d], ElemOps<T>::load(kr + H + d), dd); }
      sc[j] = a + (mask[b * L + l] != 0 ? 0.f : -3.4028235e38f);
      dpd[j] = dd;
    }
    mxl = fmaxf(mxl, sc[j]);
  }
  float mx = wave_max(mxl);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float e[NJ];
  float suml = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) { e[j] = (tid + 256 * j) < L ? expf(sc[j] - mx) : 0.f; suml += e[j]; }
  float sum = wave_sum(suml);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float invs = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  const DropCfg dc(drop_p);
  const float ks = dc.thresh ? dc.keep_scale : 1.f;
  float pl[NJ], pd[NJ], dp[NJ];
  float dotl = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int l = tid + 256 * j;
    pl[j] = e[j] * invs;
    const bool keep = !dc.thresh || (l < L && dropout_keep(seed, (uint64_t)blockIdx.x * (uint64_t)L + (uint64_t)l, dc.thresh));
    pd[j] = keep ? pl[j] * ks : 0.f;
    dp[j] = keep ? dpd[j] * ks : 0.f;
    dotl += pl[j] * dp[j];
  }
  float dot = wave_sum(dotl);
  if ((tid & 63) == 0) red[8 + (tid >> 6)] = dot;
  __syncthreads();
  dot = (red[8] + red[9]) + (red[10] + red[11]);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int l = tid + 256 * j;
    const float ds = pl[j] * (dp[j] - dot);
    sp[l] = l < L ? ds : 0.f;
    if (l < L) {
      T* dr = dkv + (b * L + l) * 2 * (int64_t)H + h * 64;
#pragma unroll 8
      for (int d = 0; d < 64; ++d) { ElemOps<T>::store(dr + d, ds * sq[d]); ElemOps<T>::store(dr + H + d, pd[j] * sd[d]); }
    }
  }
  __syncthreads();
  const int d = tid & 63, grp = tid >> 6;
  float acc = 0.f;
  for (int l = grp; l < L; l += 4) acc = fmaf(sp[l], ElemOps<T>::load(kv + (b * L + l) * 2 * (int64_t)H + h * 64 + d), acc);
  part[grp][d] = acc;
  __syncthreads();
  if (tid < 64) ElemOps<T>::store(dq + b * H + h * 64 + tid, (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
}

struct DecTape {
  char* base; size_t sl;                 // per-layer block stride
  size_t o_x0, o_v, o_x1, o_q, o_ctx, o_x2, o_f, o_f2, o_kv;
  char* xl;                              // the stack's output before the final RMSNorm
  size_t total;
  char* at(int l, size_t off) const { return base + sl * l + off; }
};
DecTape dec_tape(const OmEncoderConfig* c, int nl, int64_t B, int64_t L, char* base) {
  const size_t es = (c->dtype == OM_BF16 || c->dtype == OM_F16) ? 2 : 4;
  const size_t bh = align_up((size_t)B * c->hidden * es, 256), bf = align_up((size_t)B * c->ffn * es, 256);
  DecTape t;
  t.base = base;
  size_t off = 0;
  t.o_x0 = off; off += bh; t.o_v = off; off += bh; t.o_x1 = off; off += bh; t.o_q = off; off += bh;
  t.o_ctx = off; off += bh; t.o_x2 = off; off += bh; t.o_f = off; off += bf; t.o_f2 = off; off += bf;
  t.o_kv = off; off += align_up((size_t)B * L * 2 * c->hidden * es, 256);
  t.sl = off;
  t.xl = base + t.sl * nl;
  t.total = t.sl * nl + bh;
  return t;
}
struct DecTrainWs {
  char *n, *g, *dg, *dg2, *dd, *dxa, *dxb, *dn, *dq, *dctx, *dkv, *tl, *tr, *wt;
  float *dE, *do32;
  size_t swt, total;
};
DecTrainWs dec_train_ws(const OmEncoderConfig* c, int nl, int64_t B, int64_t L, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  const size_t es = (c->dtype == OM_BF16 || c->dtype == OM_F16) ? 2 : 4;
  const size_t H = c->hidden, F = c->ffn, M = (size_t)B * L, Mp = (M + 63) / 64 * 64;
  const size_t bh = (size_t)B * H * es, bf = (size_t)B * F * es;
  DecTrainWs w;
  w.n = take(bh); w.g = take(bf); w.dg = take(bf); w.dg2 = take(bf); w.dd = take(bh);
  w.dxa = take(bh); w.dxb = take(bh); w.dn = take(bh); w.dq = take(bh); w.dctx = take(bh);
  w.dkv = take(M * 2 * H * es);
  const size_t wide = std::max(2 * H, F);
  w.tl = take(wide * Mp * es); w.tr = take(wide * Mp * es);
  w.swt = align_up((6 * H * H + 3 * F * H) * es, 256);
  w.wt = take(w.swt * nl);
  w.dE = (float*)take(M * H * 4);
  w.do32 = (float*)take((size_t)B * H * 4);
  w.total = off;
  return w;
}
// transposed weights of decoder layer l (the data-gradient GEMMs' B operands)
struct DecWt { char *sa_v, *sa_o, *ca_q, *ca_kv, *ca_o, *f1, *f1g, *f2; };
DecWt dec_wt(const OmEncoderConfig* c, const DecTrainWs& ws, int l) {
  const size_t es = (c->dtype == OM_BF16 || c->dtype == OM_F16) ? 2 : 4, H = c->hidden, F = c->ffn;
  char* p = ws.wt + ws.swt * l;
  DecWt v;
  v.sa_v = p; p += H * H * es; v.sa_o = p; p += H * H * es; v.ca_q = p; p += H * H * es;
  v.ca_kv = p; p += 2 * H * H * es; v.ca_o = p; p += H * H * es;
  v.f1 = p; p += F * H * es; v.f1g = p; p += F * H * es; v.f2 = p;
  return v;
}
int dec_check(const OmEncoderConfig* c, const OmT5DecoderWeights* w, int64_t L) {
  if (c->arch != OM_ARCH_T5) OM_FAIL("decoder step: T5 only");
  if (c->dtype != OM_F32 && c->dtype != OM_BF16 && c->dtype != OM_F16) OM_FAIL("dtype must be OM_F32, OM_BF16 or OM_F16");
  if (c->head_dim != 64 || c->n_heads * 64 != c->hidden) OM_FAIL("head_dim must be 64 (inner dim == d_model)");
  if (L < 1 || L > DEC_MAX_L) OM_FAIL("sequence length must be in [1,1024]");
  if (!w->layers_host || w->n_layers < 1 || !w->start_emb || !w->final_ln_g) OM_FAIL("incomplete decoder weights");
  const int akind = c->act & 0xff;
  if (akind != OM_ACT_RELU && akind != OM_ACT_GELU_TANH) OM_FAIL("T5 decoder supports relu and gated gelu_new feed-forward layers");
  for (int l = 0; l < w->n_layers; ++l) {
    const OmT5DecoderLayer& lw = w->layers_host[l];
    if (!lw.sa_v_w || !lw.sa_o_w || !lw.sa_ln_g || !lw.ca_q_w || !lw.ca_kv_w || !lw.ca_o_w || !lw.ca_ln_g || !lw.ffn1_w ||
        !lw.ffn2_w || !lw.ffn_ln_g) OM_FAIL("incomplete decoder layer weights");
    if ((akind == OM_ACT_GELU_TANH) != (lw.ffn1g_w != nullptr)) OM_FAIL("the gated feed-forward (and only it) needs ffn1g_w");
  }
  return 0;
}
// dW[N,K] += dY[M,N]^T X[M,K] over M rows (train.hip's helper with an explicit row count)
int dec_wgrad(int dt, const void* dY, int N, const void* X, int K, float* dW, int64_t M, DecTrainWs& ws, hipStream_t s) {
  if (!dW) OM_FAIL("decoder gradients: a weight-gradient buffer is NULL");
  if (omk_gemm_tn_ok(dt, M, N, K, N, K)) return omk_gemm_tn(dt, dY, N, X, K, dW, K, nullptr, M, N, K, s);
  const int64_t Mp = (M + 63) / 64 * 64;
  if (omk_transpose(dt, dY, N, M, N, ws.tl, Mp, Mp, 0, s)) return 1;
  if (omk_transpose(dt, X, K, M, K, ws.tr, Mp, Mp, 0, s)) return 1;
  return omk_gemm_splitk(dt, ws.tl, Mp, ws.tr, Mp, dW, K, N, K, Mp, s);
}
// C[M,N] (out_dt) = A[M,K] W[N,K]^T (+ resid), optional dropout of the product before the residual add
int dec_gemm(int dt, const void* A, int K, const void* W, void* C, int out_dt, int64_t M, int N, const void* resid, float drop_p,
             uint64_t seed, hipStream_t s) {
  GemmEpilogue e = {};
  e.resid = resid; e.ldr = N; e.drop_p = drop_p; e.seed = seed;
  return omk_gemm(dt, A, K, W, K, out_dt, C, N, M, N, K, e, s);
}

template <typename T>
int dec_train_forward_t(const OmEncoderConfig* c, const OmT5DecoderWeights* w, const void* enc_hidden, const int64_t* attention_mask,
                        int64_t B, int64_t L, float p, uint64_t seed, const DecTape& t, DecTrainWs& ws, float* out_hidden, hipStream_t s) {
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads, nl = w->n_layers;
  const int64_t M = B * L;
  const int kind = (c->act & 0xff) == OM_ACT_GELU_TANH ? 1 : 0;
  const unsigned gbh = (unsigned)((B * H + 255) / 256);
  hipLaunchKernelGGL((dec_start_kernel<T>), dim3(gbh), dim3(256), 0, s, w->start_emb, (T*)t.at(0, t.o_x0), B, H);
  OM_LAUNCH_CHECK();
  if (p > 0.f) RUN(omk_dropout(dt, t.at(0, t.o_x0), t.at(0, t.o_x0), B * H, p, dec_site(seed, 0, 0), s));
  for (int l = 0; l < nl; ++l) {
    const OmT5DecoderLayer& lw = w->layers_host[l];
    char *x0 = t.at(l, t.o_x0), *v = t.at(l, t.o_v), *x1 = t.at(l, t.o_x1), *q = t.at(l, t.o_q), *ctx = t.at(l, t.o_ctx);
    char *x2 = t.at(l, t.o_x2), *f = t.at(l, t.o_f), *f2 = t.at(l, t.o_f2), *kv = t.at(l, t.o_kv);
    char* x3 = l + 1 < nl ? t.at(l + 1, t.o_x0) : t.xl;
    // self-attention over the single position
    RUN(omk_layernorm(dt, x0, H, ws.n, H, lw.sa_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    RUN(dec_gemm(dt, ws.n, H, lw.sa_v_w, v, dt, B, H, nullptr, 0.f, 0, s));
    if (p > 0.f) { hipLaunchKernelGGL((dec_head_drop_kernel<T>), dim3(gbh), dim3(256), 0, s, (T*)v, B, H, p, dec_site(seed, l, 1)); OM_LAUNCH_CHECK(); }
    RUN(dec_gemm(dt, v, H, lw.sa_o_w, x1, dt, B, H, x0, p, dec_site(seed, l, 6), s));
    // cross-attention
    RUN(omk_layernorm(dt, x1, H, ws.n, H, lw.ca_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    RUN(dec_gemm(dt, ws.n, H, lw.ca_q_w, q, dt, B, H, nullptr, 0.f, 0, s));
    RUN(dec_gemm(dt, enc_hidden, H, lw.ca_kv_w, kv, dt, M, 2 * H, nullptr, 0.f, 0, s));
    hipLaunchKernelGGL((dec_cross_kernel<T>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const T*)q, (const T*)kv, attention_mask,
                       (T*)ctx, (int)L, H, nh, p, dec_site(seed, l, 2));
    OM_LAUNCH_CHECK();
    RUN(dec_gemm(dt, ctx, H, lw.ca_o_w, x2, dt, B, H, x1, p, dec_site(seed, l, 3), s));
    // feed-forward
    RUN(omk_layernorm(dt, x2, H, ws.n, H, lw.ffn_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    RUN(dec_gemm(dt, ws.n, H, lw.ffn1_w, f, dt, B, F, nullptr, 0.f, 0, s));
    if (kind) RUN(dec_gemm(dt, ws.n, H, lw.ffn1g_w, f2, dt, B, F, nullptr, 0.f, 0, s));
    RUN(omk_t5_act_fwd(dt, f, kind ? f2 : nullptr, ws.g, B * F, kind, s));
    if (p > 0.f) RUN(omk_dropout(dt, ws.g, ws.g, B * F, p, dec_site(seed, l, 5), s));
    RUN(dec_gemm(dt, ws.g, F, lw.ffn2_w, x3, dt, B, H, x2, p, dec_site(seed, l, 4), s));
  }
  RUN(omk_layernorm_f32out(dt, t.xl, H, out_hidden, H, w->final_ln_g, nullptr, B, H, c->ln_eps, 1, s));
  if (p > 0.f) RUN(omk_dropout(OM_F32, out_hidden, out_hidden, B * H, p, dec_site(seed, nl, 1), s));
  return 0;
}

template <typename T>
int dec_train_backward_t(const OmEncoderConfig* c, const OmT5DecoderWeights* w, const void* enc_hidden, const int64_t* attention_mask,
                         int64_t B, int64_t L, float p, uint64_t seed, const DecTape& t, DecTrainWs& ws, const float* d_out,
                         const OmT5DecoderGrads* g, void* d_enc_hidden, hipStream_t s) {
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads, nl = w->n_layers;
  const int64_t M = B * L;
  const int kind = (c->act & 0xff) == OM_ACT_GELU_TANH ? 1 : 0;
  const unsigned gbh = (unsigned)((B * H + 255) / 256);
  // every weight transposed once (the data-gradient GEMMs' B operands)
  {
    std::vector<const void*> in; std::vector<void*> out; std::vector<int> R, C;
    auto add = [&](const void* src, void* dst, int r, int cc) { in.push_back(src); out.push_back(dst); R.push_back(r); C.push_back(cc); };
    for (int l = 0; l < nl; ++l) {
      const OmT5DecoderLayer& lw = w->layers_host[l];
      const DecWt wt = dec_wt(c, ws, l);
      add(lw.sa_v_w, wt.sa_v, H, H); add(lw.sa_o_w, wt.sa_o, H, H); add(lw.ca_q_w, wt.ca_q, H, H);
      add(lw.ca_kv_w, wt.ca_kv, 2 * H, H); add(lw.ca_o_w, wt.ca_o, H, H);
      add(lw.ffn1_w, wt.f1, F, H); if (kind) add(lw.ffn1g_w, wt.f1g, F, H);
      add(lw.ffn2_w, wt.f2, H, F);
    }
    RUN(omk_transpose_batch(dt, in.data(), out.data(), R.data(), C.data(), (int)in.size(), s));
  }
  // final dropout + RMSNorm
  const float* dof = d_out;
  if (p > 0.f) { RUN(omk_dropout(OM_F32, d_out, ws.do32, B * H, p, dec_site(seed, nl, 1), s)); dof = ws.do32; }
  const void* dy = dof;
  if (dt != OM_F32) {
    hipLaunchKernelGGL((dec_cast_kernel<T>), dim3(gbh), dim3(256), 0, s, dof, (T*)ws.dn, B * H);
    OM_LAUNCH_CHECK();
    dy = ws.dn;
  }
  char* dx = ws.dxa;
  char* dx_other = ws.dxb;
  RUN(omk_norm_bwd(dt, dy, t.xl, w->final_ln_g, dx, g->final_ln_g, nullptr, B, H, c->ln_eps, 1, nullptr, s));
  bool first_e = true;                         // dE: the first layer processed writes, the others add (f32 accumulation)
  for (int l = nl - 1; l >= 0; --l) {
    const OmT5DecoderLayer& lw = w->layers_host[l];
    const OmT5DecoderLayerGrads& lg = g->layers_host[l];
    const DecWt wt = dec_wt(c, ws, l);
    char *x0 = t.at(l, t.o_x0), *v = t.at(l, t.o_v), *x1 = t.at(l, t.o_x1), *q = t.at(l, t.o_q), *ctx = t.at(l, t.o_ctx);
    char *x2 = t.at(l, t.o_x2), *f = t.at(l, t.o_f), *f2 = t.at(l, t.o_f2), *kv = t.at(l, t.o_kv);
    // ---- feed-forward branch: x3 = x2 + drop(drop(act(n3 W1^T)) W2^T)
    const char* dO = dx;
    if (p > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, B * H, p, dec_site(seed, l, 4), s)); dO = ws.dd; }
    RUN(omk_t5_act_fwd(dt, f, kind ? f2 : nullptr, ws.g, B * F, kind, s));
    if (p > 0.f) RUN(omk_dropout(dt, ws.g, ws.g, B * F, p, dec_site(seed, l, 5), s));
    RUN(dec_wgrad(dt, dO, H, ws.g, F, lg.ffn2_w, B, ws, s));
    RUN(dec_gemm(dt, dO, H, wt.f2, ws.dg, dt, B, F, nullptr, 0.f, 0, s));                 // dg = dO W2
    if (p > 0.f) RUN(omk_dropout(dt, ws.dg, ws.dg, B * F, p, dec_site(seed, l, 5), s));
    RUN(omk_t5_act_bwd(dt, ws.dg, f, kind ? f2 : nullptr, ws.dg, ws.dg2, B * F, kind, s));
    RUN(omk_layernorm(dt, x2, H, ws.n, H, lw.ffn_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    RUN(dec_wgrad(dt, ws.dg, F, ws.n, H, lg.ffn1_w, B, ws, s));
    if (kind) RUN(dec_wgrad(dt, ws.dg2, F, ws.n, H, lg.ffn1g_w, B, ws, s));
    RUN(dec_gemm(dt, ws.dg, F, wt.f1, ws.dn, dt, B, H, nullptr, 0.f, 0, s));              // dn3 = df W1
    if (kind) RUN(dec_gemm(dt, ws.dg2, F, wt.f1g, ws.dn, dt, B, H, ws.dn, 0.f, 0, s));     //     += df2 W1g
    RUN(omk_norm_bwd(dt, ws.dn, x2, lw.ffn_ln_g, dx_other, lg.ffn_ln_g, nullptr, B, H, c->ln_eps, 1, dx, s));   // d x2
    { char* tmp = dx; dx = dx_other; dx_other = tmp; }
    // ---- cross-attention branch: x2 = x1 + drop(ctx Wo^T)
    const char* dC = dx;
    if (p > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, B * H, p, dec_site(seed, l, 3), s)); dC = ws.dd; }
    RUN(dec_wgrad(dt, dC, H, ctx, H, lg.ca_o_w, B, ws, s));
    RUN(dec_gemm(dt, dC, H, wt.ca_o, ws.dctx, dt, B, H, nullptr, 0.f, 0, s));             // dctx = dC Wo
    hipLaunchKernelGGL((dec_cross_bwd_kernel<T>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const T*)q, (const T*)kv, attention_mask,
                       (const T*)ws.dctx, (T*)ws.dq, (T*)ws.dkv, (int)L, H, nh, p, dec_site(seed, l, 2));
    OM_LAUNCH_CHECK();
    RUN(dec_wgrad(dt, ws.dkv, 2 * H, enc_hidden, H, lg.ca_kv_w, M, ws, s));               // d(Wk | Wv)
    RUN(dec_gemm(dt, ws.dkv, 2 * H, wt.ca_kv, ws.dE, OM_F32, M, H, first_e ? nullptr : ws.dE, 0.f, 0, s));   // dE (+)= dkv Wkv
    first_e = false;
    RUN(omk_layernorm(dt, x1, H, ws.n, H, lw.ca_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    RUN(dec_wgrad(dt, ws.dq, H, ws.n, H, lg.ca_q_w, B, ws, s));
    RUN(dec_gemm(dt, ws.dq, H, wt.ca_q, ws.dn, dt, B, H, nullptr, 0.f, 0, s));             // dn2 = dq Wq
    RUN(omk_norm_bwd(dt, ws.dn, x1, lw.ca_ln_g, dx_other, lg.ca_ln_g, nullptr, B, H, c->ln_eps, 1, dx, s));    // d x1
    { char* tmp = dx; dx = dx_other; dx_other = tmp; }
    // ---- self-attention branch: x1 = x0 + drop(v' Wo^T), v' = head-dropped (n1 Wv^T)
    const char* dA = dx;
    if (p > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, B * H, p, dec_site(seed, l, 6), s)); dA = ws.dd; }
    RUN(dec_wgrad(dt, dA, H, v, H, lg.sa_o_w, B, ws, s));
    RUN(dec_gemm(dt, dA, H, wt.sa_o, ws.dq, dt, B, H, nullptr, 0.f, 0, s));                // dv' = dA Wo
    if (p > 0.f) { hipLaunchKernelGGL((dec_head_drop_kernel<T>), dim3(gbh), dim3(256), 0, s, (T*)ws.dq, B, H, p, dec_site(seed, l, 1)); OM_LAUNCH_CHECK(); }
    RUN(omk_layernorm(dt, x0, H, ws.n, H, lw.sa_ln_g, nullptr, B, H, c->ln_eps, 1, s));
    RUN(dec_wgrad(dt, ws.dq, H, ws.n, H, lg.sa_v_w, B, ws, s));
    RUN(dec_gemm(dt, ws.dq, H, wt.sa_v, ws.dn, dt, B, H, nullptr, 0.f, 0, s));             // dn1 = dv Wv
    RUN(omk_norm_bwd(dt, ws.dn, x0, lw.sa_ln_g, dx_other, lg.sa_ln_g, nullptr, B, H, c->ln_eps, 1, dx, s));    // d x0
    { char* tmp = dx; dx = dx_other; dx_other = tmp; }
  }
  // the start embedding: every row is the same table row
  const char* de = dx;
  if (p > 0.f) { RUN(omk_dropout(dt, dx, ws.dd, B * H, p, dec_site(seed, 0, 0), s)); de = ws.dd; }
  if (g->start_emb) RUN(omk_colsum(dt, de, H, B, H, g->start_emb, s));
  // gradient w.r.t. the encoder output, in the compute dtype
  if (dt == OM_F32) {
    OM_HIP(hipMemcpyAsync(d_enc_hidden, ws.dE, (size_t)M * H * 4, hipMemcpyDeviceToDevice, s));
  } else {
    const int64_t n = M * H;
    hipLaunchKernelGGL((dec_cast_kernel<T>), dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, s, ws.dE, (T*)d_enc_hidden, n);
    OM_LAUNCH_CHECK();
  }
  return 0;
}
}  // namespace

extern "C" size_t om_t5_decoder_tape_bytes(const OmEncoderConfig* cfg, int n_layers, int64_t B, int64_t L) {
  if (!cfg || n_layers <= 0 || B <= 0 || L <= 0) return 0;
  return dec_tape(cfg, n_layers, B, L, nullptr).total;
}
extern "C" size_t om_t5_decoder_train_workspace_bytes(const OmEncoderConfig* cfg, int n_layers, int64_t B, int64_t L) {
  if (!cfg || n_layers <= 0 || B <= 0 || L <= 0) return 0;
  return dec_train_ws(cfg, n_layers, B, L, nullptr).total;
}

extern "C" int om_t5_decoder_train_forward(const OmEncoderConfig* c, const OmT5DecoderWeights* w, const void* enc_hidden,
                                           const int64_t* attention_mask, int64_t B, int64_t L, float dropout, uint64_t seed,
                                           void* tape_mem, size_t tape_bytes, float* out_hidden, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!c || !w || !enc_hidden || !attention_mask || !out_hidden || !tape_mem || !workspace) OM_FAIL("null argument");
  if (B <= 0) return 0;
  if (dec_check(c, w, L)) return 1;
  if (((uintptr_t)workspace & 255) || ((uintptr_t)tape_mem & 255)) OM_FAIL("tape/workspace must be 256-byte aligned");
  if (dropout < 0.f || dropout >= 1.f) OM_FAIL("dropout must be in [0, 1)");
  const DecTape t = dec_tape(c, w->n_layers, B, L, (char*)tape_mem);
  DecTrainWs ws = dec_train_ws(c, w->n_layers, B, L, (char*)workspace);
  if (t.total > tape_bytes || ws.total > workspace_bytes) OM_FAIL("tape or workspace too small");
  if (c->dtype == OM_BF16)
    return dec_train_forward_t<bf16_t>(c, w, enc_hidden, attention_mask, B, L, dropout, seed, t, ws, out_hidden, (hipStream_t)stream);
  if (c->dtype == OM_F16)
    return dec_train_forward_t<f16_t>(c, w, enc_hidden, attention_mask, B, L, dropout, seed, t, ws, out_hidden, (hipStream_t)stream);
  return dec_train_forward_t<float>(c, w, enc_hidden, attention_mask, B, L, dropout, seed, t, ws, out_hidden, (hipStream_t)stream);
}

extern "C" int om_t5_decoder_train_backward(const OmEncoderConfig* c, const OmT5DecoderWeights* w, const void* enc_hidden,
                                            const int64_t* attention_mask, int64_t B, int64_t L, float dropout, uint64_t seed,
                                            const void* tape_mem, const float* d_out, const OmT5DecoderGrads* g,
                                            void* d_enc_hidden, void* workspace, size_t workspace_bytes, void* stream) {
  if (!c || !w || !enc_hidden || !attention_mask || !d_out || !tape_mem || !workspace || !g || !d_enc_hidden) OM_FAIL("null argument");
  if (B <= 0) return 0;
  if (dec_check(c, w, L)) return 1;
  if (!g->layers_host || !g->final_ln_g) OM_FAIL("incomplete decoder gradient buffers");
  if (((uintptr_t)workspace & 255) || ((uintptr_t)tape_mem & 255)) OM_FAIL("tape/workspace must be 256-byte aligned");
  const DecTape t = dec_tape(c, w->n_layers, B, L, (char*)const_cast<void*>(tape_mem));
  DecTrainWs ws = dec_train_ws(c, w->n_layers, B, L, (char*)workspace);
  if (ws.total > workspace_bytes) OM_FAIL("workspace too small");
  if (c->dtype == OM_BF16)
    return dec_train_backward_t<bf16_t>(c, w, enc_hidden, attention_mask, B, L, dropout, seed, t, ws, d_out, g, d_enc_hidden, (hipStream_t)stream);
  if (c->dtype == OM_F16)
    return dec_train_backward_t<f16_t>(c, w, enc_hidden, attention_mask, B, L, dropout, seed, t, ws, d_out, g, d_enc_hidden, (hipStream_t)stream);
  return dec_train_backward_t<float>(c, w, enc_hidden, attention_mask, B, L, dropout, seed, t, ws, d_out, g, d_enc_hidden, (hipStream_t)stream);
}

// Backward of y[B,D] = x[B,K] W[D,K]^T in f32 (the LinearHead / the two LM-head rows behind a trained decoder position):
// dW = dy^T x (written, not added), dx = dy W.  Either output may be NULL.
extern "C" int om_linear_f32_backward(const float* dy, const float* x, const float* w, float* dw, float* dx, int B, int D, int K,
                                      void* stream) {
  if (!dy || !x || !w) OM_FAIL("null argument");
  if (B <= 0 || D <= 0 || K <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dw) RUN(omk_small_tn(dy, x, dw, B, D, K, s));
  if (dx) RUN(omk_small_nn(dy, w, dx, B, D, K, s));
  return 0;
}
