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
#include "attn_common.h"
#include "kernels.h"

namespace {
#define RUN(expr) do { if (expr) return 1; } while (0)

// x[b, :] = emb[:]  (f32 table row -> compute dtype)
template <typename T>
__global__ void dec_start_kernel(const float* __restrict__ emb, T* __restrict__ x, int64_t B, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * H) ElemOps<T>::store(x + i, emb[i % H]);
}

// Single-query cross attention.  Block = (batch b, head h), 256 threads: thread l scores key l (L <= 256), block softmax,
// then 64 threads x 4 row groups accumulate ctx[d] = sum_l p[l] V[l][d].  kv: [B, L, 2H] (K | V), q, ctx: [B, H].
template <typename T>
__global__ __launch_bounds__(256) void dec_cross_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                        const int64_t* __restrict__ mask, T* __restrict__ ctx, int L,
                                                        int H, int heads) {
  __shared__ float sq[64];
  __shared__ float sp[256];
  __shared__ float red[8];
  __shared__ float part[4][64];
  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int tid = threadIdx.x;
  if (tid < 64) sq[tid] = ElemOps<T>::load(q + b * H + h * 64 + tid);
  __syncthreads();
  float sc = -INFINITY;
  if (tid < L) {
    const T* kr = kv + (b * L + tid) * 2 * (int64_t)H + h * 64;
    float a = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) a = fmaf(sq[d], ElemOps<T>::load(kr + d), a);
    sc = a + (mask[b * L + tid] != 0 ? 0.f : -3.4028235e38f);        // HF: (1 - mask) * finfo.min
  }
  float mx = wave_max(sc);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float e = tid < L ? expf(sc - mx) : 0.f;
  float sum = wave_sum(e);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  sp[tid] = e;
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
  const size_t es = c->dtype == OM_BF16 ? 2 : 4;
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
  if (c->dtype != OM_F32 && c->dtype != OM_BF16) OM_FAIL("dtype must be OM_F32 or OM_BF16");
  if (c->head_dim != 64 || c->n_heads * 64 != c->hidden) OM_FAIL("head_dim must be 64 (inner dim == d_model)");
  if (L < 1 || L > 256) OM_FAIL("sequence length must be in [1,256]");
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
    if (dt == OM_BF16) hipLaunchKernelGGL((dec_cross_kernel<bf16_t>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const bf16_t*)ws.t, (const bf16_t*)ws.kv, attention_mask, (bf16_t*)ws.ctx, (int)L, H, nh);
    else hipLaunchKernelGGL((dec_cross_kernel<float>), dim3((unsigned)(B * nh)), dim3(256), 0, s, (const float*)ws.t, (const float*)ws.kv, attention_mask, (float*)ws.ctx, (int)L, H, nh);
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
  else hipLaunchKernelGGL((dec_final_kernel<float>), dim3((unsigned)B), dim3(256), 0, s, (const float*)ws.x, w->final_ln_g, out_hidden, H, c->ln_eps);
  OM_LAUNCH_CHECK();
#undef GEMM
  return 0;
}
