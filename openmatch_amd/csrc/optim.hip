// om_grad_sqnorm / om_adamw_step: global-norm clipping + AdamW + refresh of the packed compute-dtype weight copies in two passes
// over memory, for a whole model in three launches.  Stands in for what the reference's training loop does per optimizer step
// through HF Trainer (trainer/dense_trainer.py:27-108 inherits Trainer.train: clip_grad_norm_(max_grad_norm) -> AdamW.step) plus
// the re-packing of the encoder's 16-bit weights that the HIP forward needs after every update.
//
// Why here: torch's multi-tensor AdamW + clip ran 1.09 ms of a 10.8 ms training step (profiles/r04_train_kernel_stats_v0.csv: seven
// 110 us launches whose tensor lists are limited by the kernel-argument size, a norm pass, a scaling pass over every gradient), and
// its fused form does not bump the parameters' version counters, so the packed bf16 copies were never refreshed behind it.  One
// kernel reads g, p, m, v once and writes p, m, v and the 16-bit copies the next forward reads: 30 bytes per parameter (3.3 GB at
// bert-base, ~0.6 ms at the ~5.5 TB/s a read-modify-write stream reaches).  HBM-bound; no LDS, no MFMA.
#include <math.h>

#include "common.h"

namespace {
constexpr int CHUNK = OM_ADAM_CHUNK;       // elements per workgroup
constexpr int THREADS = 256;

__device__ __forceinline__ void store_shadow(void* dst, int dtype, int64_t i, const f32x4_t& p) {
  if (dtype == OM_BF16) {
    *(uint2*)((bf16_t*)dst + i) = make_uint2(Half16<bf16_t>::pack2(p[0], p[1]), Half16<bf16_t>::pack2(p[2], p[3]));
  } else if (dtype == OM_F16) {
    *(uint2*)((f16_t*)dst + i) = make_uint2(Half16<f16_t>::pack2(p[0], p[1]), Half16<f16_t>::pack2(p[2], p[3]));
  } else {
    *(f32x4_t*)((float*)dst + i) = p;
  }
}

// partial[c] = sum of squares of chunk c's gradient elements (fixed order inside a chunk: the result does not depend on timing)
__global__ __launch_bounds__(THREADS) void grad_sqnorm_kernel(const OmAdamTensor* __restrict__ T, const int32_t* __restrict__ chunks,
                                                              float* __restrict__ partial) {
  const OmAdamTensor t = T[chunks[2 * blockIdx.x]];
  const int64_t base = (int64_t)chunks[2 * blockIdx.x + 1] * CHUNK;
  const int64_t n = t.n - base < CHUNK ? t.n - base : CHUNK;
  const float* g = t.g ? t.g + base : nullptr;        // (an absent gradient stays NULL whatever the chunk's offset: ADVICE r5)
  float s = 0.f;
  if (g) {
    const bool vec = ((uintptr_t)g & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = threadIdx.x; i < n4; i += THREADS) {
      const f32x4_t x = __builtin_nontemporal_load((const f32x4_t*)g + i);
      s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += THREADS) s += g[i] * g[i];
  }
  __shared__ float red[THREADS / 64];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// out[0] = sum of the partials, strided per thread then a tree: one fixed order
__global__ __launch_bounds__(1024) void sqnorm_reduce_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
  __shared__ float red[16];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += red[i];
    out[0] = tot;
  }
}

struct AdamArgs { float lr_wd_unused, step_size, beta1, beta2, eps, bc2_sqrt_inv, max_norm, grad_scale, lr; int clip, skip_nonfinite; const float* scale_state; long long step; };

// torch.optim.AdamW's update (torch/optim/adamw.py, single-tensor form), element by element:
//   p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with g = grad * grad_scale * min(1, max_norm / (|grad * grad_scale|_2 + 1e-6)) (torch.nn.utils.clip_grad_norm_).
__global__ __launch_bounds__(THREADS) void adamw_kernel(const OmAdamTensor* __restrict__ T, const int32_t* __restrict__ chunks,
                                                        const float* __restrict__ gnorm_sq, AdamArgs a) {
  const OmAdamTensor t = T[chunks[2 * blockIdx.x]];
  if (!t.g) return;
  const int64_t base = (int64_t)chunks[2 * blockIdx.x + 1] * CHUNK;
  const int64_t n = t.n - base < CHUNK ? t.n - base : CHUNK;
  float gs = a.grad_scale;
  if (a.scale_state) {                                                // the dynamic loss scale's state, kept on the device (om_loss_scale_update)
    gs *= a.scale_state[1];                                           // 1 / scale
    const float skipped = a.scale_state[3];
    if (skipped > 0.f) {       // GradScaler never calls optimizer.step() on a skipped step: the bias corrections count the steps TAKEN
      __shared__ float bc[2];
      if (threadIdx.x == 0) {
        const double n = (double)a.step - (double)skipped;
        bc[0] = (float)((double)a.lr / (1.0 - pow((double)a.beta1, n)));
        bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, n)));
      }
      __syncthreads();
      a.step_size = bc[0]; a.bc2_sqrt_inv = bc[1];
    }
  }
  if (gnorm_sq) {
    const float nrm = sqrtf(gnorm_sq[0]) * fabsf(gs);
    if (a.skip_nonfinite && !(nrm <= 3.0e38f)) return;           // inf / nan gradients: the step is skipped (GradScaler semantics)
    if (a.clip) gs *= fminf(1.0f, a.max_norm / (nrm + 1e-6f));
  }
  const float decay = 1.0f - a.lr * t.weight_decay;
  float* p = t.p + base; const float* g = t.g + base; float* m = t.m + base; float* v = t.v + base;
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 &&
                   (!t.shadow0 || ((uintptr_t)t.shadow0 & 15) == 0) && (!t.shadow1 || ((uintptr_t)t.shadow1 & 15) == 0);
  const int64_t n4 = vec ? n / 4 : 0;
  for (int64_t i = threadIdx.x; i < n4; i += THREADS) {
    const f32x4_t gv = __builtin_nontemporal_load((const f32x4_t*)g + i) * gs;
    f32x4_t pv = *((const f32x4_t*)p + i), mv = *((const f32x4_t*)m + i), vv = *((const f32x4_t*)v + i);
    pv *= decay;
    mv += (gv - mv) * (1.0f - a.beta1);
    vv = vv * a.beta2 + gv * gv * (1.0f - a.beta2);
    f32x4_t den;
#pragma unroll
    for (int e = 0; e < 4; ++e) den[e] = sqrtf(vv[e]) * a.bc2_sqrt_inv + a.eps;
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[e] -= a.step_size * (mv[e] / den[e]);
    *((f32x4_t*)p + i) = pv; *((f32x4_t*)m + i) = mv; *((f32x4_t*)v + i) = vv;
    if (t.shadow0) store_shadow(t.shadow0, t.shadow0_dtype, base + 4 * i, pv);
    if (t.shadow1) store_shadow(t.shadow1, t.shadow1_dtype, base + 4 * i, pv);
  }
  for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += THREADS) {
    const float gv = g[i] * gs;
    float pv = p[i] * decay, mv = m[i], vv = v[i];
    mv += (gv - mv) * (1.0f - a.beta1);
    vv = vv * a.beta2 + gv * gv * (1.0f - a.beta2);
    pv -= a.step_size * (mv / (sqrtf(vv) * a.bc2_sqrt_inv + a.eps));
    p[i] = pv; m[i] = mv; v[i] = vv;
    for (int k = 0; k < 2; ++k) {
      void* sh = k ? t.shadow1 : t.shadow0;
      const int dt = k ? t.shadow1_dtype : t.shadow0_dtype;
      if (!sh) continue;
      if (dt == OM_BF16) ((bf16_t*)sh)[base + i] = f32_to_bf16(pv);
      else if (dt == OM_F16) ((f16_t*)sh)[base + i] = (f16_t)pv;
      else ((float*)sh)[base + i] = pv;
    }
  }
}
}  // namespace

extern "C" int om_grad_sqnorm(const OmAdamTensor* tensors, const int32_t* chunks, int n_chunks, float* partial, float* out_sq,
                              void* stream) {
  if (n_chunks < 0 || (n_chunks > 0 && (!tensors || !chunks || !partial)) || !out_sq) OM_FAIL("null argument");
  hipStream_t s = (hipStream_t)stream;
  if (n_chunks > 0) hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)n_chunks), dim3(THREADS), 0, s, tensors, chunks, partial);
  hipLaunchKernelGGL(sqnorm_reduce_kernel, dim3(1), dim3(1024), 0, s, partial, n_chunks, out_sq);
  OM_LAUNCH_CHECK();
  return 0;
}

// The dynamic loss scale of float16 training (torch.cuda.amp.GradScaler.update, which HF Trainer drives for the reference's --fp16:
// trainer/dense_trainer.py:141-149), as one thread on the device: a non-finite gradient norm halves the scale (the optimizer
// skipped that step by itself: om_adamw_step skip_nonfinite) and restarts the count of clean steps; `growth_interval` clean
// steps double it.  state = {scale, 1 / scale, clean steps, steps skipped so far}.
__global__ void loss_scale_update_kernel(const float* __restrict__ gnorm_sq, float* __restrict__ state, int growth_interval) {
  if (threadIdx.x || blockIdx.x) return;
  float scale = state[0], good = state[2];
  const bool finite = gnorm_sq[0] <= 3.0e38f;
  if (!finite) { scale = fmaxf(scale * 0.5f, 1.0f); good = 0.f; state[3] += 1.f; }
  else if (++good >= (float)growth_interval) { scale = fminf(scale * 2.0f, 16777216.0f); good = 0.f; }
  state[0] = scale; state[1] = 1.0f / scale; state[2] = good;
}
extern "C" int om_loss_scale_update(const float* gnorm_sq, float* state4, int growth_interval, void* stream) {
  if (!gnorm_sq || !state4 || growth_interval < 1) OM_FAIL("bad argument");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gnorm_sq, state4, growth_interval);
  OM_LAUNCH_CHECK();
  return 0;
}

extern "C" int om_adamw_step(const OmAdamTensor* tensors, const int32_t* chunks, int n_chunks, float lr, float beta1, float beta2,
                             float eps, int64_t step, const float* gnorm_sq, float max_norm, float grad_scale, int skip_nonfinite,
                             const float* scale_state, void* stream) {
  if (n_chunks < 0 || (n_chunks > 0 && (!tensors || !chunks))) OM_FAIL("null argument");
  if (step < 1) OM_FAIL("step counts from 1");
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) OM_FAIL("betas must lie in [0, 1)");
  if (max_norm > 0.f && !gnorm_sq) OM_FAIL("clipping needs the squared gradient norm (om_grad_sqnorm)");
  if (n_chunks == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  AdamArgs a;
  a.lr_wd_unused = 0.f;
  a.lr = lr;
  a.step_size = (float)((double)lr / bc1);
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc2_sqrt_inv = (float)(1.0 / sqrt(bc2));
  a.max_norm = max_norm; a.grad_scale = grad_scale;
  a.clip = max_norm > 0.f ? 1 : 0;
  a.skip_nonfinite = skip_nonfinite;
  a.scale_state = scale_state; a.step = (long long)step;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)n_chunks), dim3(THREADS), 0, (hipStream_t)stream, tensors, chunks, gnorm_sq, a);
  OM_LAUNCH_CHECK();
  return 0;
}
