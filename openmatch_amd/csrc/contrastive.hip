// In-batch-negatives contrastive loss, forward + backward (K10 + K11).
// Replaces  scores = q @ p.T ; CrossEntropyLoss(mean)(scores, arange(Q) * n_psg)  and its
// autograd (modeling/dense_retrieval_model.py:113-122; loss.py:9-15).  Sizes are tiny
// ([64,768] x [768,512] at the reference's 8-GPU config), so this is latency-bound: the
// score GEMM runs on the exact-f32 MFMA path, the softmax is one wavefront per row, and the
// two gradient contractions are plain coalesced f32 loops over the LOCAL rows only.
#include "kernels.h"

// Reductions of F.cross_entropy (loss.py:9-15 passes `reduction` through) and its ignore_index
#define OM_CE_MEAN 0
#define OM_CE_SUM 1
#define OM_CE_NONE 2
#define OM_CE_IGNORE (-100)

// number of rows that take part (target != ignore_index), as a float for the mean's denominator
__global__ void ce_count_kernel(const int64_t* __restrict__ target, int Qg, float* __restrict__ count) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < Qg; i += 64) acc += target[i] != OM_CE_IGNORE ? 1.f : 0.f;
  acc = wave_sum(acc);
  if (threadIdx.x == 0) *count = acc;
}

// row-wise: loss_i = logsumexp(S[i,:]) - S[i,target_i];  dS[i,:] = (softmax - onehot) * coef_i with
//   coef_i = scale / valid (mean) | scale (sum) | scale * row_grad[i] (none; row_grad NULL = 1)
// target NULL: target_i = i * n_psg (the in-batch positive, modeling :115-120).  Ignored rows: loss 0, dS 0.
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ S,
                                                      float* __restrict__ dS, int Qg, int Pg,
                                                      int n_psg, const int64_t* __restrict__ target, int reduction,
                                                      float scale, const float* __restrict__ row_grad,
                                                      const float* __restrict__ valid, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Qg) return;
  const int64_t tgt = target ? target[i] : (int64_t)i * n_psg;
  if (tgt == OM_CE_IGNORE) {
    if (lane == 0) row_loss[i] = 0.f;
    if (dS) for (int j = lane; j < Pg; j += 64) dS[(int64_t)i * Pg + j] = 0.f;
    return;
  }
  const float* s = S + (int64_t)i * Pg;
  float mx = -INFINITY;
  for (int j = lane; j < Pg; j += 64) mx = fmaxf(mx, s[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < Pg; j += 64) sum += expf(s[j] - mx);
  sum = wave_sum(sum);
  const float lse = mx + logf(sum);
  if (lane == 0) row_loss[i] = lse - s[tgt];
  if (dS) {
    float coef = scale;
    if (reduction == OM_CE_MEAN) coef = scale / fmaxf(valid ? *valid : (float)Qg, 1.f);
    else if (reduction == OM_CE_NONE && row_grad) coef = scale * row_grad[i];
    const float inv = 1.0f / sum;
    for (int j = lane; j < Pg; j += 64) {
      const float p = expf(s[j] - mx) * inv;
      dS[(int64_t)i * Pg + j] = (p - (j == tgt ? 1.f : 0.f)) * coef;
    }
  }
}

// loss = scale * mean | sum of the row losses (deterministic single-wave reduction), or the scaled rows themselves
__global__ void ce_reduce_kernel(const float* __restrict__ row_loss, int Qg, float scale, int reduction,
                                 const float* __restrict__ valid, float* __restrict__ loss) {
  if (reduction == OM_CE_NONE) {
    for (int i = threadIdx.x; i < Qg; i += 64) loss[i] = row_loss[i] * scale;
    return;
  }
  float acc = 0.f;
  for (int i = threadIdx.x; i < Qg; i += 64) acc += row_loss[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) *loss = reduction == OM_CE_MEAN ? acc / (valid ? *valid : (float)Qg) * scale : acc * scale;   // all rows ignored: nan, as torch
}

// S[i,j] = <q_i, p_j> for a training batch's score matrix (8 x 64 per GPU, 64 x 512 with cross-device negatives): one wave per
// dot product.  The exact-f32 MFMA GEMM took 58 us for this one 128 x 128 tile (a k-ordered chain of 384 dependent MFMAs on one CU,
// profiles/r05_train_timeline_v1.txt); this is ~5 us.  f32 fmaf chains per lane + a wave reduction.
__global__ __launch_bounds__(256) void scores_small_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                            float* __restrict__ S, int Qg, int Pg, int d) {
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)Qg * Pg) return;
  const int i = (int)(w / Pg), j = (int)(w % Pg), lane = threadIdx.x & 63;
  const float4* a = (const float4*)(q + (int64_t)i * d);
  const float4* b = (const float4*)(p + (int64_t)j * d);
  float acc = 0.f;
  for (int c = lane; c < d / 4; c += 64) {
    const float4 x = a[c], y = b[c];
    acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) S[w] = acc;
}

// d_q[i,c] = sum_j dS[q_row0+i, j] * p[j,c]
__global__ void dq_kernel(const float* __restrict__ dS, const float* __restrict__ p,
                          float* __restrict__ dq, int Pg, int d, int q_row0) {
  const int i = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const float* ds = dS + (int64_t)(q_row0 + i) * Pg;
  float acc = 0.f;
  for (int j = 0; j < Pg; ++j) acc = fmaf(ds[j], p[(int64_t)j * d + c], acc);
  dq[(int64_t)i * d + c] = acc;
}
// d_p[j,c] = sum_i dS[i, p_row0+j] * q[i,c]
__global__ void dp_kernel(const float* __restrict__ dS, const float* __restrict__ q,
                          float* __restrict__ dp, int Qg, int Pg, int d, int p_row0) {
  const int j = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float acc = 0.f;
  for (int i = 0; i < Qg; ++i)
    acc = fmaf(dS[(int64_t)i * Pg + p_row0 + j], q[(int64_t)i * d + c], acc);
  dp[(int64_t)j * d + c] = acc;
}

extern "C" int om_contrastive_fwd_bwd_ex(const float* q, const float* p, int Qg, int Pg, int d, const int64_t* target,
                                         int n_psg, int reduction, const float* row_grad, float loss_scale, int q_row0,
                                         int q_rows, int p_row0, int p_rows, float* loss, float* scores, float* d_q,
                                         float* d_p, float* workspace, void* stream) {
  if (Qg <= 0 || Pg <= 0) OM_FAIL("empty batch");
  if (!target && (int64_t)(Qg - 1) * n_psg >= Pg) OM_FAIL("target index out of range (Pg < Qg * n_psg)");
  if (reduction < OM_CE_MEAN || reduction > OM_CE_NONE) OM_FAIL("reduction must be 0 (mean), 1 (sum) or 2 (none)");
  if (!workspace || !loss) OM_FAIL("null argument");
  if (q_row0 < 0 || q_row0 + q_rows > Qg || p_row0 < 0 || p_row0 + p_rows > Pg)
    OM_FAIL("local slice out of range");
  hipStream_t s = (hipStream_t)stream;
  // workspace: S [Qg,Pg] (if scores == NULL) | dS [Qg,Pg] | row_loss [Qg] | valid [1]
  float* S = scores ? scores : workspace;
  float* dS = workspace + (scores ? 0 : (size_t)Qg * Pg);
  float* row_loss = dS + (size_t)Qg * Pg;
  float* valid = row_loss + Qg;
  const bool bwd = d_q || d_p;
  if ((int64_t)Qg * Pg <= 65536 && d % 4 == 0 && !(((uintptr_t)q | (uintptr_t)p) & 15)) {
    hipLaunchKernelGGL(scores_small_kernel, dim3((unsigned)(((int64_t)Qg * Pg + 3) / 4)), dim3(256), 0, s, q, p, S, Qg, Pg, d);
    OM_LAUNCH_CHECK();
  } else if (om_gemm_nt(OM_F32, q, d, p, d, OM_F32, S, Pg, Qg, Pg, d, nullptr, nullptr, 0, OM_ACT_NONE, s)) {
    return 1;
  }
  if (target) {
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(64), 0, s, target, Qg, valid);
    OM_LAUNCH_CHECK();
  } else {
    valid = nullptr;                     // every row takes part: the kernels use Qg
  }
  hipLaunchKernelGGL(ce_rows_kernel, dim3((Qg + 3) / 4), dim3(256), 0, s, S, bwd ? dS : nullptr, Qg,
                     Pg, n_psg, target, reduction, loss_scale, row_grad, valid, row_loss);
  OM_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, s, row_loss, Qg, loss_scale, reduction, valid, loss);
  OM_LAUNCH_CHECK();
  if (d_q && q_rows > 0) {
    hipLaunchKernelGGL(dq_kernel, dim3((d + 255) / 256, q_rows), dim3(256), 0, s, dS, p, d_q, Pg, d,
                       q_row0);
    OM_LAUNCH_CHECK();
  }
  if (d_p && p_rows > 0) {
    hipLaunchKernelGGL(dp_kernel, dim3((d + 255) / 256, p_rows), dim3(256), 0, s, dS, q, d_p, Qg, Pg,
                       d, p_row0);
    OM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int om_contrastive_fwd_bwd(const float* q, const float* p, int Qg, int Pg, int d,
                                      int n_psg, float loss_scale, int q_row0, int q_rows,
                                      int p_row0, int p_rows, float* loss, float* scores,
                                      float* d_q, float* d_p, float* workspace, void* stream) {
  return om_contrastive_fwd_bwd_ex(q, p, Qg, Pg, d, nullptr, n_psg, OM_CE_MEAN, nullptr, loss_scale, q_row0, q_rows, p_row0,
                                   p_rows, loss, scores, d_q, d_p, workspace, stream);
}
