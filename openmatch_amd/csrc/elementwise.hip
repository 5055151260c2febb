// HBM-bound helpers of the encoder: embedding gather + LayerNorm (K1), LayerNorm / RMSNorm
// (tails of K4/K6), pooling (K7), L2 normalise (K9), T5 relative-position bias, and the
// f32 -> f16 index shadow copy.  One wavefront (64 lanes) owns one row; 4 rows per block.
#include "common.h"
#include "kernels.h"
#include "ln_row.h"

#define ROWS_PER_BLOCK 4
#define MAX_VEC_LIMIT 8  // 4-element vectors per lane -> H <= 2048 (NV = 4 covers H <= 1024)

// Normalise the row held in x[][] (nv vectors per lane) and write it out.
// LayerNorm: two-pass mean / biased variance in f32 (torch.nn.LayerNorm);
// RMSNorm (T5LayerNorm, HF:models/t5/modeling_t5.py:59-72): x * rsqrt(mean(x^2)+eps) * g.
template <typename TOut, int MAX_VEC>
__device__ inline void norm_and_store(float (&x)[MAX_VEC][4], int nvec, int lane, int H,
                                      const float* __restrict__ g, const float* __restrict__ b,
                                      float eps, int rms, TOut* __restrict__ out, float* __restrict__ out32 = nullptr) {
  float mean, rstd;
  ln_row_stats<MAX_VEC>(x, nvec, lane, H, eps, rms, mean, rstd);
#pragma unroll
  for (int j = 0; j < MAX_VEC; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (j < nvec && c < H) {
      float gv[4], y[4];
      Vec4<float>::load(g + c, gv);
      if (b) {
        float bv[4];
        Vec4<float>::load(b + c, bv);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = ln_affine(x[j][e], mean, rstd, gv[e], bv[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (x[j][e] - mean) * rstd * gv[e];
      }
      Vec4<TOut>::store(out + c, y);
      if (out32) Vec4<float>::store(out32 + c, y);      // the unrounded copy (16-bit training: what the next residual add reads)
    }
  }
}

template <typename TIn, typename TOut, int MAX_VEC>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layernorm_kernel(
    const TIn* __restrict__ x, int64_t ldx, TOut* __restrict__ y, int64_t ldy,
    const float* __restrict__ g, const float* __restrict__ b, int64_t M, int H, float eps, int rms,
    const int* __restrict__ rows = nullptr, float* __restrict__ y32 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const int64_t src = rows ? (int64_t)rows[row] : row;      // gather (packed rows: the [CLS] row of every sequence)
  const int nvec = (H / 4 + 63) / 64;
  float v[MAX_VEC][4];
#pragma unroll
  for (int j = 0; j < MAX_VEC; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (j < nvec && c < H) Vec4<TIn>::load(x + src * ldx + c, v[j]);
  }
  norm_and_store<TOut, MAX_VEC>(v, nvec, lane, H, g, b, eps, rms, y + row * ldy, y32 ? y32 + row * ldy : nullptr);
}

// bf16 rows with 16-byte accesses: 32 lanes own one row (two rows per wave), every lane holds NV
// vectors of 8 elements.  The 8-byte accesses of the generic kernel reach ~4.1 TB/s on
// [131072, 768]; 16-byte ones are what the memory path is built for (MI355X_MICROARCH.md).
template <int NV, typename TOut = bf16_t, typename TIn = bf16_t>      // TOut = float: the LAST normalisation before pooling (the reference's
__global__ __launch_bounds__(256) void layernorm_bf16x8_kernel(     // autocast runs layer_norm in fp32 and returns fp32); TIn: bf16_t or f16_t (round 6)
    const TIn* __restrict__ x, int64_t ldx, TOut* __restrict__ y, int64_t ldy,
    const float* __restrict__ g, const float* __restrict__ b, int64_t M, int H, float eps, int rms,
    const TIn* __restrict__ x_lo,             // x_lo: second plane of a two-plane residual stream (value = x + x_lo), or NULL
    const int* __restrict__ rows = nullptr,   // gather: output row r normalises input row rows[r]
    int lo8 = 0) {                            // 1: x_lo is the eight-bit plane (kernels.h omk_lo8_offset) of the [.., H] tensor x points into
  const int lane = threadIdx.x & 31;
  const int64_t orow = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (orow >= M) return;
  const int64_t row = rows ? (int64_t)rows[orow] : orow;
  const int nv8 = H / 8;
  float v[NV][8];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 32 * j;
    if (c < nv8) {
      const uint4 t = *(const uint4*)(x + row * ldx + c * 8);
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[j][2 * e] = Half16<TIn>::lo(w[e]); v[j][2 * e + 1] = Half16<TIn>::hi(w[e]); }
      if (x_lo && lo8) {       // the row's index in the [M, H] tensor the plane belongs to: row * ldx / H (ldx = L * H gathers the CLS rows)
        const int64_t trow = row * (ldx / H);
        const unsigned char* p8 = (const unsigned char*)x_lo;
        const uint32_t wa = *(const uint32_t*)(p8 + omk_lo8_offset(trow, c * 8, H)), wb = *(const uint32_t*)(p8 + omk_lo8_offset(trow, c * 8 + 4, H));
        const om_f32x2_t a01 = __builtin_amdgcn_cvt_pk_f32_bf8((int)wa, false), a23 = __builtin_amdgcn_cvt_pk_f32_bf8((int)wa, true);
        const om_f32x2_t b01 = __builtin_amdgcn_cvt_pk_f32_bf8((int)wb, false), b23 = __builtin_amdgcn_cvt_pk_f32_bf8((int)wb, true);
        const float l8[8] = {a01[0], a01[1], a23[0], a23[1], b01[0], b01[1], b23[0], b23[1]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = fmaf(l8[e], 0.0009765625f, v[j][e]);
      } else if (x_lo) {
        const uint4 t2 = *(const uint4*)(x_lo + row * ldx + c * 8);
        const uint32_t w2[4] = {t2.x, t2.y, t2.z, t2.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][2 * e] += Half16<TIn>::lo(w2[e]); v[j][2 * e + 1] += Half16<TIn>::hi(w2[e]); }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
  }
  auto half_sum = [](float t) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  float mean = 0.f;
  if (!rms) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[j][e];
    mean = half_sum(s) / (float)H;
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 32 * j < nv8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; ss += d * d; }
    }
  const float rstd = rsqrtf(half_sum(ss) / (float)H + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 32 * j;
    if (c < nv8) {
      float gv[8], bv[8];
      Vec4<float>::load(g + c * 8, *(float(*)[4])&gv[0]);
      Vec4<float>::load(g + c * 8 + 4, *(float(*)[4])&gv[4]);
      if (b) { Vec4<float>::load(b + c * 8, *(float(*)[4])&bv[0]); Vec4<float>::load(b + c * 8 + 4, *(float(*)[4])&bv[4]); }
      uint32_t w[4];
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y0 = (v[j][2 * e] - mean) * rstd * gv[2 * e], y1 = (v[j][2 * e + 1] - mean) * rstd * gv[2 * e + 1];
        if (b) { y0 += bv[2 * e]; y1 += bv[2 * e + 1]; }
        o[2 * e] = y0; o[2 * e + 1] = y1;
        w[e] = Half16<TIn>::pack2(y0, y1);
      }
      if (sizeof(TOut) == 4) {
        Vec4<float>::store((float*)y + orow * ldy + c * 8, *(float(*)[4])&o[0]);
        Vec4<float>::store((float*)y + orow * ldy + c * 8 + 4, *(float(*)[4])&o[4]);
      } else {
        *(uint4*)(y + orow * ldy + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  }
}

// Fold a LayerNorm into the bf16 weight that consumes its output (kernels.h, GemmEpilogue::ln_*):
//   W'[n,k] = bf16(W[n,k] gamma[k]),  s[n] = sum_k W'[n,k],  b'[n] = b[n] + sum_k beta[k] W[n,k]
// so that LN(y) W^T + b = rstd (y W'^T - mu s) + b'.  One wave per output row.
template <typename T>          // bf16_t or f16_t
__global__ __launch_bounds__(256) void ln_fold_kernel(const T* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ b,
                                                      T* __restrict__ Wf, float* __restrict__ colsum,
                                                      float* __restrict__ bf, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s = 0.f, t = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {              // K % 8 == 0
    const uint4 w4 = *(const uint4*)(W + (int64_t)n * K + k);
    const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float w0 = Half16<T>::lo(w[e]), w1 = Half16<T>::hi(w[e]);
      o[e] = Half16<T>::pack2(w0 * gamma[k + 2 * e], w1 * gamma[k + 2 * e + 1]);
      s += Half16<T>::lo(o[e]) + Half16<T>::hi(o[e]);        // the column sum of what the GEMM will actually multiply
      if (beta) t += w0 * beta[k + 2 * e] + w1 * beta[k + 2 * e + 1];
    }
    *(uint4*)(Wf + (int64_t)n * K + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  s = wave_sum(s); t = wave_sum(t);
  if (lane == 0) { colsum[n] = s; bf[n] = (b ? b[n] : 0.f) + t; }
}

int omk_ln_fold(int dtype, const void* W, const float* gamma, const float* beta, const float* b, void* Wf,
                float* colsum, float* bf, int N, int K, hipStream_t s) {
  if (K % 8 != 0) OM_FAIL("K must be a multiple of 8");
  if (dtype == OM_F16)
    hipLaunchKernelGGL((ln_fold_kernel<f16_t>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, (const f16_t*)W, gamma, beta, b,
                       (f16_t*)Wf, colsum, bf, N, K);
  else
    hipLaunchKernelGGL((ln_fold_kernel<bf16_t>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, (const bf16_t*)W, gamma, beta, b,
                       (bf16_t*)Wf, colsum, bf, N, K);
  OM_LAUNCH_CHECK();
  return 0;
}

// BERT: LN(word[id] + type[tt] + pos[t])  (HF:models/bert/modeling_bert.py:68-108).
// T5  : word[id]                          (shared embedding, no norm).
template <typename TOut, int MAX_VEC>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void embed_kernel(
    const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
    const float* __restrict__ g, const float* __restrict__ b, TOut* __restrict__ out, int64_t M,
    int L, int H, int vocab, int type_vocab, float eps, int bert, const int* __restrict__ row_map, float* __restrict__ out32) {
  const int lane = threadIdx.x & 63;
  const int64_t orow = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (orow >= M) return;
  int64_t row = orow;
  if (row_map) {                                   // packed rows: output row t embeds token row_map[t]; pad rows are zero
    row = row_map[orow];
    if (row < 0) {
      for (int c = lane * 4; c < H; c += 256) { float z[4] = {0.f, 0.f, 0.f, 0.f}; Vec4<TOut>::store(out + orow * H + c, z); }
      return;
    }
  }
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const int nvec = (H / 4 + 63) / 64;
  float v[MAX_VEC][4];
  if (bert) {
    int64_t tt = type_ids ? type_ids[row] : 0;
    tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
    const int t = (int)(row % L);
#pragma unroll
    for (int j = 0; j < MAX_VEC; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (j < nvec && c < H) {
        float w[4], ty[4], p[4];
        Vec4<float>::load(word + id * H + c, w);
        Vec4<float>::load(type + tt * H + c, ty);
        Vec4<float>::load(pos + (int64_t)t * H + c, p);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = (w[e] + ty[e]) + p[e];
      }
    }
    norm_and_store<TOut, MAX_VEC>(v, nvec, lane, H, g, b, eps, 0, out + orow * H, out32 ? out32 + orow * H : nullptr);
  } else {
#pragma unroll
    for (int j = 0; j < MAX_VEC; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (j < nvec && c < H) {
        Vec4<float>::load(word + id * H + c, v[j]);
        Vec4<TOut>::store(out + orow * H + c, v[j]);
      }
    }
  }
}

// pooling == FIRST: hidden[:,0,:] ; MEAN: sum(h*m)/clamp(sum(m),1e-9)  (utils.py:233-235)
template <typename T>
__global__ void pool_kernel(const T* __restrict__ x, const int64_t* __restrict__ mask,
                            float* __restrict__ out, int L, int H, int mode, const int* __restrict__ cu) {
  const int64_t b = blockIdx.x;
  const T* xb = x + (cu ? (int64_t)cu[b] : b * (int64_t)L) * H;
  const int Lb = cu ? cu[b + 1] - cu[b] : L;        // packed rows: this sequence's own length (the mask keeps pitch L)
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (mode == OM_POOL_FIRST) {
      Vec4<T>::load(xb + c, acc);
    } else {
      float cnt = 0.f;
      for (int t = 0; t < Lb; ++t) {
        const float m = (float)mask[b * L + t];
        float v[4];
        Vec4<T>::load(xb + (int64_t)t * H + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e] * m;
        cnt += m;
      }
      cnt = fmaxf(cnt, 1e-9f);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] / cnt;
    }
    Vec4<float>::store(out + b * H + c, acc);
  }
}

// F.normalize(x, dim=1): x / max(||x||_2, 1e-12)   (modeling/dense_retrieval_model.py:153-154)
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void l2norm_kernel(const float* __restrict__ x,
                                                                     float* __restrict__ y,
                                                                     int64_t M, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[row * D + c]; ss += v * v; }
  const float denom = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  for (int c = lane; c < D; c += 64) y[row * D + c] = x[row * D + c] / denom;
}

// T5 additive attention bias  bias[h][q][k] = table[bucket(k - q)][h]
// (HF:models/t5/modeling_t5.py compute_bias; the bucket LUT is built on the host).
__global__ void t5_bias_kernel(const float* __restrict__ table, const int* __restrict__ lut,
                               float* __restrict__ out, int L, int heads) {
  const int64_t n = (int64_t)heads * L * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % L), q = (int)((i / L) % L), h = (int)(i / ((int64_t)L * L));
    out[i] = table[lut[k - q + (L - 1)] * heads + h];
  }
}

// index.add(): f16 shadow rows + rounding statistics for the certified search margin.
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void index_to_f16_kernel(
    const float* __restrict__ x, f16_t* __restrict__ y, int64_t N, int d,
    unsigned* stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= N) return;
  float e2 = 0.f, n2 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = x[row * d + c];
    const f16_t r = (f16_t)v;             // round-to-nearest-even; |v| > 65504 -> inf -> margin inf
    const float rv = (float)r;
    y[row * d + c] = r;
    e2 += (v - rv) * (v - rv);
    n2 += rv * rv;
  }
  e2 = wave_sum(e2); n2 = wave_sum(n2);
  if (lane == 0) {
    // non-negative floats order like their bit patterns; round the norms UP a little
    const unsigned ue = __float_as_uint(sqrtf(e2) * 1.0001f), un = __float_as_uint(sqrtf(n2) * 1.0001f);
    if (ue > stats[0]) atomicMax(stats + 0, ue);     // monotone max: the plain read only skips
    if (un > stats[1]) atomicMax(stats + 1, un);     // atomics that cannot change the value
  }
}

// ---- host launchers ---------------------------------------------------------
template <typename TIn, typename TOut>
static int launch_ln(const void* x, int64_t ldx, void* y, int64_t ldy, const float* g,
                     const float* b, int64_t M, int H, float eps, int rms, hipStream_t s, const int* rows = nullptr) {
  const unsigned grid = (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  if (H <= 1024)
    hipLaunchKernelGGL((layernorm_kernel<TIn, TOut, 4>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, s,
                       (const TIn*)x, ldx, (TOut*)y, ldy, g, b, M, H, eps, rms, rows);
  else
    hipLaunchKernelGGL((layernorm_kernel<TIn, TOut, 8>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, s,
                       (const TIn*)x, ldx, (TOut*)y, ldy, g, b, M, H, eps, rms, rows);
  OM_LAUNCH_CHECK();
  return 0;
}

// Row statistics from the slot partials a GEMM epilogue left (GemmEpilogue::stats_out): one thread per row adds the
// slots in slot order -- the same result whatever order the tiles ran in (round 2 accumulated with f32 atomics).
__global__ __launch_bounds__(256) void ln_stats_reduce_kernel(const float2* __restrict__ slots, int nslots, int64_t M, float2* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < nslots; ++k) { const float2 v = slots[(int64_t)k * M + m]; s1 += v.x; s2 += v.y; }
  out[m] = make_float2(s1, s2);
}
int omk_ln_stats_reduce(const float* slots, int nslots, int64_t M, float* out, hipStream_t s) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(ln_stats_reduce_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, (const float2*)slots, nslots, M, (float2*)out);
  OM_LAUNCH_CHECK();
  return 0;
}

// x in the compute format (optionally two planes, bf16) -> y in f32
int omk_layernorm_f32out(int dtype, const void* x, int64_t ldx, float* y, int64_t ldy, const float* g, const float* b,
                         int64_t M, int H, float eps, int rms, hipStream_t s, const void* x_lo, const int* rows, int lo8) {
  if (lo8 && !(x_lo && dtype == OM_F16 && H % 256 == 0 && ldx % H == 0)) OM_FAIL("eight-bit second plane: float16 rows of a whole-tile [M, H] tensor");
  if (H % 4 != 0 || H > 64 * 4 * MAX_VEC_LIMIT) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
  if (M <= 0) return 0;
  if ((dtype == OM_BF16 || (dtype == OM_F16 && x_lo)) && H % 8 == 0 && H <= 1024 && ldx % 8 == 0 && ldy % 4 == 0 &&
      !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)x_lo) & 15) && !(((uintptr_t)g | (uintptr_t)b) & 15)) {
    const unsigned grid = (unsigned)((M + 7) / 8);
    const int nv = (H / 8 + 31) / 32;
#define LN8F_(NV, TI) hipLaunchKernelGGL((layernorm_bf16x8_kernel<NV, float, TI>), dim3(grid), dim3(256), 0, s, (const TI*)x, ldx, \
                                    y, ldy, g, b, M, H, eps, rms, (const TI*)x_lo, rows, lo8)
#define LN8F(NV) do { if (dtype == OM_BF16) LN8F_(NV, bf16_t); else LN8F_(NV, f16_t); } while (0)
    if (nv <= 1) LN8F(1); else if (nv == 2) LN8F(2); else if (nv == 3) LN8F(3); else LN8F(4);
#undef LN8F
#undef LN8F_
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (x_lo) OM_FAIL("two-plane LayerNorm input: 16-bit rows of whole 16-byte vectors only");
  if (dtype == OM_BF16) return launch_ln<bf16_t, float>(x, ldx, y, ldy, g, b, M, H, eps, rms, s, rows);
  if (dtype == OM_F16) return launch_ln<f16_t, float>(x, ldx, y, ldy, g, b, M, H, eps, rms, s, rows);
  return launch_ln<float, float>(x, ldx, y, ldy, g, b, M, H, eps, rms, s, rows);
}

// packed rows: more tokens than the caller's row bound -> every representation becomes NaN (never a silently truncated batch)
__global__ void pack_overflow_poison_kernel(const int* __restrict__ cu, int64_t B, int64_t rows, float* __restrict__ out, int64_t n) {
  if ((int64_t)cu[B + 1] <= rows) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = __int_as_float(0x7fc00000);
}
int omk_pack_overflow_poison(const int* cu, int64_t B, int64_t rows, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pack_overflow_poison_kernel, dim3(64), dim3(256), 0, s, cu, B, rows, out, n);
  OM_LAUNCH_CHECK();
  return 0;
}

// f32 rows in -> the normalised rows in `dtype` (the next contraction's operand) AND in f32 (y32: what the next residual add reads):
// the LayerNorm of the 16-bit training forward, whose residual stream stays in f32 as the reference's autocast keeps it
int omk_layernorm_dual(int dtype, const float* x, int64_t ldx, void* y, float* y32, int64_t ldy, const float* g, const float* b,
                       int64_t M, int H, float eps, hipStream_t s) {
  if (H % 4 != 0 || H > 64 * 4 * MAX_VEC_LIMIT) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
  if (M <= 0) return 0;
  const unsigned grid = (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
#define LND(TO, MV) hipLaunchKernelGGL((layernorm_kernel<float, TO, MV>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, s, x, ldx, (TO*)y, ldy, g, b, M, \
                                       H, eps, 0, (const int*)nullptr, y32)
  if (dtype == OM_BF16) { if (H <= 1024) LND(bf16_t, 4); else LND(bf16_t, 8); }
  else if (dtype == OM_F16) { if (H <= 1024) LND(f16_t, 4); else LND(f16_t, 8); }
  else OM_FAIL("a 16-bit output format");
#undef LND
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* g,
                  const float* b, int64_t M, int H, float eps, int rms, hipStream_t s, const void* x_lo, int lo8) {
  if (lo8 && !(x_lo && dtype == OM_F16 && H % 256 == 0 && ldx % H == 0)) OM_FAIL("eight-bit second plane: float16 rows of a whole-tile [M, H] tensor");
  if (H % 4 != 0 || H > 64 * 4 * MAX_VEC_LIMIT) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
  if (M <= 0) return 0;
  if (x_lo && !((dtype == OM_BF16 || dtype == OM_F16) && H % 8 == 0 && H <= 1024 && ldx % 8 == 0 && ldy % 8 == 0 &&
                !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)x_lo) & 15) && !(((uintptr_t)g | (uintptr_t)b) & 15)))
    OM_FAIL("two-plane LayerNorm input: 16-bit rows of whole 16-byte vectors only");
  if ((dtype == OM_BF16 || (dtype == OM_F16 && x_lo)) && H % 8 == 0 && H <= 1024 && ldx % 8 == 0 && ldy % 8 == 0 &&
      !(((uintptr_t)x | (uintptr_t)y) & 15) && !(((uintptr_t)g | (uintptr_t)b) & 15)) {
    const unsigned grid = (unsigned)((M + 7) / 8);
    const int nv = (H / 8 + 31) / 32;
#define LN8_(NV, TI) hipLaunchKernelGGL((layernorm_bf16x8_kernel<NV, TI, TI>), dim3(grid), dim3(256), 0, s, (const TI*)x, ldx, \
                                   (TI*)y, ldy, g, b, M, H, eps, rms, (const TI*)x_lo, (const int*)nullptr, lo8)
#define LN8(NV) do { if (dtype == OM_BF16) LN8_(NV, bf16_t); else LN8_(NV, f16_t); } while (0)
    if (nv <= 1) LN8(1); else if (nv == 2) LN8(2); else if (nv == 3) LN8(3); else LN8(4);
#undef LN8
#undef LN8_
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == OM_BF16) return launch_ln<bf16_t, bf16_t>(x, ldx, y, ldy, g, b, M, H, eps, rms, s);
  if (dtype == OM_F16) return launch_ln<f16_t, f16_t>(x, ldx, y, ldy, g, b, M, H, eps, rms, s);
  return launch_ln<float, float>(x, ldx, y, ldy, g, b, M, H, eps, rms, s);
}

int omk_embed(int dtype, const int64_t* ids, const int64_t* type_ids, const float* word,
              const float* pos, const float* type, const float* g, const float* b, void* out,
              int64_t M, int L, int H, int vocab, int type_vocab, float eps, int bert,
              hipStream_t s, const int* row_map, float* out32) {
  if (H % 4 != 0 || H > 64 * 4 * MAX_VEC_LIMIT) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
  if (M <= 0) return 0;
  if (out32 && (!bert || row_map)) OM_FAIL("embedding: the f32 copy goes with the BERT LayerNorm on unpacked rows");
  const unsigned grid = (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
#define EMBED_LAUNCH(TT, NV)                                                                   \
  hipLaunchKernelGGL((embed_kernel<TT, NV>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, s, ids, \
                     type_ids, word, pos, type, g, b, (TT*)out, M, L, H, vocab, type_vocab, eps, bert, row_map, out32)
  if (dtype == OM_BF16) {
    if (H <= 1024) EMBED_LAUNCH(bf16_t, 4); else EMBED_LAUNCH(bf16_t, 8);
  } else if (dtype == OM_F16) {
    if (H <= 1024) EMBED_LAUNCH(f16_t, 4); else EMBED_LAUNCH(f16_t, 8);
  } else {
    if (H <= 1024) EMBED_LAUNCH(float, 4); else EMBED_LAUNCH(float, 8);
  }
#undef EMBED_LAUNCH
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_pool(int dtype, const void* x, const int64_t* mask, float* out, int64_t B, int L, int H,
             int mode, hipStream_t s, const int* cu) {
  if (B <= 0) return 0;
  if (H % 4 != 0) OM_FAIL("hidden size must be a multiple of 4");
  if (dtype == OM_BF16)
    hipLaunchKernelGGL((pool_kernel<bf16_t>), dim3((unsigned)B), dim3(256), 0, s, (const bf16_t*)x,
                       mask, out, L, H, mode, cu);
  else if (dtype == OM_F16)
    hipLaunchKernelGGL((pool_kernel<f16_t>), dim3((unsigned)B), dim3(256), 0, s, (const f16_t*)x,
                       mask, out, L, H, mode, cu);
  else
    hipLaunchKernelGGL((pool_kernel<float>), dim3((unsigned)B), dim3(256), 0, s, (const float*)x,
                       mask, out, L, H, mode, cu);
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_l2norm(const float* x, float* y, int64_t M, int D, hipStream_t s) {
  if (M <= 0) return 0;
  const unsigned grid = (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  hipLaunchKernelGGL(l2norm_kernel, dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, s, x, y, M, D);
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_t5_bias(const float* table, const int* lut, float* out, int L, int heads, hipStream_t s) {
  hipLaunchKernelGGL(t5_bias_kernel, dim3(256), dim3(256), 0, s, table, lut, out, L, heads);
  OM_LAUNCH_CHECK();
  return 0;
}

extern "C" int om_index_to_f16(const float* rows_f32, int64_t N, int d, void* rows_f16,
                               float* stats, void* stream) {
  if (N <= 0) return 0;
  const unsigned grid = (unsigned)((N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  hipLaunchKernelGGL(index_to_f16_kernel, dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0,
                     (hipStream_t)stream, rows_f32, (f16_t*)rows_f16, N, d, (unsigned*)stats);
  OM_LAUNCH_CHECK();
  return 0;
}

// ---- self-check of ln_row.h's row reduction against the __shfl_xor butterfly (om_debug_wave_sum_check) ----
__global__ __launch_bounds__(256) void wave_sum_check_kernel(const float* __restrict__ in, float* __restrict__ a, float* __restrict__ b, int64_t groups) {
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= groups) return;
  const float v = in[g * 64 + (threadIdx.x & 63)];
  const float s0 = wave_sum(v);
  float s1[1] = {v};
  ln_wave_sums<1>(s1);
  if ((threadIdx.x & 63) == (int)(g % 64)) { a[g] = s0; b[g] = s1[0]; }      // a different lane reports for every group: all lanes hold the sum
}
extern "C" int om_debug_wave_sum_check(const float* in, float* out_shuffle, float* out_dpp, int64_t groups, void* stream) {
  if (!in || !out_shuffle || !out_dpp || groups <= 0) OM_FAIL("om_debug_wave_sum_check: null argument");
  hipLaunchKernelGGL(wave_sum_check_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out_shuffle, out_dpp, groups);
  OM_LAUNCH_CHECK();
  return 0;
}
