// Backward-pass kernels of the BERT encoder (training path, K1..K9 reversed):
// operand transposes for the weight-gradient GEMMs, bias column sums, dropout masks,
// LayerNorm / embedding / pooling / normalise backward, and the fused attention backward.
// The dense contractions themselves (dgrad, wgrad) run on the MFMA GEMM of gemm.hip.
#include "attn_common.h"
#include "train_kernels.h"

// ---------------------------------------------------------------------------------------
// out[c][r] = op(in[r][c]),  r < R (rows R..Rp-1 of the output pitch are zero-filled so the
// result can be the K-padded operand of an NT GEMM).  OP 1 = erf-GELU (recompute of the FFN
// activation instead of saving it).
template <typename T, int OP>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t ldi,
                                                        int64_t R, int C, T* __restrict__ out,
                                                        int64_t ldo, int64_t Rp) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t r = r0 + ty + 4 * i;
    const int c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      v = ElemOps<T>::load(in + r * ldi + c);
      if (OP == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i;
    const int64_t r = r0 + tx;
    if (c < C && r < Rp) ElemOps<T>::store(out + (int64_t)c * ldo + r, tile[tx][ty + 4 * i]);
  }
}

int omk_transpose(int dtype, const void* in, int64_t ldi, int64_t R, int C, void* out, int64_t ldo,
                  int64_t Rp, int op, hipStream_t s) {
  if (R <= 0 || C <= 0) return 0;
  dim3 grid((unsigned)((Rp + 63) / 64), (unsigned)((C + 63) / 64));
#define TR(TT, OPV) hipLaunchKernelGGL((transpose_kernel<TT, OPV>), grid, dim3(256), 0, s, (const TT*)in, ldi, R, C, (TT*)out, ldo, Rp)
  if (dtype == OM_BF16) { if (op) TR(bf16_t, 1); else TR(bf16_t, 0); }
  else if (dtype == OM_F16) { if (op) TR(f16_t, 1); else TR(f16_t, 0); }
  else { if (op) TR(float, 1); else TR(float, 0); }
#undef TR
  OM_LAUNCH_CHECK();
  return 0;
}

// Many small transposes in one launch (every weight matrix of the encoder, once per backward: the dgrad GEMMs read W^T).
// 48 separate launches of ~10 us each were 0.5 ms of a 15 ms training step (profiles/r02_train_kernel_stats_v3.csv).
struct TransposeJob { const void* in; void* out; int R, C; int tile0; };     // out[c][r] = in[r][c], both dense
struct TransposeBatch { TransposeJob job[40]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch b) {
  __shared__ float tile[64][65];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.job[j + 1].tile0) ++j;
  const TransposeJob& jb = b.job[j];
  const int t = blockIdx.x - jb.tile0, tc = (jb.C + 63) / 64;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const T* in = (const T*)jb.in;
  T* out = (T*)jb.out;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < jb.R && c < jb.C) ? ElemOps<T>::load(in + (int64_t)r * jb.C + c) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < jb.C && r < jb.R) ElemOps<T>::store(out + (int64_t)c * jb.R + r, tile[tx][ty + 4 * i]);
  }
}

int omk_transpose_batch(int dtype, const void* const* in, void* const* out, const int* R, const int* C, int n, hipStream_t s) {
  for (int at = 0; at < n;) {
    TransposeBatch b;
    int tiles = 0;
    b.n = 0;
    while (at < n && b.n < 40) {
      b.job[b.n] = TransposeJob{in[at], out[at], R[at], C[at], tiles};
      tiles += ((R[at] + 63) / 64) * ((C[at] + 63) / 64);
      ++b.n; ++at;
    }
    if (!tiles) continue;
    if (dtype == OM_BF16) hipLaunchKernelGGL((transpose_batch_kernel<bf16_t>), dim3((unsigned)tiles), dim3(256), 0, s, b);
    else if (dtype == OM_F16) hipLaunchKernelGGL((transpose_batch_kernel<f16_t>), dim3((unsigned)tiles), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((transpose_batch_kernel<float>), dim3((unsigned)tiles), dim3(256), 0, s, b);
    OM_LAUNCH_CHECK();
  }
  return 0;
}

// out[c] += sum_r x[r][c]      (out is f32 and zero-initialised by the caller)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ld, int64_t M,
                                                     int N, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = (r0 + rows_per) < M ? (r0 + rows_per) : M;
  float acc = 0.f;
  if (c < N)
    for (int64_t r = r0 + w; r < r1; r += 4) acc += ElemOps<T>::load(x + r * ld + c);
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < N) atomicAdd(out + c, (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}

int omk_colsum(int dtype, const void* x, int64_t ld, int64_t M, int N, float* out, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 255) / 256 > 64 ? 64 : (M + 255) / 256));
  if (dtype == OM_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, ld, M, N, out);
  else if (dtype == OM_F16) hipLaunchKernelGGL((colsum_kernel<f16_t>), grid, dim3(256), 0, s, (const f16_t*)x, ld, M, N, out);
  else hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)x, ld, M, N, out);
  OM_LAUNCH_CHECK();
  return 0;
}

// y[i] = keep(seed, i) ? x[i] / (1 - p) : 0     (forward and backward of inverted dropout)
// rows (packed rows; with the row width H): element i = (row r, column c) is keyed as (rows[r], c) -- the token the row holds --
// so that the packed and the padded step of one batch draw the same mask
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p,
                               uint64_t seed, const int* __restrict__ rows, int H) {
  const DropCfg dc(p);
  const uint32_t thresh = dc.thresh;
  const float scale = dc.keep_scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t key = (uint64_t)i;
    if (rows) { const int64_t r = i / H; key = (uint64_t)((int64_t)rows[r] * H + (i - r * H)); }
    ElemOps<T>::store(y + i, dropout_keep(seed, key, thresh) ? ElemOps<T>::load(x + i) * scale : 0.f);
  }
}

int omk_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, hipStream_t s, const int* rows, int H) {
  if (n <= 0) return 0;
  if (rows && (H <= 0 || n % H)) OM_FAIL("dropout with a row map: n must be rows * H");
  const unsigned grid = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  if (dtype == OM_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n, p, seed, rows, H);
  else if (dtype == OM_F16) hipLaunchKernelGGL((dropout_kernel<f16_t>), dim3(grid), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, n, p, seed, rows, H);
  else hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, n, p, seed, rows, H);
  OM_LAUNCH_CHECK();
  return 0;
}

// four consecutive elements as one 8-byte (bf16) / 16-byte (f32) access: p must be that aligned
template <typename T> __device__ __forceinline__ void unpack4(const uint2& w, float (&v)[4]);      // four 16-bit values of a loaded dword pair
template <> __device__ __forceinline__ void unpack4<bf16_t>(const uint2& w, float (&v)[4]) {
  v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
  v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack4<f16_t>(const uint2& w, float (&v)[4]) {
  v[0] = Half16<f16_t>::lo(w.x); v[1] = Half16<f16_t>::hi(w.x); v[2] = Half16<f16_t>::lo(w.y); v[3] = Half16<f16_t>::hi(w.y);
}
template <> __device__ __forceinline__ void unpack4<float>(const uint2&, float (&)[4]) {}          // (never used: PF is a 16-bit path)
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) { unpack4<bf16_t>(*(const uint2*)p, v); }
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float (&v)[4]) { unpack4<f16_t>(*(const uint2*)p, v); }
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const float4 w = *(const float4*)p;
  v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  *(uint2*)p = make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                          (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float (&v)[4]) {
  *(uint2*)p = make_uint2(Half16<f16_t>::pack2(v[0], v[1]), Half16<f16_t>::pack2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------
// LayerNorm backward.  One wavefront per row, 4 rows per block pass, grid-stride over rows;
// d_gamma / d_beta are accumulated per lane in registers and flushed once per block.
//   xhat = (x - mean) * rstd ; dxhat = dy * g
//   dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
// MODE 0: x is read from memory (the saved pre-LN sum).
// MODE 1: x = word[id] + type[tt] + pos[t] is recomputed and dx is scattered into the three
//         embedding-table gradients (BERT embeddings backward).
// Eight waves per block and at most one block per CU: every block ends with one f32 atomic per column into d_gamma /
// d_beta, and 1024 blocks adding into the same 96 cache lines serialised at the memory side (41 us per call at
// 9216 x 768, profiles/r02_train_kernel_stats_v3.csv) -- the block count, not the bytes, set the time.
// (Four waves for hidden sizes above 1024: the wider per-lane state needs the 256-register budget.)
// PF (MODE 0, 16-bit T, x32 given, no dy32): the NEXT row's x32 / dy are loaded (raw) while this row is reduced -- two rows of loads in
// flight per wave; with the 128-register budget that holds two blocks on a CU (round 5: 130 registers had left room for one block of
// eight waves, eight rows in flight per CU: 33 us per call at 9 216 x 768 where the bytes need 13)
template <typename T, int NV, int MODE, bool PF = false>
__global__ __launch_bounds__((NV == 8 || MODE == 1) ? 256 : 512, (NV == 8 || MODE == 1) ? 1 : (NV == 3 ? 4 : 3)) void ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ g,
    T* __restrict__ dx, float* __restrict__ dg, float* __restrict__ db, int64_t M, int H, float eps,
    const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
    float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype_, int L, int vocab,
    int type_vocab, int rms, const T* __restrict__ add, T* __restrict__ dx_drop, float drop_p, uint64_t drop_seed,
    const float* __restrict__ dy32, const float* __restrict__ x32, float* __restrict__ partial, const int* __restrict__ cu) {
  // cu != NULL (MODE 1, packed rows): dy holds sequence b's rows at cu[b] .. cu[b + 1] - 1; ids / type_ids keep the [B, L] layout
  // partial != NULL (MODE 0): the block's column sums of d_gamma / d_beta go to partial[blockIdx.x][2][H] as plain stores and
  // omk_ln_param_reduce adds them up in block order -- instead of gridDim.x same-address atomics per column, which cost ~7 us of a
  // 27 us call at 9 216 x 768 and made the sums depend on the arrival order (round 5)
  // dy32 / x32 != NULL (MODE 0): the incoming gradient / the normalisation's input are read from these f32 tensors instead of dy / x
  // (round 5: the gradient of the pooled rows enters the last LayerNorm unrounded; the pre-LayerNorm sums of the 16-bit training
  // forward are kept in f32 -- tools/emulate_train_dataflow.py)
  constexpr int LNB_WAVES = (NV == 8 || MODE == 1) ? 4 : 8;
  // dx_drop != NULL: also writes dropout(dx) with the forward's mask (seed, element index) -- the gradient entering the
  // dense layer in front of the residual add -- so that no separate dropout pass re-reads dx.
  // rms != 0: T5LayerNorm (no mean, no bias): xhat = x * rstd, rstd = rsqrt(mean(x^2) + eps).
  // add != NULL: dx = (this backward) + add  (the residual stream's gradient, pre-norm stacks).
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;                       // [2][LNB_WAVES][H]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float gacc[NV][4], bacc[NV][4], gv[NV][4];
  // MODE 1: every token adds into one of (usually) two token-type rows -- thousands of atomics per
  // address if done per token (528 us per call at 8448 tokens).  Types 0 and 1 are summed in registers
  // and flushed once per block; other type ids keep the per-token atomic.
  float tacc[MODE == 1 ? 2 : 1][NV][4];
  // MODE 1, position table: a block takes ONE position t (all sequences, or a share of them), sums d x over its rows in
  // registers and adds the total once -- instead of one atomic per token and element into 128 x H addresses
  // (7 M contended atomics, 283 us per call at 72 x 128 tokens: profiles/r02_train_kernel_stats_v4.csv).
  float pacc[MODE == 1 ? NV : 1][4];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (lane + 64 * j) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gacc[j][e] = 0.f; bacc[j][e] = 0.f; gv[j][e] = (c + e < H) ? g[c + e] : 0.f;
      tacc[0][j][e] = 0.f;
      if (MODE == 1) { tacc[MODE == 1 ? 1 : 0][j][e] = 0.f; pacc[MODE == 1 ? j : 0][e] = 0.f; }
    }
  }
  // rows of this wave: MODE 0 grid-stride over all rows; MODE 1 rows b * L + t of the block's position t
  int64_t row_first = (int64_t)blockIdx.x * LNB_WAVES + w, row_step = (int64_t)gridDim.x * LNB_WAVES, row_end = M;
  int my_t = 0;
  if (MODE == 1) {
    const int parts = (int)gridDim.x / L > 0 ? (int)gridDim.x / L : 1;        // blocks per position (grid >= L is the launcher's job)
    my_t = blockIdx.x % L;
    const int part = blockIdx.x / L;
    row_first = ((int64_t)part * LNB_WAVES + w) * L + my_t;
    row_step = (int64_t)parts * LNB_WAVES * L;
    if (part >= parts) row_end = 0;
  }
  float4 rx[PF ? NV : 1];
  uint2 rd[PF ? NV : 1];
#define LNB_FETCH(R_)                                                                   \
  _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                      \
    const int c = (lane + 64 * j) * 4;                                                  \
    if (c < H) {                                                                        \
      rx[PF ? j : 0] = *(const float4*)(x32 + (R_) * H + c);                            \
      rd[PF ? j : 0] = *(const uint2*)(dy + (R_) * H + c);                              \
    }                                                                                   \
  }
  if (PF && row_first < row_end) { LNB_FETCH(row_first) }
  for (int64_t row = row_first; row < row_end; row += row_step) {
    float xv[NV][4], dv[NV][4];
    int64_t id = 0, tt = 0;
    int t = 0;
    int64_t drow = row;                      // the row of dy
    if (MODE == 1 && cu) {
      const int64_t sq = row / L;
      const int c0 = cu[sq], c1 = cu[sq + 1];
      if (my_t >= c1 - c0) continue;         // (wave-uniform) this sequence has no row at position my_t
      drow = c0 + my_t;
    }
    if (MODE == 1) {
      id = ids[row]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
      tt = type_ids ? type_ids[row] : 0; tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
      t = (int)(row % L);
    }
    float s1 = 0.f;
    if (PF) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if ((lane + 64 * j) * 4 < H) {
          const float4 q = rx[PF ? j : 0];
          xv[j][0] = q.x; xv[j][1] = q.y; xv[j][2] = q.z; xv[j][3] = q.w;
          unpack4<T>(rd[PF ? j : 0], dv[j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) s1 += xv[j][e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { xv[j][e] = 0.f; dv[j][e] = 0.f; }
        }
      }
      const int64_t nr = row + row_step;
      if (nr < row_end) { LNB_FETCH(nr) }
    } else
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < H) {
        if (MODE == 0) {
          if (x32) load4<float>(x32 + row * H + c, xv[j]); else load4<T>(x + row * H + c, xv[j]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[j][e] = (word[id * H + c + e] + type[tt * H + c + e]) + pos[(int64_t)t * H + c + e];
        }
        if (MODE == 0 && dy32) load4<float>(dy32 + row * H + c, dv[j]); else load4<T>(dy + drow * H + c, dv[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s1 += xv[j][e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[j][e] = 0.f; dv[j][e] = 0.f; }
      }
    }
    const float mean = rms ? 0.f : wave_sum(s1) / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if ((lane + 64 * j) * 4 < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[j][e] - mean; s2 += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(s2) / (float)H + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if ((lane + 64 * j) * 4 < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[j][e] - mean) * rstd;
          const float dxh = dv[j][e] * gv[j][e];
          xv[j][e] = xh;
          m1 += dxh; m2 += dxh * xh;
          gacc[j][e] += dv[j][e] * xh;
          bacc[j][e] += dv[j][e];
        }
      }
    m1 = rms ? 0.f : wave_sum(m1) / (float)H; m2 = wave_sum(m2) / (float)H;
    const DropCfg dc((MODE == 0 && dx_drop) ? drop_p : 0.f);
    const uint32_t drop_thresh = dc.thresh;
    const float drop_scale = dc.keep_scale;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < H) {
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = rstd * (dv[j][e] * gv[j][e] - m1 - xv[j][e] * m2);
        if (MODE == 0) {
          if (add) {
            float av[4];
            load4<T>(add + row * H + c, av);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] += av[e];
          }
          store4<T>(dx + row * H + c, out);
          if (dx_drop) {               // dropout of the value as stored (rounded to T), like the separate pass it replaces
            float dr[4];
            // MODE 0: `cu` carries the packed rows' token map; a wave works on ONE row, so the look-up is a scalar load (a per-lane
            // load cost the three registers this 128-register kernel does not have)
            int krow = __builtin_amdgcn_readfirstlane((int)row);
            if (MODE == 0 && cu) krow = cu[krow];
            const uint64_t bits = dropout_bits(drop_seed, (uint64_t)((int64_t)krow * H + c) >> 2);  // H % 4 == 0: one group
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float vb = sizeof(T) == 2 ? Half16<T>::value(Half16<T>::bits(out[e])) : out[e];
              dr[e] = dropout_field(bits, e, drop_thresh) ? vb * drop_scale : 0.f;
            }
            store4<T>(dx_drop + row * H + c, dr);
          }
        } else {
          // the word-table scatter leaves through a wave-private LDS row so that every atomic instruction covers 64
          // CONSECUTIVE columns (the registers hold 4 consecutive columns per lane: 16 cache lines per instruction)
          *(float4*)(red + w * H + c) = make_float4(out[0], out[1], out[2], out[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = out[e];
            if (tt == 0) tacc[0][j][e] += v;
            else if (tt == 1) tacc[MODE == 1 ? 1 : 0][j][e] += v;
            else atomicAdd(dtype_ + tt * H + c + e, v);
            pacc[MODE == 1 ? j : 0][e] += v;
          }
        }
      }
    }
    if (MODE == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations execute in order)
      for (int cc = lane; cc < H; cc += 64) atomicAdd(dword + id * H + cc, red[w * H + cc]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // read before the next row overwrites it
    }
  }
  if (MODE == 1) __syncthreads();          // `red` is the block-reduction buffer from here on
  // block reduction of the parameter gradients
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (c < H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(0 * LNB_WAVES + w) * H + c + e] = gacc[j][e]; red[(1 * LNB_WAVES + w) * H + c + e] = bacc[j][e]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 64 * LNB_WAVES) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < LNB_WAVES; ++k) { sg += red[k * H + c]; sb += red[(LNB_WAVES + k) * H + c]; }
    if (MODE == 0 && partial) {
      partial[((size_t)blockIdx.x * 2 + 0) * H + c] = sg;
      partial[((size_t)blockIdx.x * 2 + 1) * H + c] = sb;
    } else {
      atomicAdd(dg + c, sg);
      if (db) atomicAdd(db + c, sb);
    }
  }
  if (MODE == 1) {                     // token-type rows 0 and 1: same block reduction, then one atomic per column
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(0 * LNB_WAVES + w) * H + c + e] = tacc[0][j][e]; red[(1 * LNB_WAVES + w) * H + c + e] = tacc[MODE == 1 ? 1 : 0][j][e]; }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 64 * LNB_WAVES) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < LNB_WAVES; ++k) { s0 += red[k * H + c]; s1 += red[(LNB_WAVES + k) * H + c]; }
      atomicAdd(dtype_ + c, s0);
      if (type_vocab > 1) atomicAdd(dtype_ + H + c, s1);
    }
    __syncthreads();                   // the block's position row
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[w * H + c + e] = pacc[MODE == 1 ? j : 0][e];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 64 * LNB_WAVES) {
      float s0 = 0.f;
#pragma unroll
      for (int k = 0; k < LNB_WAVES; ++k) s0 += red[k * H + c];
      atomicAdd(dpos + (int64_t)my_t * H + c, s0);
    }
  }
}

template <typename T, int MODE>
static int launch_ln_bwd(const void* dy, const void* x, const float* g, void* dx, float* dg, float* db,
                         int64_t M, int H, float eps, const int64_t* ids, const int64_t* tt,
                         const float* word, const float* pos, const float* type, float* dword,
                         float* dpos, float* dtype_, int L, int vocab, int type_vocab, hipStream_t s,
                         int rms = 0, const void* add = nullptr, void* dx_drop = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0,
                         const float* dy32 = nullptr, const float* x32 = nullptr, float* partial = nullptr, int* partial_blocks = nullptr,
                         const int* cu = nullptr) {
  const int waves = (H <= 1024 && MODE == 0) ? 8 : 4;
  const int64_t want = (M + waves - 1) / waves;
  // two blocks per CU: ~12 MB of loads in flight (one row per wave at a time), what ~5 TB/s x ~2 us of latency needs;
  // more blocks only add same-address atomics on d_gamma / d_beta (1024 blocks: 41 us, 256: 30 us per call)
  unsigned grid = (unsigned)(want > OM_LNB_MAX_BLOCKS ? OM_LNB_MAX_BLOCKS : want);
  if (partial_blocks) *partial_blocks = (int)grid;
  if (MODE == 1) {                     // one position per block, 256 / L blocks per position (kernel comment)
    if (L < 1 || M % L) OM_FAIL("embedding backward: M must be B * L");
    const int parts = 256 / L > 0 ? 256 / L : 1;
    grid = (unsigned)(L * parts);
  }
  const size_t lds = (size_t)2 * waves * H * sizeof(float);       // <= 64 KiB for both shapes
#define LNB_(NV, PF_) hipLaunchKernelGGL((ln_bwd_kernel<T, NV, MODE, PF_>), dim3(grid), dim3(64 * waves), lds, s, (const T*)dy, (const T*)x, g, (T*)dx, dg, db, M, H, eps, ids, tt, word, pos, type, dword, dpos, dtype_, L, vocab, type_vocab, rms, (const T*)add, (T*)dx_drop, drop_p, drop_seed, dy32, x32, partial, cu)
#define LNB(NV) LNB_(NV, false)
  constexpr bool can_pf = MODE == 0 && sizeof(T) == 2;
  if (can_pf && x32 && !dy32 && H <= 768 && (om_option(OM_OPT_TRAIN_WGRAD_STREAM) & 8) == 0) {        // (bit 3, A/B: no prefetch;
    LNB_(3, can_pf);                                                                                   //  wider rows: the prefetch spills)
  } else if (H <= 768 && MODE == 0) LNB(3);
  else if (H <= 1024) LNB(4);
  else LNB(8);
#undef LNB
#undef LNB_
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_ln_bwd(int dtype, const void* dy, const void* x, const float* g, void* dx, float* dg,
               float* db, int64_t M, int H, float eps, hipStream_t s) {
  return omk_norm_bwd(dtype, dy, x, g, dx, dg, db, M, H, eps, 0, nullptr, s);
}

int omk_ln_bwd_drop(int dtype, const void* dy, const void* x, const float* g, void* dx, void* dx_drop, float drop_p,
                    uint64_t drop_seed, float* dg, float* db, int64_t M, int H, float eps, hipStream_t s, const float* dy32,
                    const float* x32, float* partial, int* partial_blocks, const int* drop_rows) {
  if (M <= 0) { if (partial_blocks) *partial_blocks = 0; return 0; }
  if (H % 4 || H > 2048) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
  if (drop_p <= 0.f) dx_drop = nullptr;
#define OM_LNBD(TT) return launch_ln_bwd<TT, 0>(dy, x, g, dx, dg, db, M, H, eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 1, s, 0, nullptr, dx_drop, drop_p, drop_seed, dy32, x32, partial, partial_blocks, drop_rows)
  if (dtype == OM_BF16) OM_LNBD(bf16_t);
  if (dtype == OM_F16) OM_LNBD(f16_t);
  OM_LNBD(float);
#undef OM_LNBD
}

// d_gamma[c] += sum over blocks of partial[b][0][c], d_beta likewise, for n LayerNorm sites in one launch (the partial sums that
// omk_ln_bwd_drop left): grid (ceil(H / 64), n), 16 groups of blocks per column, added in a fixed order -- deterministic.
struct LnSiteTable { OmLnSite site[OM_LN_SITES_MAX]; };
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(LnSiteTable tab, int H) {
  __shared__ float red[2][16][64];
  const OmLnSite st = tab.site[blockIdx.y];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 64 + col;
  float sg = 0.f, sb = 0.f;
  if (c < H)
    for (int b = grp; b < st.blocks; b += 16) {
      sg += st.partial[((size_t)b * 2 + 0) * H + c];
      sb += st.partial[((size_t)b * 2 + 1) * H + c];
    }
  red[0][grp][col] = sg; red[1][grp][col] = sb;
  __syncthreads();
  if (grp < 2 && c < H) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[grp][k][col];
    float* dst = grp == 0 ? st.dg : st.db;
    if (dst) dst[c] += v;
  }
}
int omk_ln_param_reduce(const OmLnSite* sites, int n, int H, hipStream_t s) {
  for (int i = 0; i < n; i += OM_LN_SITES_MAX) {
    LnSiteTable tab;
    const int m = n - i < OM_LN_SITES_MAX ? n - i : OM_LN_SITES_MAX;
    for (int k = 0; k < m; ++k) tab.site[k] = sites[i + k];
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((unsigned)((H + 63) / 64), (unsigned)m), dim3(1024), 0, s, tab, H);
    OM_LAUNCH_CHECK();
  }
  return 0;
}

int omk_norm_bwd(int dtype, const void* dy, const void* x, const float* g, void* dx, float* dg,
                 float* db, int64_t M, int H, float eps, int rms, const void* add, hipStream_t s) {
  if (M <= 0) return 0;
  if (H % 4 || H > 2048) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
  if (dtype == OM_BF16)
    return launch_ln_bwd<bf16_t, 0>(dy, x, g, dx, dg, db, M, H, eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 1, s, rms, add);
  if (dtype == OM_F16)
    return launch_ln_bwd<f16_t, 0>(dy, x, g, dx, dg, db, M, H, eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 1, s, rms, add);
  return launch_ln_bwd<float, 0>(dy, x, g, dx, dg, db, M, H, eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 1, s, rms, add);
}

int omk_embed_bwd(int dtype, const void* dy, const int64_t* ids, const int64_t* type_ids,
                  const float* word, const float* pos, const float* type, const float* g,
                  float* dword, float* dpos, float* dtype_, float* dg, float* db, int64_t M, int L,
                  int H, int vocab, int type_vocab, float eps, hipStream_t s, const int* cu) {
  if (M <= 0) return 0;
  if (H % 4 || H > 2048) OM_FAIL("hidden size must be a multiple of 4 and <= 2048");
#define OM_EB(TT) return launch_ln_bwd<TT, 1>(dy, nullptr, g, nullptr, dg, db, M, H, eps, ids, type_ids, word, pos, type, dword, dpos, dtype_, L, vocab, type_vocab, s, 0, nullptr, nullptr, 0.f, 0, nullptr, nullptr, nullptr, nullptr, cu)
  if (dtype == OM_BF16) OM_EB(bf16_t);
  if (dtype == OM_F16) OM_EB(f16_t);
  OM_EB(float);
#undef OM_EB
}

// rows [*first, M) of a row-major tensor <- 0 (packed rows: the pad rows behind the last sequence; `first` lives on the device)
__global__ __launch_bounds__(256) void zero_rows_from_kernel(char* __restrict__ p, int64_t row_bytes, const int* __restrict__ first, int64_t M) {
  const int64_t vecs = row_bytes >> 4;
  for (int64_t row = (int64_t)first[0] + blockIdx.x; row < M; row += gridDim.x)
    for (int64_t v = threadIdx.x; v < vecs; v += 256) *(uint4*)(p + row * row_bytes + v * 16) = make_uint4(0, 0, 0, 0);
}
int omk_zero_rows_from(void* p, int64_t row_bytes, const int* first, int64_t M, hipStream_t s) {
  if (M <= 0) return 0;
  if (row_bytes % 16 || ((uintptr_t)p & 15)) OM_FAIL("zero_rows_from: rows of whole 16-byte vectors");
  hipLaunchKernelGGL(zero_rows_from_kernel, dim3((unsigned)(M < 512 ? M : 512)), dim3(256), 0, s, (char*)p, row_bytes, first, M);
  OM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// pooling backward: d_hidden[b,t,:] from d_pooled[b,:]
// grid (B, parts): block (b, part) writes rows part, part + parts, .. of sequence b, four elements per thread and access
// (one block per sequence took 97 us for 72 x 128 x 768 outputs; H % 4 == 0 is checked by the callers of the encoder)
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dp, const int64_t* __restrict__ mask,
                                                       T* __restrict__ dh, int L, int H, int mode, const int* __restrict__ cu) {
  // cu != NULL (packed rows): sequence b's rows are cu[b] .. cu[b + 1] - 1 of dh; the mask keeps its [B, L] layout
  const int64_t b = blockIdx.x;
  const int64_t row0 = cu ? cu[b] : b * L;
  const int Lb = cu ? cu[b + 1] - cu[b] : L;
  __shared__ float cnt_s;
  if (mode == OM_POOL_MEAN) {
    if (threadIdx.x < 64) {
      float c = 0.f;
      for (int t = threadIdx.x; t < L; t += 64) c += (float)mask[b * L + t];
      c = wave_sum(c);
      if (threadIdx.x == 0) cnt_s = fmaxf(c, 1e-9f);
    }
    __syncthreads();
  }
  const float cnt = mode == OM_POOL_MEAN ? cnt_s : 1.f;
  const int h4 = H >> 2;
  for (int t = blockIdx.y; t < Lb; t += gridDim.y) {
    const float wgt = mode == OM_POOL_FIRST ? (t == 0 ? 1.f : 0.f) : (float)mask[b * L + t] / cnt;
    for (int c4 = threadIdx.x; c4 < h4; c4 += 256) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (wgt != 0.f) {
        const float4 d = *(const float4*)(dp + b * H + c4 * 4);
        if (mode == OM_POOL_FIRST) { v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w; }
        else { v[0] = d.x * (float)mask[b * L + t] / cnt; v[1] = d.y * (float)mask[b * L + t] / cnt; v[2] = d.z * (float)mask[b * L + t] / cnt; v[3] = d.w * (float)mask[b * L + t] / cnt; }
      }
      store4<T>(dh + (row0 + t) * H + c4 * 4, v);
    }
  }
}

int omk_pool_bwd(int dtype, const float* dp, const int64_t* mask, void* dh, int64_t B, int L, int H,
                 int mode, hipStream_t s, const int* cu) {
  if (B <= 0) return 0;
  if (H % 4) OM_FAIL("hidden size must be a multiple of 4");
  int parts = (int)(2048 / B);
  parts = parts < 1 ? 1 : (parts > L ? L : parts);
  const dim3 grid((unsigned)B, (unsigned)parts);
  if (dtype == OM_BF16) hipLaunchKernelGGL((pool_bwd_kernel<bf16_t>), grid, dim3(256), 0, s, dp, mask, (bf16_t*)dh, L, H, mode, cu);
  else if (dtype == OM_F16) hipLaunchKernelGGL((pool_bwd_kernel<f16_t>), grid, dim3(256), 0, s, dp, mask, (f16_t*)dh, L, H, mode, cu);
  else hipLaunchKernelGGL((pool_bwd_kernel<float>), grid, dim3(256), 0, s, dp, mask, (float*)dh, L, H, mode, cu);
  OM_LAUNCH_CHECK();
  return 0;
}

// y = x / max(|x|, eps):  dx = (dy - y (y . dy)) / max(|x|, eps)     (eps clamp inactive branch: dx = dy/eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ dy,
                                                         float* __restrict__ dx, int64_t M, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[row * D + c]; ss += v * v; dot += v * dy[row * D + c]; }
  ss = wave_sum(ss); dot = wave_sum(dot);
  const float nrm = sqrtf(ss);
  if (nrm > 1e-12f) {
    const float inv = 1.0f / nrm;
    for (int c = lane; c < D; c += 64) {
      const float y = x[row * D + c] * inv;
      dx[row * D + c] = (dy[row * D + c] - y * (dot * inv)) * inv;
    }
  } else {
    for (int c = lane; c < D; c += 64) dx[row * D + c] = dy[row * D + c] * 1e12f;
  }
}

int omk_l2norm_bwd(const float* x, const float* dy, float* dx, int64_t M, int D, hipStream_t s) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, dy, dx, M, D);
  OM_LAUNCH_CHECK();
  return 0;
}

// small f32 contractions (LinearHead backward, [B,768]-sized):
//   nn: C[i,c] = sum_j A[i,j] * Bm[j,c]      tn: C[j,c] = sum_i A[i,j] * Bm[i,c]
__global__ void small_nn_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                float* __restrict__ C, int J, int Cc) {
  const int i = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cc) return;
  float acc = 0.f;
  for (int j = 0; j < J; ++j) acc = fmaf(A[(int64_t)i * J + j], Bm[(int64_t)j * Cc + c], acc);
  C[(int64_t)i * Cc + c] = acc;
}
__global__ void small_tn_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                float* __restrict__ C, int I, int J, int Cc) {
  const int j = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cc) return;
  float acc = 0.f;
  for (int i = 0; i < I; ++i) acc = fmaf(A[(int64_t)i * J + j], Bm[(int64_t)i * Cc + c], acc);
  C[(int64_t)j * Cc + c] = acc;
}
int omk_small_nn(const float* A, const float* Bm, float* C, int I, int J, int Cc, hipStream_t s) {
  if (I <= 0 || Cc <= 0) return 0;
  hipLaunchKernelGGL(small_nn_kernel, dim3((Cc + 255) / 256, I), dim3(256), 0, s, A, Bm, C, J, Cc);
  OM_LAUNCH_CHECK();
  return 0;
}
int omk_small_tn(const float* A, const float* Bm, float* C, int I, int J, int Cc, hipStream_t s) {
  if (J <= 0 || Cc <= 0) return 0;
  hipLaunchKernelGGL(small_tn_kernel, dim3((Cc + 255) / 256, J), dim3(256), 0, s, A, Bm, C, I, J, Cc);
  OM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Attention backward for one (batch, head) per workgroup, L <= 256 (KT <= 8 key/query tiles; KT = 6, 8 serve the
// cross-encoder's 162-token pairs and long passages at one wave per SIMD).
//   P = softmax(scale QK^T + mask), Pd = dropout(P), O = Pd V           (forward, recomputed)
//   dPd = dO V^T ; dP = dropout'(dPd) ; dS = P o (dP - rowsum(P o dP)) * scale
//   dQ = dS K ; dK = dS^T Q ; dV = Pd^T dO
// Phase A (wave w <-> query block w): scores in the swapped orientation (lane <-> query, as in
// the forward) give the row statistics (max, 1/sum, delta) and dQ.
// Phase B (wave w <-> key block w): the same scores in the direct orientation (lane <-> key,
// registers <-> queries) are exactly the A-operand layout needed to contract over queries, so
// dV and dK need no cross-lane traffic; the row statistics come from LDS.
// Only the three transposed images (K^T, Q^T, dO^T: [64][L+4]) live in LDS; row fragments of
// Q / K / V / dO are read straight from global memory (L2 resident, 16 KiB each).
template <typename T, int KT>
__global__ __launch_bounds__(64 * KT) void attention_bwd_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dctx, T* __restrict__ dqkv,
    const int64_t* __restrict__ mask, int Lm, int H, int heads, float scale, float drop_p,
    uint64_t seed, const float* __restrict__ pos_bias, float* __restrict__ drel, const int* __restrict__ cu) {
  // cu != NULL (packed rows; with pos_bias since round 6: T5 training): sequence b is rows cu[b] .. cu[b + 1] - 1, L its own row count; the mask's pitch stays Lm
  // pos_bias [heads][L][L] (T5): added to the scaled scores; drel [heads][2L-1] accumulates the
  // gradient of that bias per relative position key - query (+ L-1), summed over the batch.
  typedef AttnGeom<T> G;
  typedef typename MmaOps<T>::frag_t frag_t;
  constexpr int LP = KT * 32 + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sKt = (T*)smem;
  T* sQt = sKt + 64 * LP;
  T* sDOt = sQt + 64 * LP;
  float* sM = (float*)(sDOt + 64 * LP);        // additive key mask
  float* sMax = sM + KT * 32;                  // per query: row max, 1/row sum, delta
  float* sInv = sMax + KT * 32;
  float* sDelta = sInv + KT * 32;
  float* sRel = sDelta + KT * 32;              // [2 * KT * 32] bias gradient per relative position

  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  int64_t row0 = b * Lm;
  int L = Lm;
  if (cu) { row0 = cu[b]; L = cu[b + 1] - cu[b]; if (L <= 0) return; }
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t ld = 3 * (int64_t)H;
  const T* base = qkv + row0 * ld + h * 64;
  const T* dob = dctx + row0 * H + h * 64;
  T* dbase = dqkv + row0 * ld + h * 64;
  const AttnDrop dr_(drop_p);
  const uint32_t thresh = dr_.thresh;
  const float keep_scale = dr_.keep_scale;

  for (int idx = tid; idx < KT * 32 * G::CPR; idx += nthr) {
    const int row = idx / G::CPR, c = idx % G::CPR;
    uint4 kv = make_uint4(0, 0, 0, 0), qv = kv, dv = kv;
    if (row < L) {
      qv = *(const uint4*)(base + (int64_t)row * ld + c * G::EPC);
      kv = *(const uint4*)(base + (int64_t)row * ld + H + c * G::EPC);
      dv = *(const uint4*)(dob + (int64_t)row * H + c * G::EPC);
    }
    const T* ke = (const T*)&kv; const T* qe = (const T*)&qv; const T* de = (const T*)&dv;
#pragma unroll
    for (int e = 0; e < G::EPC; ++e) {
      sKt[(c * G::EPC + e) * LP + row] = ke[e];
      sQt[(c * G::EPC + e) * LP + row] = qe[e];
      sDOt[(c * G::EPC + e) * LP + row] = de[e];
    }
  }
  for (int k = tid; k < KT * 32; k += nthr)
    sM[k] = k < L ? (mask[b * Lm + k] != 0 ? 0.f : -3.4028235e38f) : -INFINITY;
  if (drel)
    for (int k = tid; k < 2 * KT * 32; k += nthr) sRel[k] = 0.f;
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int blk0 = wave * 32;                  // first query (phase A) / key (phase B) of this wave
  const bool active = blk0 < L;
  const int myrow = (blk0 + l31) < L ? (blk0 + l31) : (L - 1);

  // ------------------------------------------------------------------ phase A
  if (active) {
    frag_t qf[G::NKK], dof[G::NKK];
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
      qf[kk] = *(const frag_t*)(base + (int64_t)myrow * ld + (kk * 2 + half) * G::EPC);
      dof[kk] = *(const frag_t*)(dob + (int64_t)myrow * H + (kk * 2 + half) * G::EPC);
    }
    f32x16_t s[KT], dp[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[t][r] = 0.f; dp[t][r] = 0.f; }
      const int krow = (t * 32 + l31) < L ? (t * 32 + l31) : (L - 1);
      const T* kp = base + (int64_t)krow * ld + H;
      const T* vp = base + (int64_t)krow * ld + 2 * H;
#pragma unroll
      for (int kk = 0; kk < G::NKK; ++kk) {
        const frag_t ka = *(const frag_t*)(kp + (kk * 2 + half) * G::EPC);
        const frag_t va = *(const frag_t*)(vp + (kk * 2 + half) * G::EPC);
        MmaOps<T>::mma(ka, qf[kk], s[t]);      // S^T[key][query]
        MmaOps<T>::mma(va, dof[kk], dp[t]);    // dPd^T[key][query]
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t mb = *(const f32x4_t*)(sM + t * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = s[t][4 * g + e] * scale + mb[e];
          if (pos_bias) {
            const int kc = (t * 32 + 8 * g + 4 * half + e) < L ? (t * 32 + 8 * g + 4 * half + e) : (L - 1);
            v += pos_bias[((int64_t)h * Lm + myrow) * Lm + kc];      // (the table's pitch: the padded length, also for packed rows)
          }
          s[t][4 * g + e] = v; mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float e = G::exp_(s[t][r] - mx); s[t][r] = e; sum += e; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = s[t][r] * inv;
        float dpp = dp[t][r];
        if (thresh) {
          const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          dpp = attn_drop_keep1(seed, b, h, heads, Lm, blk0 + l31, key, thresh) ? dpp * keep_scale : 0.f;
        }
        s[t][r] = p; dp[t][r] = dpp;
        delta += p * dpp;
      }
    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dlogit = s[t][r] * (dp[t][r] - delta);          // d loss / d (scaled score + bias)
        if (drel) {
          const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, q = blk0 + l31;
          if (q < L && key < L) atomicAdd(&sRel[key - q + (Lm - 1)], dlogit);
        }
        s[t][r] = dlogit * scale;                                     // dS
      }
    if (half == 0 && blk0 + l31 < L) { sMax[blk0 + l31] = mx; sInv[blk0 + l31] = inv; sDelta[blk0 + l31] = delta; }
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) SlabMma<T>::run(s[t], sKt + l31 * LP + t * 32 + 4 * half, LP, o);   // dQ = dS K
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = blk0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (q < L) ElemOps<T>::store(dbase + (int64_t)q * ld + dt * 32 + l31, o[dt][r]);
      }
  }
  __syncthreads();
  if (drel)
    for (int k = tid; k < 2 * Lm - 1; k += nthr) atomicAdd(drel + (int64_t)h * (2 * Lm - 1) + k, sRel[k]);
  if (!active) return;

  // ------------------------------------------------------------------ phase B
  {
    const bool kvalid = (blk0 + l31) < L;
    const float mbk = sM[blk0 + l31];
    frag_t kf[G::NKK], vf[G::NKK];
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
      kf[kk] = *(const frag_t*)(base + (int64_t)myrow * ld + H + (kk * 2 + half) * G::EPC);
      vf[kk] = *(const frag_t*)(base + (int64_t)myrow * ld + 2 * H + (kk * 2 + half) * G::EPC);
    }
    f32x16_t dv[2], dk[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
#pragma unroll
    for (int tq = 0; tq < KT; ++tq) {
      if (tq * 32 >= L) break;
      const int qr = (tq * 32 + l31) < L ? (tq * 32 + l31) : (L - 1);
      f32x16_t sb, dpb;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sb[r] = 0.f; dpb[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < G::NKK; ++kk) {
        const frag_t qa = *(const frag_t*)(base + (int64_t)qr * ld + (kk * 2 + half) * G::EPC);
        const frag_t da = *(const frag_t*)(dob + (int64_t)qr * H + (kk * 2 + half) * G::EPC);
        MmaOps<T>::mma(qa, kf[kk], sb);        // S[query][key]
        MmaOps<T>::mma(da, vf[kk], dpb);       // dPd[query][key]
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q4 = tq * 32 + 8 * g + 4 * half;
        const f32x4_t m4 = *(const f32x4_t*)(sMax + q4);
        const f32x4_t i4 = *(const f32x4_t*)(sInv + q4);
        const f32x4_t d4 = *(const f32x4_t*)(sDelta + q4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = q4 + e;
          float p = 0.f, pd = 0.f, dpp = 0.f;
          if (q < L && kvalid) {
            float lg = sb[4 * g + e] * scale + mbk;
            if (pos_bias) lg += pos_bias[((int64_t)h * Lm + q) * Lm + (blk0 + l31)];
            p = G::exp_(lg - m4[e]) * i4[e];
            pd = p; dpp = dpb[4 * g + e];
            if (thresh) {
              const bool keep = attn_drop_keep1(seed, b, h, heads, Lm, q, blk0 + l31, thresh);
              pd = keep ? p * keep_scale : 0.f;
              dpp = keep ? dpp * keep_scale : 0.f;
            }
          }
          sb[4 * g + e] = pd;                                   // Pd[q][key]
          dpb[4 * g + e] = (q < L && kvalid) ? p * (dpp - d4[e]) * scale : 0.f;   // dS[q][key]
        }
      }
      SlabMma<T>::run(sb, sDOt + l31 * LP + tq * 32 + 4 * half, LP, dv);    // dV += Pd^T dO
      SlabMma<T>::run(dpb, sQt + l31 * LP + tq * 32 + 4 * half, LP, dk);    // dK += dS^T Q
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = blk0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key < L) {
          ElemOps<T>::store(dbase + (int64_t)key * ld + H + dt * 32 + l31, dk[dt][r]);
          ElemOps<T>::store(dbase + (int64_t)key * ld + 2 * H + dt * 32 + l31, dv[dt][r]);
        }
      }
  }
}

template <typename T, int KT>
static int launch_attn_bwd(const void* qkv, const void* dctx, void* dqkv, const int64_t* mask,
                           int64_t B, int L, int H, int heads, float scale, float drop_p,
                           uint64_t seed, const float* pos_bias, float* drel, hipStream_t s, const int* cu = nullptr) {
  constexpr int LP = KT * 32 + 4;
  const int lds = 3 * 64 * LP * (int)sizeof(T) + 6 * KT * 32 * 4;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)attention_bwd_kernel<T, KT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int waves = (L + 31) / 32;
  hipLaunchKernelGGL((attention_bwd_kernel<T, KT>), dim3((unsigned)(heads * B)), dim3(64 * waves), lds, s,
                     (const T*)qkv, (const T*)dctx, (T*)dqkv, mask, L, H, heads, scale, drop_p, seed, pos_bias, drel, cu);
  OM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Attention backward beyond 256 tokens (round 6: training up to 512), 16-bit formats.  The kernel above keeps a whole row of scores per
// lane in registers (s[KT], dp[KT]: 32 registers per key tile) and three transposed images in LDS; at 512 tokens neither fits.  Here:
//  * TWO kernels, each with one wave per 32-row block and four blocks per workgroup (grid: (sequence, head) x ceil(L / 128)):
//      A  wave <-> query block: over the key tiles TWICE with one tile in registers -- an online softmax for the row statistics, then the
//         gradients -- with delta = rowsum(P o dP) taken as dO . O from the forward's output (the identity every flash-attention backward
//         uses; O is on the tape); writes dQ and the statistics (max, 1 / sum, delta per query) to `stats`;
//      B  wave <-> key block: a loop over the query tiles (as phase B of the kernel above) with the statistics read back;
//  * LDS holds what each needs: K^T under A; Q^T, dO^T and the statistics under B.
// Same masks (attn_common.h hash on (sequence, head, query, key)), same bias and bias-gradient handling as the kernel above.
#define OM_ABL_LMAX 512
#define OM_ABL_LP (OM_ABL_LMAX + 4)
template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_long_a_kernel(
    const T* __restrict__ qkv, const T* __restrict__ ctx, const T* __restrict__ dctx, T* __restrict__ dqkv,
    const int64_t* __restrict__ mask, int L, int H, int heads, float scale, float drop_p,
    uint64_t seed, const float* __restrict__ pos_bias, float* __restrict__ drel, float* __restrict__ stats) {
  typedef AttnGeom<T> G;
  typedef typename MmaOps<T>::frag_t frag_t;
  constexpr int LMAX = OM_ABL_LMAX, LP = OM_ABL_LP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sKt = (T*)smem;                           // K^T: [64][LP]
  float* sM = (float*)(sKt + 64 * LP);         // additive key mask
  float* sRel = sM + LMAX;                     // [2 * LMAX] bias gradient per relative position

  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int KT = (L + 31) / 32;
  const int64_t ld = 3 * (int64_t)H;
  const T* base = qkv + b * L * ld + h * 64;
  const T* dob = dctx + b * L * H + h * 64;
  const T* ob = ctx + b * L * H + h * 64;
  T* dbase = dqkv + b * L * ld + h * 64;
  float* st = stats + (int64_t)blockIdx.x * 3 * LMAX;      // [3][LMAX]: max, 1 / sum, delta
  const AttnDrop dr_(drop_p);
  const uint32_t thresh = dr_.thresh;
  const float keep_scale = dr_.keep_scale;

  for (int idx = tid; idx < KT * 32 * G::CPR; idx += nthr) {
    const int row = idx / G::CPR, c = idx % G::CPR;
    uint4 kv = make_uint4(0, 0, 0, 0);
    if (row < L) kv = *(const uint4*)(base + (int64_t)row * ld + H + c * G::EPC);
    const T* ke = (const T*)&kv;
#pragma unroll
    for (int e = 0; e < G::EPC; ++e) sKt[(c * G::EPC + e) * LP + row] = ke[e];
  }
  for (int k = tid; k < KT * 32; k += nthr)
    sM[k] = k < L ? (mask[b * L + k] != 0 ? 0.f : -3.4028235e38f) : -INFINITY;
  if (drel)
    for (int k = tid; k < 2 * LMAX; k += nthr) sRel[k] = 0.f;
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int blk0 = (blockIdx.y * 4 + wave) * 32;       // this wave's query block
  if (blk0 < L) {
    const int myrow = (blk0 + l31) < L ? (blk0 + l31) : (L - 1);
    frag_t qf[G::NKK], dof[G::NKK];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
      qf[kk] = *(const frag_t*)(base + (int64_t)myrow * ld + (kk * 2 + half) * G::EPC);
      dof[kk] = *(const frag_t*)(dob + (int64_t)myrow * H + (kk * 2 + half) * G::EPC);
      const uint4 dw = *(const uint4*)(dob + (int64_t)myrow * H + (kk * 2 + half) * G::EPC);
      const uint4 ow = *(const uint4*)(ob + (int64_t)myrow * H + (kk * 2 + half) * G::EPC);
      const uint32_t dws[4] = {dw.x, dw.y, dw.z, dw.w}, ows[4] = {ow.x, ow.y, ow.z, ow.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) delta += Half16<T>::lo(dws[e]) * Half16<T>::lo(ows[e]) + Half16<T>::hi(dws[e]) * Half16<T>::hi(ows[e]);
    }
    delta += __shfl_xor(delta, 32, 64);          // dO . O over the 64 features of the head: = rowsum(P o dP)
    // one key tile of scaled, masked, biased scores for this lane's query: registers <-> keys t*32 + (r&3) + 8(r>>2) + 4 half
#define OM_ABL_SCORES(T_, S_)                                                                            \
    do {                                                                                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) S_[r] = 0.f;                                        \
      const int krow_ = ((T_) * 32 + l31) < L ? ((T_) * 32 + l31) : (L - 1);                             \
      const T* kp_ = base + (int64_t)krow_ * ld + H;                                                     \
      _Pragma("unroll") for (int kk = 0; kk < G::NKK; ++kk) {                                            \
        const frag_t ka_ = *(const frag_t*)(kp_ + (kk * 2 + half) * G::EPC);                             \
        MmaOps<T>::mma(ka_, qf[kk], S_);                                                                 \
      }                                                                                                  \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
        const f32x4_t mb_ = *(const f32x4_t*)(sM + (T_) * 32 + 8 * g + 4 * half);                        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
          float v_ = S_[4 * g + e] * scale + mb_[e];                                                     \
          if (pos_bias) {                                                                                \
            const int kc_ = ((T_) * 32 + 8 * g + 4 * half + e) < L ? ((T_) * 32 + 8 * g + 4 * half + e) : (L - 1); \
            v_ += pos_bias[((int64_t)h * L + myrow) * L + kc_];                                          \
          }                                                                                              \
          S_[4 * g + e] = v_;                                                                            \
        }                                                                                                \
      }                                                                                                  \
    } while (0)
    // pass 1: the row's maximum and normaliser, online over the key tiles
    float m_run = -INFINITY, l_run = 0.f;
    for (int t = 0; t < KT; ++t) {
      f32x16_t sc;
      OM_ABL_SCORES(t, sc);
      float mx = m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // (tile 0 holds key 0: unmasked or finfo.min, finite -- mx is finite from here on)
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += G::exp_(sc[r] - mx);
      sum += __shfl_xor(sum, 32, 64);
      l_run = l_run * G::exp_(m_run - mx) + sum;
      m_run = mx;
    }
    const float inv = 1.0f / l_run;
    // pass 2: P, dP, dS tile by tile; dQ accumulates
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    for (int t = 0; t < KT; ++t) {
      f32x16_t sc, dp;
      OM_ABL_SCORES(t, sc);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
      {
        const int krow = (t * 32 + l31) < L ? (t * 32 + l31) : (L - 1);
        const T* vp = base + (int64_t)krow * ld + 2 * H;
#pragma unroll
        for (int kk = 0; kk < G::NKK; ++kk) {
          const frag_t va = *(const frag_t*)(vp + (kk * 2 + half) * G::EPC);
          MmaOps<T>::mma(va, dof[kk], dp);       // dPd^T[key][query]
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float pr = G::exp_(sc[r] - m_run) * inv;
        float dpp = dp[r];
        if (thresh) dpp = attn_drop_keep1(seed, b, h, heads, L, blk0 + l31, key, thresh) ? dpp * keep_scale : 0.f;
        const float dlogit = pr * (dpp - delta);                      // d loss / d (scaled score + bias)
        if (drel && (blk0 + l31) < L && key < L) atomicAdd(&sRel[key - (blk0 + l31) + (L - 1)], dlogit);
        sc[r] = dlogit * scale;                                        // dS
      }
      SlabMma<T>::run(sc, sKt + l31 * LP + t * 32 + 4 * half, LP, o);   // dQ += dS K
    }
#undef OM_ABL_SCORES
    if (half == 0 && blk0 + l31 < L) { st[blk0 + l31] = m_run; st[LMAX + blk0 + l31] = inv; st[2 * LMAX + blk0 + l31] = delta; }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = blk0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (q < L) ElemOps<T>::store(dbase + (int64_t)q * ld + dt * 32 + l31, o[dt][r]);
      }
  }
  if (drel) {
    __syncthreads();
    for (int k = tid; k < 2 * L - 1; k += nthr) atomicAdd(drel + (int64_t)h * (2 * L - 1) + k, sRel[k]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_long_b_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dctx, T* __restrict__ dqkv,
    const int64_t* __restrict__ mask, int L, int H, int heads, float scale, float drop_p,
    uint64_t seed, const float* __restrict__ pos_bias, const float* __restrict__ stats) {
  typedef AttnGeom<T> G;
  typedef typename MmaOps<T>::frag_t frag_t;
  constexpr int LMAX = OM_ABL_LMAX, LP = OM_ABL_LP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sQt = (T*)smem;                           // Q^T: [64][LP]
  T* sDOt = sQt + 64 * LP;                     // dO^T
  float* sMax = (float*)(sDOt + 64 * LP);      // per query: row max, 1 / row sum, delta (kernel A)
  float* sInv = sMax + LMAX;
  float* sDelta = sInv + LMAX;

  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int KT = (L + 31) / 32;
  const int64_t ld = 3 * (int64_t)H;
  const T* base = qkv + b * L * ld + h * 64;
  const T* dob = dctx + b * L * H + h * 64;
  T* dbase = dqkv + b * L * ld + h * 64;
  const float* st = stats + (int64_t)blockIdx.x * 3 * LMAX;
  const AttnDrop dr_(drop_p);
  const uint32_t thresh = dr_.thresh;
  const float keep_scale = dr_.keep_scale;

  for (int idx = tid; idx < KT * 32 * G::CPR; idx += nthr) {
    const int row = idx / G::CPR, c = idx % G::CPR;
    uint4 qv = make_uint4(0, 0, 0, 0), dv = qv;
    if (row < L) {
      qv = *(const uint4*)(base + (int64_t)row * ld + c * G::EPC);
      dv = *(const uint4*)(dob + (int64_t)row * H + c * G::EPC);
    }
    const T* qe = (const T*)&qv; const T* de = (const T*)&dv;
#pragma unroll
    for (int e = 0; e < G::EPC; ++e) {
      sQt[(c * G::EPC + e) * LP + row] = qe[e];
      sDOt[(c * G::EPC + e) * LP + row] = de[e];
    }
  }
  for (int k = tid; k < KT * 32; k += nthr) {
    sMax[k] = k < L ? st[k] : 0.f; sInv[k] = k < L ? st[LMAX + k] : 0.f; sDelta[k] = k < L ? st[2 * LMAX + k] : 0.f;
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int blk0 = (blockIdx.y * 4 + wave) * 32;       // this wave's key block
  if (blk0 >= L) return;
  const int myrow = (blk0 + l31) < L ? (blk0 + l31) : (L - 1);
  const bool kvalid = (blk0 + l31) < L;
  const float mbk = kvalid ? (mask[b * L + blk0 + l31] != 0 ? 0.f : -3.4028235e38f) : -INFINITY;
  frag_t kf[G::NKK], vf[G::NKK];
#pragma unroll
  for (int kk = 0; kk < G::NKK; ++kk) {
    kf[kk] = *(const frag_t*)(base + (int64_t)myrow * ld + H + (kk * 2 + half) * G::EPC);
    vf[kk] = *(const frag_t*)(base + (int64_t)myrow * ld + 2 * H + (kk * 2 + half) * G::EPC);
  }
  f32x16_t dv[2], dk[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
  for (int tq = 0; tq < KT; ++tq) {
    const int qr = (tq * 32 + l31) < L ? (tq * 32 + l31) : (L - 1);
    f32x16_t sb, dpb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sb[r] = 0.f; dpb[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
      const frag_t qa = *(const frag_t*)(base + (int64_t)qr * ld + (kk * 2 + half) * G::EPC);
      const frag_t da = *(const frag_t*)(dob + (int64_t)qr * H + (kk * 2 + half) * G::EPC);
      MmaOps<T>::mma(qa, kf[kk], sb);        // S[query][key]
      MmaOps<T>::mma(da, vf[kk], dpb);       // dPd[query][key]
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q4 = tq * 32 + 8 * g + 4 * half;
      const f32x4_t m4 = *(const f32x4_t*)(sMax + q4);
      const f32x4_t i4 = *(const f32x4_t*)(sInv + q4);
      const f32x4_t d4 = *(const f32x4_t*)(sDelta + q4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q4 + e;
        float pr = 0.f, pd = 0.f, dpp = 0.f;
        if (q < L && kvalid) {
          float lg = sb[4 * g + e] * scale + mbk;
          if (pos_bias) lg += pos_bias[((int64_t)h * L + q) * L + (blk0 + l31)];
          pr = G::exp_(lg - m4[e]) * i4[e];
          pd = pr; dpp = dpb[4 * g + e];
          if (thresh) {
            const bool keep = attn_drop_keep1(seed, b, h, heads, L, q, blk0 + l31, thresh);
            pd = keep ? pr * keep_scale : 0.f;
            dpp = keep ? dpp * keep_scale : 0.f;
          }
        }
        sb[4 * g + e] = pd;                                   // Pd[q][key]
        dpb[4 * g + e] = (q < L && kvalid) ? pr * (dpp - d4[e]) * scale : 0.f;   // dS[q][key]
      }
    }
    SlabMma<T>::run(sb, sDOt + l31 * LP + tq * 32 + 4 * half, LP, dv);    // dV += Pd^T dO
    SlabMma<T>::run(dpb, sQt + l31 * LP + tq * 32 + 4 * half, LP, dk);    // dK += dS^T Q
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = blk0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key < L) {
        ElemOps<T>::store(dbase + (int64_t)key * ld + H + dt * 32 + l31, dk[dt][r]);
        ElemOps<T>::store(dbase + (int64_t)key * ld + 2 * H + dt * 32 + l31, dv[dt][r]);
      }
    }
}

template <typename T>
static int launch_attn_bwd_long(const void* qkv, const void* ctx, const void* dctx, void* dqkv, const int64_t* mask,
                                int64_t B, int L, int H, int heads, float scale, float drop_p,
                                uint64_t seed, const float* pos_bias, float* drel, float* stats, hipStream_t s) {
  const int lds_a = 64 * OM_ABL_LP * (int)sizeof(T) + 3 * OM_ABL_LMAX * 4;
  const int lds_b = 2 * 64 * OM_ABL_LP * (int)sizeof(T) + 3 * OM_ABL_LMAX * 4;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)attention_bwd_long_a_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
    OM_HIP(hipFuncSetAttribute((const void*)attention_bwd_long_b_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
    attr_set = true;
  }
  const dim3 grid((unsigned)(heads * B), (unsigned)((L + 127) / 128));
  hipLaunchKernelGGL((attention_bwd_long_a_kernel<T>), grid, dim3(256), lds_a, s,
                     (const T*)qkv, (const T*)ctx, (const T*)dctx, (T*)dqkv, mask, L, H, heads, scale, drop_p, seed, pos_bias, drel, stats);
  OM_LAUNCH_CHECK();
  hipLaunchKernelGGL((attention_bwd_long_b_kernel<T>), grid, dim3(256), lds_b, s,
                     (const T*)qkv, (const T*)dctx, (T*)dqkv, mask, L, H, heads, scale, drop_p, seed, pos_bias, (const float*)stats);
  OM_LAUNCH_CHECK();
  return 0;
}

// 256 < L <= 512, 16-bit formats: needs the forward's output (the tape's ctx)
size_t omk_attention_bwd_long_stats_bytes(int64_t B, int heads) { return (size_t)B * heads * 3 * OM_ABL_LMAX * 4; }
int omk_attention_bwd_long(int dtype, const void* qkv, const void* ctx, const void* dctx, void* dqkv, const int64_t* mask,
                           int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                           const float* pos_bias, float* drel, float* stats, hipStream_t s) {
  if (B <= 0) return 0;
  if (L < 1 || L > 512) OM_FAIL("attention backward (tile-at-a-time form): up to 512 tokens");      // (taken from 257 on; below that only when a test forces it)
  if (H != heads * 64) OM_FAIL("head_dim must be 64");
  if (!ctx || !stats) OM_FAIL("attention backward beyond 256 tokens needs the forward's output and a statistics buffer (omk_attention_bwd_long_stats_bytes)");
  if (dtype == OM_BF16) return launch_attn_bwd_long<bf16_t>(qkv, ctx, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, stats, s);
  if (dtype == OM_F16) return launch_attn_bwd_long<f16_t>(qkv, ctx, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, stats, s);
  OM_FAIL("attention backward beyond 256 tokens: 16-bit formats");
}

int omk_attention_bwd(int dtype, const void* qkv, const void* dctx, void* dqkv, const int64_t* mask,
                      int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                      hipStream_t s, const int* cu) {
  if (cu) {           // packed rows (16-bit formats): the transposing-read kernel up to 128 tokens, the generic one up to 256
    if (dtype != OM_BF16 && dtype != OM_F16) OM_FAIL("packed rows: attention backward for 16-bit formats");
    if (B <= 0) return 0;
    if (omk_attention_bwd16_ok(dtype, L, H, heads))
      return omk_attention_bwd16(dtype, qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, s, cu);
    if (L < 1 || L > 256 || H != heads * 64) OM_FAIL("packed rows: attention backward up to 256 tokens, head_dim 64");
#define ABP(TT)                                                                                      \
  do {                                                                                               \
    if (L <= 32) return launch_attn_bwd<TT, 1>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, nullptr, nullptr, s, cu); \
    if (L <= 64) return launch_attn_bwd<TT, 2>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, nullptr, nullptr, s, cu); \
    if (L <= 128) return launch_attn_bwd<TT, 4>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, nullptr, nullptr, s, cu); \
    if (L <= 192) return launch_attn_bwd<TT, 6>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, nullptr, nullptr, s, cu); \
    return launch_attn_bwd<TT, 8>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, nullptr, nullptr, s, cu);    \
  } while (0)
    if (dtype == OM_BF16) ABP(bf16_t);
    ABP(f16_t);
#undef ABP
  }
  return omk_attention_bwd_bias(dtype, qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, nullptr, nullptr, s, nullptr);
}

int omk_attention_bwd_bias(int dtype, const void* qkv, const void* dctx, void* dqkv, const int64_t* mask,
                           int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                           const float* pos_bias, float* drel, hipStream_t s, const int* cu) {
  if (B <= 0) return 0;
  if (cu && dtype != OM_BF16 && dtype != OM_F16) OM_FAIL("packed rows: attention backward for 16-bit formats");
  if (omk_attention_bwd16_ok(dtype, L, H, heads) && (pos_bias != nullptr) == (drel != nullptr))      // (with the T5 bias too since round 6)
    return omk_attention_bwd16(dtype, qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, s, cu, pos_bias, drel);
  if (L < 1 || L > 256) OM_FAIL("training supports sequence lengths up to 256");
  // the three transposed [64][L + 4] images of the backward kernel must fit the 160 KiB of LDS: 256 keys in 16 bits, 192 in f32
  if (dtype == OM_F32 && L > 192) OM_FAIL("float32 training supports sequence lengths up to 192 (16-bit formats: 256)");
  if (H != heads * 64) OM_FAIL("head_dim must be 64");
#define AB(TT)                                                                                       \
  do {                                                                                               \
    if (L <= 32) return launch_attn_bwd<TT, 1>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, s, cu); \
    if (L <= 64) return launch_attn_bwd<TT, 2>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, s, cu); \
    if (L <= 128) return launch_attn_bwd<TT, 4>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, s, cu); \
    if (L <= 192) return launch_attn_bwd<TT, 6>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, s, cu); \
    return launch_attn_bwd<TT, 8>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, pos_bias, drel, s, cu);    \
  } while (0)
  if (dtype == OM_BF16) AB(bf16_t);
  if (dtype == OM_F16) AB(f16_t);
  AB(float);
#undef AB
}


// ---------------------------------------------------------------------------------------
// T5 feed-forward activations (HF:models/t5/modeling_t5.py T5DenseActDense / T5DenseGatedActDense)
//   kind 0 (relu):        g = relu(f)                      df  = dg (f > 0)
//   kind 1 (gated gelu):  g = gelu_new(f) * f2             df  = dg f2 gelu_new'(f),  df2 = dg gelu_new(f)
__device__ inline float gelu_new_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ inline float gelu_new_grad_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float t = tanhf(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}
template <typename T>
__global__ void t5_act_fwd_kernel(const T* __restrict__ f, const T* __restrict__ f2, T* __restrict__ g, int64_t n, int kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = ElemOps<T>::load(f + i);
  ElemOps<T>::store(g + i, kind == 0 ? fmaxf(x, 0.f) : gelu_new_f(x) * ElemOps<T>::load(f2 + i));
}
template <typename T>
__global__ void t5_act_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ f, const T* __restrict__ f2,
                                  T* __restrict__ df, T* __restrict__ df2, int64_t n, int kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = ElemOps<T>::load(dg + i), x = ElemOps<T>::load(f + i);
  if (kind == 0) {
    ElemOps<T>::store(df + i, x > 0.f ? d : 0.f);
  } else {
    const float y = ElemOps<T>::load(f2 + i);
    ElemOps<T>::store(df + i, d * y * gelu_new_grad_f(x));
    ElemOps<T>::store(df2 + i, d * gelu_new_f(x));
  }
}
int omk_t5_act_fwd(int dtype, const void* f, const void* f2, void* g, int64_t n, int kind, hipStream_t s) {
  if (n <= 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == OM_BF16) hipLaunchKernelGGL(t5_act_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)f, (const bf16_t*)f2, (bf16_t*)g, n, kind);
  else if (dtype == OM_F16) hipLaunchKernelGGL(t5_act_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)f, (const f16_t*)f2, (f16_t*)g, n, kind);
  else hipLaunchKernelGGL(t5_act_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)f, (const float*)f2, (float*)g, n, kind);
  OM_LAUNCH_CHECK();
  return 0;
}
int omk_t5_act_bwd(int dtype, const void* dg, const void* f, const void* f2, void* df, void* df2, int64_t n, int kind, hipStream_t s) {
  if (n <= 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == OM_BF16) hipLaunchKernelGGL(t5_act_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dg, (const bf16_t*)f, (const bf16_t*)f2, (bf16_t*)df, (bf16_t*)df2, n, kind);
  else if (dtype == OM_F16) hipLaunchKernelGGL(t5_act_bwd_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)dg, (const f16_t*)f, (const f16_t*)f2, (f16_t*)df, (f16_t*)df2, n, kind);
  else hipLaunchKernelGGL(t5_act_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dg, (const float*)f, (const float*)f2, (float*)df, (float*)df2, n, kind);
  OM_LAUNCH_CHECK();
  return 0;
}

// T5 embedding backward: d word_emb[id] += dy[row]   (shared embedding, no norm, no positions)
template <typename T>
__global__ void t5_embed_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ ids, float* __restrict__ dword,
                                    int64_t M, int H, int vocab, const int* __restrict__ row_map) {
  const int64_t row = blockIdx.x;
  int64_t tok = row;
  if (row_map) { tok = row_map[row]; if (tok < 0) return; }
  int64_t id = ids[tok]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  for (int c = threadIdx.x; c < H; c += blockDim.x) atomicAdd(dword + id * H + c, ElemOps<T>::load(dy + row * H + c));
}
int omk_t5_embed_bwd(int dtype, const void* dy, const int64_t* ids, float* dword, int64_t M, int H, int vocab, hipStream_t s, const int* row_map) {
  if (M <= 0) return 0;
  if (dtype == OM_BF16) hipLaunchKernelGGL(t5_embed_bwd_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, s, (const bf16_t*)dy, ids, dword, M, H, vocab, row_map);
  else if (dtype == OM_F16) hipLaunchKernelGGL(t5_embed_bwd_kernel<f16_t>, dim3((unsigned)M), dim3(256), 0, s, (const f16_t*)dy, ids, dword, M, H, vocab, row_map);
  else hipLaunchKernelGGL(t5_embed_bwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (const float*)dy, ids, dword, M, H, vocab, row_map);
  OM_LAUNCH_CHECK();
  return 0;
}

// relative-position bias backward: d table[bucket(rel)][h] += d rel[h][rel]   (table is [buckets][heads])
__global__ void t5_bias_bwd_kernel(const float* __restrict__ drel, const int* __restrict__ lut, float* __restrict__ dtable,
                                   int L, int heads) {
  const int h = blockIdx.x;
  for (int r = threadIdx.x; r < 2 * L - 1; r += blockDim.x)
    atomicAdd(dtable + (int64_t)lut[r] * heads + h, drel[(int64_t)h * (2 * L - 1) + r]);
}
int omk_t5_bias_bwd(const float* drel, const int* lut, float* dtable, int L, int heads, hipStream_t s) {
  hipLaunchKernelGGL(t5_bias_bwd_kernel, dim3((unsigned)heads), dim3(256), 0, s, drel, lut, dtable, L, heads);
  OM_LAUNCH_CHECK();
  return 0;
}
