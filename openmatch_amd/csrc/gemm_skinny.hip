// C[M,N] = epilogue(A[M,K] W[N,K]^T) for FEW rows (M <= a few hundred): the nn.Linear of a forward over one query or a handful of
// sequences -- what the reference runs per search request (retriever/dense_retriever.py:151 encodes the query batch; a served
// query is ONE 32-token row block).  Round 5.
//
// Why a kernel of its own: the tile generations of gemm.hip stream W through 128- or 256-column tiles, so at M = 32 a
// [768 x 3072] weight is read by SIX workgroups, 786 KB each, one K step after another: ~20 us per launch whatever the batch,
// x 84 launches = the 1.9 ms floor under every small forward (docs/history.md: "a captured graph changes nothing: the floor is
// the GPU-side chain").  The operation is a weight STREAM: 2 K N bytes once from HBM, M x that in flops, nothing to reuse.
// So: workgroups of MT x NT output tiles of 16 x 16 (about one workgroup per CU: 48 ... 192 of them where the tile kernels had
// 6 ... 24), their 4 or 8 waves split K, every lane loads its MFMA fragments straight from global memory (16 bytes per lane:
// row l % 16, 8 consecutive k) -- no LDS staging, up to 12 K steps of loads in flight per wave before the first MFMA -- and the
// partial tiles meet in LDS in a fixed order.  A row's result does not depend on M or on the rows beside it.
// Measured (profiles/r05_skinny_sweep_*.txt, kernel durations): 3.1 - 8 us per launch up to 64 rows, 5.7 - 14 us at 256, 7.6 - 21 us
// at 512, where the tile kernels take 18 - 56 us whatever the row count; past ~1 000 rows the re-reads of A (N / (16 NT) times,
// from L2) cost more than the tiles' idle CUs and the tile kernels take over (OM_OPT_GEMM_SKINNY_M).
//
// Roofline: HBM; algorithmic bytes per launch = 2 N K (the weight) + 2 M (K + N) (activations in, out).
#include <algorithm>
#include <atomic>
#include "kernels.h"
#include "gemm_epilogue.h"
#include "ln_row.h"

namespace {
typedef uint32_t sk_u32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct SkMma;
template <> struct SkMma<bf16_t> {
  __device__ static inline f32x4_t mma(const sk_u32x4_t& a, const sk_u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct SkMma<f16_t> {
  __device__ static inline f32x4_t mma(const sk_u32x4_t& a, const sk_u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ float sk_act(float v, int act) {
  switch (act) {
    case OM_ACT_GELU_ERF: return act_apply<OM_ACT_GELU_ERF, true>(v);
    case OM_ACT_RELU: return act_apply<OM_ACT_RELU, true>(v);
    case OM_ACT_GELU_TANH: return act_apply<OM_ACT_GELU_TANH, true>(v);
  }
  return v;
}

// pending LayerNorms of a launch (GemmEpilogue a_ln32 ... rln32_stats)
struct SkLn {
  const float *a32, *a_g, *a_b;
  float* stats_out;
  const float *r32, *r_stats, *r_g, *r_b;
  float eps;
};

// grid (N / (16 NT), ceil(M / (16 MT))), NW waves that split K; MFMA operand 1 = 16 rows of A, operand 2 = 16 rows of W (the output
// columns): lane l holds k = 8 (l >> 4) .. + 7 of row l & 15 of either; D[row 4 (l >> 4) + i][column l & 15].
// MT x NT tiles per wave: (MT + NT) loads feed MT NT MFMAs per K step -- NT > 1 cuts the re-reads of A (every workgroup of a row
// block reads all of A's K slice: N / (16 NT) times per launch, from L2), NW sets the length of a wave's dependent load chain.
//
// LNA (round 6): the A operand is a PENDING LayerNorm -- f32 rows a_ln32 [M, K] to be normalised with (gamma, beta).  The weight
// fragments of the first batch are requested first; under their flight the workgroup's 16 MT rows are normalised by its waves exactly as
// layernorm_kernel would (ln_row.h: one wave per row, the same reductions, the same fused multiply-add, the same rounding to T) into
// LDS, from where every wave takes its K slice of all rows as fragments.  The first column block leaves (mean, rstd) per row for the
// contraction that later adds the same LayerNorm's output as its residual (rln32: the element is re-derived, bit for bit, from the
// row's statistics).  A forward over <= 64 token rows then has no normalisation launches at all (encoder.hip).
template <typename T, int MT, int NT, int NW, bool LNA>
__global__ __launch_bounds__(64 * NW) void gemm_nt_skinny_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw,
                                                                T* C, int64_t ldc, int M, int N, int K, const float* __restrict__ bias,
                                                                const T* resid, int64_t ldr, int act, int mul, const float* resid32, float* out32,
                                                                SkLn ln) {
  constexpr int UNR = LNA ? (NW == 8 ? ((MT + NT) <= 3 ? 4 : 2) : ((MT + NT) <= 4 ? 6 : 2))      // (eight waves: 256 registers per lane, and the rows' f32 copies live beside the weight fragments)
                          : ((MT + NT) <= 3 ? 12 : ((MT + NT) <= 5 ? 8 : 4));      // K steps in flight per wave: (MT + NT) UNR x 4 VGPRs
  constexpr int RED_BYTES = NW * MT * NT * 4 * 64 * 4;
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  float (*red)[MT * NT][4][64] = (float (*)[MT * NT][4][64])sk_smem;     // [NW][MT * NT][4][64], after the K loop
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * (16 * MT);
  const int ksl = K / NW;                                   // this wave's K slice
  const T* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = W + (int64_t)(n0 + 16 * t + r) * ldw + wave * ksl + kg * 8;
  const int steps = ksl >> 5;
  const int xs_pitch = 2 * K + 16;                          // LNA: the normalised rows in LDS; + 16 bytes: the 16 rows of a fragment read on distinct banks
  const T* ap[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    int m = m0 + 16 * j + r;
    m = m < M ? m : M - 1;                                  // rows past M: computed from a valid row, never stored
    ap[j] = LNA ? (const T*)(sk_smem + (size_t)(16 * j + r) * xs_pitch) + wave * ksl + kg * 8 : A + (int64_t)m * lda + wave * ksl + kg * 8;
  }
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < steps; s0 += UNR) {
    sk_u32x4_t wq[NT][UNR], aq[MT][UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (s0 + u < steps) {
#pragma unroll
        for (int t = 0; t < NT; ++t) wq[t][u] = __builtin_nontemporal_load((const sk_u32x4_t*)(wp[t] + (s0 + u) * 32));      // the weight passes once
        if (!LNA) {
#pragma unroll
          for (int j = 0; j < MT; ++j) aq[j][u] = *(const sk_u32x4_t*)(ap[j] + (s0 + u) * 32);
        }
      }
    if (LNA && s0 == 0) {
      // the workgroup's rows, normalised: wave w takes rows w, w + NW, ... (all their loads first: one trip to memory)
      constexpr int RPW = (16 * MT + NW - 1) / NW;
      const int nvec = (K / 4 + 63) / 64;
      float x[RPW][4][4];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int rr = wave + q * NW;
        int m = m0 + rr;
        m = m < M ? m : M - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (lane + 64 * j) * 4;
          if (rr < 16 * MT && j < nvec && c < K) {
            const float4 t4 = *(const float4*)(ln.a32 + (int64_t)m * K + c);
            x[q][j][0] = t4.x; x[q][j][1] = t4.y; x[q][j][2] = t4.z; x[q][j][3] = t4.w;
          }
        }
      }
      float gv[4][4], bv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (j < nvec && c < K) {
          const float4 g4 = *(const float4*)(ln.a_g + c), b4 = *(const float4*)(ln.a_b + c);
          gv[j][0] = g4.x; gv[j][1] = g4.y; gv[j][2] = g4.z; gv[j][3] = g4.w;
          bv[j][0] = b4.x; bv[j][1] = b4.y; bv[j][2] = b4.z; bv[j][3] = b4.w;
        }
      }
      float mean[RPW], rstd[RPW];
      ln_rows_stats<RPW, 4>(x, nvec, lane, K, ln.eps, 0, mean, rstd);
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int rr = wave + q * NW;
        if (rr < 16 * MT) {
          if (ln.stats_out && blockIdx.x == 0 && lane == 0 && m0 + rr < M) *(float2*)(ln.stats_out + (int64_t)(m0 + rr) * 2) = make_float2(mean[q], rstd[q]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (j < nvec && c < K) {
              float y[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) y[e] = ln_affine(x[q][j][e], mean[q], rstd[q], gv[j][e], bv[j][e]);
              Vec4<T>::store((T*)(sk_smem + (size_t)rr * xs_pitch) + c, y);
            }
          }
        }
      }
      __syncthreads();
    }
    if (LNA) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (s0 + u < steps) {
#pragma unroll
          for (int j = 0; j < MT; ++j) aq[j][u] = *(const sk_u32x4_t*)(ap[j] + (s0 + u) * 32);
        }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (s0 + u < steps) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[j][t] = SkMma<T>::mma(aq[j][u], wq[t][u], acc[j][t]);
      }
  }
  if (LNA) __syncthreads();                                 // the partials below reuse the rows' LDS
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][j * NT + t][i][lane] = acc[j][t][i];
  __syncthreads();
  // MT NT 4 accumulator registers x 64 lanes to finish, one (tile, register) per wave at a time; partials added in wave order
  for (int q = wave; q < MT * NT * 4; q += NW) {
    const int tile = q >> 2, i = q & 3, j = tile / NT, t = tile % NT;
    const int m = m0 + 16 * j + 4 * kg + i, n = n0 + 16 * t + r;
    if (m >= M) continue;
    float v = red[0][tile][i][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[w][tile][i][lane];
    v = sk_act(v + (bias ? bias[n] : 0.f), act);
    if (ln.r32) {
      const float2 st = *(const float2*)(ln.r_stats + (int64_t)m * 2);      // the residual is a pending LayerNorm: this element of its output
      v += ln_affine(ln.r32[(int64_t)m * ldr + n], st.x, st.y, ln.r_g[n], ln.r_b[n]);
    } else if (resid32) {
      v += resid32[(int64_t)m * ldr + n];                                  // the f32 residual stream of the few-rows path (may alias out32)
    } else if (resid) {
      const float rv = ElemOps<T>::load(resid + (int64_t)m * ldr + n);      // may alias C: read and written by this lane only
      v = mul ? v * rv : v + rv;
    }
    if (out32) out32[(int64_t)m * ldc + n] = v;
    else ElemOps<T>::store(C + (int64_t)m * ldc + n, v);
  }
}

// (MT, NT, NW) per shape, from the sweep over the four contractions of a bert-base layer at 16 ... 1024 rows
// (profiles/r05_skinny_sweep_*.txt): about one workgroup per CU -- the 16 x 16 output tiles of the problem over 256, as tiles per
// workgroup (rows first: a second row tile re-uses the weight fragment, a second column tile the activation fragment).
struct SkCfg { int mt, nt, nw; };
static SkCfg sk_choose(int64_t M, int64_t N, int64_t K) {
  const int forced = om_option(OM_OPT_GEMM_SKINNY_CFG);            // A/B: MT * 10000 + NT * 100 + NW
  if (forced > 0) return SkCfg{forced / 10000, forced / 100 % 100, forced % 100};
  const int64_t rt = (M + 15) / 16, tiles = rt * (N / 16);
  int p = 1;
  while (p < 8 && tiles > 256 * p) p <<= 1;
  SkCfg c{1, 1, 4};
  if (p >= 2) { if (rt >= 2) c.mt = 2; else c.nt = 2; }
  if (p >= 4) { if (c.nt == 1 && N % 32 == 0) c.nt = 2; else if (rt >= 4) c.mt = 4; }
  if (p >= 8) { if (rt >= 4 && c.mt < 4) c.mt = 4; else if (N % 64 == 0) c.nt = 4; }
  while (c.nt > 1 && N % (16 * c.nt)) c.nt >>= 1;
  // The K split is a function of (N, K) ONLY: the order in which a row's products are added then does not depend on the rows
  // that ride along -- a query encoded alone and in a batch gives the same bits.  Eight slices where N is narrow (few workgroups,
  // long K chains: out-proj, FFN2), four where N is wide (within 15 % of the best split at every row count of the sweep).
  c.nw = (K % 256 == 0 && N <= 1024) ? 8 : 4;
  return c;
}

template <typename T, int MT, int NT, int NW, bool LNA>
int launch_skinny_cfg(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff, mul = (ep.act & OM_ACT_MUL_RESID) ? 1 : 0;
  const SkLn ln = {ep.a_ln32, ep.a_ln_g, ep.a_ln_b, ep.a_ln_stats_out, ep.rln32, ep.rln32_stats, ep.rln_g, ep.rln_b, ep.ln_eps};
  size_t lds = (size_t)NW * MT * NT * 4 * 64 * 4;                                   // the partial tiles
  if (LNA) lds = std::max(lds, (size_t)16 * MT * (2 * (size_t)K + 16));              // ... or the normalised rows
  if (lds > 64 * 1024) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_skinny_kernel<T, MT, NT, NW, LNA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((gemm_nt_skinny_kernel<T, MT, NT, NW, LNA>), dim3((unsigned)(N / (16 * NT)), (unsigned)((M + 16 * MT - 1) / (16 * MT))),
                     dim3(64 * NW), lds, s, (const T*)A, lda, (const T*)W, ldw, (T*)C, ldc, (int)M, (int)N, (int)K, ep.bias,
                     (const T*)ep.resid, ep.ldr, act, mul, ep.resid32, ep.out32, ln);
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int launch_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                  const GemmEpilogue& ep, hipStream_t s) {
  const SkCfg c = sk_choose(M, N, K);
  if (N % (16 * c.nt) || K % (32 * c.nw)) OM_FAIL("gemm_skinny: (NT, NW) does not divide the problem");
  const bool lna = ep.a_ln32 != nullptr;
  if (lna && (size_t)16 * c.mt * (2 * (size_t)K + 16) > 160 * 1024) OM_FAIL("gemm_skinny: pending-LayerNorm operand rows do not fit in LDS");
#define SK_GO(MT_, NT_, NW_)                                                                                              \
  if (c.mt == MT_ && c.nt == NT_ && c.nw == NW_)                                                                          \
    return lna ? launch_skinny_cfg<T, MT_, NT_, NW_, true>(A, lda, W, ldw, C, ldc, M, N, K, ep, s)                          \
               : launch_skinny_cfg<T, MT_, NT_, NW_, false>(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
  SK_GO(1, 1, 4) SK_GO(1, 1, 8) SK_GO(1, 2, 4) SK_GO(1, 2, 8) SK_GO(1, 4, 4) SK_GO(1, 4, 8)
  SK_GO(2, 1, 4) SK_GO(2, 1, 8) SK_GO(2, 2, 4) SK_GO(2, 2, 8) SK_GO(2, 4, 4) SK_GO(2, 4, 8)
#undef SK_GO
  if (lna) OM_FAIL("gemm_skinny: a pending-LayerNorm operand takes at most 32 rows per workgroup (omk_gemm_skinny_ok)");      // 64 rows of f32 per workgroup spill
#define SK_GO(MT_, NT_, NW_) \
  if (c.mt == MT_ && c.nt == NT_ && c.nw == NW_) return launch_skinny_cfg<T, MT_, NT_, NW_, false>(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
  SK_GO(4, 1, 4) SK_GO(4, 1, 8) SK_GO(4, 2, 4) SK_GO(4, 2, 8) SK_GO(4, 4, 4)
#undef SK_GO
  OM_FAIL("gemm_skinny: no kernel for this (MT, NT, NW)");
}
}  // namespace

// shapes and epilogues the kernel takes (omk_gemm asks before it picks a tile generation)
bool omk_gemm_skinny_ok(int in_dtype, int out_dtype, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep) {
  const int max_m = om_option(OM_OPT_GEMM_SKINNY_M);
  if (max_m <= 0 || M > max_m || M < 1) return false;
  if (in_dtype != out_dtype || (in_dtype != OM_BF16 && in_dtype != OM_F16)) return false;
  if (N % 16 || K % 128 || N > (int64_t)65535 * 16 || M > (int64_t)65535 * 16) return false;
  const int act = ep.act & 0xff;
  if (act != OM_ACT_NONE && act != OM_ACT_GELU_ERF && act != OM_ACT_RELU && act != OM_ACT_GELU_TANH) return false;
  if (ep.pre_act || ep.drop_p > 0.f || ep.ln_stats || ep.rln_stats || ep.stats_out || ep.resid_lo || ep.out_lo) return false;
  if ((ep.act & OM_ACT_MUL_RESID) && (!ep.resid || ep.resid32 || ep.rln32)) return false;
  if (ep.a_ln32 && (!ep.a_ln_g || !ep.a_ln_b || K % 4 || K > 1024 || sk_choose(M, N, K).mt > 2)) return false;   // LayerNorm (with shift), rows of <= 1024 elements, <= 32 rows per workgroup
  if (ep.rln32 && (!ep.rln32_stats || !ep.rln_g || !ep.rln_b)) return false;
  return true;
}

int omk_gemm_skinny(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  if (dtype == OM_F16) return launch_skinny<f16_t>(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
  return launch_skinny<bf16_t>(A, lda, W, ldw, C, ldc, M, N, K, ep, s);
}
