// Epilogue of the one-wave-per-SIMD GEMM (gemm_core6.h).  The MFMA operands are swapped there, so a
// lane owns ONE output row and every four consecutive accumulator registers are four consecutive
// output columns:
//     acc[mi][ni][r]  =  C[ mrow0 + mi*32 + (lane & 31) ][ ncol0 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3) ]
// That makes the LDS transposition cheap (tools/gemm_epilogue_probe.hip: the f32 `ds_write_b32`
// staging of the column-per-lane layout cost ~7 k cycles per 256 x 256 tile before any store):
//   * 16-bit output: add the residual (if any) in f32 in registers, pack to bf16, ds_write_b64 (4 columns),
//     rows of 264 B.  The residual patch (32 rows x 256 B) comes in through the same LDS-DMA the K loop
//     uses (8 wave instructions per patch, chunk position XOR row so the row-per-lane read-back is
//     2-way instead of 32-way conflicted), one to two patches ahead -- a single rounding and the cheap
//     bf16 staging.  Loading it as 16-byte vectors into VGPRs after the read-back cost 20 k cycles per
//     tile (tools/gemm_epilogue_probe.hip): HBM latency per patch, and `vmcnt` retiring in order ties
//     those loads to the stores in front of them.
//   * f32 output: ds_write_b128 of four f32, rows of 528 B; residual (VGPR loads) added after the read-back
// then whole 16-byte vectors are read back row-major and stored fully coalesced (a wave instruction
// covers 4 (bf16) or 2 (f32) complete 256 / 512-byte row segments).
// One patch = one mi block (32 rows x 128 columns).  With a single wave per SIMD nothing else hides
// latency, so patches are software pipelined by hand: read back patch p, write patch p+1, then
// finish patch p.  A wave's LDS operations execute in order, so reusing
// the one region needs no waits beyond the register dependencies the compiler tracks itself.
#pragma once
#include "gemm_epilogue.h"

// erf-GELU for 16-bit outputs without transcendentals, two elements per instruction
// (v_pk_mul_f32 / v_pk_fma_f32):  gelu(x) = x * (0.5 + h(xc)),  h(x) = 0.5 erf(x / sqrt 2) ~ x Q(x^2)
// evaluated at xc = clamp(x, +-4.2), Q the degree-8 minimax fit (|h error| <= 7.4e-6) scaled so that
// h(4.2) = 0.5 (1 + 2e-7): beyond the clamp the factor is 1.0000001 / -1e-7, i.e. the tails are x and 0
// to 1e-7 |x|.  |gelu error| <= 6e-5 for |x| <= 8 -- far inside a bf16 ulp.  13 instructions per two
// elements.  The f32 output path keeps erff.
__device__ __forceinline__ f32x2_t gelu_erf_poly2(f32x2_t x) {
  const f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.2f, 4.2f), __builtin_amdgcn_fmed3f(x[1], -4.2f, 4.2f)};
  const f32x2_t t = xc * xc;
  f32x2_t q = {5.998145036e-11f, 5.998145036e-11f};
#define OM_G2(K) q = __builtin_elementwise_fma(q, t, (f32x2_t){K, K})
  OM_G2(-5.633389311e-09f); OM_G2(2.343703613e-07f); OM_G2(-5.760840850e-06f); OM_G2(9.457556007e-05f);
  OM_G2(-1.114161685e-03f); OM_G2(9.830250405e-03f); OM_G2(-6.636118144e-02f); OM_G2(3.989123106e-01f);
#undef OM_G2
  return x * __builtin_elementwise_fma(xc, q, (f32x2_t){0.5f, 0.5f});
}

// gelu(x) and gelu'(x) = Phi(x) + x phi(x) from ONE evaluation of the polynomial (training forward, OM_ACT_PRE_GRAD): the
// backward's GELU' epilogue (the same polynomial + exp2 on a [tokens, 3072] tensor: 160 us per layer against ~60 for a plain
// multiply, profiles/r03_bench_v4_kernel_stats.csv) becomes a multiplication by a stored tensor.
__device__ __forceinline__ f32x2_t gelu_erf_poly2_both(f32x2_t x, f32x2_t& grad) {
  const f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.2f, 4.2f), __builtin_amdgcn_fmed3f(x[1], -4.2f, 4.2f)};
  const f32x2_t t = xc * xc;
  f32x2_t q = {5.998145036e-11f, 5.998145036e-11f};
#define OM_G2(K) q = __builtin_elementwise_fma(q, t, (f32x2_t){K, K})
  OM_G2(-5.633389311e-09f); OM_G2(2.343703613e-07f); OM_G2(-5.760840850e-06f); OM_G2(9.457556007e-05f);
  OM_G2(-1.114161685e-03f); OM_G2(9.830250405e-03f); OM_G2(-6.636118144e-02f); OM_G2(3.989123106e-01f);
#undef OM_G2
  const f32x2_t Phi = __builtin_elementwise_fma(xc, q, (f32x2_t){0.5f, 0.5f});
  const f32x2_t x2 = x * x;
  const f32x2_t pdf = {0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x2[0]),
                       0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x2[1])};
  grad = __builtin_elementwise_fma(x, pdf, Phi);
  return x * Phi;
}

// The same on EIGHT elements at once.  One Horner chain is nine dependent packed FMAs, and the compiler issues the
// chains of a patch one after the other (each instruction waits for the previous one's result: the GELU epilogue of the
// persistent GEMM ran at ~9 cycles per VALU instruction).  Vector-of-8 arithmetic expands every step into four
// independent packed instructions -- four chains in flight.
typedef float f32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x8_t gelu_erf_poly8(f32x8_t x) {
  f32x8_t xc;
#pragma unroll
  for (int i = 0; i < 8; ++i) xc[i] = __builtin_amdgcn_fmed3f(x[i], -4.2f, 4.2f);
  const f32x8_t t = xc * xc;
  f32x8_t q = 5.998145036e-11f;
#define OM_G8(K) q = __builtin_elementwise_fma(q, t, (f32x8_t)(K))
  OM_G8(-5.633389311e-09f); OM_G8(2.343703613e-07f); OM_G8(-5.760840850e-06f); OM_G8(9.457556007e-05f);
  OM_G8(-1.114161685e-03f); OM_G8(9.830250405e-03f); OM_G8(-6.636118144e-02f); OM_G8(3.989123106e-01f);
#undef OM_G8
  return x * __builtin_elementwise_fma(xc, q, (f32x8_t)(0.5f));
}

// gelu and gelu' of eight values from one evaluation of the polynomial (gelu_erf_poly2_both's arithmetic, four chains in flight):
// the two-output training epilogue of the continuous 256 x 256 kernel (gemm_wide7.h, TRAIN)
__device__ __forceinline__ f32x8_t gelu_erf_poly8_both(f32x8_t x, f32x8_t& grad) {
  f32x8_t xc;
#pragma unroll
  for (int i = 0; i < 8; ++i) xc[i] = __builtin_amdgcn_fmed3f(x[i], -4.2f, 4.2f);
  const f32x8_t t = xc * xc;
  f32x8_t q = 5.998145036e-11f;
#define OM_G8(K) q = __builtin_elementwise_fma(q, t, (f32x8_t)(K))
  OM_G8(-5.633389311e-09f); OM_G8(2.343703613e-07f); OM_G8(-5.760840850e-06f); OM_G8(9.457556007e-05f);
  OM_G8(-1.114161685e-03f); OM_G8(9.830250405e-03f); OM_G8(-6.636118144e-02f); OM_G8(3.989123106e-01f);
#undef OM_G8
  const f32x8_t Phi = __builtin_elementwise_fma(xc, q, (f32x8_t)(0.5f));
  const f32x8_t e = x * x * (f32x8_t)(-0.7213475204444817f);
  f32x8_t pdf;
#pragma unroll
  for (int i = 0; i < 8; ++i) pdf[i] = 0.3989422804014327f * __builtin_amdgcn_exp2f(e[i]);
  grad = __builtin_elementwise_fma(x, pdf, Phi);
  return x * Phi;
}

// two adjacent output elements (columns n, n+1 of row m) before the residual
// dbits: the dropout hash of the four-column group holding (m, n) (dropout_bits; N % 4 == 0), e0 = n & 3 (0 or 2)
template <int ACT, bool TRAIN, typename OutT>
__device__ __forceinline__ f32x2_t epi_pair(f32x2_t v, int64_t m, int64_t n, int64_t M, int64_t N,
                                           const GemmEpilogue& ep, const EpiScalars& es, uint64_t dbits, int e0) {
  if (ACT == OM_ACT_GELU_ERF && sizeof(OutT) == 2) {
    if (TRAIN) {
      if (ep.act & OM_ACT_PRE_GRAD) {      // the tape keeps gelu'(v): Phi(v) is the polynomial the forward evaluates anyway, + v phi(v) by exp2
        f32x2_t gr;
        v = gelu_erf_poly2_both(v, gr);
        if (ep.pre_act && m < M && n < N)
          *(uint32_t*)((OutT*)ep.pre_act + m * ep.ldp + n) = Half16<OutT>::pack2(gr[0], gr[1]);
      } else {
        if (ep.pre_act && m < M && n < N)
          *(uint32_t*)((OutT*)ep.pre_act + m * ep.ldp + n) = Half16<OutT>::pack2(v[0], v[1]);
        v = gelu_erf_poly2(v);
      }
    } else {
      v = gelu_erf_poly2(v);
    }
    if (TRAIN) {
      if (es.drop_thresh) {
        v[0] = dropout_field(dbits, e0, es.drop_thresh) ? v[0] * es.drop_scale : 0.f;
        v[1] = dropout_field(dbits, e0 + 1, es.drop_thresh) ? v[1] * es.drop_scale : 0.f;
      }
    }
    return v;
  }
  f32x2_t o = {epi_value<ACT, TRAIN, OutT>(v[0], m, n, M, N, ep, 0u, 1.f),
               epi_value<ACT, TRAIN, OutT>(v[1], m, n + 1, M, N, ep, 0u, 1.f)};
  if (TRAIN && ACT != OM_ACT_GELU_ERF_GRAD) {
    if (es.drop_thresh) {
      o[0] = dropout_field(dbits, e0, es.drop_thresh) ? o[0] * es.drop_scale : 0.f;
      o[1] = dropout_field(dbits, e0 + 1, es.drop_thresh) ? o[1] * es.drop_scale : 0.f;
    }
  }
  return o;
}

#define G6E_STRIDE16 264      // bf16 staging row: 128 columns x 2 B + 8 (ds_write_b64 / ds_read_b64 conflict-free)
#define G6E_STRIDE32 528      // f32 staging row: 128 columns x 4 B + 16 (ds_write_b128 / ds_read_b128 conflict-free)
#define G6E_REGION_BYTES (32 * G6E_STRIDE32)
#define G6E_RES_LDS_BYTES (65536 + 4 * 2 * 8192)      // LDS of a 16-bit kernel with residual: staging + residual ring (= the K ring)

// PROBE (tools/gemm_epilogue_probe.hip only): bit 0 drops the LDS writes, bit 2 the global stores.
// LNF: which fused-LayerNorm features are compiled in -- 0 none, 1 the A-operand side (GemmEpilogue::ln_*),
// 2 the output side (rln_* and stats_out); kept out of the kernels that do not use them, whose
// register allocation they would otherwise burden.
template <typename OutT, int ACT, bool TRAIN, bool RESID, int LNF = 0, int PROBE = 0>
__device__ __forceinline__ void store_wave_tile6(f32x16_t (&acc)[4][4], int64_t mrow0, int64_t ncol0, OutT* C,
                                                 int64_t ldc, int64_t M, int64_t N, const GemmEpilogue& ep,
                                                 const EpiScalars& es, char* region, const float (&rs)[4]) {
  constexpr bool RES_DMA = RESID && sizeof(OutT) == 2;      // residual through LDS-DMA, added before packing
  constexpr bool STAGE16 = sizeof(OutT) == 2;
  constexpr int STRIDE = STAGE16 ? G6E_STRIDE16 : G6E_STRIDE32;
  constexpr int SB = STAGE16 ? 2 : 4;           // staged bytes per element
  constexpr int VEC = OutVec<OutT>::VEC;        // elements per 16-byte output vector
  constexpr int CPR = 128 / VEC;                // vectors per patch row
  constexpr int RPI = 64 / CPR;                 // rows covered by one 64-lane pass
  constexpr int ITER = 32 / RPI;
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int rl = lane / CPR, c = lane % CPR;
  const int64_t rows_left64 = M - mrow0, cols_left64 = N - ncol0;
  const int rows_left = rows_left64 > 128 ? 128 : (int)rows_left64;
  const int cols_left = cols_left64 > 128 ? 128 : (int)cols_left64;
  const bool col_ok = c * VEC < cols_left;
  OutT* cp = C + (mrow0 + rl) * ldc + ncol0 + c * VEC;
  const OutT* rp = (RESID && !RES_DMA) ? (const OutT*)ep.resid + (mrow0 + rl) * ep.ldr + ncol0 + c * VEC : nullptr;   // may alias C
  // residual patches in LDS: two 8 KiB buffers per wave in the upper half of the (now idle) ring;
  // the bf16 staging regions (8448 B at wave * G6E_REGION_BYTES) all end below 64 KiB
  const int wave_ = threadIdx.x >> 6;
  char* resbuf = region - wave_ * G6E_REGION_BYTES + 65536 + wave_ * 16384;
  // Fused LayerNorm pieces (GemmEpilogue::ln_* / rln_* / stats_out; all wave-uniform switches):
  //   rs[mi]        row scale rstd_m of an A operand that is a raw pre-LayerNorm tensor (the rest of
  //                 that affine sits in the accumulators' initial value, gemm_wide6.h)
  //   ra, rc        the residual row's (rstd, -mu rstd): LN(r) = (r ra + rc) gamma_n + beta_n
  //   gbtab         this wave's 128 gamma | 128 beta (a static LDS object, so the compiler may move
  //                 its reads across the staging writes)
  __shared__ float gbtab[LNF == 2 ? 4 : 1][2][128];
  const bool ln_in = LNF == 1 && ep.ln_stats != nullptr, res_ln = LNF == 2 && RES_DMA && ep.rln_stats != nullptr;
  const bool stats_out = LNF == 2 && ep.stats_out != nullptr;
  float ra[4] = {1.f, 1.f, 1.f, 1.f}, rc[4] = {0.f, 0.f, 0.f, 0.f};
  if (res_ln) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      int64_t m = mrow0 + mi * 32 + l31; if (m > M - 1) m = M - 1;
      const float2 st = ((const float2*)ep.rln_stats)[m];
      const float mu = st.x * ep.ln_inv_h;
      const float rstd = rsqrtf(fmaxf(st.y * ep.ln_inv_h - mu * mu, 0.f) + ep.ln_eps);
      ra[mi] = rstd; rc[mi] = -mu * rstd;
    }
    const int64_t n4 = ncol0 + l31 * 4;
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
    if (n4 < N) t = *(const f32x4_t*)((half ? ep.rln_b : ep.rln_g) + n4);
    *(f32x4_t*)&gbtab[LNF == 2 ? wave_ : 0][half][l31 * 4] = t;
  }
  char* lds_wr = region + l31 * STRIDE + half * 4 * SB;
  const char* lds_rd = region + rl * STRIDE + c * VEC * SB;

  uint4 rres[2][ITER];   // residual vectors of the patch in flight and of the next one
#define G6E_ROW_OK(MI, IT) ((MI) * 32 + (IT) * RPI + rl < rows_left && col_ok)
#define G6E_WRITE(MI)                                                                                      \
  do {                                                                                                     \
    const int64_t m = mrow0 + (MI) * 32 + l31;                                                             \
    /* all 16 residual reads of the patch first: behind the staging writes each would expose its     \
       LDS latency (the compiler cannot prove the two LDS areas distinct and keeps program order) */ \
    uint2 rpatch[4][4];                                                                                    \
    f32x2_t ssum = {0.f, 0.f}, ssq = {0.f, 0.f};     /* this row's (sum, sum of squares) over my 64 columns */ \
    if (RES_DMA) {                                                                                         \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
          rpatch[ni][j] = *(const uint2*)(resbuf + ((MI) & 1) * 8192 + l31 * 256 +                         \
                                          (((ni * 4 + j) ^ (l31 & 15)) << 4) + 8 * half);                  \
    }                                                                                                      \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
        const int64_t n = ncol0 + ni * 32 + 8 * j + 4 * half;                                              \
        f32x2_t a_lo = {acc[MI][ni][4 * j], acc[MI][ni][4 * j + 1]}, a_hi = {acc[MI][ni][4 * j + 2], acc[MI][ni][4 * j + 3]};  \
        if (ln_in) { a_lo *= rs[MI]; a_hi *= rs[MI]; }                                                     \
        uint64_t dbits = 0;            /* n % 4 == 0, N % 8 == 0: one hash for the four columns */        \
        if (TRAIN) { if (es.drop_thresh) dbits = dropout_bits(ep.seed, (((ep.drop_rows && m < M) ? (uint64_t)(int64_t)ep.drop_rows[m] : (uint64_t)m) * (uint64_t)N + (uint64_t)n) >> 2); } \
        const f32x2_t lo = epi_pair<ACT, TRAIN, OutT>(a_lo, m, n, M, N, ep, es, dbits, 0);                 \
        const f32x2_t hi = epi_pair<ACT, TRAIN, OutT>(a_hi, m, n + 2, M, N, ep, es, dbits, 2);             \
        f32x2_t lo_ = lo, hi_ = hi;                                                                        \
        if (RES_DMA) {                                                                                     \
          const uint2 rr = rpatch[ni][j];                                                                  \
          float r0 = Half16<OutT>::lo(rr.x), r1 = Half16<OutT>::hi(rr.x);                                  \
          float r2 = Half16<OutT>::lo(rr.y), r3 = Half16<OutT>::hi(rr.y);                                  \
          if (res_ln) {                                                                                    \
            const f32x4_t g4 = *(const f32x4_t*)&gbtab[LNF == 2 ? wave_ : 0][0][ni * 32 + 8 * j + 4 * half];              \
            const f32x4_t b4 = *(const f32x4_t*)&gbtab[LNF == 2 ? wave_ : 0][1][ni * 32 + 8 * j + 4 * half];              \
            r0 = fmaf(fmaf(r0, ra[MI], rc[MI]), g4[0], b4[0]); r1 = fmaf(fmaf(r1, ra[MI], rc[MI]), g4[1], b4[1]); \
            r2 = fmaf(fmaf(r2, ra[MI], rc[MI]), g4[2], b4[2]); r3 = fmaf(fmaf(r3, ra[MI], rc[MI]), g4[3], b4[3]); \
          }                                                                                                \
          if (es.mul) { lo_[0] *= r0; lo_[1] *= r1; hi_[0] *= r2; hi_[1] *= r3; }                          \
          else {                                                                                           \
            lo_[0] = epi_resid<ACT, sizeof(OutT) == 2>(lo_[0], r0, false); lo_[1] = epi_resid<ACT, sizeof(OutT) == 2>(lo_[1], r1, false);        \
            hi_[0] = epi_resid<ACT, sizeof(OutT) == 2>(hi_[0], r2, false); hi_[1] = epi_resid<ACT, sizeof(OutT) == 2>(hi_[1], r3, false);        \
          }                                                                                                \
        }                                                                                                  \
        if (stats_out && n < N) {                                                                          \
          ssum += lo_ + hi_;                                                                               \
          ssq = __builtin_elementwise_fma(lo_, lo_, __builtin_elementwise_fma(hi_, hi_, ssq));             \
        }                                                                                                  \
        char* dst = lds_wr + (ni * 32 + 8 * j) * SB;                                                       \
        if (PROBE & 1) { asm volatile("" :: "v"(lo_), "v"(hi_)); }                                         \
        else if (STAGE16) *(uint2*)dst = make_uint2(Half16<OutT>::pack2(lo_[0], lo_[1]), Half16<OutT>::pack2(hi_[0], hi_[1])); \
        else *(f32x4_t*)dst = (f32x4_t){lo_[0], lo_[1], hi_[0], hi_[1]};                                   \
      }                                                                                                    \
    if (stats_out) {          /* the other half of the row sits in lane ^ 32; one atomic pair per row */  \
      float s1 = ssum[0] + ssum[1], s2 = ssq[0] + ssq[1];                                                  \
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);                                          \
      if (half == 0 && m < M) { atomicAdd(ep.stats_out + 2 * m, s1); atomicAdd(ep.stats_out + 2 * m + 1, s2); } \
    }                                                                                                      \
  } while (0)
#define G6E_RES_DMA(MI)                                                                                    \
  do {                                                                                                     \
    char* buf = resbuf + ((MI) & 1) * 8192;                                                                \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                        \
      const int row = 4 * k + (lane >> 4), src_chunk = (lane & 15) ^ (row & 15);                           \
      int64_t gr = mrow0 + (MI) * 32 + row; if (gr > M - 1) gr = M - 1;                                     \
      int64_t gc = ncol0 + src_chunk * 8; if (gc > N - 8) gc = ncol0;                                      \
      __builtin_amdgcn_global_load_lds((gptr_t)((const OutT*)ep.resid + gr * ep.ldr + gc),                  \
                                       (lptr_t)(buf + k * 1024), 16, 0, 0);                                \
    }                                                                                                      \
  } while (0)
#define G6E_RESID(MI)                                                                                      \
  do {                                                                                                     \
    _Pragma("unroll") for (int it = 0; it < ITER; ++it)                                                    \
      if (G6E_ROW_OK(MI, it)) rres[(MI) & 1][it] = *(const uint4*)(rp + (int64_t)((MI) * 32 + it * RPI) * ep.ldr); \
  } while (0)

  // Residual patches 0 and 1 are requested up front, patch p+2 as soon as WRITE(p) has consumed its
  // buffer.  vmcnt retires in order and stores are predicated, so each wait counts only the DMA
  // instructions that are certainly younger than the patch it needs (8 per patch); the inline asm
  // (memory clobber) keeps the LDS reads behind the wait.
  if (RES_DMA) {
    G6E_RES_DMA(0);
    G6E_RES_DMA(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else if (RESID) {
    G6E_RESID(0);
  }
  G6E_WRITE(0);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    // read back patch mi
    uint4 st16[STAGE16 ? ITER : 1];
    f32x4_t st32[STAGE16 ? 1 : ITER][VEC / 4];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const char* src = lds_rd + it * RPI * STRIDE;
      if (STAGE16) {
        const uint2 a = *(const uint2*)src, b = *(const uint2*)(src + 8);
        st16[it] = make_uint4(a.x, a.y, b.x, b.y);
      } else {
#pragma unroll
        for (int e = 0; e < VEC / 4; ++e) st32[it][e] = *(const f32x4_t*)(src + e * 16);
      }
    }
    if (mi + 1 < 4) {
      if (RES_DMA) {
        if (mi + 2 < 4) { G6E_RES_DMA(mi + 2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }   // younger: patch mi+2
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      G6E_WRITE(mi + 1);
      if (RESID && !RES_DMA) G6E_RESID(mi + 1);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (G6E_ROW_OK(mi, it)) {
        uint4 packed;
        if (STAGE16) {
          packed = st16[it];
        } else {
          float xv[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) xv[e] = st32[it][e >> 2][e & 3];
          if (RESID && !RES_DMA) {
            float rv[VEC];
            OutVec<OutT>::unpack(rres[mi & 1][it], rv);
            if (es.mul) {                                  // T5 gated FFN: act(wi_0 x) * (wi_1 x)
#pragma unroll
              for (int e = 0; e < VEC; ++e) xv[e] *= rv[e];
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) xv[e] = epi_resid<ACT, sizeof(OutT) == 2>(xv[e], rv[e], false);
            }
          }
          packed = OutVec<OutT>::pack(xv);
        }
        if (PROBE & 4) asm volatile("" :: "v"(packed.x), "v"(packed.y), "v"(packed.z), "v"(packed.w));
        else *(uint4*)(cp + (int64_t)(mi * 32 + it * RPI) * ldc) = packed;
      }
    }
  }
#undef G6E_ROW_OK
#undef G6E_WRITE
#undef G6E_RESID
#undef G6E_RES_DMA
}
