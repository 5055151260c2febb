// Shared by gemm_wide6_bf16.hip / gemm_wide6_f32.hip (one translation unit per input type so the
// epilogue variants compile in parallel).
#pragma once
#include "gemm_core6.h"
#include "gemm_epilogue6.h"

// ---- v6: 256x256 tile, four waves of 128x128, software-pipelined fragment reads (gemm_core6.h) ----
// The epilogue variant (activation, training extras) is a KERNEL template parameter chosen on the
// host: with all variants behind one in-kernel switch hipcc spills the 256 accumulators to scratch
// at the switch (1 KiB per lane) and takes minutes to compile.
template <typename T, typename OutT, int ACT, bool TRAIN, bool RESID, int LNF = 0>
__global__ __launch_bounds__(G6_THREADS) void gemm_nt_kernel6(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g6_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[4][4];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) { tr[0] = clock64(); tr[30] = wall_clock64(); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t mc = m0 + wm * 128, nc = n0 + wn * 128;
  const char* pa[4];
  const char* pb[4];
  g6_point<T>(pa, pb, A, lda, B, ldb, M, N, m0, n0, wave, lane);
  const int nk = (int)((K * (int64_t)sizeof(T)) / G6_ROW_BYTES);
  // Initial value of the accumulators (loads issued BEFORE the first operand DMA: vmcnt retires in
  // order, so they return first and the arithmetic runs while the operands are in flight):
  //   * the bias; or
  //   * LNF == 1 and A is a raw pre-LayerNorm tensor (GemmEpilogue::ln_stats):  b'_n / rstd_m - mu_m s_n,
  //     so that the epilogue's rstd_m * acc = rstd_m (A W'^T - mu_m s_n) + b'_n.  s | b' go through a
  //     wave-private static LDS table so that only one column quad is in registers at a time.
  __shared__ float lntab[LNF == 1 ? 4 : 1][2][128];
  float rs[4] = {1.f, 1.f, 1.f, 1.f};                                  // rstd_m (row scale of the epilogue)
  const int l31 = lane & 31, half = lane >> 5;
  const bool ln_in = LNF == 1 && ep.ln_stats != nullptr;
  if (ln_in) {
    float mu[4], inv[4];
    float2 st[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      int64_t m = mc + mi * 32 + l31; if (m > M - 1) m = M - 1;
      st[mi] = ((const float2*)ep.ln_stats)[m];
    }
    const int64_t n4 = nc + l31 * 4;                    // lanes 0-31: s, lanes 32-63: b'
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
    const float* tsrc = half ? ep.bias : ep.ln_colsum;      // either may be absent (RMSNorm: no shift, no mean)
    if (tsrc && n4 < N) t = *(const f32x4_t*)(tsrc + n4);
    if (tr && threadIdx.x == 0) tr[1] = clock64();
    g6_begin(pa, pb, nk, smem, wave);
    float* tab = &lntab[LNF == 1 ? wave : 0][0][0];
    *(f32x4_t*)(tab + half * 128 + l31 * 4) = t;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      mu[mi] = ep.ln_rms ? 0.f : st[mi].x * ep.ln_inv_h;
      const float var = fmaxf(st[mi].y * ep.ln_inv_h - mu[mi] * mu[mi], 0.f) + ep.ln_eps;
      rs[mi] = rsqrtf(var);
      inv[mi] = sqrtf(var);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t s4 = *(const f32x4_t*)(tab + ni * 32 + 8 * j + 4 * half);
        const f32x4_t b4 = *(const f32x4_t*)(tab + 128 + ni * 32 + 8 * j + 4 * half);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][4 * j + e] = fmaf(-mu[mi], s4[e], b4[e] * inv[mi]);
      }
  } else {
    f32x4_t bn[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = nc + ni * 32 + 8 * j + 4 * half;
        bn[ni][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (ep.bias && n < N) bn[ni][j] = *(const f32x4_t*)(ep.bias + n);   // N % 4 == 0, 16-byte aligned (wide_ok)
      }
    if (tr && threadIdx.x == 0) tr[1] = clock64();
    g6_begin(pa, pb, nk, smem, wave);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][4 * j + e] = bn[ni][j][e];
  }
  gemm_mainloop6_run<T>(pa, pb, nk, smem, acc, tr);   // ends on a barrier
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const EpiScalars es(ep);
  char* region = smem + wave * G6E_REGION_BYTES;
  store_wave_tile6<OutT, ACT, TRAIN, RESID, LNF>(acc, mc, nc, C, ldc, M, N, ep, es, region, rs);
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; tr[31] = wall_clock64(); }
}

// A persistent form (one workgroup per CU walking tiles, the next tile's first two K steps issued
// before the epilogue) was measured and dropped twice (v5, v6p: 3-8 % SLOWER): vmcnt retires in
// order, so waiting for the prefetched operands also waits for the acknowledgement of every store of
// the epilogue in front of them, which costs more than the cold start it hides.
// row tiles that sweep the column tiles together (OM_OPT_GEMM_GROUP_M, for A/B measurements)
static int g6_group_m() {
  const int v = om_option(OM_OPT_GEMM_GROUP_M);        // (abi.cpp reads the environment once, for every option)
  return v > 0 ? v : 8;
}

template <typename T, typename OutT, int ACT, bool TRAIN, bool RESID, int LNF = 0>
static int launch6(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t nwg = ((M + G6_BM - 1) / G6_BM) * ((N + G6_BN - 1) / G6_BN);
  if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");
  // 16-bit kernels with a residual keep three residual patches per wave in LDS next to the staging
  constexpr int lds_bytes = (RESID && sizeof(OutT) == 2 && G6E_RES_LDS_BYTES > G6_LDS_BYTES) ? G6E_RES_LDS_BYTES : G6_LDS_BYTES;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel6<T, OutT, ACT, TRAIN, RESID, LNF>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  const int tclass = sizeof(T) == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32;
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(tclass, s);
  // sweep order: 8 row tiles stay resident while the column tiles are walked (L2 reuse per XCD)
  hipLaunchKernelGGL((gemm_nt_kernel6<T, OutT, ACT, TRAIN, RESID, LNF>), dim3((unsigned)nwg), dim3(G6_THREADS), lds_bytes, s,
                     (const T*)A, lda, (const T*)B, ldb, (OutT*)C, ldc, M, N, K, ep, g6_group_m());
  if (timing) om_timing_end(tclass, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

// The (activation, training, residual) combinations the callers use, for one dtype pair; anything
// else reports "no wide kernel" and stays on the older generations.
static bool launch6_has(int act, bool resid) {
  switch (act) {
    case OM_ACT_NONE: case OM_ACT_GELU_TANH: return true;
    case OM_ACT_GELU_ERF: case OM_ACT_RELU: return !resid;
    case OM_ACT_GELU_ERF_GRAD: return resid;
  }
  return false;
}

template <typename T, typename OutT>
static int launch6_any(int act, bool train, bool resid, const void* A, int64_t lda, const void* B, int64_t ldb,
                       void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep,
                       hipStream_t s) {
#define OM_L6(A_, R_)                                                                          \
  do {                                                                                         \
    if (train) return launch6<T, OutT, A_, true, R_>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);  \
    return launch6<T, OutT, A_, false, R_>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);            \
  } while (0)
  switch (act) {
    case OM_ACT_NONE:          if (resid) OM_L6(OM_ACT_NONE, true); else OM_L6(OM_ACT_NONE, false);
    case OM_ACT_GELU_TANH:     if (resid) OM_L6(OM_ACT_GELU_TANH, true); else OM_L6(OM_ACT_GELU_TANH, false);
    case OM_ACT_GELU_ERF:      if (!resid) OM_L6(OM_ACT_GELU_ERF, false); break;
    case OM_ACT_RELU:          if (!resid) OM_L6(OM_ACT_RELU, false); break;
    case OM_ACT_GELU_ERF_GRAD: if (resid) OM_L6(OM_ACT_GELU_ERF_GRAD, true); break;
  }
#undef OM_L6
  OM_FAIL("no wide kernel for this epilogue combination");
}
