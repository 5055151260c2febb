// Shared by gemm_wide6_bf16.hip / gemm_wide6_f32.hip (one translation unit per input type so the
// epilogue variants compile in parallel).
#pragma once
#include "gemm_core6.h"
#include "gemm_epilogue6.h"

// ---- v6: 256x256 tile, four waves of 128x128, software-pipelined fragment reads (gemm_core6.h) ----
// The epilogue variant (activation, training extras) is a KERNEL template parameter chosen on the
// host: with all variants behind one in-kernel switch hipcc spills the 256 accumulators to scratch
// at the switch (1 KiB per lane) and takes minutes to compile.
template <typename T, typename OutT, int ACT, bool TRAIN, bool RESID>
__global__ __launch_bounds__(G6_THREADS) void gemm_nt_kernel6(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g4_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[4][4];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = clock64();
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t nc = n0 + wn * 128;
  {
    f32x16_t init[4];          // the bias rides in the accumulators' initial value
    g6_bias_init(init, ep.bias, nc, N);
    gemm_mainloop6<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, init, tr);   // ends on a barrier
  }
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const EpiScalars es(ep);
  char* region = smem + wave * G6E_REGION_BYTES;
  store_wave_tile6<OutT, ACT, TRAIN, RESID>(acc, m0 + wm * 128, nc, C, ldc, M, N, ep, es, region);
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; }
}

// A persistent form (one workgroup per CU walking tiles, the next tile's first two K steps issued
// before the epilogue) was measured and dropped twice (v5, v6p: 3-8 % SLOWER): vmcnt retires in
// order, so waiting for the prefetched operands also waits for the acknowledgement of every store of
// the epilogue in front of them, which costs more than the cold start it hides.
// row tiles that sweep the column tiles together (OM_GEMM_GROUP_M overrides, for A/B measurements)
static int g6_group_m() {
  static const int v = getenv("OM_GEMM_GROUP_M") ? atoi(getenv("OM_GEMM_GROUP_M")) : 8;
  return v > 0 ? v : 8;
}

template <typename T, typename OutT, int ACT, bool TRAIN, bool RESID>
static int launch6(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t nwg = ((M + G4_BM - 1) / G4_BM) * ((N + G4_BN - 1) / G4_BN);
  if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");
  // 16-bit kernels with a residual keep three residual patches per wave in LDS next to the staging
  constexpr int lds_bytes = (RESID && sizeof(OutT) == 2 && G6E_RES_LDS_BYTES > G6_LDS_BYTES) ? G6E_RES_LDS_BYTES : G6_LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel6<T, OutT, ACT, TRAIN, RESID>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  const int tclass = sizeof(T) == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32;
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(tclass, s);
  // sweep order: 8 row tiles stay resident while the column tiles are walked (L2 reuse per XCD)
  hipLaunchKernelGGL((gemm_nt_kernel6<T, OutT, ACT, TRAIN, RESID>), dim3((unsigned)nwg), dim3(G6_THREADS), lds_bytes, s,
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
