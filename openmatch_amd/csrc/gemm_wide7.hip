// om_gemm_nt, tile generation 7 (gemm_core7.h): 256 x 256 tile, four waves, 128-byte K steps, 16-bit inputs.
// Stage A of the generation: the K loop on the v6 epilogue, one tile per workgroup, selected with
// om_debug_gemm_gen(7 | 71) for A/B measurements against generation 6 (71 = alternate DMA placement).
#include "gemm_core7.h"
#include "gemm_epilogue6.h"

template <typename T, typename OutT, int ACT, bool RESID, int BPOS>
__global__ __launch_bounds__(G6_THREADS) void gemm_nt_kernel7a(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g4_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[4][4];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) { tr[0] = clock64(); tr[30] = wall_clock64(); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t mc = m0 + wm * 128, nc = n0 + wn * 128;
  G7Src src;
  g7_point<T>(src, A, lda, B, ldb, M, N, m0, n0, wave, lane);
  const int nk = (int)((K * (int64_t)sizeof(T)) / G7_ROW_BYTES);
  const int half = lane >> 5;
  f32x4_t bn[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = nc + ni * 32 + 8 * j + 4 * half;
      bn[ni][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (ep.bias && n < N) bn[ni][j] = *(const f32x4_t*)(ep.bias + n);
    }
  if (tr && threadIdx.x == 0) tr[1] = clock64();
  g7_begin(src, nk, smem, wave);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][4 * j + e] = bn[ni][j][e];
  gemm_mainloop7_run<T, BPOS>(src, nk, smem, acc, tr);   // ends on a barrier
  if (tr && threadIdx.x == 0) tr[15] = clock64();
  const EpiScalars es(ep);
  char* region = smem + wave * G6E_REGION_BYTES;
  const float rs[4] = {1.f, 1.f, 1.f, 1.f};
  store_wave_tile6<OutT, ACT, false, RESID, 0>(acc, mc, nc, C, ldc, M, N, ep, es, region, rs);
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; tr[31] = wall_clock64(); }
}

template <typename T, typename OutT, int ACT, bool RESID, int BPOS>
static int launch7a(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int64_t nwg = ((M + G4_BM - 1) / G4_BM) * ((N + G4_BN - 1) / G4_BN);
  if (nwg > 0x7fffffffLL) OM_FAIL("grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel7a<T, OutT, ACT, RESID, BPOS>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    attr_set = true;
  }
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(OM_TIMING_GEMM_BF16, s);
  hipLaunchKernelGGL((gemm_nt_kernel7a<T, OutT, ACT, RESID, BPOS>), dim3((unsigned)nwg), dim3(G6_THREADS), G7_LDS_BYTES, s,
                     (const T*)A, lda, (const T*)B, ldb, (OutT*)C, ldc, M, N, K, ep, 8);
  if (timing) om_timing_end(OM_TIMING_GEMM_BF16, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

// bf16 -> bf16 inference epilogues without fused LayerNorm (the stage-A measurement set)
bool omk_gemm_wide7a_has(int in_dtype, int out_dtype, int act, bool train, bool resid, int64_t K, const GemmEpilogue& ep) {
  if (in_dtype != OM_BF16 || out_dtype != OM_BF16 || train || (K * 2) % G7_ROW_BYTES) return false;
  if (ep.ln_stats || ep.rln_stats || ep.stats_out) return false;
  return (act == OM_ACT_NONE) || (act == OM_ACT_GELU_ERF && !resid);
}

int omk_gemm_wide7a(int bpos, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s) {
  const int act = ep.act & 0xff;
  const bool resid = ep.resid != nullptr;
#define OM_L7(A_, R_)                                                                                     \
  do {                                                                                                    \
    if (bpos) return launch7a<bf16_t, bf16_t, A_, R_, 1>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);         \
    return launch7a<bf16_t, bf16_t, A_, R_, 0>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                   \
  } while (0)
  if (act == OM_ACT_NONE && !resid) OM_L7(OM_ACT_NONE, false);
  if (act == OM_ACT_NONE && resid) OM_L7(OM_ACT_NONE, true);
  if (act == OM_ACT_GELU_ERF && !resid) OM_L7(OM_ACT_GELU_ERF, false);
#undef OM_L7
  OM_FAIL("no generation-7 kernel for this epilogue");
}
