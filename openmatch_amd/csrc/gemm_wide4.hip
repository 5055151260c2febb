// om_gemm_nt, wide tile generation 4 (see gemm_core4.h); its own translation unit so the
// epilogue specialisations of the generations compile in parallel.
#include "gemm_core4.h"
#include "gemm_epilogue.h"

// ---- v4: 256x256 tile, 4-deep ring of 64-byte K steps (gemm_core4.h) -------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(G4_THREADS) void gemm_nt_kernel4(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g4_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[4][2];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = clock64();
  gemm_mainloop4<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, tr);   // ends on a barrier
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const EpiScalars es(ep);
  const int64_t nc = n0 + wn * 64;
  const float b0 = (ep.bias && nc + (lane & 31) < N) ? ep.bias[nc + (lane & 31)] : 0.f;
  const float b1 = (ep.bias && nc + 32 + (lane & 31) < N) ? ep.bias[nc + 32 + (lane & 31)] : 0.f;
  char* region = smem + wave * (32 * PATCH_STRIDE);
#define OM_V4_CALL(A, TR)                                                                                        \
  store_patch<OutT, A, TR>(acc[0][0], acc[0][1], b0, b1, m0 + wm * 128, nc, C, ldc, M, N, ep, es, region);        \
  store_patch<OutT, A, TR>(acc[1][0], acc[1][1], b0, b1, m0 + wm * 128 + 32, nc, C, ldc, M, N, ep, es, region);   \
  store_patch<OutT, A, TR>(acc[2][0], acc[2][1], b0, b1, m0 + wm * 128 + 64, nc, C, ldc, M, N, ep, es, region);   \
  store_patch<OutT, A, TR>(acc[3][0], acc[3][1], b0, b1, m0 + wm * 128 + 96, nc, C, ldc, M, N, ep, es, region)
  OM_EPI_SWITCH(es.act, es.train, OM_V4_CALL)
#undef OM_V4_CALL
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; }
}

OM_DEFINE_LAUNCHER(launch_wide, gemm_nt_kernel4, G4_THREADS, G4_LDS_BYTES, G4_BM, G4_BN)

int omk_gemm_wide4(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
                   void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep,
                   hipStream_t s) {
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16) return launch_wide<bf16_t, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_BF16 && out_dtype == OM_F32) return launch_wide<bf16_t, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F32 && out_dtype == OM_F32) return launch_wide<float, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F16 && out_dtype == OM_F32) return launch_wide<f16_t, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  OM_FAIL("unsupported dtype combination");
}
