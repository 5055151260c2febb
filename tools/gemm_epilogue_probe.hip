// Developer tool: attributes the cycles of the v6 GEMM epilogue (store_wave_tile in gemm_epilogue.h)
// by switching parts of it off.  Each workgroup stores its own 256 x 256 bf16 tile `reps` times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_epilogue_probe.hip -o build/epilogue_probe
#include <stdio.h>
#include <vector>
#include "../openmatch_amd/csrc/gemm_core6.h"
#include "../openmatch_amd/csrc/gemm_epilogue6.h"

void om_set_error(const std::string&) {}
bool om_timing_on() { return false; }
void om_timing_begin(int, hipStream_t) {}
void om_timing_end(int, hipStream_t, double) {}

template <int PROBE, int ACT, bool RESID>
__global__ __launch_bounds__(G6_THREADS) void probe(bf16_t* C, const bf16_t* R, int64_t ldc, int64_t M, int64_t N, long long* ticks, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16_t acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.01f * (float)(i * 64 + j * 16 + r) + 0.001f * lane;
  GemmEpilogue ep = {};
  ep.act = ACT; ep.resid = RESID ? R : nullptr; ep.ldr = ldc;
  const EpiScalars es(ep);
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  const int64_t ntn = N / 256;
  const int64_t m0 = (int64_t)(blockIdx.x / ntn) * 256, n0 = (int64_t)(blockIdx.x % ntn) * 256;
  char* region = smem + wave * G6E_REGION_BYTES;
  __syncthreads();
  const long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    store_wave_tile6<bf16_t, ACT, false, RESID, 0, PROBE>(acc, m0 + wm * 128, n0 + wn * 128, C, ldc, M, N, ep, es, region, one);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int PROBE, int ACT, bool RESID> static void run(const char* what, bf16_t* C, const bf16_t* R, int64_t M, int64_t N, long long* ticks, int blocks) {
  const int reps = 8;
  hipFuncSetAttribute((const void*)probe<PROBE, ACT, RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, G6E_RES_LDS_BYTES);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<PROBE, ACT, RESID>), dim3(blocks), dim3(G6_THREADS), G6E_RES_LDS_BYTES, 0, C, R, N, M, N, ticks, reps);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  printf("%-40s blocks=%4d  %.0f ticks per 256x256 tile\n", what, blocks, sum / blocks / reps);
}

int main() {
  const int64_t N = 3072, M = 256 * 1024 / (N / 256) ;   // 1024 tiles
  bf16_t *C, *R; long long* ticks;
  hipMalloc(&C, M * N * 2); hipMalloc(&R, M * N * 2); hipMemset(R, 0, M * N * 2); hipMalloc(&ticks, 4096 * 8);
  for (int blocks : {1, 256}) {
    run<0, 0, false>("act none: full", C, R, M, N, ticks, blocks);
    run<1, 0, false>("act none: no LDS writes", C, R, M, N, ticks, blocks);
    run<4, 0, false>("act none: no global stores", C, R, M, N, ticks, blocks);
    run<0, 0, true>("act none + resid: full", C, R, M, N, ticks, blocks);
    run<4, 0, true>("act none + resid: no global stores", C, R, M, N, ticks, blocks);
    run<0, 1, false>("gelu erf: full", C, R, M, N, ticks, blocks);
    run<4, 1, false>("gelu erf: no global stores", C, R, M, N, ticks, blocks);
  }
  return 0;
}
