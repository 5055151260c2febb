// Developer tool: attributes the cycles of one K step of the v6 GEMM main loop (gemm_core6.h) by
// switching parts of it off (PROBE bits).  Operands alias one cache-hot row, so memory never stalls.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_loop_probe.hip -o build/loop_probe
#include <stdio.h>
#include <vector>
#include "../openmatch_amd/csrc/gemm_core6.h"

void om_set_error(const std::string&) {}

template <int PROBE>
__global__ __launch_bounds__(G6_THREADS) void probe(const bf16_t* A, const bf16_t* B, float* sink, long long* ticks, int64_t K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x16_t acc[4][4];
  f32x16_t zero[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) zero[i][r] = 0.f;
  const long long t0 = clock64();
  gemm_mainloop6<bf16_t, PROBE>(A, 0, B, 0, 1 << 20, 1 << 20, K, (int64_t)blockIdx.x * 256, 0, smem, acc, zero, nullptr);
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  sink[blockIdx.x * G6_THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int PROBE> static void run(const char* what, const bf16_t* A, const bf16_t* B, float* sink, long long* ticks, int64_t K, int blocks) {
  hipFuncSetAttribute((const void*)probe<PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, G6_LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<PROBE>, dim3(blocks), dim3(G6_THREADS), G6_LDS_BYTES, 0, A, B, sink, ticks, K);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<PROBE>, dim3(blocks), dim3(G6_THREADS), G6_LDS_BYTES, 0, A, B, sink, ticks, K);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  const double nk = (double)(K * 2 / 64);
  printf("%-32s blocks=%4d  %.0f ticks per K step (32 MFMA = 1024 ideal)  kernel %.3f ms -> %.2f GHz tick rate, %.0f TFLOP/s\n", what, blocks,
         sum / blocks / nk, ms, sum / blocks / (ms * 1e6), (double)blocks * 2.0 * 256 * 256 * K / (ms * 1e9));
}

int main(int argc, char** argv) {
  const bool random_data = argc > 1;   // any argument: N(0,1) bf16 operands instead of zeros (MFMA power)
  const int64_t K = 16384;
  bf16_t *A, *B; float* sink; long long* ticks;
  hipMalloc(&A, K * 2 + 4096); hipMalloc(&B, K * 2 + 4096); hipMemset(A, 0, K * 2 + 4096); hipMemset(B, 0, K * 2 + 4096);
  if (random_data) {
    std::vector<bf16_t> h(K + 2048);
    unsigned long long x = 88172645463325252ull;
    for (auto& v : h) {   // sum of 4 uniforms ~ normal enough; what matters is toggling mantissas and signs
      float acc = 0; for (int i = 0; i < 4; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; acc += (float)(x & 0xffff) / 65536.0f - 0.5f; }
      v = f32_to_bf16(acc * 1.7f);
    }
    hipMemcpy(A, h.data(), K * 2 + 4096, hipMemcpyHostToDevice);
    for (auto& v : h) v ^= 0x8000 * (&v - h.data() & 1);
    hipMemcpy(B, h.data(), K * 2 + 4096, hipMemcpyHostToDevice);
    printf("operands: random\n");
  }
  hipMalloc(&sink, 1024 * G6_THREADS * 4); hipMalloc(&ticks, 1024 * 8);
  for (int blocks : {256, 1024}) {
    run<0>("full loop", A, B, sink, ticks, K, blocks);
    run<1>("no DMA issue", A, B, sink, ticks, K, blocks);
    run<2>("no fragment reads", A, B, sink, ticks, K, blocks);
    run<3>("no DMA, no fragment reads", A, B, sink, ticks, K, blocks);
    run<4>("no barrier (results garbage)", A, B, sink, ticks, K, blocks);
    run<7>("MFMA only", A, B, sink, ticks, K, blocks);
  }
  return 0;
}
