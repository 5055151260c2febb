// Developer tool: sustained matrix-core rate under the chip's power budget, MFMA-only loops on register operands
// (no LDS, no memory in the loop), 32x32x16 vs 16x16x32 bf16, random vs zero operands.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_power_probe.hip -o build/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void k32(const uint4* __restrict__ src, float* out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 u = src[(threadIdx.x + 256 * i) & 4095], v = src[(threadIdx.x + 256 * (i + 4)) & 4095];
    a[i] = __builtin_bit_cast(bf16x8, u); b[i] = __builtin_bit_cast(bf16x8, v);
  }
  f32x16 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(const uint4* __restrict__ src, float* out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint4 u = src[(threadIdx.x + 256 * i) & 4095], v = src[(threadIdx.x + 256 * (i + 8)) & 4095];
    a[i] = __builtin_bit_cast(bf16x8, u); b[i] = __builtin_bit_cast(bf16x8, v);
  }
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  uint4* src; float* out;
  hipMalloc(&src, 4096 * 16); hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (int zero = 0; zero < 2; ++zero) {
      std::vector<unsigned short> h(4096 * 8);
      unsigned long long x = 88172645463325252ull;
      for (auto& v : h) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        float f = ((float)(x & 0xffff) / 65536.0f - 0.5f) * 2.0f;
        unsigned u; memcpy(&u, &f, 4);
        v = zero ? 0 : (unsigned short)(u >> 16);
      }
      hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
      for (int shape = 0; shape < 2; ++shape) {
        const int iters = shape == 0 ? 60000 : 30000;                 // ~equal flops: 16 x 32x32x16 vs 64 x 16x16x32 per iteration
        const double flop = (double)wgs * 4 * iters * (shape == 0 ? 16.0 * 32 * 32 * 16 * 2 : 64.0 * 16 * 16 * 32 * 2);
        for (int l = 0; l < 4; ++l) {
          hipEventRecord(e0);
          if (shape == 0) hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, src, out, iters);
          else hipLaunchKernelGGL(k16, dim3(wgs), dim3(256), 0, 0, src, out, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          if (l > 0) printf("%s %s launch %d: %8.2f ms  %7.1f TFLOP/s\n", shape == 0 ? "32x32x16" : "16x16x32", zero ? "zeros " : "random", l, ms, flop / ms * 1e-9);
        }
      }
    }
  return 0;
}
