// Developer probe (round 4): the branch-free continuous-ring K loop on 32 x 32 x 16 against 16 x 16 x 32 MFMAs, K loop + tile
// walk only (no epilogue: one accumulator word per lane is stored so that nothing is optimised away), random operands,
// hipEvents around 20 launches.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DG7_CONT16_PROBE tools/gemm7c16_probe.hip
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <atomic>
#include <algorithm>
#include "../openmatch_amd/csrc/gemm_wide7.h"

void om_set_error(const std::string& s) { fprintf(stderr, "error: %s\n", s.c_str()); }
bool om_timing_on() { return false; }
void om_timing_begin(int, hipStream_t) {}
void om_timing_end(int, hipStream_t, double) {}
int om_option(int o) { return o == OM_OPT_GEMM_CONT ? 7 : (o == OM_OPT_GEMM_MAX_GRID ? 0 : 8); }

template <typename T, bool M16>
__global__ __launch_bounds__(G6_THREADS) void kloop_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                            float* __restrict__ out, int64_t M, int64_t N, int64_t K, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntm = M / 256, ntn = N / 256;
  const int nk = (int)((K * 2) / G7_ROW_BYTES);
  int it = 0;
  int64_t m0, n0;
  if (!g7_tile(0, ntm, ntn, group_m, m0, n0)) return;
  G7SrcU src;
  g7_offsets_u<T>(src, lda, ldb, wave, lane0);
  G7Ring ring;
  g7_ring_reset(ring);
  const char* cur_a = (const char*)(A + m0 * lda);
  const char* cur_b = (const char*)(B + n0 * ldb);
  g7_fill_a(src, cur_a, smem + ring.ac, wave);
  g7_fill_b(src, cur_b, smem + ring.bc, wave);
  g7_fill_a(src, cur_a + G7_ROW_BYTES, smem + ring.an, wave);
  if (M16) g7_fill_b(src, cur_b + G7_ROW_BYTES, smem + ring.bn, wave);
  else {
#pragma unroll
    for (int i = 0; i < 4; ++i) g7_issue_b(src, cur_b + G7_ROW_BYTES, i, g7_lds_addr(smem + ring.bn) + (i * 4 + wave) * 1024);
  }
  G7_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
  float keep = 0.f;
  for (;;) {
    ++it;
    int64_t m1 = m0, n1 = n0;
    const bool has_next = g7_tile(it, ntm, ntn, group_m, m1, n1);
    const char* const next_a = (const char*)(A + m1 * lda);
    const char* const next_b = (const char*)(B + n1 * ldb);
    if (M16) {
      f32x4_t acc[8][8];
      const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 64; ++q) acc[q >> 3][q & 7] = z4;
      gemm_mainloop7_cont16<T>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc, nullptr);
#pragma unroll
      for (int q = 0; q < 64; ++q) { asm volatile("" : "+a"(acc[q >> 3][q & 7])); keep += acc[q >> 3][q & 7][0]; }
    } else {
      f32x16_t acc[4][4];
      const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q >> 2][q & 3] = z;
      gemm_mainloop7_cont<T>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc);
#pragma unroll
      for (int q = 0; q < 16; ++q) { asm volatile("" : "+a"(acc[q >> 2][q & 3])); keep += acc[q >> 2][q & 3][0] + acc[q >> 2][q & 3][5]; }
    }
    if (!has_next) break;
    cur_a = next_a; cur_b = next_b; m0 = m1; n0 = n1;
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = keep;
  G7_WAIT_VM(0);
}

static void fill_bf16(bf16_t* d, size_t n, float scale, unsigned long long seed) {
  const size_t chunk = std::min<size_t>(n, (size_t)1 << 22);
  std::vector<bf16_t> h(chunk + 37);
  unsigned long long x = 88172645463325252ull ^ seed;
  for (auto& v : h) {
    float acc = 0;
    for (int i = 0; i < 4; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; acc += (float)(x & 0xffff) / 65536.0f - 0.5f; }
    v = f32_to_bf16(acc * 1.7f * scale);
  }
  for (size_t o = 0; o < n; o += chunk + 37) hipMemcpy(d + o, h.data(), std::min(chunk + 37, n - o) * 2, hipMemcpyHostToDevice);
}

template <bool M16>
static void run(const char* what, int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, float* out) {
  hipFuncSetAttribute((const void*)kloop_kernel<bf16_t, M16>, hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() { hipLaunchKernelGGL((kloop_kernel<bf16_t, M16>), dim3(256), dim3(G6_THREADS), G7_LDS_BYTES, 0, A, K, B, K, out, M, N, K, 8); };
  for (int i = 0; i < 3; ++i) go();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) go();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-10s %-26s M=%ld N=%ld K=%ld : %8.1f us  %7.1f TFLOP/s (K loop + tile walk only)\n", M16 ? "16x16x32" : "32x32x16", what, (long)M, (long)N, (long)K, ms * 1e3,
         2.0 * M * N * K / (ms * 1e9));
}

int main() {
  const int64_t M = 131072;
  bf16_t *A, *B; float* out;
  hipMalloc(&A, (size_t)M * 3072 * 2); hipMalloc(&B, (size_t)3072 * 3072 * 2); hipMalloc(&out, 256 * 256 * 4);
  fill_bf16(A, (size_t)M * 3072, 1.0f, 1); fill_bf16(B, (size_t)3072 * 3072, 0.05f, 2);
  for (int round = 0; round < 3; ++round) {
    run<false>("qkv shape", M, 2304, 768, A, B, out);   run<true>("qkv shape", M, 2304, 768, A, B, out);
    run<false>("ffn1 shape", M, 3072, 768, A, B, out);  run<true>("ffn1 shape", M, 3072, 768, A, B, out);
    run<false>("ffn2 shape", M, 768, 3072, A, B, out);  run<true>("ffn2 shape", M, 768, 3072, A, B, out);
    run<false>("K = 3072 square-ish", 32768, 3072, 3072, A, B, out); run<true>("K = 3072 square-ish", 32768, 3072, 3072, A, B, out);
  }
  return 0;
}
