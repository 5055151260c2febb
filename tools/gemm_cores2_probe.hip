// Developer probe (round 5): TWO co-resident workgroups per CU -- 256 x 128 tiles, four waves of
// 128 x 64 (128 accumulator registers: the 256-register budget of two waves per SIMD), 64-byte K steps through a three-stage
// LDS ring (3 x 24 KiB = 72 KiB per workgroup, two fit the 160 KiB LDS), one tile per workgroup so that the hardware keeps a
// second workgroup's K loop on every SIMD while the first stores its tile -- against the product's ONE 256 x 256 tile stream per CU
// (kernel 7c16 through om_gemm_nt), whole launches with the plain bias epilogue, same box, interleaved, hipEvents around 20 launches.
// The question: does overlapping one stream's epilogue with another's K loop pay for the 2 x LDS fragment traffic per flop of the
// narrower wave tile and the half-line DMA requests of 64-byte K steps?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_cores2_probe.hip -Lopenmatch_amd/csrc -lopenmatch_hip -Wl,-rpath,$PWD/openmatch_amd/csrc -o build/gemm_cores2_probe
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../openmatch_amd/csrc/gemm_core.h"

#define P_ROW 64                      // bytes of K per step
#define P_A (256 * P_ROW)             // 16 KiB
#define P_B (128 * P_ROW)             // 8 KiB
#define P_STAGE (P_A + P_B)
#define P_STAGES 3
#define P_LDS (P_STAGES * P_STAGE)    // 72 KiB
#define P_ESTRIDE 136                 // staging row: 64 columns x 2 B + 8

template <typename T, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void cores2_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, T* __restrict__ C, int64_t ldc, int64_t M, int64_t N,
    int64_t K, const float* __restrict__ bias) {
  typedef typename MmaOps<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntn = N / 128;
  // eight row tiles share a sweep over the column tiles (the panels of a group stay in L2)
  const int64_t per_group = 8 * ntn, g = blockIdx.x / per_group, in_g = blockIdx.x % per_group;
  const int64_t m0 = (g * 8 + in_g % 8) * 256, n0 = (in_g / 8) * 128;
  if (m0 >= M) return;
  const char* pa[4];
  const char* pb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 16 + (lane >> 2);
    pa[i] = (const char*)(A + (m0 + r) * lda) + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (i * 4 + wave) * 16 + (lane >> 2);
    pb[i] = (const char*)(B + (n0 + r) * ldb) + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
  }
  const int nk = (int)(K * 2 / P_ROW);
  auto stage = [&](int t, char* slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + (size_t)t * P_ROW), (lptr_t)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + (size_t)t * P_ROW), (lptr_t)(slot + P_A + (i * 4 + wave) * 1024), 16, 0, 0);
  };
  const int l31 = lane & 31, half = lane >> 5, key = (lane >> 2) & 3;
  const int slot0 = (half ^ key) << 4, slot1 = ((2 | half) ^ key) << 4;
  const int rowa = (wm * 128 + l31) * P_ROW, rowb = P_A + (wn * 64 + l31) * P_ROW;
  f32x16_t acc[4][2];
  {
    f32x4_t bn[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[ni][j] = *(const f32x4_t*)(bias + n0 + wn * 64 + ni * 32 + 8 * j + 4 * half);
    stage(0, smem);
    stage(1, smem + P_STAGE);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][4 * j + e] = bn[ni][j][e];
  }
  __builtin_amdgcn_s_waitcnt(0x0076);          // vmcnt(6): stage 0 has landed
  __builtin_amdgcn_s_barrier();
  int o_cur = 0, o_nxt = P_STAGE, o_far = 2 * P_STAGE;
  frag_t a0[4], b0[2], a1[4], b1[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const frag_t*)(smem + rowa + i * 32 * P_ROW + slot0);
#pragma unroll
  for (int i = 0; i < 2; ++i) b0[i] = *(const frag_t*)(smem + rowb + i * 32 * P_ROW + slot0);
  for (int t = 0; t < nk; ++t) {
    const char* cur = smem + o_cur;
    const char* nxt = smem + o_nxt;
    const bool issue = t + 2 < nk;
    // sub-step 0: 8 MFMAs, the fragments of sub-step 1 behind the first six, the DMA of step t + 2 behind the rest
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      MmaOps<T>::mma(b0[q & 1], a0[q >> 1], acc[q >> 1][q & 1]);
      if (q < 4) a1[q] = *(const frag_t*)(cur + rowa + q * 32 * P_ROW + slot1);
      else if (q < 6) b1[q - 4] = *(const frag_t*)(cur + rowb + (q - 4) * 32 * P_ROW + slot1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (issue) stage(t + 2, smem + o_far);       // the slot of step t - 1: every wave passed the barrier that ended it
    __builtin_amdgcn_sched_barrier(0);
    if (issue) __builtin_amdgcn_s_waitcnt(0x0076); else __builtin_amdgcn_s_waitcnt(0x0070);     // step t + 1 landed; my reads done
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      MmaOps<T>::mma(b1[q & 1], a1[q >> 1], acc[q >> 1][q & 1]);
      if (t + 1 < nk) {
        if (q < 4) a0[q] = *(const frag_t*)(nxt + rowa + q * 32 * P_ROW + slot0);
        else if (q < 6) b0[q - 4] = *(const frag_t*)(nxt + rowb + (q - 4) * 32 * P_ROW + slot0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    o_cur = o_nxt;
    o_nxt = o_nxt + P_STAGE == P_LDS ? 0 : o_nxt + P_STAGE;
    o_far = o_far + P_STAGE == P_LDS ? 0 : o_far + P_STAGE;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // epilogue: per 32-row block stage bf16 through this wave's LDS region, store whole 128-byte rows
  char* reg = smem + wave * (32 * P_ESTRIDE);
  T* cbase = C + (m0 + wm * 128) * ldc + n0 + wn * 64;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint2 pk = make_uint2(Half16<T>::pack2(acc[mi][ni][4 * j], acc[mi][ni][4 * j + 1]), Half16<T>::pack2(acc[mi][ni][4 * j + 2], acc[mi][ni][4 * j + 3]));
        *(uint2*)(reg + l31 * P_ESTRIDE + (ni * 32 + 8 * j + 4 * half) * 2) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int v = it * 64 + lane, row = v >> 3, c = v & 7;
      const uint2 x = *(const uint2*)(reg + row * P_ESTRIDE + c * 16), y = *(const uint2*)(reg + row * P_ESTRIDE + c * 16 + 8);
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(u32x4{x.x, x.y, y.x, y.y}, (u32x4*)(cbase + (int64_t)(mi * 32 + row) * ldc + c * 8));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

static void fill_bf16(bf16_t* d, size_t n, float scale, unsigned long long seed, std::vector<bf16_t>* keep = nullptr) {
  std::vector<bf16_t> h(n);
  unsigned long long x = 88172645463325252ull ^ seed;
  for (auto& v : h) {
    float acc = 0;
    for (int i = 0; i < 4; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; acc += (float)(x & 0xffff) / 65536.0f - 0.5f; }
    v = f32_to_bf16(acc * 1.7f * scale);
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  if (keep) keep->swap(h);
}

template <int WPE>
static float time_cores2(int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, bf16_t* C, const float* bias) {
  // WPE = 1 (the control: the same kernel with ONE workgroup per CU): 100 KiB of LDS are requested so that a second one cannot be placed
  const size_t lds = WPE == 1 ? 100 * 1024 : P_LDS;
  hipFuncSetAttribute((const void*)cores2_kernel<bf16_t, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = (unsigned)((M / 256) * (N / 128));
  auto go = [&]() { hipLaunchKernelGGL((cores2_kernel<bf16_t, WPE>), dim3(grid), dim3(256), lds, 0, A, K, B, K, C, N, M, N, K, bias); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) go();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) go();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}
static float time_product(int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, bf16_t* C, const float* bias) {
  auto go = [&]() { if (om_gemm_nt(OM_BF16, A, K, B, K, OM_BF16, C, N, M, N, K, bias, nullptr, 0, OM_ACT_NONE, nullptr)) { fprintf(stderr, "om_gemm_nt: %s\n", om_last_error()); exit(1); } };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) go();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) go();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}

int main() {
  const int64_t M = 131072, KMAX = 3072, NMAX = 3072;
  bf16_t *A, *B, *C, *C2; float* bias;
  hipMalloc(&A, (size_t)M * KMAX * 2); hipMalloc(&B, (size_t)NMAX * KMAX * 2); hipMalloc(&C, (size_t)M * NMAX * 2); hipMalloc(&C2, (size_t)M * NMAX * 2);
  hipMalloc(&bias, NMAX * 4);
  {
    std::vector<float> hb(NMAX);
    for (int i = 0; i < NMAX; ++i) hb[i] = 0.01f * (float)(i % 17 - 8);
    hipMemcpy(bias, hb.data(), NMAX * 4, hipMemcpyHostToDevice);
  }
  {   // random operands in chunks (131072 x 3072 is 805 MB)
    const size_t chunk = (size_t)1 << 24;
    for (size_t o = 0; o < (size_t)M * KMAX; o += chunk) fill_bf16(A + o, std::min(chunk, (size_t)M * KMAX - o), 1.0f, 1 + o);
    fill_bf16(B, (size_t)NMAX * KMAX, 0.05f, 2);
  }
  struct Shape { const char* what; int64_t N, K; } shapes[] = {{"QKV", 2304, 768}, {"FFN1 (plain)", 3072, 768}, {"FFN2 (plain)", 768, 3072}};
  // correctness of the probe kernel against the product's kernel on one shape (same operands, same bias)
  {
    time_cores2<2>(M, 2304, 768, A, B, C, bias);
    time_product(M, 2304, 768, A, B, C2, bias);
    std::vector<bf16_t> h1(4096), h2(4096);
    double worst = 0;
    for (int64_t row : {0L, 257L, 65535L, 131071L}) {
      hipMemcpy(h1.data(), C + row * 2304, 2304 * 2, hipMemcpyDeviceToHost);
      hipMemcpy(h2.data(), C2 + row * 2304, 2304 * 2, hipMemcpyDeviceToHost);
      for (int i = 0; i < 2304; ++i) worst = std::max(worst, (double)fabsf(bf16_to_f32(h1[i]) - bf16_to_f32(h2[i])));
    }
    printf("CHECK probe vs product, 4 rows x 2304 columns: max |diff| %.4g %s\n", worst, worst < 0.05 ? "ok" : "MISMATCH");
  }
  for (int round = 0; round < 3; ++round)
    for (const Shape& s : shapes) {
      const float t_p = time_product(M, s.N, s.K, A, B, C2, bias);
      const float t_2 = time_cores2<2>(M, s.N, s.K, A, B, C, bias);
      const float t_1 = time_cores2<1>(M, s.N, s.K, A, B, C, bias);
      const double fl = 2.0 * M * s.N * s.K;
      printf("%-14s N=%4ld K=%4ld : product (one 256x256 stream per CU) %7.1f us %7.1f TFLOP/s | two 256x128 workgroups per CU %7.1f us %7.1f TFLOP/s | "
             "the same kernel, one workgroup per CU %7.1f us %7.1f TFLOP/s\n", s.what, (long)s.N, (long)s.K, t_p * 1e3, fl / (t_p * 1e9), t_2 * 1e3, fl / (t_2 * 1e9),
             t_1 * 1e3, fl / (t_1 * 1e9));
    }
  return 0;
}
