// Developer probe: what does a global_store_dwordx4 cost the wave that issues it, and does VALU work between two
// stores hide it?  (Round 4: the GELU epilogue of the persistent GEMM = plain epilogue + polynomial, cycle for cycle,
// although one store sits behind every ~300 cycles of VALU work -- tile traces profiles/r04_probe2_*.)
//   one workgroup of 1 / 2 / 4 waves per CU (one wave per SIMD), every wave: R rounds of [V packed FMAs, one store of 1 KiB]
//   V = 0, 16, 32, 64, 128; whole 128-byte lines (8 rows x 128 B per instruction, rows `ld` bytes apart) as the GEMM writes them
//   reported: shader cycles per round per wave (s_memtime), and the same with the store compiled out
// hipcc -O3 --offload-arch=gfx950 tools/store_issue_probe.hip -o build/store_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int V, bool STORE, bool NT>
__global__ __launch_bounds__(256) void probe(char* out, size_t ld, int rounds, unsigned long long* cyc, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave w of block b owns rows [ (b * 4 + w) * 8 * rounds, ... ): every round 8 new rows x 128 B
  char* base = out + ((size_t)blockIdx.x * 4 + wave) * 8 * (size_t)rounds * ld + (size_t)(lane >> 3) * ld + (lane & 7) * 16;
  f32x2 a0 = {seed, seed + 1.f}, a1 = {seed + 2.f, seed + 3.f}, a2 = {seed + 4.f, seed + 5.f}, a3 = {seed + 6.f, seed + 7.f};
  const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, -0.5f};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int v = 0; v < V; v += 4) {
      a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c);
      a2 = __builtin_elementwise_fma(a2, m, c); a3 = __builtin_elementwise_fma(a3, m, c);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (STORE) {
      const u32x4 d = {__float_as_uint(a0[0]), __float_as_uint(a1[0]), __float_as_uint(a2[0]), __float_as_uint(a3[0])};
      if (NT) __builtin_nontemporal_store(d, (u32x4*)(base + (size_t)r * 8 * ld));
      else *(u32x4*)(base + (size_t)r * 8 * ld) = d;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  if (a0[0] + a1[0] + a2[0] + a3[0] == 12345.678f) out[0] = 1;     // keep the chains alive without a store
}

template <int V, bool STORE, bool NT>
static double run(char* out, size_t ld, int rounds, int waves, unsigned long long* dcyc, int blocks) {
  hipLaunchKernelGGL((probe<V, STORE, NT>), dim3(blocks), dim3(64 * waves), 0, 0, out, ld, rounds, dcyc, 1.0f);
  hipLaunchKernelGGL((probe<V, STORE, NT>), dim3(blocks), dim3(64 * waves), 0, 0, out, ld, rounds, dcyc, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { s += (double)h[b * 4 + w]; ++n; }
  return s / n / rounds;
}

int main() {
  const int rounds = 64;
  const size_t ld = 6144;                                  // FFN1's output row: 3072 x 2 bytes
  const int blocks = 256;
  char* out; unsigned long long* dcyc;
  hipMalloc(&out, (size_t)blocks * 4 * 8 * rounds * ld + 4096); hipMalloc(&dcyc, blocks * 4 * 8);
  hipMemset(dcyc, 0, blocks * 4 * 8);
  printf("cycles per round per wave: [V packed FMAs + one 1 KiB store]; 256 workgroups (one per CU), 64 rounds\n");
#define ROW(V)                                                                                                  \
  for (int waves : {1, 2, 4})                                                                                   \
    printf("V=%-3d waves/CU=%d : VALU only %7.1f | + store %7.1f | + nt store %7.1f\n", V, waves,                 \
           run<V, false, false>(out, ld, rounds, waves, dcyc, blocks), run<V, true, false>(out, ld, rounds, waves, dcyc, blocks), \
           run<V, true, true>(out, ld, rounds, waves, dcyc, blocks));
  ROW(0) ROW(16) ROW(32) ROW(64) ROW(128) ROW(256)
  // the same with only 8 CUs busy (is the cost the CU's or the chip's?)
  printf("8 workgroups only:\n");
  for (int waves : {1, 4})
    printf("V=64  waves/CU=%d : VALU only %7.1f | + store %7.1f | + nt store %7.1f\n", waves, run<64, false, false>(out, ld, rounds, waves, dcyc, 8),
           run<64, true, false>(out, ld, rounds, waves, dcyc, 8), run<64, true, true>(out, ld, rounds, waves, dcyc, 8));
  return 0;
}
