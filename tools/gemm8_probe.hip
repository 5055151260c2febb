// Developer probe: is an EIGHT-wave form of the generation-7 K loop (two waves per SIMD, wave tile 128 x 64, 128
// accumulator registers) faster than the four-wave one?  tools/gemm7_probe.hip showed the four-wave loop losing ~20 %
// to its own LDS-DMA issues (the issuing wave is the only MFMA source of its SIMD) and ~10 % to fragment reads.
// Same 256 x 256 tile, same five 32 KiB LDS units and rotation, same swizzle; no epilogue (MODE 0, timing: compare with
// gemm7_probe -DG7_ABL=4) or plain direct stores (MODE 1, correctness).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm8_probe.hip -o build/g8probe
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../openmatch_amd/csrc/gemm_core7.h"

void om_set_error(const std::string& s) { fprintf(stderr, "error: %s\n", s.c_str()); }

#ifndef G8_PRIO
#define G8_PRIO 0
#endif

struct G8Src { const char *a, *b; uint32_t oa[4], ob[4]; };

template <typename T, int MODE>
__global__ __launch_bounds__(512) void gemm8_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                    T* C, int64_t ldc, int64_t M, int64_t N, int64_t K) {
  typedef typename MmaOps<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t ntm = M / 256, ntn = N / 256;
  const int nk = (int)(K * 2 / G7_ROW_BYTES);
  const uint32_t ntiles = (uint32_t)(ntm * ntn);
  const int key = (lane >> 1) & 7, half = lane >> 5;
  int slot[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) slot[kk] = (((kk << 1) | half) ^ key) << 4;
  const int rowa = (wm * 128 + (lane & 31)) * G7_ROW_BYTES;
  const int rowb = (wn * 64 + (lane & 31)) * G7_ROW_BYTES;
  G8Src src;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    src.oa[i] = (uint32_t)(r * lda * 2) + c * 16;
    src.ob[i] = (uint32_t)(r * ldb * 2) + c * 16;
  }
  const uint32_t lds0 = g7_lds_addr(smem);
  auto tile_of = [&](uint32_t w, int64_t& m0, int64_t& n0) {      // row-panel major: a workgroup's consecutive tiles share A
    m0 = (int64_t)(w / (uint32_t)ntn) * 256; n0 = (int64_t)(w % (uint32_t)ntn) * 256;
  };
  uint32_t w = blockIdx.x;
  if (w >= ntiles) return;
  int64_t m0, n0;
  tile_of(w, m0, n0);
  src.a = (const char*)(A + m0 * lda); src.b = (const char*)(B + n0 * ldb);
#define G8_FILL(BASE, OFF, UNIT) _Pragma("unroll") for (int i = 0; i < 4; ++i) g7_dma(BASE, OFF[i], lds0 + (UNIT) + (i * 8 + wave) * 1024)
  G8_FILL(src.a, src.oa, 0);
  G8_FILL(src.b, src.ob, G7_UNIT_BYTES);
  for (;;) {
    if (nk > 1) { G8_FILL(src.a + G7_ROW_BYTES, src.oa, 2 * G7_UNIT_BYTES); G8_FILL(src.b + G7_ROW_BYTES, src.ob, 3 * G7_UNIT_BYTES); }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* ka = src.a + 2 * G7_ROW_BYTES;
    const char* kb = src.b + 2 * G7_ROW_BYTES;
    int u_ac = 0, u_bc = G7_UNIT_BYTES, u_an = 2 * G7_UNIT_BYTES, u_bn = 3 * G7_UNIT_BYTES, u_sp = 4 * G7_UNIT_BYTES;
    frag_t a0[4], b0[2], a1[4], b1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = *(const frag_t*)(smem + u_ac + rowa + i * 32 * G7_ROW_BYTES + slot[0]);
#pragma unroll
    for (int i = 0; i < 2; ++i) b0[i] = *(const frag_t*)(smem + u_bc + rowb + i * 32 * G7_ROW_BYTES + slot[0]);
#define G8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define G8_DMA(P, I, UNIT) g7_dma(k##P, src.o##P[I], lds0 + (UNIT) + ((I) * 8 + wave) * 1024)
#define G8_SUB(AF, BF, AN, BN, UA, UB, SLOT, DO_READ, DO_DMA, P, UNIT, DBASE, COND)                       \
  _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                         \
    MmaOps<T>::mma(BF[q & 1], AF[q >> 1], acc[q >> 1][q & 1]);                                            \
    if (q < 6 && (DO_READ)) {                                                                             \
      if (q < 4) AN[q] = *(const frag_t*)(smem + (UA) + rowa + q * 32 * G7_ROW_BYTES + (SLOT));           \
      else BN[q - 4] = *(const frag_t*)(smem + (UB) + rowb + (q - 4) * 32 * G7_ROW_BYTES + (SLOT));       \
    }                                                                                                     \
    if ((DO_DMA) && q >= 6) { if (COND) G8_DMA(P, (DBASE) + q - 6, UNIT); }                               \
    G8_FENCE();                                                                                           \
  }
#define G8_STEP(ISSUE, NEXT, B2H)                                                                         \
  do {                                                                                                    \
    { const char* const kb_cur = kb; kb -= G7_ROW_BYTES;                                                  \
      G8_SUB(a0, b0, a1, b1, u_ac, u_bc, slot[1], true, true, b, u_bn, 2, B2H)                            \
      kb = kb_cur; }                                                                                      \
    G8_SUB(a1, b1, a0, b0, u_ac, u_bc, slot[2], true, ISSUE, a, u_sp, 0, true)                            \
    G8_SUB(a0, b0, a1, b1, u_ac, u_bc, slot[3], true, ISSUE, a, u_sp, 2, true)                            \
    if (ISSUE) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_s_barrier();                                                                         \
    G8_FENCE();                                                                                           \
    G8_SUB(a1, b1, a0, b0, u_an, u_bn, slot[0], NEXT, ISSUE, b, u_ac, 0, true)                            \
    { const int o_ac = u_ac, o_bc = u_bc; u_ac = u_an; u_bc = u_bn; u_an = u_sp; u_bn = o_ac; u_sp = o_bc; } \
    ka += G7_ROW_BYTES; kb += G7_ROW_BYTES;                                                               \
  } while (0)
    if (G8_PRIO) { if (wave >> 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    int t = 0;
    for (; t + 2 < nk; ++t) G8_STEP(true, true, t > 0);
    if (t + 1 < nk) { G8_STEP(false, true, t > 0); ++t; }
    G8_STEP(false, false, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // next tile's first K step, then whatever stands in for the epilogue
    const int64_t mc = m0 + wm * 128, nc = n0 + wn * 64;
    w += gridDim.x;
    const bool has_next = w < ntiles;
    if (has_next) tile_of(w, m0, n0);
    src.a = (const char*)(A + m0 * lda); src.b = (const char*)(B + n0 * ldb);
    G8_FILL(src.a, src.oa, 0);
    G8_FILL(src.b, src.ob, G7_UNIT_BYTES);
    if (MODE == 1) {
      const int l31 = lane & 31;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            T* p = C + (mc + mi * 32 + l31) * ldc + nc + ni * 32 + 8 * j + 4 * half;
            uint2 v = make_uint2(Half16<T>::pack2(acc[mi][ni][4 * j], acc[mi][ni][4 * j + 1]),
                                 Half16<T>::pack2(acc[mi][ni][4 * j + 2], acc[mi][ni][4 * j + 3]));
            *(uint2*)p = v;
          }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" ::"a"(acc[q >> 1][q & 1]));
    }
    if (!has_next) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
  for (size_t o = 0; o < n; o += chunk + 37) (void)hipMemcpy(d + o, h.data(), std::min(chunk + 37, n - o) * 2, hipMemcpyHostToDevice);
}

template <int MODE>
static void launch(const bf16_t* A, const bf16_t* B, bf16_t* C, int64_t M, int64_t N, int64_t K) {
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)gemm8_kernel<bf16_t, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES); set = true; }
  const int64_t ntiles = (M / 256) * (N / 256);
  const int grid = (int)std::min<int64_t>(256, ntiles);
  hipLaunchKernelGGL((gemm8_kernel<bf16_t, MODE>), dim3(grid), dim3(512), G7_LDS_BYTES, 0, A, K, B, K, C, N, M, N, K);
}

static void check(int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, bf16_t* C) {
  (void)hipMemset(C, 0xff, (size_t)M * N * 2);
  launch<1>(A, B, C, M, N, K);
  (void)hipDeviceSynchronize();
  std::vector<bf16_t> a(K), b(K);
  double worst = 0; int bad = 0;
  unsigned long long x = 99;
  for (int i = 0; i < 512; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const int64_t m = (i < 8) ? (i & 1 ? M - 1 - i : i) : (int64_t)(x % (unsigned long long)M);
    const int64_t n = (int64_t)((x >> 32) % (unsigned long long)N);
    (void)hipMemcpy(a.data(), A + m * K, K * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), B + n * K, K * 2, hipMemcpyDeviceToHost);
    bf16_t c; (void)hipMemcpy(&c, C + m * N + n, 2, hipMemcpyDeviceToHost);
    double ref = 0;
    for (int64_t k = 0; k < K; ++k) ref += (double)bf16_to_f32(a[k]) * (double)bf16_to_f32(b[k]);
    const double err = fabs((double)bf16_to_f32(c) - ref), tol = 0.01 * fabs(ref) + 0.02;
    if (!(err <= tol)) ++bad;
    worst = std::max(worst, err);
  }
  printf("CHECK g8 M=%ld N=%ld K=%ld: %s (max |err| %.4f over 512 samples, %d bad)\n", (long)M, (long)N, (long)K, bad ? "FAILED" : "ok", worst, bad);
}

static void bench(const char* what, int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, bf16_t* C) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch<0>(A, B, C, M, N, K);
  const int reps = 20;
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch<0>(A, B, C, M, N, K);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("G8p%d ABL=4   %-28s M=%ld N=%ld K=%ld : %8.1f us  %7.1f TFLOP/s\n", G8_PRIO, what, (long)M, (long)N, (long)K, ms * 1e3, 2.0 * M * N * K / (ms * 1e9));
}

int main() {
  const int64_t M = 131072;
  bf16_t *A, *B, *C;
  (void)hipMalloc(&A, (size_t)M * 3072 * 2); (void)hipMalloc(&B, (size_t)3072 * 3072 * 2); (void)hipMalloc(&C, (size_t)M * 3072 * 2);
  fill_bf16(A, (size_t)M * 3072, 1.0f, 1); fill_bf16(B, (size_t)3072 * 3072, 0.05f, 2);
  check(4096, 768, 768, A, B, C); check(2048, 2304, 768, A, B, C); check(2048, 768, 3072, A, B, C);
  check(512, 512, 128, A, B, C); check(512, 256, 64, A, B, C); check(65536, 768, 192, A, B, C);
  for (int round = 0; round < 2; ++round) {
    bench("qkv (ln-folded A)", M, 2304, 768, A, B, C);
    bench("out-proj (+LN resid, stats)", M, 768, 768, A, B, C);
    bench("ffn1 shape, no gelu", M, 3072, 768, A, B, C);
    bench("ffn2 (+LN resid, stats)", M, 768, 3072, A, B, C);
    bench("plain", 32768, 3072, 3072, A, B, C);
  }
  return 0;
}
