// Developer probe (round 4, late): the continuous-ring K loop on 16 x 16 x 32 MFMAs over 128 x 256 tiles (TIN = 4: four 16-row
// blocks per wave, the A operand in the first half of its unit) against 256 x 256 tiles (TIN = 8), at the TRAINING step's token
// count -- M = 9 216 rows is 36 row blocks of 256: 108 tiles at N = 768 on 256 CUs.  K loop + tile walk only (no epilogue),
// random operands, hipEvents around 20 launches.  The question: is a half-height tile's K loop fast enough (its prefetch distance
// is two steps of ~1.4 k cycles) to replace the 256 x 128 generation (kernel 2: ~2.7 k cycles per step, 70 / 52 / 18 / 50 us on the
// four shapes below)?     hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm7h_probe.hip -o build/g7h_probe
// Also the round's K-step ablation: -DG7H_ABL=<bits> compiles parts of the loop out (bit 0 DMA issues, 1 fragment reads, 2 barrier,
// 3 vmcnt wait, 4 scheduling fences, 5 every second MFMA), -DG7_DMA_EARLY=n front-loads a sub-step's DMA issues, -DG7_DMA_FORM=0|1|2
// selects the LDS-DMA helper's form (gemm_core7.h).  `build/g7h_probe [M]`, recipe and results: tools/r4/probe17.sh,
// profiles/r04_probe17_* / r04_probe18_*.
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
int om_option(int o) { return o == OM_OPT_GEMM_CONT ? 15 : (o == OM_OPT_GEMM_MAX_GRID ? 0 : 8); }

// ---- gemm_mainloop7_cont16 of gemm_core7.h with the tile height as a parameter (the library's loop is TIN = 8 only) -------------
// TIN: 16-row blocks per wave -- 8 (a 256 x 256 tile) or 4 (a 128 x 256 tile: the A operand fills the first half of its unit, four
// DMA instructions per wave and step instead of eight).
// G7_DMA_EARLY (probe): the DMA issues of a sub-step sit behind its FIRST MFMAs, one per G7_DMA_EARLY MFMAs, instead of evenly
// over the sub-step (0: evenly) -- the last issue of B(t+2) then has most of a step to land before the next mid-step barrier
#ifndef G7_DMA_EARLY
#define G7_DMA_EARLY 0
#endif
#define G7_DMA_EARLY_ (G7_DMA_EARLY)
template <typename T, bool TAIL = false, typename TailFn = G7NoTail, int TIN = 8>
__device__ __forceinline__ void gemm_mainloop7_cont16h(const G7SrcU& src, const char* cur_a, const char* cur_b,
                                                      const char* next_a, const char* next_b, int nk, char* smem, G7Ring& ring,
                                                      f32x4_t (&acc)[TIN][8], unsigned long long* tr = nullptr, TailFn tail = TailFn()) {
  typedef typename MmaOps<T>::frag_t frag_t;
  static_assert(sizeof(T) == 2, "128-byte K steps: 16-bit operands only");
  static_assert(TIN == 8 || (TIN == 4 && !TAIL), "256-row tiles, or 128-row tiles without a tail hook");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 1) & 7;           // == ((row >> 1) & 7) for row = 16*x + (lane & 15)
  const int kb4 = lane >> 4;
  int slot[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) slot[kk] = (((kk << 2) | kb4) ^ key) << 4;
  const int rowa = (wm * (TIN * 16) + (lane & 15)) * G7_ROW_BYTES;
  const int rowb = (wn * 128 + (lane & 15)) * G7_ROW_BYTES;
  const uint32_t lds0 = g7_lds_addr(smem);
  const char* ka = cur_a + 2 * G7_ROW_BYTES;
  const char* kb = cur_b + 2 * G7_ROW_BYTES;
  int u_ac = ring.ac, u_bc = ring.bc, u_an = ring.an, u_bn = ring.bn, u_sp = ring.sp;
  frag_t a0[TIN], b0[8], a1[TIN], b1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b0[i] = *(const frag_t*)(smem + u_bc + rowb + i * 16 * G7_ROW_BYTES + slot[0]);
#pragma unroll
  for (int i = 0; i < TIN; ++i) a0[i] = *(const frag_t*)(smem + u_ac + rowa + i * 16 * G7_ROW_BYTES + slot[0]);
  // G7H_ABL (timing probes; results are garbage): bit 0 no DMA issues, bit 1 no fragment reads (the first step's are reused),
  // bit 2 no barrier, bit 3 no vmcnt wait, bit 4 no scheduling fences, bit 5 every second MFMA dropped
#ifndef G7H_ABL
#define G7H_ABL 0
#endif
#define G7_FENCE() do { if (!(G7H_ABL & 16)) __builtin_amdgcn_sched_barrier(0); } while (0)
  // TIN * 8 MFMAs from (AF, BF); sixteen of them each cover one fragment read into (BN, then AN) from (UA, UB) chunk SLOT (every
  // fourth MFMA of 64, every second of 32); NISS of them each cover one DMA issue of operand P (PTR, instruction index) into UNIT
  // -- or, in the last step of a TAIL loop, tail(TBASE + index)
#define G7C_SUB16(AF, BF, AN, BN, UA, UB, SLOT, P, PTR, UNIT, TBASE, LASTSTEP, NISS)                     \
  _Pragma("unroll") for (int q = 0; q < TIN * 8; ++q) {                                                  \
    constexpr int RSTR = TIN / 2, ISTR = TIN * 8 / (NISS);      /* MFMAs per fragment read / per DMA issue */ \
    if (!(G7H_ABL & 32) || !(q & 1)) Mma16c<T>::mma(BF[q & 7], AF[q >> 3], acc[q >> 3][q & 7]);          \
    if (!(G7H_ABL & 2) && q % RSTR == 0) {                                                               \
      const int r_ = q / RSTR;                                                                           \
      if (r_ < 8) BN[r_] = *(const frag_t*)(smem + (UB) + rowb + r_ * 16 * G7_ROW_BYTES + (SLOT));       \
      else if (r_ < 8 + TIN) AN[r_ - 8] = *(const frag_t*)(smem + (UA) + rowa + (r_ - 8) * 16 * G7_ROW_BYTES + (SLOT)); \
    }                                                                                                    \
    if (!(G7H_ABL & 1) && (G7_DMA_EARLY_ ? (q < (NISS) * G7_DMA_EARLY_ && q % G7_DMA_EARLY_ == G7_DMA_EARLY_ - 1) : (q % ISTR == (ISTR == 8 ? 5 : 1)))) { \
      const int di_ = G7_DMA_EARLY_ ? q / (G7_DMA_EARLY_ ? G7_DMA_EARLY_ : 1) : q / ISTR;                \
      if (LASTSTEP) tail((TBASE) + di_, u_sp, u_ac, u_bc);                                                   \
      else g7_issue_##P(src, PTR, di_, lds0 + (UNIT) + (di_ * 4 + wave) * 1024);                         \
    }                                                                                                    \
    G7_FENCE();                                                                                          \
  }
#define G7C_STEP16(LASTSTEP)                                                                             \
  do {                                                                                                   \
    if (tr && tid == 0 && t < 12) tr[3 + t] = clock64();                                                 \
    G7C_SUB16(a0, b0, a1, b1, u_ac, u_bc, slot[1], a, ka, u_sp, 0, LASTSTEP, TIN)                        \
    if (G7H_ABL & 8) __builtin_amdgcn_s_waitcnt(0xC07F); else __builtin_amdgcn_s_waitcnt(0x0070 | TIN);   /* vmcnt(TIN) lgkmcnt(0): all but A(t+2) */ \
    if (!(G7H_ABL & 4)) __builtin_amdgcn_s_barrier();                                                    \
    G7_FENCE();                                                                                          \
    G7C_SUB16(a1, b1, a0, b0, u_an, u_bn, slot[0], b, kb, u_ac, 8, LASTSTEP, 8)                          \
    { const int o_ac = u_ac, o_bc = u_bc; u_ac = u_an; u_bc = u_bn; u_an = u_sp; u_bn = o_ac; u_sp = o_bc; } \
    if (t + 3 == nk) { ka = next_a; kb = next_b; } else { ka += G7_ROW_BYTES; kb += G7_ROW_BYTES; }      \
  } while (0)
  int t = 0;
  const int nplain = TAIL ? nk - 1 : nk;
  for (; t < nplain; ++t) G7C_STEP16(false);
  if (TAIL) G7C_STEP16(true);
#undef G7C_STEP16
#undef G7C_SUB16
#undef G7_FENCE
  ring.ac = u_ac; ring.bc = u_bc; ring.an = u_an; ring.bn = u_bn; ring.sp = u_sp;
}



template <typename T, int TIN>
__global__ __launch_bounds__(G6_THREADS) void kloop_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                            float* __restrict__ out, int64_t M, int64_t N, int64_t K, int group_m,
                                                            unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = TIN * 32;
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntm = M / BM, ntn = N / 256;
  const int nk = (int)((K * 2) / G7_ROW_BYTES);
  int it = 0;
  int64_t m0, n0;
  if (!g7_tile(0, ntm, ntn, group_m, m0, n0)) return;
  m0 = m0 / 256 * BM;
  G7SrcU src;
  g7_offsets_u<T>(src, lda, ldb, wave, lane0);
  G7Ring ring;
  g7_ring_reset(ring);
  const char* cur_a = (const char*)(A + m0 * lda);
  const char* cur_b = (const char*)(B + n0 * ldb);
  const uint32_t l0 = g7_lds_addr(smem);
#pragma unroll
  for (int i = 0; i < TIN; ++i) g7_issue_a(src, cur_a, i, l0 + ring.ac + (i * 4 + wave) * 1024);
  g7_fill_b(src, cur_b, smem + ring.bc, wave);
#pragma unroll
  for (int i = 0; i < TIN; ++i) g7_issue_a(src, cur_a + G7_ROW_BYTES, i, l0 + ring.an + (i * 4 + wave) * 1024);
  g7_fill_b(src, cur_b + G7_ROW_BYTES, smem + ring.bn, wave);
  G7_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
  float keep = 0.f;
  unsigned long long t_loop = 0;
  int tiles = 0;
  for (;;) {
    ++it;
    int64_t m1 = m0, n1 = n0;
    const bool has_next = g7_tile(it, ntm, ntn, group_m, m1, n1);
    m1 = m1 / 256 * BM;
    const char* const next_a = (const char*)(A + m1 * lda);
    const char* const next_b = (const char*)(B + n1 * ldb);
    f32x4_t acc[TIN][8];
    const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < TIN * 8; ++q) acc[q >> 3][q & 7] = z4;
    const unsigned long long c0 = clock64();
    gemm_mainloop7_cont16h<T, false, G7NoTail, TIN>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc, nullptr);
    t_loop += clock64() - c0;
    ++tiles;
#pragma unroll
    for (int q = 0; q < TIN * 8; ++q) { asm volatile("" : "+a"(acc[q >> 3][q & 7])); keep += acc[q >> 3][q & 7][0]; }
    if (!has_next) break;
    cur_a = next_a; cur_b = next_b; m0 = m1; n0 = n1;
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = keep;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t_loop; cyc[blockIdx.x * 2 + 1] = (unsigned long long)tiles * nk; }
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

template <int TIN>
static void run(const char* what, int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, float* out, unsigned long long* cyc) {
  hipFuncSetAttribute((const void*)kloop_kernel<bf16_t, TIN>, hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int64_t tiles = (M / (TIN * 32)) * (N / 256);
  const unsigned grid = (unsigned)std::min<int64_t>(256, tiles);
  hipMemset(cyc, 0, 512 * 8);
  auto go = [&]() { hipLaunchKernelGGL((kloop_kernel<bf16_t, TIN>), dim3(grid), dim3(G6_THREADS), G7_LDS_BYTES, 0, A, K, B, K, out, M, N, K, 8, cyc); };
  for (int i = 0; i < 3; ++i) go();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) go();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost);
  double c = 0, st = 0;
  for (unsigned b = 0; b < grid; ++b) { c += (double)h[b * 2]; st += (double)h[b * 2 + 1]; }
  printf("%3d x 256 tiles  %-22s M=%ld N=%ld K=%ld : %7.1f us  %7.1f TFLOP/s  %ld tiles on %u CUs, %.0f cycles per K step (K loop + tile walk only)\n",
         TIN * 32, what, (long)M, (long)N, (long)K, ms * 1e3, 2.0 * M * N * K / (ms * 1e9), (long)tiles, grid, st > 0 ? c / st : 0.0);
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 9216;      // 9216: the training step's token rows; 131072: the encoder benchmark's
  printf("G7_DMA_EARLY=%d  G7H_ABL=%d  M=%ld\n", (int)G7_DMA_EARLY, (int)G7H_ABL, (long)M);
  bf16_t *A, *B; float* out; unsigned long long* cyc;
  hipMalloc(&A, (size_t)M * 3072 * 2); hipMalloc(&B, (size_t)3072 * 3072 * 2); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 512 * 8);
  fill_bf16(A, (size_t)M * 3072, 1.0f, 1); fill_bf16(B, (size_t)3072 * 3072, 0.05f, 2);
  for (int round = 0; round < (M > 20000 ? 1 : 2); ++round) {
    run<8>("dgrad, K = 3072", M, 768, 3072, A, B, out, cyc);  run<4>("dgrad, K = 3072", M, 768, 3072, A, B, out, cyc);
    run<8>("dgrad, K = 2304", M, 768, 2304, A, B, out, cyc);  run<4>("dgrad, K = 2304", M, 768, 2304, A, B, out, cyc);
    run<8>("dgrad / out-proj, K = 768", M, 768, 768, A, B, out, cyc); run<4>("dgrad / out-proj, K = 768", M, 768, 768, A, B, out, cyc);
    run<8>("qkv forward", M, 2304, 768, A, B, out, cyc);      run<4>("qkv forward", M, 2304, 768, A, B, out, cyc);
    run<8>("ffn1 shape", M, 3072, 768, A, B, out, cyc);       run<4>("ffn1 shape", M, 3072, 768, A, B, out, cyc);
  }
  return 0;
}
