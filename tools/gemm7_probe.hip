// Developer tool: times the generation-7 persistent GEMM (gemm_wide7.h) on the encoder's four shapes with parts of
// the tile compiled out (-DG7_ABL=<bits>, gemm_core7.h), on random operands, hipEvents around 20 launches.
//   for a in 0 1 2 4 8 16 32 64 ...; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DG7_ABL=$a tools/gemm7_probe.hip -o build/g7probe_$a; done
#define G7_TRACE_STEPS 1      // per-step stamps (step0 / later steps below): compiled out of the library
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
static int g_grid_cap = 0, g_stagger = 0, g_cont = 1;
int om_option(int o) { return o == OM_OPT_GEMM_MAX_GRID ? g_grid_cap : (o == OM_OPT_GEMM_CONT ? g_cont : 8); }      // OM_OPT_GEMM_GROUP_M default 8

static void fill_bf16(bf16_t* d, size_t n, float scale, unsigned long long seed) {
  const size_t chunk = std::min<size_t>(n, (size_t)1 << 22);      // 4 M random values, repeated (rows differ: 4 M is not a multiple of any row)
  std::vector<bf16_t> h(chunk + 37);
  unsigned long long x = 88172645463325252ull ^ seed;
  for (auto& v : h) {
    float acc = 0;
    for (int i = 0; i < 4; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; acc += (float)(x & 0xffff) / 65536.0f - 0.5f; }
    v = f32_to_bf16(acc * 1.7f * scale);
  }
  for (size_t o = 0; o < n; o += chunk + 37) hipMemcpy(d + o, h.data(), std::min(chunk + 37, n - o) * 2, hipMemcpyHostToDevice);
}
static void fill_f32(float* d, size_t n, float a, float b) {
  std::vector<float> h(n);
  unsigned long long x = 1234567ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = a + (b - a) * (float)(x & 0xffff) / 65536.0f; }
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
}

static bool g_zero = false;
static unsigned long long* g_trace = nullptr;
static bf16_t *g_rlo = nullptr, *g_clo = nullptr;
template <int ACT, bool RESID, int LNF>
static void run(const char* what, int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, bf16_t* C, const bf16_t* R,
                const float* vecs, float* stats_in, float* stats_out) {
  GemmEpilogue ep = {};
  ep.bias = vecs;
  ep.act = ACT;
  ep.ln_inv_h = 1.0f / 768.0f; ep.ln_eps = 1e-12f;
  if (LNF == 1) { ep.ln_stats = stats_in; ep.ln_colsum = vecs + 4096; }
  if (RESID) { ep.resid = R; ep.ldr = N; }
  if (LNF >= 2) { ep.rln_stats = stats_in; ep.rln_g = vecs + 8192; ep.rln_b = vecs + 12288; ep.stats_out = stats_out; }
  if (LNF == 3) { ep.resid_lo = g_rlo; ep.out_lo = g_clo; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch7<bf16_t, ACT, RESID, LNF>(A, K, B, K, C, N, M, N, K, ep, 0);
  const int reps = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) { ep.reverse = i & 1; launch7<bf16_t, ACT, RESID, LNF>(A, K, B, K, C, N, M, N, K, ep, 0); }
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double tiles_per_cu = (double)(M / 256) * (N / 256) / 256.0;
  printf("%s%s ABL=%-3d pol=%d cap=%-3d cont=%d %-28s M=%ld N=%ld K=%ld : %8.1f us  %7.1f TFLOP/s  %7.2f us per tile-slot (%.1f tiles/CU)\n", VARIANT, g_zero ? "z" : "", G7_ABL, G7_ST_POLICY, g_grid_cap, g_cont, what, (long)M,
         (long)N, (long)K, ms * 1e3, 2.0 * M * N * K / (ms * 1e9), ms * 1e3 / tiles_per_cu, tiles_per_cu);
  if (g_trace) {            // one traced launch: average phase lengths of a tile in shader ticks, and ticks per wall microsecond
    const size_t nblk = 8192;
    hipMemset(g_trace, 0, nblk * 32 * 8);
    ep.trace = g_trace; ep.reverse = 0;
    launch7<bf16_t, ACT, RESID, LNF>(A, K, B, K, C, N, M, N, K, ep, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 32);
    hipMemcpy(h.data(), g_trace, nblk * 32 * 8, hipMemcpyDeviceToHost);
    double init = 0, pro = 0, loop = 0, epi = 0, tot = 0, clk = 0, st0 = 0, strest = 0; size_t n = 0;
    const int nk = (int)(K / 64);
    for (size_t b = 0; b < nblk; ++b) {
      const unsigned long long* t = &h[b * 32];
      if (!(t[0] && t[28] && t[15] && t[3] && t[1] && t[31] > t[30])) continue;
      init += (double)(t[1] - t[0]); pro += (double)(t[3] - t[1]); loop += (double)(t[15] - t[3]); epi += (double)(t[28] - t[15]); tot += (double)(t[28] - t[0]);
      clk += (double)(t[28] - t[0]) / ((double)(t[31] - t[30]) * 10.0);      // ticks per ns (100 MHz wall counter)
      if (nk >= 3) { st0 += (double)(t[4] - t[3]); strest += (double)(t[3 + std::min(nk, 12) - 1] - t[4]) / (std::min(nk, 12) - 2); }
      ++n;
    }
    if (n && RESID) {      // per patch iteration of the residual epilogue: stamps 17..24, then the end (28)
      double it[9] = {0}; size_t m = 0;
      for (size_t b = 0; b < nblk; ++b) {
        const unsigned long long* t = &h[b * 32];
        if (!(t[15] && t[16] && t[17] && t[24] && t[28])) continue;
        it[0] += (double)(t[17] - t[15]);
        for (int p = 0; p < 7; ++p) it[1 + p] += (double)(t[18 + p] - t[17 + p]);
        it[8] += (double)(t[28] - t[24]); ++m;
      }
      if (m) { printf("    residual epilogue: to first patch %.0f | iterations", it[0] / m); for (int p = 1; p < 9; ++p) printf(" %.0f", it[p] / m); printf(" ticks\n"); }
    }
    if (n) printf("    trace over %zu tiles: wait-for-tables %.0f  acc-init %.0f  K loop %.0f (MFMA %ld, x%.3f; step0 %.0f, later steps %.0f)  epilogue %.0f  tile %.0f ticks; %.2f GHz shader clock (per-tile mean)\n",
                  n, init / n, pro / n, loop / n, (long)(K * 32), loop / n / (K * 32.0), st0 / n, strest / n, epi / n, tot / n, clk / n);
  }
}

// spot check of the plain variant (bias only): 512 sampled outputs against a host dot product in double
static void check_plain(int64_t M, int64_t N, int64_t K, const bf16_t* A, const bf16_t* B, bf16_t* C, const float* vecs) {
  GemmEpilogue ep = {};
  ep.bias = vecs;
  hipMemset(C, 0xff, (size_t)M * N * 2);
  launch7<bf16_t, OM_ACT_NONE, false, 0>(A, K, B, K, C, N, M, N, K, ep, 0);
  hipDeviceSynchronize();
  std::vector<bf16_t> a(K), b(K); std::vector<float> bias(N);
  hipMemcpy(bias.data(), vecs, N * 4, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  unsigned long long x = 99;
  for (int i = 0; i < 512; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const int64_t m = (i < 8) ? (i & 1 ? M - 1 - i : i) : (int64_t)(x % (unsigned long long)M);
    const int64_t n = (int64_t)((x >> 32) % (unsigned long long)N);
    hipMemcpy(a.data(), A + m * K, K * 2, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), B + n * K, K * 2, hipMemcpyDeviceToHost);
    bf16_t c; hipMemcpy(&c, C + m * N + n, 2, hipMemcpyDeviceToHost);
    double ref = bias[n];
    for (int64_t k = 0; k < K; ++k) ref += (double)bf16_to_f32(a[k]) * (double)bf16_to_f32(b[k]);
    const double err = fabs((double)bf16_to_f32(c) - ref), tol = 0.01 * fabs(ref) + 0.02;
    if (!(err <= tol)) ++bad;
    worst = std::max(worst, err);
  }
  printf("CHECK cont=%d ABL=%d M=%ld N=%ld K=%ld: %s (max |err| %.4f over 512 samples)\n", g_cont, G7_ABL, (long)M, (long)N, (long)K,
         bad ? "FAILED" : "ok", worst);
}

int main(int argc, char** argv) {
  const int64_t M = 131072;
  bf16_t *A, *B, *C, *R; float *vecs, *st_in, *st_out;
  hipMalloc(&A, (size_t)M * 3072 * 2); hipMalloc(&B, (size_t)3072 * 3072 * 2); hipMalloc(&C, (size_t)M * 3072 * 2); hipMalloc(&R, (size_t)M * 768 * 2);
  hipMalloc(&vecs, 16384 * 4); hipMalloc(&st_in, (size_t)M * 8); hipMalloc(&st_out, (size_t)M * 8 * 8);
  const bool zero = argc > 1; g_zero = zero;          // any argument: zero-filled operands (how much of the rate is the power limit?)
  if (zero) { hipMemset(A, 0, (size_t)M * 3072 * 2); hipMemset(B, 0, (size_t)3072 * 3072 * 2); hipMemset(R, 0, (size_t)M * 768 * 2); printf("operands: zeros\n"); }
  else { fill_bf16(A, (size_t)M * 3072, 1.0f, 1); fill_bf16(B, (size_t)3072 * 3072, 0.05f, 2); fill_bf16(R, (size_t)M * 768, 1.0f, 3); }
  fill_f32(vecs, 16384, -0.5f, 0.5f);
  {  // plausible (sum, sum of squares) of 768 N(0,1) values
    std::vector<float> h((size_t)M * 2);
    for (size_t i = 0; i < (size_t)M; ++i) { h[2 * i] = (float)((i * 37) % 23) - 11.f; h[2 * i + 1] = 700.f + (float)((i * 13) % 101); }
    hipMemcpy(st_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  hipMemset(st_out, 0, (size_t)M * 8);
  hipMalloc(&g_rlo, (size_t)M * 768 * 2); hipMalloc(&g_clo, (size_t)M * 768 * 2); fill_bf16(g_rlo, (size_t)M * 768, 0.004f, 7);
  if (!(G7_ABL)) for (int gc_ : {0, 3, 11}) { g_cont = gc_;
    check_plain(4096, 768, 768, A, B, C, vecs); check_plain(2048, 2304, 768, A, B, C, vecs); check_plain(2048, 768, 3072, A, B, C, vecs);
    check_plain(512, 512, 128, A, B, C, vecs); check_plain(512, 256, 64, A, B, C, vecs); check_plain(65536, 768, 192, A, B, C, vecs);
  }
  hipMalloc(&g_trace, 8192 * 32 * 8);
  if (getenv("G7_MODE") && atoi(getenv("G7_MODE")) == 1) {
    // (a) how a tile's phases change with the number of CUs that run at once (same tiles per CU): burst contention or a per-CU limit?
    for (int cap : {256, 128, 64, 32, 8}) {
      g_grid_cap = cap; g_stagger = 0;
      const int64_t Mc = M * cap / 256;
      run<OM_ACT_NONE, false, 1>("qkv (ln-folded A)", Mc, 2304, 768, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_NONE, true, 2>("out-proj (+LN resid, stats)", Mc, 768, 768, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_NONE, true, 2>("ffn2 (+LN resid, stats)", Mc, 768, 3072, A, B, C, R, vecs, st_in, st_out);
    }
    // (b) staggered start of the CUs of an XCD (four phases, `stag` x 256 cycles apart), full grid
    g_grid_cap = 0;
    for (int rep = 0; rep < 2; ++rep)
      for (int stag : {0, 8, 16, 24, 40, 64}) {
        g_stagger = stag;
        run<OM_ACT_NONE, false, 1>("qkv (ln-folded A)", M, 2304, 768, A, B, C, R, vecs, st_in, st_out);
        run<OM_ACT_NONE, true, 2>("out-proj (+LN resid, stats)", M, 768, 768, A, B, C, R, vecs, st_in, st_out);
        run<OM_ACT_GELU_ERF, false, 1>("ffn1 + gelu (ln-folded A)", M, 3072, 768, A, B, C, R, vecs, st_in, st_out);
        run<OM_ACT_NONE, true, 2>("ffn2 (+LN resid, stats)", M, 768, 3072, A, B, C, R, vecs, st_in, st_out);
      }
    return 0;
  }
  for (int round = 0; round < 2; ++round) {
    for (int cont : {0, 3, 11}) {      // the ring restarted per tile (round 3) / continuous (round 4: 7c without, 7r with a residual) / the same on 16 x 16 x 32 MFMAs
      g_cont = cont;
      printf("-- continuous ring %s\n", cont == 0 ? "off" : (cont & 8 ? "ON, 16x16x32" : "ON, 32x32x16"));
      run<OM_ACT_NONE, false, 1>("qkv (ln-folded A)", M, 2304, 768, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_NONE, true, 2>("out-proj (+LN resid, stats)", M, 768, 768, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_GELU_ERF, false, 1>("ffn1 + gelu (ln-folded A)", M, 3072, 768, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_NONE, true, 2>("ffn2 (+LN resid, stats)", M, 768, 3072, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_NONE, true, 0>("ffn2 shape, plain residual", M, 768, 3072, A, B, C, R, vecs, st_in, st_out);
      run<OM_ACT_NONE, false, 0>("plain", 32768, 3072, 3072, A, B, C, R, vecs, st_in, st_out);
    }
    run<OM_ACT_NONE, true, 3>("out-proj two planes", M, 768, 768, A, B, C, R, vecs, st_in, st_out);
    run<OM_ACT_NONE, true, 3>("ffn2 two planes", M, 768, 3072, A, B, C, R, vecs, st_in, st_out);
  }
  return 0;
}
