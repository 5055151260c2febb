// om_gemm_nt: C = act(A · B^T + bias) (+|*) resid  on MFMA (bf16 / f16 or exact-f32), gfx950.
// Stands in for the ATen/BLAS GEMM under every nn.Linear on the hot path
// (HF:models/bert/modeling_bert.py:175-177,289-293,334-351; linear.py:22-23).
//
// Four tile shapes, each with a job none of the others does:
//   v1  128x128 tile, 2-stage LDS, direct stores             any M, N: small / ragged problems, odd alignments (this file)
//   v2  256x128 tile, 3-deep LDS ring (gemm_core2.h)          M >= 512 with few column tiles: the training batch (this file)
//   v6  256x256 tile, 4 waves of 128x128, 64-byte K steps     f32 (3 x bf16 split), training epilogues (gemm_wide6_*.hip)
//   v7  256x256 tile, persistent, 128-byte K steps            16-bit inference: whole tiles, fused LayerNorm (gemm_wide7*.hip)
// (the eight-wave 256x256 generation 4 that v6 replaced is gone: what it still caught -- a bias that is not 16-byte
// aligned, an epilogue v6 does not instantiate -- runs on v2)
#include "gemm_core2.h"
#include "gemm_epilogue.h"

// ---- v1: 128x128 tile, direct (scattered) stores; any M, N --------------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t ntm = (M + GEMM_BM - 1) / GEMM_BM, ntn = (N + GEMM_BN - 1) / GEMM_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  const int64_t m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
  f32x16_t acc[2][2];
  gemm_mainloop<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const EpiScalars es(ep);
#define OM_V1_CALL(A, TR) store_direct<OutT, A, TR>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], m0 + wm * 64, n0 + wn * 64, C, ldc, M, N, ep, es)
  OM_EPI_SWITCH(es.act, es.train, OM_V1_CALL)
#undef OM_V1_CALL
}

// ---- split-K: C (f32, zero-initialised by the caller) += A[:, ks] · B[:, ks]^T per K slice --------
// Weight-gradient contractions have a long K (the token count) and a small output (768 x 768 is
// 9 tiles of 256^2): slicing K over blockIdx.y fills the chip; partial sums meet in f32 atomics.
template <typename T>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_splitk_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, int steps_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t ntn = (N + GEMM_BN - 1) / GEMM_BN;
  const int64_t m0 = (blockIdx.x / ntn) * GEMM_BM, n0 = (blockIdx.x % ntn) * GEMM_BN;
  f32x16_t acc[2][2];
  gemm_mainloop<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc,
                   (int64_t)blockIdx.y * steps_per_slice * GEMM_ROW_BYTES, steps_per_slice);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + (lane & 31);
    if (n >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int64_t mbase = m0 + wm * 64 + mi * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
        if (m < M) atomicAdd(C + m * ldc + n, acc[mi][ni][r]);
      }
    }
  }
}

// ---- v2: 256x128 tile, 3-deep LDS ring -----------------------------------------------------------
template <typename T, typename OutT>
__global__ __launch_bounds__(G2_THREADS) void gemm_nt_kernel2(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, OutT* C,
    int64_t ldc, int64_t M, int64_t N, int64_t K, GemmEpilogue ep, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g2_tile_coords(M, N, group_m, m0, n0);
  f32x16_t acc[2][2];
  unsigned long long* tr = ep.trace ? ep.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = clock64();
  gemm_mainloop2<T>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, tr);
  if (tr && threadIdx.x == 0) tr[15] = clock64();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const EpiScalars es(ep);
  const int64_t nc = n0 + wn * 64;
  const float b0 = (ep.bias && nc + (lane & 31) < N) ? ep.bias[nc + (lane & 31)] : 0.f;
  const float b1 = (ep.bias && nc + 32 + (lane & 31) < N) ? ep.bias[nc + 32 + (lane & 31)] : 0.f;
  char* region = smem + wave * (32 * PATCH_STRIDE);
  __syncthreads();                                  // every wave is done reading the ring
#define OM_V2_CALL(A, TR)                                                                                   \
  store_patch<OutT, A, TR>(acc[0][0], acc[0][1], b0, b1, m0 + wm * 64, nc, C, ldc, M, N, ep, es, region);    \
  store_patch<OutT, A, TR>(acc[1][0], acc[1][1], b0, b1, m0 + wm * 64 + 32, nc, C, ldc, M, N, ep, es, region)
  OM_EPI_SWITCH(es.act, es.train, OM_V2_CALL)
#undef OM_V2_CALL
  if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; }
}

OM_DEFINE_LAUNCHER(launch_gemm, gemm_nt_kernel, GEMM_THREADS, GEMM_LDS_BYTES, GEMM_BM, GEMM_BN)
OM_DEFINE_LAUNCHER(launch_gemm2, gemm_nt_kernel2, G2_THREADS, G2_LDS_BYTES, G2_BM, G2_BN)

static int gemm_variant() {     // OM_OPT_GEMM_VARIANT = 1|2|6 pins a kernel generation (A/B measurements; 0: automatic)
  return om_option(OM_OPT_GEMM_VARIANT);
}

// The 256-row kernels write whole 16-byte output segments; small or ragged problems use v1.
static bool wide_ok(int out_dtype, const void* C, int64_t ldc, int64_t M, int64_t N, const GemmEpilogue& ep) {
  if (gemm_variant() == 1) return false;
  const int64_t vec = out_dtype == OM_F32 ? 4 : 8;
  if (M < 512 || N % vec || ldc % vec || ((uintptr_t)C & 15)) return false;
  if (ep.resid && (ep.ldr % vec || ((uintptr_t)ep.resid & 15))) return false;
  return true;
}

static int g_debug_gen_value();
bool omk_gemm_ln_fusable(int dtype, int64_t M, int64_t N, int64_t K) {
  // float16 has the persistent generation only: whole 256 x 256 tiles (the encoder pads its token rows)
  if (dtype == OM_F16) return M >= 512 && M % 256 == 0 && N % 256 == 0 && (K * 2) % 128 == 0 && gemm_variant() == 0;
  // bfloat16 likewise since round 3: the fused-LayerNorm epilogues (slot statistics, two-plane residual) exist in
  // generation 7 only
  return dtype == OM_BF16 && M >= 512 && M % 256 == 0 && N % 256 == 0 && (K * 2) % 128 == 0 && gemm_variant() == 0 && g_debug_gen_value() != 6;
}

static unsigned long long* g_trace = nullptr;
extern "C" void om_debug_gemm_trace(unsigned long long* buf) { g_trace = buf; }
unsigned long long* omk_debug_trace() { return g_trace; }      // the scan kernel of search.hip stamps into the same buffer
static int g_debug_gen = 0;     // 0: default selection; 6: never generation 7; 70: generation 7 with one tile per workgroup (A/B)
extern "C" void om_debug_gemm_gen(int gen) { g_debug_gen = gen; }
static int g_debug_gen_value() { return g_debug_gen; }
bool omk_gemm_wide7_has(int act, bool resid, int lnf);
bool omk_gemm_wide7_f16_has(int act, bool resid, int lnf);
int omk_gemm_wide7_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);
int omk_gemm_wide7(bool persist, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);

bool omk_gemm_skinny_ok(int in_dtype, int out_dtype, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep);      // gemm_skinny.hip
int omk_gemm_skinny(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, const GemmEpilogue& ep, hipStream_t s);
bool omk_gemm_wide7_train_ok(int64_t M, int64_t N, int64_t K, int64_t ldc, const GemmEpilogue& ep);
int omk_gemm_wide7_train(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                         int64_t N, int64_t K, const GemmEpilogue& ep, hipStream_t s);

int omk_gemm(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int out_dtype,
             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep_in,
             hipStream_t s) {
  GemmEpilogue ep = ep_in;
  ep.trace = g_trace;
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0) OM_FAIL("K must be positive");
  const int64_t es = in_dtype == OM_F32 ? 4 : 2;
  if ((K * es) % GEMM_ROW_BYTES != 0) OM_FAIL("K*sizeof(elem) must be a multiple of 128 bytes");
  if ((lda * es) % 16 != 0 || (ldb * es) % 16 != 0) OM_FAIL("lda/ldb must keep rows 16-byte aligned");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) OM_FAIL("A/B must be 16-byte aligned");
  if ((ep.act & 0xff) == OM_ACT_GELU_ERF_GRAD && !ep.resid) OM_FAIL("gelu-grad epilogue needs resid");
  // few rows (a query, a handful of sequences): the weight-streaming kernel, N / 16 workgroups instead of N / 128
  if (gemm_variant() == 0 && g_debug_gen == 0 && omk_gemm_skinny_ok(in_dtype, out_dtype, M, N, K, ep))
    return omk_gemm_skinny(in_dtype, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (ep.resid32 || ep.out32 || ep.a_ln32 || ep.rln32) OM_FAIL("f32 residual / f32 sum / pending-LayerNorm epilogue: the few-rows kernel only (gemm_skinny.hip)");
  const bool wide = wide_ok(out_dtype, C, ldc, M, N, ep);
  // Pick the tile generation that finishes first: whole rounds of (256 CUs x resident workgroups)
  // times the tile's work over its measured relative efficiency (profiles/r01_selftest_gemm_v4.log).
  int gen = 1;
  if (wide) {
    // cost = rounds x (work a CU has in flight per round) / efficiency; v1 keeps 2 workgroups per CU
    auto rounds = [&](int64_t bm, int64_t bn, int64_t slots) {
      const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
      return (double)((tiles + slots - 1) / slots);
    };
    const double c4 = N >= 256 ? rounds(256, 256, 256) * (256.0 * 256.0) / 1.00 : 1e30;
    const double c2 = rounds(256, 128, 256) * (256.0 * 128.0) / 0.92;
    const double c1 = rounds(128, 128, 512) * (2 * 128.0 * 128.0) / 0.70;
    gen = c4 <= c2 && c4 <= c1 ? 6 : (c2 <= c1 ? 2 : 1);     // 6 falls back to 2 where it has no variant
    if (gemm_variant() == 2) gen = 2;
    if (gemm_variant() == 6) gen = N >= 256 ? 6 : 2;
    // A/B (OM_OPT_GEMM_CONT bit 4): plain 16-bit shapes of whole 256 x 256 tiles go to the continuous-ring kernels even when
    // those leave CUs idle (training: N = 768 at 9 216 token rows is 108 tiles -- the cost model above prefers 216 tiles of
    // 256 x 128 on generation 2; the idle CUs are not idle in a training step, the weight-gradient lane runs beside)
    const bool f16_model = in_dtype == OM_F16 && (om_option(OM_OPT_GEMM_CONT) & 128);      // bit 7 (round 5): float16 follows the same rules
    const bool plain16 = (in_dtype == OM_BF16 || f16_model) && out_dtype == in_dtype && !ep.pre_act && ep.drop_p == 0.f && M % 256 == 0 && N % 256 == 0 &&
                         K * 2 >= 3 * 128;
    if ((om_option(OM_OPT_GEMM_CONT) & 16) && plain16) gen = 6;
    // Round 5 (bit 6, default on): the model above prices generation 2 at 0.92 of a 256 x 256 tile's rate; measured in the training
    // step (profiles/r05_train_timeline_v0.txt) its K step takes ~2 650 cycles for 1 024 cycles of MFMA against 2 425 for 2 048 on
    // the continuous ring -- 0.55.  With that figure the QKV projection of the training forward (9 216 x 2 304: 324 whole tiles, two
    // rounds) moves to the continuous kernel (51 -> ~40 us); the N = 768 shapes (108 tiles on 256 CUs) stay where they are.
    if ((om_option(OM_OPT_GEMM_CONT) & 64) && plain16 && gen == 2 && gemm_variant() == 0) {
      const double c7 = rounds(256, 256, 256) * (256.0 * 256.0), c2r = rounds(256, 128, 256) * (256.0 * 128.0) / 0.55;
      if (c7 < c2r) gen = 6;
    }
  }
  const bool ln_fused = ep.ln_stats || ep.rln_stats || ep.stats_out;
  // the training forward's FFN1 (gelu + gelu' to the tape): the continuous 256 x 256 kernel with its two-output epilogue
  if (wide && (in_dtype == OM_BF16 || in_dtype == OM_F16) && out_dtype == in_dtype && gemm_variant() == 0 && g_debug_gen != 6 &&
      omk_gemm_wide7_train_ok(M, N, K, ldc, ep))
    return omk_gemm_wide7_train(in_dtype, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  if (in_dtype == OM_F16 && out_dtype == OM_F16) {
    // float16 -> float16 (the inference encoder's float16 mode): the persistent 256 x 256 kernel where the problem is
    // made of whole tiles, else the generic 128 / 256-row tiles (which also take the training epilogues of float16 training)
    const int act = ep.act & 0xff;
    const bool resid = ep.resid != nullptr;
    const bool train16 = ep.pre_act != nullptr || ep.drop_p > 0.f || act == OM_ACT_GELU_ERF_GRAD;      // float16 training (round 5): the generic tiles
    const int lnf = ep.ln_stats ? 1 : ((ep.rln_stats || ep.stats_out) ? (ep.out_lo ? (ep.lo8 ? 4 : 3) : 2) : 0);
    const bool g7 = wide && !train16 && gemm_variant() == 0 && M % 256 == 0 && N % 256 == 0 && (K * 2) % 128 == 0 &&
                    (((uintptr_t)ep.bias & 15) == 0) && !(ep.ln_stats && (ep.rln_stats || ep.stats_out)) &&
                    !(lnf >= 2 && !ep.stats_out) && (!resid || (ep.ldr * 2) % 128 == 0) && !((ep.act & OM_ACT_MUL_RESID) && (lnf != 0 || !resid)) &&
                    omk_gemm_wide7_f16_has(act, resid, lnf);
    // Round 5 (OM_OPT_GEMM_CONT bit 7, default on): a PLAIN float16 contraction goes to the persistent kernel only where the tile-choice
    // model above says so, as bfloat16 does.  Before, every whole-tile float16 shape went there: the N = 768 data gradients of a training
    // step (108 tiles on 256 CUs) took 66 / 51 us where generation 2 takes 62 / 47, and 60 / 48 against 50 / 40 at the 5 120 rows of a
    // packed batch (profiles/r05_gemm_variant_probe.json).  The fused-LayerNorm variants exist in generation 7 only.
    const bool want7 = lnf != 0 || gen == 6 || !(om_option(OM_OPT_GEMM_CONT) & 128);
    if (g7 && want7) return omk_gemm_wide7_f16(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    if (ln_fused) OM_FAIL("float16: the fused LayerNorm epilogues need whole 256 x 256 tiles");
    if (wide && gen != 1) return launch_gemm2<f16_t, f16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    return launch_gemm<f16_t, f16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
  }
  if (ln_fused && !(wide && in_dtype == OM_BF16 && out_dtype == OM_BF16 && N >= 256)) OM_FAIL("fused LayerNorm epilogue needs the 256x256 bf16 kernel");
  if (ln_fused) gen = 6;
  if (gen == 6) {
    const int act = ep.act & 0xff;
    const bool train = ep.pre_act != nullptr || ep.drop_p > 0.f;
    const bool resid = ep.resid != nullptr;
    // v6 reads the bias as float4 and writes pre-activation pairs
    const bool aligned = (((uintptr_t)ep.bias & 15) == 0) && (ep.ldp % 2 == 0) && (((uintptr_t)ep.pre_act & 3) == 0);
    if (!aligned && ln_fused) OM_FAIL("fused LayerNorm epilogue needs 16-byte aligned bias");
    if (!aligned) gen = 2;
    else if (g_debug_gen != 6 && gemm_variant() == 0 && in_dtype == OM_BF16 && out_dtype == OM_BF16 && !train && M % 256 == 0 &&
             N % 256 == 0 && (K * 2) % 128 == 0 && !(ep.ln_stats && (ep.rln_stats || ep.stats_out)) &&
             !((ep.rln_stats || ep.stats_out) && !ep.stats_out) && (!resid || (ep.ldr * 2) % 128 == 0) &&
             omk_gemm_wide7_has(act, resid, ep.ln_stats ? 1 : ((ep.rln_stats || ep.stats_out) ? (ep.out_lo ? 3 : 2) : 0)))
      return omk_gemm_wide7(g_debug_gen != 70, A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    else if (ln_fused) OM_FAIL("fused LayerNorm epilogue: whole 256 x 256 tiles of bf16, inference only (generation 7)");
    else if (omk_gemm_wide6_b16_has(in_dtype, out_dtype, act, train, resid))
      return omk_gemm_wide6_b16(in_dtype, A, lda, B, ldb, out_dtype, C, ldc, M, N, K, ep, s);
    else if (omk_gemm_wide6_f32_has(in_dtype, out_dtype, act, train, resid))
      return omk_gemm_wide6_f32(in_dtype, A, lda, B, ldb, out_dtype, C, ldc, M, N, K, ep, s);
    if (ln_fused) OM_FAIL("no kernel for this fused LayerNorm epilogue");
    gen = 2;
  }
#define OM_GEMM_GO(TI, TO)                                                                            \
  do {                                                                                                \
    if (gen == 2) return launch_gemm2<TI, TO>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                 \
    return launch_gemm<TI, TO>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);                                \
  } while (0)
  if (in_dtype == OM_BF16 && out_dtype == OM_BF16) OM_GEMM_GO(bf16_t, bf16_t);
  if (in_dtype == OM_BF16 && out_dtype == OM_F32) OM_GEMM_GO(bf16_t, float);
  if (in_dtype == OM_F32 && out_dtype == OM_F32) OM_GEMM_GO(float, float);
  if (in_dtype == OM_F16 && out_dtype == OM_F32) OM_GEMM_GO(f16_t, float);
  if (in_dtype == OM_F32 && out_dtype == OM_BF16) return launch_gemm<float, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
#undef OM_GEMM_GO
  OM_FAIL("unsupported dtype combination");
}

int omk_gemm_splitk(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int64_t es = in_dtype == OM_F32 ? 4 : 2;
  if ((K * es) % GEMM_ROW_BYTES != 0) OM_FAIL("K*sizeof(elem) must be a multiple of 128 bytes");
  const int64_t tiles = ((M + GEMM_BM - 1) / GEMM_BM) * ((N + GEMM_BN - 1) / GEMM_BN);
  const int nk = (int)(K * es / GEMM_ROW_BYTES);
  int slices = (int)((1024 + tiles - 1) / tiles);          // aim at ~4 workgroups per CU
  if (slices > nk / 4) slices = nk / 4 > 0 ? nk / 4 : 1;   // but at least 4 K steps per slice
  const int per = (nk + slices - 1) / slices;
  slices = (nk + per - 1) / per;
  static std::atomic<bool> attr_bf16{false}, attr_f32{false};
  const bool timing = om_timing_on();
  const int tclass = es == 2 ? OM_TIMING_GEMM_BF16 : OM_TIMING_GEMM_F32;
  if (timing) om_timing_begin(tclass, s);
  if (in_dtype == OM_BF16) {
    if (!attr_bf16) { OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_splitk_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)); attr_bf16 = true; }
    hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16_t>), dim3((unsigned)tiles, slices), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s,
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, per);
  } else if (in_dtype == OM_F16) {
    static std::atomic<bool> attr_f16{false};
    if (!attr_f16) { OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_splitk_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)); attr_f16 = true; }
    hipLaunchKernelGGL((gemm_nt_splitk_kernel<f16_t>), dim3((unsigned)tiles, slices), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s,
                       (const f16_t*)A, lda, (const f16_t*)B, ldb, C, ldc, M, N, K, per);
  } else if (in_dtype == OM_F32) {
    if (!attr_f32) { OM_HIP(hipFuncSetAttribute((const void*)gemm_nt_splitk_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)); attr_f32 = true; }
    hipLaunchKernelGGL((gemm_nt_splitk_kernel<float>), dim3((unsigned)tiles, slices), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s,
                       (const float*)A, lda, (const float*)B, ldb, C, ldc, M, N, K, per);
  } else {
    OM_FAIL("unsupported dtype");
  }
  if (timing) om_timing_end(tclass, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

extern "C" int om_gemm_nt(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                          int out_dtype, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const float* bias, const void* resid, int64_t ldr, int act,
                          void* stream) {
  GemmEpilogue ep = {};
  ep.bias = bias; ep.resid = resid; ep.ldr = ldr; ep.act = act;
  return omk_gemm(in_dtype, A, lda, B, ldb, out_dtype, C, ldc, M, N, K, ep, (hipStream_t)stream);
}
