// Weight-gradient contraction for the training backward (stands in for autograd's  dW = dY^T X  of every nn.Linear in
// HF BertModel / T5 under DRModel.forward + loss.backward(); reference: modeling/dense_retrieval_model.py:89-131):
//
//     C[n, k] += sum_m A[m, n] * B[m, k]          A = dY [M, N],  B = X [M, K],  both ROW-major as the forward /
//     bias[n] += sum_m A[m, n]                     backward kernels leave them, bf16;  C, bias f32 (accumulated)
//
// The contraction index m is the SLOW index of both operands, so neither tile has the k-contiguous fragment layout the
// matrix core wants.  Round 1 transposed both operands in HBM first (transpose_kernel x2 + colsum_kernel + an NT
// split-K GEMM: 10 of the 26.7 ms of a training step, profiles/r01_train_kernel_stats_v1.csv).  Here the tiles are staged
// row-major and read back through gfx950's transposing LDS read (ds_read_b64_tr_b16): a 16-lane group fetches a
// [4 m][16 col] block, 8 bytes per lane, and each lane receives the four m values of ITS column -- two such reads
// are one 32x32x16 MFMA fragment.  LDS image per operand tile (64 m x 128 col): eight [64 m][16 col] sub-tiles of
// 32-byte rows, sub-tile stride 2048 + 128 bytes (the two column blocks a half-wave reads land on disjoint banks).
//
// Workgroup: 128 (n) x 128 (k) output tile, four waves of 64 x 64, over one slice of the token axis (split-M, f32
// atomics into the caller-zeroed / caller-accumulated C); register-staged double buffer, one barrier per 64 tokens.
// The bias column sums ride on the matrix core too (one extra MFMA per A fragment against a fragment of ones) in the
// workgroups of k tile 0.
#include <atomic>

#include "gemm_core.h"
#include "kernels.h"

namespace {
constexpr int TN_BM = 64;                       // tokens per step
constexpr int TN_SUB = TN_BM * 32 + 128;        // bytes per [64][16] sub-tile, padded
constexpr int TN_TILE = 8 * TN_SUB;             // one operand tile
constexpr int TN_LDS = 4 * TN_TILE;             // A, B double-buffered: 69 632 bytes
constexpr int TN_THREADS = 256;

typedef short v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s tn_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
}

__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, float* __restrict__ bias, int64_t M, int N, int K, int rows_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int ntk = K / 128;
  const int n0 = (blockIdx.x / ntk) * 128, k0 = (blockIdx.x % ntk) * 128;
  const int64_t m_begin = (int64_t)blockIdx.y * rows_per_slice;
  const int64_t m_end = m_begin + rows_per_slice < M ? m_begin + rows_per_slice : M;
  if (m_begin >= m_end) return;
  const bool do_bias = bias != nullptr && k0 == 0;

  // staging: wave instruction j = wave * 4 + i covers rows (j >> 1) * 8 .. + 7 and column blocks (j & 1) * 4 .. + 3;
  // lane -> (half h of the 32-byte row, row r8, column block cbl): 16 lanes write 256 contiguous bytes of one sub-tile
  const int h = lane & 1, r8 = (lane >> 1) & 7, cbl = lane >> 4;
  int s_row[4], s_lds[4];
  const bf16_t* pa[4];
  const bf16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = wave * 4 + i;
    const int m = (j >> 1) * 8 + r8, cb = (j & 1) * 4 + cbl;
    s_row[i] = m;
    s_lds[i] = cb * TN_SUB + m * 32 + h * 16;
    pa[i] = A + (m_begin + m) * lda + n0 + cb * 16 + h * 8;
    pb[i] = B + (m_begin + m) * ldb + k0 + cb * 16 + h * 8;
  }
  uint4 ra[4], rb[4];
  auto fetch = [&](int64_t m_at) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = m_at + s_row[i] < m_end;
      ra[i] = ok ? *(const uint4*)(pa[i] + (m_at - m_begin) * lda) : make_uint4(0u, 0u, 0u, 0u);
      rb[i] = ok ? *(const uint4*)(pb[i] + (m_at - m_begin) * ldb) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto stash = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(buf + s_lds[i]) = ra[i];
      *(uint4*)(buf + TN_TILE + s_lds[i]) = rb[i];
    }
  };

  // fragment reads: 16-lane group g -> column block (g & 1) of the 32-column fragment, m half (g >> 1) of the k16 step
  const int g = lane >> 4;
  const int frag_off = (g & 1) * TN_SUB + (g >> 1) * 8 * 32 + (lane & 15) * 8;
  f32x16_t acc[2][2];
  f32x16_t accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accb[i][r] = 0.f; }
  }
  const short one = (short)0x3f80;
  const bf16x8_t ones = {one, one, one, one, one, one, one, one};

  fetch(m_begin);
  stash(smem);
  __syncthreads();
  int cur = 0;
  for (int64_t m_at = m_begin; m_at < m_end; m_at += TN_BM) {
    const bool more = m_at + TN_BM < m_end;
    if (more) fetch(m_at + TN_BM);
    const char* ia = smem + cur * 2 * TN_TILE + (wn * 4) * TN_SUB + frag_off;
    const char* ib = smem + cur * 2 * TN_TILE + TN_TILE + (wk * 4) * TN_SUB + frag_off;
#pragma unroll
    for (int ks = 0; ks < TN_BM / 16; ++ks) {
      bf16x8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const v4s a0 = tn_read(ia + i * 2 * TN_SUB + ks * 512), a1 = tn_read(ia + i * 2 * TN_SUB + ks * 512 + 128);
        const v4s b0 = tn_read(ib + i * 2 * TN_SUB + ks * 512), b1 = tn_read(ib + i * 2 * TN_SUB + ks * 512 + 128);
        fa[i] = (bf16x8_t){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        fb[i] = (bf16x8_t){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
      // acc[i][j][r]: n = n0 + wn*64 + i*32 + 8*(r>>2) + 4*(lane>>5) + (r&3),  k = k0 + wk*64 + j*32 + (lane & 31)
      // (k, the contiguous index of C, runs along the lanes: every atomic instruction below covers whole 128-byte rows)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) MmaOps<bf16_t>::mma(fa[i], fb[j], acc[i][j]);
        if (do_bias && wk == 0) MmaOps<bf16_t>::mma(fa[i], ones, accb[i]);
      }
    }
    if (more) {
      stash(smem + (cur ^ 1) * 2 * TN_TILE);
      __syncthreads();
      cur ^= 1;
    }
  }

  const int kcol = k0 + wk * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nbase = n0 + wn * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nbase + 8 * (r >> 2) + (r & 3);
      float* crow = C + (int64_t)n * ldc + kcol;
      atomicAdd(crow, acc[i][0][r]);
      atomicAdd(crow + 32, acc[i][1][r]);
      if (do_bias && wk == 0 && (lane & 31) == 0) atomicAdd(bias + n, accb[i][r]);
    }
  }
}
}  // namespace

bool omk_gemm_tn_ok(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
  return dtype == OM_BF16 && M > 0 && N % 128 == 0 && K % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0 && N <= (1 << 20) && K <= (1 << 20);
}

int omk_gemm_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* bias,
                int64_t M, int64_t N, int64_t K, hipStream_t s) {
  if (!omk_gemm_tn_ok(dtype, M, N, K, lda, ldb)) OM_FAIL("gemm_tn: bf16 operands with N and K multiples of 128 only");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) OM_FAIL("gemm_tn: operands must be 16-byte aligned");
  const int64_t tiles = (N / 128) * (K / 128);
  const int64_t steps = (M + TN_BM - 1) / TN_BM;
  int64_t slices = (1024 + tiles - 1) / tiles;            // ~4 workgroups per CU (two are resident at a time)
  if (slices > steps / 4) slices = steps / 4 > 0 ? steps / 4 : 1;     // at least 256 tokens per slice
  const int64_t per = (steps + slices - 1) / slices;
  slices = (steps + per - 1) / per;
  static std::atomic<bool> attr{false};
  if (!attr) {
    OM_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS));
    attr = true;
  }
  const bool timing = om_timing_on();
  if (timing) om_timing_begin(OM_TIMING_GEMM_BF16, s);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)tiles, (unsigned)slices), dim3(TN_THREADS), TN_LDS, s, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, C, ldc, bias, M, (int)N, (int)K, (int)(per * TN_BM));
  if (timing) om_timing_end(OM_TIMING_GEMM_BF16, s, 2.0 * (double)M * (double)N * (double)K);
  OM_LAUNCH_CHECK();
  return 0;
}

extern "C" int om_gemm_tn_acc(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                              float* bias, int64_t M, int64_t N, int64_t K, void* stream) {
  if (!A || !B || !C) OM_FAIL("null argument");
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return omk_gemm_tn(in_dtype, A, lda, B, ldb, C, ldc, bias, M, N, K, (hipStream_t)stream);
}
