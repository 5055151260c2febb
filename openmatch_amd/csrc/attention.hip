// Fused self-attention for short sequences (L <= 256, head_dim 64), gfx950.
// Replaces BertSelfAttention / T5Attention score+softmax+context
// (HF:models/bert/modeling_bert.py:111-136, HF:models/t5/modeling_t5.py:176-370, eval mode).
//
// One workgroup per (batch, head); one wavefront per 32 query rows.  K is staged in LDS
// row-major (XOR-swizzled 16-byte slots), V is staged TRANSPOSED ([d][key], +4 pad) so both
// MFMA B-operands are contiguous LDS reads.  Scores are computed swapped (S^T = K Q^T): with
// the 32x32 accumulator map (col = lane&31, rows in registers) every lane then owns ONE query
// row and 16 keys per key tile, so the softmax is lane-local plus one exchange with lane^32,
// and the probabilities already sit in the A-operand layout of the P·V MFMA.
#include "attn_common.h"

template <typename T, int KT>
__global__ __launch_bounds__(64 * KT) void attention_kernel(
    const T* __restrict__ qkv, T* __restrict__ ctx, const int64_t* __restrict__ mask,
    const float* __restrict__ pos_bias, int L, int H, int heads, float scale, float drop_p,
    uint64_t seed) {
  typedef AttnGeom<T> G;
  typedef typename MmaOps<T>::frag_t frag_t;
  constexpr int LP = KT * 32 + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  T* sVt = (T*)(smem + KT * 32 * G::ROWB);
  float* sM = (float*)(smem + KT * 32 * G::ROWB + 64 * LP * (int)sizeof(T));

  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t ld = 3 * (int64_t)H;
  const T* base = qkv + b * L * ld + h * 64;

  for (int idx = tid; idx < KT * 32 * G::CPR; idx += nthr) {
    const int row = idx / G::CPR, c = idx % G::CPR;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (row < L) {
      kv = *(const uint4*)(base + (int64_t)row * ld + H + c * G::EPC);
      vv = *(const uint4*)(base + (int64_t)row * ld + 2 * H + c * G::EPC);
    }
    *(uint4*)(sK + row * G::ROWB + ((c ^ G::key(row)) << 4)) = kv;
    const T* ve = (const T*)&vv;
#pragma unroll
    for (int e = 0; e < G::EPC; ++e) sVt[(c * G::EPC + e) * LP + row] = ve[e];
  }
  // additive key mask: padded keys get finfo.min (HF extended mask); keys past L do not exist
  for (int k = tid; k < KT * 32; k += nthr)
    sM[k] = k < L ? (mask[b * L + k] != 0 ? 0.f : -3.4028235e38f) : -INFINITY;
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int q0 = wave * 32;
  if (q0 >= L) return;
  const int qrow = (q0 + l31) < L ? (q0 + l31) : (L - 1);

  frag_t qf[G::NKK];
#pragma unroll
  for (int kk = 0; kk < G::NKK; ++kk)
    qf[kk] = *(const frag_t*)(base + (int64_t)qrow * ld + (kk * 2 + half) * G::EPC);

  // S^T = K Q^T : lane owns query l31, keys (r&3)+8(r>>2)+4*half of each 32-key tile
  f32x16_t s[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    const int row = t * 32 + l31;
    const char* krow = sK + row * G::ROWB;
    const int key = G::key(row);
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
      const frag_t a = *(const frag_t*)(krow + (((kk * 2 + half) ^ key) << 4));
      MmaOps<T>::mma(a, qf[kk], s[t]);
    }
  }

  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k0 = t * 32 + 8 * g + 4 * half;
      const f32x4_t mb = *(const f32x4_t*)(sM + k0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = s[t][4 * g + e] * scale;
        if (pos_bias) {
          const int kc = (k0 + e) < L ? (k0 + e) : (L - 1);
          v += pos_bias[((int64_t)h * L + qrow) * L + kc];
        }
        v += mb[e];
        s[t][4 * g + e] = v;
        mx = fmaxf(mx, v);
      }
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = G::exp_(s[t][r] - mx);
      s[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[t][r] *= inv;
  if (drop_p > 0.f) {      // training: attention_probs dropout, mask regenerated in the backward
    const uint32_t thresh = (uint32_t)(drop_p * 4294967296.0);
    const float keep_scale = 1.0f / (1.0f - drop_p);
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        s[t][r] = dropout_keep(seed, attn_drop_idx(b, h, heads, L, q0 + l31, key), thresh) ? s[t][r] * keep_scale : 0.f;
      }
  }

  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t) SlabMma<T>::run(s[t], sVt + l31 * LP + t * 32 + 4 * half, LP, o);

  // Context rows leave through LDS: the wave parks its [32 x 64] block in the K rows of its own
  // queries (every wave is past K and V^T after the barrier; waves without query rows have exited)
  // and writes whole 16-byte vectors, 8 (bf16) / 4 (f32) complete rows per instruction, instead of 32
  // two-byte stores per lane.
  __syncthreads();
  T* so = (T*)(sK + (size_t)q0 * G::ROWB);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = (r & 3) + 8 * (r >> 2) + 4 * half;
      ElemOps<T>::store(so + q * 64 + dt * 32 + l31, o[dt][r]);
    }
  T* out = ctx + (b * L + q0) * H + h * 64;
  constexpr int VPR = G::ROWB / 16;                 // 16-byte vectors per row
#pragma unroll
  for (int it = 0; it < 32 * VPR / 64; ++it) {
    const int idx = it * 64 + lane, row = idx / VPR, c = idx % VPR;
    const uint4 v = *(const uint4*)((const char*)so + row * G::ROWB + c * 16);
    if (q0 + row < L) *(uint4*)((char*)(out + (int64_t)row * H) + c * 16) = v;
  }
}

template <typename T, int KT>
static int launch_attn(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias,
                       int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                       hipStream_t s) {
  constexpr int LP = KT * 32 + 4;
  const int lds = KT * 32 * AttnGeom<T>::ROWB + 64 * LP * (int)sizeof(T) + KT * 32 * 4;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)attention_kernel<T, KT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int waves = (L + 31) / 32;
  hipLaunchKernelGGL((attention_kernel<T, KT>), dim3((unsigned)(heads * B)),
                     dim3(64 * waves), lds, s, (const T*)qkv, (T*)ctx, mask, pos_bias, L, H, heads,
                     scale, drop_p, seed);
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int dispatch_attn(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias,
                         int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                         hipStream_t s) {
  if (L <= 32) return launch_attn<T, 1>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
  if (L <= 64) return launch_attn<T, 2>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
  if (L <= 128) return launch_attn<T, 4>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
  if (L <= 192) return launch_attn<T, 6>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
  return launch_attn<T, 8>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
}

int omk_attention(int dtype, const void* qkv, void* ctx, const int64_t* mask,
                  const float* pos_bias, int64_t B, int L, int H, int heads, float scale,
                  float drop_p, uint64_t seed, hipStream_t s) {
  if (B <= 0) return 0;
  if (L < 1 || L > 256) OM_FAIL("sequence length must be in [1,256]");
  if (H != heads * 64) OM_FAIL("head_dim must be 64");
  if (B * heads > 0x7fffffffLL) OM_FAIL("batch too large for one launch");
  if (dtype == OM_BF16) return dispatch_attn<bf16_t>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
  return dispatch_attn<float>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
}
