// Attention backward for bf16 training, L <= 128, one (batch, head) per workgroup -- the counterpart of
// attention.hip's generic attention_kernel (forward with dropout) for HF BertSelfAttention under autograd
// (HF:models/bert/modeling_bert.py:111-136; reached from DRModel.forward + loss.backward(), reference
// modeling/dense_retrieval_model.py:89-131).
//
//   P = softmax(scale Q K^T + mask), Pd = dropout(P), O = Pd V                    (forward, recomputed)
//   dPd = dO V^T ; dP = dropout'(dPd) ; delta = rowsum(P o dP) ; dS = P o (dP - delta) * scale
//   dQ = dS K ; dK = dS^T Q ; dV = Pd^T dO
//
// The generic kernel (train_kernels.hip, still used for f32, T5 biases and L > 128) recomputes the scores in a second,
// lane-per-key orientation for dK / dV -- exponentials and dropout hashes twice -- and builds three transposed LDS
// images with 2-byte stores: 153 us per layer at 72 x 12 heads x 128 tokens for ~15 MFLOP per head
// (profiles/r02_train_kernel_stats_v3.csv).  Here:
//   phase A  wave w <-> queries 32 w .. 32 w + 31, lane <-> query (S^T = K Q^T as in the forward): softmax, dropout (one
//            hash per four probabilities), delta, dS; dQ^T = K^T dS^T with K^T fragments from transposing LDS reads;
//            Pd and dS go to LDS as bf16 [query][key];
//   phase B  wave w <-> keys 32 w .. + 31: dV^T = dO^T Pd, dK^T = Q^T dS -- all four operand kinds are transposing reads
//            (ds_read_b64_tr_b16) of row-major LDS images, nothing is recomputed.
// A transposing read hands lane (column c, half h) the four rows r0 + 4 h .. + 3 of its column: two of them are the
// eight k slots of a 32x32x16 fragment in the order (half, e) <-> row 16 u + 8 (e >> 2) + 4 half + (e & 3) -- the same
// order in which an accumulator's registers enumerate their row index, so accumulators become operands by a bf16 pack
// alone (no cross-lane traffic anywhere in this kernel).
// LDS: K then Q, V then dO as [L][64] rows of 192 bytes (128 + 64: the four rows of a transposing read fall into the four
// 64-byte quarters of the bank cycle); Pd, dS as [L][L] rows of 2 L + 8 bytes (conflict-free 8-byte stores from lane-per-
// query registers; their reads are 4-way conflicted but few).
#include <atomic>

#include "attn_common.h"
#include "train_kernels.h"

namespace {
typedef bf16x8_t frag_t;
typedef short v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s trd(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ frag_t frag_of(v4s a, v4s b) { return (frag_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
// T = bf16_t or f16_t (round 5: float16 training): the kernel moves raw 16-bit words; only the packing of f32 values and the MFMA opcode
// depend on the format
template <typename T> __device__ __forceinline__ uint32_t pk2(float a, float b) { return Half16<T>::pack2(a, b); }
template <typename T> __device__ __forceinline__ frag_t pack8(const f32x16_t& v, int u) {
  const uint4 w = make_uint4(pk2<T>(v[8 * u + 0], v[8 * u + 1]), pk2<T>(v[8 * u + 2], v[8 * u + 3]), pk2<T>(v[8 * u + 4], v[8 * u + 5]), pk2<T>(v[8 * u + 6], v[8 * u + 7]));
  return __builtin_bit_cast(frag_t, w);
}
template <typename T> __device__ __forceinline__ f32x16_t mma16(const frag_t& a, const frag_t& b, const f32x16_t& c);
template <> __device__ __forceinline__ f32x16_t mma16<bf16_t>(const frag_t& a, const frag_t& b, const f32x16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_t mma16<f16_t>(const frag_t& a, const frag_t& b, const f32x16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
constexpr int PITCH = 192;            // bytes per row of the [L][64] images

// o[dt][r] = OUT[row l31 of this wave][d = 32 dt + (r&3) + 8 (r>>2) + 4 half]  ->  16-byte stores, one row per lane
template <typename T>
__device__ __forceinline__ void store_rows(const f32x16_t (&o)[2], char* out, int half, bool valid) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const uint32_t a0 = pk2<T>(o[dt][8 * gp + 0], o[dt][8 * gp + 1]), a1 = pk2<T>(o[dt][8 * gp + 2], o[dt][8 * gp + 3]);
      const uint32_t b0 = pk2<T>(o[dt][8 * gp + 4], o[dt][8 * gp + 5]), b1 = pk2<T>(o[dt][8 * gp + 6], o[dt][8 * gp + 7]);
      // (a: d group 2 gp, b: d group 2 gp + 1) -- lanes 32-63 of a swap with lanes 0-31 of b
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (valid) *(uint4*)(out + (32 * dt + 16 * gp + 8 * half) * 2) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

template <typename T, int KT, bool BIAS>
__global__ __launch_bounds__(64 * KT) void attention_bwd16_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx, bf16_t* __restrict__ dqkv,
    const int64_t* __restrict__ mask, int Lm, int H, int heads, float scale, float drop_p, uint64_t seed, const int* __restrict__ cu,
    const float* __restrict__ pos_bias, float* __restrict__ drel) {
  // BIAS (round 6: T5 training, which ran the generic kernel at 281 us per layer where this one takes ~50): pos_bias [heads][Lm][Lm] is added
  // to the scaled scores (phase A recomputes them; phase B reads Pd / dS from LDS and never sees it); drel [heads][2 Lm - 1] accumulates the
  // gradient of the bias per relative position key - query + (Lm - 1), through an LDS histogram per workgroup.
  // cu != NULL (packed rows, round 5): sequence b is rows cu[b] .. cu[b + 1] - 1 of qkv / dctx / dqkv, L its own row count; the
  // mask keeps its pitch Lm.  Rows past L are neither read (clamped) nor written, as for a padded sequence shorter than the tile.
  constexpr int LT = KT * 32;
  constexpr int PP = LT * 2 + 8;                       // bytes per row of the Pd / dS images
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;                               // K, later Q
  char* const sB = sA + LT * PITCH;                    // V, later dO
  char* const sP = sB + LT * PITCH;                    // Pd [query][key]
  char* const sD = sP + LT * PP;                       // dS [query][key]
  float* const sM = (float*)(sD + LT * PP);            // additive key mask (log2 domain)
  float* const sRel = sM + LT;                         // BIAS: [2 LT] bias gradient per relative position
  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  int64_t row0 = b * Lm;
  int L = Lm;
  if (cu) {
    const int c0 = __builtin_amdgcn_readfirstlane(cu[b]), c1 = __builtin_amdgcn_readfirstlane(cu[b + 1]);
    row0 = c0; L = c1 - c0;
    if (L <= 0) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ld2 = 6 * (int64_t)H;                  // row pitch of qkv / dqkv in bytes
  const char* const base = (const char*)(qkv + row0 * 3 * (int64_t)H + h * 64);
  const char* const dob = (const char*)(dctx + row0 * (int64_t)H + h * 64);
  char* const dbase = (char*)(dqkv + row0 * 3 * (int64_t)H + h * 64);
  const AttnDrop dr(drop_p);

  // ---- stage K, V (now) and fetch Q, dO (for phase B) : thread -> 16-byte chunk c of row r, four rows apart per pass
  const int sc = tid & 7, sr = tid >> 3;               // 8 chunks per row, 8 KT rows per pass
  // (named registers: hipcc keeps a uint4 array that lives across a barrier in scratch)
  uint4 rq0, rq1, rq2, rq3, rdo0, rdo1, rdo2, rdo3;
#define BW_STAGE(I)                                                                                   \
  {                                                                                                   \
    const int r = sr + (I) * 8 * KT;                                                                  \
    const int rr = r < L ? r : L - 1;                  /* rows past L repeat row L - 1 (masked / zeroed below) */ \
    const uint4 kv = *(const uint4*)(base + rr * ld2 + 2 * H + sc * 16);                              \
    const uint4 vv = *(const uint4*)(base + rr * ld2 + 4 * H + sc * 16);                              \
    rq##I = *(const uint4*)(base + rr * ld2 + sc * 16);                                               \
    rdo##I = *(const uint4*)(dob + (int64_t)rr * (2 * H) + sc * 16);                                  \
    *(uint4*)(sA + r * PITCH + sc * 16) = kv;                                                         \
    *(uint4*)(sB + r * PITCH + sc * 16) = vv;                                                         \
  }
  BW_STAGE(0) BW_STAGE(1) BW_STAGE(2) BW_STAGE(3)
#undef BW_STAGE
  const float LOG2E = 1.4426950408889634f;
  for (int k = tid; k < LT; k += 64 * KT) sM[k] = k < L ? (mask[b * Lm + k] != 0 ? 0.f : -1e30f) : -INFINITY;
  if (BIAS)
    for (int k = tid; k < 2 * LT; k += 64 * KT) sRel[k] = 0.f;

  const int q0 = wave * 32;
  const int qrow = (q0 + l31) < L ? (q0 + l31) : (L - 1);
  const bool qvalid = q0 + l31 < L;
  frag_t qf[4], dof[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    qf[kk] = *(const frag_t*)(base + (int64_t)qrow * ld2 + (kk * 2 + half) * 16);
    dof[kk] = *(const frag_t*)(dob + (int64_t)qrow * (2 * H) + (kk * 2 + half) * 16);
  }
  __syncthreads();

  // ================================================================== phase A: lane <-> query q0 + l31
  {
    f32x16_t s[KT], dp[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[t][r] = 0.f; dp[t][r] = 0.f; }
      const char* krow = sA + (t * 32 + l31) * PITCH;
      const char* vrow = sB + (t * 32 + l31) * PITCH;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const frag_t ka = *(const frag_t*)(krow + (kk * 2 + half) * 16);
        const frag_t va = *(const frag_t*)(vrow + (kk * 2 + half) * 16);
        s[t] = mma16<T>(ka, qf[kk], s[t]);       // S^T[key][query]
        dp[t] = mma16<T>(va, dof[kk], dp[t]);    // dPd^T[key][query]
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float c2 = scale * LOG2E;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t mb = *(const f32x4_t*)(sM + t * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = fmaf(s[t][4 * g + e], c2, mb[e]);
          if (BIAS) {
            const int kc = (t * 32 + 8 * g + 4 * half + e) < L ? (t * 32 + 8 * g + 4 * half + e) : (L - 1);
            v = fmaf(pos_bias[((int64_t)h * Lm + qrow) * Lm + kc], LOG2E, v);      // (the table's pitch: the padded length, also for packed rows)
          }
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
        const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // p, dropout (four decisions per hash, remembered as one bit each), delta;  s <- P, dp <- dP
    float delta = 0.f;
    uint64_t kept = 0;                               // bit 16 t + 4 g + e
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint64_t bits = ~0ull;
        if (dr.thresh) bits = attn_drop_bits(seed, b, h, heads, Lm, q0 + l31, (t * 32 + 8 * g + 4 * half) >> 2);      // (the forward's key: mask pitch)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = s[t][4 * g + e] * inv;
          const bool keep = attn_drop_keep(bits, e, dr.thresh);
          kept |= (uint64_t)keep << (16 * t + 4 * g + e);
          const float dpp = keep ? dp[t][4 * g + e] * dr.keep_scale : 0.f;
          delta = fmaf(p, dpp, delta);
          s[t][4 * g + e] = p;                  // P (the undropped probability is what dS needs)
          dp[t][4 * g + e] = dpp;               // dP
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    delta += __shfl_xor(delta, 32, 64);
    // dS = P (dP - delta) scale -> LDS + dp ;  Pd = P keep_scale [kept] -> LDS.  Rows of queries past L are zero.
    char* const prow = sP + (q0 + l31) * PP;
    char* const drow = sD + (q0 + l31) * PP;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float pd[4], ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = qvalid ? s[t][4 * g + e] : 0.f;
          pd[e] = ((kept >> (16 * t + 4 * g + e)) & 1) ? p * dr.keep_scale : 0.f;
          const float dlogit = p * (dp[t][4 * g + e] - delta);          // d loss / d (scaled score + bias)
          if (BIAS) {
            const int key = t * 32 + 8 * g + 4 * half + e;
            if (qvalid && key < L) atomicAdd(&sRel[key - (q0 + l31) + (Lm - 1)], dlogit);
          }
          ds[e] = dlogit * scale;
          dp[t][4 * g + e] = ds[e];
        }
        const int koff = (t * 32 + 8 * g + 4 * half) * 2;
        *(uint2*)(prow + koff) = make_uint2(pk2<T>(pd[0], pd[1]), pk2<T>(pd[2], pd[3]));
        *(uint2*)(drow + koff) = make_uint2(pk2<T>(ds[0], ds[1]), pk2<T>(ds[2], ds[3]));
      }
    // dQ^T[d][query] = sum_key K^T[d][key] dS[query][key] : K^T by transposing reads, dS from the registers
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    const char* const kt0 = sA + (4 * half + ((lane & 15) >> 2)) * PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const frag_t dsf = pack8<T>(dp[t], u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p = kt0 + (t * 32 + 16 * u) * PITCH + dt * 64;
          const frag_t kf = frag_of(trd(p), trd(p + 8 * PITCH));
          o[dt] = mma16<T>(kf, dsf, o[dt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    store_rows<T>(o, dbase + (int64_t)qrow * ld2, half, qvalid);
  }
  __syncthreads();                                       // Pd, dS complete; K, V no longer needed
  if (BIAS)
    for (int k = tid; k < 2 * Lm - 1; k += 64 * KT) atomicAdd(drel + (int64_t)h * (2 * Lm - 1) + k, sRel[k]);
#define BW_PUT(I)                                                                                     \
  {                                                                                                   \
    const int r = sr + (I) * 8 * KT;                                                                  \
    *(uint4*)(sA + r * PITCH + sc * 16) = rq##I;                                                      \
    *(uint4*)(sB + r * PITCH + sc * 16) = rdo##I;                                                     \
  }
  BW_PUT(0) BW_PUT(1) BW_PUT(2) BW_PUT(3)
#undef BW_PUT
  __syncthreads();

  // ================================================================== phase B: lane <-> key k0 + l31
  {
    const int k0 = wave * 32;
    f32x16_t dv[2], dk[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
    // per-lane bases of the transposing reads: row (query) 4 half + (i >> 2) of the 16-query step, column block of the lane
    const int i16 = lane & 15, gg = (lane >> 4) & 1;
    const int rowl = 4 * half + (i16 >> 2);
    const char* const xo = sB + rowl * PITCH + (16 * gg + 4 * (i16 & 3)) * 2;        // dO^T : columns d
    const char* const xq = sA + rowl * PITCH + (16 * gg + 4 * (i16 & 3)) * 2;        // Q^T
    const char* const yp = sP + rowl * PP + (k0 + 16 * gg + 4 * (i16 & 3)) * 2;      // Pd   : columns key
    const char* const yd = sD + rowl * PP + (k0 + 16 * gg + 4 * (i16 & 3)) * 2;      // dS
#pragma unroll
    for (int v = 0; v < 2 * KT; ++v) {                   // 16 queries per step
      const frag_t pf = frag_of(trd(yp + 16 * v * PP), trd(yp + (16 * v + 8) * PP));
      const frag_t df = frag_of(trd(yd + 16 * v * PP), trd(yd + (16 * v + 8) * PP));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const frag_t of = frag_of(trd(xo + 16 * v * PITCH + dt * 64), trd(xo + (16 * v + 8) * PITCH + dt * 64));
        const frag_t qf2 = frag_of(trd(xq + 16 * v * PITCH + dt * 64), trd(xq + (16 * v + 8) * PITCH + dt * 64));
        dv[dt] = mma16<T>(of, pf, dv[dt]);     // dV^T[d][key]
        dk[dt] = mma16<T>(qf2, df, dk[dt]);    // dK^T[d][key]
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool kvalid = k0 + l31 < L;
    const int krow = kvalid ? k0 + l31 : L - 1;
    store_rows<T>(dk, dbase + (int64_t)krow * ld2 + 2 * H, half, kvalid);
    store_rows<T>(dv, dbase + (int64_t)krow * ld2 + 4 * H, half, kvalid);
  }
}

template <typename T, int KT, bool BIAS>
int launch_bwd16(const void* qkv, const void* dctx, void* dqkv, const int64_t* mask, int64_t B, int L, int H, int heads,
                 float scale, float drop_p, uint64_t seed, hipStream_t s, const int* cu, const float* pos_bias, float* drel) {
  constexpr int LT = KT * 32;
  const int lds = 2 * LT * PITCH + 2 * LT * (LT * 2 + 8) + LT * 4 + (BIAS ? 2 * LT * 4 : 0);
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)attention_bwd16_kernel<T, KT, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((attention_bwd16_kernel<T, KT, BIAS>), dim3((unsigned)(heads * B)), dim3(64 * KT), lds, s, (const bf16_t*)qkv,
                     (const bf16_t*)dctx, (bf16_t*)dqkv, mask, L, H, heads, scale, drop_p, seed, cu, pos_bias, drel);
  OM_LAUNCH_CHECK();
  return 0;
}
}  // namespace

bool omk_attention_bwd16_ok(int dtype, int L, int H, int heads) {
  return (dtype == OM_BF16 || dtype == OM_F16) && L >= 1 && L <= 128 && H == heads * 64 && om_option(OM_OPT_ATTENTION_FAST);
}

int omk_attention_bwd16(int dtype, const void* qkv, const void* dctx, void* dqkv, const int64_t* mask, int64_t B, int L, int H,
                        int heads, float scale, float drop_p, uint64_t seed, hipStream_t s, const int* cu, const float* pos_bias, float* drel) {
  if (B <= 0) return 0;
  if ((pos_bias != nullptr) != (drel != nullptr)) OM_FAIL("attention backward: the position bias and its gradient buffer go together");
#define BWD16_(TT, BB)                                                                                                            \
  do {                                                                                                                            \
    if (L <= 32) return launch_bwd16<TT, 1, BB>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, s, cu, pos_bias, drel);   \
    if (L <= 64) return launch_bwd16<TT, 2, BB>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, s, cu, pos_bias, drel);   \
    return launch_bwd16<TT, 4, BB>(qkv, dctx, dqkv, mask, B, L, H, heads, scale, drop_p, seed, s, cu, pos_bias, drel);                \
  } while (0)
#define BWD16(TT) do { if (pos_bias) BWD16_(TT, true); else BWD16_(TT, false); } while (0)
  if (dtype == OM_F16) BWD16(f16_t);
  BWD16(bf16_t);
#undef BWD16
#undef BWD16_
}
