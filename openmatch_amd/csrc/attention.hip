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
#include <type_traits>

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
  if (drop_p > 0.f) {      // training: attention_probs dropout, mask regenerated in the backward (attn_common.h)
    const AttnDrop dr(drop_p);
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint64_t bits = attn_drop_bits(seed, b, h, heads, L, q0 + l31, (t * 32 + 8 * g + 4 * half) >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[t][4 * g + e] = attn_drop_keep(bits, e, dr.thresh) ? s[t][4 * g + e] * dr.keep_scale : 0.f;
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

// ---------------------------------------------------------------------------------------------------------------
// bf16 inference kernel (no dropout): the same mathematics with ~2.5x fewer instructions per wave.  The kernel above
// is ISSUE-bound at the encoder's shape (L = 128: ~1400 instructions per wave, 4 waves per SIMD -- profiles/
// r01_bench_v8_kernel_stats.csv: 244 us per layer against a 146 us HBM floor): staging through VGPRs, V transposed
// with 256 two-byte LDS stores per lane, the context transposed back through LDS.  Here
//   * K and V go to LDS row-major by LDS-DMA (4 + 4 instructions per wave, swizzle on the source address);
//   * V^T fragments come from transposing LDS reads (ds_read_b64_tr_b16) of the row-major V image: two reads hand a
//     lane its d column's eight keys in exactly the k-slot order the probabilities' registers have, so they are the
//     A operand of  O^T = V^T P^T  as read (the first version transposed V on the matrix core: D = V_tile . I, two
//     MFMAs + eight cvt_pk per fragment pair -- a third of the kernel's MFMAs and ~15 % of its instructions);
//   * O^T puts a query row in each lane: one v_permlane32_swap per dword pair makes 16 contiguous bytes, parked in the
//     wave's own K rows and written out as whole 128-byte lines;
//   * softmax in the log2 domain: v = fma(s, scale log2e, mask), exp2(v - max) -- five instructions per score.
// Masked keys carry -1e30 (finite: a fully masked row stays uniform, as with HF's finfo.min), keys past L -inf.
#include "gemm_core7.h"

typedef short v4s_a_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s_a_t vtrd(const char* p) {           // ds_read_b64_tr_b16
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_a_t __attribute__((address_space(3)))*)(p));
}
template <typename F>
__device__ __forceinline__ F vfrag_of(v4s_a_t a, v4s_a_t b) { return __builtin_bit_cast(F, (bf16x8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}); }

// DBG (timing experiments, OM_OPT_ATTENTION_DEBUG): bit 0 no K / V fetch, bit 1 no arithmetic, bit 2 no stores
// T: bf16_t or f16_t (the float16 inference mode) -- same instruction stream, the other MFMA / conversion opcodes
// KTV: the key tiles this (batch, head) walks -- KT, or fewer when its trailing tiles hold no unmasked key (round 4).  Every key
// at or past kmax[b] is masked and its probability is exactly 0 (exp2(-1e30 - max) with the max over an unmasked key): its K / V
// rows need not be fetched, scored or multiplied.  One straight-line body per tile count (the kernel below switches once per
// workgroup): run-time guards around each tile's code cost the full-length case more than the short ones gained (161 vs 148 us,
// profiles/r04_probe12_*).
template <typename T, int KT, bool BIAS, bool DROP, int DBG, int KTV>
__device__ __forceinline__ void attention_fwd16_body(
    const T* __restrict__ qkv, T* __restrict__ ctx, const int64_t* __restrict__ mask,
    const float* __restrict__ pos_bias, int L, int H, int heads, float scale, float drop_p, uint64_t seed, int rev,
    int64_t row0, int Lm) {      // row0: first row of this sequence in qkv / ctx (b * L, or its packed offset); L: ITS length; Lm: row pitch of `mask`
  typedef typename MmaOps<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem;
  char* const sV = smem + KT * 32 * 128;
  float* const sM = (float*)(smem + 2 * KT * 32 * 128);
  const int h = blockIdx.x % heads;
  // rev: batch rows last to first (workgroups start in blockIdx order) -- the rows the QKV projection wrote last are
  // still in the memory-side cache, and the output projection then starts on the context rows written last
  const int64_t b = rev ? (int64_t)(gridDim.x / heads) - 1 - blockIdx.x / heads : blockIdx.x / heads;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ld2 = 6 * (int64_t)H;                       // row pitch of qkv in bytes
  const char* const base = (const char*)(qkv + row0 * 3 * (int64_t)H + h * 64);
  // K and V rows of this (batch, head): instruction i of wave w moves rows (i * KT + w) * 8 .. + 7, lane -> row
  // (lane >> 3), physical 16-byte chunk (lane & 7) <- source chunk (lane & 7) ^ ((row >> 1) & 7) for K (row-per-lane
  // 16-byte reads), ^ 4 ((row >> 1) & 1) for V: the four rows of a transposing read then sit in the four 64-byte
  // quarters of the bank cycle.  Rows past L repeat row L - 1 (their scores are masked to -inf).
  if (!(DBG & 1))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (KTV < KT && i * KT + wave >= KTV * 4) continue;         // (wave-uniform) rows of a fully masked key tile
    const int r = (i * KT + wave) * 8 + (lane >> 3);
    const int rr = r < L ? r : L - 1;
    const uint32_t off = (uint32_t)(rr * ld2) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    const uint32_t offv = (uint32_t)(rr * ld2) + (((lane & 7) ^ (((r >> 1) & 1) << 2)) << 4);
    const uint32_t dst = (uint32_t)((i * KT + wave) * 1024);
    g7_dma(base + 2 * H, off, g7_lds_addr(sK) + dst);
    g7_dma(base + 4 * H, offv, g7_lds_addr(sV) + dst);
  }
  const float LOG2E = 1.4426950408889634f;
  for (int k = tid; k < KT * 32; k += 64 * KT)
    sM[k] = k < L ? (mask[b * Lm + k] != 0 ? 0.f : -1e30f) : -INFINITY;

  const int q0 = wave * 32;
  const int qrow = (q0 + l31) < L ? (q0 + l31) : (L - 1);
  frag_t qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const frag_t*)(base + (int64_t)qrow * ld2 + (kk * 2 + half) * 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMA above is not in hipcc's bookkeeping
  __syncthreads();
  if (q0 >= L) return;

  constexpr int KTC = (DBG & 2) ? 0 : KTV;         // key tiles the arithmetic walks
  // S^T = K Q^T : lane owns query l31, keys (r&3) + 8(r>>2) + 4 half of each 32-key tile
  const int key = (l31 >> 1) & 7;
  f32x16_t s[KTV];
#pragma unroll
  for (int t = 0; t < KTC; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    const char* krow = sK + (t * 32 + l31) * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const frag_t a = *(const frag_t*)(krow + (((kk * 2 + half) ^ key) << 4));
      MmaOps<T>::mma(a, qf[kk], s[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const float c2 = scale * LOG2E;
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < KTC; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k0 = t * 32 + 8 * g + 4 * half;
      const f32x4_t mb = *(const f32x4_t*)(sM + k0);
      f32x4_t pb = {0.f, 0.f, 0.f, 0.f};
      if (BIAS) {              // T5: bias[h][q][k], four consecutive keys per load (L % 4 == 0 is not required: clamp)
        const float* pr = pos_bias + ((int64_t)h * Lm + qrow) * Lm;      // (the table's pitch: the padded length, also for packed rows)
#pragma unroll
        for (int e = 0; e < 4; ++e) pb[e] = pr[(k0 + e) < Lm ? (k0 + e) : (Lm - 1)] * LOG2E;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(s[t][4 * g + e], c2, mb[e]) + pb[e];
        s[t][4 * g + e] = v;
        mx = fmaxf(mx, v);
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the 16 mask vectors from being fetched all at once
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < KTC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
      s[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  float inv = 1.0f / sum;
  if (DROP) {       // training forward: attention_probs dropout on the (still unnormalised) probabilities; the mask is
                    // regenerated by the backward from (seed, b, h, q, key group) -- attn_common.h
    const AttnDrop dr(drop_p);
#pragma unroll
    for (int t = 0; t < KTC; ++t)
  #pragma unroll
      for (int g = 0; g < 4; ++g) {
        // (keyed with the mask's row pitch Lm, not the sequence's own length: a packed and a padded step draw the same mask)
        const uint64_t bits = attn_drop_bits(seed, b, h, heads, Lm, q0 + l31, (t * 32 + 8 * g + 4 * half) >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[t][4 * g + e] = attn_drop_keep(bits, e, dr.thresh) ? s[t][4 * g + e] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
      }
    inv *= dr.keep_scale;
  }

  // V^T fragments by transposing LDS reads: lane (i = lane & 15, gg = lane bit 4, half) names 8-byte chunk i of the
  // [4 keys][16 d] block (keys 4 half .. + 3 of a 16-key step, d block 16 gg) and receives its d column's four keys --
  // two reads are the eight k slots of a fragment in the order the probabilities' registers have.
  const int i16 = lane & 15;
  const char* const vt0 = sV + (4 * half + (i16 >> 2)) * 128 + 32 * ((lane >> 4) & 1) + 8 * (i16 & 3);
  const int vsw = (i16 >> 3) & 1;               // the row's swizzle bit: (row >> 1) & 1 with row = 8 n + 4 half + (i16 >> 2)
  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int t = 0; t < KTC; ++t) {
    uint4 pa[2];      // probabilities of this key tile as two k slabs (k slot e of half h <-> register 8u + e)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      pa[u] = make_uint4(Half16<T>::pack2(s[t][8 * u + 0], s[t][8 * u + 1]), Half16<T>::pack2(s[t][8 * u + 2], s[t][8 * u + 3]),
                         Half16<T>::pack2(s[t][8 * u + 4], s[t][8 * u + 5]), Half16<T>::pack2(s[t][8 * u + 6], s[t][8 * u + 7]));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const char* p = vt0 + (t * 32 + 16 * u) * 128 + ((dt ^ vsw) << 6);
        const frag_t vf = vfrag_of<frag_t>(vtrd(p), vtrd(p + 8 * 128));
        MmaOps<T>::mma(vf, __builtin_bit_cast(frag_t, pa[u]), o[dt]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (DBG & 2) {                                    // keep the Q loads alive
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o[0][kk] = (float)qf[kk][0];
  }
  // o[dt][r] = O[query l31][d = 32 dt + (r&3) + 8(r>>2) + 4 half]: scale by 1/sum, pack to bf16 -- 16 bytes per
  // (dt, register group of 8) after pairing the two halves' dwords.  Stored straight from this layout an instruction
  // writes 32 bytes to each of 32 rows; instead the block goes through the wave's own K rows (all waves are past K and
  // V after the barrier; waves without queries have exited and do not count) and leaves as whole 128-byte rows, 8 per
  // instruction.  16-byte chunk c of row r sits at chunk c ^ (r & 7): writes and reads are both conflict-free.
  // (Either way L2 writes exactly the context bytes -- WRITE_SIZE 201 MB per launch, profiles/r02_hbm_traffic.json -- and
  // the kernel runs at the memory system's pace: 805 MB per layer at ~5.2 TB/s, profiles/r02_attention_variants.log.)
  __syncthreads();
  char* const so = sK + q0 * 128;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      uint32_t a0 = Half16<T>::pack2(o[dt][8 * gp + 0] * inv, o[dt][8 * gp + 1] * inv), a1 = Half16<T>::pack2(o[dt][8 * gp + 2] * inv, o[dt][8 * gp + 3] * inv);
      uint32_t b0 = Half16<T>::pack2(o[dt][8 * gp + 4] * inv, o[dt][8 * gp + 5] * inv), b1 = Half16<T>::pack2(o[dt][8 * gp + 6] * inv, o[dt][8 * gp + 7] * inv);
      // (a: d group 2gp, b: d group 2gp + 1) -- lanes 32-63 of a swap with lanes 0-31 of b
      auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      // lanes 0-31: d = 32 dt + 16 gp + 0..7 ; lanes 32-63: d = 32 dt + 16 gp + 8..15   (chunk 4 dt + 2 gp + half)
      *(uint4*)(so + l31 * 128 + (((4 * dt + 2 * gp + half) ^ (l31 & 7)) << 4)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
  char* const out = (char*)(ctx + (row0 + q0) * (int64_t)H + h * 64);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), c = lane & 7;
    const uint4 v = *(const uint4*)(so + row * 128 + ((c ^ (row & 7)) << 4));
    if (!(DBG & 4) || v.x == 0x12345u) { if (q0 + row < L) *(uint4*)(out + (int64_t)row * H * 2 + c * 16) = v; }
  }
}

template <typename T, int KT, bool BIAS, bool DROP, int DBG = 0>
__global__ __launch_bounds__(64 * KT, KT <= 4 ? ((BIAS || DROP) ? 3 : 4) : 2) void attention_fwd16_kernel(
    const T* __restrict__ qkv, T* __restrict__ ctx, const int64_t* __restrict__ mask,
    const float* __restrict__ pos_bias, int L, int H, int heads, float scale, float drop_p, uint64_t seed, int rev,
    const int* __restrict__ kmax, const int* __restrict__ cu) {
#define OM_ATTN_BODY(V) attention_fwd16_body<T, KT, BIAS, DROP, DBG, V>(qkv, ctx, mask, pos_bias, Lb, H, heads, scale, drop_p, seed, rev, row0, L)
  const int64_t b = rev ? (int64_t)(gridDim.x / heads) - 1 - blockIdx.x / heads : blockIdx.x / heads;
  int64_t row0 = b * L;
  int Lb = L;
  if (cu) {                                          // packed rows (om_encoder_forward_packed): sequence b is rows cu[b] .. cu[b + 1] - 1
    const int c0 = __builtin_amdgcn_readfirstlane(cu[b]), c1 = __builtin_amdgcn_readfirstlane(cu[b + 1]);
    row0 = c0; Lb = c1 - c0;
    if (Lb <= 0) return;
  }
  if (KT == 4 && !BIAS && !DROP && (kmax || cu)) {   // (the encoder's inference shape; kmax[b] = L for a row without any unmasked key)
    const int kt = ((cu ? Lb : __builtin_amdgcn_readfirstlane(kmax[b])) + 31) >> 5;
    if (kt <= 1) { OM_ATTN_BODY(1); return; }
    if (kt == 2) { OM_ATTN_BODY(2); return; }
    if (kt == 3) { OM_ATTN_BODY(3); return; }
  }
  OM_ATTN_BODY(KT);
#undef OM_ATTN_BODY
}

template <typename T, int KT, bool BIAS, bool DROP>
static int launch_attn16_(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias, int64_t B, int L, int H,
                         int heads, float scale, float drop_p, uint64_t seed, hipStream_t s, int rev, const int* kmax, const int* cu = nullptr) {
  const int lds = 2 * KT * 32 * 128 + KT * 32 * 4;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)attention_fwd16_kernel<T, KT, BIAS, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  // The timing variants (parts of the kernel compiled out: their results are WRONG) exist in probe builds only
  // (-DOM_PROBE_KERNELS: python -m openmatch_amd._build --probe); the product library ignores OM_OPT_ATTENTION_DEBUG.
#ifdef OM_PROBE_KERNELS
  const int dbg = (sizeof(T) == 2 && std::is_same<T, bf16_t>::value && KT == 4 && !BIAS && !DROP) ? om_option(OM_OPT_ATTENTION_DEBUG) : 0;
#else
  const int dbg = 0;
#endif
  if (dbg) {
#ifdef OM_PROBE_KERNELS
#define OM_ATTN_DBG(D)                                                                                                        \
  case D:                                                                                                                     \
    hipLaunchKernelGGL((attention_fwd16_kernel<bf16_t, 4, false, false, D>), dim3((unsigned)(heads * B)), dim3(256), lds, s,          \
                       (const bf16_t*)qkv, (bf16_t*)ctx, mask, pos_bias, L, H, heads, scale, drop_p, seed, rev, kmax, cu);    \
    break;
    switch (dbg) { OM_ATTN_DBG(1) OM_ATTN_DBG(2) OM_ATTN_DBG(3) OM_ATTN_DBG(4) OM_ATTN_DBG(5) OM_ATTN_DBG(6) OM_ATTN_DBG(7) default: break; }
#undef OM_ATTN_DBG
#endif
    OM_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((attention_fwd16_kernel<T, KT, BIAS, DROP>), dim3((unsigned)(heads * B)), dim3(64 * KT), lds, s, (const T*)qkv,
                     (T*)ctx, mask, pos_bias, L, H, heads, scale, drop_p, seed, rev, kmax, cu);
  OM_LAUNCH_CHECK();
  return 0;
}
template <typename T, int KT>
static int launch_attn16(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias, int64_t B, int L, int H,
                         int heads, float scale, float drop_p, uint64_t seed, hipStream_t s, int rev, const int* kmax, const int* cu = nullptr) {
  if (std::is_same<T, f16_t>::value) {            // float16: BERT-family encoders; dropout for float16 training, T5's bias table for inference (round 5)
    if (pos_bias && drop_p > 0.f) return launch_attn16_<f16_t, KT, true, true>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, rev, kmax, cu);   // float16 T5 training (round 6)
    if (pos_bias) return launch_attn16_<f16_t, KT, true, false>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, 0.f, 0, s, rev, kmax, cu);
    if (drop_p > 0.f) return launch_attn16_<f16_t, KT, false, true>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, rev, kmax, cu);
    return launch_attn16_<T, KT, false, false>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, 0.f, 0, s, rev, kmax, cu);
  }
  if (drop_p > 0.f) {
    if (pos_bias) return launch_attn16_<bf16_t, KT, true, true>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, rev, kmax, cu);
    return launch_attn16_<bf16_t, KT, false, true>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, rev, kmax, cu);
  }
  if (pos_bias) return launch_attn16_<bf16_t, KT, true, false>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, 0.f, 0, s, rev, kmax, cu);
  return launch_attn16_<bf16_t, KT, false, false>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, 0.f, 0, s, rev, kmax, cu);
}

// ---------------------------------------------------------------------------------------------------------------
// Beyond 256 tokens in the 16-bit formats (round 6): the body of the kernel above inside a loop over 128-key chunks with the online
// softmax.  A workgroup owns 128 queries of one (sequence, head) -- four waves of 32 -- and per chunk: K and V rows by LDS-DMA
// (row-major, swizzled on the source address), S^T = K Q^T, exp2 softmax against the running maximum, V^T fragments by transposing
// LDS reads.  O^T keeps a query per LANE, so the rescaling by exp2(m_old - m_new) is one multiply per accumulator register with the
// lane's own factor (the f32 kernel below keeps queries in registers and sends the factors through an LDS table).  Dropout, the T5
// bias table and packed rows as in the kernel above.  Measured at 16 x 512 tokens, 12 heads: profiles/r06_train_long_sequences.txt.
template <typename T, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attention_fwd16c_kernel(
    const T* __restrict__ qkv, T* __restrict__ ctx, const int64_t* __restrict__ mask,
    const float* __restrict__ pos_bias, int Lm, int H, int heads, float scale, float drop_p, uint64_t seed, const int* __restrict__ cu) {
  typedef typename MmaOps<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem;
  char* const sV = smem + 128 * 128;
  float* const sM = (float*)(smem + 2 * 128 * 128);
  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  int64_t row0 = b * Lm;
  int L = Lm;
  if (cu) {
    const int c0 = __builtin_amdgcn_readfirstlane(cu[b]), c1 = __builtin_amdgcn_readfirstlane(cu[b + 1]);
    row0 = c0; L = c1 - c0;
  }
  const int qb = blockIdx.y * 128;
  if (qb >= L) return;                                      // (whole workgroup)
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ld2 = 6 * (int64_t)H;                       // row pitch of qkv in bytes
  const char* const base = (const char*)(qkv + row0 * 3 * (int64_t)H + h * 64);
  const float LOG2E = 1.4426950408889634f;
  const int q0 = qb + wave * 32;
  const bool active = q0 < L;                               // (wave-uniform; an inactive wave still fetches its share of every chunk)
  const int qrow = (q0 + l31) < L ? (q0 + l31) : (L - 1);
  frag_t qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const frag_t*)(base + (int64_t)qrow * ld2 + (kk * 2 + half) * 16);
  const float c2 = scale * LOG2E;
  const AttnDrop dr(drop_p);
  const int key = (l31 >> 1) & 7;
  const int i16 = lane & 15;
  const char* const vt0 = sV + (4 * half + (i16 >> 2)) * 128 + 32 * ((lane >> 4) & 1) + 8 * (i16 & 3);
  const int vsw = (i16 >> 3) & 1;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  for (int kc = 0; kc < L; kc += 128) {
    __syncthreads();                                         // the previous chunk has been consumed by every wave
    // K and V rows kc .. kc + 127: instruction i of wave w moves rows (i * 4 + w) * 8 .. + 7, lane -> row (lane >> 3), physical
    // 16-byte chunk (lane & 7) <- source chunk (lane & 7) ^ ((row >> 1) & 7) for K, ^ 4 ((row >> 1) & 1) for V (the kernel above)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (i * 4 + wave) * 8 + (lane >> 3);
      const int rr = (kc + r) < L ? (kc + r) : (L - 1);
      const uint32_t off = (uint32_t)(rr * ld2) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
      const uint32_t offv = (uint32_t)(rr * ld2) + (((lane & 7) ^ (((r >> 1) & 1) << 2)) << 4);
      const uint32_t dst = (uint32_t)((i * 4 + wave) * 1024);
      g7_dma(base + 2 * H, off, g7_lds_addr(sK) + dst);
      g7_dma(base + 4 * H, offv, g7_lds_addr(sV) + dst);
    }
    if (tid < 128) sM[tid] = (kc + tid) < L ? (mask[b * Lm + kc + tid] != 0 ? 0.f : -1e30f) : -INFINITY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMA above is not in hipcc's bookkeeping
    __syncthreads();
    if (!active) continue;

    f32x16_t s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      const char* krow = sK + (t * 32 + l31) * 128;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const frag_t a = *(const frag_t*)(krow + (((kk * 2 + half) ^ key) << 4));
        MmaOps<T>::mma(a, qf[kk], s[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float mx = m_run;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = t * 32 + 8 * g + 4 * half;
        const f32x4_t mb = *(const f32x4_t*)(sM + k0);
        f32x4_t pb = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) {
          const float* pr = pos_bias + ((int64_t)h * Lm + qrow) * Lm;      // (the table's pitch: the padded length, also for packed rows)
#pragma unroll
          for (int e = 0; e < 4; ++e) pb[e] = pr[(kc + k0 + e) < Lm ? (kc + k0 + e) : (Lm - 1)] * LOG2E;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(s[t][4 * g + e], c2, mb[e]) + pb[e];
          s[t][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // the first chunk always holds key 0 (unmasked, or -1e30: finite): mx is finite from here on
    const float alpha = __builtin_amdgcn_exp2f(m_run - mx);   // exp2(-inf) = 0 on the first chunk
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = mx;
    if (DROP) {       // (keyed with the mask's row pitch Lm: the backward kernels and a packed step regenerate the same mask)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint64_t bits = attn_drop_bits(seed, b, h, heads, Lm, q0 + l31, (kc + t * 32 + 8 * g + 4 * half) >> 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[t][4 * g + e] = attn_drop_keep(bits, e, dr.thresh) ? s[t][4 * g + e] : 0.f;
          __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;        // O^T: this lane's query in every register
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint4 pa[2];      // probabilities of this key tile as two k slabs (k slot e of half h <-> register 8u + e)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        pa[u] = make_uint4(Half16<T>::pack2(s[t][8 * u + 0], s[t][8 * u + 1]), Half16<T>::pack2(s[t][8 * u + 2], s[t][8 * u + 3]),
                           Half16<T>::pack2(s[t][8 * u + 4], s[t][8 * u + 5]), Half16<T>::pack2(s[t][8 * u + 6], s[t][8 * u + 7]));
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p = vt0 + (t * 32 + 16 * u) * 128 + ((dt ^ vsw) << 6);
          const frag_t vf = vfrag_of<frag_t>(vtrd(p), vtrd(p + 8 * 128));
          MmaOps<T>::mma(vf, __builtin_bit_cast(frag_t, pa[u]), o[dt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // O / l (x the keep scale): through the wave's own K rows, out as whole 128-byte rows (the kernel above)
  __syncthreads();
  if (!active) return;
  const float inv = (DROP ? dr.keep_scale : 1.0f) / l_run;
  char* const so = sK + (wave * 32) * 128;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      uint32_t a0 = Half16<T>::pack2(o[dt][8 * gp + 0] * inv, o[dt][8 * gp + 1] * inv), a1 = Half16<T>::pack2(o[dt][8 * gp + 2] * inv, o[dt][8 * gp + 3] * inv);
      uint32_t b0 = Half16<T>::pack2(o[dt][8 * gp + 4] * inv, o[dt][8 * gp + 5] * inv), b1 = Half16<T>::pack2(o[dt][8 * gp + 6] * inv, o[dt][8 * gp + 7] * inv);
      auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      *(uint4*)(so + l31 * 128 + (((4 * dt + 2 * gp + half) ^ (l31 & 7)) << 4)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
  char* const out = (char*)(ctx + (row0 + q0) * (int64_t)H + h * 64);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), c = lane & 7;
    const uint4 v = *(const uint4*)(so + row * 128 + ((c ^ (row & 7)) << 4));
    if (q0 + row < L) *(uint4*)(out + (int64_t)row * H * 2 + c * 16) = v;
  }
}

template <typename T, bool BIAS, bool DROP>
static int launch_attn16c_(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias, int64_t B, int L, int H,
                           int heads, float scale, float drop_p, uint64_t seed, hipStream_t s, const int* cu) {
  const int lds = 2 * 128 * 128 + 128 * 4;
  hipLaunchKernelGGL((attention_fwd16c_kernel<T, BIAS, DROP>), dim3((unsigned)(heads * B), (unsigned)((L + 127) / 128)), dim3(256), lds, s,
                     (const T*)qkv, (T*)ctx, mask, pos_bias, L, H, heads, scale, drop_p, seed, cu);
  OM_LAUNCH_CHECK();
  return 0;
}
template <typename T>
static int launch_attn16c(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias, int64_t B, int L, int H,
                          int heads, float scale, float drop_p, uint64_t seed, hipStream_t s, const int* cu) {
  if (pos_bias && drop_p > 0.f) return launch_attn16c_<T, true, true>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, cu);
  if (pos_bias) return launch_attn16c_<T, true, false>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, 0.f, 0, s, cu);
  if (drop_p > 0.f) return launch_attn16c_<T, false, true>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, cu);
  return launch_attn16c_<T, false, false>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, 0.f, 0, s, cu);
}

// ---------------------------------------------------------------------------------------------------------------
// Long sequences (256 < L <= 1024, inference): the kernels above keep a whole score row per lane (L / 2 registers) and
// all of K, V in LDS, which stops at 256 keys.  Document corpora are encoded at 512 tokens (the reference accepts anything
// up to max_position_embeddings).  Here a workgroup owns 128 QUERIES of one (batch, head) -- four waves of 32 -- and walks
// the keys in chunks of 128 with the online softmax: running maximum m and sum l per query, O rescaled by
// exp(m_old - m_new) per chunk.  K chunk row-major (swizzled), V chunk transposed, as in attention_kernel<T, 4>; the
// accumulators O[query][d] keep queries in REGISTERS and d in lanes (SlabMma), so the per-query factors travel through a
// 32-float LDS table per wave.  Constant registers and LDS for any L; K / V are re-read once per 128 queries (L2).
template <typename T>
__global__ __launch_bounds__(256) void attention_long_kernel(
    const T* __restrict__ qkv, T* __restrict__ ctx, const int64_t* __restrict__ mask,
    const float* __restrict__ pos_bias, int Lm, int H, int heads, float scale, float drop_p, uint64_t seed, const int* __restrict__ cu) {
  // cu != NULL (packed rows beyond 256 tokens, round 6): sequence b is rows cu[b] .. cu[b + 1] - 1 of qkv / ctx, L its own row count; the mask,
  // the bias table and the dropout hash keep the padded pitch Lm.
  // drop_p > 0 (round 6: training beyond 256 tokens): the probabilities that meet V are masked with the (sequence, head, query, key)
  // hash the backward regenerates (attn_common.h); the normaliser is the sum of the unmasked ones, as in the other kernels
  typedef AttnGeom<T> G;
  typedef typename MmaOps<T>::frag_t frag_t;
  constexpr int LP = 128 + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  T* sVt = (T*)(smem + 128 * G::ROWB);
  float* sM = (float*)(smem + 128 * G::ROWB + 64 * LP * (int)sizeof(T));
  float* sF = sM + 128;                                     // [4 waves][32] per-query factors

  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int qb = blockIdx.y * 128;
  const int tid = threadIdx.x;
  const int64_t ld = 3 * (int64_t)H;
  int64_t row0 = b * Lm;
  int L = Lm;
  if (cu) { row0 = cu[b]; L = cu[b + 1] - cu[b]; }
  if (qb >= L) return;                                       // (whole workgroup: a sequence shorter than this query block)
  const T* base = qkv + row0 * ld + h * 64;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int q0 = qb + wave * 32;
  const int qrow = (q0 + l31) < L ? (q0 + l31) : (L - 1);
  frag_t qf[G::NKK];
#pragma unroll
  for (int kk = 0; kk < G::NKK; ++kk) qf[kk] = *(const frag_t*)(base + (int64_t)qrow * ld + (kk * 2 + half) * G::EPC);

  const AttnDrop dr_(drop_p);
  const uint32_t thresh = dr_.thresh;
  const float keep_scale = dr_.keep_scale;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  for (int kc = 0; kc < L; kc += 128) {
    __syncthreads();                                         // the previous chunk has been consumed by every wave
    for (int idx = tid; idx < 128 * G::CPR; idx += 256) {
      const int row = idx / G::CPR, c = idx % G::CPR;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (kc + row < L) {
        kv = *(const uint4*)(base + (int64_t)(kc + row) * ld + H + c * G::EPC);
        vv = *(const uint4*)(base + (int64_t)(kc + row) * ld + 2 * H + c * G::EPC);
      }
      *(uint4*)(sK + row * G::ROWB + ((c ^ G::key(row)) << 4)) = kv;
      const T* ve = (const T*)&vv;
#pragma unroll
      for (int e = 0; e < G::EPC; ++e) sVt[(c * G::EPC + e) * LP + row] = ve[e];
    }
    if (tid < 128) sM[tid] = (kc + tid) < L ? (mask[b * Lm + kc + tid] != 0 ? 0.f : -3.4028235e38f) : -INFINITY;
    __syncthreads();

    f32x16_t s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
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
    float mx = m_run;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = t * 32 + 8 * g + 4 * half;
        const f32x4_t mb = *(const f32x4_t*)(sM + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = s[t][4 * g + e] * scale;
          if (pos_bias) {
            const int kcol = (kc + k0 + e) < L ? (kc + k0 + e) : (L - 1);
            v += pos_bias[((int64_t)h * Lm + qrow) * Lm + kcol];
          }
          v += mb[e];
          s[t][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // the first chunk always holds key 0 (unmasked [CLS] or finfo.min, finite): mx is finite from here on
    const float alpha = G::exp_(m_run - mx);                 // exp(-inf) = 0 on the first chunk
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = G::exp_(s[t][r] - mx);
        sum += e;
        s[t][r] = (thresh && !attn_drop_keep1(seed, b, h, heads, Lm, q0 + l31, kc + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, thresh)) ? 0.f : (thresh ? e * keep_scale : e);
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = mx;
    // rescale O: the factor of query q lives in lane q; O holds queries in registers -> through the wave's table
    if (half == 0) sF[wave * 32 + l31] = alpha;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (same wave: LDS operations execute in order)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_t a4 = *(const f32x4_t*)(sF + wave * 32 + 8 * g + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[0][4 * g + e] *= a4[e]; o[1][4 * g + e] *= a4[e]; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) SlabMma<T>::run(s[t], sVt + l31 * LP + t * 32 + 4 * half, LP, o);
  }
  // O / l, parked in the wave's own K rows, stored as whole 16-byte vectors
  __syncthreads();
  if (half == 0) sF[wave * 32 + l31] = 1.0f / l_run;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  T* so = (T*)(sK + (size_t)(wave * 32) * G::ROWB);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4_t i4 = *(const f32x4_t*)(sF + wave * 32 + 8 * g + 4 * half);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = 8 * g + 4 * half + e;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ElemOps<T>::store(so + q * 64 + dt * 32 + l31, o[dt][4 * g + e] * i4[e]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (q0 < L) {
    T* out = ctx + (row0 + q0) * H + h * 64;
    constexpr int VPR = G::ROWB / 16;
#pragma unroll
    for (int it = 0; it < 32 * VPR / 64; ++it) {
      const int idx = it * 64 + lane, row = idx / VPR, c = idx % VPR;
      const uint4 v = *(const uint4*)((const char*)so + row * G::ROWB + c * 16);
      if (q0 + row < L) *(uint4*)((char*)(out + (int64_t)row * H) + c * 16) = v;
    }
  }
}

template <typename T>
static int launch_attn_long(const void* qkv, void* ctx, const int64_t* mask, const float* pos_bias, int64_t B, int L, int H,
                            int heads, float scale, hipStream_t s, float drop_p = 0.f, uint64_t seed = 0, const int* cu = nullptr) {
  const int lds = 128 * AttnGeom<T>::ROWB + 64 * 132 * (int)sizeof(T) + 128 * 4 + 128 * 4;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)attention_long_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((attention_long_kernel<T>), dim3((unsigned)(heads * B), (unsigned)((L + 127) / 128)), dim3(256), lds, s,
                     (const T*)qkv, (T*)ctx, mask, pos_bias, L, H, heads, scale, drop_p, seed, cu);
  OM_LAUNCH_CHECK();
  return 0;
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

// kmax[b] = 1 + the last unmasked key of batch row b (L when the row has none): once per forward, read by every layer's
// attention launch (attention_fwd16_kernel: key tiles past it are skipped)
__global__ void mask_extent_kernel(const int64_t* __restrict__ mask, int64_t B, int L, int* __restrict__ kmax) {
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  int last = 0;
  for (int k = lane; k < L; k += 64) if (mask[b * L + k] != 0) last = k + 1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
  if (lane == 0) kmax[b] = last ? last : L;
}
// packed rows: cu[0..B] = exclusive scan of kmax clamped to `rows`; row_map[t] = b * L + pos of packed row t (-1 for the pad rows
// up to `rows`).  One workgroup (B is a few thousand at most).  More tokens than `rows`: the sequences past the bound get the
// rows that are left (possibly none) so that no kernel leaves the buffers, cu[B + 1] keeps the true count, and the pooling
// tail turns every representation into NaN (omk_pack_overflow_poison) instead of returning a silently truncated batch.
__global__ __launch_bounds__(1024) void pack_rows_kernel(const int* __restrict__ kmax, int64_t B, int L, int64_t rows, int* __restrict__ cu,
                                                          int* __restrict__ cls_rows, int* __restrict__ row_map) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (B + 1023) / 1024, lo = tid * per, hi = lo + per < B ? lo + per : B;
  int sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += kmax[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;                       // exclusive prefix of this thread's rows
  const int cap = (int)rows;
  for (int64_t i = lo; i < hi; ++i) {
    cu[i] = run < cap ? run : cap;
    cls_rows[i] = run < cap ? run : cap - 1;
    const int n = kmax[i];
    for (int k = 0; k < n; ++k) if (run + k < cap) row_map[run + k] = (int)(i * L + k);
    run += n;
  }
  const int total = part[1023];
  if (tid == 1023) { cu[B] = total < cap ? total : cap; cu[B + 1] = total; }
  for (int64_t t = total + tid; t < rows; t += 1024) row_map[t] = -1;
}
int omk_pack_rows(const int* kmax, int64_t B, int L, int64_t rows, int* cu, int* cls_rows, int* row_map, hipStream_t s) {
  if (B <= 0) return 0;
  if (rows > 0x7fffffff || B * (int64_t)L > 0x7fffffff) OM_FAIL("packed rows: token counts below 2^31");
  hipLaunchKernelGGL(pack_rows_kernel, dim3(1), dim3(1024), 0, s, kmax, B, L, rows, cu, cls_rows, row_map);
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_mask_extent(const int64_t* mask, int64_t B, int L, int* kmax, hipStream_t s) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(mask_extent_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, mask, B, L, kmax);
  OM_LAUNCH_CHECK();
  return 0;
}

int omk_attention(int dtype, const void* qkv, void* ctx, const int64_t* mask,
                  const float* pos_bias, int64_t B, int L, int H, int heads, float scale,
                  float drop_p, uint64_t seed, hipStream_t s, int reverse, const int* kmax, const int* cu) {
  if (B <= 0) return 0;
  if (cu && !(dtype == OM_F16 || (dtype == OM_BF16 && om_option(OM_OPT_ATTENTION_FAST))))
    OM_FAIL("packed rows: the 16-bit attention kernels");      // (with dropout too: the packed training forward, round 5; beyond 256 tokens: round 6)
  if (L < 1 || L > 1024) OM_FAIL("sequence length must be in [1,1024]");
  if (L > 256 && drop_p > 0.f && (L > 512 || dtype == OM_F32)) OM_FAIL("attention with dropout: up to 512 tokens in the 16-bit formats (float32: 256)");
  if (H != heads * 64) OM_FAIL("head_dim must be 64");
  if (B * heads > 0x7fffffffLL) OM_FAIL("batch too large for one launch");
  // OM_OPT_ATTENTION_FAST bit 1 (tests): the tile-at-a-time kernels of L > 256 at every length (their masks and results must agree with the others')
  const bool force_long = (om_option(OM_OPT_ATTENTION_FAST) & 2) != 0 && dtype != OM_F32;
  if (dtype == OM_F16) {                                      // float16 inference mode: the fast kernel only
    if (L > 256 || force_long) {
      if (!(om_option(OM_OPT_ATTENTION_FAST) & 4)) return launch_attn16c<f16_t>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, cu);
      return launch_attn_long<f16_t>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, s, drop_p, seed, cu);      // (bit 2, A/B: the first online-softmax kernel)
    }
    if (L <= 32) return launch_attn16<f16_t, 1>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    if (L <= 64) return launch_attn16<f16_t, 2>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    if (L <= 128) return launch_attn16<f16_t, 4>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    if (L <= 192) return launch_attn16<f16_t, 6>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    return launch_attn16<f16_t, 8>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
  }
  if (L > 256 || force_long) {                                // online-softmax kernel, any dtype
    if (dtype == OM_BF16 && om_option(OM_OPT_ATTENTION_FAST) && !(om_option(OM_OPT_ATTENTION_FAST) & 4))
      return launch_attn16c<bf16_t>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, cu);
    if (dtype == OM_BF16) return launch_attn_long<bf16_t>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, s, drop_p, seed, cu);
    return launch_attn_long<float>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, s);
  }
  if (dtype == OM_BF16 && om_option(OM_OPT_ATTENTION_FAST)) {        // the low-instruction-count kernel (inference, and training with dropout)
    if (L <= 32) return launch_attn16<bf16_t, 1>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    if (L <= 64) return launch_attn16<bf16_t, 2>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    if (L <= 128) return launch_attn16<bf16_t, 4>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    if (L <= 192) return launch_attn16<bf16_t, 6>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
    return launch_attn16<bf16_t, 8>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s, reverse, kmax, cu);
  }
  if (dtype == OM_BF16) return dispatch_attn<bf16_t>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
  return dispatch_attn<float>(qkv, ctx, mask, pos_bias, B, L, H, heads, scale, drop_p, seed, s);
}

extern "C" int om_debug_attention(const void* qkv, void* ctx, const int64_t* mask, int64_t B, int L, int H, int heads, void* stream) {
  if (!qkv || !ctx || !mask) OM_FAIL("null argument");
  return omk_attention(OM_BF16, qkv, ctx, mask, nullptr, B, L, H, heads, 0.125f, 0.f, 0, (hipStream_t)stream);
}
