// Exact inner-product top-k over a resident index shard (K12 + K13 + K14), gfx950.
// Replaces faiss.IndexFlatIP.search and the faiss-GPU shard merge
// (retriever/dense_retriever.py:38-58,180; utils.py:215-229).
//
// The [Q x N] score matrix is never materialised.  The scan is a threshold-filtered GEMM:
//   1. bootstrap: score the first few thousand rows densely, sort, theta_q = k-th best so far
//      (the k-th best of ANY subset is a lower bound of the final k-th best);
//   2. scan the remaining rows in geometrically growing chunks with the MFMA main loop of
//      gemm_core.h (A = index rows, B = queries -> every lane owns one query column); the
//      epilogue appends (score,row) to the query's candidate list only if score >= theta_q
//      -- the expected number of survivors per chunk is k * chunk / rows_seen;
//   3. after each chunk a per-query bitonic sort (LDS, 8192 keys) keeps the best k and
//      raises theta_q.  If a list overflows (adversarial row order) the chunk is redone
//      through the dense path, which is always exact.
// Scores in OM_SEARCH_F32 mode come from the exact-f32 MFMA (k-ordered fmaf chain).
// In OM_SEARCH_F16_RESCORE mode the scan runs on the IEEE-f16 shadow index with a CERTIFIED
// margin: |s_f16 - s_f32| <= E_q = ||q||*max||p-f16(p)|| + ||q-f16(q)||*max||f16(p)|| (+ f32
// accumulation slop), the list keeps everything within 2*E_q of the running k-th best, and
// the survivors are re-scored with exact f32 dot products before the final top-k -- so the
// returned ids are those of the f32 scan.
#include <stdlib.h>

#include <algorithm>

#include "gemm_core2.h"
#include "gemm_core6.h"
#include "gemm_wide7.h"
#include "kernels.h"

#define SORT_CAP 8192      // keys one workgroup sorts in LDS (64 KiB)
#define DENSE_CHUNK 4096   // rows scored densely per bootstrap / fallback step
#define LIST_MAX (SORT_CAP - DENSE_CHUNK)
#define K_MAX 2048
#define SORT_THREADS 512

typedef unsigned long long u64;
static thread_local int64_t g_info[8];   // last om_sim_topk call: see om_sim_topk_info()

__host__ __device__ inline u64 pack_key(float score, uint32_t payload) {
  return ((u64)f32_orderable(score) << 32) | (u64)(uint32_t)~payload;
}
__host__ __device__ inline float key_score(u64 k) { return orderable_f32((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t key_payload(u64 k) { return ~(uint32_t)k; }

// ---- filtered scan: A = index rows [row0, row0+nrows), B = queries ----------------------
template <typename T>
__global__ __launch_bounds__(G2_THREADS) void sim_filter_kernel(
    const T* __restrict__ rows, int64_t nrows, uint32_t row_base, const T* __restrict__ queries,
    int64_t nq, int64_t d, const float* __restrict__ thr, u64* __restrict__ keys,
    unsigned* __restrict__ cnt, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t m0, n0;
  g2_tile_coords(nrows, nq, group_m, m0, n0);       // 256 index rows x 128 queries per workgroup
  f32x16_t acc[2][2];
  gemm_mainloop2<T>(rows, d, queries, d, nrows, nq, d, m0, n0, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t q = n0 + wn * 64 + ni * 32 + (lane & 31);
    if (q >= nq) continue;
    const float th = thr[q];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int64_t mbase = m0 + wm * 64 + mi * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
        const float v = acc[mi][ni][r];
        if (m < nrows && v >= th) {
          const unsigned pos = atomicAdd(cnt + q, 1u);
          if (pos < SORT_CAP) keys[q * SORT_CAP + pos] = pack_key(v, row_base + (uint32_t)m);
        }
      }
    }
  }
}

// The scan for query batches wider than one 128 tile: 256 queries x 256 index rows per workgroup on
// the one-wave-per-SIMD main loop of gemm_core6.h.  The QUERIES are its A operand, so (operands
// swapped inside, see there) a lane owns one query and its registers walk index rows:
//   acc[mi][ni][r] = <query q0 + wm*128 + mi*32 + (lane&31),  row r0 + wn*128 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)>
// The filter first takes the maximum of each 16-register tile (8 v_max3) and only walks the tile
// when some lane of the wave has a survivor -- survivors are rare by construction of theta.
template <typename T>
__global__ __launch_bounds__(G6_THREADS) void sim_filter_kernel6(
    const T* __restrict__ rows, int64_t nrows, uint32_t row_base, const T* __restrict__ queries,
    int64_t nq, int64_t d, const float* __restrict__ thr, u64* __restrict__ keys,
    unsigned* __restrict__ cnt, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // `group_m` index-row tiles stay resident (L2) while every query tile sweeps over them
  const int64_t ntr = (nrows + G6_BN - 1) / G6_BN, ntq = (nq + G6_BM - 1) / G6_BM;
  int64_t tr_, tq_;
  gemm_tile_coords(ntr, ntq, group_m, tr_, tq_);
  const int64_t q0 = tq_ * G6_BM, r0 = tr_ * G6_BN;
  f32x16_t acc[4][4];
  {
    f32x16_t zero[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) zero[i][r] = 0.f;
    gemm_mainloop6<T>(queries, d, rows, d, nq, nrows, d, q0, r0, smem, acc, zero);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t q = q0 + wm * 128 + mi * 32 + (lane & 31);
    const float th = q < nq ? thr[q] : __builtin_inff();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const f32x16_t a = acc[mi][ni];
      float mx = fmaxf(fmaxf(a[0], a[1]), a[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, a[r]), a[r + 1]);
      mx = fmaxf(mx, a[15]);
      if (mx >= th) {
        const int64_t nbase = r0 + wn * 128 + ni * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t n = nbase + (r & 3) + 8 * (r >> 2);
          const float v = a[r];
          if (n < nrows && v >= th) {
            const unsigned pos = atomicAdd(cnt + q, 1u);
            if (pos < SORT_CAP) keys[q * SORT_CAP + pos] = pack_key(v, row_base + (uint32_t)n);
          }
        }
      }
    }
  }
}

// Survivors of the generation-7 scan.  A score passes its list's threshold about once in a thousand, but with 1024
// scores per 32 x 32 block most blocks hold one: appending from inside the block walk (returning atomic on the list
// counter -> wait -> store, per survivor, as generation 6 does) serialises ~7 L2 round trips per wave and tile
// (profiles/r02_scan_trace_v0.log: 11.2k cycles of a 44.9k-cycle tile).  Instead the walk only STAGES survivors in
// wave-private LDS -- position from the wave's ballot, no atomics.  Staging: keys in the wave's OWN DMA slices of units 2
// and 3 (only this wave ever writes them, and not before its next tile starts), 2048 of them; their query numbers in
// unit 4 behind the threshold tables.  A block adds at most 1024 records, so one capacity check per block suffices.
//
// Round 3 (the tile trace had 12k of a tile's 40k cycles outside the K loop, and the four waves meet at the next tile's
// first barrier, so the slowest filter sets the pace):
//  * two-level walk: the block test keeps the maxima of its four register quads; a block with a survivor tests the
//    quads and walks only the (one, almost always) quad that holds it -- ~8 ballot + branch steps instead of 16;
//  * the append is split around the NEXT tile's K loop: at the end of a tile's filter its (<= 64, one per lane)
//    records move from the staging area into registers; after the next K loop the list-counter atomics are issued,
//    their return is consumed half a filter later and the key stores leave half a filter before the tile start that
//    would otherwise wait for their acknowledgement (vmcnt retires in order).  Neither round trip is exposed.  More
//    than 64 records in a tile, or a full staging area, take the synchronous flush (rare);
//  * the tile walk advances incrementally (no division per tile), and the DMA lane offsets are computed once: the
//    query panel is padded with zero rows to a whole tile (query_prep_kernel), so no tile needs clamped rows.
#define SC7_STAGE_CAP 2048
#define SC7_QIDX_OFF (G7_TAB_OFF + 4096)
__device__ __forceinline__ char* sc7_key_slot(char* smem, int wave, unsigned pos) {
  return smem + (2 + (pos >> 10)) * G7_UNIT_BYTES + wave * 1024 + ((pos >> 7) & 7) * 4096 + (pos & 127) * 8;
}
__device__ __forceinline__ uint16_t* sc7_q_slot(char* smem, int wave, unsigned pos) {
  return (uint16_t*)(smem + SC7_QIDX_OFF + wave * (SC7_STAGE_CAP * 2) + pos * 2);
}
// synchronous append of staged records [from, wcount): one atomic round trip, then the stores
__device__ __forceinline__ void sc7_flush(char* smem, int wave, unsigned from, unsigned wcount, int lane, int64_t q0,
                                          u64* __restrict__ keys, unsigned* __restrict__ cnt) {
  for (unsigned i = from + (unsigned)lane; i < wcount; i += 64) {
    const u64 key = *(const u64*)sc7_key_slot(smem, wave, i);
    const int64_t q = q0 + *sc7_q_slot(smem, wave, i);
    const unsigned pos = atomicAdd(cnt + q, 1u);
    if (pos < SORT_CAP) keys[q * SORT_CAP + pos] = key;
  }
}
// query blocks MI0 .. MI1-1 of the wave's 128 x 128 scores; whole tiles only: every row of the tile exists
template <int MI0, int MI1>
__device__ __forceinline__ void sc7_filter(f32x16_t (&acc)[4][4], const float (&th)[4], uint32_t id0, uint32_t ql0,
                                           int lane, char* smem, int wave, unsigned& wcount, int64_t q0,
                                           u64* __restrict__ keys, unsigned* __restrict__ cnt) {
#pragma unroll
  for (int mi = MI0; mi < MI1; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      asm volatile("" : "+a"(acc[mi][ni]));            // stays in its AGPRs until this point
      const f32x16_t a = acc[mi][ni];
      float gm[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) gm[g] = fmaxf(fmaxf(fmaxf(a[4 * g], a[4 * g + 1]), a[4 * g + 2]), a[4 * g + 3]);
      const float mx = fmaxf(fmaxf(fmaxf(gm[0], gm[1]), gm[2]), gm[3]);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx >= th[mi]) != 0, 0)) {          // wave-uniform: some lane holds a survivor (cold: laid out behind the 16 block tests)
        if (wcount > SC7_STAGE_CAP - 1024) { sc7_flush(smem, wave, 0u, wcount, lane, q0, keys, cnt); wcount = 0; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (__builtin_amdgcn_ballot_w64(gm[g] >= th[mi]) == 0) continue;
#pragma unroll
          for (int r = 4 * g; r < 4 * g + 4; ++r) {
            const uint32_t off = (uint32_t)(ni * 32 + (r & 3) + 8 * (r >> 2));
            const bool pass = a[r] >= th[mi];
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (m != 0) {
              if (pass) {
                const unsigned pos = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                *(u64*)sc7_key_slot(smem, wave, pos) = pack_key(a[r], id0 + off);
                *sc7_q_slot(smem, wave, pos) = (uint16_t)(ql0 + mi * 32);
              }
              wcount += (unsigned)__builtin_popcountll(m);
            }
          }
        }
      }
      G7_FENCE_();
    }
  }
}

// Work id -> (row tile, query tile) of the generation-7 scan.  With the GEMM's walk (g7_tile: groups of row tiles swept by
// ALL query tiles, 32 consecutive tiles per XCD and round) every XCD touches 4 query panels and 8 row panels per
// round and hops to another group next round: nothing is reused across rounds, and the L2 -> fabric traffic of one
// 6980-query search measured 441 GB, 32x the 13.6 GB index (FETCH_SIZE, profiles/r02_hbm_traffic_search.json) -- 4.7 TB/s,
// close to what the memory system delivers at all.  Here every XCD OWNS a group of at most 8 query tiles (<= 3 MiB of
// query panels, resident in its 4 MiB L2 for the whole launch) and a share of the row tiles, and walks its rows with
// the query tile running fastest: a row panel is fetched once per query GROUP instead of once per 4 query tiles.
// The walk is an iterator: work item w = it * nslots + slot of the XCD's (row, query tile) grid, advanced by nslots per
// tile with one compare instead of a division.
struct Sc7Walk {
  bool owned;                      // XCD-owned query groups (else the GEMM's walk, by index)
  int it;
  int64_t ntr, ntq; int group_m;
  uint32_t q_lo, qn, r_lo, rn, row, qi, d_row, d_qi;
  __device__ __forceinline__ bool init(int64_t ntr_, int64_t ntq_, int group_m_, int qgroup, int64_t& r0, int64_t& q0) {
    ntr = ntr_; ntq = ntq_; group_m = group_m_; it = 0;
    owned = (gridDim.x & 7) == 0 && ntq <= 8 * qgroup;
    if (!owned) return g7_tile(0, ntr, ntq, group_m, r0, q0);
    const uint32_t x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const uint32_t tq = (uint32_t)ntq, tr = (uint32_t)ntr;
    uint32_t ng = 1;                                        // query groups: 1, 2, 4 or 8 -- at most 8 query tiles each
    while (ng < 8 && (tq + ng - 1) / ng > (uint32_t)qgroup) ng <<= 1;
    const uint32_t g = x % ng, part = x / ng, nparts = 8 / ng;
    q_lo = g * tq / ng; qn = (g + 1) * tq / ng - q_lo;
    r_lo = (uint32_t)((uint64_t)part * tr / nparts); rn = (uint32_t)((uint64_t)(part + 1) * tr / nparts) - r_lo;
    if (qn == 0) return false;
    row = slot / qn; qi = slot - row * qn;
    d_row = nslots / qn; d_qi = nslots - d_row * qn;
    if (row >= rn) return false;
    r0 = (int64_t)(r_lo + row) * 256; q0 = (int64_t)(q_lo + qi) * 256;
    return true;
  }
  __device__ __forceinline__ bool next(int64_t& r0, int64_t& q0) {
    ++it;
    if (!owned) return g7_tile(it, ntr, ntq, group_m, r0, q0);
    qi += d_qi; row += d_row;
    if (qi >= qn) { qi -= qn; ++row; }
    if (row >= rn) return false;
    r0 = (int64_t)(r_lo + row) * 256; q0 = (int64_t)(q_lo + qi) * 256;
    return true;
  }
};

// The same scan on generation 7 (gemm_core7.h / gemm_wide7.h): 128-byte K steps (whole-line LDS-DMA requests) in a
// PERSISTENT kernel -- one workgroup per CU walks (query tile, row tile) pairs, the first K step of the next pair and its
// 256 thresholds are fetched while the current pair is filtered.  The only wait at a tile start is vmcnt(16): K step 1
// may be outstanding, everything older -- the prefetch, and the appends of the tile before, issued half a filter
// earlier -- has landed.  16-bit inputs only (the f32 scan stays on generation 6).
// `thr` must be readable up to round_up(nq, 256) + 256 entries (+inf beyond nq: carve_search / init_lists_kernel);
// `queries` holds round_up(nq, 256) rows, zeros beyond nq (query_prep_kernel).
template <typename T>
__global__ __launch_bounds__(G6_THREADS) void sim_filter_kernel7(
    const T* __restrict__ rows, int64_t nrows, uint32_t row_base, const T* __restrict__ queries,
    int64_t nq, int64_t d, const float* __restrict__ thr, u64* __restrict__ keys,
    unsigned* __restrict__ cnt, int group_m, unsigned long long* __restrict__ trace) {
  typedef typename MmaOps<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntr = (nrows + 255) / 256, ntq = (nq + 255) / 256;
  const int nk = (int)((d * 2) / G7_ROW_BYTES);
  unsigned wcount = 0;                     // records in this wave's staging area (wave-uniform)
  unsigned pend_n = 0;                     // records of the previous tile held in registers, one per lane (wave-uniform)
  u64 pend_key = 0; uint32_t pend_q = 0;
  int64_t r0, q0;
  const int qgroup = group_m >> 8;                 // (the host packs both walk parameters into one argument)
  group_m &= 255;
  Sc7Walk walk;
  if (!walk.init(ntr, ntq, group_m, qgroup, r0, q0)) return;
  G7Src src;
  g7_offsets<T>(src, d, d, wave, lane0);           // every row of every tile exists (padded query panel, whole row tiles)
  src.a = (const char*)(queries + q0 * d);
  src.b = (const char*)(rows + r0 * d);
  g7_dma((const char*)(thr + q0 + wm * 128), lane0 * 16, g7_lds_addr(smem + G7_TAB_OFF + wave * 1024));
  g7_fill(src.a, src.oa, smem, wave);
  g7_fill(src.b, src.ob, smem + G7_UNIT_BYTES, wave);
  for (;;) {
    unsigned long long* tr = nullptr;          // debug: phase stamps of the first 8192 tiles (om_debug_gemm_trace)
    if (trace) {
      const int64_t tile_id = (r0 / 256) * ntq + q0 / 256;
      if (tile_id < 8192) tr = trace + tile_id * 32;
    }
    if (tr && threadIdx.x == 0) { tr[0] = clock64(); tr[30] = wall_clock64(); }
    if (nk > 1) {
      g7_fill(src.a + G7_ROW_BYTES, src.oa, smem + 2 * G7_UNIT_BYTES, wave);
      g7_fill(src.b + G7_ROW_BYTES, src.ob, smem + 3 * G7_UNIT_BYTES, wave);
      G7_WAIT_VM(16);
    } else {
      G7_WAIT_VM(0);
    }
    if (tr && threadIdx.x == 0) tr[1] = clock64();
    float th[4];
    f32x16_t acc[4][4];
    {
      int lane_i = lane0;
      asm volatile("" : "+v"(lane_i));
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) th[mi] = *(const float*)(smem + G7_TAB_OFF + wave * 1024 + (mi * 32 + (lane_i & 31)) * 4);
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
      const frag_t zf = __builtin_bit_cast(frag_t, z4);
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q) {       // zero by the matrix core, one MFMA per accumulator tile: with ONE shared zero fragment the
        frag_t zq = zf;                    // compiler issues one product and copies it into the other fifteen tiles with 240
        asm volatile("" : "+v"(zq));       // v_accvgpr_mov (~1 k of the 1.5 k cycles this block took per pair)
        acc[q >> 2][q & 3] = zero;
        MmaOps<T>::mma(zq, zq, acc[q >> 2][q & 3]);
      }
    }
    gemm_mainloop7_run<T>(src, nk, smem, acc, tr, false);
    if (tr && threadIdx.x == 0) tr[15] = clock64();
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    // the previous tile's records: list positions requested now, consumed half a filter later
    unsigned pend_pos = 0;
    if (pend_n && (unsigned)lane < pend_n) pend_pos = atomicAdd(cnt + pend_q, 1u);
    // next pair: its first K step and its thresholds are fetched under the filter below
    const int64_t rc = r0, qc = q0;
    const bool has_next = walk.next(r0, q0);       // (r0, q0 unchanged when there is none: a harmless re-fetch)
    src.a = (const char*)(queries + q0 * d);
    src.b = (const char*)(rows + r0 * d);
    g7_dma((const char*)(thr + q0 + wm * 128), lane0 * 16, g7_lds_addr(smem + G7_TAB_OFF + wave * 1024));
    g7_fill(src.a, src.oa, smem, wave);
    g7_fill(src.b, src.ob, smem + G7_UNIT_BYTES, wave);
    G7_FENCE_();
    {
      const uint32_t id0 = row_base + (uint32_t)rc + (uint32_t)(wn * 128 + 4 * (lane >> 5));    // row id of (ni = 0, r = 0)
      const uint32_t ql0 = (uint32_t)(wm * 128 + (lane & 31));
      sc7_filter<0, 2>(acc, th, id0, ql0, lane, smem, wave, wcount, qc, keys, cnt);
      if (pend_n) {
        if ((unsigned)lane < pend_n && pend_pos < SORT_CAP) keys[(int64_t)pend_q * SORT_CAP + pend_pos] = pend_key;
        pend_n = 0;
      }
      G7_FENCE_();
      sc7_filter<2, 4>(acc, th, id0, ql0, lane, smem, wave, wcount, qc, keys, cnt);
      if (wcount > 64) sc7_flush(smem, wave, 64u, wcount, lane, qc, keys, cnt);
      pend_n = wcount < 64u ? wcount : 64u;
      if ((unsigned)lane < pend_n) {
        pend_key = *(const u64*)sc7_key_slot(smem, wave, (unsigned)lane);
        pend_q = (uint32_t)qc + *sc7_q_slot(smem, wave, (unsigned)lane);
      }
      wcount = 0;
    }
    if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; tr[31] = wall_clock64(); }
    if (!has_next) break;
  }
  if (pend_n && (unsigned)lane0 < pend_n) {
    const unsigned pos = atomicAdd(cnt + pend_q, 1u);
    if (pos < SORT_CAP) keys[(int64_t)pend_q * SORT_CAP + pos] = pend_key;
  }
  G7_WAIT_VM(0);
}

// ---- the same scan on the CONTINUOUS ring (round 4; gemm_core7.h gemm_mainloop7_cont, gemm_wide7.h kernels 7c / 7r) ------
// What the kernel above pays per tile beside its 12 k-cycle filter: 16 wave-uniform branches in every K step (one around each
// DMA issue: ~800 of a step's ~3200 cycles -- the branch-free loop runs 2400), 16 DMA issues of K step 1 at every tile start
// and a cold first step.  Here the K loop of a pair prefetches step 0 of the next pair (its last step issues nothing else:
// TAIL_EMPTY), A(1) of the next pair is issued right behind the K loop and lands under the filter, the first half of B(1)
// behind the filter.  The filter's staging lives in the units the last step frees -- this wave's own slices of them, so no
// barrier brackets the filter:
//     ring.bn (the unit A(nk - 1) left)   slices 0-7   keys 0 .. 1023
//     ring.sp (the unit B(nk - 1) left)   slices 0-3   keys 1024 .. 1535;  4-6 query numbers;  7 the next pair's thresholds
//     ring.an (the spare of the last step)              A(1) of the next pair
#define SC7C_CAP 1536
struct Sc7cStage { char* bn; char* sp; };
__device__ __forceinline__ char* sc7c_key_slot(const Sc7cStage& st, unsigned pos) {
  const unsigned sl = pos >> 7;
  return (sl < 8 ? st.bn + sl * 4096 : st.sp + (sl - 8) * 4096) + (pos & 127) * 8;
}
__device__ __forceinline__ uint16_t* sc7c_q_slot(const Sc7cStage& st, unsigned pos) { return (uint16_t*)(st.sp + (4 + (pos >> 9)) * 4096 + (pos & 511) * 2); }
__device__ __forceinline__ void sc7c_flush(const Sc7cStage& st, unsigned from, unsigned wcount, int lane, int64_t q0, u64* __restrict__ keys,
                                           unsigned* __restrict__ cnt) {
  for (unsigned i = from + (unsigned)lane; i < wcount; i += 64) {
    const u64 key = *(const u64*)sc7c_key_slot(st, i);
    const int64_t q = q0 + *sc7c_q_slot(st, i);
    const unsigned pos = atomicAdd(cnt + q, 1u);
    if (pos < SORT_CAP) keys[q * SORT_CAP + pos] = key;
  }
}
__device__ __forceinline__ float sc7c_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float sc7c_max16(const f32x16_t& a) {      // eight v_max3 (the compiler's fmaxf tree: 13 v_max + 5 v_max3)
  const float t0 = sc7c_max3(a[0], a[1], a[2]), t1 = sc7c_max3(a[3], a[4], a[5]), t2 = sc7c_max3(a[6], a[7], a[8]);
  const float t3 = sc7c_max3(a[9], a[10], a[11]), t4 = sc7c_max3(a[12], a[13], a[14]);
  return sc7c_max3(sc7c_max3(t0, t1, t2), sc7c_max3(t3, t4, a[15]), t0);
}
// One row block of queries (mi) against the wave's 128 index rows: the common case -- no score of the block's 64 per lane reaches
// its query's threshold -- is 64 accumulator reads, 33 v_max3, ONE compare and ONE branch; the survivor path is laid out as cold
// code behind the straight line.  Tile trace at the benchmark's size (profiles/r04_probe16_scan_filter_phases.log): filter 5.4 k
// cycles per tile with one test per column block and the survivor code inline (each test 5 KB from the next), 4.7 k like this;
// 3.4 survivors per wave and tile cost ~850 cycles each, the 16 block tests ~1.8 k.  A COMPACT survivor path (scores parked in LDS,
// run-time loop over the score indices: 70 KB of code down to 12) measured 8.9 k -- the unrolled per-score code is the faster one,
// the instruction cache is not what it waits for.
template <int MI0, int MI1>
__device__ __forceinline__ void sc7c_filter(f32x16_t (&acc)[4][4], const float (&th)[4], uint32_t id0, uint32_t ql0, int lane, const Sc7cStage& st,
                                            unsigned& wcount, int64_t q0, u64* __restrict__ keys, unsigned* __restrict__ cnt) {
#pragma unroll
  for (int mi = MI0; mi < MI1; ++mi) {
    f32x16_t a[4];
    float bm[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      asm volatile("" : "+a"(acc[mi][ni]));            // stays in its AGPRs until this point
      a[ni] = acc[mi][ni];
      bm[ni] = sc7c_max16(a[ni]);
    }
    const float mx = sc7c_max3(sc7c_max3(bm[0], bm[1], bm[2]), bm[3], bm[3]);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx >= th[mi]) != 0, 0)) {          // wave-uniform: some lane holds a survivor
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        if (__builtin_amdgcn_ballot_w64(bm[ni] >= th[mi]) == 0) continue;
        if (wcount > SC7C_CAP - 1024) { sc7c_flush(st, 0u, wcount, lane, q0, keys, cnt); wcount = 0; }      // a block adds at most 1024
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float gm = fmaxf(fmaxf(a[ni][4 * g], a[ni][4 * g + 1]), fmaxf(a[ni][4 * g + 2], a[ni][4 * g + 3]));
          if (__builtin_amdgcn_ballot_w64(gm >= th[mi]) == 0) continue;
#pragma unroll
          for (int r = 4 * g; r < 4 * g + 4; ++r) {
            const uint32_t off = (uint32_t)(ni * 32 + (r & 3) + 8 * (r >> 2));
            const bool pass = a[ni][r] >= th[mi];
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (m != 0) {
              if (pass) {
                const unsigned pos = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                *(u64*)sc7c_key_slot(st, pos) = pack_key(a[ni][r], id0 + off);
                *sc7c_q_slot(st, pos) = (uint16_t)(ql0 + mi * 32);
              }
              wcount += (unsigned)__builtin_popcountll(m);
            }
          }
        }
      }
    }
    G7_FENCE_();
  }
}

template <typename T>
__global__ __launch_bounds__(G6_THREADS) void sim_filter_kernel7c(
    const T* __restrict__ rows, int64_t nrows, uint32_t row_base, const T* __restrict__ queries,
    int64_t nq, int64_t d, const float* __restrict__ thr, u64* __restrict__ keys,
    unsigned* __restrict__ cnt, int group_m, unsigned long long* __restrict__ trace) {
  typedef typename MmaOps<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntr = (nrows + 255) / 256, ntq = (nq + 255) / 256;
  const int nk = (int)((d * 2) / G7_ROW_BYTES);
  unsigned wcount = 0;                     // records in this wave's staging area (wave-uniform)
  unsigned pend_n = 0;                     // records of the previous tile held in registers, one per lane (wave-uniform)
  u64 pend_key = 0; uint32_t pend_q = 0;
  int64_t r0, q0;
  const int qgroup = group_m >> 8;
  group_m &= 255;
  Sc7Walk walk;
  if (!walk.init(ntr, ntq, group_m, qgroup, r0, q0)) return;
  G7SrcU src;
  g7_offsets_u<T>(src, d, d, wave, lane0);         // every row of every tile exists (padded query panel, whole row tiles)
  G7Ring ring;
  g7_ring_reset(ring);
  const uint32_t lds_base = g7_lds_addr(smem);
  const char* cur_a = (const char*)(queries + q0 * d);
  const char* cur_b = (const char*)(rows + r0 * d);
  g7_dma((const char*)(thr + q0 + wm * 128), lane0 * 16, lds_base + ring.sp + (7 * 4 + wave) * 1024);
  g7_fill_a(src, cur_a, smem + ring.ac, wave);
  g7_fill_b(src, cur_b, smem + ring.bc, wave);
  g7_fill_a(src, cur_a + G7_ROW_BYTES, smem + ring.an, wave);
#pragma unroll
  for (int i = 0; i < 4; ++i) g7_issue_b(src, cur_b + G7_ROW_BYTES, i, lds_base + ring.bn + (i * 4 + wave) * 1024);
  G7_WAIT_VM(12);                                  // the thresholds and K step 0 (A(1) and half of B(1) may be outstanding)
  __builtin_amdgcn_s_barrier();                    // first pair only: K step 0 published
  int64_t r1 = r0, q1 = q0;
  bool has_next = walk.next(r1, q1);
  for (;;) {
    unsigned long long* tr = nullptr;
    if (trace) {
      const int64_t tile_id = (r0 / 256) * ntq + q0 / 256;
      if (tile_id < 8192) tr = trace + tile_id * 32;
    }
    if (tr && threadIdx.x == 0) { tr[0] = tr[1] = clock64(); tr[30] = wall_clock64(); }
    float th[4];
    f32x16_t acc[4][4];          // (starts from zero inside the K loop: ZERO_FIRST -- no initialising pass)
    {
      int lane_i = lane0;
      asm volatile("" : "+v"(lane_i));
      const char* const tab = smem + ring.sp + (7 * 4 + wave) * 1024;      // this pair's thresholds (fetched under the previous filter)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) th[mi] = *(const float*)(tab + (mi * 32 + (lane_i & 31)) * 4);
    }
    const char* const next_a = (const char*)(queries + q1 * d);      // (no next pair: this one again -- a harmless prefetch)
    const char* const next_b = (const char*)(rows + r1 * d);
    gemm_mainloop7_cont<T, true, G7NoTail, true, true>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc, tr);
    if (tr && threadIdx.x == 0) tr[15] = clock64();
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const Sc7cStage st = {smem + ring.bn + wave * 1024, smem + ring.sp + wave * 1024};
    // the previous tile's records: list positions requested now, consumed half a filter later
    unsigned pend_pos = 0;
    if (pend_n && (unsigned)lane < pend_n) pend_pos = atomicAdd(cnt + pend_q, 1u);
    // under the filter: A(1) of the next pair (into the unit the filter does not use) and its thresholds
    g7_fill_a(src, next_a + G7_ROW_BYTES, smem + ring.an, wave);
    g7_dma((const char*)(thr + q1 + wm * 128), lane0 * 16, g7_lds_addr(st.sp) + 7 * 4096);
    G7_FENCE_();
    if (tr && threadIdx.x == 0) tr[16] = clock64();
    {
      const uint32_t id0 = row_base + (uint32_t)r0 + (uint32_t)(wn * 128 + 4 * (lane >> 5));    // row id of (ni = 0, r = 0)
      const uint32_t ql0 = (uint32_t)(wm * 128 + (lane & 31));
      sc7c_filter<0, 2>(acc, th, id0, ql0, lane, st, wcount, q0, keys, cnt);
      if (pend_n) {
        if ((unsigned)lane < pend_n && pend_pos < SORT_CAP) keys[(int64_t)pend_q * SORT_CAP + pend_pos] = pend_key;
        pend_n = 0;
      }
      G7_FENCE_();
      if (tr && threadIdx.x == 0) tr[17] = clock64();
      sc7c_filter<2, 4>(acc, th, id0, ql0, lane, st, wcount, q0, keys, cnt);
      if (tr && threadIdx.x == 0) { tr[18] = clock64(); tr[20] = wcount; }
      if (wcount > 64) sc7c_flush(st, 64u, wcount, lane, q0, keys, cnt);
      pend_n = wcount < 64u ? wcount : 64u;
      if ((unsigned)lane < pend_n) {
        pend_key = *(const u64*)sc7c_key_slot(st, (unsigned)lane);
        pend_q = (uint32_t)q0 + *sc7c_q_slot(st, (unsigned)lane);
      }
      wcount = 0;
    }
    // A(1) and the thresholds were fetched a whole filter ago; everything issued since (the key stores of the pending records,
    // half a filter old; a synchronous flush, rare) is older than any DMA of the next K loop.  The staging area has been read
    // (lgkmcnt): the first half of B(1) may overwrite it.
    if (tr && threadIdx.x == 0) tr[19] = clock64();
    G7_WAIT_VM(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tr && threadIdx.x == 0) tr[21] = clock64();
#pragma unroll
    for (int i = 0; i < 4; ++i) g7_issue_b(src, next_b + G7_ROW_BYTES, i, lds_base + ring.bn + (i * 4 + wave) * 1024);
    if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; tr[31] = wall_clock64(); }
    if (!has_next) break;
    cur_a = next_a; cur_b = next_b; r0 = r1; q0 = q1;
    has_next = walk.next(r1, q1);
  }
  if (pend_n && (unsigned)lane0 < pend_n) {
    const unsigned pos = atomicAdd(cnt + pend_q, 1u);
    if (pos < SORT_CAP) keys[(int64_t)pend_q * SORT_CAP + pos] = pend_key;
  }
  G7_WAIT_VM(0);
}

// ---- the continuous-ring scan on 16 x 16 x 32 MFMAs (round 6) -------------------------------------------------------------------
// The K loop of the encoder's kernels 7c16 / 7r16 (gemm_core7.h gemm_mainloop7_cont16: 28.3 k cycles per K = 768 tile where the
// 32 x 32 x 16 loop above takes ~30 k) under the same filter.  Accumulator layout: acc[ti][fj][r] = score(query q0 + wm*128 + ti*16 +
// (lane & 15), row r0 + wn*128 + fj*16 + 4*(lane >> 4) + r): a lane holds EIGHT queries' thresholds and 32 scores per query block.
// Ring protocol of that loop: it enters with A(1) AND all of B(1) issued (the 32 x 32 loop issues the second half of B(1) itself), so
// all eight B(1) requests go out behind the filter; its last step issues nothing (empty tail: vmcnt(0) at the last barrier) and the
// tile starts from zero inside the loop (ZERO_FIRST).  Same staging, same survivor records, same lists as kernel 7c.
// Measured (profiles/r06_scan_16x16x32_ab.txt): in shader cycles the tile is LONGER (K loop 29.3 k against 26.8 k, filter 6.3 k against
// 4.7 k: eight query blocks per lane instead of four) and the clock higher (the small MFMA shape draws less); end to end +1.4 % on one
// box, -0.8 % on another.  A wash: the 32 x 32 x 16 kernel stays the default, this one is OM_GEMM_CONT bit 10.
// One query block (ti: 16 queries) against the wave's 128 index rows: 32 scores per lane, 11 v_max3, one compare and one branch when
// none reaches its query's threshold; the survivor path behind it as cold code.  (Tried: ONE branch per four blocks with the survivor
// path re-reading the accumulators -- 7.7 k cycles per tile against 6.3 k: with 3.4 survivors per wave and tile most groups of four
// blocks hold one, and then pay the reads and the maxima twice.)
template <int TI0, int TI1>
__device__ __forceinline__ void sc7c16_filter(f32x4_t (&acc)[8][8], const float (&th)[8], uint32_t id0, uint32_t ql0, int lane, const Sc7cStage& st,
                                              unsigned& wcount, int64_t q0, u64* __restrict__ keys, unsigned* __restrict__ cnt) {
#pragma unroll
  for (int ti = TI0; ti < TI1; ++ti) {
    f32x4_t a[8];
    float bm[8];
#pragma unroll
    for (int fj = 0; fj < 8; ++fj) {
      asm volatile("" : "+a"(acc[ti][fj]));            // stays in its AGPRs until this point
      a[fj] = acc[ti][fj];
      bm[fj] = sc7c_max3(sc7c_max3(a[fj][0], a[fj][1], a[fj][2]), a[fj][3], a[fj][3]);
    }
    const float mx = sc7c_max3(sc7c_max3(bm[0], bm[1], bm[2]), sc7c_max3(bm[3], bm[4], bm[5]), sc7c_max3(bm[6], bm[7], bm[7]));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx >= th[ti]) != 0, 0)) {          // wave-uniform: some lane holds a survivor
#pragma unroll
      for (int fj = 0; fj < 8; ++fj) {
        if (__builtin_amdgcn_ballot_w64(bm[fj] >= th[ti]) == 0) continue;
        if (wcount > SC7C_CAP - 256) { sc7c_flush(st, 0u, wcount, lane, q0, keys, cnt); wcount = 0; }      // a block adds at most 256
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool pass = a[fj][r] >= th[ti];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
          if (m != 0) {
            if (pass) {
              const unsigned pos = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
              *(u64*)sc7c_key_slot(st, pos) = pack_key(a[fj][r], id0 + (uint32_t)(fj * 16 + r));
              *sc7c_q_slot(st, pos) = (uint16_t)(ql0 + ti * 16);
            }
            wcount += (unsigned)__builtin_popcountll(m);
          }
        }
      }
    }
    G7_FENCE_();
  }
}

template <typename T>
__global__ __launch_bounds__(G6_THREADS) void sim_filter_kernel7c16(
    const T* __restrict__ rows, int64_t nrows, uint32_t row_base, const T* __restrict__ queries,
    int64_t nq, int64_t d, const float* __restrict__ thr, u64* __restrict__ keys,
    unsigned* __restrict__ cnt, int group_m, unsigned long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t ntr = (nrows + 255) / 256, ntq = (nq + 255) / 256;
  const int nk = (int)((d * 2) / G7_ROW_BYTES);
  unsigned wcount = 0;                     // records in this wave's staging area (wave-uniform)
  unsigned pend_n = 0;                     // records of the previous tile held in registers, one per lane (wave-uniform)
  u64 pend_key = 0; uint32_t pend_q = 0;
  int64_t r0, q0;
  const int qgroup = group_m >> 8;
  group_m &= 255;
  Sc7Walk walk;
  if (!walk.init(ntr, ntq, group_m, qgroup, r0, q0)) return;
  G7SrcU src;
  g7_offsets_u<T>(src, d, d, wave, lane0);         // every row of every tile exists (padded query panel, whole row tiles)
  G7Ring ring;
  g7_ring_reset(ring);
  const uint32_t lds_base = g7_lds_addr(smem);
  const char* cur_a = (const char*)(queries + q0 * d);
  const char* cur_b = (const char*)(rows + r0 * d);
  g7_dma((const char*)(thr + q0 + wm * 128), lane0 * 16, lds_base + ring.sp + (7 * 4 + wave) * 1024);
  g7_fill_a(src, cur_a, smem + ring.ac, wave);
  g7_fill_b(src, cur_b, smem + ring.bc, wave);
  g7_fill_a(src, cur_a + G7_ROW_BYTES, smem + ring.an, wave);
  g7_fill_b(src, cur_b + G7_ROW_BYTES, smem + ring.bn, wave);
  G7_WAIT_VM(16);                                  // the thresholds and K step 0 (A(1), B(1) may be outstanding)
  __builtin_amdgcn_s_barrier();                    // first pair only: K step 0 published
  int64_t r1 = r0, q1 = q0;
  bool has_next = walk.next(r1, q1);
  for (;;) {
    unsigned long long* tr = nullptr;
    if (trace) {
      const int64_t tile_id = (r0 / 256) * ntq + q0 / 256;
      if (tile_id < 8192) tr = trace + tile_id * 32;
    }
    if (tr && threadIdx.x == 0) { tr[0] = tr[1] = clock64(); tr[30] = wall_clock64(); }
    float th[8];
    f32x4_t acc[8][8];           // (starts from zero inside the K loop: ZERO_FIRST -- no initialising pass)
    {
      int lane_i = lane0;
      asm volatile("" : "+v"(lane_i));
      const char* const tab = smem + ring.sp + (7 * 4 + wave) * 1024;      // this pair's thresholds (fetched under the previous filter)
#pragma unroll
      for (int ti = 0; ti < 8; ++ti) th[ti] = *(const float*)(tab + (ti * 16 + (lane_i & 15)) * 4);
    }
    const char* const next_a = (const char*)(queries + q1 * d);      // (no next pair: this one again -- a harmless prefetch)
    const char* const next_b = (const char*)(rows + r1 * d);
    gemm_mainloop7_cont16<T, true, G7NoTail, true>(src, cur_a, cur_b, next_a, next_b, nk, smem, ring, acc, tr);
    if (tr && threadIdx.x == 0) tr[15] = clock64();
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const Sc7cStage st = {smem + ring.bn + wave * 1024, smem + ring.sp + wave * 1024};
    // the previous tile's records: list positions requested now, consumed half a filter later
    unsigned pend_pos = 0;
    if (pend_n && (unsigned)lane < pend_n) pend_pos = atomicAdd(cnt + pend_q, 1u);
    // under the filter: A(1) of the next pair (into the unit the filter does not use) and its thresholds
    g7_fill_a(src, next_a + G7_ROW_BYTES, smem + ring.an, wave);
    g7_dma((const char*)(thr + q1 + wm * 128), lane0 * 16, g7_lds_addr(st.sp) + 7 * 4096);
    G7_FENCE_();
    if (tr && threadIdx.x == 0) tr[16] = clock64();
    {
      const uint32_t id0 = row_base + (uint32_t)r0 + (uint32_t)(wn * 128 + 4 * (lane >> 4));    // row id of (fj = 0, r = 0)
      const uint32_t ql0 = (uint32_t)(wm * 128 + (lane & 15));
      sc7c16_filter<0, 4>(acc, th, id0, ql0, lane, st, wcount, q0, keys, cnt);
      if (pend_n) {
        if ((unsigned)lane < pend_n && pend_pos < SORT_CAP) keys[(int64_t)pend_q * SORT_CAP + pend_pos] = pend_key;
        pend_n = 0;
      }
      G7_FENCE_();
      if (tr && threadIdx.x == 0) tr[17] = clock64();
      sc7c16_filter<4, 8>(acc, th, id0, ql0, lane, st, wcount, q0, keys, cnt);
      if (tr && threadIdx.x == 0) { tr[18] = clock64(); tr[20] = wcount; }
      if (wcount > 64) sc7c_flush(st, 64u, wcount, lane, q0, keys, cnt);
      pend_n = wcount < 64u ? wcount : 64u;
      if ((unsigned)lane < pend_n) {
        pend_key = *(const u64*)sc7c_key_slot(st, (unsigned)lane);
        pend_q = (uint32_t)q0 + *sc7c_q_slot(st, (unsigned)lane);
      }
      wcount = 0;
    }
    // A(1) and the thresholds were fetched a whole filter ago; everything issued since (the key stores of the pending records, half a
    // filter old; a synchronous flush, rare) is older than any DMA of the next K loop.  The staging area has been read (lgkmcnt): B(1)
    // may overwrite it -- all eight requests (the 16 x 16 x 32 loop enters with the whole unit issued).
    if (tr && threadIdx.x == 0) tr[19] = clock64();
    G7_WAIT_VM(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tr && threadIdx.x == 0) tr[21] = clock64();
    g7_fill_b(src, next_b + G7_ROW_BYTES, smem + ring.bn, wave);
    if (tr && threadIdx.x == 0) { tr[28] = clock64(); tr[29] = blockIdx.x; tr[31] = wall_clock64(); }
    if (!has_next) break;
    cur_a = next_a; cur_b = next_b; r0 = r1; q0 = q1;
    has_next = walk.next(r1, q1);
  }
  if (pend_n && (unsigned)lane0 < pend_n) {
    const unsigned pos = atomicAdd(cnt + pend_q, 1u);
    if (pos < SORT_CAP) keys[(int64_t)pend_q * SORT_CAP + pos] = pend_key;
  }
  G7_WAIT_VM(0);
}

// Small query batches (<= 128 queries): the scan is a pass over the whole f16 index (13.6 GB at 8.8 M x 768) that has to
// run at HBM speed.  The generic 128-query tile spends 1.7 PFLOP of matrix-core time on padding at Q = 1 (3.99 ms per
// search in round 1, profiles/r01_search_shapes.jsonl).  Round 2 kept one or two 32-query blocks resident in LDS beside a
// three- / four-slot ring (2.56 ms at Q = 1, 3.35 at Q = 64, nothing for 65-128 queries: 4.6 ms on the generation-2 filter
// kernel) -- LDS shared between the resident blocks (48 KiB each) and the ring had no room for a third block.  Round 3 moves
// the queries into registers.  A wave's B operand never changes: its 32-query block is 12 K steps x 4 MFMA sub-steps x 16 bytes per lane = 192
// VGPRs of the 512 this one-wave-per-SIMD kernel owns, loaded once from global memory in fragment order.  The whole LDS
// (160 KiB) is the ring: ten slots of 128 rows x 128 bytes, nine units = 144 KiB per CU in flight.
//     NB = 1 (Q <= 32)    the four waves share the block and take 32 rows of a unit each
//     NB = 2 (Q <= 64)    waves 0, 2 / 1, 3 hold block 0 / 1 and take 64 rows each
//     NB = 4 (Q <= 128)   one block per wave, every wave takes all 128 rows (each reads the whole unit: 64 KiB of fragment
//                         reads + 16 KiB of DMA per unit is the LDS port's limit at ~15 TB/s, above what HBM delivers)
// Persistent over 128-row tiles (blockIdx, blockIdx + grid, ...); whole tiles only; 256 <= d <= 768 (K steps past d / 64 are
// compiled but skipped: the K loop is unrolled so that the query fragments are indexed statically).  The index stream is
// fetched with the non-temporal policy (read once, by one CU).  Measured (8 841 823 x 768, k = 1000, profiles/r03_search_small_batches.log):
// Q = 1 2.42-2.62 ms (round 2: 2.56-2.65 on like boxes), Q = 64 3.14-3.22 (3.35), Q = 128 3.59-3.68 (4.63).  What more
// bytes in flight did NOT buy: the kernel's own time still grows with the resident blocks (2.15 / 2.9 / 3.3 ms for one / two /
// four, rocprofv3) although neither the matrix core (16 MFMAs per 16 KiB unit at most) nor the LDS port is near a limit.
#define SR_RING 10
#define SR_UROWS 128
#define SR_UBYTES (SR_UROWS * G7_ROW_BYTES)
#define SR_LDS (SR_RING * SR_UBYTES)
#define SR_IPU (SR_UROWS / 32)                  // DMA instructions per wave and unit
#define SR_NKMAX 12
// DBG (timing probes only, OM_OPT_SEARCH_DEBUG bits 3 / 4; the results are wrong): 1 = every second MFMA sub-step skipped (does
// the pass's time follow its matrix-core work?), 2 = no MFMA at all
template <typename T, int NB, int DBG = 0>
__global__ __launch_bounds__(G6_THREADS) void sim_stream_reg_kernel(
    const T* __restrict__ rows, int64_t nrows, uint32_t row_base, const T* __restrict__ queries, int64_t nq, int64_t d,
    const float* __restrict__ thr, u64* __restrict__ keys, unsigned* __restrict__ cnt, int nt) {
  typedef typename MmaOps<T>::frag_t frag_t;
  constexpr int RBW = NB;                        // 32-row blocks per wave and unit: 4 / NB waves share a query block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = (int)((d * 2) / G7_ROW_BYTES);
  const int64_t ntiles = nrows / SR_UROWS;
  const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  if (my_tiles <= 0) return;
  const int units = my_tiles * nk;
  const uint32_t lds0 = g7_lds_addr(smem);
  const int half = lane >> 5, l31 = lane & 31, key = (l31 >> 1) & 7;
  const int nb = wave % NB, rgroup = wave / NB;   // this wave's query block and its share of a unit's rows

  // the wave's query block, in MFMA fragment order, straight from global memory (once per launch)
  frag_t bq[SR_NKMAX][4];
  {
    const int qr = nb * 32 + l31;
    const int rr = qr < nq ? qr : (int)nq - 1;                  // padding rows repeat the last query (their threshold is +inf)
    const T* qrow = queries + (int64_t)rr * d + half * 8;
#pragma unroll
    for (int ks = 0; ks < SR_NKMAX; ++ks)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        bq[ks][kk] = ks < nk ? *(const frag_t*)(qrow + ks * 64 + kk * 16) : __builtin_bit_cast(frag_t, z4);
      }
  }
  const float th = thr[nb * 32 + l31];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // before the first hand-counted DMA: nothing of the compiler's is in flight
#pragma unroll
  for (int ks = 0; ks < SR_NKMAX; ++ks)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+a"(bq[ks][kk]));    // in the accumulation half of the register file (the matrix core
                                                                            // reads B from there as well): 192 + 16 NB of its 256; the other half stays free for the row fragments

  // DMA offsets of a unit: instruction i of this wave moves rows (i*4 + wave)*8 .. +7, lane -> row (lane >> 3),
  // physical chunk (lane & 7) <- source chunk (lane & 7) ^ ((row >> 1) & 7)
  uint32_t off[SR_IPU];
#pragma unroll
  for (int i = 0; i < SR_IPU; ++i) {
    const int r = (i * 4 + wave) * 8 + (lane >> 3);
    off[i] = (uint32_t)(r * d * 2) + ((((lane & 7) ^ ((r >> 1) & 7))) << 4);
  }
  // issue cursor: (source address, K step, slot) of the next unit to fetch -- advanced incrementally (the per-unit scalar
  // work counts: at two and four resident blocks a unit's instruction path is as long as its share of the HBM stream)
  const char* i_base = (const char*)(rows + (int64_t)blockIdx.x * SR_UROWS * d);
  const int64_t i_tile_step = (int64_t)gridDim.x * SR_UROWS * d * 2 - (int64_t)nk * G7_ROW_BYTES;   // last K step of a tile -> first of the next
  int i_ks = 0, i_slot = 0, issued = 0;
  auto issue = [&]() {
    const char* base = i_base;
    const uint32_t dst = lds0 + i_slot * SR_UBYTES + wave * 1024;
    if (nt) {                                                    // (wave-uniform) the index is read once per pass, by one CU
#pragma unroll
      for (int i = 0; i < SR_IPU; ++i) g7_dma_nt(base, off[i], dst + i * 4096);
    } else {
#pragma unroll
      for (int i = 0; i < SR_IPU; ++i) g7_dma(base, off[i], dst + i * 4096);
    }
    i_base += G7_ROW_BYTES;
    if (++i_ks == nk) { i_ks = 0; i_base += i_tile_step; }
    if (++i_slot == SR_RING) i_slot = 0;
    ++issued;
  };
  for (int u = 0; u < SR_RING - 1; ++u)
    if (u < units) issue();

  const int arow = (rgroup * (32 * RBW) + l31) * G7_ROW_BYTES;   // + rt * 32 rows
  f32x16_t acc[RBW];
#pragma unroll
  for (int rt = 0; rt < RBW; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  // Appends are split around the next tile.  A 32 x 32 score block holds a survivor about once per tile and query block, and
  // the list position comes from a returning atomic: ~1.1 us under a streaming load (MI355X_MICROARCH.md "dequeue"), during
  // which this wave issues nothing -- written in line (returning atomic, then the stores) that was RBW round trips per
  // 8-12 us tile, and the compiler's wait for the result is vmcnt(0): the whole ring.  Instead the tile's scores move to a
  // shadow register set, the atomics of all its blocks leave together (inline assembly: invisible to the compiler's
  // counting), and four units into the NEXT tile -- 16 younger DMA instructions in the in-order queue -- the positions are
  // there and the keys are stored.
  f32x16_t sh[RBW];
  unsigned pn[RBW], ppos[RBW];
#pragma unroll
  for (int rt = 0; rt < RBW; ++rt) { pn[rt] = 0; ppos[rt] = 0; }
  bool pend = false;                                             // (wave-uniform) a tile's appends are in flight
  int64_t ptile = 0;
  int since = 0;                                                 // units issued since its atomics
  const int qi = nb * 32 + l31;
  auto retire = [&]() {
    // every atomic has returned once at most the younger operations are outstanding
    if (since >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rt = 0; rt < RBW; ++rt) asm volatile("" : "+v"(ppos[rt]));      // (the results are read after the wait)
#pragma unroll
    for (int rt = 0; rt < RBW; ++rt) {
      if (pn[rt]) {
        const uint32_t id0 = row_base + (uint32_t)(ptile * SR_UROWS) + rgroup * (32 * RBW) + rt * 32 + 4 * half;
        unsigned pos = ppos[rt];
        u64* const kq = keys + (int64_t)qi * SORT_CAP;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (sh[rt][r] >= th) {
            if (pos < SORT_CAP) kq[pos] = pack_key(sh[rt][r], id0 + (r & 3) + 8 * (r >> 2));
            ++pos;
          }
      }
    }
    pend = false;
  };
  int c_slot = 0, u = 0;
  int64_t tile = blockIdx.x;
  // Round 4, what the pass's time is made of (timing probes with half / none of the MFMAs compiled out, DBG above;
  // profiles/r04_probe9_small_batch_mfma_share.log): with HALF the matrix-core work the pass runs at the no-MFMA time (2.16-2.40 ms
  // whatever the batch: the HBM stream), with all of it +0.2 ms (Q = 1) to +1.0 ms (Q = 64, 128) -- a threshold, not a slope,
  // and not the clock round 3 suspected.  Reading the first fragments of unit u + 1 under the last MFMAs of unit u (the
  // barrier of u certifying u + 1 as well, eight units in flight instead of nine) changed nothing (Q = 64 3.20, Q = 128 3.32 ms:
  // profiles/r04_probe10_small_batch_cross_unit_prefetch.log) and was removed: the chain that is too long for a unit's share of
  // the stream is the four dependent read -> MFMA stages themselves.
  frag_t a[2][RBW];
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int ks = 0; ks < SR_NKMAX; ++ks) {
      if (ks < nk) {                                             // wave-uniform
        // unit u landed; up to RING - 1 younger units may be in flight
        const int younger = issued - 1 - u;
        // steady state: the eight units behind unit u are in flight (the ninth is issued behind the barrier below).  While
        // the ring drains -- the last units of a launch -- everything outstanding is waited for: a ladder of exact counts
        // compiled to ~40 scalar instructions and a dozen taken branches per unit, a third of what a unit may cost at all
        if (younger == SR_RING - 2) G7_WAIT_VM((SR_RING - 2) * SR_IPU);
        else G7_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();                            // ... for every wave; and unit u - 1 has been read by all
        if (issued < units) { issue(); ++since; }                // into the slot of unit u - 1
        const char* ua = smem + c_slot * SR_UBYTES + arow;
        // the row fragments of sub-step kk + 1 are read under the MFMAs of sub-step kk (order pinned: left alone the
        // compiler reads two fragments, waits, multiplies, reads two ... and every LDS round trip is exposed)
#pragma unroll
        for (int rt = 0; rt < RBW; ++rt) a[0][rt] = *(const frag_t*)(ua + rt * 32 * G7_ROW_BYTES + ((half ^ key) << 4));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          G7_FENCE_();
          if (kk < 3) {
            const int slot = ((((kk + 1) << 1) | half) ^ key) << 4;
#pragma unroll
            for (int rt = 0; rt < RBW; ++rt) a[(kk + 1) & 1][rt] = *(const frag_t*)(ua + rt * 32 * G7_ROW_BYTES + slot);
          }
          G7_FENCE_();
#pragma unroll
          for (int rt = 0; rt < RBW; ++rt)
            if (DBG == 0 || (DBG == 1 && !(kk & 1))) MmaOps<T>::mma(a[kk & 1][rt], bq[ks][kk], acc[rt]); // acc[rt][r]: row 8(r>>2) + 4 half + (r&3) of the 32-row block, query 32 nb + l31
        }
        G7_FENCE_();
        // the accumulators' home is the AGPR file: without the pin the compiler keeps them in VGPRs across the (wave-uniform)
        // K-step branches and copies all of them in and out around every unit's MFMAs -- 32 RBW v_accvgpr moves per unit that
        // also wait for the matrix core to drain (rocprofv3: 88 VALU instructions per unit, the pass 35 / 55 % longer with two /
        // four resident blocks than with one)
#pragma unroll
        for (int rt = 0; rt < RBW; ++rt) asm volatile("" : "+a"(acc[rt]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (++c_slot == SR_RING) c_slot = 0;
        ++u;
        if (ks == 3 && pend) retire();                           // the previous tile's appends: positions are back by now
      }
#pragma unroll
      for (int rt = 0; rt < RBW; ++rt) asm volatile("" : "+a"(acc[rt]));   // (also where the skipped and the taken K step meet)
    }
    // (the dispatcher sends d < 256 -- fewer than four K steps, no retire point inside a tile -- to the generic kernels)
    bool any = false;
#pragma unroll
    for (int rt = 0; rt < RBW; ++rt) {
      sh[rt] = acc[rt];
      unsigned n = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) n += sh[rt][r] >= th ? 1u : 0u;
      pn[rt] = n;
      // one atomic per lane and block, not per survivor (with one query every append of a round lands on one counter)
      if (n) {
        unsigned* const cp = cnt + qi;
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(ppos[rt]) : "v"(cp), "v"(n) : "memory");
      }
      any = any || __builtin_amdgcn_ballot_w64(n != 0) != 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    }
    if (any) { pend = true; ptile = tile; since = 0; }
    tile += gridDim.x;
  }
  if (pend) { since = 0; retire(); }
  G7_WAIT_VM(0);
}

// append a dense score block S[q, 0:n] (rows row_base..) to every list
__global__ void append_dense_kernel(const float* __restrict__ S, int64_t ldS, int n,
                                    uint32_t row_base, u64* __restrict__ keys,
                                    unsigned* __restrict__ cnt) {
  const int64_t q = blockIdx.x;
  const unsigned base = cnt[q];
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    if (base + j < SORT_CAP) keys[q * SORT_CAP + base + j] = pack_key(S[q * ldS + j], row_base + j);
  __syncthreads();
  if (threadIdx.x == 0) cnt[q] = base + n;
}

// flag[0] |= any list overflowed
__global__ void check_overflow_kernel(const unsigned* __restrict__ cnt, int64_t nq,
                                      unsigned* __restrict__ flag) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq && cnt[q] > SORT_CAP) atomicOr(flag, 1u);
}
__global__ void restore_cnt_kernel(unsigned* __restrict__ cnt, const unsigned* __restrict__ prev,
                                   int64_t nq) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) cnt[q] = prev[q];
}

// descending bitonic sort of P (power of two) keys in LDS
__device__ inline void bitonic_sort_desc(u64* s, int P, int tid, int nthr) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += nthr) {
        const int x = i ^ j;
        if (x > i) {
          const u64 a = s[i], b = s[x];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { s[i] = b; s[x] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// Sort one list, keep the best k (exact mode) or everything within margin[q] of the k-th
// best (certified f16 mode), publish the new threshold.  flag[1] = max list length,
// flag[2] |= a certified list outgrew LIST_MAX.
__global__ __launch_bounds__(SORT_THREADS) void select_kernel(
    u64* __restrict__ keys, unsigned* __restrict__ cnt, unsigned* __restrict__ cnt_prev,
    float* __restrict__ thr, const float* __restrict__ margin, int k, unsigned* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* s = (u64*)smem;
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x;
  unsigned n = cnt[q];
  if (n > SORT_CAP) n = SORT_CAP;
  int P = 2;
  while (P < (int)n) P <<= 1;
  u64* list = keys + q * SORT_CAP;
  for (int i = tid; i < P; i += SORT_THREADS) s[i] = i < (int)n ? list[i] : 0ull;
  __syncthreads();
  bitonic_sort_desc(s, P, tid, SORT_THREADS);

  unsigned& keep_s = *(unsigned*)(smem + SORT_CAP * 8);  // all LDS in the one dynamic array
  if (tid == 0) keep_s = 0;
  __syncthreads();
  float th = -INFINITY;
  unsigned keep;
  if (!margin) {
    keep = n < (unsigned)k ? n : (unsigned)k;
    if (n >= (unsigned)k) th = key_score(s[k - 1]);
  } else {
    if (n >= (unsigned)k) {
      th = key_score(s[k - 1]) - 2.0f * margin[q];
      unsigned local = 0;
      for (int i = tid; i < (int)n; i += SORT_THREADS) local += key_score(s[i]) >= th ? 1u : 0u;
      atomicAdd(&keep_s, local);
      __syncthreads();
      keep = keep_s;
    } else {
      keep = n;
    }
  }
  for (int i = tid; i < (int)keep; i += SORT_THREADS) list[i] = s[i];
  if (tid == 0) {
    cnt[q] = keep;
    cnt_prev[q] = keep;
    thr[q] = th;
    atomicMax(flag + 1, keep);
    if (keep > LIST_MAX) atomicOr(flag + 2, 1u);
  }
}

// The per-round selection WITHOUT a sort: a round only needs the new threshold (the k-th best score so far) and the
// list cut down to what can still matter; order is irrelevant until the very end.  Radix select over the 32 score bits
// of the keys (4 passes of a 256-bin histogram in LDS, suffix sums by wave shuffles), then an unordered compaction of
// everything >= the cut (exact mode: the k-th score, ties included; certified mode: k-th - 2 margin).  ~30x cheaper
// than the 8192-key bitonic sort it replaces between rounds (profiles/r01_bench_v8_kernel_stats.csv: 8 x 2.3 ms).
__global__ __launch_bounds__(256) void select_radix_kernel(
    u64* __restrict__ keys, unsigned* __restrict__ cnt, float* __restrict__ thr, const float* __restrict__ margin,
    int k, unsigned* __restrict__ flag, int cap) {
  // cap: keys the LDS copy holds (SORT_CAP, or less when the host expects short lists: 64 KiB of LDS per workgroup
  // leaves two workgroups per CU, 32 KiB four -- this kernel is a chain of barriers and wants the occupancy)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* s = (u64*)smem;
  unsigned* hist = (unsigned*)(smem + (size_t)cap * 8);       // 256 bins
  unsigned* misc = hist + 256;                                // [0] bucket, [1] remaining, [2] keep counter, [4..7] wave totals
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned n = cnt[q];
  if (n > SORT_CAP) {                                        // appends beyond the capacity were dropped: the round is void
    if (tid == 0) atomicOr(flag, 1u);                        // (sticky; the step-by-step loop uses check_overflow_kernel)
    n = SORT_CAP;
  }
  u64* list = keys + q * SORT_CAP;
  // a list longer than the LDS copy (rare: the host sizes `cap` with a 2x margin) is read from global memory in every
  // pass (64 KiB, L2-resident) and only the survivors are staged
  const bool staged = n <= (unsigned)cap;
  const u64* src = staged ? (const u64*)s : (const u64*)list;
  if (staged)
    for (unsigned i = tid; i < n; i += 256) s[i] = list[i];
  if (tid == 0) misc[2] = 0;
  __syncthreads();
  if (n <= (unsigned)k) {                                     // nothing to cut: keep everything
    float mn = INFINITY;
    for (unsigned i = tid; i < n; i += 256) mn = fminf(mn, key_score(src[i]));
    mn = -wave_max(-mn);
    float* wmin = (float*)(misc + 4);
    if (lane == 0) wmin[wave] = mn;
    __syncthreads();
    if (tid == 0) {
      float th = -INFINITY;                                   // fewer than k candidates so far: no threshold yet
      if (n == (unsigned)k) {                                 // exactly k: the k-th best is the minimum
        const float m4 = fminf(fminf(wmin[0], wmin[1]), fminf(wmin[2], wmin[3]));
        th = margin ? m4 - 2.0f * margin[q] : m4;
      }
      thr[q] = th;
      atomicMax(flag + 1, n);
    }
    return;
  }
  uint32_t prefix = 0, mask = 0;
  unsigned remaining = (unsigned)k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    // (round 6 probe: one LDS atomic per DISTINCT bin of a wave for the two high bytes -- where nearly every lane names the same bin -- ran
    // 170 us per launch against 152: the same-address atomics are not what this kernel waits for; it is 3 % of a search)
    for (unsigned i = tid; i < n; i += 256) {
      const uint32_t v = (uint32_t)(src[i] >> 32);
      if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    // suffix sums S[t] = sum_{b >= t} hist[b]: inside each wave by shuffles, across waves through misc[4..7]
    const unsigned c = hist[tid];
    unsigned suf = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_down(suf, o, 64);
      if (lane + o < 64) suf += up;
    }
    if (lane == 0) misc[4 + wave] = suf;                      // total of this wave's 64 bins
    __syncthreads();
    unsigned above = 0;                                       // bins of higher waves
    for (int w = wave + 1; w < 4; ++w) above += misc[4 + w];
    const unsigned S = suf + above, Snext = S - c;            // S[t] and S[t+1]
    if (S >= remaining && Snext < remaining) { misc[0] = (unsigned)tid; misc[1] = remaining - Snext; }
    __syncthreads();
    prefix |= misc[0] << shift;
    mask |= 255u << shift;
    remaining = misc[1];
    __syncthreads();
  }
  const float kth = orderable_f32(prefix);
  const float cut = margin ? kth - 2.0f * margin[q] : kth;
  if (staged) {
    for (unsigned i = tid; i < n; i += 256) {
      const u64 key = s[i];
      if (key_score(key) >= cut) list[atomicAdd(&misc[2], 1u)] = key;
    }
    __syncthreads();
  } else {                                                    // in place is not safe without the copy: survivors via LDS
    for (unsigned i = tid; i < n; i += 256) {
      const u64 key = list[i];
      if (key_score(key) >= cut) {
        const unsigned pos = atomicAdd(&misc[2], 1u);
        if (pos < (unsigned)cap) s[pos] = key;
      }
    }
    __syncthreads();
    unsigned kept = misc[2];                                  // one value for the whole workgroup (no writer until the barrier below)
    if (kept > (unsigned)cap) {                               // masses of ties (duplicated rows): treated as an overflow
      if (tid == 0) atomicOr(flag, 1u);
      kept = (unsigned)cap;
    }
    for (unsigned i = tid; i < kept; i += 256) list[i] = s[i];
    __syncthreads();
    if (tid == 0) misc[2] = kept;                             // read back by this same thread below
  }
  if (tid == 0) {
    const unsigned keep = misc[2];
    cnt[q] = keep;
    thr[q] = cut;
    atomicMax(flag + 1, keep);
    if (margin && keep > LIST_MAX) atomicOr(flag + 2, 1u);
  }
}

// per-query certified margin and bf16 copy of the queries
__global__ __launch_bounds__(256) void query_prep_kernel(const float* __restrict__ q,
                                                         f16_t* __restrict__ qb,
                                                         float* __restrict__ margin,
                                                         const float* __restrict__ stats,
                                                         int64_t nq, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nq) {            // padding of the query panel up to a whole 256-query tile: zero rows (the generation-7 scan reads them)
    if (row < (nq + 255) / 256 * 256)
      for (int c = lane; c < d; c += 64) qb[row * d + c] = (f16_t)0.f;
    return;
  }
  float n2 = 0.f, e2 = 0.f, b2 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = q[row * d + c];
    const f16_t r = (f16_t)v;
    const float rv = (float)r;
    qb[row * d + c] = r;
    n2 += v * v; e2 += (v - rv) * (v - rv); b2 += rv * rv;
  }
  n2 = wave_sum(n2); e2 = wave_sum(e2); b2 = wave_sum(b2);
  if (lane == 0) {
    const float qn = sqrtf(n2), qe = sqrtf(e2), qbn = sqrtf(b2);
    const float Ep = stats[0], Pn = stats[1];
    // f32 accumulation of the 16-bit scan and of the f32 re-score: each <= d * 2^-24 * |q||p|
    const float slop = 3.0f * (float)d * 5.9604645e-8f * fmaxf(qn, qbn) * (Pn + Ep);
    margin[row] = 1.01f * (qn * Ep + qe * Pn) + slop;
  }
}

// exact f32 re-score of every surviving candidate: one wavefront per (query, slot)
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ q,
                                                      const float* __restrict__ index,
                                                      u64* __restrict__ keys,
                                                      const unsigned* __restrict__ cnt, int d) {
  const int64_t qi = blockIdx.y;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slot >= (int)cnt[qi]) return;
  u64* kp = keys + qi * SORT_CAP + slot;
  const uint32_t row = key_payload(*kp);
  const float* qv = q + qi * d;
  const float* pv = index + (int64_t)row * d;
  float acc = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 a = *(const float4*)(qv + c);
    const float4 b = *(const float4*)(pv + c);
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) *kp = pack_key(acc, row);
}

__global__ void emit_kernel(const u64* __restrict__ keys, const unsigned* __restrict__ cnt, int k,
                            int64_t id_offset, float* __restrict__ out_scores,
                            int64_t* __restrict__ out_ids) {
  const int64_t q = blockIdx.x;
  const unsigned n = cnt[q];
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    if (j < (int)n) {
      const u64 key = keys[q * SORT_CAP + j];
      out_scores[q * k + j] = key_score(key);
      out_ids[q * k + j] = id_offset + (int64_t)key_payload(key);
    } else {
      out_scores[q * k + j] = -3.4028235e38f;  // faiss pads (D,I) with (-FLT_MAX, -1)
      out_ids[q * k + j] = -1;
    }
  }
}

__global__ void init_lists_kernel(unsigned* cnt, unsigned* cnt_prev, float* thr, int64_t nq, unsigned* flag) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < 16) flag[q] = 0;                                            // the 64-byte flag block (one launch less than a memset)
  if (q < nq) { cnt[q] = 0; cnt_prev[q] = 0; thr[q] = -INFINITY; }
  else if (q < (nq + 255) / 256 * 256 + 256) thr[q] = INFINITY;       // queries that do not exist never have survivors
}

// ---- K14: merge W sorted partial lists per query ------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void merge_kernel(
    const float* __restrict__ ps, const int64_t* __restrict__ pi, int W, int64_t nq, int k_in,
    int k_out, float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* s = (u64*)smem;
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x;
  const int n = W * k_in;
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += SORT_THREADS) {
    u64 key = 0ull;
    if (i < n) {
      const int w = i / k_in, j = i % k_in;
      const int64_t src = ((int64_t)w * nq + q) * k_in + j;
      if (pi[src] >= 0) key = pack_key(ps[src], (uint32_t)i);  // ties: (part, position) order
    }
    s[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc(s, P, tid, SORT_THREADS);
  for (int j = tid; j < k_out; j += SORT_THREADS) {
    const u64 key = j < P ? s[j] : 0ull;
    if (key != 0ull) {
      const int i = (int)key_payload(key);
      const int w = i / k_in, jj = i % k_in;
      const int64_t src = ((int64_t)w * nq + q) * k_in + jj;
      out_scores[q * k_out + j] = ps[src];
      out_ids[q * k_out + j] = pi[src];
    } else {
      out_scores[q * k_out + j] = -3.4028235e38f;
      out_ids[q * k_out + j] = -1;
    }
  }
}

extern "C" int om_topk_merge(const float* part_scores, const int64_t* part_ids, int W,
                             int64_t n_queries, int k_in, int k_out, float* out_scores,
                             int64_t* out_ids, void* stream) {
  if (n_queries <= 0 || k_out <= 0) return 0;
  if (W <= 0 || k_in <= 0) OM_FAIL("W and k_in must be positive");
  if ((int64_t)W * k_in > SORT_CAP) OM_FAIL("W*k_in exceeds the 8192-key merge capacity");
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               SORT_CAP * 8));
    attr_set = true;
  }
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)n_queries), dim3(SORT_THREADS), SORT_CAP * 8,
                     (hipStream_t)stream, part_scores, part_ids, W, n_queries, k_in, k_out,
                     out_scores, out_ids);
  OM_LAUNCH_CHECK();
  return 0;
}

// ---- host orchestration -------------------------------------------------------------------
struct SearchWs {
  u64* keys; unsigned *cnt, *cnt_prev, *flag; float *thr, *margin, *dense; f16_t* qb;
  size_t total;
};
static SearchWs carve_search(int64_t nq, int d, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base + o; };
  SearchWs w;
  w.keys = (u64*)take((size_t)nq * SORT_CAP * 8);
  w.cnt = (unsigned*)take((size_t)nq * 4);
  w.cnt_prev = (unsigned*)take((size_t)nq * 4);
  w.flag = (unsigned*)take(64);
  w.thr = (float*)take(((size_t)(nq + 255) / 256 * 256 + 256) * 4);      // +inf beyond nq: the generation-7 scan reads whole tiles of thresholds
  w.margin = (float*)take((size_t)nq * 4);
  w.dense = (float*)take((size_t)nq * DENSE_CHUNK * 4);
  w.qb = (f16_t*)take((size_t)((nq + 255) / 256 * 256) * d * 2);            // whole query tiles, zero rows beyond nq
  w.total = off;
  return w;
}

extern "C" void om_sim_topk_info(int64_t out[8]) {
  for (int i = 0; i < 8; ++i) out[i] = g_info[i];
}

extern "C" size_t om_sim_topk_workspace_bytes(int64_t n_queries, int d, int k) {
  (void)k;
  if (n_queries <= 0 || d <= 0) return 0;
  return carve_search(n_queries, d, nullptr).total;
}

namespace {
struct Scan {
  int mode; const float* q32; const float* idx32; const f16_t* idx16; int64_t nq, N; int d, k;
  SearchWs ws; hipStream_t s;

  int select(bool certified) {
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)nq), dim3(SORT_THREADS), SORT_CAP * 8 + 16, s,
                       ws.keys, ws.cnt, ws.cnt_prev, ws.thr, certified ? ws.margin : nullptr, k,
                       ws.flag);
    OM_LAUNCH_CHECK();
    return 0;
  }
  int select_radix(bool certified, int cap = SORT_CAP) {
    hipLaunchKernelGGL(select_radix_kernel, dim3((unsigned)nq), dim3(256), (size_t)cap * 8 + 1024 + 64, s,
                       ws.keys, ws.cnt, ws.thr, certified ? ws.margin : nullptr, k, ws.flag, cap);
    OM_LAUNCH_CHECK();
    return 0;
  }
  int read_flags(unsigned (&f)[4]) {
    static thread_local unsigned* pinned = nullptr;        // page-locked: the copy is one DMA, no staging through the runtime
    if (!pinned) OM_HIP(hipHostMalloc((void**)&pinned, 64, hipHostMallocPortable));
    OM_HIP(hipMemcpyAsync(pinned, ws.flag, sizeof(f), hipMemcpyDeviceToHost, s));
    OM_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 4; ++i) f[i] = pinned[i];
    return 0;
  }
  int rescore(unsigned maxlist) {
    hipLaunchKernelGGL(rescore_kernel, dim3((maxlist + 3) / 4, (unsigned)nq), dim3(256), 0, s, q32,
                       idx32, ws.keys, ws.cnt, d);
    OM_LAUNCH_CHECK();
    return 0;
  }
  // dense-score rows [r0, r0+n) and merge them into the lists (always exact, HBM heavy)
  int dense_gemm_append(int64_t r0, int n, bool bf16) {
    if (bf16) {
      if (om_gemm_nt(OM_F16, ws.qb, d, idx16 + r0 * d, d, OM_F32, ws.dense, DENSE_CHUNK, nq, n, d,
                     nullptr, nullptr, 0, OM_ACT_NONE, s)) return 1;
    } else {
      if (om_gemm_nt(OM_F32, q32, d, idx32 + r0 * d, d, OM_F32, ws.dense, DENSE_CHUNK, nq, n, d,
                     nullptr, nullptr, 0, OM_ACT_NONE, s)) return 1;
    }
    hipLaunchKernelGGL(append_dense_kernel, dim3((unsigned)nq), dim3(256), 0, s, ws.dense,
                       (int64_t)DENSE_CHUNK, n, (uint32_t)r0, ws.keys, ws.cnt);
    OM_LAUNCH_CHECK();
    return 0;
  }
  int dense_step(int64_t r0, int n, bool bf16) {
    if (dense_gemm_append(r0, n, bf16)) return 1;
    return select(bf16);
  }
  int filter_step(int64_t r0, int64_t n, bool bf16, bool check = true /* false: the selection that follows flags overflows */) {
    const bool wide = nq > 128;
    const int64_t ntm = (n + 255) / 256, ntn = wide ? (nq + G6_BM - 1) / G6_BM : (nq + G2_BN - 1) / G2_BN;
    if (ntm * ntn > 0x7fffffffLL) OM_FAIL("scan grid too large");
    const bool timing = om_timing_on();
    if (timing) om_timing_begin(OM_TIMING_SCAN, s);
    const dim3 grid((unsigned)(ntm * ntn));
#define SCAN(KERNEL, THREADS, LDS, TT, ROWS, QUERIES)                                              \
  hipLaunchKernelGGL((KERNEL<TT>), grid, dim3(THREADS), LDS, s, ROWS + r0 * d, n, (uint32_t)r0, QUERIES, \
                     nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, 8)
    if (bf16) {
      if (wide && om_option(OM_OPT_SCAN_GEN7)) {
        // generation 7 takes the whole 256-row tiles, generation 6 the ragged tail of the chunk (if any)
        const int64_t whole = n & ~(int64_t)255;
        if (whole) {
          int ncu = g7_num_cus();
          const int64_t tiles = (whole / 256) * ntn;
          if (tiles < ncu) ncu = (int)tiles;
          if ((d * 2) / G7_ROW_BYTES >= 3 && (om_option(OM_OPT_GEMM_CONT) & 4) && (om_option(OM_OPT_GEMM_CONT) & 1024))      // bit 10 (round 6): ... on 16 x 16 x 32 MFMAs
            hipLaunchKernelGGL((sim_filter_kernel7c16<f16_t>), dim3((unsigned)ncu), dim3(G6_THREADS), G7_LDS_BYTES, s, idx16 + r0 * d, whole,
                               (uint32_t)r0, ws.qb, nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, 8 | (std::max(1, om_option(OM_OPT_SCAN_QGROUP)) << 8), omk_debug_trace());
          else if ((d * 2) / G7_ROW_BYTES >= 3 && (om_option(OM_OPT_GEMM_CONT) & 4))      // bit 2: the scan on the continuous ring
            hipLaunchKernelGGL((sim_filter_kernel7c<f16_t>), dim3((unsigned)ncu), dim3(G6_THREADS), G7_LDS_BYTES, s, idx16 + r0 * d, whole,
                               (uint32_t)r0, ws.qb, nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, 8 | (std::max(1, om_option(OM_OPT_SCAN_QGROUP)) << 8), omk_debug_trace());
          else
          hipLaunchKernelGGL((sim_filter_kernel7<f16_t>), dim3((unsigned)ncu), dim3(G6_THREADS), G7_LDS_BYTES, s, idx16 + r0 * d, whole,
                             (uint32_t)r0, ws.qb, nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, 8 | (std::max(1, om_option(OM_OPT_SCAN_QGROUP)) << 8), omk_debug_trace());
        }
        if (n > whole)
          hipLaunchKernelGGL((sim_filter_kernel6<f16_t>), dim3((unsigned)ntn), dim3(G6_THREADS), G6_LDS_BYTES, s, idx16 + (r0 + whole) * d,
                             n - whole, (uint32_t)(r0 + whole), ws.qb, nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, 8);
      } else if (wide) SCAN(sim_filter_kernel6, G6_THREADS, G6_LDS_BYTES, f16_t, idx16, ws.qb);
      else if (nq <= 128 && (d * 2) / G7_ROW_BYTES <= SR_NKMAX && (d * 2) / G7_ROW_BYTES >= 4 && om_option(OM_OPT_SCAN_GEN7)) {
        // the HBM-speed pass for small batches (queries in registers, the whole LDS a ring) over the whole 128-row tiles;
        // the generic kernel takes the ragged tail
        const int64_t whole = n & ~(int64_t)127;
        if (whole) {
          int ncu = g7_num_cus();
          if (whole / SR_UROWS < ncu) ncu = (int)(whole / SR_UROWS);
#define STREAM_(NB_, DBG_) hipLaunchKernelGGL((sim_stream_reg_kernel<f16_t, NB_, DBG_>), dim3((unsigned)ncu), dim3(G6_THREADS), SR_LDS, s, idx16 + r0 * d, whole, \
                                       (uint32_t)r0, ws.qb, nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, (om_option(OM_OPT_SEARCH_DEBUG) & 4) ? 0 : 1)
#ifdef OM_PROBE_KERNELS      // timing variants with MFMAs compiled out (WRONG results): probe builds only (python -m openmatch_amd._build --probe)
#define STREAM(NB_) do { const int dbg_ = (om_option(OM_OPT_SEARCH_DEBUG) >> 3) & 3; if (dbg_ == 1) STREAM_(NB_, 1); else if (dbg_ == 2) STREAM_(NB_, 2); else STREAM_(NB_, 0); } while (0)
#else
#define STREAM(NB_) STREAM_(NB_, 0)
#endif
          if (nq <= 32) STREAM(1); else if (nq <= 64) STREAM(2); else STREAM(4);
#undef STREAM
#undef STREAM_
        }
        if (n > whole)
          hipLaunchKernelGGL((sim_filter_kernel<f16_t>), dim3((unsigned)((nq + G2_BN - 1) / G2_BN)), dim3(G2_THREADS), G2_LDS_BYTES, s, idx16 + (r0 + whole) * d, n - whole,
                             (uint32_t)(r0 + whole), ws.qb, nq, (int64_t)d, ws.thr, ws.keys, ws.cnt, 8);
      } else SCAN(sim_filter_kernel, G2_THREADS, G2_LDS_BYTES, f16_t, idx16, ws.qb);
    } else {
      if (wide) SCAN(sim_filter_kernel6, G6_THREADS, G6_LDS_BYTES, float, idx32, q32);
      else SCAN(sim_filter_kernel, G2_THREADS, G2_LDS_BYTES, float, idx32, q32);
    }
#undef SCAN
    if (timing) om_timing_end(OM_TIMING_SCAN, s, 2.0 * (double)n * (double)nq * (double)d);
    OM_LAUNCH_CHECK();
    if (!check) return 0;
    hipLaunchKernelGGL(check_overflow_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s,
                       ws.cnt, nq, ws.flag);
    OM_LAUNCH_CHECK();
    return 0;
  }

  void trace(const char* what, int64_t done, int64_t chunk, const unsigned (&f)[4]) {
    g_info[1]++;                                  // rounds
    if (f[1] > (unsigned)g_info[3]) g_info[3] = f[1];
    const bool dbg = (om_option(OM_OPT_SEARCH_DEBUG) & 1) != 0;
    if (dbg) fprintf(stderr, "[om_sim_topk] %-8s done=%ld chunk=%ld overflow=%u max_list=%u too_wide=%u\n",
                     what, (long)done, (long)chunk, f[0], f[1], f[2]);
  }

  // returns 0 ok, 1 error, 2 certified margin too wide (caller retries in f32)
  int run(bool bf16) {
    hipLaunchKernelGGL(init_lists_kernel, dim3((unsigned)((nq + 255) / 256 + 2)), dim3(256), 0, s,
                       ws.cnt, ws.cnt_prev, ws.thr, nq, ws.flag);
    OM_LAUNCH_CHECK();
    if (bf16) {
      hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)(((nq + 255) / 256 * 256 + 3) / 4)), dim3(256), 0, s, q32,
                         ws.qb, ws.margin, stats, nq, d);
      OM_LAUNCH_CHECK();
    }
    int64_t done = 0;
    unsigned f[4] = {0, 0, 0, 0};
    bool unsorted = false;
    // Fast schedule (default): bootstrap on a dense-scored chunk, then filtered scan -> overflow check (sticky flag) ->
    // radix selection per round, back to back with NO host synchronisation at all -- the list length after a selection
    // hardly moves (k plus ties / the certified margin), so the chunk sizes are fixed on the host from an ESTIMATE of
    // it.  One read of the flags at the very end; an overflow or a too-wide margin anywhere (adversarial row order,
    // heavily duplicated rows) falls back to the step-by-step loop below from scratch, which is always exact.
    bool careful = true;
    if (om_option(OM_OPT_SCAN_GEN7) && N > DENSE_CHUNK) {
      if (dense_gemm_append(0, DENSE_CHUNK, bf16)) return 1;
      if (select_radix(bf16, DENSE_CHUNK <= 4096 ? 4096 : SORT_CAP)) return 1;
      g_info[1]++;
      const int64_t list = bf16 ? (int64_t)k * 13 / 10 + 64 : (int64_t)k + 16;       // estimate (certified: + the margin's ties)
      // few queries: the scan is one HBM pass whatever the thresholds, the per-round launches are what costs -- few, long
      // rounds; many queries: an append costs the scan a slow path, the selection ~0.3 ms -- many short rounds with
      // tight thresholds (profiles/r02_scan_trace_*.log).  Expected survivors of a chunk ~ list * chunk / at.
      // (33-128 queries, one index pass per round set on the register-resident stream kernel: 100 % measured best of 60 / 100 / 200)
      const double growth = nq <= 32 ? 400.0 : (nq <= 128 && !om_option_is_set(OM_OPT_SCAN_GROWTH) ? 100.0 : (double)std::max(5, om_option(OM_OPT_SCAN_GROWTH)));
      const double room = 0.75 * (double)(SORT_CAP - list);
      const double want = std::min(room, (double)list * growth / 100.0);
      // selection launches of the rounds: lists are ~list + want keys long; when twice that margin fits 4096 keys the
      // LDS copy is sized for 4096 (four workgroups per CU instead of two); a longer list is read from global memory
      const int sel_cap = (double)list + 2.0 * want + 256.0 < 4096.0 ? 4096 : SORT_CAP;
      int64_t at = DENSE_CHUNK;
      while (at < N) {
        int64_t chunk = (int64_t)((double)at * want / (double)list);
        chunk = std::max<int64_t>(chunk, DENSE_CHUNK) & ~(int64_t)255;      // whole tiles (DENSE_CHUNK is one)
        chunk = std::min<int64_t>(chunk, N - at);
        if (N - at - chunk < chunk / 4) chunk = N - at;           // no sliver of a last round
        if (filter_step(at, chunk, bf16, false)) return 1;
        if (select_radix(bf16, sel_cap)) return 1;
        g_info[1]++;
        at += chunk;
      }
      // Few queries: the exact re-score and the final sort are queued BEHIND the last round before the flags are read, so
      // the host's wake-up overlaps them instead of idling the device (a 34 us hole per search at one query).  They
      // cover lists up to twice the estimate; a longer list (never seen) counts as a failed fast schedule.
      // (round 6: up to 128 queries, the whole small-batch regime of the register-resident stream kernel -- a search of 2.6-3.3 ms; at
      // thousands of queries the one wake-up per 90 ms search is nothing, and a re-score grid sized for twice the estimate would launch
      // millions of empty workgroups.  om_sim_topk has exactly ONE host synchronisation per search on this schedule: this read)
      const bool spec = bf16 && nq <= 128;
      const unsigned spec_cover = (unsigned)std::min<int64_t>(2 * list, SORT_CAP);
      if (spec) {
        if (rescore(spec_cover)) return 1;
        if (select(false)) return 1;
      }
      unsigned g[4];
      if (read_flags(g)) return 1;
      if (g[1] > (unsigned)g_info[3]) g_info[3] = g[1];
      if (!g[0] && !g[2] && !(spec && g[1] > spec_cover)) {
        if (spec) return 0;                      // everything is done
        done = N;
        f[1] = g[1];
        unsorted = true;                         // the lists are cut but not ordered: one sort at the very end
        careful = false;
      } else {                                   // rare: start over on the careful path
        g_info[2]++;
        hipLaunchKernelGGL(init_lists_kernel, dim3((unsigned)((nq + 255) / 256 + 2)), dim3(256), 0, s,
                           ws.cnt, ws.cnt_prev, ws.thr, nq, ws.flag);
        OM_LAUNCH_CHECK();
      }
    }
    if (careful) {                               // bootstrap of the step-by-step loop
      const int n = (int)std::min<int64_t>(N, DENSE_CHUNK);
      if (dense_step(0, n, bf16)) return 1;
      done = n;
      if (read_flags(f)) return 1;
      trace("boot", 0, n, f);
      if (f[2]) return 2;
    }
    while (done < N) {
      const int64_t list = std::max<unsigned>(f[1], 1u);
      // expected survivors of a chunk ~ list * chunk / done.  The per-query sort pads its list to the next
      // power of two, so aim just under 4096 keys (list + survivors) while the list is short enough --
      // filling half the free slots lands at ~4700 and sorts 8192 every round -- else half the free slots.
      const double want = list <= 2048 ? 3800.0 - (double)list : (double)(SORT_CAP - list) / 2.0;
      int64_t chunk = (int64_t)((double)done * want / (double)list);
      chunk = std::max<int64_t>(chunk, DENSE_CHUNK);
      chunk = std::min<int64_t>(chunk, N - done);
      OM_HIP(hipMemsetAsync(ws.flag, 0, 64, s));
      if (filter_step(done, chunk, bf16)) return 1;
      if (read_flags(f)) return 1;
      if (!f[0]) {
        if (select(bf16)) return 1;
      } else {
        g_info[2]++;
        // a list overflowed: rewind to the pre-chunk lists and redo the chunk densely
        hipLaunchKernelGGL(restore_cnt_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s,
                           ws.cnt, ws.cnt_prev, nq);
        OM_LAUNCH_CHECK();
        for (int64_t r = done; r < done + chunk; r += DENSE_CHUNK) {
          OM_HIP(hipMemsetAsync(ws.flag, 0, 64, s));
          if (dense_step(r, (int)std::min<int64_t>(DENSE_CHUNK, done + chunk - r), bf16)) return 1;
          if (bf16) {
            if (read_flags(f)) return 1;
            if (f[2]) return 2;
          }
        }
      }
      if (read_flags(f)) return 1;
      trace("scan", done, chunk, f);
      if (f[2]) return 2;
      done += chunk;
    }
    if (bf16) {
      // exact f32 re-score of the certified candidate set, then the true top-k
      if (rescore(std::max<unsigned>(f[1], 1u))) return 1;
      if (select(false)) return 1;
    } else if (unsorted) {
      if (select(false)) return 1;               // exact mode after the radix rounds: best k, in order
    }
    return 0;
  }
  const float* stats;
};
}  // namespace

extern "C" int om_sim_topk(int mode, const float* queries, int64_t n_queries,
                           const float* index_f32, const void* index_f16, const float* stats,
                           int64_t N, int d, int k, int64_t id_offset, float* out_scores,
                           int64_t* out_ids, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (n_queries <= 0) return 0;
  if (k < 1 || k > K_MAX) OM_FAIL("k must be in [1,2048]");
  if (N < 0 || N > 0xffffffffLL) OM_FAIL("shard rows must fit in 32 bits");
  if (d <= 0 || (d * 2) % GEMM_ROW_BYTES != 0) OM_FAIL("d must be a multiple of 64 (pad with zeros)");
  if (n_queries > 65535) OM_FAIL("at most 65535 queries per call");
  if (!queries || !out_scores || !out_ids) OM_FAIL("null argument");
  if (!workspace || ((uintptr_t)workspace & 255)) OM_FAIL("workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  Scan sc;
  sc.ws = carve_search(n_queries, d, (char*)workspace);
  if (sc.ws.total > workspace_bytes) OM_FAIL("workspace too small");
  sc.mode = mode; sc.q32 = queries; sc.idx32 = index_f32; sc.idx16 = (const f16_t*)index_f16;
  sc.nq = n_queries; sc.N = N; sc.d = d; sc.k = k; sc.s = s; sc.stats = stats;

  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               SORT_CAP * 8 + 16));
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel<float>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel<f16_t>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel6<float>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G6_LDS_BYTES));
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel6<f16_t>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G6_LDS_BYTES));
#define SR_ATTR(NB_, DBG_) OM_HIP(hipFuncSetAttribute((const void*)sim_stream_reg_kernel<f16_t, NB_, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, SR_LDS))
    SR_ATTR(1, 0); SR_ATTR(2, 0); SR_ATTR(4, 0); SR_ATTR(1, 1); SR_ATTR(2, 1); SR_ATTR(4, 1); SR_ATTR(1, 2); SR_ATTR(2, 2); SR_ATTR(4, 2);
#undef SR_ATTR
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel7<f16_t>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel7c<f16_t>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    OM_HIP(hipFuncSetAttribute((const void*)sim_filter_kernel7c16<f16_t>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, G7_LDS_BYTES));
    OM_HIP(hipFuncSetAttribute((const void*)select_radix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               SORT_CAP * 8 + 1024 + 64));
    attr_set = true;
  }
  for (auto& v : g_info) v = 0;
  if (N > 0) {
    if (!index_f32) OM_FAIL("index_f32 is null");
    int rc = 2;
    if (mode == OM_SEARCH_F16_RESCORE) {
      if (!index_f16 || !stats) OM_FAIL("f16 mode needs index_f16 and stats");
      rc = sc.run(true);
      if (rc == 1) return 1;
      g_info[0] = 1;
      if (rc == 2) g_info[4] = 1;
    }
    if (rc == 2) {                           // f32 scan (requested, or margin too wide)
      g_info[0] = 0;
      if (sc.run(false)) return 1;
    }
  } else {
    hipLaunchKernelGGL(init_lists_kernel, dim3((unsigned)((n_queries + 255) / 256)), dim3(256), 0, s,
                       sc.ws.cnt, sc.ws.cnt_prev, sc.ws.thr, n_queries, sc.ws.flag);
    OM_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(emit_kernel, dim3((unsigned)n_queries), dim3(256), 0, s, sc.ws.keys, sc.ws.cnt, k,
                     id_offset, out_scores, out_ids);
  OM_LAUNCH_CHECK();
  return 0;
}
