// Fourth-generation NT GEMM main loop for gfx950: a 256 x 256 tile on eight waves fed through a
// 4-deep ring of HALF-depth stages (K step = 64 bytes: 32 bf16 / 16 f32).
//
// Why (profiles/r01_gemm_variants_trace.log): with two 64 KiB stages only ONE stage (64 KiB per CU)
// can be in flight while the other is consumed, the LDS-DMA latency under load is ~3500 cycles, so
// every 2100-cycle MFMA step waits another ~1500 cycles for its successor.  Four 32 KiB stages keep
// THREE (96 KiB) in flight behind a counted `vmcnt(8)`; a step is 1024 MFMA cycles per SIMD, so the
// memory system gets ~3 steps of cover instead of 1.
//
//   stage = A tile [256 rows][64 B] (16 KiB) + B tile [256 rows][64 B] (16 KiB)
//   one global_load_lds_dwordx4 wave instruction = 1 KiB = 16 rows; per wave and stage 2 + 2
//   swizzle for 64-byte rows: physical 16-B slot = logical ^ ((row >> 2) & 3)  (a 256-B bank row holds
//   4 tile rows; every ds_read_b128 lane group then covers 16 distinct slots)
#pragma once
#include "gemm_core.h"

#define G4_BM 256
#define G4_BN 256
#define G4_THREADS 512
#define G4_ROW_BYTES 64
#define G4_OPERAND_BYTES (256 * 64)
#define G4_STAGE_BYTES (2 * G4_OPERAND_BYTES)
#define G4_STAGES 4
#define G4_LDS_BYTES (G4_STAGES * G4_STAGE_BYTES)

__device__ inline void g4_stage(const char* const (&pa)[2], const char* const (&pb)[2], size_t kbyte,
                                char* slot, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + kbyte), (lptr_t)(slot + (i * 8 + wave) * 1024),
                                     16, 0, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + kbyte),
                                     (lptr_t)(slot + G4_OPERAND_BYTES + (i * 8 + wave) * 1024), 16, 0, 0);
}

// acc[mi][ni] as in gemm_mainloop3 (wm = wave>>2, wn = wave&3; wave tile 128 x 64).
template <typename T>
__device__ inline void gemm_mainloop4(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                      int64_t ldb, int64_t M, int64_t N, int64_t K, int64_t m0,
                                      int64_t n0, char* smem, f32x16_t (&acc)[4][2],
                                      unsigned long long* tr = nullptr) {
  typedef typename MmaOps<T>::frag_t frag_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 2, wn = wave & 3;

  const char* pa[2];
  const char* pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (i * 8 + wave) * 16 + (lane >> 2);         // tile row fed by this lane
    const int c = (lane & 3) ^ ((r >> 2) & 3);               // logical chunk stored at slot lane&3
    int64_t ra = m0 + r; if (ra > M - 1) ra = M - 1;
    int64_t rb = n0 + r; if (rb > N - 1) rb = N - 1;
    pa[i] = (const char*)(A + ra * lda) + c * 16;
    pb[i] = (const char*)(B + rb * ldb) + c * 16;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = (int)((K * (int64_t)sizeof(T)) / G4_ROW_BYTES);
  const int key = (lane >> 2) & 3;   // == ((row>>2)&3) for row = 32*x + (lane&31)
  const int half = lane >> 5;
  const int rowa = (wm * 128 + (lane & 31)) * G4_ROW_BYTES;
  const int rowb = G4_OPERAND_BYTES + (wn * 64 + (lane & 31)) * G4_ROW_BYTES;

  if (tr && tid == 0) tr[1] = clock64();
  // prologue: three stages in flight
  g4_stage(pa, pb, 0, smem, wave);
  if (nk > 1) g4_stage(pa, pb, G4_ROW_BYTES, smem + G4_STAGE_BYTES, wave);
  if (nk > 2) g4_stage(pa, pb, 2 * G4_ROW_BYTES, smem + 2 * G4_STAGE_BYTES, wave);

  for (int t = 0; t < nk; ++t) {
    // tile t has landed for THIS wave once only the (up to two) newer tiles' DMAs are outstanding
    const int newer = nk - 1 - t;
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my LDS reads of tile t-1 are done
    __builtin_amdgcn_s_barrier();                           // -> true for every wave
    if (tr && tid == 0 && t < 12) tr[3 + t] = clock64();
    if (t + 3 < nk)                                         // slot (t+3)&3 == slot of tile t-1: free now
      g4_stage(pa, pb, (size_t)(t + 3) * G4_ROW_BYTES, smem + ((t + 3) & 3) * G4_STAGE_BYTES, wave);
    const char* cur = smem + (t & 3) * G4_STAGE_BYTES;
    frag_t a[2][4], b[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int slot = (((kk << 1) | half) ^ key) << 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[kk][mi] = *(const frag_t*)(cur + rowa + mi * 32 * G4_ROW_BYTES + slot);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[kk][ni] = *(const frag_t*)(cur + rowb + ni * 32 * G4_ROW_BYTES + slot);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) MmaOps<T>::mma(a[kk][mi], b[kk][ni], acc[mi][ni]);
    // issue order: all 12 LDS reads of the step, then sub-step 0's MFMAs (which only wait for the
    // first 6 reads), then sub-step 1's
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MmaOps<T>::kMfmaPerMma * 16, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                             // everyone is done with the ring
}

__device__ inline void g4_tile_coords(int64_t M, int64_t N, int group_m, int64_t& m0, int64_t& n0) {
  const int64_t ntm = (M + G4_BM - 1) / G4_BM, ntn = (N + G4_BN - 1) / G4_BN;
  int64_t tm, tn;
  gemm_tile_coords(ntm, ntn, group_m, tm, tn);
  m0 = tm * G4_BM;
  n0 = tn * G4_BN;
}
