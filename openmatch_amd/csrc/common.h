// Shared device/host helpers for the gfx950 kernels.  Wavefront = 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/openmatch_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef _Float16 f16_t;                                         // IEEE half (search shadow index)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define OM_WAVE 64

// ---- error plumbing -------------------------------------------------------
void om_set_error(const std::string& msg);
#define OM_FAIL(msg)                                                        \
  do {                                                                      \
    om_set_error(std::string(__func__) + ": " + (msg));                     \
    return 1;                                                               \
  } while (0)
#define OM_HIP(expr)                                                        \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) {                                                 \
      om_set_error(std::string(__func__) + ": " #expr " -> " + hipGetErrorString(_e)); \
      (void)hipGetLastError(); /* reported here: do not leave it sticky for the host framework's next check */ \
      return 1;                                                             \
    }                                                                       \
  } while (0)
#define OM_LAUNCH_CHECK() OM_HIP(hipGetLastError())

// ---- bf16 <-> f32 ----------------------------------------------------------
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
// round-to-nearest-even, NaN preserved (same rule as torch's c10::BFloat16).  On the device the
// __bf16 cast lets hipcc use gfx950's v_cvt_pk_bf16_f32 (one instruction per two values).
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
  union { uint32_t u; float f; } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((c.u >> 16) | 0x40);
  uint32_t lsb = (c.u >> 16) & 1u;
  return (bf16_t)((c.u + 0x7fffu + lsb) >> 16);
}

template <typename T> struct ElemOps;
template <> struct ElemOps<float> {
  static constexpr int dtype = OM_F32;
  __device__ static inline float load(const float* p) { return *p; }
  __device__ static inline void store(float* p, float v) { *p = v; }
};
template <> struct ElemOps<f16_t> {
  static constexpr int dtype = OM_F16;
  __device__ static inline float load(const f16_t* p) { return (float)*p; }
  __device__ static inline void store(f16_t* p, float v) { *p = (f16_t)v; }
};
template <> struct ElemOps<bf16_t> {
  static constexpr int dtype = OM_BF16;
  __device__ static inline float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static inline void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- packed pairs of the 16-bit formats: one dword <-> two floats ------------------------------
// The 16-bit kernels (GEMM epilogues, attention, LayerNorm) are written once for both storage formats: bfloat16
// (bf16_t, raw bits) and IEEE half (f16_t).  pack2 rounds to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).
#if defined(__HIPCC__)
typedef __bf16 om_bf16x2_hw_t __attribute__((ext_vector_type(2)));
typedef _Float16 om_f16x2_hw_t __attribute__((ext_vector_type(2)));
typedef float om_f32x2_t __attribute__((ext_vector_type(2)));
template <typename T> struct Half16;
template <> struct Half16<bf16_t> {
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    const om_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, om_bf16x2_hw_t));
  }
  __device__ static __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  __device__ static __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  __device__ static __forceinline__ uint32_t bits(float v) { return f32_to_bf16(v); }          // one value -> its 16 bits
  __device__ static __forceinline__ float value(uint32_t b) { return __uint_as_float(b << 16); }
};
template <> struct Half16<f16_t> {
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    const om_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, om_f16x2_hw_t));
  }
  __device__ static __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(om_f16x2_hw_t, w)[0]; }
  __device__ static __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(om_f16x2_hw_t, w)[1]; }
  __device__ static __forceinline__ uint32_t bits(float v) { return __builtin_bit_cast(unsigned short, (f16_t)v); }
  __device__ static __forceinline__ float value(uint32_t b) { return (float)__builtin_bit_cast(f16_t, (unsigned short)b); }
};
template <> struct Half16<float> {     // never executed: lets `if (sizeof(OutT) == 2)` branches of f32 instantiations compile
  __device__ static __forceinline__ uint32_t pack2(float lo, float) { return __float_as_uint(lo); }
  __device__ static __forceinline__ float lo(uint32_t w) { return __uint_as_float(w); }
  __device__ static __forceinline__ float hi(uint32_t w) { return __uint_as_float(w); }
  __device__ static __forceinline__ uint32_t bits(float v) { return __float_as_uint(v); }
  __device__ static __forceinline__ float value(uint32_t b) { return __uint_as_float(b); }
};
#endif

// ---- wave reductions (64 lanes) ---------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- order-preserving float <-> uint32 (for sort keys / atomic max) ---------
__host__ __device__ inline uint32_t f32_orderable(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ inline float orderable_f32(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return c.f;
}

// XCD-aware bijective remap of a 1-D grid: consecutive work ids land on one XCD
// (block b runs on XCD b % 8 on MI355X; a speed hint only, never correctness).
__device__ inline unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  unsigned xcd = bid % nx, q = nwg / nx, r = nwg % nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / nx;
}

// ---- run-time switches and caches (abi.cpp) -------------------------------------
bool om_option_is_set(int opt);                                           // its environment variable was given (OM_OPT_SCAN_GROWTH only)
int om_option(int opt);                                                   // OM_OPT_* of include/openmatch_hip.h
int om_t5_lut_device(int L, int buckets, int max_dist, const int** out);  // device-resident bucket table, built once

// ---- optional per-launch timing (abi.cpp) ------------------------------------
bool om_timing_on();
void om_timing_begin(int kernel_class, hipStream_t s);
void om_timing_end(int kernel_class, hipStream_t s, double flops);

static inline int64_t ceil_div_i64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
