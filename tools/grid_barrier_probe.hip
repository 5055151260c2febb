// What a device-wide barrier costs on MI355X against a kernel boundary (round 6; the question behind "one persistent kernel for a
// few-rows forward"): G workgroups of 512 threads run R rounds of
//   v0  arrive (relaxed atomic add, agent scope) + spin on the counter                      -- the bare rendezvous
//   v1  the same with release / acquire ordering (L2 write-back + invalidate on gfx950: 8 XCDs, 8 L2s)
//   v2  v1 + every workgroup writes a 512-byte slice of a 48 KB "activation" and reads all of it back after the barrier (checked)
//   v3  v2 with the slice written and read by system-coherent (nontemporal, L2-bypassing) accesses and a relaxed rendezvous
// and, for comparison, R empty launches of the same grid on one stream.          hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* counter, unsigned* err, float* act, float* sink, int R) {
  const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) {
    if (V >= 2) {                                   // this workgroup's slice: 128 floats, value = round
      if (t < 128) {
        if (V == 3) __builtin_nontemporal_store((float)(r + 1), act + g * 128 + t);
        else act[g * 128 + t] = (float)(r + 1);
      }
    }
    __syncthreads();
    if (t == 0) {
      const unsigned target = (unsigned)(r + 1) * G;
      if (V == 0 || V == 3) {
        if (V == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // orders the nontemporal stores before the arrive
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *err = 1; break; }
        }
      } else {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *err = 1; break; }
        }
      }
    }
    __syncthreads();
    if (V >= 2) {                                   // read the whole activation: G * 128 floats = 128 KB at G = 256
      for (int i = t; i < G * 128; i += 512) {
        const float v = V == 3 ? __builtin_nontemporal_load(act + i) : act[i];
        if (v != (float)(r + 1)) *err = 2;
        acc += v;
      }
      __syncthreads();                              // nobody overwrites before everyone has read: second rendezvous is the next round's
      if (t == 0) {                                 // (a real pipeline alternates two buffers instead; here: a second barrier, counted)
      }
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

// v2 / v3 need the read of round r to finish before the write of round r + 1: alternate two activation buffers
template <int V>
__global__ __launch_bounds__(512) void pipeline_kernel(unsigned* counter, unsigned* err, float* act0, float* act1, float* sink, int R) {
  const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) {
    float* act = (r & 1) ? act1 : act0;
    if (t < 128) {
      if (V == 3) __builtin_nontemporal_store((float)(r + 1), act + g * 128 + t);
      else act[g * 128 + t] = (float)(r + 1);
    }
    __syncthreads();
    if (t == 0) {
      const unsigned target = (unsigned)(r + 1) * G;
      if (V == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *err = 1; break; }
        }
      } else {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *err = 1; break; }
        }
      }
    }
    __syncthreads();
    for (int i = t; i < G * 128; i += 512) {
      const float v = V == 3 ? __builtin_nontemporal_load(act + i) : act[i];
      if (v != (float)(r + 1)) *err = 2;
      acc += v;
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

__global__ __launch_bounds__(512) void empty_kernel(float* sink) { if (sink[0] == 12345.f) sink[1] = 1.f; }

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int G = p.multiProcessorCount, R = 2000;
  unsigned *counter, *err; float *act0, *act1, *sink;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&act0, G * 512)); CK(hipMalloc(&act1, G * 512)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("{\"workgroups\": %d, \"rounds\": %d", G, R);
  for (int pass = 0; pass < 2; ++pass)
  for (int v = 0; v < 4; ++v) {
    CK(hipMemset(counter, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(sink, 0, 64));
    CK(hipEventRecord(e0, 0));
    if (v == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(G), dim3(512), 0, 0, counter, err, act0, sink, R);
    if (v == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(G), dim3(512), 0, 0, counter, err, act0, sink, R);
    if (v == 2) hipLaunchKernelGGL(pipeline_kernel<2>, dim3(G), dim3(512), 0, 0, counter, err, act0, act1, sink, R);
    if (v == 3) hipLaunchKernelGGL(pipeline_kernel<3>, dim3(G), dim3(512), 0, 0, counter, err, act0, act1, sink, R);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    if (pass) printf(", \"v%d_us_per_round\": %.3f, \"v%d_err\": %u", v, ms * 1e3 / R, v, herr);
  }
  for (int pass = 0; pass < 2; ++pass) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(512), 0, 0, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf(", \"empty_launch_us\": %.3f", ms * 1e3 / R);
  }
  printf("}\n");
  return 0;
}
