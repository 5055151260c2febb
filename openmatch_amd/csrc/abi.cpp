// Error plumbing + trivial queries of the C ABI.
#include "common.h"

static thread_local std::string g_last_error;
void om_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" const char* om_last_error(void) { return g_last_error.c_str(); }
extern "C" int om_abi_version(void) { return OM_ABI_VERSION; }
extern "C" int om_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    om_set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return -1;
  }
  return n;
}

// ---- per-launch kernel timing ------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct Span { hipEvent_t a, b; };
struct ClassStat { std::vector<Span> spans; double flops = 0; size_t used = 0; };
bool g_timing = false;
ClassStat g_stat[3];
std::mutex g_mu;
}  // namespace
bool om_timing_on() { return g_timing; }
void om_timing_begin(int c, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  ClassStat& st = g_stat[c];
  if (st.used == st.spans.size()) {
    Span sp;
    (void)hipEventCreate(&sp.a);
    (void)hipEventCreate(&sp.b);
    st.spans.push_back(sp);
  }
  (void)hipEventRecord(st.spans[st.used].a, s);
}
void om_timing_end(int c, hipStream_t s, double flops) {
  std::lock_guard<std::mutex> lk(g_mu);
  ClassStat& st = g_stat[c];
  (void)hipEventRecord(st.spans[st.used].b, s);
  st.used++;
  st.flops += flops;
}
extern "C" int om_kernel_timing_enable(int enable) { g_timing = enable != 0; return 0; }
extern "C" int om_kernel_timing_read(int c, double* total_ms, int64_t* launches, double* flops) {
  if (c < 0 || c > 2) { om_set_error("om_kernel_timing_read: bad class"); return 1; }
  std::lock_guard<std::mutex> lk(g_mu);
  ClassStat& st = g_stat[c];
  double ms = 0;
  for (size_t i = 0; i < st.used; ++i) {
    hipError_t e = hipEventSynchronize(st.spans[i].b);
    float t = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, st.spans[i].a, st.spans[i].b);
    if (e != hipSuccess) { om_set_error(std::string("om_kernel_timing_read: ") + hipGetErrorString(e)); return 1; }
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int64_t)st.used;
  if (flops) *flops = st.flops;
  st.used = 0; st.flops = 0;
  return 0;
}
