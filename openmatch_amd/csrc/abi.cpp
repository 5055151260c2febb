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

// ---- run-time switches (A/B measurements, tests) ------------------------------------------------
// Read once from the environment, overridable through om_debug_option(); the hot paths read an atomic.
#include <atomic>
#include <map>
#include <tuple>
namespace {
std::atomic<int> g_opt[OM_OPT_COUNT];
std::atomic<bool> g_opt_init{false};
std::mutex g_opt_mu;
void opt_init() {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  if (g_opt_init.load()) return;
  const char* e = getenv("OM_ENCODER_FUSED_LN");
  g_opt[OM_OPT_ENCODER_FUSED_LN] = e ? atoi(e) : 1;
  g_opt[OM_OPT_ENCODER_DEBUG] = getenv("OM_ENCODER_DEBUG") ? 1 : 0;
  e = getenv("OM_ATTENTION_FAST");
  g_opt[OM_OPT_ATTENTION_FAST] = e ? atoi(e) : 1;
  e = getenv("OM_SCAN_GEN7");
  g_opt[OM_OPT_SCAN_GEN7] = e ? atoi(e) : 1;
  e = getenv("OM_SCAN_GROWTH");
  g_opt[OM_OPT_SCAN_GROWTH] = e ? atoi(e) : 60;
  g_opt[OM_OPT_WGRAD_DEBUG] = 0;
  g_opt[OM_OPT_ATTENTION_DEBUG] = 0;
  e = getenv("OM_ENCODER_PINGPONG");
  g_opt[OM_OPT_ENCODER_PINGPONG] = e ? atoi(e) : 1;
  e = getenv("OM_TRAIN_WGRAD_STREAM");
  g_opt[OM_OPT_TRAIN_WGRAD_STREAM] = e ? atoi(e) : 1;
  e = getenv("OM_SCAN_QGROUP");
  g_opt[OM_OPT_SCAN_QGROUP] = e ? atoi(e) : 8;
  e = getenv("OM_GEMM_GROUP_M");
  g_opt[OM_OPT_GEMM_GROUP_M] = e ? atoi(e) : 8;
  e = getenv("OM_ENCODER_TWO_PLANE");
  g_opt[OM_OPT_ENCODER_TWO_PLANE] = e ? atoi(e) : 3;
  e = getenv("OM_GEMM_VARIANT");
  g_opt[OM_OPT_GEMM_VARIANT] = e ? atoi(e) : 0;
  e = getenv("OM_SEARCH_DEBUG");
  g_opt[OM_OPT_SEARCH_DEBUG] = e ? (atoi(e) ? atoi(e) : 1) : 0;
  e = getenv("OM_TRAIN_WGRAD_BATCH");
  g_opt[OM_OPT_TRAIN_WGRAD_BATCH] = e ? atoi(e) : 4;
  e = getenv("OM_GEMM_MAX_GRID");
  g_opt[OM_OPT_GEMM_MAX_GRID] = e ? atoi(e) : 0;
  e = getenv("OM_TRAIN_TAPE_GRAD");
  g_opt[OM_OPT_TRAIN_TAPE_GRAD] = e ? atoi(e) : 1;
  e = getenv("OM_TRAIN_RES32");
  g_opt[OM_OPT_TRAIN_RES32] = e ? atoi(e) : 1;
  e = getenv("OM_GEMM_CONT");
  g_opt[OM_OPT_GEMM_CONT] = e ? atoi(e) : 495;
  e = getenv("OM_GEMM_SKINNY_M");
  g_opt[OM_OPT_GEMM_SKINNY_M] = e ? atoi(e) : 1024;
  e = getenv("OM_FEW_ROWS_LN_FUSE");
  g_opt[OM_OPT_FEW_ROWS_LN_FUSE] = e ? atoi(e) : 64;
  g_opt_init.store(true);
}
}  // namespace
// true when the option's environment variable was given (a default that depends on the call's shape must not override it)
bool om_option_is_set(int opt) {
  if (opt == OM_OPT_SCAN_GROWTH) return getenv("OM_SCAN_GROWTH") != nullptr;
  return false;
}
int om_option(int opt) {
  if (!g_opt_init.load(std::memory_order_acquire)) opt_init();
  return g_opt[opt].load(std::memory_order_relaxed);
}
extern "C" int om_debug_option(int opt, int value) {
  if (opt < 0 || opt >= OM_OPT_COUNT) { om_set_error("om_debug_option: unknown option"); return 1; }
  if (!g_opt_init.load(std::memory_order_acquire)) opt_init();
  g_opt[opt].store(value);
  return 0;
}

// T5 relative-position bucket table of a sequence length, resident on the device: built and uploaded ONCE per
// (device, L, buckets, max distance) -- no pageable copy and no stream synchronisation on later forwards.
int om_t5_lut_device(int L, int buckets, int max_dist, const int** out) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int>, int*> cache;
  int dev = 0;
  OM_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto key = std::make_tuple(dev, L, buckets, max_dist);
  auto it = cache.find(key);
  if (it == cache.end()) {
    std::vector<int> lut(2 * L);
    for (int rel = -(L - 1); rel <= L - 1; ++rel) lut[rel + (L - 1)] = om_t5_relative_bucket(rel, buckets, max_dist);
    int* d = nullptr;
    OM_HIP(hipMalloc(&d, (size_t)(2 * L) * sizeof(int)));
    OM_HIP(hipMemcpy(d, lut.data(), (size_t)(2 * L - 1) * sizeof(int), hipMemcpyHostToDevice));
    it = cache.emplace(key, d).first;
  }
  *out = it->second;
  return 0;
}
