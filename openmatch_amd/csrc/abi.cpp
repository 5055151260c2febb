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
