// Error channel + device probe of libdeepim_hip.so.
#include "common.h"
#include <cstring>

namespace dim {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace dim

extern "C" {

const char* dim_last_error(void) { return dim::err_buf(); }

int dim_device_info(char* name, int n) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return dim::set_err(DIM_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return dim::set_err(DIM_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (name && n > 0) {
    snprintf(name, n, "%s (%s)", p.name, p.gcnArchName);
  }
  return p.multiProcessorCount;
}

}  // extern "C"
