// Shared helpers for libdeepim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/deepim_hip.h"

namespace dim {

constexpr int kWave = 64;

// thread-local message behind dim_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DIM_OK;
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dim

#define DIM_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return dim::set_err(DIM_ERR_ARG, __VA_ARGS__);  \
  } while (0)
