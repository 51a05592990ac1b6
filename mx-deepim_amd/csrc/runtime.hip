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

// device-to-device copy of 4-byte words as a KERNEL node (16-byte path when both pointers and the count allow it)
__global__ __launch_bounds__(256) void copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, long nwords, int vec4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec4) {
    if (i * 4 < nwords) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  } else if (i < nwords) {
    dst[i] = src[i];
  }
}

// rows x width words between arrays with different row pitches (a channel window of an (O, I, kh, kw) weight: width = channels * kh * kw)
__global__ void copy_rows_kernel(unsigned* __restrict__ dst, long dst_pitch, const unsigned* __restrict__ src, long src_pitch, long rows, long width) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * width) return;
  const long r = i / width, c = i - r * width;
  dst[r * dst_pitch + c] = src[r * src_pitch + c];
}

}  // namespace dim

extern "C" {

int dim_copy_rows(void* dst, long dst_pitch_words, const void* src, long src_pitch_words, long rows, long width_words, void* stream) {
  if (rows == 0 || width_words == 0) return DIM_OK;
  DIM_REQUIRE(dst && src, "null pointer");
  DIM_REQUIRE(rows > 0 && width_words > 0 && dst_pitch_words >= width_words && src_pitch_words >= width_words, "bad geometry");
  hipLaunchKernelGGL(dim::copy_rows_kernel, dim3(dim::ceil_div(rows * width_words, 256)), dim3(256), 0, dim::as_stream(stream),
                     reinterpret_cast<unsigned*>(dst), dst_pitch_words, reinterpret_cast<const unsigned*>(src), src_pitch_words, rows,
                     width_words);
  return dim::check_launch("copy_rows");
}

int dim_copy_words(void* dst, const void* src, long nwords, void* stream) {
  if (nwords == 0) return DIM_OK;
  DIM_REQUIRE(dst && src, "null pointer");
  const int vec4 = (nwords % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0);
  const long n = vec4 ? nwords / 4 : nwords;
  hipLaunchKernelGGL(dim::copy_words_kernel, dim3(dim::ceil_div(n, 256)), dim3(256), 0, dim::as_stream(stream),
                     reinterpret_cast<unsigned*>(dst), reinterpret_cast<const unsigned*>(src), nwords, vec4);
  return dim::check_launch("copy_words");
}

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
