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

// rows x width words between arrays with different row pitches (a channel window of an (O, I, kh, kw) weight: width = channels * kh * kw;
// a channel range of an NHWC map: rows = pixels).  ADD: dst += src as floats.  V4: 16 bytes per thread (width, pitches, pointers aligned).
template <bool ADD, bool V4>
__global__ void copy_rows_kernel(float* __restrict__ dst, long dst_pitch, const float* __restrict__ src, long src_pitch, long rows, long width) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long wq = V4 ? width / 4 : width;
  if (i >= rows * wq) return;
  const long r = i / wq, c = (i - r * wq) * (V4 ? 4 : 1);
  if (V4) {
    float4 v = *reinterpret_cast<const float4*>(src + r * src_pitch + c);
    float4* o = reinterpret_cast<float4*>(dst + r * dst_pitch + c);
    if (ADD) {
      const float4 a = *o;
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    *o = v;
  } else {
    const float v = src[r * src_pitch + c];
    float* o = dst + r * dst_pitch + c;
    *o = ADD ? *o + v : v;
  }
}

__global__ void fill_words_kernel(unsigned* __restrict__ dst, long nwords, unsigned value) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) dst[i] = value;
}

template <bool ADD>
static int copy_rows_launch(void* dst, long dst_pitch, const void* src, long src_pitch, long rows, long width, void* stream) {
  if (rows == 0 || width == 0) return DIM_OK;
  DIM_REQUIRE(dst && src, "null pointer");
  DIM_REQUIRE(rows > 0 && width > 0 && dst_pitch >= width && src_pitch >= width, "bad geometry");
  const bool v4 = width % 4 == 0 && dst_pitch % 4 == 0 && src_pitch % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  const long n = rows * (v4 ? width / 4 : width);
  float* d = reinterpret_cast<float*>(dst);
  const float* s = reinterpret_cast<const float*>(src);
  if (v4)
    hipLaunchKernelGGL((copy_rows_kernel<ADD, true>), dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), d, dst_pitch, s, src_pitch, rows, width);
  else
    hipLaunchKernelGGL((copy_rows_kernel<ADD, false>), dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), d, dst_pitch, s, src_pitch, rows, width);
  return check_launch(ADD ? "add_rows" : "copy_rows");
}

}  // namespace dim

extern "C" {

int dim_copy_rows(void* dst, long dst_pitch_words, const void* src, long src_pitch_words, long rows, long width_words, void* stream) {
  return dim::copy_rows_launch<false>(dst, dst_pitch_words, src, src_pitch_words, rows, width_words, stream);
}

int dim_add_rows(float* dst, long dst_pitch, const float* src, long src_pitch, long rows, long width, void* stream) {
  return dim::copy_rows_launch<true>(dst, dst_pitch, src, src_pitch, rows, width, stream);
}

int dim_fill_words(void* dst, long nwords, unsigned value, void* stream) {
  if (nwords == 0) return DIM_OK;
  DIM_REQUIRE(dst && nwords > 0, "null pointer");
  hipLaunchKernelGGL(dim::fill_words_kernel, dim3(dim::ceil_div(nwords, 256)), dim3(256), 0, dim::as_stream(stream),
                     reinterpret_cast<unsigned*>(dst), nwords, value);
  return dim::check_launch("fill_words");
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
