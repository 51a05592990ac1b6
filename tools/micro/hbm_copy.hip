// micro-benchmark: streaming copy / read / write bandwidth of HBM (buffers far larger than the 256 MiB Infinity Cache)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ src, float* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i < n; i += stride) { float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void write_k(float4* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const long bytes = 4L << 30;  // 4 GiB per buffer
  const long n = bytes / 16;
  float4 *a, *b; float* o;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 4);
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 16;
  float ms;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) {
        if (mode == 0) copy_k<<<grid, 256>>>(a, b, n);
        else if (mode == 1) read_k<<<grid, 256>>>(a, o, n);
        else write_k<<<grid, 256>>>(b, n);
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double moved = (mode == 0 ? 2.0 : 1.0) * bytes * 5;
    printf("%s: %.3f ms per pass, %.2f TB/s\n", mode == 0 ? "copy (read+write)" : mode == 1 ? "read" : "write", ms / 5, moved / (ms * 1e-3) / 1e12);
  }
  return 0;
}
