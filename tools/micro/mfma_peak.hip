// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate vs accumulators per wave and waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int threads, int iters = 2000) {
  int cus = 256;
  float* out;
  hipMalloc(&out, (size_t)cus * blocks_per_cu * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<cus * blocks_per_cu, threads>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<cus * blocks_per_cu, threads>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)cus * blocks_per_cu * threads / 64;
  double flops = waves * iters * 16.0 * NACC * 32 * 32 * 2 * 2;
  printf("NACC=%d blocks/CU=%d threads=%d iters=%d (waves/SIMD=%.1f): %.3f ms  %.1f TF\n", NACC, blocks_per_cu, threads, iters,
         blocks_per_cu * threads / 64 / 4.0, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<1>(1, 256); run<1>(2, 256); run<1>(4, 256); run<1>(6, 256);
  run<2>(1, 256); run<2>(2, 256); run<4>(1, 256); run<4>(2, 256); run<4>(3, 256);
  // sustained: the same loop for ~10 ms, ~100 ms, ~0.5 s (power management lowers the clock under a long MFMA burst)
  run<4>(1, 256, 6000); run<4>(1, 256, 60000); run<4>(1, 256, 300000); run<4>(1, 256, 2000);
  return 0;
}
