// Which f32 MFMA shape holds the higher clock / delivers more FLOP/s under sustained load on this chip?  Bare loops, operands in
// registers (random data), one or two waves per SIMD, every CU busy; in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_shape_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFF]; }
  long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + k) & 3], acc[k], 0, 0, 0);   // 16 x 4096 FLOP
    }
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) sum += acc[k][r];
  } else {
    f32x4 acc[16];
    for (int k = 0; k < 16; ++k) for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + k) & 3], b[(u + (k >> 2)) & 3], acc[k], 0, 0, 0);   // 32 x 2048 FLOP
    }
    for (int k = 0; k < 16; ++k) for (int r = 0; r < 4; ++r) sum += acc[k][r];
  }
  long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  out[tid] = sum;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = c1 - c0; }
}

int main() {
  const int CUS = 256, WG = 512, iters = 20000;
  float *in, *out; long long* st;
  hipMalloc(&in, 65536 * 4); hipMalloc(&out, CUS * WG * 4 * 2); hipMalloc(&st, CUS * 2 * 2 * 8);
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  for (int blocks_per_cu = 1; blocks_per_cu <= 1; ++blocks_per_cu)
    for (int shape : {32, 16, 32, 16}) {
      const int grid = CUS * blocks_per_cu;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {   // a few back-to-back launches so the clock settles
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(grid), dim3(WG), 0, 0, in, out, st, iters);
        else hipLaunchKernelGGL(probe<16>, dim3(grid), dim3(WG), 0, 0, in, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> s(grid * 2);
      hipMemcpy(s.data(), st, grid * 2 * 8, hipMemcpyDeviceToHost);
      std::vector<double> clk;
      for (int i = 0; i < grid; ++i) clk.push_back((double)s[2 * i + 1] / ((double)s[2 * i] * 10.0));
      std::sort(clk.begin(), clk.end());
      const double flop = (double)grid * (WG / 64) * iters * 16.0 * 4096.0;
      printf("shape %2dx%2d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s  in-kernel clock GHz med %.2f (min %.2f max %.2f)\n", shape, shape, WG / 64 / 4 * blocks_per_cu, ms,
             flop / ms / 1e9, clk[grid / 2], clk[0], clk[grid - 1]);
    }
  return 0;
}
