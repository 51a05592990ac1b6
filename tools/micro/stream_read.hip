// What does a SHORT streaming read reach on this part?  Reads `mb` MB once per launch (float4 loads, `U` loads in flight per thread,
// persistent grid of G workgroups x 256 threads, each workgroup a contiguous range), one float written per thread-group; between two
// timed launches a 1 GiB buffer is streamed so the bytes come from HBM.  The fc6 weight stream (84 MB, csrc/fc.hip) is this shape.
// build: hipcc -O3 --offload-arch=gfx950 stream_read.hip -o stream_read ; run: ./stream_read [mb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const v4f* __restrict__ p, long n4, float* __restrict__ out) {
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long b = (long)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = b + threadIdx.x; i < e; i += 256L * U) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + 256L * u;
      v[u] = j < e ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : (v4f){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

__global__ void touch_kernel(v4f* p, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] += (v4f){1.f, 1.f, 1.f, 1.f};
}

template <int U, bool NT>
static void run(const v4f* p, long n4, float* out, v4f* big, long big4, int G, double mb) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, 0, big, big4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((read_kernel<U, NT>), dim3(G), dim3(256), 0, 0, p, n4, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  printf("%5.0f MB  G=%5d  U=%d %s: median %.1f us  min %.1f us  -> %.2f TB/s\n", mb, G, U, NT ? "nt" : "  ", ts[ts.size() / 2], ts[0],
         mb * 1e6 / ts[ts.size() / 2] / 1e6);
}

int main(int argc, char** argv) {
  const double mb = argc > 1 ? atof(argv[1]) : 83.9;
  const long n4 = (long)(mb * 1e6 / 16);
  v4f *p, *big;
  float* out;
  const long big4 = (1L << 30) / 16;
  hipMalloc(&p, n4 * 16);
  hipMalloc(&big, big4 * 16);
  hipMalloc(&out, 1 << 20);
  hipMemset(p, 0, n4 * 16);
  hipMemset(big, 0, big4 * 16);
  for (int G : {256, 512, 1024, 2048, 4096}) {
    run<4, false>(p, n4, out, big, big4, G, mb);
    run<8, false>(p, n4, out, big, big4, G, mb);
    run<8, true>(p, n4, out, big, big4, G, mb);
    run<16, true>(p, n4, out, big, big4, G, mb);
  }
  return 0;
}
