// depth -> optical-flow label kernel on device-resident tensors.
// Replaces /root/reference/lib/flow_c/gpu_flow_kernel.cu:32-69 (flow_kernel) and its host wrapper
// _flow (:82-148) which cudaMalloc/H2D/D2H/cudaFree on every call.  Same per-pixel arithmetic
// (float32, same operation order), output flow in (dy, dx) order, valid in {0,1}.
// HBM-bound: 2 planes read + 3 planes written per sample = 6.144 MB at 480x640.
#include <hip/hip_fp16.h>

#include "common.h"

namespace dim {

// one pixel of flow_kernel (gpu_flow_kernel.cu:32-69), arithmetic as written there
__device__ __forceinline__ void depth_flow_pixel(float d, int w, int h, const float* __restrict__ kt, const float* __restrict__ tgt_b,
                                                 float i0, float i1, float i2, float i3, float i4, float i5, int H, int W, float& fh,
                                                 float& fw, float& va) {
  // no FMA contraction: the CUDA reference compiles these as written only if nvcc does not fuse;
  // fusing changes results by <= 1 ulp, the oracle comparison tolerates that on `flow` and the
  // tests exclude pixels within 1e-6 of the 3e-3 / bounds predicates.
  float x = (w * i0 + h * i1 + i2) * d;
  float y = (w * i3 + h * i4 + i5) * d;
  float z = d;
  fh = 0.f; fw = 0.f; va = 0.f;
  if (d > 1E-3) {
    float xp = x * kt[0] + y * kt[1] + z * kt[2] + kt[3];
    float yp = x * kt[4] + y * kt[5] + z * kt[6] + kt[7];
    float zp = (float)((double)(x * kt[8] + y * kt[9] + z * kt[10] + kt[11]) + 1E-15);
    float wp = xp / zp, hp = yp / zp;
    int wi = (int)round((double)wp), hi = (int)round((double)hp);
    if (wp >= 0 && wp <= W - 1 && hp >= 0 && hp <= H - 1) {
      float dt = tgt_b[(long)hi * W + wi];
      if (fabsf(zp - dt) < 3E-3) {
        fh = hp - h;
        fw = wp - w;
        va = 1.f;
      }
    }
  }
}

// one thread per pixel (any W)
__global__ __launch_bounds__(256) void depth_flow_kernel(const float* __restrict__ depth_src, const float* __restrict__ depth_tgt,
                                                         const float* __restrict__ KT, float i0, float i1, float i2, float i3,
                                                         float i4, float i5, int H, int W, float* __restrict__ flow,
                                                         float* __restrict__ valid) {
  const int b = blockIdx.z, h = blockIdx.y;
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const long plane = (long)H * W;
  const long index = (long)b * plane + (long)h * W + w;
  float fh, fw, va;
  depth_flow_pixel(depth_src[index], w, h, KT + 12 * b, depth_tgt + (long)b * plane, i0, i1, i2, i3, i4, i5, H, W, fh, fw, va);
  flow[((long)b * 2 + 0) * plane + (long)h * W + w] = fh;
  flow[((long)b * 2 + 1) * plane + (long)h * W + w] = fw;
  valid[index] = va;
}

// W % 4 == 0 and 16-byte aligned planes: one thread per four consecutive pixels -- one 16-byte load of the source depth, three
// 16-byte stores (the one-pixel-per-thread form moved 4 bytes per lane and instruction and ran at 3.7 TB/s of its 6.1 MB per pair);
// the target-depth gather stays one dword per pixel (it lands within a few rows of the source pixel: cache hits)
__global__ __launch_bounds__(256) void depth_flow_quad_kernel(const float* __restrict__ depth_src, const float* __restrict__ depth_tgt,
                                                              const float* __restrict__ KT, float i0, float i1, float i2, float i3,
                                                              float i4, float i5, int H, int W, float* __restrict__ flow,
                                                              float* __restrict__ valid) {
  const int b = blockIdx.y;
  const long plane = (long)H * W;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;   // quad index inside the sample
  const long pix = 4 * q;
  if (pix >= plane) return;
  const int h = (int)(pix / W), w0 = (int)(pix - (long)h * W);
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f d4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(depth_src + (long)b * plane + pix));
  const float d[4] = {d4.x, d4.y, d4.z, d4.w};
  float fh[4], fw[4], va[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    depth_flow_pixel(d[k], w0 + k, h, KT + 12 * b, depth_tgt + (long)b * plane, i0, i1, i2, i3, i4, i5, H, W, fh[k], fw[k], va[k]);
  __builtin_nontemporal_store((v4f){fh[0], fh[1], fh[2], fh[3]}, reinterpret_cast<v4f*>(flow + ((long)b * 2 + 0) * plane + pix));
  __builtin_nontemporal_store((v4f){fw[0], fw[1], fw[2], fw[3]}, reinterpret_cast<v4f*>(flow + ((long)b * 2 + 1) * plane + pix));
  __builtin_nontemporal_store((v4f){va[0], va[1], va[2], va[3]}, reinterpret_cast<v4f*>(valid + (long)b * plane + pix));
}

// Test-time flow error (reference deepim/core/tester.py:500-512 accumulation, :719-736 calc_EPE_one_pair, :706-716 the
// [flow, visible, bg] list of par_generate_gt):  point_diff = sqrt((gt0 - pred0)^2 + (gt1 - pred1)^2) with the prediction
// stored as float16 first (tester.py:485-487 `.astype("float16")`, round to nearest even) and the arithmetic in float64
// (calc_flow returns float64, so numpy promotes).  Per sample: sum over all pixels, over visible == 1, over
// visible or bg (bg = visible == 0 and depth_rendered == 0), and the two pixel counts.
// Two deterministic stages: every workgroup leaves its five partial sums (fixed tree), one wave per sample adds them in order.
constexpr int kEpeBlocks = 120;  // workgroups per sample
__global__ __launch_bounds__(256) void flow_epe_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                               const float* __restrict__ visible, const float* __restrict__ depth_ren,
                                                               long plane, double* __restrict__ partial) {
  const int b = blockIdx.y;
  const float* p0 = pred + (long)b * 2 * plane;
  const float* g0 = gt + (long)b * 2 * plane;
  const float* vi = visible + (long)b * plane;
  const float* dr = depth_ren + (long)b * plane;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < plane; o += (long)gridDim.x * blockDim.x) {
    const double q0 = (double)__half2float(__float2half_rn(p0[o])), q1 = (double)__half2float(__float2half_rn(p0[plane + o]));
    const double d0 = (double)g0[o] - q0, d1 = (double)g0[plane + o] - q1;
    const double e = sqrt(d0 * d0 + d1 * d1);
    const float v = vi[o];
    const bool bg = v == 0.f && dr[o] == 0.f;
    s[0] += e;
    if (v == 1.f) s[1] += e;
    if (v != 0.f || bg) s[2] += e;   // np.logical_or(visible, bg)
    s[3] += (double)v;               // visible.sum()
    s[4] += (v != 0.f || bg) ? 1.0 : 0.0;
  }
  __shared__ double red[5][256];
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) partial[((long)b * gridDim.x + blockIdx.x) * 5 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void flow_epe_final_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ sums, int accumulate) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= 5) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += partial[((long)b * nblk + i) * 5 + k];
  sums[b * 5 + k] = (accumulate ? sums[b * 5 + k] : 0.0) + s;
}

}  // namespace dim

using namespace dim;

extern "C" int dim_depth_to_flow(const float* depth_src, const float* depth_tgt, const float* KT, const float* Kinv9, int B, int H,
                                 int W, float* flow, float* valid, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(depth_src && depth_tgt && KT && Kinv9 && flow && valid, "null pointer");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (W % 4 == 0 && al16(depth_src) && al16(flow) && al16(valid))
    hipLaunchKernelGGL(depth_flow_quad_kernel, dim3(ceil_div((long)H * W / 4, 256), B), dim3(256), 0, as_stream(stream), depth_src, depth_tgt,
                       KT, Kinv9[0], Kinv9[1], Kinv9[2], Kinv9[3], Kinv9[4], Kinv9[5], H, W, flow, valid);
  else
    hipLaunchKernelGGL(depth_flow_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, as_stream(stream), depth_src, depth_tgt, KT,
                       Kinv9[0], Kinv9[1], Kinv9[2], Kinv9[3], Kinv9[4], Kinv9[5], H, W, flow, valid);
  return check_launch("depth_to_flow");
}

extern "C" long dim_flow_epe_workspace_bytes(int B) { return (long)B * kEpeBlocks * 5 * (long)sizeof(double); }

extern "C" int dim_flow_epe_sums(const float* flow_pred, const float* flow_gt, const float* visible, const float* depth_rendered, int B,
                                 int H, int W, void* workspace, double* sums, int accumulate, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(flow_pred && flow_gt && visible && depth_rendered && workspace && sums, "null pointer");
  DIM_REQUIRE(H > 0 && W > 0, "empty image");
  hipLaunchKernelGGL(flow_epe_partial_kernel, dim3(kEpeBlocks, B), dim3(256), 0, as_stream(stream), flow_pred, flow_gt, visible,
                     depth_rendered, (long)H * W, (double*)workspace);
  hipLaunchKernelGGL(flow_epe_final_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const double*)workspace, kEpeBlocks, sums,
                     accumulate);
  return check_launch("flow_epe_sums");
}
