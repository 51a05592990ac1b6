// Device half of the data layer: the loader ships the pixels as they sit in the image files -- 8-bit BGR colour, 16-bit depth: 2.1 MB
// per 480x640 pair -- and this kernel builds the float blobs the network reads (9.8 MB per pair) on the GPU, instead of building them
// on the host and pushing 4.6x the bytes through PCIe.  Restates, per pixel:
//   image_* [c]   = bgr[2 - c] - PIXEL_MEANS[2 - c]                        lib/utils/image.py:709-720  transform()
//   mask_rendered = d > 0.2 ? 1 : d,  d = depth / DEPTH_FACTOR             lib/utils/image.py:478-488  (the depth itself below 0.2 m)
//   bbox          = {min_x, max_x, min_y, max_y} of d > 0.2                lib/utils/image.py:437-460  (TEST.INIT_MASK box_rendered;
//                   dim_box_mask then fills the end-exclusive rectangle)
#include "common.h"

namespace dim {

__global__ void blobs_bbox_init_kernel(int* bbox, int n, int H, int W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    bbox[4 * i + 0] = W; bbox[4 * i + 1] = -1; bbox[4 * i + 2] = H; bbox[4 * i + 3] = -1;
  }
}

// one thread per 4 consecutive pixels (W % 4 == 0): 12 + 12 + 8 bytes in, 7 float4 out
__global__ __launch_bounds__(256) void test_blobs_from_raw_kernel(const unsigned char* __restrict__ obs_bgr,
                                                                  const unsigned char* __restrict__ ren_bgr,
                                                                  const unsigned short* __restrict__ depth_ren, int H, int W,
                                                                  float depth_factor, float mb, float mg, float mr, float thr,
                                                                  float* __restrict__ image_observed, float* __restrict__ image_rendered,
                                                                  float* __restrict__ mask_rendered, int* __restrict__ bbox) {
  const int b = blockIdx.y;
  const int plane = H * W;
  const int pix = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  const bool live = pix < plane;
  float d[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const long o = (long)b * plane + pix;
    const float means[3] = {mb, mg, mr};
#pragma unroll
    for (int img = 0; img < 2; ++img) {
      const unsigned char* src = img == 0 ? obs_bgr : ren_bgr;
      float* dst = img == 0 ? image_observed : image_rendered;
      if (!src || !dst) continue;
      const uint3 raw = *reinterpret_cast<const uint3*>(src + o * 3);  // 12 bytes = 4 pixels x (B, G, R); o * 3 is a multiple of 12
      const unsigned w[3] = {raw.x, raw.y, raw.z};
      float v[4][3];
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k / 3][k % 3] = (float)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
#pragma unroll
      for (int c = 0; c < 3; ++c)  // plane c holds BGR channel 2 - c
        *reinterpret_cast<float4*>(dst + ((long)b * 3 + c) * plane + pix) =
            make_float4(v[0][2 - c] - means[2 - c], v[1][2 - c] - means[2 - c], v[2][2 - c] - means[2 - c], v[3][2 - c] - means[2 - c]);
    }
    if (depth_ren) {
      const uint2 raw = *reinterpret_cast<const uint2*>(depth_ren + o);
      d[0] = __fdiv_rn((float)(raw.x & 0xFFFFu), depth_factor);  // a true division, as `depth / DEPTH_FACTOR` on the host
      d[1] = __fdiv_rn((float)(raw.x >> 16), depth_factor);  // a true division, as `depth / DEPTH_FACTOR` on the host
      d[2] = __fdiv_rn((float)(raw.y & 0xFFFFu), depth_factor);  // a true division, as `depth / DEPTH_FACTOR` on the host
      d[3] = __fdiv_rn((float)(raw.y >> 16), depth_factor);  // a true division, as `depth / DEPTH_FACTOR` on the host
      if (mask_rendered)
        *reinterpret_cast<float4*>(mask_rendered + o) = make_float4(d[0] > thr ? 1.f : d[0], d[1] > thr ? 1.f : d[1], d[2] > thr ? 1.f : d[2],
                                                                    d[3] > thr ? 1.f : d[3]);
    }
  }
  if (!bbox) return;
  const int y = pix / W, x0 = pix - y * W;
  int lo = 0x7FFFFFFF, hi = -1, ylo = 0x7FFFFFFF, yhi = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (live && d[k] > thr) {
      lo = min(lo, x0 + k);
      hi = max(hi, x0 + k);
      ylo = yhi = y;
    }
  if (__ballot(hi >= 0) == 0) return;  // wave-uniform
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    lo = min(lo, __shfl_xor(lo, s));
    hi = max(hi, __shfl_xor(hi, s));
    ylo = min(ylo, __shfl_xor(ylo, s));
    yhi = max(yhi, __shfl_xor(yhi, s));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&bbox[4 * b + 0], lo);
    atomicMax(&bbox[4 * b + 1], hi);
    atomicMin(&bbox[4 * b + 2], ylo);
    atomicMax(&bbox[4 * b + 3], yhi);
  }
}

}  // namespace dim

using namespace dim;

extern "C" {

int dim_test_blobs_from_raw(const unsigned char* obs_bgr, const unsigned char* ren_bgr, const unsigned short* depth_rendered, int B, int H,
                            int W, float depth_factor, const float* pixel_means_bgr3, float mask_thr, float* image_observed,
                            float* image_rendered, float* mask_rendered, int* bbox, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(pixel_means_bgr3, "null pointer");
  DIM_REQUIRE(W % 4 == 0 && H > 0 && depth_factor > 0.f, "W must be a multiple of 4, depth_factor positive");
  DIM_REQUIRE(!(mask_rendered || bbox) || depth_rendered, "mask_rendered / bbox need depth_rendered");
  DIM_REQUIRE(((reinterpret_cast<uintptr_t>(obs_bgr) | reinterpret_cast<uintptr_t>(ren_bgr)) & 3) == 0 &&
                  (reinterpret_cast<uintptr_t>(depth_rendered) & 7) == 0 &&
                  ((reinterpret_cast<uintptr_t>(image_observed) | reinterpret_cast<uintptr_t>(image_rendered) |
                    reinterpret_cast<uintptr_t>(mask_rendered)) & 15) == 0,
              "pointer alignment: raw images 4 bytes, raw depth 8 bytes, float planes 16 bytes");
  hipStream_t st = as_stream(stream);
  if (bbox) hipLaunchKernelGGL(blobs_bbox_init_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, bbox, B, H, W);
  hipLaunchKernelGGL(test_blobs_from_raw_kernel, dim3(ceil_div((long)H * W / 4, 256), B), dim3(256), 0, st, obs_bgr, ren_bgr, depth_rendered,
                     H, W, depth_factor, pixel_means_bgr3[0], pixel_means_bgr3[1], pixel_means_bgr3[2], mask_thr, image_observed,
                     image_rendered, mask_rendered, bbox);
  return check_launch("test_blobs_from_raw");
}

}  // extern "C"
