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


// ---------------------------------------------------------------------------------------------------------------------------------
// General form for training AND test batches (lib/pair_matching/data_pair.py:22-72, :144-265 via lib/utils/image.py:65-553): every
// input that is absent (null) switches its outputs off.  Per pixel:
//   image_observed[c] = (use_bg && label == 0 ? bg : obs)[2 - c] - PIXEL_MEANS[2 - c]      image.py:107-175 (VOC background behind the
//                                                                                          object), :709-720 transform()
//   image_rendered[c] = ren[2 - c] - PIXEL_MEANS[2 - c]
//   d = depth_ren / DEPTH_FACTOR;  depth_rendered = d;  mask_rendered = d > 0.2 ? 1 : d;   image.py:478-488;  bbox_ren of d > 0.2
//   depth_a_out = depth_a / DEPTH_FACTOR, depth_b_out likewise                             image.py:186-233 (depth_gt_observed, depth_observed)
//   mask_label = (label == mask_idx) ? 1 : 0;  label_raw = (float)label;  bbox_label of mask_label     image.py:292-318, :385-440
__global__ __launch_bounds__(256) void pair_blobs_from_raw_kernel(
    const unsigned char* __restrict__ obs_bgr, const unsigned char* __restrict__ bg_bgr, const int* __restrict__ use_bg,
    const unsigned char* __restrict__ ren_bgr, const unsigned short* __restrict__ depth_ren, const unsigned short* __restrict__ depth_a,
    const unsigned short* __restrict__ depth_b, const unsigned char* __restrict__ label, const int* __restrict__ mask_idx, int H, int W,
    float depth_factor, float mb, float mg, float mr, float thr, float* __restrict__ image_observed, float* __restrict__ image_rendered,
    float* __restrict__ mask_rendered, float* __restrict__ depth_rendered, float* __restrict__ depth_a_out, float* __restrict__ depth_b_out,
    float* __restrict__ mask_label, float* __restrict__ label_raw, int* __restrict__ bbox_ren, int* __restrict__ bbox_label) {
  const int b = blockIdx.y;
  const int plane = H * W;
  const int pix = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  const bool live = pix < plane;
  float d[4] = {0.f, 0.f, 0.f, 0.f};
  bool on[4] = {false, false, false, false};   // label == mask_idx
  if (live) {
    const long o = (long)b * plane + pix;
    const float means[3] = {mb, mg, mr};
    unsigned lab[4] = {0u, 0u, 0u, 0u};
    if (label) {
      const unsigned raw = *reinterpret_cast<const unsigned*>(label + o);
#pragma unroll
      for (int k = 0; k < 4; ++k) lab[k] = (raw >> (8 * k)) & 0xFFu;
      const unsigned idx = mask_idx ? (unsigned)mask_idx[b] : 1u;
#pragma unroll
      for (int k = 0; k < 4; ++k) on[k] = lab[k] == idx;
      if (mask_label)
        *reinterpret_cast<float4*>(mask_label + o) = make_float4(on[0] ? 1.f : 0.f, on[1] ? 1.f : 0.f, on[2] ? 1.f : 0.f, on[3] ? 1.f : 0.f);
      if (label_raw) *reinterpret_cast<float4*>(label_raw + o) = make_float4((float)lab[0], (float)lab[1], (float)lab[2], (float)lab[3]);
    }
    const bool paste = bg_bgr && (!use_bg || use_bg[b] != 0);
#pragma unroll
    for (int img = 0; img < 2; ++img) {
      const unsigned char* src = img == 0 ? obs_bgr : ren_bgr;
      float* dst = img == 0 ? image_observed : image_rendered;
      if (!src || !dst) continue;
      const uint3 raw = *reinterpret_cast<const uint3*>(src + o * 3);  // 12 bytes = 4 pixels x (B, G, R)
      unsigned w[3] = {raw.x, raw.y, raw.z};
      float v[4][3];
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k / 3][k % 3] = (float)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
      if (img == 0 && paste) {
        const uint3 braw = *reinterpret_cast<const uint3*>(bg_bgr + o * 3);
        const unsigned bw[3] = {braw.x, braw.y, braw.z};
#pragma unroll
        for (int k = 0; k < 12; ++k)
          if (lab[k / 3] == 0u) v[k / 3][k % 3] = (float)((bw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)  // plane c holds BGR channel 2 - c
        *reinterpret_cast<float4*>(dst + ((long)b * 3 + c) * plane + pix) =
            make_float4(v[0][2 - c] - means[2 - c], v[1][2 - c] - means[2 - c], v[2][2 - c] - means[2 - c], v[3][2 - c] - means[2 - c]);
    }
    if (depth_ren) {
      const uint2 raw = *reinterpret_cast<const uint2*>(depth_ren + o);
      d[0] = __fdiv_rn((float)(raw.x & 0xFFFFu), depth_factor);  // true divisions, as `depth / DEPTH_FACTOR` on the host
      d[1] = __fdiv_rn((float)(raw.x >> 16), depth_factor);
      d[2] = __fdiv_rn((float)(raw.y & 0xFFFFu), depth_factor);
      d[3] = __fdiv_rn((float)(raw.y >> 16), depth_factor);
      if (depth_rendered) *reinterpret_cast<float4*>(depth_rendered + o) = make_float4(d[0], d[1], d[2], d[3]);
      if (mask_rendered)
        *reinterpret_cast<float4*>(mask_rendered + o) = make_float4(d[0] > thr ? 1.f : d[0], d[1] > thr ? 1.f : d[1], d[2] > thr ? 1.f : d[2],
                                                                    d[3] > thr ? 1.f : d[3]);
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const unsigned short* src = which == 0 ? depth_a : depth_b;
      float* dst = which == 0 ? depth_a_out : depth_b_out;
      if (!src || !dst) continue;
      const uint2 raw = *reinterpret_cast<const uint2*>(src + o);
      *reinterpret_cast<float4*>(dst + o) = make_float4(__fdiv_rn((float)(raw.x & 0xFFFFu), depth_factor), __fdiv_rn((float)(raw.x >> 16), depth_factor),
                                                        __fdiv_rn((float)(raw.y & 0xFFFFu), depth_factor), __fdiv_rn((float)(raw.y >> 16), depth_factor));
    }
  }
  const int y = pix / W, x0 = pix - y * W;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    int* bbox = which == 0 ? bbox_ren : bbox_label;
    if (!bbox) continue;
    int lo = 0x7FFFFFFF, hi = -1, ylo = 0x7FFFFFFF, yhi = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (live && (which == 0 ? d[k] > thr : on[k])) {
        lo = min(lo, x0 + k);
        hi = max(hi, x0 + k);
        ylo = yhi = y;
      }
    if (__ballot(hi >= 0) == 0) continue;  // wave-uniform
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
}

// lib/utils/mask_dilate.py:10-55 with the random draws made by the caller: thick[b] = {t_down, t_up, t_right, t_left}, 0 = that side
// stays put.  out = in + (number of sides whose displaced boundary hits the pixel), clamped at 1 -- the reference's arithmetic, which
// also turns the raw label values of TRAIN.INIT_MASK 'mask_gt' into 1.
__global__ __launch_bounds__(256) void mask_dilate_kernel(const float* __restrict__ in, const int* __restrict__ thick, int H, int W,
                                                          float* __restrict__ out) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  const float* m = in + (long)b * H * W;
  const int t0 = thick[4 * b], t1 = thick[4 * b + 1], t2 = thick[4 * b + 2], t3 = thick[4 * b + 3];
  const float v = m[(long)y * W + x];
  float acc = v;
  if (v == 0.f) {
    if (t0 > 0 && y >= t0 && m[(long)(y - t0) * W + x] != 0.f) acc += 1.f;
    if (t1 > 0 && y + t1 < H && m[(long)(y + t1) * W + x] != 0.f) acc += 1.f;
    if (t2 > 0 && x >= t2 && m[(long)y * W + x - t2] != 0.f) acc += 1.f;
    if (t3 > 0 && x + t3 < W && m[(long)y * W + x + t3] != 0.f) acc += 1.f;
  }
  out[(long)b * H * W + (long)y * W + x] = acc > 1.f ? 1.f : acc;
}

// First-iteration flow labels: calc_flow of lib/pair_matching/flow.py:12-81 (numpy, float64 per pixel; NOT the predicate of the CUDA
// kernel that re-labels later iterations, csrc/flow.hip) + the weights of image.py:531-545.
//   X = d K^-1 [u, v, 1]  (float64; Kinv64 = inv(K) in float64);  Xp = P [X; 1], P = K se3_mul(tgt, se3_inverse(src)) formed on the
//   host as the reference forms it (float32 se3 helpers) and handed over as float64;  pz = Xp.z + 1e-15;  (pw, ph) = Xp.xy / pz
//   visible = d != 0 and rint(pw, ph) inside and |d_tgt[rint] - pz| < 3e-3 and |d_tgt[rint]| > 1e-10
//   flow = (ph - v, pw - u) ["[h, w]" representation] or (pw - u, ph - v) [standard_rep], 0 where not visible
//   weights: 0 all ones, 1 visible, 2 (d == 0) or visible; written to both channels
struct FlowK {
  double kinv[9];
};
__global__ __launch_bounds__(256) void calc_flow_labels_kernel(const float* __restrict__ depth_src, const float* __restrict__ depth_tgt,
                                                               const double* __restrict__ P12, FlowK kk, int H, int W, double thresh,
                                                               int standard_rep, int weight_type, float* __restrict__ flow,
                                                               float* __restrict__ weights) {
  const int b = blockIdx.z, v = blockIdx.y;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= W) return;
  const long plane = (long)H * W;
  const long o = (long)v * W + u;
  const float ds = depth_src[(long)b * plane + o];
  const double d = (double)ds;
  const double rx = kk.kinv[0] * u + kk.kinv[1] * v + kk.kinv[2], ry = kk.kinv[3] * u + kk.kinv[4] * v + kk.kinv[5],
               rz = kk.kinv[6] * u + kk.kinv[7] * v + kk.kinv[8];
  const double X = d * rx, Y = d * ry, Z = d * rz;
  const double* P = P12 + 12 * b;
  const double xp = P[0] * X + P[1] * Y + P[2] * Z + P[3];
  const double yp = P[4] * X + P[5] * Y + P[6] * Z + P[7];
  const double pz = (P[8] * X + P[9] * Y + P[10] * Z + P[11]) + 1e-15;
  const double pw = xp / pz, ph = yp / pz;
  bool vis = false;
  if (ds != 0.f) {
    const double cw = rint(pw), ch = rint(ph);   // np.round: half to even
    if (cw >= 0.0 && cw < (double)W && ch >= 0.0 && ch < (double)H) {
      const double dt = (double)depth_tgt[(long)b * plane + (long)ch * W + (long)cw];
      vis = fabs(dt - pz) < thresh && fabs(dt) > 1e-10;
    }
  }
  const float fw = vis ? (float)(pw - (double)u) : 0.f, fh = vis ? (float)(ph - (double)v) : 0.f;
  flow[((long)b * 2 + 0) * plane + o] = standard_rep ? fw : fh;
  flow[((long)b * 2 + 1) * plane + o] = standard_rep ? fh : fw;
  if (weights) {
    const float wv = weight_type == 0 ? 1.f : (weight_type == 1 ? (vis ? 1.f : 0.f) : ((ds == 0.f || vis) ? 1.f : 0.f));
    weights[((long)b * 2 + 0) * plane + o] = wv;
    weights[((long)b * 2 + 1) * plane + o] = wv;
  }
}

// image.py:559-600: model[b, :, j] = table[table_off[b] + idx[b, j]] (idx < 0: zero-padded slot, weight 0), observed = R P + t
__global__ void point_clouds_kernel(const float* __restrict__ table, const int* __restrict__ table_off, const int* __restrict__ idx,
                                    const float* __restrict__ pose_obs, int n, float* __restrict__ model, float* __restrict__ weights,
                                    float* __restrict__ observed) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = idx[(long)b * n + j];
  float p[3] = {0.f, 0.f, 0.f};
  if (i >= 0) {
    const float* src = table + 3 * ((long)table_off[b] + i);
    p[0] = src[0]; p[1] = src[1]; p[2] = src[2];
  }
  const float* T = pose_obs + 12 * b;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const long o = ((long)b * 3 + r) * n + j;
    model[o] = p[r];
    weights[o] = i >= 0 ? 1.f : 0.f;
    observed[o] = (float)((double)T[4 * r] * p[0] + (double)T[4 * r + 1] * p[1] + (double)T[4 * r + 2] * p[2] + (double)T[4 * r + 3]);
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


int dim_pair_blobs_from_raw(const unsigned char* obs_bgr, const unsigned char* bg_bgr, const int* use_bg, const unsigned char* ren_bgr,
                            const unsigned short* depth_ren, const unsigned short* depth_a, const unsigned short* depth_b,
                            const unsigned char* label, const int* mask_idx, int B, int H, int W, float depth_factor,
                            const float* pixel_means_bgr3, float mask_thr, float* image_observed, float* image_rendered,
                            float* mask_rendered, float* depth_rendered, float* depth_a_out, float* depth_b_out, float* mask_label,
                            float* label_raw, int* bbox_ren, int* bbox_label, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(pixel_means_bgr3, "null pointer");
  DIM_REQUIRE(W % 4 == 0 && H > 0 && depth_factor > 0.f, "W must be a multiple of 4, depth_factor positive");
  DIM_REQUIRE(!(mask_rendered || depth_rendered || bbox_ren) || depth_ren, "mask_rendered / depth_rendered / bbox_ren need depth_ren");
  DIM_REQUIRE(!(mask_label || label_raw || bbox_label || bg_bgr) || label, "mask_label / label_raw / bbox_label / background paste need label");
  DIM_REQUIRE((!depth_a_out || depth_a) && (!depth_b_out || depth_b), "depth output without its input");
  auto al = [](const void* p, unsigned m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
  DIM_REQUIRE(al(obs_bgr, 3) && al(bg_bgr, 3) && al(ren_bgr, 3) && al(label, 3) && al(depth_ren, 7) && al(depth_a, 7) && al(depth_b, 7) &&
                  al(image_observed, 15) && al(image_rendered, 15) && al(mask_rendered, 15) && al(depth_rendered, 15) && al(depth_a_out, 15) &&
                  al(depth_b_out, 15) && al(mask_label, 15) && al(label_raw, 15),
              "pointer alignment: raw images / labels 4 bytes, raw depth 8 bytes, float planes 16 bytes");
  hipStream_t st = as_stream(stream);
  if (bbox_ren) hipLaunchKernelGGL(blobs_bbox_init_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, bbox_ren, B, H, W);
  if (bbox_label) hipLaunchKernelGGL(blobs_bbox_init_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, bbox_label, B, H, W);
  hipLaunchKernelGGL(pair_blobs_from_raw_kernel, dim3(ceil_div((long)H * W / 4, 256), B), dim3(256), 0, st, obs_bgr, bg_bgr, use_bg, ren_bgr,
                     depth_ren, depth_a, depth_b, label, mask_idx, H, W, depth_factor, pixel_means_bgr3[0], pixel_means_bgr3[1],
                     pixel_means_bgr3[2], mask_thr, image_observed, image_rendered, mask_rendered, depth_rendered, depth_a_out, depth_b_out,
                     mask_label, label_raw, bbox_ren, bbox_label);
  return check_launch("pair_blobs_from_raw");
}

int dim_mask_dilate(const float* mask_in, const int* thickness4, float* mask_out, int B, int H, int W, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(mask_in && thickness4 && mask_out && mask_in != mask_out, "null pointer, or in place (the rule reads displaced pixels of the ORIGINAL mask)");
  hipLaunchKernelGGL(mask_dilate_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, as_stream(stream), mask_in, thickness4, H, W, mask_out);
  return check_launch("mask_dilate");
}

int dim_calc_flow_labels(const float* depth_src, const float* depth_tgt, const double* P12, const double* Kinv9_f64, int B, int H, int W,
                         double thresh, int standard_rep, int weight_type, float* flow, float* flow_weights, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(depth_src && depth_tgt && P12 && Kinv9_f64 && flow, "null pointer");
  DIM_REQUIRE(weight_type >= 0 && weight_type <= 2, "weight_type: 0 all, 1 viz, 2 valid");
  FlowK kk;
  for (int i = 0; i < 9; ++i) kk.kinv[i] = Kinv9_f64[i];
  hipLaunchKernelGGL(calc_flow_labels_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, as_stream(stream), depth_src, depth_tgt, P12, kk, H,
                     W, thresh, standard_rep, weight_type, flow, flow_weights);
  return check_launch("calc_flow_labels");
}

int dim_point_clouds(const float* table, const int* table_off, const int* idx, const float* pose_observed, int B, int n, float* model,
                     float* weights, float* observed, void* stream) {
  if (B == 0 || n == 0) return DIM_OK;
  DIM_REQUIRE(table && table_off && idx && pose_observed && model && weights && observed, "null pointer");
  hipLaunchKernelGGL(point_clouds_kernel, dim3(ceil_div(n, 256), B), dim3(256), 0, as_stream(stream), table, table_off, idx, pose_observed, n,
                     model, weights, observed);
  return check_launch("point_clouds");
}

}  // extern "C"
