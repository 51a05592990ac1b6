// Fused zoom: bbox reduction -> zoom_factor -> affine bilinear gather.
//
// Replaces the reference's Python custom ops that bounce through numpy and call
// mx.nd.GridGenerator + mx.nd.BilinearSampler per sample:
//   ZoomMask            deepim/operator_py/zoom_mask.py:29-134
//   ZoomImageWithFactor deepim/operator_py/zoom_image_with_factor.py:31-75
//   ZoomImage           deepim/operator_py/zoom_image.py:26-119
//   ZoomMaskWithFactor  deepim/operator_py/zoom_mask_with_factor.py:29-68
//   ZoomFlow            deepim/operator_py/zoom_flow.py:28-77
//   ZoomDepth           deepim/operator_py/zoom_depth.py:24-50
// The sampling grid is never materialised.  HBM-bound: every input plane is read once
// (neighbouring output pixels share source texels through L1/L2), every output written once.
#include "common.h"

namespace dim {

// ------------------------------------------------------------------ bbox reduction
// bbox[b] = {min_x, max_x, min_y, max_y} over pixels where pred(b,y,x) holds; empty -> {W,-1,H,-1}.
__global__ void bbox_init_kernel(int* bbox, int n, int H, int W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    bbox[4 * i + 0] = W;
    bbox[4 * i + 1] = -1;
    bbox[4 * i + 2] = H;
    bbox[4 * i + 3] = -1;
  }
}

// mode 0: plane > thr (C == 1).   mode 1: sum_c (x_c + mean_c) > thr (ZoomImage validity).
// One block per (row-group, sample); float4 loads; wave shuffle + LDS reduce; 4 atomics per block.
template <int MODE>
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ x, int C, int H, int W, float thr, float m0, float m1,
                                                   float m2, int rows_per_block, int* __restrict__ bbox) {
  const int b = blockIdx.y;
  const int y_begin = blockIdx.x * rows_per_block;
  const int y_end = min(H, y_begin + rows_per_block);
  const int W4 = W >> 2;
  int minx = W, maxx = -1, miny = H, maxy = -1;
  const long plane = (long)H * W;
  const float* base = x + (long)b * C * plane;
  const int total = (y_end - y_begin) * W4;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    int y = y_begin + i / W4;
    int x4 = (i - (i / W4) * W4) * 4;
    float4 v = *reinterpret_cast<const float4*>(base + (long)y * W + x4);
    if (MODE == 1) {
      float4 v1 = *reinterpret_cast<const float4*>(base + plane + (long)y * W + x4);
      float4 v2 = *reinterpret_cast<const float4*>(base + 2 * plane + (long)y * W + x4);
      v.x = ((v.x + m0) + (v1.x + m1)) + (v2.x + m2);
      v.y = ((v.y + m0) + (v1.y + m1)) + (v2.y + m2);
      v.z = ((v.z + m0) + (v1.z + m1)) + (v2.z + m2);
      v.w = ((v.w + m0) + (v1.w + m1)) + (v2.w + m2);
    }
    bool p0 = v.x > thr, p1 = v.y > thr, p2 = v.z > thr, p3 = v.w > thr;
    if (p0 | p1 | p2 | p3) {
      int lo = p0 ? 0 : (p1 ? 1 : (p2 ? 2 : 3));
      int hi = p3 ? 3 : (p2 ? 2 : (p1 ? 1 : 0));
      minx = min(minx, x4 + lo);
      maxx = max(maxx, x4 + hi);
      miny = min(miny, y);
      maxy = max(maxy, y);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    minx = min(minx, __shfl_down(minx, off, 64));
    maxx = max(maxx, __shfl_down(maxx, off, 64));
    miny = min(miny, __shfl_down(miny, off, 64));
    maxy = max(maxy, __shfl_down(maxy, off, 64));
  }
  __shared__ int red[4][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[wave][0] = minx; red[wave][1] = maxx; red[wave][2] = miny; red[wave][3] = maxy;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      minx = min(minx, red[w][0]); maxx = max(maxx, red[w][1]);
      miny = min(miny, red[w][2]); maxy = max(maxy, red[w][3]);
    }
    if (maxx >= 0) {
      atomicMin(&bbox[4 * b + 0], minx);
      atomicMax(&bbox[4 * b + 1], maxx);
      atomicMin(&bbox[4 * b + 2], miny);
      atomicMax(&bbox[4 * b + 3], maxy);
    }
  }
}

// ------------------------------------------------------------------ zoom window rule
// zoom_mask.py:50-117 (identical in zoom_image.py:41-100).  Host code there is float32 for
// K.t and float64 afterwards; restated the same way.  status[b] bit0: observed bbox empty
// (the reference raises ValueError from np.min of an empty array), bit1: rendered bbox empty
// (reference prints "NO POINT VALID IN MASK rendered" and falls back to the observed box).
__global__ void zoom_factor_kernel(const int* __restrict__ bbox_obs, const int* __restrict__ bbox_ren,
                                   const float* __restrict__ src_pose, float k00, float k01, float k02, float k10, float k11,
                                   float k12, float k20, float k21, float k22, int H, int W, int B,
                                   float* __restrict__ zoom_factor, int* __restrict__ status) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int* bo = bbox_obs + 4 * b;
  const int* br = bbox_ren + 4 * b;
  int st = 0;
  if (bo[1] < 0) {
    st |= 1;
    zoom_factor[4 * b + 0] = 1.f; zoom_factor[4 * b + 1] = 1.f; zoom_factor[4 * b + 2] = 0.f; zoom_factor[4 * b + 3] = 0.f;
    if (status) status[b] = st | ((br[1] < 0) ? 2 : 0);
    return;
  }
  double rsx = bo[0], rex = bo[1], rsy = bo[2], rey = bo[3];
  double rcx = (rsx + rex) * 0.5, rcy = (rsy + rey) * 0.5;
  const float* p = src_pose + 12 * b;
  float tx_ = p[3], ty_ = p[7], tz_ = p[11];
  float c0 = k00 * tx_ + k01 * ty_ + k02 * tz_;
  float c1 = k10 * tx_ + k11 * ty_ + k12 * tz_;
  float c2 = k20 * tx_ + k21 * ty_ + k22 * tz_;
  double dsx, dex, dsy, dey, zcx, zcy;
  if (br[1] < 0) {
    st |= 2;
    dsx = rsx; dex = rex; dsy = rsy; dey = rey; zcx = rcx; zcy = rcy;
  } else {
    dsx = br[0]; dex = br[1]; dsy = br[2]; dey = br[3];
    zcx = (double)(c0 / c2);
    zcy = (double)(c1 / c2);
  }
  double left = fmax(zcx - dsx, zcx - rsx);
  double right = fmax(dex - zcx, rex - zcx);
  double up = fmax(zcy - dsy, zcy - rsy);
  double down = fmax(rey - zcy, dey - zcy);
  double crop_h = fmax(fmax(0.75 * right, 0.75 * left), fmax(up, down)) * 1.4 * 2;
  double wx = crop_h / H;
  double tx = zcx / W * 2 - 1;
  double ty = zcy / H * 2 - 1;
  zoom_factor[4 * b + 0] = (float)wx;
  zoom_factor[4 * b + 1] = (float)wx;
  zoom_factor[4 * b + 2] = (float)tx;
  zoom_factor[4 * b + 3] = (float)ty;
  if (status) status[b] = st;
}

// ------------------------------------------------------------------ sampling
struct Affine {
  float wx, wy, tx, ty;
};

// forward: theta = zoom_factor.  inverse: zoom_flow.py:35-44 (float32 arithmetic on numpy scalars)
__device__ inline Affine load_affine(const float* zf, bool inverse, int H, int W) {
  Affine a;
  float wx_in = zf[0], wy_in = zf[1], tx_in = zf[2], ty_in = zf[3];
  if (!inverse) {
    a.wx = wx_in; a.wy = wy_in; a.tx = tx_in; a.ty = ty_in;
  } else {
    // numpy float32 scalars mixed with python floats/ints stay float32 (weak python scalars)
    a.wx = 1.f / wx_in;
    a.wy = 1.f / wy_in;
    float crop_w = wx_in * (float)W, crop_h = wy_in * (float)H;
    float cx = tx_in * 0.5f * (float)W + 0.5f * (float)W;
    float cy = ty_in * 0.5f * (float)H + 0.5f * (float)H;
    a.tx = ((float)W * 0.5f - cx) / crop_w * 2.f;
    a.ty = ((float)H * 0.5f - cy) / crop_h * 2.f;
  }
  return a;
}

struct Tap {
  int o00, o01, o10, o11;      // offsets inside a plane (clamped)
  bool v00, v01, v10, v11;     // corner validity (BilinearSampler zero padding)
  float wx0, wy0;              // top_left_x_w, top_left_y_w
};

// GridGenerator(affine) + BilinearSampler coordinate rule (MXNet 1.2 semantics, float32):
//   x_t = -1 + j*2/(W-1);  x_s = wx*x_t + tx;  x_real = (x_s + 1)*(W-1)/2;  zero outside.
// Every operation is individually rounded (__f*_rn: no FMA contraction) so coordinates and weights are
// bit-identical to a plain float32 evaluation: with 8-bit image data a 1-ulp coordinate change at x~600
// already moves a sample by ~1e-2, which would otherwise dominate the parity budget.
__device__ inline Tap make_tap(const Affine& a, int y, int x, int H, int W) {
  const float sx = (float)(2.0 / (double)(W - 1)), sy = (float)(2.0 / (double)(H - 1));
  float xt = __fadd_rn(-1.f, __fmul_rn((float)x, sx));
  float yt = __fadd_rn(-1.f, __fmul_rn((float)y, sy));
  float xs = __fadd_rn(__fmul_rn(a.wx, xt), a.tx);
  float ys = __fadd_rn(__fmul_rn(a.wy, yt), a.ty);
  float xr = __fdiv_rn(__fmul_rn(__fadd_rn(xs, 1.f), (float)(W - 1)), 2.f);
  float yr = __fdiv_rn(__fmul_rn(__fadd_rn(ys, 1.f), (float)(H - 1)), 2.f);
  float x0f = floorf(xr), y0f = floorf(yr);
  Tap t;
  t.wx0 = __fsub_rn(1.f, __fsub_rn(xr, x0f));
  t.wy0 = __fsub_rn(1.f, __fsub_rn(yr, y0f));
  // clamp before the int cast so absurd coordinates cannot overflow; those corners are invalid anyway
  int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)(W + 1));
  int y0 = (int)fminf(fmaxf(y0f, -2.f), (float)(H + 1));
  bool vx0 = x0 >= 0 && x0 <= W - 1, vx1 = x0 + 1 >= 0 && x0 + 1 <= W - 1;
  bool vy0 = y0 >= 0 && y0 <= H - 1, vy1 = y0 + 1 >= 0 && y0 + 1 <= H - 1;
  int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
  int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
  t.o00 = yc0 * W + xc0; t.o01 = yc0 * W + xc1; t.o10 = yc1 * W + xc0; t.o11 = yc1 * W + xc1;
  t.v00 = vy0 && vx0; t.v01 = vy0 && vx1; t.v10 = vy1 && vx0; t.v11 = vy1 && vx1;
  return t;
}

enum { PRE_NONE = 0, PRE_BIN02 = 1 };                        // mask > 0.2 -> {0,1} before sampling
enum { POST_NONE = 0, POST_ROUND = 1, POST_ROUND_M045 = 2 }; // mx.nd.round(x) / round(x - 0.45)

__device__ inline float mx_round(float v) { return copysignf(floorf(fabsf(v) + 0.5f), v); }

template <int PRE>
__device__ inline float fetch(const float* p, int o) {
  float v = p[o];
  if (PRE == PRE_BIN02) v = v > 0.2f ? 1.f : 0.f;
  return v;
}

// out = tl*wy0*wx0 + tr*wy0*(1-wx0) + bl*(1-wy0)*wx0 + br*(1-wy0)*(1-wx0), evaluated left to right with
// one rounding per operation, exactly as BilinearSampler's expression reads.
// `add` is the per-plane constant added before sampling and removed afterwards (pixel mean).
template <int PRE>
__device__ inline float sample(const float* plane, const Tap& t, float add) {
  // the offsets are clamped into the plane, so the four loads are unconditional (independent, all in flight together) and the
  // zero padding is a select afterwards; with the loads inside the conditionals every corner was a branch + its own vmcnt wait
  const float f00 = fetch<PRE>(plane, t.o00), f01 = fetch<PRE>(plane, t.o01), f10 = fetch<PRE>(plane, t.o10), f11 = fetch<PRE>(plane, t.o11);
  float tl = t.v00 ? __fadd_rn(f00, add) : 0.f;
  float tr = t.v01 ? __fadd_rn(f01, add) : 0.f;
  float bl = t.v10 ? __fadd_rn(f10, add) : 0.f;
  float br = t.v11 ? __fadd_rn(f11, add) : 0.f;
  const float wx1 = __fsub_rn(1.f, t.wx0), wy1 = __fsub_rn(1.f, t.wy0);
  float r = __fmul_rn(__fmul_rn(tl, t.wy0), t.wx0);
  r = __fadd_rn(r, __fmul_rn(__fmul_rn(tr, t.wy0), wx1));
  r = __fadd_rn(r, __fmul_rn(__fmul_rn(bl, wy1), t.wx0));
  r = __fadd_rn(r, __fmul_rn(__fmul_rn(br, wy1), wx1));
  return r;
}

// Generic NCHW plane sampler: y[b,c] = post( sample(pre(x[b,c] + add[c])) - add[c] ) * scale(b)
// scale_mode: 0 none, 1 divide by zoom_factor[b,0], 2 multiply by zoom_factor[b,0]   (ZoomFlow)
template <int PRE, int POST>
__global__ __launch_bounds__(256) void zoom_planes_kernel(const float* __restrict__ x, const float* __restrict__ zoom_factor,
                                                          float* __restrict__ y, int C, int H, int W, int inverse,
                                                          float add0, float add1, float add2, int scale_mode) {
  const int b = blockIdx.z;
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  const int py = blockIdx.y;
  if (px >= W) return;
  const Affine a = load_affine(zoom_factor + 4 * b, inverse != 0, H, W);
  const Tap t = make_tap(a, py, px, H, W);
  const long plane = (long)H * W;
  const float zwx = zoom_factor[4 * b];
  for (int c = 0; c < C; ++c) {
    float add = c == 0 ? add0 : (c == 1 ? add1 : add2);
    if (C > 3) add = 0.f;
    float v = sample<PRE>(x + ((long)b * C + c) * plane, t, add);
    v = __fsub_rn(v, add);
    if (POST == POST_ROUND) v = mx_round(v);
    if (POST == POST_ROUND_M045) v = mx_round(__fsub_rn(v, 0.45f));
    if (scale_mode == 1) v = __fdiv_rn(v, zwx);
    if (scale_mode == 2) v = __fmul_rn(v, zwx);
    y[((long)b * C + c) * plane + (long)py * W + px] = v;
  }
}

// Fused network input (the test/train graph's Concat, deepIM_flownet.py:53-60):
//   X[b, y, x, 0:3] = zoom(image_observed)/255, [3:6] = zoom(image_rendered)/255,
//   X[.., 6] = round(zoom(mask_observed)), X[.., 7] = round(zoom(bin02(mask_rendered)))   (NHWC, 8 ch)
// plus optional NCHW copies of the four zoomed tensors for callers that want the op outputs.
// MODE 0: the two masks (shipped graph); MODE 1: no masks (INPUT_MASK off: channels 6, 7 = 0, the first layer's weights for them are
// zero too); MODE 2: no masks, depth_observed / depth_rendered in their place: plain bilinear samples (ZoomDepth, zoom_depth.py:24-50)
// divided by 255 like the images (deepIM_flownet.py:33-51)
template <int MODE>
__global__ __launch_bounds__(256) void zoom_net_input_kernel(const float* __restrict__ img_obs, const float* __restrict__ img_ren,
                                                             const float* __restrict__ mask_obs, const float* __restrict__ mask_ren,
                                                             const float* __restrict__ zoom_factor, float* __restrict__ X,
                                                             int H, int W, float m0, float m1, float m2,
                                                             float* __restrict__ z_img_obs, float* __restrict__ z_img_ren,
                                                             float* __restrict__ z_mask_obs, float* __restrict__ z_mask_ren) {
  const int b = blockIdx.z;
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  const int py = blockIdx.y;
  if (px >= W) return;
  const Affine a = load_affine(zoom_factor + 4 * b, false, H, W);
  const Tap t = make_tap(a, py, px, H, W);
  const long plane = (long)H * W;
  if (MODE == 3) {   // the two mask lanes alone (second 8-lane group of the 10-channel first layer): [mask_obs, mask_ren, 0 x 6]
    const float mo = mx_round(sample<PRE_NONE>(mask_obs + (long)b * plane, t, 0.f));
    const float mr = mx_round(sample<PRE_BIN02>(mask_ren + (long)b * plane, t, 0.f));
    float4* dst = reinterpret_cast<float4*>(X + ((long)b * plane + (long)py * W + px) * 8);
    dst[0] = make_float4(mo, mr, 0.f, 0.f);
    dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float* io = img_obs + (long)b * 3 * plane;
  const float* ir = img_ren + (long)b * 3 * plane;
  float v[8];
  v[0] = __fsub_rn(sample<PRE_NONE>(io, t, m0), m0);
  v[1] = __fsub_rn(sample<PRE_NONE>(io + plane, t, m1), m1);
  v[2] = __fsub_rn(sample<PRE_NONE>(io + 2 * plane, t, m2), m2);
  v[3] = __fsub_rn(sample<PRE_NONE>(ir, t, m0), m0);
  v[4] = __fsub_rn(sample<PRE_NONE>(ir + plane, t, m1), m1);
  v[5] = __fsub_rn(sample<PRE_NONE>(ir + 2 * plane, t, m2), m2);
  if (MODE == 0) {
    v[6] = mx_round(sample<PRE_NONE>(mask_obs + (long)b * plane, t, 0.f));
    v[7] = mx_round(sample<PRE_BIN02>(mask_ren + (long)b * plane, t, 0.f));
  } else if (MODE == 2) {
    v[6] = __fdiv_rn(sample<PRE_NONE>(mask_obs + (long)b * plane, t, 0.f), 255.f);
    v[7] = __fdiv_rn(sample<PRE_NONE>(mask_ren + (long)b * plane, t, 0.f), 255.f);
  } else {
    v[6] = v[7] = 0.f;
  }
  const long o = (long)py * W + px;
  if (MODE == 0 && z_img_obs) {
    for (int c = 0; c < 3; ++c) {
      z_img_obs[((long)b * 3 + c) * plane + o] = v[c];
      z_img_ren[((long)b * 3 + c) * plane + o] = v[3 + c];
    }
    z_mask_obs[(long)b * plane + o] = v[6];
    z_mask_ren[(long)b * plane + o] = v[7];
  }
  float4 lo = make_float4(__fdiv_rn(v[0], 255.f), __fdiv_rn(v[1], 255.f), __fdiv_rn(v[2], 255.f), __fdiv_rn(v[3], 255.f));
  float4 hi = make_float4(__fdiv_rn(v[4], 255.f), __fdiv_rn(v[5], 255.f), v[6], v[7]);
  float4* dst = reinterpret_cast<float4*>(X + ((long)b * plane + o) * 8);
  dst[0] = lo;
  dst[1] = hi;
}

}  // namespace dim

using namespace dim;

extern "C" {

int dim_mask_bbox(const float* x, int B, int C, int H, int W, int mode, float thr, const float* means3, int* bbox, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(x && bbox, "null pointer");
  DIM_REQUIRE(W % 4 == 0, "W must be a multiple of 4 (got %d)", W);
  DIM_REQUIRE(mode == 0 || (mode == 1 && C == 3 && means3), "mode 0 (plane>thr, C=1) or mode 1 (sum of 3 planes + means)");
  DIM_REQUIRE(mode == 1 || C == 1, "mode 0 expects C == 1");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bbox_init_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, bbox, B, H, W);
  const int rows = 8;
  dim3 grid(ceil_div(H, rows), B);
  if (mode == 0)
    hipLaunchKernelGGL(bbox_kernel<0>, grid, dim3(256), 0, st, x, C, H, W, thr, 0.f, 0.f, 0.f, rows, bbox);
  else
    hipLaunchKernelGGL(bbox_kernel<1>, grid, dim3(256), 0, st, x, C, H, W, thr, means3[0], means3[1], means3[2], rows, bbox);
  return check_launch("mask_bbox");
}

int dim_zoom_factor(const int* bbox_observed, const int* bbox_rendered, const float* src_pose, const float* K9, int B, int H, int W,
                    float* zoom_factor, int* status, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(bbox_observed && bbox_rendered && src_pose && K9 && zoom_factor, "null pointer");
  hipLaunchKernelGGL(zoom_factor_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), bbox_observed, bbox_rendered,
                     src_pose, K9[0], K9[1], K9[2], K9[3], K9[4], K9[5], K9[6], K9[7], K9[8], H, W, B, zoom_factor, status);
  return check_launch("zoom_factor");
}

int dim_zoom_planes(const float* x, const float* zoom_factor, float* y, int B, int C, int H, int W, int inverse, int pre, int post,
                    const float* add3, int scale_mode, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(x && zoom_factor && y, "null pointer");
  DIM_REQUIRE(pre == PRE_NONE || pre == PRE_BIN02, "pre must be 0 or 1");
  DIM_REQUIRE(post >= 0 && post <= 2, "post must be 0..2");
  DIM_REQUIRE(!add3 || C <= 3, "per-plane add constants only for C <= 3");
  float a0 = add3 ? add3[0] : 0.f, a1 = add3 && C > 1 ? add3[1] : 0.f, a2 = add3 && C > 2 ? add3[2] : 0.f;
  dim3 grid(ceil_div(W, 256), H, B), block(256);
  hipStream_t st = as_stream(stream);
#define DIM_ZP(PRE, POST) \
  hipLaunchKernelGGL((zoom_planes_kernel<PRE, POST>), grid, block, 0, st, x, zoom_factor, y, C, H, W, inverse, a0, a1, a2, scale_mode)
  if (pre == 0 && post == 0) DIM_ZP(0, 0);
  else if (pre == 0 && post == 1) DIM_ZP(0, 1);
  else if (pre == 0 && post == 2) DIM_ZP(0, 2);
  else if (pre == 1 && post == 0) DIM_ZP(1, 0);
  else if (pre == 1 && post == 1) DIM_ZP(1, 1);
  else DIM_ZP(1, 2);
#undef DIM_ZP
  return check_launch("zoom_planes");
}

int dim_zoom_net_input(const float* image_observed, const float* image_rendered, const float* mask_observed,
                       const float* mask_rendered, const float* zoom_factor, float* X_nhwc8, int B, int H, int W,
                       const float* means3, float* z_image_observed, float* z_image_rendered, float* z_mask_observed,
                       float* z_mask_rendered, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(image_observed && image_rendered && mask_observed && mask_rendered && zoom_factor && X_nhwc8 && means3, "null pointer");
  bool any = z_image_observed || z_image_rendered || z_mask_observed || z_mask_rendered;
  bool all = z_image_observed && z_image_rendered && z_mask_observed && z_mask_rendered;
  DIM_REQUIRE(!any || all, "pass all four NCHW outputs or none");
  const int bx = (W % 256 != 0 && W % 128 == 0) ? 128 : 256;  // W = 640: 5 x 128 leaves no idle lanes (3 x 256 idles 17 %)
  dim3 grid(ceil_div(W, bx), H, B), block(bx);
  hipLaunchKernelGGL(zoom_net_input_kernel<0>, grid, block, 0, as_stream(stream), image_observed, image_rendered, mask_observed,
                     mask_rendered, zoom_factor, X_nhwc8, H, W, means3[0], means3[1], means3[2], z_image_observed,
                     z_image_rendered, z_mask_observed, z_mask_rendered);
  return check_launch("zoom_net_input");
}

int dim_zoom_net_input_ex(const float* image_observed, const float* image_rendered, const float* extra_observed, const float* extra_rendered,
                          const float* zoom_factor, float* X_nhwc8, int B, int H, int W, const float* means3, int mode, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(image_observed && image_rendered && zoom_factor && X_nhwc8 && means3, "null pointer");
  DIM_REQUIRE(mode >= 0 && mode <= 3, "mode 0 (masks), 1 (images only), 2 (depth planes) or 3 (mask lanes alone)");
  DIM_REQUIRE(mode == 1 || (extra_observed && extra_rendered), "modes 0, 2 and 3 need the two extra planes");
  const int bx = (W % 256 != 0 && W % 128 == 0) ? 128 : 256;
  dim3 grid(ceil_div(W, bx), H, B), block(bx);
  hipStream_t st = as_stream(stream);
#define DIM_ZNI(M)                                                                                                                    \
  hipLaunchKernelGGL(zoom_net_input_kernel<M>, grid, block, 0, st, image_observed, image_rendered, extra_observed, extra_rendered, \
                     zoom_factor, X_nhwc8, H, W, means3[0], means3[1], means3[2], (float*)nullptr, (float*)nullptr, (float*)nullptr, \
                     (float*)nullptr)
  if (mode == 0) DIM_ZNI(0);
  else if (mode == 1) DIM_ZNI(1);
  else if (mode == 2) DIM_ZNI(2);
  else DIM_ZNI(3);
#undef DIM_ZNI
  return check_launch("zoom_net_input_ex");
}

}  // extern "C"
