// Training-only kernels: loss gradients, head / decoder backward pieces, weight-layout converters, SGD.
//
// Reference semantics (deepim/symbols/deepIM_flownet.py get_loss :303-560, executor deepim/core/module.py:1205-1209):
//   flow loss   MakeLoss(w * (f_est - f/NORMALIZE_FLOW)^2, grad_scale = LW_FLOW/(480*640))          :344-352
//   PM loss     MakeLoss(w * |(P_est - P_obs)/NORMALIZE_3D_POINT|, grad_scale = LW_PM/NUM_3D_SAMPLE) :446-499 (type L1)
//   mask loss   LogisticRegressionOutput(x, y, grad_scale = LW_MASK)                                :531-536
// MXNet-internal conventions assumed (source not in /root/reference -> "parity unpinned"):
//   MakeLoss backward = grad_scale * d(loss)/dx (normalization 'null'); LogisticRegressionOutput backward =
//   grad_scale / num_output * (sigmoid(x) - y) with num_output = label.size / batch; gradients are SUMMED over the batch
//   (rescale_grad = 1.0, deepim/train.py:383); SGD: mom = momentum*mom - lr*(g + wd*w), w += mom, wd_mult = 0 for biases.
#include "common.h"

namespace dim {

// ---------------------------------------------------------------------------------------------- weight layout converters
// packed [chunk][Cout][32] (chunk = (32-channel slice, kh, kw) | first layer: 4 consecutive flat taps) -> OIHW.  Inverse of
// pack_conv_weight_kernel (conv.hip); used to bring wgrad's output into the flat MXNet-layout gradient bucket.
__global__ void unpack_conv_weight_kernel(const float* __restrict__ wp, float* __restrict__ w, int Cout, int CoutPad, int Cin, int KH, int KW,
                                          int cin8, float scale, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cout * Cin * KH * KW;
  if (idx >= total) return;
  int kw = (int)(idx % KW);
  long t = idx / KW;
  int kh = (int)(t % KH); t /= KH;
  int ci = (int)(t % Cin);
  int co = (int)(t / Cin);
  int kc, kin;
  if (cin8) {
    const int t = kh * KW + kw;  // flat tap: 4 taps x 8 channels per chunk
    kc = t >> 2;
    kin = (t & 3) * 8 + ci;
  } else {
    kc = (ci >> 5) * KH * KW + kh * KW + kw;
    kin = ci & 31;
  }
  float v = wp[((long)kc * CoutPad + co) * 32 + kin] * scale;
  w[idx] = accumulate ? w[idx] + v : v;
}

// fc packed [chunk = (32-channel slice, h, w)][Out][32] -> (Out, C*H*W) in MXNet's (c,h,w) flatten order
__global__ void unpack_fc_weight_kernel(const float* __restrict__ wp, float* __restrict__ w, int Out, int C, int H, int W) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Out * C * H * W;
  if (idx >= total) return;
  long k = idx % ((long)C * H * W);
  int o = (int)(idx / ((long)C * H * W));
  int c = (int)(k / (H * W));
  int hw = (int)(k % (H * W));
  long kc = (long)(c >> 5) * H * W + hw;
  w[idx] = wp[(kc * Out + o) * 32 + (c & 31)];
}

// fc dgrad weights: dX (B, (h,w,c)) = dz (B, Out) * W  as a 1x1 "convolution" with Cin' = Out, Cout' = C*H*W (NHWC order):
// wp[kc][n = (h,w,c)][kin] = W[o = kc*32+kin][c*H*W + h*W + w]
template <typename PT>
__global__ void pack_fc_dgrad_weight_kernel(const float* __restrict__ w, PT* __restrict__ wp, int Out, int C, int H, int W) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Out * C * H * W;
  if (idx >= total) return;
  int kin = (int)(idx % 32);
  long t = idx / 32;
  long n = t % ((long)C * H * W);
  int kc = (int)(t / ((long)C * H * W));
  int c = (int)(n % C);
  long hw = n / C;
  wp[idx] = (PT)w[(long)(kc * 32 + kin) * C * H * W + (long)c * H * W + hw];
}

// ---------------------------------------------------------------------------------------------- loss gradients
// flow: g = gs * w * 2 * (f_est - f/nf); also accumulates sum(w * (f_est - f/nf)^2) per block into loss_sum (metric)
__global__ __launch_bounds__(256) void flow_loss_grad_kernel(const float* __restrict__ f_est, const float* __restrict__ f_lab,
                                                             const float* __restrict__ wgt, float* __restrict__ grad, long n, float inv_nf,
                                                             float gs, float* __restrict__ loss_sum) {
  // grid-stride: at most 1024 workgroups, so at most 1024 adds meet on the one loss_sum address (same-line atomics retire one after
  // the other, ~10 ns each: one per wave cost 0.5 ms, one per workgroup of 1024 elements still 0.09 of this kernel's 0.13 ms)
  float local = 0.f;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    float4 a = *reinterpret_cast<const float4*>(f_est + i), b = *reinterpret_cast<const float4*>(f_lab + i),
           w = *reinterpret_cast<const float4*>(wgt + i);
    float4 d = make_float4(a.x - b.x * inv_nf, a.y - b.y * inv_nf, a.z - b.z * inv_nf, a.w - b.w * inv_nf);
    *reinterpret_cast<float4*>(grad + i) = make_float4(gs * w.x * 2.f * d.x, gs * w.y * 2.f * d.y, gs * w.z * 2.f * d.z, gs * w.w * 2.f * d.w);
    local += w.x * d.x * d.x + w.y * d.y * d.y + w.z * d.z * d.z + w.w * d.w * d.w;
  }
  if (loss_sum) {  // metric only; gradients never depend on it.  One atomic per workgroup
    __shared__ float red[4];
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
  }
}

// mask: x = pre-sigmoid logits; g = gs/num_output * (sigmoid(x) - y); prob output optional
__global__ __launch_bounds__(256) void logistic_grad_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ grad,
                                                            float* __restrict__ prob, long n, float gs_over_n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p = 1.f / (1.f + expf(-x[i]));
  if (prob) prob[i] = p;
  grad[i] = gs_over_n * (p - y[i]);
}

// The element losses get_loss offers besides |x| (deepIM_flownet.py:414-426, :460-487): value and derivative of
//   type 0  |x|        type 1  x^2        type 2  mx.sym.smooth_l1(x, scalar = s): 0.5 (s x)^2 if |x| < 1 / s^2, else |x| - 0.5 / s^2
__device__ __forceinline__ float elem_loss(int type, float x, float s, float* dfdx) {
  const float ax = fabsf(x), sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
  if (type == 1) { *dfdx = 2.f * x; return x * x; }
  if (type == 2) {
    const float s2 = s * s;
    if (ax < 1.f / s2) { *dfdx = s2 * x; return 0.5f * s2 * x * x; }
    *dfdx = sg;
    return ax - 0.5f / s2;
  }
  *dfdx = sg;
  return ax;
}

// point-matching loss of any of the three types (dim_pm_l1_grad is type 0): grad = gs * w * f'(d) / norm, d = (p_est - p_obs) / norm
__global__ __launch_bounds__(256) void pm_loss_grad_kernel(const float* __restrict__ p_est, const float* __restrict__ p_obs, const float* __restrict__ wgt,
                                    float* __restrict__ grad, long n, float inv_norm, float gs, int type, float sl1, float* __restrict__ loss_sum) {
  float local = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float df;
    const float v = elem_loss(type, (p_est[i] - p_obs[i]) * inv_norm, sl1, &df);
    grad[i] = gs * wgt[i] * df * inv_norm;
    local += wgt[i] * v;
  }
  if (loss_sum) {  // one add per workgroup, at most 128 workgroups (see flow_loss_grad_kernel)
    __shared__ float red[4];
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (red[0] + red[1]) + (red[2] + red[3]));
  }
}

// SE3_DIST_LOSS (deepIM_flownet.py:396-437), one thread per sample:
//   rot_loss   = 1 - (rot_gt . rot_est_norm)^2                      d / d rot_est_norm = -2 (rot_gt . rot_est_norm) rot_gt
//   trans_loss = f(zoom_trans_est - zoom_trans_gt)  (3 values)      zoom_trans_est = the raw output of the `trans` layer, recomputed
//                                                                    here from fc7 (3 x 256 multiply-adds) so the forward keeps its outputs
// MakeLoss(grad_scale = LW_ROT / LW_TRANS), no normalisation: the gradients are ADDED to d_rot_norm / d_ztrans (which hold the
// point-matching gradients, or zeros); loss_sums[0..1] accumulate the un-scaled sums for the Rot_L2Loss / Trans_L2Loss metrics.
__global__ void se3_dist_loss_grad_kernel(const float* __restrict__ rot_norm, const float* __restrict__ rot_gt, const float* __restrict__ fc7,
                                          const float* __restrict__ wt, const float* __restrict__ bt, const float* __restrict__ ztrans_gt,
                                          float* __restrict__ d_rot_norm, float* __restrict__ d_ztrans, int B, float lw_rot, float lw_trans,
                                          int trans_type, float sl1, float* __restrict__ loss_sums) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* q = rot_norm + 4 * b;
  const float* g = rot_gt + 4 * b;
  const float dot = g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3];
  for (int i = 0; i < 4; ++i) d_rot_norm[4 * b + i] += -2.f * lw_rot * dot * g[i];
  float tl = 0.f;
  for (int j = 0; j < 3; ++j) {
    float tz = bt[j];
    for (int k = 0; k < 256; ++k) tz = fmaf(fc7[(long)b * 256 + k], wt[j * 256 + k], tz);
    float df;
    tl += elem_loss(trans_type, tz - ztrans_gt[3 * b + j], sl1, &df);
    d_ztrans[3 * b + j] += lw_trans * df;
  }
  if (loss_sums) {
    atomicAdd(loss_sums, 1.f - dot * dot);
    atomicAdd(loss_sums + 1, tl);
  }
}

// ---------------------------------------------------------------------------------------------- pose head backward
// L2Normalization(instance, eps=1e-10) forward: y = x / sqrt(sum x^2 + eps)
__global__ void quat_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* q = x + 4 * b;
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + 1e-10f);
  for (int i = 0; i < 4; ++i) y[4 * b + i] = q[i] / n;
}

// one block per sample: d_rot_norm, d_trans -> d_rot (through L2Normalization), d_fc7 -> dz7 (LeakyReLU') -> d_fc6a -> dz6
__global__ __launch_bounds__(256) void pose_head_bwd_kernel(const float* __restrict__ fc6a, const float* __restrict__ fc7,
                                                            const float* __restrict__ rot_raw, const float* __restrict__ d_rot_norm,
                                                            const float* __restrict__ d_trans, const float* __restrict__ w7,
                                                            const float* __restrict__ wr, const float* __restrict__ wt,
                                                            float* __restrict__ d_rot, float* __restrict__ dz7, float* __restrict__ dz6) {
  const int b = blockIdx.x, t = threadIdx.x;
  __shared__ float s_dr[4], s_dt[3], s_dz7[256];
  if (t == 0) {
    const float* q = rot_raw + 4 * b;
    const float* g = d_rot_norm + 4 * b;
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + 1e-10f);
    float y[4], dot = 0.f;
    for (int i = 0; i < 4; ++i) { y[i] = q[i] / n; dot += y[i] * g[i]; }
    for (int i = 0; i < 4; ++i) { s_dr[i] = (g[i] - y[i] * dot) / n; d_rot[4 * b + i] = s_dr[i]; }
    for (int i = 0; i < 3; ++i) s_dt[i] = d_trans[3 * b + i];  // inverse ZoomTrans backward = identity (b_zoom_grad False)
  }
  __syncthreads();
  // d_fc7[t] = sum_o d_rot[o] Wr[o][t] + sum_o d_ztrans[o] Wt[o][t]
  float g7 = 0.f;
  for (int o = 0; o < 4; ++o) g7 = fmaf(s_dr[o], wr[o * 256 + t], g7);
  for (int o = 0; o < 3; ++o) g7 = fmaf(s_dt[o], wt[o * 256 + t], g7);
  g7 *= fc7[(long)b * 256 + t] > 0.f ? 1.f : 0.1f;
  s_dz7[t] = g7;
  dz7[(long)b * 256 + t] = g7;
  __syncthreads();
  // d_fc6a[t] = sum_o dz7[o] W7[o][t]
  float g6 = 0.f;
  for (int o = 0; o < 256; ++o) g6 = fmaf(s_dz7[o], w7[o * 256 + t], g6);
  g6 *= fc6a[(long)b * 256 + t] > 0.f ? 1.f : 0.1f;
  dz6[(long)b * 256 + t] = g6;
}

// dW[o][i] = sum_b A[b][o] * X[b][i];  db[o] = sum_b A[b][o]      (tiny fully-connected layers)
__global__ void fc_wgrad_kernel(const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ dW, float* __restrict__ db,
                                int B, int Out, int In) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Out * In) return;
  int i = (int)(idx % In), o = (int)(idx / In);
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    float a = A[(long)b * Out + o];
    s = fmaf(a, X[(long)b * In + i], s);
    sb += a;
  }
  dW[idx] = s;
  if (i == 0 && db) db[o] = sb;
}

// fc6's weight gradient straight in the MXNet layout: dW[o][c HW + q] = sum_b dz[b][o] x[b][q][c]   (x NHWC: B x HW pixels x C channels;
// MXNet flattens (c, h, w)).  B is the batch: 16 terms -- there is nothing for a matrix pipe to do, the job is writing 84 MB once.  The
// general path computed it as an 8 x 10 "convolution" into the packed layout (17.6 us) and converted that (58.8 us at 16 pairs).
// Workgroup = CB channels x 64 outputs: the x block sits in LDS transposed to [b][c][q] (q fastest, as the gradient rows want it), a
// thread keeps its CB HW / 256 columns' batch values in registers and walks the outputs; every store is a run of the row.
template <int CB, int BMAX, int OB, int VEC>
__global__ __launch_bounds__(256) void fc_wgrad_nhwc_kernel(const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ dW,
                                                           int B, int Out, int C, int HW) {
  typedef float vf __attribute__((ext_vector_type(VEC)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int J = CB * HW;                 // contiguous gradient columns of one output row in this block
  float* xs = smem;                      // [BMAX][J]
  float* dzs = smem + BMAX * J;          // [BMAX][OB]
  const int c0 = blockIdx.x * CB, o0 = blockIdx.y * OB;
  // global reads: CB contiguous channels of pixel (b, q); eight loads in flight per thread (one at a time: 40 L2 round trips in series
  // made the whole kernel 55-62 us whatever the store width)
  for (int i0 = threadIdx.x; i0 < BMAX * J; i0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + 256 * e;
      const int c = i % CB, q = (i / CB) % HW, b = i / (CB * HW);
      v[e] = (i < BMAX * J && b < B) ? x[((long)b * HW + q) * C + c0 + c] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + 256 * e;
      const int c = i % CB, q = (i / CB) % HW, b = i / (CB * HW);
      if (i < BMAX * J) xs[b * J + c * HW + q] = v[e];
    }
  }
  for (int i = threadIdx.x; i < BMAX * OB; i += 256) {
    const int o = i % OB, b = i / OB;
    dzs[i] = (b < B && o0 + o < Out) ? dz[(long)b * Out + o0 + o] : 0.f;
  }
  __syncthreads();
  constexpr int NJ = (CB * 80 / VEC + 255) / 256;   // column groups per thread (HW <= 80)
  const int JV = J / VEC;
  vf xv[NJ][BMAX];
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    const int j = threadIdx.x + 256 * k;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) xv[k][b] = j < JV ? *reinterpret_cast<const vf*>(xs + b * J + j * VEC) : (vf)(0.f);
  }
#pragma unroll 2
  for (int o = 0; o < OB; ++o) {
    if (o0 + o >= Out) break;
    float a[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) a[b] = dzs[b * OB + o];   // broadcast reads
    float* row = dW + (long)(o0 + o) * C * HW + (long)c0 * HW;
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      const int j = threadIdx.x + 256 * k;
      vf sacc = (vf)(0.f);
#pragma unroll
      for (int b = 0; b < BMAX; ++b) sacc += a[b] * xv[k][b];
      if (j < JV) *reinterpret_cast<vf*>(row + j * VEC) = sacc;
    }
  }
}

// ---------------------------------------------------------------------------------------------- head backward pieces
// backward of Deconvolution(k=32, s=16, group=C) + Crop(crop): dF[n,iy,ix,c] = sum_{ky,kx} dOut[n,c,16iy+ky-crop,16ix+kx-crop] wk[c,ky,kx]
// one wave per output element
__global__ __launch_bounds__(256) void upsample16_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ wk,
                                                             float* __restrict__ df, int N, int C, int h, int w, int OH, int OW, int crop,
                                                             float scale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long e = (long)blockIdx.x * 4 + wave;
  if (e >= (long)N * h * w * C) return;
  int c = (int)(e % C);
  long t = e / C;
  int ix = (int)(t % w); t /= w;
  int iy = (int)(t % h);
  int n = (int)(t / h);
  // all 16 + 16 loads first, from clamped addresses, then select: behind a per-iteration `if` every load was a round trip of its own
  // (42 + 24 us for the two heads at 16 x 30 x 40; the products and their order are unchanged)
  const int kx = lane & 31;
  const int ox = 16 * ix + kx - crop;
  const bool okx = (unsigned)ox < (unsigned)OW;
  const float* plane = dout + ((long)n * C + c) * OH * OW + (okx ? ox : 0);
  const float* wrow = wk + (long)c * 1024 + kx;
  float dv[16], wv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int ky = 2 * j + (lane >> 5);
    const int oy = 16 * iy + ky - crop;
    const bool ok = okx && (unsigned)oy < (unsigned)OH;
    const float v = plane[(long)(ok ? oy : 0) * OW];
    dv[j] = ok ? v : 0.f;
    wv[j] = wrow[ky * 32];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s = fmaf(dv[j], wv[j], s);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) df[e] = s * scale;
}

// small-Cout conv backward, data: dX[pix][ci] (+)= sum_{co,kh,kw} dY[pix - tap][co] * W[co][ci][kh][kw]   (stride 1)
// one thread per (pixel, 4 channels).  wt = the weight transposed to [tap][co][CinPad4] (ci contiguous: one float4 per (tap, co);
// reading the MXNet layout directly was a 9-float-stride gather per channel)
__global__ void small_cout_transpose_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int CinPad, int taps) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)taps * Cout * CinPad) return;
  int ci = (int)(idx % CinPad);
  long t = idx / CinPad;
  int co = (int)(t % Cout), tap = (int)(t / Cout);
  wt[idx] = ci < Cin ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
}

// Version 2: a thread keeps its 4 channels' K*K*COUT weights in registers and walks pixels.  Workgroup = a chunk of 64 consecutive
// pixels x 64 channel quads (lane = quad, the 4 waves take every 4th pixel); the dY window of the chunk sits in LDS (broadcast
// reads).  The first version -- one thread per (pixel, quad) -- re-loaded the 18 weight float4 per thread and was bound by the
// vector L1 (61 us for a 120 MB read-modify-write at 16 x 30 x 40 x 770).
template <int COUT, int K>
__global__ __launch_bounds__(256) void conv_small_cout_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wt,
                                                                    float* __restrict__ dx, int N, int H, int W, int Cin, int CinPad,
                                                                    int dx_cstride, int pad, int accumulate, int chunk_px) {
  extern __shared__ float s_dy[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long P = (long)N * H * W;
  const long p0 = (long)blockIdx.x * chunk_px, p1 = min(P, p0 + chunk_px);
  // dX[p] pairs with dY[p + (pad - kh) W + (pad - kw)]: window = pixels p0 - mlo .. p1 - 1 + mhi of dY
  const int mhi = pad * W + pad, mlo = (K - 1 - pad) * W + (K - 1 - pad);
  const int wn = (int)(p1 - p0) + mlo + mhi;
  for (int i = threadIdx.x; i < wn * COUT; i += 256) {
    const long pp = p0 - mlo + i / COUT;
    s_dy[i] = (pp >= 0 && pp < P) ? dy[pp * COUT + i % COUT] : 0.f;
  }
  __syncthreads();
  const int c0 = (blockIdx.y * 64 + lane) * 4;
  if (c0 >= CinPad) return;
  float4 wv[K * K][COUT];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int co = 0; co < COUT; ++co) wv[t][co] = *reinterpret_cast<const float4*>(wt + ((long)t * COUT + co) * CinPad + c0);
  const bool full = c0 + 3 < Cin;
  for (long p = p0 + wave; p < p1; p += 4) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    float* o = dx + p * dx_cstride + c0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (accumulate) {
      if (full) {
        acc = *reinterpret_cast<const float4*>(o);
      } else {
        acc.x = o[0];
        if (c0 + 1 < Cin) acc.y = o[1];
        if (c0 + 2 < Cin) acc.z = o[2];
      }
    }
    const int o0 = (int)(p - p0) + mlo;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
      const int kh = t / K, kw = t - kh * K;
      const int oy = y + pad - kh, ox = x + pad - kw;
      const bool ok = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
      const int oi = o0 + (pad - kh) * W + (pad - kw);
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float gl = s_dy[oi * COUT + co];
        const float g = ok ? gl : 0.f;
        s.x = fmaf(g, wv[t][co].x, s.x);
        s.y = fmaf(g, wv[t][co].y, s.y);
        s.z = fmaf(g, wv[t][co].z, s.z);
        s.w = fmaf(g, wv[t][co].w, s.w);
      }
    }
    acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
    if (full) {
      *reinterpret_cast<float4*>(o) = acc;
    } else {
      o[0] = acc.x;
      if (c0 + 1 < Cin) o[1] = acc.y;
      if (c0 + 2 < Cin) o[2] = acc.z;
    }
  }
}

// small-Cout conv backward, weights: dW[co][ci][kh][kw] = sum_pix dY[pix][co] * X[pix + tap][ci]; db[co] = sum_pix dY[pix][co]
//   = sum over X pixels q of X[q][ci] * dY[q - tap][co]: each X row is read ONCE per wave (coalesced over ci) and feeds
//   the KH*KW*Cout accumulators of the thread; the dY values of the q - tap neighbours are wave-uniform scalars.
// stage 1: grid (pixel chunks, ceil(Cin/256)); partial[chunk][co][tap][ci] (deterministic, no atomics);  stage 2 sums the chunks.
template <int COUT, int TAPS>
__global__ __launch_bounds__(64) void conv_small_cout_wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                           float* __restrict__ partial, float* __restrict__ partial_b,
                                                                           int N, int H, int W, int Cin, int CinP, int in_cstride, int KH, int KW,
                                                                           int pad, int chunk_px) {
  // one wave = 256 input channels, a lane = 4 adjacent channels (one 16-byte load per pixel and tap; the first version loaded one
  // dword per lane and was bound by the number of wave-loads: 2.8 M of them for the 770-channel head at 16 x 30 x 40)
  const int chunk = blockIdx.x;
  const int ci = (blockIdx.y * 64 + threadIdx.x) * 4;
  const long P = (long)N * H * W;
  const long p0 = (long)chunk * chunk_px, p1 = min(P, p0 + chunk_px);
  float4 acc[TAPS][COUT];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[t][c] = make_float4(0.f, 0.f, 0.f, 0.f);
  // dW[tap] pairs input pixel q with output pixel q - (tap - pad): walk the INPUT pixels of the chunk, one 16-byte load each, and feed
  // all TAPS x COUT accumulators from it; the dY values of the neighbours and the border tests are wave-uniform (scalar loads, selected
  // branch-free so that the loads of the following pixels are not held behind a branch).  The first version walked output pixels and
  // loaded X once per tap behind a border branch: 576 dependent L2 round trips per wave, 173 us for the 770-channel head.
  // the dY window of the chunk -- output pixels p0 - mhi .. p1 - 1 + mlo -- goes to LDS first (scalar loads per pixel and tap left
  // the wave waiting on the scalar cache five times per pixel: 0.55 us per pixel)
  extern __shared__ float s_dy[];
  const int mlo = pad * W + pad, mhi = (KH - 1 - pad) * W + (KW - 1 - pad);
  const int wn = (int)(p1 - p0) + mlo + mhi;
  for (int i = threadIdx.x; i < wn * COUT; i += 64) {
    const long pp = p0 - mhi + i / COUT;
    s_dy[i] = (pp >= 0 && pp < P) ? dy[pp * COUT + i % COUT] : 0.f;
  }
  __syncthreads();
  if (ci < CinP) {
    int xx = (int)(p0 % W), y = (int)((p0 / W) % H);
#pragma unroll 4
    for (long q = p0; q < p1; ++q) {
      const float4 xv = *reinterpret_cast<const float4*>(x + q * in_cstride + ci);
      const int o0 = (int)(q - p0) + mhi;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int kh = t / KW, kw = t - kh * KW;
        const int oy = y + pad - kh, ox = xx + pad - kw;
        const bool ok = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
        const int o = o0 + (pad - kh) * W + (pad - kw);
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          const float gl = s_dy[o * COUT + c];
          const float g = ok ? gl : 0.f;
          acc[t][c].x = fmaf(g, xv.x, acc[t][c].x);
          acc[t][c].y = fmaf(g, xv.y, acc[t][c].y);
          acc[t][c].z = fmaf(g, xv.z, acc[t][c].z);
          acc[t][c].w = fmaf(g, xv.w, acc[t][c].w);
        }
      }
      if (++xx == W) { xx = 0; if (++y == H) y = 0; }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int c = 0; c < COUT; ++c)
        *reinterpret_cast<float4*>(partial + (((long)chunk * COUT + c) * TAPS + t) * CinP + ci) = acc[t][c];
  }
  if (blockIdx.y == 0 && threadIdx.x < COUT) {
    float sb = 0.f;
    for (long p = p0; p < p1; ++p) sb += dy[p * COUT + threadIdx.x];
    partial_b[(long)chunk * COUT + threadIdx.x] = sb;
  }
}

// workgroup = 32 outputs x 8 chunk groups, four loads in flight per thread (the chunk count is in the hundreds: a serial loop per
// output took 100 us, 64 outputs x 4 groups without unrolling 40 us); fixed summation order: per group in chunk order, then a tree
__global__ __launch_bounds__(256) void conv_small_cout_wgrad_final_kernel(const float* __restrict__ partial, const float* __restrict__ partial_b,
                                                                          float* __restrict__ dw, float* __restrict__ db, int nchunk, int Cin,
                                                                          int CinP, int Cout, int KH, int KW) {
  const int ol = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long idx = (long)blockIdx.x * 32 + ol;
  const long per = (long)Cout * KH * KW * Cin, perP = (long)Cout * KH * KW * CinP;   // partial rows are CinP (multiple of 4) long
  __shared__ float red[8][32];
  float s = 0.f;
  if (idx < per) {
    const long pidx = (idx / Cin) * CinP + idx % Cin;
    int c = grp;
    for (; c + 24 < nchunk; c += 32) {
      const float v0 = partial[(long)c * perP + pidx], v1 = partial[(long)(c + 8) * perP + pidx], v2 = partial[(long)(c + 16) * perP + pidx],
                  v3 = partial[(long)(c + 24) * perP + pidx];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; c < nchunk; c += 8) s += partial[(long)c * perP + pidx];
  }
  red[grp][ol] = s;
  __syncthreads();
  if (grp == 0 && idx < per) {
    const int ci = (int)(idx % Cin);
    const long t = idx / Cin;
    const int tap = (int)(t % (KH * KW));
    const int co = (int)(t / (KH * KW));
    dw[((long)co * Cin + ci) * KH * KW + tap] =
        ((red[0][ol] + red[1][ol]) + (red[2][ol] + red[3][ol])) + ((red[4][ol] + red[5][ol]) + (red[6][ol] + red[7][ol]));
  }
  if (blockIdx.x == 0 && db && threadIdx.x < Cout) {
    float sb = 0.f;
    for (int c = 0; c < nchunk; ++c) sb += partial_b[(long)c * Cout + threadIdx.x];
    db[threadIdx.x] = sb;
  }
}

// tiny deconv (k4 s2, Cin,Cout <= 4) backward: dX, dW, db.  dy is a channel range of a concat buffer over the crop window.
__global__ void deconv_tiny_bwd_data_kernel(const float* __restrict__ dy, int dy_cstride, int dy_coff, const float* __restrict__ w,
                                            float* __restrict__ dx, int N, int H, int W, int Cin, int Cout, int OH, int OW, int crop) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * H * W * Cin) return;
  int ci = (int)(idx % Cin);
  long t = idx / Cin;
  int ix = (int)(t % W); t /= W;
  int iy = (int)(t % H);
  int n = (int)(t / H);
  float s = 0.f;
  for (int ky = 0; ky < 4; ++ky) {
    int oy = 2 * iy + ky - crop;
    if ((unsigned)oy >= (unsigned)OH) continue;
    for (int kx = 0; kx < 4; ++kx) {
      int ox = 2 * ix + kx - crop;
      if ((unsigned)ox >= (unsigned)OW) continue;
      const float* g = dy + ((long)(n * OH + oy) * OW + ox) * dy_cstride + dy_coff;
      for (int co = 0; co < Cout; ++co) s = fmaf(g[co], w[(((long)ci * Cout + co) * 4 + ky) * 4 + kx], s);
    }
  }
  dx[idx] = s;
}

__global__ void deconv_tiny_bwd_weight_kernel(const float* __restrict__ x, int x_cstride, const float* __restrict__ dy, int dy_cstride,
                                              int dy_coff, float* __restrict__ dw, float* __restrict__ db, int N, int H, int W, int Cin,
                                              int Cout, int OH, int OW, int crop) {
  // one block per (ci, co, ky, kx) [+ one extra set for db]; threads stride the input pixels
  const int e = blockIdx.x;
  const int kx = e % 4, ky = (e / 4) % 4, co = (e / 16) % Cout, ci = e / (16 * Cout);
  float s = 0.f, sb = 0.f;
  for (long p = threadIdx.x; p < (long)N * H * W; p += blockDim.x) {
    int ix = (int)(p % W);
    int iy = (int)((p / W) % H);
    int n = (int)(p / ((long)W * H));
    int oy = 2 * iy + ky - crop, ox = 2 * ix + kx - crop;
    if ((unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW)
      s = fmaf(x[p * x_cstride + ci], dy[((long)(n * OH + oy) * OW + ox) * dy_cstride + dy_coff + co], s);
  }
  if (ci == 0 && ky == 0 && kx == 0)
    for (long p = threadIdx.x; p < (long)N * OH * OW; p += blockDim.x) sb += dy[p * dy_cstride + dy_coff + co];
  __shared__ float red[256], redb[256];
  red[threadIdx.x] = s;
  redb[threadIdx.x] = sb;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { red[threadIdx.x] += red[threadIdx.x + off]; redb[threadIdx.x] += redb[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dw[e] = red[0];
    if (ci == 0 && ky == 0 && kx == 0 && db) db[co] = redb[0];
  }
}

// ---------------------------------------------------------------------------------------------- SGD
// mx.optimizer.SGD with momentum: mom = momentum*mom - lr*(rescale*g + wd*w); w += mom     (float4)
__global__ void sgd_momentum_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ mom, long n, float lr,
                                    float momentum, float wd, float rescale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m = momentum * mom[i] - lr * (rescale * g[i] + wd * w[i]);
  mom[i] = m;
  w[i] += m;
}

// mx.optimizer.Adam (adam_update): g' = rescale*g + wd*w; mean = b1*mean + (1-b1)*g'; var = b2*var + (1-b2)*g'^2;
// w -= lr_t * mean / (sqrt(var) + eps), with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller
__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ mean, float* __restrict__ var, long n,
                            float lr_t, float beta1, float beta2, float eps, float wd, float rescale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = rescale * g[i] + wd * w[i];
  float m = beta1 * mean[i] + (1.f - beta1) * gi;
  float v = beta2 * var[i] + (1.f - beta2) * gi * gi;
  mean[i] = m;
  var[i] = v;
  w[i] -= lr_t * m / (sqrtf(v) + eps);
}

}  // namespace dim

using namespace dim;

extern "C" {

// packed = the sum of nslab arrays slab_stride floats apart (a weight gradient's pixel-split slabs, added in slab order)
extern "C++" int dim::conv2d_unpack_weight_slabs(const float* w_packed, int nslab, long slab_stride, float* w_oihw, int Cout, int CoutPad, int Cin,
                                                 int KH, int KW, float scale, int accumulate, void* stream) {
  DIM_REQUIRE(w_packed && w_oihw, "null pointer");
  DIM_REQUIRE(CoutPad >= Cout, "CoutPad < Cout");
  long total = (long)Cout * Cin * KH * KW;
  const int T = KH * KW, G = (Cin % 32 == 0) ? wtile_group(CoutPad, T, Cin / 32) : 0;
  if (G) {  // the inverse of dim_conv2d_pack_weight's tiling
    WTileArgs a = {};
    a.src = w_packed; a.dst = w_oihw;
    a.G = G; a.Q = T; a.gmax = Cout; a.rmax = Cin; a.g_fast = 0; a.nj = 0;
    a.sg = (long)Cin * T; a.sr = T; a.rows_x = 32L * T; a.rows_y = (long)G * Cin * T;
    a.dq = (long)CoutPad * 32; a.packed_x = (long)T * CoutPad * 32;
    a.scale = scale; a.accumulate = accumulate;
    a.nslab = nslab; a.slab_stride = slab_stride;
    wtile_launch<false, float>(a, Cin / 32, CoutPad / G, as_stream(stream));
  } else {
    if (nslab > 1) {  // no tiling for this shape (the 8-lane first layer): sum into slab 0 first
      DIM_REQUIRE(slab_stride % 4 == 0, "slab stride must be a multiple of 4 floats");
      int rc = dim_splitk_reduce(w_packed, nullptr, const_cast<float*>(w_packed), slab_stride / 4, 4, nslab, 1.0f, stream);
      if (rc != DIM_OK) return rc;
    }
    hipLaunchKernelGGL(unpack_conv_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_packed, w_oihw, Cout, CoutPad,
                       Cin, KH, KW, Cin == 8, scale, accumulate);
  }
  return check_launch("unpack_conv_weight");
}

int dim_conv2d_unpack_weight(const float* w_packed, float* w_oihw, int Cout, int CoutPad, int Cin, int KH, int KW, float scale,
                             int accumulate, void* stream) {
  return conv2d_unpack_weight_slabs(w_packed, 1, 0, w_oihw, Cout, CoutPad, Cin, KH, KW, scale, accumulate, stream);
}

int dim_fc_unpack_weight(const float* w_packed, float* w_out_in, int Out, int C, int H, int W, void* stream) {
  DIM_REQUIRE(w_packed && w_out_in, "null pointer");
  long total = (long)Out * C * H * W;
  const int HW = H * W, G = C % 32 == 0 ? wtile_group(Out, HW, C / 32) : 0;
  if (G) {  // the inverse of dim_fc_pack_weight's tiling
    WTileArgs a = {};
    a.src = w_packed; a.dst = w_out_in;
    a.G = G; a.Q = HW; a.gmax = Out; a.rmax = C; a.g_fast = 0; a.nj = 0;
    a.sg = (long)C * HW; a.sr = HW; a.rows_x = 32L * HW; a.rows_y = (long)G * C * HW;
    a.dq = (long)Out * 32; a.packed_x = (long)HW * Out * 32;
    a.scale = 1.0f; a.accumulate = 0;
    wtile_launch<false, float>(a, C / 32, Out / G, as_stream(stream));
  } else {
    hipLaunchKernelGGL(unpack_fc_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_packed, w_out_in, Out, C, H, W);
  }
  return check_launch("unpack_fc_weight");
}

// workgroup (output slice ob, G channels): rows = w[ob * 32 + r][c][q], packed run q at ((ob * HW + q) * C + c) * 32
extern "C++" template <typename PT>
int fc_dgrad_pack_weight_any(const float* w_out_in, PT* w_packed, int Out, int C, int H, int W, void* stream) {
  DIM_REQUIRE(w_out_in && w_packed, "null pointer");
  DIM_REQUIRE(Out % 32 == 0, "Out must be a multiple of 32");
  long total = (long)Out * C * H * W;
  const int HW = H * W, G = wtile_group(C, HW, Out / 32);
  if (G) {
    WTileArgs a = {};
    a.src = w_out_in; a.dst = w_packed;
    a.G = G; a.Q = HW; a.gmax = C; a.rmax = Out; a.g_fast = 1; a.nj = 0;
    a.sg = HW; a.sr = (long)C * HW; a.rows_x = 32L * C * HW; a.rows_y = (long)G * HW;
    a.dq = (long)C * 32; a.packed_x = (long)HW * C * 32;
    wtile_launch<true, PT>(a, Out / 32, C / G, as_stream(stream));
  } else {
    hipLaunchKernelGGL((pack_fc_dgrad_weight_kernel<PT>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_out_in, w_packed, Out, C, H,
                       W);
  }
  return check_launch("pack_fc_dgrad_weight");
}

int dim_fc_dgrad_pack_weight(const float* w_out_in, float* w_packed, int Out, int C, int H, int W, void* stream) {
  return fc_dgrad_pack_weight_any(w_out_in, w_packed, Out, C, H, W, stream);
}

int dim_fc_dgrad_pack_weight_bf16(const float* w_out_in, void* w_packed_bf16, int Out, int C, int H, int W, void* stream) {
  return fc_dgrad_pack_weight_any(w_out_in, reinterpret_cast<__bf16*>(w_packed_bf16), Out, C, H, W, stream);
}

int dim_flow_loss_grad(const float* flow_est, const float* flow_label, const float* flow_weights, float* grad, long n, float normalize_flow,
                       float grad_scale, float* loss_sum, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(flow_est && flow_label && flow_weights && grad, "null pointer");
  DIM_REQUIRE(n % 4 == 0, "element count must be a multiple of 4");
  const long wgs = ceil_div(n / 4, 256);
  hipLaunchKernelGGL(flow_loss_grad_kernel, dim3((unsigned)(wgs < 1024 ? wgs : 1024)), dim3(256), 0, as_stream(stream), flow_est, flow_label, flow_weights,
                     grad, n, 1.0f / normalize_flow, grad_scale, loss_sum);
  return check_launch("flow_loss_grad");
}

int dim_logistic_grad(const float* logits, const float* label, float* grad, float* prob, long n, float grad_scale_over_num_output,
                      void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(logits && label && grad, "null pointer");
  hipLaunchKernelGGL(logistic_grad_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), logits, label, grad, prob, n,
                     grad_scale_over_num_output);
  return check_launch("logistic_grad");
}

int dim_pm_l1_grad(const float* p_est, const float* p_obs, const float* weights, float* grad, long n, float norm_term, float grad_scale,
                   float* loss_sum, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(p_est && p_obs && weights && grad, "null pointer");
  const long l1_wgs = ceil_div(n, 256);
  hipLaunchKernelGGL(pm_loss_grad_kernel, dim3((unsigned)(l1_wgs < 128 ? l1_wgs : 128)), dim3(256), 0, as_stream(stream), p_est, p_obs, weights, grad, n,
                     1.0f / norm_term, grad_scale, 0, 1.0f, loss_sum);
  return check_launch("pm_l1_grad");
}

int dim_pm_loss_grad(const float* p_est, const float* p_obs, const float* weights, float* grad, long n, float norm_term, float grad_scale,
                     int loss_type, float smooth_l1_scalar, float* loss_sum, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(p_est && p_obs && weights && grad, "null pointer");
  DIM_REQUIRE(loss_type >= 0 && loss_type <= 2 && (loss_type != 2 || smooth_l1_scalar > 0.f), "loss_type 0 L1 | 1 L2 | 2 smooth_L1 (scalar > 0)");
  const long pm_wgs = ceil_div(n, 256);
  hipLaunchKernelGGL(pm_loss_grad_kernel, dim3((unsigned)(pm_wgs < 128 ? pm_wgs : 128)), dim3(256), 0, as_stream(stream), p_est, p_obs, weights, grad, n,
                     1.0f / norm_term, grad_scale, loss_type, smooth_l1_scalar, loss_sum);
  return check_launch("pm_loss_grad");
}

int dim_se3_dist_loss_grad(const float* rot_est_norm, const float* rot_gt, const float* fc7, const float* trans_w, const float* trans_b,
                           const float* zoom_trans_gt, float* d_rot_norm, float* d_zoom_trans, int B, float lw_rot, float lw_trans,
                           int trans_loss_type, float smooth_l1_scalar, float* loss_sums2, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(rot_est_norm && rot_gt && fc7 && trans_w && trans_b && zoom_trans_gt && d_rot_norm && d_zoom_trans, "null pointer");
  DIM_REQUIRE(trans_loss_type >= 0 && trans_loss_type <= 2 && (trans_loss_type != 2 || smooth_l1_scalar > 0.f),
              "trans_loss_type 0 L1 | 1 L2 | 2 smooth_L1 (scalar > 0)");
  hipLaunchKernelGGL(se3_dist_loss_grad_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), rot_est_norm, rot_gt, fc7, trans_w,
                     trans_b, zoom_trans_gt, d_rot_norm, d_zoom_trans, B, lw_rot, lw_trans, trans_loss_type, smooth_l1_scalar, loss_sums2);
  return check_launch("se3_dist_loss_grad");
}

int dim_quat_normalize(const float* rot, float* rot_norm, int B, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(rot && rot_norm, "null pointer");
  hipLaunchKernelGGL(quat_normalize_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), rot, rot_norm, B);
  return check_launch("quat_normalize");
}

int dim_pose_head_bwd(const float* fc6a, const float* fc7, const float* rot_raw, const float* d_rot_norm, const float* d_trans,
                      const float* fc7_w, const float* rot_w, const float* trans_w, float* d_rot, float* dz7, float* dz6, int B,
                      void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(fc6a && fc7 && rot_raw && d_rot_norm && d_trans && fc7_w && rot_w && trans_w && d_rot && dz7 && dz6, "null pointer");
  hipLaunchKernelGGL(pose_head_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), fc6a, fc7, rot_raw, d_rot_norm, d_trans, fc7_w, rot_w,
                     trans_w, d_rot, dz7, dz6);
  return check_launch("pose_head_bwd");
}

int dim_fc_wgrad(const float* dz, const float* x, float* dW, float* db, int B, int Out, int In, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(dz && x && dW, "null pointer");
  hipLaunchKernelGGL(fc_wgrad_kernel, dim3(ceil_div((long)Out * In, 256)), dim3(256), 0, as_stream(stream), dz, x, dW, db, B, Out, In);
  return check_launch("fc_wgrad");
}

// dW (Out, C*H*W in MXNet's (c, h, w) order) = dz (B, Out)^T . x (B, H, W, C NHWC), batches of up to 32 (see fc_wgrad_nhwc_kernel)
int dim_fc_wgrad_nhwc(const float* dz, const float* x, float* dW, int B, int Out, int C, int H, int W, void* stream) {
  DIM_REQUIRE(dz && x && dW, "null pointer");
  const int HW = H * W;
  DIM_REQUIRE(B >= 1 && B <= 32, "batch %d: this entry is built for 1 .. 32 rows (use dim_conv2d_wgrad + dim_fc_unpack_weight beyond)", B);
  DIM_REQUIRE(C % 8 == 0 && HW <= 80 && HW >= 1, "C %% 8 == 0 and H * W <= 80 required");
  hipStream_t st = as_stream(stream);
#define DIM_FC_WGRAD(CBc, BMc, OBc, VECc)                                                                                               \
  {                                                                                                                                     \
    const size_t lds = (size_t)(BMc * CBc * HW + BMc * OBc) * 4;                                                                        \
    static bool attr = false;                                                                                                           \
    if (!attr) {                                                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_wgrad_nhwc_kernel<CBc, BMc, OBc, VECc>),                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)(BMc * CBc * 80 + BMc * OBc) * 4));  \
      if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));                        \
      attr = true;                                                                                                                      \
    }                                                                                                                                   \
    hipLaunchKernelGGL((fc_wgrad_nhwc_kernel<CBc, BMc, OBc, VECc>), dim3(C / CBc, ceil_div(Out, OBc)), dim3(256), lds, st, dz, x, dW, B, \
                       Out, C, HW);                                                                                                     \
  }
  // 8 channels x 32 outputs per workgroup (43 KB of LDS: three workgroups per CU, 1024 workgroups for fc6); 16-byte stores when the
  // map size allows.  History at fc6, 16 pairs: 16 channels x 64 outputs (86 KB, one workgroup per CU) 58 us; 8 x 32 with 4-byte
  // stores 55 us; the packed "convolution" + conversion it replaces: 86 us
  const bool v4 = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(dW) & 15) == 0;
  if (B <= 16) { if (v4) DIM_FC_WGRAD(8, 16, 64, 4) else DIM_FC_WGRAD(8, 16, 32, 1) }
  else { if (v4) DIM_FC_WGRAD(8, 32, 32, 4) else DIM_FC_WGRAD(8, 32, 32, 1) }
#undef DIM_FC_WGRAD
  return check_launch("fc_wgrad_nhwc");
}

int dim_upsample16_bwd(const float* dout_nchw, const float* w_c1_32_32, float* df_nhwc, int N, int C, int h, int w, int OH, int OW, int crop,
                       float scale, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(dout_nchw && w_c1_32_32 && df_nhwc, "null pointer");
  long total = (long)N * h * w * C;
  hipLaunchKernelGGL(upsample16_bwd_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, as_stream(stream), dout_nchw, w_c1_32_32, df_nhwc, N, C,
                     h, w, OH, OW, crop, scale);
  return check_launch("upsample16_bwd");
}

static const int kSmallCoutChunkPx = 64;  // output pixels per wave of the wgrad partial pass (300 x 4 waves at 16x30x40)

long dim_conv_small_cout_bwd_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW) {
  long nchunk = ceil_div((long)N * H * W, kSmallCoutChunkPx);
  long cin_pad = (Cin + 3) / 4 * 4;
  return nchunk * ((long)Cout * KH * KW * cin_pad + Cout) + 4 + (long)KH * KW * Cout * cin_pad;  // chunk partials + transposed weight
}

int dim_conv_small_cout_bwd(const float* x, const float* dy, const float* w_oihw, float* dx, float* dw_oihw, float* db, float* workspace,
                            int N, int H, int W, int Cin, int in_cstride, int dx_cstride, int Cout, int KH, int KW, int pad,
                            int accumulate_dx, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && dy && w_oihw && dw_oihw && workspace, "null pointer");
  DIM_REQUIRE(KH == 3 && KW == 3 && (Cout == 1 || Cout == 2), "small-Cout backward is built for 3x3 kernels with 1 or 2 output channels");
  hipStream_t st = as_stream(stream);
  const int nchunk = ceil_div((long)N * H * W, kSmallCoutChunkPx);
  const int CinPad = (Cin + 3) / 4 * 4;
  float* partial = workspace;
  float* partial_b = workspace + (long)nchunk * Cout * KH * KW * CinPad;
  float* wt = workspace + (((long)nchunk * Cout * KH * KW * CinPad + (long)nchunk * Cout + 3) / 4) * 4;  // 16-byte aligned
  DIM_REQUIRE(in_cstride % 4 == 0 && in_cstride >= CinPad && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "small-Cout backward reads x 16 bytes at a time: in_cstride %% 4 == 0, >= Cin rounded up to 4, x and workspace 16-byte aligned");
  if (dx) {
    DIM_REQUIRE(dx_cstride % 4 == 0, "dx_cstride must be a multiple of 4");
    long wtot = (long)KH * KW * Cout * CinPad;
    hipLaunchKernelGGL(small_cout_transpose_weight_kernel, dim3(ceil_div(wtot, 256)), dim3(256), 0, st, w_oihw, wt, Cout, Cin, CinPad, KH * KW);
  }
  dim3 grid(nchunk, ceil_div(CinPad, 256));
  const size_t dy_lds = (size_t)(kSmallCoutChunkPx + (KH - 1) * W + (KW - 1)) * Cout * 4;  // the chunk's dY window
  DIM_REQUIRE(dy_lds <= 65536, "small-Cout backward: image too wide for the dY window in LDS (W = %d)", W);
  if (dx) {
    DIM_REQUIRE((reinterpret_cast<uintptr_t>(dx) & 15) == 0, "dx must be 16-byte aligned");
    if (Cout == 2)
      hipLaunchKernelGGL((conv_small_cout_dgrad_kernel<2, 3>), grid, dim3(256), dy_lds, st, dy, wt, dx, N, H, W, Cin, CinPad, dx_cstride, pad,
                         accumulate_dx, kSmallCoutChunkPx);
    else
      hipLaunchKernelGGL((conv_small_cout_dgrad_kernel<1, 3>), grid, dim3(256), dy_lds, st, dy, wt, dx, N, H, W, Cin, CinPad, dx_cstride, pad,
                         accumulate_dx, kSmallCoutChunkPx);
  }
  if (Cout == 2)
    hipLaunchKernelGGL((conv_small_cout_wgrad_partial_kernel<2, 9>), grid, dim3(64), dy_lds, st, x, dy, partial, partial_b, N, H, W, Cin, CinPad,
                       in_cstride, KH, KW, pad, kSmallCoutChunkPx);
  else
    hipLaunchKernelGGL((conv_small_cout_wgrad_partial_kernel<1, 9>), grid, dim3(64), dy_lds, st, x, dy, partial, partial_b, N, H, W, Cin, CinPad,
                       in_cstride, KH, KW, pad, kSmallCoutChunkPx);
  long per = (long)Cout * KH * KW * Cin;
  hipLaunchKernelGGL(conv_small_cout_wgrad_final_kernel, dim3(ceil_div(per, 32)), dim3(256), 0, st, partial, partial_b, dw_oihw, db,
                     nchunk, Cin, CinPad, Cout, KH, KW);
  return check_launch("conv_small_cout_bwd");
}

int dim_deconv4x4s2_tiny_bwd(const float* x, int x_cstride, const float* dy, int dy_cstride, int dy_coff, const float* w_iohw, float* dx,
                             float* dw_iohw, float* db, int N, int H, int W, int Cin, int Cout, int OH, int OW, int crop, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && dy && w_iohw && dw_iohw, "null pointer");
  hipStream_t st = as_stream(stream);
  if (dx) {
    long total = (long)N * H * W * Cin;
    hipLaunchKernelGGL(deconv_tiny_bwd_data_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, dy, dy_cstride, dy_coff, w_iohw, dx, N, H,
                       W, Cin, Cout, OH, OW, crop);
  }
  hipLaunchKernelGGL(deconv_tiny_bwd_weight_kernel, dim3(Cin * Cout * 16), dim3(256), 0, st, x, x_cstride, dy, dy_cstride, dy_coff, dw_iohw,
                     db, N, H, W, Cin, Cout, OH, OW, crop);
  return check_launch("deconv_tiny_bwd");
}

int dim_sgd_momentum(float* w, const float* grad, float* mom, long n, float lr, float momentum, float wd, float rescale_grad, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(w && grad && mom, "null pointer");
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), w, grad, mom, n, lr, momentum, wd,
                     rescale_grad);
  return check_launch("sgd_momentum");
}

int dim_adam(float* w, const float* grad, float* mean, float* var, long n, float lr_t, float beta1, float beta2, float epsilon, float wd,
             float rescale_grad, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(w && grad && mean && var, "null pointer");
  hipLaunchKernelGGL(adam_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), w, grad, mean, var, n, lr_t, beta1, beta2, epsilon,
                     wd, rescale_grad);
  return check_launch("adam");
}

}  // extern "C"
