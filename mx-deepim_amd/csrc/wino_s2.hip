// 3x3 / stride-2 / pad-1 convolutions (conv4, conv5 of deepim/symbols/deepIM_flownet.py:118-126, 143-151) in the Winograd domain
// through their phase images -- minimal filtering per phase.
//
//   y[i][j] = sum_{a,b} w[a][b] x[2i + a - 1][2j + b - 1]
// Per axis the even phase E[r] = x[2r] meets ONE tap (w1: F(4,1) is the identity on four points) and the odd phase O[r] = x[2r + 1]
// two (w0 on O[i - 1], w2 on O[i]: F(4,2), five points), so a 4 x 4 output tile costs 4x4 + 4x5 + 5x4 + 5x5 = 81 multiplies per channel
// pair against 144 direct (1.78x).  Every pixel of the tile's 9 x 9 input patch belongs to exactly one phase: the input transform reads
// 81 values and writes 81.  The 81 planes run through the persistent stream-K GEMM of the other Winograd layers (wino_gemm.hip, P = 81):
//   V [T][81][Cin]   U [81][Cin/32][Cout][32]   M [T][81][Cout]
// Plane order: ee 0..15 (a*4 + b), eo 16..35 (a*5 + beta), oe 36..55 (alpha*4 + b), oo 56..80 (alpha*5 + beta); a, b = even rows / columns
// of the tile, alpha, beta = transform points of the odd rows / columns.
// F(4,2) with points {0, 1, -1, 2, inf}, B^T scaled to small integers (the fractions live in G, applied once at pack time in f64):
//   B^T = [2 -1 -2 1 0; 0 2 1 -1 0; 0 -2 3 -1 0; 0 -1 0 1 0; 0 2 -1 -2 1]   on d[r] = O[i0 - 1 + r] = x[2 (i0 + r) - 1]
//   G   = [1/2 0; 1/2 1/2; 1/6 -1/6; 1/6 1/3; 0 1]                           on (w0, w2)
//   A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 0; 0 1 -1 8 1]
// tools/wino_s2_proto.py is the same algorithm in numpy: exact in f64, 1.4e-6 of max|y| in f32 on a conv4-shaped layer (direct f32: 3e-7).
#include <cstdlib>

#include "common.h"

namespace dim {

constexpr int kVec = 2;  // channels per thread of the transforms
typedef float vf __attribute__((ext_vector_type(kVec)));
constexpr int kPlanes = 81, kEO = 16, kOE = 36, kOO = 56;

#define DIM_S2_BT(O, D, S)                                  \
  {                                                         \
    O[0 * S] = 2.f * D[0] - D[1] - 2.f * D[2] + D[3];       \
    O[1 * S] = 2.f * D[1] + D[2] - D[3];                    \
    O[2 * S] = 3.f * D[2] - 2.f * D[1] - D[3];              \
    O[3 * S] = D[3] - D[1];                                 \
    O[4 * S] = 2.f * D[1] - D[2] - 2.f * D[3] + D[4];       \
  }
#define DIM_S2_AT(O, M, S)                                  \
  {                                                         \
    const vf s1 = M[1] + M[2], d1 = M[1] - M[2];            \
    O[0 * S] = M[0] + s1 + M[3];                            \
    O[1 * S] = d1 + 2.f * M[3];                             \
    O[2 * S] = s1 + 4.f * M[3];                             \
    O[3 * S] = d1 + 8.f * M[3] + M[4];                      \
  }

// thread = (tile t, kVec channels); blocks past nblk zero the M tiles that two workgroups of the following stream-K GEMM share
__global__ __launch_bounds__(256) void wino_s2_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W, int C,
                                                            int in_cstride, int th, int tw, FastDiv div_cq, FastDiv div_tw, FastDiv div_th,
                                                            unsigned nblk, WGemmArgs plan) {
#pragma clang fp contract(fast)
  if (blockIdx.x >= nblk) {
    wino_gemm_zero_tile(plan, (int)(blockIdx.x - nblk) + 1);
    return;
  }
  // XCD-contiguous numbering: neighbouring tiles share a row / column of their 9 x 9 patches
  const unsigned idx = (unsigned)wg_xcd_contiguous((int)blockIdx.x, (int)nblk) * 256u + threadIdx.x;
  const unsigned CQ = C / kVec;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const int y0 = 8 * (int)ty - 1, x0 = 8 * (int)tx - 1;   // input pixel of patch position (0, 0): patch row q = x row y0 + q
  const float* base = x + (long)n * H * W * in_cstride + cq * kVec;
  float* out = V + (long)t * kPlanes * C + cq * kVec;
  const long plane = C;
  // load from a clamped address, then select: a conditional load would compile to a branch with a wait per load
  auto ld = [&](int py, int px) -> vf {
    const int yy = y0 + py, xx = x0 + px;
    const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    const vf v = *reinterpret_cast<const vf*>(base + ((long)(ok ? yy : 0) * W + (ok ? xx : 0)) * in_cstride);
    return ok ? v : (vf)(0.f);
  };
  // patch rows / columns: odd phase d[r] at patch index 2r (r = 0..4), even phase e[r] at patch index 2r + 1 (r = 0..3)
  // ---- ee: copies
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) __builtin_nontemporal_store(ld(2 * a + 1, 2 * b + 1), reinterpret_cast<vf*>(out + (a * 4 + b) * plane));
  // ---- eo: B^T along x on the odd columns of every even row
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    vf d[5], o[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) d[q] = ld(2 * a + 1, 2 * q);
    DIM_S2_BT(o, d, 1)
#pragma unroll
    for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(o[q], reinterpret_cast<vf*>(out + (kEO + a * 5 + q) * plane));
  }
  // ---- oe: B^T along y on the odd rows of every even column
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    vf d[5], o[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) d[q] = ld(2 * q, 2 * b + 1);
    DIM_S2_BT(o, d, 1)
#pragma unroll
    for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(o[q], reinterpret_cast<vf*>(out + (kOE + q * 4 + b) * plane));
  }
  // ---- oo: both
  vf tmp[25];
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    vf d[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) d[q] = ld(2 * q, 2 * b);
    vf* o = tmp + b;
    DIM_S2_BT(o, d, 5)
  }
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    vf o[5];
    const vf* d = tmp + 5 * a;
    DIM_S2_BT(o, d, 1)
#pragma unroll
    for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(o[q], reinterpret_cast<vf*>(out + (kOO + a * 5 + q) * plane));
  }
}

// Y = M_ee + M_eo A + A^T M_oe + A^T M_oo A, + bias, LeakyReLU; thread = (tile t, kVec output channels)
__global__ __launch_bounds__(256) void wino_s2_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y,
                                                             int N, int Ho, int Wo, int C, int out_cstride, int out_coff, int th, int tw,
                                                             float slope, FastDiv div_cq, FastDiv div_tw, FastDiv div_th) {
#pragma clang fp contract(fast)
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const unsigned CQ = C / kVec;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const long plane = C;
  const unsigned co = cq * kVec;
  const float* in = M + (long)t * kPlanes * C + co;
  auto ld = [&](int p) -> vf { return __builtin_nontemporal_load(reinterpret_cast<const vf*>(in + p * plane)); };
  vf acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = ld(k);   // ee
#pragma unroll
  for (int a = 0; a < 4; ++a) {                  // eo: A^T along x
    vf m[5], o[4];
#pragma unroll
    for (int q = 0; q < 5; ++q) m[q] = ld(kEO + a * 5 + q);
    DIM_S2_AT(o, m, 1)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a * 4 + b] += o[b];
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {                  // oe: A^T along y
    vf m[5], o[4];
#pragma unroll
    for (int q = 0; q < 5; ++q) m[q] = ld(kOE + q * 4 + b);
    DIM_S2_AT(o, m, 1)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a * 4 + b] += o[a];
  }
  {                                              // oo: both
    vf rr[20];   // A^T m: [4 rows][5 columns]
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      vf m[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) m[q] = ld(kOO + q * 5 + b);
      vf* o = rr + b;
      DIM_S2_AT(o, m, 5)
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      vf o[4];
      const vf* m = rr + 5 * a;
      DIM_S2_AT(o, m, 1)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a * 4 + b] += o[b];
    }
  }
  vf bv = (vf)(0.f);
  if (bias) {   // element loads: the bias of a flat parameter blob is only 4-byte aligned (the other convolution paths accept that too)
#pragma unroll
    for (int e = 0; e < kVec; ++e) bv[e] = bias[co + e];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int oy = 4 * (int)ty + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ox = 4 * (int)tx + b;
      if (oy < Ho && ox < Wo) {
        vf v = acc[a * 4 + b] + bv;
#pragma unroll
        for (int e = 0; e < kVec; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        *reinterpret_cast<vf*>(y + (((long)n * Ho + oy) * Wo + ox) * out_cstride + out_coff + co) = v;
      }
    }
  }
}

// (Cout,Cin,3,3) -> the packed 1x1 weights of each of the 81 GEMMs: [p][ci/32][co][ci%32]; f64 inside (runs once per weight update)
__global__ void wino_s2_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cout * Cin) return;
  const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
  const float* g = w + ((long)co * Cin + ci) * 9;
  const double G[5][2] = {{0.5, 0.}, {0.5, 0.5}, {1. / 6, -1. / 6}, {1. / 6, 1. / 3}, {0., 1.}};
  float* o = wp + ((long)(ci >> 5) * Cout + co) * 32 + (ci & 31);
  const long per = (long)Cin * Cout;
  for (int k = 0; k < 16; ++k) o[k * per] = g[4];                                  // ee: w[1][1]
  for (int a = 0; a < 4; ++a)
    for (int q = 0; q < 5; ++q) o[(kEO + a * 5 + q) * per] = (float)(G[q][0] * g[3] + G[q][1] * g[5]);   // eo: G (w[1][0], w[1][2])
  for (int q = 0; q < 5; ++q)
    for (int b = 0; b < 4; ++b) o[(kOE + q * 4 + b) * per] = (float)(G[q][0] * g[1] + G[q][1] * g[7]);   // oe: G (w[0][1], w[2][1])
  for (int a = 0; a < 5; ++a)
    for (int b = 0; b < 5; ++b)
      o[(kOO + a * 5 + b) * per] = (float)(G[a][0] * (G[b][0] * g[0] + G[b][1] * g[2]) + G[a][1] * (G[b][0] * g[6] + G[b][1] * g[8]));
}

}  // namespace dim

using namespace dim;

// one slice of the batch: T * 81 * max(Cin, Cout) floats must stay below 2^32 bytes (32-bit buffer offsets in the GEMM)
static long s2_slice_images(long tiles_per_image, long K, long Cout) {
  const long per_image = tiles_per_image * kPlanes * (K > Cout ? K : Cout) * 4;
  return per_image < (1L << 32) ? ((1L << 32) - 1) / per_image : 0;
}

extern "C" {

long dim_winograd3x3s2_packed_weight_floats(int Cout, int Cin) { return wino_packed_with_split((long)kPlanes * Cout * Cin); }

// the one rule for every caller (FlowNetHip, dim_refiner_create): a 3x3 / stride-2 / pad-1 layer takes this path when its OUTPUT map has
// at least 300 pixels -- conv4 (30 x 40: 0.36 vs 0.40 ms at 16 pairs) and conv5 (15 x 20: 0.20 vs 0.22), not conv6 (8 x 10: its 2 x 3 tiles
// pad the map by 20 %, 96 GEMM rows fill one 128-row tile, and 170 MB of plane weights replace 19 MB: 0.13-0.15 vs 0.124 direct).
// DIM_WINO_S2K3=0: never.
int dim_winograd3x3s2_use(int H, int W, int Cin, int Cout) {
  static const int on = [] { const char* e = getenv("DIM_WINO_S2K3"); return e ? atoi(e) : 1; }();
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  // with the three-term plane GEMMs (wino_gemm_split.hip) conv6 gains too: 102 us for its three launches at 16 pairs against 128-135
  // direct (the 81 weight planes are then 255 MB of bf16 terms per forward: it is bound by reading them)
  return on && Cin % 32 == 0 && Cout % 64 == 0 && (Ho * Wo >= 300 || (wino_get_split() && Ho * Wo >= 80 && Cout % 128 == 0));
}

long dim_winograd3x3s2_workspace_floats(int N, int H, int W, int Cin, int Cout) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long per = (long)((Ho + 3) / 4) * ((Wo + 3) / 4);
  const long ns = s2_slice_images(per, Cin, Cout);
  const long n = ns <= 0 ? N : (ns < N ? ns : N);
  return kPlanes * n * per * ((long)Cin + Cout);
}

int dim_winograd3x3s2_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null pointer");
  DIM_REQUIRE(Cout % 64 == 0 && Cin % 32 == 0, "Cout %% 64 == 0 and Cin %% 32 == 0 required");
  const long total = (long)Cout * Cin;
  hipLaunchKernelGGL(wino_s2_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cout,
                     Cin);
  int rc = check_launch("winograd3x3s2_pack_weight");
  return rc != DIM_OK ? rc : wino_split_weights(w_packed, (long)kPlanes * (Cin / 32), Cout, as_stream(stream));
}

// y (N,Ho,Wo,out_cstride)[out_coff:+Cout] = LeakyReLU_slope(conv3x3 / stride 2 / pad 1 (x (N,H,W,in_cstride)[:Cin]) + bias), Ho = floor((H - 1) / 2) + 1.
// tile: 0 = auto, 3 / 4 / 5 = the plane GEMM's workgroup tile (dim_winograd_gemm_tile).  events4: optional 4 hipEvent_t recorded around
// the three launches (input transform, GEMMs, output transform) of the first slice.
int dim_conv2d_fwd_winograd3x3s2(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                                 int Cin, int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, void** events4,
                                 void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && w_packed && y && workspace, "null pointer");
  DIM_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "Cin %% 32 == 0 and Cout %% 64 == 0 required");
  if (in_cstride == 0) in_cstride = Cin;
  if (out_cstride == 0) out_cstride = Cout;
  DIM_REQUIRE(in_cstride >= Cin && in_cstride % 2 == 0 && out_cstride >= out_coff + Cout && out_cstride % 2 == 0 && out_coff % 2 == 0,
              "channel strides / offsets must be even and cover the channels");
  DIM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "x and y must be 8-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int th = (Ho + 3) / 4, tw = (Wo + 3) / 4;
  const long ns = s2_slice_images((long)th * tw, Cin, Cout);
  DIM_REQUIRE(ns > 0, "one image alone exceeds the 32-bit offsets of the plane GEMMs");
  int n_slice = ns < N ? (int)ns : N;
  if (const char* e = getenv("DIM_WINO_MAX_SLICE")) {  // test hook: force the slicing path at sizes a unit test can check
    const int cap = atoi(e);
    if (cap > 0 && cap < n_slice) n_slice = cap;
  }
  hipStream_t st = as_stream(stream);
  const FastDiv dtw = make_fastdiv((unsigned)tw), dth = make_fastdiv((unsigned)th);
  for (int n0 = 0; n0 < N; n0 += n_slice) {
    const int n = N - n0 < n_slice ? N - n0 : n_slice;
    const long T = (long)n * th * tw;
    float* V = workspace;
    float* M = workspace + kPlanes * T * Cin;
    const float* xs = x + (long)n0 * H * W * in_cstride;
    float* ys = y + (long)n0 * Ho * Wo * out_cstride;
    void** ev = n0 == 0 ? events4 : nullptr;
#define DIM_S2_EVENT(I)                                                                    \
  if (ev && ev[I]) {                                                                       \
    hipError_t e = hipEventRecord(reinterpret_cast<hipEvent_t>(ev[I]), st);                \
    if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipEventRecord: %s", hipGetErrorString(e)); \
  }
    int gt = tile;
    if (gt == 0) gt = dim_winograd_gemm_tile_planes(Cout, T, kPlanes);
    WGemmArgs plan;
    int rc = wino_gemm_plan(&plan, V, w_packed, M, (int)T, Cin, Cout, kPlanes, gt);
    if (rc != DIM_OK) return rc;
    DIM_S2_EVENT(0)
    const unsigned nblk = (unsigned)((T * (Cin / kVec) + 255) / 256);
    hipLaunchKernelGGL(wino_s2_input_kernel, dim3(nblk + plan.G - 1), dim3(256), 0, st, xs, V, n, H, W, Cin, in_cstride, th, tw,
                       make_fastdiv((unsigned)(Cin / kVec)), dtw, dth, nblk, plan);
    rc = check_launch("winograd3x3s2_input");
    if (rc != DIM_OK) return rc;
    DIM_S2_EVENT(1)
    rc = wino_gemm_run(plan, true, st);
    if (rc != DIM_OK) return rc;
    DIM_S2_EVENT(2)
    hipLaunchKernelGGL(wino_s2_output_kernel, dim3((unsigned)((T * (Cout / kVec) + 255) / 256)), dim3(256), 0, st, M, bias, ys, n, Ho, Wo, Cout,
                       out_cstride, out_coff, th, tw, slope, make_fastdiv((unsigned)(Cout / kVec)), dtw, dth);
    rc = check_launch("winograd3x3s2_output");
    if (rc != DIM_OK) return rc;
    DIM_S2_EVENT(3)
#undef DIM_S2_EVENT
  }
  return DIM_OK;
}

}  // extern "C"
