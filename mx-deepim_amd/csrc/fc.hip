// fc6 of the pose head (deepim/symbols/deepIM_flownet.py:196-198: FullyConnected 81920 -> 256 on the flattened conv6_1 map) for the
// small batches of the refinement loop: y[b][o] = LeakyReLU(sum_k x[b][k] W[o][k] + bias[o]).
//
// With 16 rows this is a pure weight stream (84 MB per forward) and the general MFMA kernel -- 64x64 tiles, split-K 40 = 160
// workgroups, every chunk's weights one chunk ahead -- had 1.3 MB in flight and ran at 1.5 TB/s (58 us + 8 us slab reduce; now 23 us = 3.6 TB/s + 5 us).  Here every
// workgroup owns a contiguous range of K chunks and ALL 256 outputs: a wave takes 64 output columns (two 32x32 MFMA tiles), loads its
// B fragments straight from the packed [chunk][Out][32] weights (16 KB per wave and chunk in flight, one chunk ahead) and the <= 32
// activation rows straight from L2 (x is 5 MB) -- no LDS, no barrier -- and writes one partial tile; a second kernel sums the
// partials in a fixed order (deterministic), adds the bias and applies the activation.
#include <cstdlib>

#include "common.h"

namespace dim {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FcArgs {
  const float* x;  // (B, H*W, C) = NHWC map, per sample HW*C floats
  const float* w;  // packed [chunk = (C/32 slice, hw)][Out][32]  (dim_fc_pack_weight)
  float* partial;  // [G][32][Out]
  int B, C, HW, Out, nchunks, chunks_per_wg, row0;
  unsigned x_bytes, w_bytes;
};

__global__ __launch_bounds__(256) void fc_stream_kernel(FcArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 31, khalf = lane >> 5;
  const int n0 = blockIdx.y * 256 + wave * 64;
  const int c_begin = blockIdx.x * a.chunks_per_wg;
  const int c_end = min(a.nchunks, c_begin + a.chunks_per_wg);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  // A fragment: lane (row b, k half) reads 4 consecutive k of its sample; rows past the batch read zeros (offset 0xFFFFFFFF)
  const int brow = a.row0 + frow;
  const int a_voff = brow < a.B ? (brow * a.HW * a.C + 4 * khalf) * 4 : -1;
  const int b_voff = ((n0 + frow) * 32 + 4 * khalf) * 4;
  const int wchunk_bytes = a.Out * 32 * 4;

  float4 fa[2][4], fb[2][4][2];
#define FC_LOAD(SET, KC)                                                                      \
  {                                                                                           \
    const int cs = (KC) / a.HW, hw = (KC) - cs * a.HW; /* chunk = (channel slice, pixel) */   \
    const int xoff = (hw * a.C + cs * 32) * 4;                                                \
    const int woff = (KC) * wchunk_bytes;                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                           \
      fa[SET][g] = buf_load16(rx, a_voff == -1 ? -1 : a_voff + g * 32, xoff);                 \
      fb[SET][g][0] = buf_load16_nt(rw, b_voff + g * 32, woff);                               \
      fb[SET][g][1] = buf_load16_nt(rw, b_voff + 32 * 32 * 4 + g * 32, woff);                 \
    }                                                                                         \
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#define FC_CHUNK(SET, KC)                                                                                  \
  {                                                                                                        \
    FC_LOAD(1 - SET, min((KC) + 1, a.nchunks - 1))                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int j = 0; j < 2; ++j) {           \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][g].x, fb[SET][g][j].x, acc[j], 0, 0, 0);        \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][g].y, fb[SET][g][j].y, acc[j], 0, 0, 0);        \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][g].z, fb[SET][g][j].z, acc[j], 0, 0, 0);        \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][g].w, fb[SET][g][j].w, acc[j], 0, 0, 0);        \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
  }
  if (c_begin < c_end) FC_LOAD(0, c_begin)
  for (int kc = c_begin; kc < c_end; kc += 2) {
    FC_CHUNK(0, kc)
    if (kc + 1 < c_end) FC_CHUNK(1, kc + 1)
  }
#undef FC_CHUNK
#undef FC_LOAD
  // D layout: col = lane & 31 (output), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (batch row)
  float* out = a.partial + (long)blockIdx.x * 32 * a.Out + n0 + frow;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (a.row0 + row < a.B) out[(long)row * a.Out + 32 * j] = acc[j][r];
    }
}

// ---- batches of at most 16 rows (the refinement loop: 16 pairs per GPU): v_mfma_f32_16x16x4_f32 instead of the 32-row tile (half of
// whose rows multiplied zeros), and the activations of the workgroup's whole K range staged ONCE in LDS.  Measured on MI355X
// (rocprofv3, 84 MB of weights, cold): the 32-row kernel 33 us; this shape with per-chunk activation loads from L2 (every wave fetching
// the same 2 KB, 16 half-used cache lines per instruction) 23.5 us; timing-only ablations of that: no MFMAs 21.0, activations loaded
// once 19.4, no stores 22.4, all three 17.9 -- against 14.5-16.4 us for a bare non-temporal read of the same 84 MB
// (tools/micro/stream_read.hip: 5.1 TB/s; WITHOUT the nt hint the same read takes 33 us behind a kernel that left dirty lines).
// Two or three chunks of weights in flight instead of one: 33.8 / 36.5 us against 32.9 for the pair of launches -- not latency-bound.
// Lane (i = lane & 15, kg = lane >> 4) supplies, for the half c of a chunk and k-step s, A[row i][k] and B[k][col i] with
// k = 16 c + 4 kg + s: 16 consecutive bytes per lane, half and operand; over s and kg every k of the half is summed once.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kFc16Ldk = 36;          // floats per (chunk, row) of the LDS image: ds_read_b128 of 16 rows x 4 k-groups is conflict-free
constexpr int kFc16MaxChunks = 24;    // chunks of a workgroup's K range (24 x 16 x 36 x 4 B = 55 KB of LDS)

__global__ __launch_bounds__(256) void fc_stream16_kernel(FcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sx[];   // [chunk][16 rows][kFc16Ldk]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.y * 256 + wave * 64;
  const int c_begin = blockIdx.x * a.chunks_per_wg;
  const int c_end = min(a.nchunks, c_begin + a.chunks_per_wg);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int b_voff = ((n0 + li) * 32 + 4 * kg) * 4;
  const int wchunk_bytes = a.Out * 32 * 4;
  float4 fb[2][4][2];
#define FC16_LOAD(SET, KC)                                                                                                     \
  {                                                                                                                            \
    const int woff = (KC) * wchunk_bytes;                                                                                      \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int t = 0; t < 4; ++t)                                \
      fb[SET][t][c] = buf_load16_nt(rw, b_voff + t * 16 * 32 * 4 + c * 64, woff);                                              \
  }
  // the first chunk of weights leaves for HBM before anything else; the activations of the whole range follow (x is L2 / Infinity
  // Cache resident: conv6_1's output transform has just written it)
  FC16_LOAD(0, min(c_begin, a.nchunks - 1))
  const int nloc = c_end - c_begin;
  for (int idx = threadIdx.x; idx < nloc * 128; idx += 256) {   // 128 float4 per chunk: 16 rows x 8
    const int j = idx >> 7, row = (idx >> 3) & 15, q = idx & 7;
    const int kc = c_begin + j;
    const int cs = kc / a.HW, hw = kc - cs * a.HW;   // chunk = (channel slice, pixel)
    const float4 v = buf_load16(rx, row < a.B ? ((row * a.HW + hw) * a.C + cs * 32 + q * 4) * 4 : -1, 0);   // rows past the batch: zeros
    *reinterpret_cast<float4*>(sx + (j * 16 + row) * kFc16Ldk + q * 4) = v;
  }
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* sa = sx + li * kFc16Ldk + 4 * kg;
#define FC16_STEP(SET, KC)                                                                                                     \
  {                                                                                                                            \
    FC16_LOAD(1 - SET, min((KC) + 1, a.nchunks - 1))                                                                           \
    float4 fa[2];                                                                                                              \
    fa[0] = *reinterpret_cast<const float4*>(sa + ((KC) - c_begin) * 16 * kFc16Ldk);                                           \
    fa[1] = *reinterpret_cast<const float4*>(sa + ((KC) - c_begin) * 16 * kFc16Ldk + 16);                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                                            \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].x, fb[SET][t][c].x, acc[t], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].y, fb[SET][t][c].y, acc[t], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].z, fb[SET][t][c].z, acc[t], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].w, fb[SET][t][c].w, acc[t], 0, 0, 0); \
    }                                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  }
  for (int kc = c_begin; kc < c_end; kc += 2) {
    FC16_STEP(0, kc)
    if (kc + 1 < c_end) FC16_STEP(1, kc + 1)
  }
#undef FC16_STEP
#undef FC16_LOAD
  // D layout of the 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + r
  float* out = a.partial + (long)blockIdx.x * 16 * a.Out + n0 + li;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      if (row < a.B) out[(long)row * a.Out + 16 * t] = acc[t][r];
    }
}

// y[row0 + r][o] = act(sum_g partial[g][r][o] + bias[o]) for the rows of the batch (prow = 16 or 32 rows per partial tile); workgroup = 8 output quads x 32 partial groups
// (16-byte loads), fixed summation order: per group in g order, then the 32 group sums as a binary tree
__global__ __launch_bounds__(256) void fc_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                        float* __restrict__ y, int G, int B, int row0, int Out, float slope, int prow) {
  __shared__ float4 red[32][8];
  const int qi = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int q = blockIdx.x * 8 + qi;  // output quad within the pass: (row, 4 outputs)
  const int OQ = Out / 4;
  const int r = q / OQ, o = (q - r * OQ) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row0 + r < B)
    for (int g = part; g < G; g += 32) {
      const float4 v = *reinterpret_cast<const float4*>(partial + ((long)g * prow + r) * Out + o);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[part][qi] = s;
  __syncthreads();
#pragma unroll
  for (int h = 16; h >= 1; h >>= 1) {
    if (part < h) {
      const float4 u = red[part + h][qi];
      float4 t = red[part][qi];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      red[part][qi] = t;
    }
    __syncthreads();
  }
  if (part == 0 && row0 + r < B) {
    float4 v = red[0][qi];
    if (bias) { v.x += bias[o]; v.y += bias[o + 1]; v.z += bias[o + 2]; v.w += bias[o + 3]; }
    v.x = v.x > 0.f ? v.x : v.x * slope;
    v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope;
    v.w = v.w > 0.f ? v.w : v.w * slope;
    *reinterpret_cast<float4*>(y + (long)(row0 + r) * Out + o) = v;
  }
}

static int fc_num_wgs(int nchunks) {
  static const int g_env = [] { const char* e = getenv("DIM_FC_WGS"); return e ? atoi(e) : 0; }();
  int g = g_env > 0 ? g_env : 256;  // measured at B = 16 (32-row kernel): 256 workgroups 23.4 + 4.7 us (stream + reduce), 512: 25.4 + 6.5, 1024: 31.3 + 9.1
  if ((long)g * kFc16MaxChunks < nchunks) g = (nchunks + kFc16MaxChunks - 1) / kFc16MaxChunks;   // the 16-row kernel's LDS image of x
  if (g > nchunks) g = nchunks;
  return g;
}

}  // namespace dim

using namespace dim;

extern "C" {

long dim_fc_fwd_workspace_floats(int C, int H, int W, int Out) {
  const int nchunks = (C / 32) * H * W;
  return (long)fc_num_wgs(nchunks) * 32 * Out;
}

int dim_fc_fwd(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int B, int C, int H, int W, int Out,
               float slope, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(x && w_packed && y && workspace, "null pointer");
  DIM_REQUIRE(C % 32 == 0 && Out % 256 == 0, "fc_fwd: C %% 32 == 0 and Out %% 256 == 0 required");
  DIM_REQUIRE((long)B * C * H * W * 4 < (1L << 31) && (long)Out * C * H * W * 4 < (1L << 32), "fc_fwd: operands too large for 32-bit offsets");
  FcArgs a = {};
  a.x = x;
  a.w = w_packed;
  a.partial = workspace;
  a.B = B;
  a.C = C;
  a.HW = H * W;
  a.Out = Out;
  a.nchunks = (C / 32) * H * W;
  const int G0 = fc_num_wgs(a.nchunks);
  a.chunks_per_wg = (a.nchunks + G0 - 1) / G0;
  const int G = (a.nchunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
  a.x_bytes = (unsigned)((long)B * C * H * W * 4);
  a.w_bytes = (unsigned)((long)Out * C * H * W * 4);
  hipStream_t st = as_stream(stream);
  if (B <= 16) {   // the refinement loop's batch: 16-row MFMA tiles, activations staged once per workgroup (fc_stream16_kernel)
    a.row0 = 0;
    hipLaunchKernelGGL(fc_stream16_kernel, dim3(G, Out / 256), dim3(256), (size_t)a.chunks_per_wg * 16 * kFc16Ldk * sizeof(float), st, a);
    hipLaunchKernelGGL(fc_reduce_kernel, dim3(ceil_div((long)B * (Out / 4), 8)), dim3(256), 0, st, workspace, bias, y, G, B, 0, Out, slope, 16);
    return check_launch("fc_fwd");
  }
  for (int row0 = 0; row0 < B; row0 += 32) {  // 32 rows per pass (one MFMA row tile); the weights stream once per pass
    a.row0 = row0;
    hipLaunchKernelGGL(fc_stream_kernel, dim3(G, Out / 256), dim3(256), 0, st, a);
    const int rows = B - row0 < 32 ? B - row0 : 32;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3(ceil_div((long)rows * (Out / 4), 8)), dim3(256), 0, st, workspace, bias, y, G, B, row0, Out,
                       slope, 32);
  }
  return check_launch("fc_fwd");
}

}  // extern "C"
