// fc6 of the pose head (deepim/symbols/deepIM_flownet.py:196-198: FullyConnected 81920 -> 256 on the flattened conv6_1 map) for the
// small batches of the refinement loop: y[b][o] = LeakyReLU(sum_k x[b][k] W[o][k] + bias[o]).
//
// With 16 rows this is a pure weight stream (84 MB per forward) and the general MFMA kernel -- 64x64 tiles, split-K 40 = 160
// workgroups, every chunk's weights one chunk ahead -- had 1.3 MB in flight and ran at 1.5 TB/s (58 us + 8 us slab reduce; now 23 us = 3.6 TB/s + 5 us).  Here every
// workgroup owns a contiguous range of K chunks and ALL 256 outputs: a wave takes 64 output columns (two 32x32 MFMA tiles), loads its
// B fragments straight from the packed [chunk][Out][32] weights (16 KB per wave and chunk in flight, one chunk ahead) and the <= 32
// activation rows straight from L2 (x is 5 MB) -- no LDS, no barrier -- and writes one partial tile; a second kernel sums the
// partials in a fixed order (deterministic), adds the bias and applies the activation.
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

// y[row0 + r][o] = act(sum_g partial[g][r][o] + bias[o]) for the rows of the batch; workgroup = 8 output quads x 32 partial groups
// (16-byte loads), fixed summation order: per group in g order, then the 32 group sums as a binary tree
__global__ __launch_bounds__(256) void fc_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                        float* __restrict__ y, int G, int B, int row0, int Out, float slope) {
  __shared__ float4 red[32][8];
  const int qi = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int q = blockIdx.x * 8 + qi;  // output quad within the pass: (row, 4 outputs)
  const int OQ = Out / 4;
  const int r = q / OQ, o = (q - r * OQ) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row0 + r < B)
    for (int g = part; g < G; g += 32) {
      const float4 v = *reinterpret_cast<const float4*>(partial + ((long)g * 32 + r) * Out + o);
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
  int g = 256;  // measured at B = 16: 256 workgroups 23.4 + 4.7 us (stream + reduce), 512: 25.4 + 6.5, 1024: 31.3 + 9.1
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
  for (int row0 = 0; row0 < B; row0 += 32) {  // 32 rows per pass (one MFMA row tile); the weights stream once per pass
    a.row0 = row0;
    hipLaunchKernelGGL(fc_stream_kernel, dim3(G, Out / 256), dim3(256), 0, st, a);
    const int rows = B - row0 < 32 ? B - row0 : 32;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3(ceil_div((long)rows * (Out / 4), 8)), dim3(256), 0, st, workspace, bias, y, G, B, row0, Out,
                       slope);
  }
  return check_launch("fc_fwd");
}

}  // extern "C"
