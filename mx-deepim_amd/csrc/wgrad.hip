// Weight gradient of the direct convolution, im2col-free, on the f32 MFMA pipe.
//
//   dWp[chunk][co][kin] = sum over pixels m=(n,ho,wo) of  dZ[m][co] * X[n, ho*s-p+kh, wo*s-p+kw, c0+kin]
//
// written DIRECTLY in the forward kernel's packed weight layout ([chunk][Cout][32], chunk = (32-channel slice, kh, kw) or, for
// the 8-channel first layer, four consecutive flat taps x 8 channels); dim_conv2d_unpack_weight brings it into the MXNet-layout
// gradient bucket the optimizer works on.  Replaces the cuDNN backward-filter calls behind MXNet's Convolution / FullyConnected backward
// (reference graph: deepim/symbols/deepIM_flownet.py:67-208; executor: deepim/core/module.py:1205-1209).
//
// Implicit GEMM with the PIXELS as the contraction dimension: A = dZ^T (co x pixels), B = gathered X (pixels x 32).
// Workgroup = NW waves, each owning a 32(co) x 32(kin) accumulator; 32 pixels per step are staged in LDS in their natural
// layouts ([pixel][co], [pixel][kin]) -- lane (i, h) reads A[co=i][pixel=2s+h] and B[pixel=2s+h][kin=i] with conflict-free
// ds_read_b32.  grid = (chunks, Cout/(32*NW), pixel splits); splits write slabs that dim_splitk_reduce sums (deterministic).
#include "common.h"

namespace dim {
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* x;   // (N,H,W,in_cstride)
  const float* dz;  // (N,Ho,Wo,dz_cstride)  gradient w.r.t. the pre-activation output
  float* dw;        // packed layout, or slab base when splits > 1
  int N, H, W, Cin, in_cstride, Ho, Wo, Cout, dz_cstride, dz_coff;
  int KH, KW, stride, pad;
  int M, nchunks, steps_per_split, nsteps;
  int accumulate;
  unsigned x_bytes, dz_bytes;  // extents for the buffer descriptors
  FastDiv div_wo, div_ho;      // pixel index -> (n, ho, wo) every step without integer division sequences
  // Winograd planes (dim_conv2d_wgrad_winograd): the K chunks [p * chunks_per_plane, (p + 1) * chunks_per_plane) belong to plane p,
  // whose dZ operand sits dz_plane_stride channels further in the row (0 / 0: plain convolution)
  int chunks_per_plane, dz_plane_stride;
  int bf16;  // products on v_mfma_f32_32x32x16_bf16 (conv_wgrad_bf16_kernel); the result is f32 in the same packed layout
  int xcd;   // number the workgroups XCD-contiguously (wgrad_block below)
  int dbg;   // timing-only ablation switches of the bf16 kernel (DIM_WGB_DBG; results are wrong when set): 1 no LDS stores after the
             // first step, 8 no MFMAs, 16 no global loads after the prologue, 32 8-byte instead of 16-byte operand loads, 64 half as many 16-byte
             // loads (two ways of moving the bytes bf16 operands in HBM would move; compile-time twins: DESIGN.md section 6b)
};

// (chunk tile, output-channel tile, pixel split) of this workgroup.  The hardware deals linear block ids round-robin to the 8 XCDs, so
// with the chunk tile as the fastest grid index the workgroups that share one dZ tile landed on 8 different L2s; wg_xcd_contiguous
// hands every XCD a contiguous run of the (x fastest) order instead: chunk tiles of one (channel tile, split) follow each other on
// one XCD and find dZ -- and the next channel tile its X -- in that L2.
__device__ __forceinline__ void wgrad_block(const WgradArgs& a, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (a.xcd) {
    const int G = gridDim.x * gridDim.y * gridDim.z;
    const int m = wg_xcd_contiguous(bx + gridDim.x * (by + gridDim.y * bz), G);
    bx = m % gridDim.x;
    const int t = m / gridDim.x;
    by = t % gridDim.y;
    bz = t / gridDim.y;
  }
}

template <int NW, bool CIN8, int NCH>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int BP = 32;            // pixels per step
  constexpr int BM = 32 * NW;       // output channels per workgroup (one 32-row MFMA tile per wave)
  constexpr int LDZ = BM + 4;       // [pixel][co] rows; float4-aligned
  constexpr int LDX = 32 * NCH + 4; // [pixel][NCH chunks x 32 kin]
  constexpr int NT = 64 * NW;
  constexpr int ZQ = BM / 4;                       // float4 per dZ row
  constexpr int Z_PER_T = (BP * ZQ) / NT;          // = 4
  constexpr int ZR_STEP = NT / ZQ;
  constexpr int X_PER_T = (BP * 8 * NCH) / NT;     // float4 of X per thread and step
  constexpr int XR_STEP = NT / 8;
  __shared__ __attribute__((aligned(16))) float sZ[2][BP * LDZ];
  __shared__ __attribute__((aligned(16))) float sX[2][BP * LDX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx, by, split;
  wgrad_block(a, bx, by, split);
  const int kc0 = bx * NCH;
  const int co0 = by * BM;
  const int step_begin = split * a.steps_per_split;
  const int step_end = min(a.nsteps, step_begin + a.steps_per_split);

  // chunk -> tap / channel slice (same order as conv_fwd_kernel / pack_conv_weight_kernel), one decode per chunk of the block
  int kh[NCH], kw[NCH], c0[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int kc = kc0 + c;
    if (CIN8) {
      // 8-channel layer: 4 taps x 8 channels per chunk, taps flat over KH x KW; this THREAD stages tap 4 kc + (xq >> 1)
      // (kh >= KH for the padding taps of the last chunk: rejected by the bounds test)
      const int t = 4 * kc + ((tid & 7) >> 1);
      kh[c] = t / a.KW; kw[c] = t - kh[c] * a.KW; c0[c] = 0;
    } else {
      int taps = a.KH * a.KW;
      int cc = kc / taps, tap = kc - cc * taps;
      c0[c] = cc << 5; kh[c] = tap / a.KW; kw[c] = tap - kh[c] * a.KW;
    }
  }
  const int zq = tid % ZQ, zr0 = tid / ZQ;
  const int xq = tid & 7, xr0 = tid >> 3;

  constexpr int XP = X_PER_T / NCH;  // pixel rows per thread per chunk (1 for NW=4, 2 for NW=2)
  static_assert(XP >= 1 && XP <= 2 && Z_PER_T == 4 && NCH <= 2, "staging register plan");
  // named staging registers: float4 arrays indexed inside macros / lambdas ended up in scratch (hipcc, ROCm 7.2)
  float4 rz0, rz1, rz2, rz3, rx00, rx01, rx10, rx11;

  f32x16 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  // Buffer descriptors: a padding tap / a pixel past M is a load at offset 0xFFFFFFFF (range check -> zeros).  No pointer
  // select (it compiled to flat loads, which also count on lgkmcnt and so serialised against the LDS fragment reads).
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, a.dz_bytes, 0x00020000);
  const int z_voff = (a.dz_coff + (a.chunks_per_plane ? (kc0 / a.chunks_per_plane) * a.dz_plane_stride : 0) + co0 + zq * 4) * 4;
  int tapb[NCH];  // byte offset of each chunk's tap / channel slice
#pragma unroll
  for (int c = 0; c < NCH; ++c) tapb[c] = ((kh[c] * a.W + kw[c]) * a.in_cstride + c0[c]) * 4;
#define DIM_WG_LZ(REG, I)                                                                                         \
  {                                                                                                               \
    int m = p0 + zr0 + ZR_STEP * I;                                                                               \
    REG = buf_load16(rsz, (pf && m < a.M) ? m * (a.dz_cstride * 4) + z_voff : -1, 0);                             \
  }
#define DIM_WG_LX(REG, C, I)                                                                                      \
  if (C < NCH && I < XP) {                                                                                        \
    int hi = hb##I + kh[C];                                                                                       \
    int wi = wb##I + kw[C];                                                                                       \
    bool ok = okm##I && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W && (!CIN8 || kh[C] < a.KH);  \
    REG = buf_load16(rsx, ok ? pix##I + tapb[C] : -1, 0);                                                         \
  }
#define DIM_WG_PIX(I)                                                                                             \
  int m##I = p0 + xr0 + XR_STEP * I;                                                                              \
  bool okm##I = pf && (I < XP) && m##I < a.M;                                                                     \
  unsigned mm##I = okm##I ? m##I : 0;                                                                             \
  unsigned t##I = fastdiv(mm##I, a.div_wo), n##I = fastdiv(t##I, a.div_ho);                                       \
  int wo##I = mm##I - t##I * a.Wo, ho##I = t##I - n##I * a.Ho;                                                    \
  int hb##I = ho##I * a.stride - a.pad, wb##I = wo##I * a.stride - a.pad;                                         \
  int pix##I = (((int)n##I * a.H + hb##I) * a.W + wb##I) * (a.in_cstride * 4) + (CIN8 ? (xq & 1) : xq) * 16;
  // global -> registers for step `st` (pf = false: everything reads the zero block; used for the prefetch past the end)
#define DIM_WG_LOAD(st, pf_ok)                                      \
  {                                                                 \
    const int p0 = (st) * BP;                                       \
    const bool pf = (pf_ok);                                        \
    DIM_WG_LZ(rz0, 0) DIM_WG_LZ(rz1, 1) DIM_WG_LZ(rz2, 2) DIM_WG_LZ(rz3, 3) \
    DIM_WG_PIX(0) DIM_WG_PIX(1)                                     \
    DIM_WG_LX(rx00, 0, 0) DIM_WG_LX(rx01, 0, 1) DIM_WG_LX(rx10, 1, 0) DIM_WG_LX(rx11, 1, 1) \
  }
#define DIM_WG_SZ(REG, I) *reinterpret_cast<float4*>(&sZ[buf_][(zr0 + ZR_STEP * I) * LDZ + zq * 4]) = REG;
#define DIM_WG_SX(REG, C, I) \
  if (C < NCH && I < XP) *reinterpret_cast<float4*>(&sX[buf_][(xr0 + XR_STEP * I) * LDX + C * 32 + xq * 4]) = REG;
#define DIM_WG_STORE(buf)                                           \
  {                                                                 \
    const int buf_ = (buf);                                         \
    DIM_WG_SZ(rz0, 0) DIM_WG_SZ(rz1, 1) DIM_WG_SZ(rz2, 2) DIM_WG_SZ(rz3, 3) \
    DIM_WG_SX(rx00, 0, 0) DIM_WG_SX(rx01, 0, 1) DIM_WG_SX(rx10, 1, 0) DIM_WG_SX(rx11, 1, 1) \
  }

  if (step_begin < step_end) {
    DIM_WG_LOAD(step_begin, true)
    DIM_WG_STORE(0)
  }
  __syncthreads();
  const int fi = lane & 31, fh = lane >> 5;
  int buf = 0;
  for (int st = step_begin; st < step_end; ++st) {
    DIM_WG_LOAD(min(st + 1, a.nsteps - 1), st + 1 < step_end)
    __builtin_amdgcn_sched_barrier(0);  // prefetch stays ahead of the MFMA block
    const float* cz = &sZ[buf][fh * LDZ + wave * 32 + fi];
    const float* cx = &sX[buf][fh * LDX + fi];
#pragma unroll
    for (int s = 0; s < BP / 2; ++s) {
      float av = cz[2 * s * LDZ];
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, cx[2 * s * LDX + 32 * c], acc[c], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    DIM_WG_STORE(buf ^ 1)
    __syncthreads();
    buf ^= 1;
  }
#undef DIM_WG_LOAD
#undef DIM_WG_STORE
#undef DIM_WG_LZ
#undef DIM_WG_LX
#undef DIM_WG_PIX
#undef DIM_WG_SZ
#undef DIM_WG_SX
  // D: col = lane&31 -> kin, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> co within the wave's 32
  const bool add = gridDim.z == 1 && a.accumulate;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (kc0 + c >= a.nchunks) break;  // odd chunk count: the last workgroup's second chunk does not exist (its loads read zeros)
    float* out = (gridDim.z == 1 ? a.dw : a.dw + (long)split * a.nchunks * a.Cout * 32) +
                 ((long)(kc0 + c) * a.Cout + co0 + wave * 32 + 4 * fh) * 32 + fi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* o = out + (long)((r & 3) + 8 * (r >> 2)) * 32;
      float v = acc[c][r];
      if (add) v += *o;
      *o = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- bf16 MFMA
// The same contraction on v_mfma_f32_32x32x16_bf16 (training mode of BASELINE configs[2]; f32 accumulate, f32 result).
// Both operands are needed with the CONTRACTION index (pixels) running inside a lane -- A[co][8 consecutive pixels], B[8 consecutive
// pixels][kin] -- while memory is channel-contiguous.  The tiles are therefore staged in their natural [pixel][channel] layout
// (float4 -> 4 bf16 -> ds_write_b64) and read back through the transposing LDS read ds_read_b64_tr_b16: a 16-lane group fetches a
// 4-pixel x 16-channel block and every lane receives one channel's four pixels.  Row strides of 192 / 320 bytes put the four rows of
// a block on disjoint bank quarters.  One workgroup = 32 NW output channels x NCH chunks (up to 128 packed columns) sharing one dZ
// tile: with the matrix pipe 16x faster the kernel is bound by L2 -> LDS operand traffic, which is what the wide tile cuts.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ wbf16x4 wg_to_bf16x4(const float4& v) {
  wbf16x4 p = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return p;
}

// Occupancy: 40 KB of LDS per workgroup of the wide instantiation (NW = 4, NCH = 4) = 3 workgroups per CU, whatever the register count
// (measured: a grid of 768 = 3 x 256 workgroups runs 20-25 % faster than one of 1024; amdgpu_waves_per_eu(4, 4) fits the kernel into 124
// registers without spilling and changes nothing).  The phases of a step -- global loads, convert + LDS stores, barrier, transposing
// reads, MFMAs -- do not overlap inside one workgroup (timing-only ablations: 221 us = 49 skeleton + epilogue + slab sum, + 33 MFMA,
// + ~35 LDS reads, + ~25 stores, + ~80 global loads on conv3_1), so the co-resident workgroups are what hides them: the launcher's
// callers size the pixel split so that the whole grid is resident at once, and the global loads run TWO steps ahead (two register
// sets: the registers are there, occupancy is LDS-bound).
template <int NW, bool CIN8, int NCH, int HALF = 0>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_bf16_kernel(WgradArgs a) {
  constexpr int BP = 32;              // pixels per step = two k-steps of the instruction
  constexpr int BM = 32 * NW;         // output channels per workgroup
  constexpr int LDZ = BM + 32;        // bf16 elements per [pixel] row of the dZ tile: 192 B (NW = 2) / 320 B (NW = 4)
  constexpr int LDX = 32 * NCH + 32;  // ... of the X tile: 128 / 192 / 320 B for NCH = 1 / 2 / 4
  constexpr int NT = 64 * NW;
  constexpr int ZQ = BM / 4;          // float4 per dZ row
  constexpr int ZR_STEP = NT / ZQ;    // = 8
  constexpr int XP = (BP * 8) / NT;   // pixel rows per thread: 1 (NW = 4) or 2 (NW = 2)
  constexpr int XR_STEP = NT / 8;
  static_assert(BP * ZQ / NT == 4 && (XP == 1 || XP == 2) && (NCH == 1 || NCH == 2 || NCH == 4), "staging plan");
  __shared__ __attribute__((aligned(16))) __bf16 sZ[2][BP * LDZ];
  __shared__ __attribute__((aligned(16))) __bf16 sX[2][BP * LDX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx, by, split;
  wgrad_block(a, bx, by, split);
  const int kc0 = bx * NCH;
  const int co0 = by * BM;
  const int step_begin = split * a.steps_per_split;
  const int step_end = min(a.nsteps, step_begin + a.steps_per_split);

  const int zq = tid % ZQ, zr0 = tid / ZQ;
  const int xq = tid & 7, xr0 = tid >> 3;
  int kh[NCH], kw[NCH], tapb[NCH];
  bool cok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int kc = kc0 + c;
    int c0;
    if (CIN8) {
      const int t = 4 * kc + (xq >> 1);  // this THREAD's flat tap of the chunk (4 taps x 8 channels)
      kh[c] = t / a.KW; kw[c] = t - kh[c] * a.KW; c0 = 0;
      cok[c] = kc < a.nchunks && kh[c] < a.KH;
    } else {
      const int taps = a.KH * a.KW;
      const int cc = kc / taps, tap = kc - cc * taps;
      c0 = cc << 5; kh[c] = tap / a.KW; kw[c] = tap - kh[c] * a.KW;
      cok[c] = kc < a.nchunks;
    }
    tapb[c] = ((kh[c] * a.W + kw[c]) * a.in_cstride + c0) * 4;
  }
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, a.dz_bytes, 0x00020000);
  const int z_voff = (a.dz_coff + co0 + zq * 4) * 4;

  // two staging register sets: while set A (step st + 1) waits to be written to LDS, the loads of step st + 2 fill set B
  float4 rzA[4], rxA[XP][NCH], rzB[4], rxB[XP][NCH];
  auto load_step = [&](float4 (&rz)[4], float4 (&rx)[XP][NCH], int st, bool pf) {
    const int p0 = st * BP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = p0 + zr0 + ZR_STEP * i;
      if constexpr (HALF == 2) {  // timing only: half as many 16-byte loads (bf16 operands in HBM fetched 8 elements at a time)
        rz[i] = (i & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : buf_load16(rsz, (pf && m < a.M) ? m * (a.dz_cstride * 4) + z_voff : -1, 0);
      } else if constexpr (HALF == 1) {  // timing only: half the bytes per load, as many loads
        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rsz, (pf && m < a.M) ? m * (a.dz_cstride * 4) + z_voff : -1, 0, 0);
        rz[i] = make_float4(__uint_as_float(h.x), __uint_as_float(h.y), 0.f, 0.f);
      } else {
        rz[i] = buf_load16(rsz, (pf && m < a.M) ? m * (a.dz_cstride * 4) + z_voff : -1, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = p0 + xr0 + XR_STEP * i;
      const bool okm = pf && m < a.M;
      const unsigned mm = okm ? m : 0;
      const unsigned t = fastdiv(mm, a.div_wo), n = fastdiv(t, a.div_ho);
      const int wo = mm - t * a.Wo, ho = t - n * a.Ho;
      const int hb = ho * a.stride - a.pad, wb = wo * a.stride - a.pad;
      const int pix = (((int)n * a.H + hb) * a.W + wb) * (a.in_cstride * 4) + (CIN8 ? (xq & 1) : xq) * 16;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool ok = okm && cok[c] && (unsigned)(hb + kh[c]) < (unsigned)a.H && (unsigned)(wb + kw[c]) < (unsigned)a.W;
        if constexpr (HALF == 2) {
          rx[i][c] = (c & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : buf_load16(rsx, ok ? pix + tapb[c] : -1, 0);
        } else if constexpr (HALF == 1) {
          const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rsx, ok ? pix + tapb[c] : -1, 0, 0);
          rx[i][c] = make_float4(__uint_as_float(h.x), __uint_as_float(h.y), 0.f, 0.f);
        } else {
          rx[i][c] = buf_load16(rsx, ok ? pix + tapb[c] : -1, 0);
        }
      }
    }
  };
  auto store_step = [&](const float4 (&rz)[4], const float4 (&rx)[XP][NCH], int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<wbf16x4*>(&sZ[buf][(zr0 + ZR_STEP * i) * LDZ + zq * 4]) = wg_to_bf16x4(rz[i]);
#pragma unroll
    for (int i = 0; i < XP; ++i)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        *reinterpret_cast<wbf16x4*>(&sX[buf][(xr0 + XR_STEP * i) * LDX + c * 32 + xq * 4]) = wg_to_bf16x4(rx[i][c]);
  };

  // Wave -> accumulator tiles.  NW = 4, NCH = 4 (the wide layers): the four waves sit 2 x 2 on the 128 (co) x 128 (packed columns)
  // workgroup tile, 64 x 64 each -- per k-step 2 A + 2 B fragments (8 transposing reads) feed 4 MFMAs; the first layout, one 32-row
  // strip x all 4 chunks per wave, needed 10 reads per 4 MFMAs, and with both operands zeroed at the source the kernel still took 146
  // of its 237 us on conv3_1: the LDS read path, not the global loads, sets its pace.  Other shapes keep the strip layout.
  constexpr bool SQ = NW == 4 && NCH == 4;
  constexpr int TI = SQ ? 2 : 1, TJ = SQ ? 2 : NCH;
  const int wr = SQ ? wave >> 1 : wave, wc = SQ ? wave & 1 : 0;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposing-read addresses: lane 4q+p of 16-lane group g points at pixel row 8 (g >> 1) + q, channels 16 (g & 1) + 4p .. +3 of
  // the block; it receives channel 16 (g & 1) + (lane & 15) for the four pixels 8 (g >> 1) + {0..3} (+4 for the second read)
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int z_el = (8 * (g >> 1) + tq) * LDZ + wr * (32 * TI) + 16 * (g & 1) + 4 * tp;
  const int x_el = (8 * (g >> 1) + tq) * LDX + wc * (32 * TJ) + 16 * (g & 1) + 4 * tp;
  typedef ws16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

  const int dbg = a.dbg;
  union Frag { ws16x4 h[2]; wbf16x8 v; };
  // the MFMAs of the step in LDS buffer `buf`
  auto compute = [&](int buf) {
    Frag fa[TI], fb;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const __bf16* pz = &sZ[buf][z_el + 16 * ks * LDZ + 32 * i];
        fa[i].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pz));
        fa[i].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pz + 4 * LDZ));
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const __bf16* px = &sX[buf][x_el + 16 * ks * LDX + 32 * j];
        fb.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px));
        fb.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px + 4 * LDX));
        if (!(dbg & 8)) {
#pragma unroll
          for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fb.v, acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  const int last = a.nsteps - 1;
  if (step_begin < step_end) {
    load_step(rzA, rxA, step_begin, true);
    store_step(rzA, rxA, 0);
    load_step(rzA, rxA, min(step_begin + 1, last), step_begin + 1 < step_end);   // step st + 1 is in flight when the loop starts
  }
  __syncthreads();
  // two steps per trip so that the register sets keep their names: LDS buffer 0 holds step st, set A step st + 1
  for (int st = step_begin; st < step_end; st += 2) {
    // LDS-only barriers (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() also waits for vmcnt(0), i.e. for the loads of step st + 2
    // issued a few lines above, so every barrier drained the prefetch the second staging set exists for.  Measured after the change:
    // 6.17 vs 6.17 ms per iteration -- the kernel is bound by the NUMBER of vector loads it issues, not by their latency or their bytes
    // (the compile-time twins below); kept because it is what the code means.  The compiler waits for a set's loads where store_step reads them.
    if (!(dbg & 16)) load_step(rzB, rxB, min(st + 2, last), st + 2 < step_end);
    compute(0);
    if (!(dbg & 1)) store_step(rzA, rxA, 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 1 >= step_end) break;
    if (!(dbg & 16)) load_step(rzA, rxA, min(st + 3, last), st + 3 < step_end);
    compute(1);
    if (!(dbg & 1)) store_step(rzB, rxB, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // D: col = lane&31 -> kin, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> co within the tile's 32
  const int fi = lane & 31, fh = lane >> 5;
  const bool add = gridDim.z == 1 && a.accumulate;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int c = wc * TJ + j;
    if (kc0 + c >= a.nchunks) break;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      float* out = (gridDim.z == 1 ? a.dw : a.dw + (long)split * a.nchunks * a.Cout * 32) +
                   ((long)(kc0 + c) * a.Cout + co0 + wr * (32 * TI) + 32 * i + 4 * fh) * 32 + fi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* o = out + (long)((r & 3) + 8 * (r >> 2)) * 32;
        float v = acc[i][j][r];
        if (add) v += *o;
        *o = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------- bf16, patch form
// The gathered-tap kernel above moves 32 KB of f32 operands through the vector memory pipe per 32-pixel step of a 128 x 128 tile:
// 512 cycles of a CU's 64 B/clk against 258 cycles of MFMA work -- it is bound by that pipe, not by latency (timing-only ablations
// and a two-step-ahead prefetch that changed nothing).  For the 3x3 / stride-1 layers this kernel cuts the bytes per MFMA 3.2x:
//   workgroup  = 128 output channels x ONE 32-channel input slice x ALL nine taps (288 packed columns), 4 waves = 4 strips of 32 rows
//   step       = one 8 x 8 block of output pixels: the dZ tile (64 px x 128 co) and ONE 10 x 10 input patch of the slice are staged in
//                LDS (bf16); every tap reads its own shifted window of the patch through the transposing read (a lane supplies the
//                address of its own pixel row, so a window is just nine different base offsets)
//   per step   : 44.8 KB through the memory pipe (700 clk) for 1160 clk of MFMA work -- the gathered form needs 144 KB for the same work
// Patch pixels are 64 B apart: the four pixels a 16-lane group transposes sit on four disjoint quarters of the 64 banks.
// Pixel blocks that hang over the map edge load zeros (buffer range check), so partial blocks need no special path.
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_bf16_patch_kernel(WgradArgs a) {
  constexpr int BM = 128, TAPS = KS * KS, PW = 8 + KS - 1, PP = PW * PW;   // patch: (8 + K - 1)^2 input pixels
  constexpr int LDZ = BM + 32;                   // dZ rows: 320 B apart (bank quarters of the transposing read, as above)
  constexpr int ZF4 = 64 * (BM / 4) / 256;       // float4 per thread of the dZ tile = 8
  constexpr int XF4 = (PP * 8 + 255) / 256;      // float4 per thread of the patch (100 px x 8) = 4
  __shared__ __attribute__((aligned(16))) __bf16 sZ[2][64 * LDZ];
  __shared__ __attribute__((aligned(16))) __bf16 sX[2][PP * 32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cc = blockIdx.x;           // input-channel slice
  const int co0 = blockIdx.y * BM;
  const int split = blockIdx.z;
  const int nbx = (a.Wo + 7) >> 3, nby = (a.Ho + 7) >> 3;
  const int nblocks = a.N * nby * nbx;
  const int blk_begin = split * a.steps_per_split, blk_end = min(nblocks, blk_begin + a.steps_per_split);

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, a.dz_bytes, 0x00020000);
  const int zq = tid & 31, zc = tid >> 5;   // dZ: float4 column zq of pixel column zc, block rows 0..7
  float4 rzA[ZF4], rxA[XF4];   // ONE staging set: the loads of block b + 1 are issued before the ~1200 cycles of MFMAs of block b
  auto load_block = [&](float4 (&rz)[ZF4], float4 (&rx)[XF4], int blk, bool pf) {
    const int n = blk / (nby * nbx), r = blk - n * (nby * nbx), by = r / nbx, bx = r - by * nbx;
    const int y0 = by * 8, x0 = bx * 8;
#pragma unroll
    for (int i = 0; i < ZF4; ++i) {
      const int y = y0 + i, x = x0 + zc;
      const bool ok = pf && y < a.Ho && x < a.Wo;
      rz[i] = buf_load16(rsz, ok ? (((n * a.Ho + y) * a.Wo + x) * a.dz_cstride + a.dz_coff + co0 + zq * 4) * 4 : -1, 0);
    }
#pragma unroll
    for (int i = 0; i < XF4; ++i) {
      const int idx = tid + 256 * i, pp = idx >> 3, c4 = idx & 7;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = y0 - a.pad + py, ix = x0 - a.pad + px;
      const bool ok = pf && idx < PP * 8 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      rx[i] = buf_load16(rsx, ok ? (((n * a.H + iy) * a.W + ix) * a.in_cstride + cc * 32 + c4 * 4) * 4 : -1, 0);
    }
  };
  auto store_block = [&](const float4 (&rz)[ZF4], const float4 (&rx)[XF4], int buf) {
#pragma unroll
    for (int i = 0; i < ZF4; ++i) *reinterpret_cast<wbf16x4*>(&sZ[buf][(8 * i + zc) * LDZ + zq * 4]) = wg_to_bf16x4(rz[i]);
#pragma unroll
    for (int i = 0; i < XF4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < PP * 8) *reinterpret_cast<wbf16x4*>(&sX[buf][idx * 4]) = wg_to_bf16x4(rx[i]);
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing-read addresses (see the gathered kernel): lane 4q + p of 16-lane group g points at pixel 8 (g >> 1) + q of the k-step,
  // channels 16 (g & 1) + 4p .. +3.  A k-step = two rows of the 8 x 8 block: pixel j -> block row 2 ks + (j >> 3), column j & 7.
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int z_el = (8 * (g >> 1) + tq) * LDZ + wave * 32 + 16 * (g & 1) + 4 * tp;
  const int x_el = ((g >> 1) * PW + tq) * 32 + 16 * (g & 1) + 4 * tp;   // patch pixel (row g >> 1, column tq) of tap (0, 0)
  typedef ws16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  union Frag { ws16x4 h[2]; wbf16x8 v; };
  auto compute = [&](int buf) {
#pragma unroll 1
    for (int ks = 0; ks < 4; ++ks) {   // not unrolled: hipcc hoists every fragment read of an unrolled body (36 x 4 registers)
      Frag fa;
      const __bf16* pz = &sZ[buf][z_el + 16 * ks * LDZ];
      fa.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pz));
      fa.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pz + 4 * LDZ));
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = t / KS, dx = t - dy * KS;
        Frag fb;
        const __bf16* px = &sX[buf][x_el + ((2 * ks + dy) * PW + dx) * 32];
        fb.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px));
        fb.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px + 4 * 32));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[t], 0, 0, 0);
      }
    }
  };

  if (blk_begin < blk_end) {
    load_block(rzA, rxA, blk_begin, true);
    store_block(rzA, rxA, 0);
  }
  __syncthreads();
  int buf = 0;
  for (int blk = blk_begin; blk < blk_end; ++blk) {
    load_block(rzA, rxA, min(blk + 1, nblocks - 1), blk + 1 < blk_end);
    compute(buf);
    store_block(rzA, rxA, buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // D: col = lane & 31 -> kin, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> co within the wave's 32
  const int fi = lane & 31, fh = lane >> 5;
  float* base = gridDim.z == 1 ? a.dw : a.dw + (long)split * a.nchunks * a.Cout * 32;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    float* out = base + ((long)(cc * TAPS + t) * a.Cout + co0 + wave * 32 + 4 * fh) * 32 + fi;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)((r & 3) + 8 * (r >> 2)) * 32] = acc[t][r];
  }
}

// column sums: db[c] = sum_m dz[m][coff + c]; grid (C/64, nsplit) -> partial[nsplit][C]; then a tiny reduce
// thread = 4 adjacent channels (one 16-byte load per row); workgroup = 64 channels x 16 row lanes, 4 rows in flight per lane
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dz, int M, int C, int cstride, int coff,
                                                             int rows_per_block, float* __restrict__ partial) {
  const int cg = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (c < C) {
    const float* base = dz + coff + c;
    int m = r0 + rsub;
    for (; m + 48 < r1; m += 64) {
      float4 v0 = *reinterpret_cast<const float4*>(base + (long)m * cstride);
      float4 v1 = *reinterpret_cast<const float4*>(base + (long)(m + 16) * cstride);
      float4 v2 = *reinterpret_cast<const float4*>(base + (long)(m + 32) * cstride);
      float4 v3 = *reinterpret_cast<const float4*>(base + (long)(m + 48) * cstride);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
      s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
      s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
      s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
    }
    for (; m < r1; m += 16) {
      float4 v0 = *reinterpret_cast<const float4*>(base + (long)m * cstride);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    }
  }
  __shared__ float4 red[16][16];
  red[rsub][cg] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                              (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (threadIdx.x < 64) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    const float* r = reinterpret_cast<const float*>(&red[0][0]) + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += r[k * 64];
    if (col < C) partial[(long)blockIdx.y * C + col] = t;
  }
}

// LeakyReLU' and the bias gradient in ONE pass over the gradient map: dz = dy * (y > 0 ? 1 : slope) in place, partial column sums of
// dz as in colsum_partial_kernel (same grid, same partial layout, same final fold).  The two separate kernels read dz twice.
__global__ __launch_bounds__(256) void lrelu_bwd_colsum_kernel(const float* __restrict__ y, int y_cstride, int y_coff, float* __restrict__ dy,
                                                               int dy_cstride, int dy_coff, int M, int C, float slope, int rows_per_block,
                                                               float* __restrict__ partial) {
  const int cg = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (c < C) {
    const float* yb = y + y_coff + c;
    float* gb = dy + dy_coff + c;
    auto one = [&](int m, float4& s) {
      const float4 yv = *reinterpret_cast<const float4*>(yb + (long)m * y_cstride);
      float4* g = reinterpret_cast<float4*>(gb + (long)m * dy_cstride);
      float4 gv = *g;
      gv.x *= yv.x > 0.f ? 1.f : slope;
      gv.y *= yv.y > 0.f ? 1.f : slope;
      gv.z *= yv.z > 0.f ? 1.f : slope;
      gv.w *= yv.w > 0.f ? 1.f : slope;
      *g = gv;
      s.x += gv.x; s.y += gv.y; s.z += gv.z; s.w += gv.w;
    };
    int m = r0 + rsub;
    for (; m + 16 < r1; m += 32) {
      one(m, s0);
      one(m + 16, s1);
    }
    for (; m < r1; m += 16) one(m, s0);
  }
  __shared__ float4 red[16][16];
  red[rsub][cg] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  if (threadIdx.x < 64) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    const float* r = reinterpret_cast<const float*>(&red[0][0]) + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += r[k * 64];
    if (col < C) partial[(long)blockIdx.y * C + col] = t;
  }
}

// workgroup = 64 columns x 4 partial-row groups; LDS fold
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int nsplit, int C, float* __restrict__ db,
                                                            int accumulate) {
  // 64 columns x 16 row groups, four independent partial sums per thread: the fold of up to 1024 partial rows is a chain of dependent
  // L2 round trips otherwise (4 groups x 1 sum: 25-40 us per call at 1024 rows)
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int k = rg;
    for (; k + 48 < nsplit; k += 64) {
      s0 += partial[(long)k * C + c];
      s1 += partial[(long)(k + 16) * C + c];
      s2 += partial[(long)(k + 32) * C + c];
      s3 += partial[(long)(k + 48) * C + c];
    }
    for (; k < nsplit; k += 16) s0 += partial[(long)k * C + c];
  }
  __shared__ float red[16][64];
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][cl];
    db[c] = accumulate ? db[c] + t : t;
  }
}

// dz = dy * (y > 0 ? 1 : slope) elementwise on a channel range of NHWC rows (LeakyReLU backward)
__global__ void lrelu_bwd_kernel(const float* __restrict__ y, int y_cstride, int y_coff, float* __restrict__ dy, int dy_cstride, int dy_coff,
                                 long M, int C, float slope) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = M * (C / 4);
  if (idx >= total) return;
  long m = idx / (C / 4);
  int c4 = (int)(idx % (C / 4)) * 4;
  float4 yv = *reinterpret_cast<const float4*>(y + m * y_cstride + y_coff + c4);
  float4* g = reinterpret_cast<float4*>(dy + m * dy_cstride + dy_coff + c4);
  float4 gv = *g;
  gv.x *= yv.x > 0.f ? 1.f : slope;
  gv.y *= yv.y > 0.f ? 1.f : slope;
  gv.z *= yv.z > 0.f ? 1.f : slope;
  gv.w *= yv.w > 0.f ? 1.f : slope;
  *g = gv;
}

}  // namespace dim

namespace dim {
// keep_slabs != nullptr: the split slabs stay in `workspace` (also for one split), unreduced; *keep_slabs = how many
int wgrad_launch(WgradArgs& a, float* dw_packed, float* workspace, int splits, int accumulate, void* stream, int* keep_slabs = nullptr);
// shapes the patch form of the bf16 kernel takes: 3x3 / stride 1 / pad 1 on maps large enough that 8 x 8 blocks waste little
static bool wgrad_patch_ok(const WgradArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 32 == 0 && a.Cout % 128 == 0 && a.chunks_per_plane == 0 &&
         a.H == a.Ho && a.W == a.Wo && a.Ho * a.Wo >= 1200;
}
}
using namespace dim;

extern "C" {

long dim_conv2d_wgrad_workspace_floats(int Cout, int Cin, int KH, int KW, int splits) {
  if (splits <= 1) return 0;
  long n = (Cin == 8) ? (long)((KH * KW + 3) / 4) * 32 * Cout : (long)KH * KW * Cin * Cout;
  return n * splits;
}

static int wgrad_args(WgradArgs& a, const float* x, const float* dz, int N, int H, int W, int Cin, int in_cstride, int Ho, int Wo, int Cout,
                      int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad) {
  DIM_REQUIRE(x && dz, "null pointer");
  DIM_REQUIRE(Cin == 8 || Cin % 32 == 0, "Cin must be 8 or a multiple of 32 (got %d)", Cin);
  DIM_REQUIRE(Cout % 64 == 0, "Cout must be a multiple of 64 (got %d)", Cout);
  DIM_REQUIRE((long)N * H * W * in_cstride < (1L << 29) && (long)N * Ho * Wo * dz_cstride < (1L << 29),
              "tensor too large for 32-bit byte offsets");
  DIM_REQUIRE(dz_cstride % 4 == 0 && dz_coff % 4 == 0 && in_cstride % 4 == 0, "channel strides / offsets must be multiples of 4");
  a = WgradArgs{};
  a.x = x; a.dz = dz;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.in_cstride = in_cstride; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.dz_cstride = dz_cstride; a.dz_coff = dz_coff; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.M = N * Ho * Wo;
  a.x_bytes = (unsigned)((long)N * H * W * in_cstride * 4);
  a.dz_bytes = (unsigned)((long)N * Ho * Wo * dz_cstride * 4);
  a.div_wo = make_fastdiv((unsigned)Wo);
  a.div_ho = make_fastdiv((unsigned)Ho);
  a.nchunks = (Cin == 8) ? (KH * KW + 3) / 4 : KH * KW * (Cin / 32);
  return DIM_OK;
}

// dw_packed (+)= wgrad(x, dz).  splits > 1: pixel range split through `workspace` (slabs) and summed deterministically.
int dim_conv2d_wgrad(const float* x, const float* dz, float* dw_packed, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                     int Ho, int Wo, int Cout, int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad, int splits,
                     int accumulate, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(dw_packed, "null pointer");
  WgradArgs a;
  int rc = wgrad_args(a, x, dz, N, H, W, Cin, in_cstride, Ho, Wo, Cout, dz_cstride, dz_coff, KH, KW, stride, pad);
  if (rc != DIM_OK) return rc;
  return wgrad_launch(a, dw_packed, workspace, splits, accumulate, stream);
}

// The weight gradient straight into the MXNet layout: the pixel-split slabs stay in `workspace` and ONE layout-converter launch sums
// them in slab order on its way to dw_oihw -- the same bits as dim_conv2d_wgrad[_bf16] + dim_conv2d_unpack_weight whenever those sum
// their slabs serially (fewer than 64 slabs, or large ones), without the packed intermediate's round trip and the reduce launch.
int dim_conv2d_wgrad_oihw(const float* x, const float* dz, float* dw_oihw, float* workspace, int N, int H, int W, int Cin, int in_cstride, int Ho,
                          int Wo, int Cout, int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad, int splits, int bf16_mfma,
                          int Cout_rows, float scale, int accumulate, void* stream) {
  DIM_REQUIRE(dw_oihw && workspace, "null pointer");
  DIM_REQUIRE(Cout_rows >= 1 && Cout_rows <= Cout, "Cout_rows must be in 1 .. Cout");
  const long slab = (Cin == 8) ? (long)((KH * KW + 3) / 4) * 32 * Cout : (long)KH * KW * Cin * Cout;
  if (N == 0) {
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(dw_oihw, 0, (size_t)Cout_rows * Cin * KH * KW * 4, as_stream(stream));
      if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    return DIM_OK;
  }
  WgradArgs a;
  int rc = wgrad_args(a, x, dz, N, H, W, Cin, in_cstride, Ho, Wo, Cout, dz_cstride, dz_coff, KH, KW, stride, pad);
  if (rc != DIM_OK) return rc;
  a.bf16 = bf16_mfma ? 1 : 0;
  int nslab = 0;
  rc = wgrad_launch(a, nullptr, workspace, splits, 0, stream, &nslab);
  if (rc != DIM_OK) return rc;
  const float* src = workspace;
  if (nslab >= 64 && slab / 4 < 65536) {  // many small slabs: the lane-parallel reduce (dim_splitk_reduce picks it), into the slot after them
    float* sum = workspace + (long)nslab * slab;
    rc = dim_splitk_reduce(workspace, nullptr, sum, slab / 4, 4, nslab, 1.0f, stream);
    if (rc != DIM_OK) return rc;
    src = sum;
    nslab = 1;
  }
  return conv2d_unpack_weight_slabs(src, nslab, slab, dw_oihw, Cout_rows, Cout, Cin, KH, KW, scale, accumulate, stream);
}

// The pixel-split count that makes the whole grid of dim_conv2d_wgrad_bf16 resident at once on a chip of n_cu compute units (one
// plan for every caller: the training executor sizes its slab workspace with it).  Gathered form: 3 workgroups of the 128-row kernel
// per CU (40 KB of LDS each), 4 of the 64-row one, a workgroup = 4 K chunks; patch form: 2 workgroups per CU (235 registers), a
// workgroup = one 32-channel slice x all taps, the split unit is an 8 x 8 pixel block.  Every split keeps at least 4 steps.
int dim_conv2d_wgrad_bf16_splits(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int n_cu) {
  if (N <= 0 || n_cu <= 0) return 1;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  WgradArgs a = {};
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  static const int patch_env = [] { const char* e = getenv("DIM_WGB_PATCH"); return e ? atoi(e) : 1; }();
  long tiles, steps, slots;
  if (patch_env && wgrad_patch_ok(a)) {
    tiles = (long)(Cin / 32) * (Cout / 128);
    steps = (long)N * ((Ho + 7) / 8) * ((Wo + 7) / 8);
    slots = 2L * n_cu;
  } else {
    const bool wide = Cout % 128 == 0;
    const long nchunks = Cin == 8 ? (KH * KW + 3) / 4 : (long)KH * KW * (Cin / 32);
    tiles = (nchunks + 3) / 4 * (wide ? Cout / 128 : Cout / 64);
    steps = ((long)N * Ho * Wo + 31) / 32;
    slots = (wide ? 3L : 4L) * n_cu;
  }
  long sp = slots / (tiles > 0 ? tiles : 1);
  if (sp > steps / 4) sp = steps / 4;
  return (int)(sp < 1 ? 1 : sp);
}

int dim_conv2d_wgrad_bf16(const float* x, const float* dz, float* dw_packed, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                          int Ho, int Wo, int Cout, int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad, int splits,
                          int accumulate, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(dw_packed, "null pointer");
  WgradArgs a;
  int rc = wgrad_args(a, x, dz, N, H, W, Cin, in_cstride, Ho, Wo, Cout, dz_cstride, dz_coff, KH, KW, stride, pad);
  if (rc != DIM_OK) return rc;
  a.bf16 = 1;
  return wgrad_launch(a, dw_packed, workspace, splits, accumulate, stream);
}

}  // extern "C"

namespace dim {

int wgrad_launch(WgradArgs& a, float* dw_packed, float* workspace, int splits, int accumulate, void* stream, int* keep_slabs) {
  const int Cin = a.Cin, Cout = a.Cout;
  a.nsteps = ceil_div(a.M, 32);
  if (splits < 1) splits = 1;
  if (splits > a.nsteps) splits = a.nsteps;
  a.steps_per_split = ceil_div(a.nsteps, splits);
  splits = ceil_div(a.nsteps, a.steps_per_split);
  DIM_REQUIRE((splits == 1 && !keep_slabs) || workspace, "split wgrad needs a workspace (dim_conv2d_wgrad_workspace_floats)");
  a.dw = (splits > 1 || keep_slabs) ? workspace : dw_packed;
  a.accumulate = accumulate;
  static const int xcd_env = [] { const char* e = getenv("DIM_WGRAD_XCD"); return e ? atoi(e) : 1; }();
  a.xcd = xcd_env;
  static const int dbg_env = [] { const char* e = getenv("DIM_WGB_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg_env;
  hipStream_t st = as_stream(stream);
  const bool nw4 = Cout % 128 == 0;
  static const int patch_env = [] { const char* e = getenv("DIM_WGB_PATCH"); return e ? atoi(e) : 1; }();
  if (a.bf16 && patch_env && wgrad_patch_ok(a)) {
    // patch form: the split unit is an 8 x 8 pixel block, the grid (Cin / 32, Cout / 128, splits)
    const int nblocks = a.N * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
    if (splits > nblocks) splits = nblocks;
    a.steps_per_split = ceil_div(nblocks, splits);
    splits = ceil_div(nblocks, a.steps_per_split);
    a.dw = (splits > 1 || keep_slabs) ? workspace : dw_packed;
    DIM_REQUIRE(!accumulate, "accumulate is not supported by the patch form");
    hipLaunchKernelGGL((conv_wgrad_bf16_patch_kernel<3>), dim3(Cin / 32, Cout / 128, splits), dim3(256), 0, st, a);
    int rcp = check_launch("conv_wgrad_bf16_patch");
    if (rcp != DIM_OK) return rcp;
    if (keep_slabs) { *keep_slabs = splits; return DIM_OK; }
    if (splits > 1) return dim_splitk_reduce(workspace, nullptr, dw_packed, (long)a.nchunks * Cout * 32 / 4, 4, splits, 1.0f, stream);
    return DIM_OK;
  }
  if (a.bf16) {
    const int nch = a.nchunks >= 4 ? 4 : a.nchunks >= 2 ? 2 : 1;  // chunks (x 32 packed columns) per workgroup sharing one dZ tile
    dim3 gridb(ceil_div(a.nchunks, nch), Cout / (nw4 ? 128 : 64), splits);
#define DIM_WGB_LAUNCH(NW, C8, NCH)                                                                              \
  do {                                                                                                           \
    if ((a.dbg & 32) && NW == 4 && !C8 && NCH == 4)   /* timing-only twin with 8-byte operand loads */           \
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<4, false, 4, 1>), gridb, dim3(256), 0, st, a);                  \
    else if ((a.dbg & 64) && NW == 4 && !C8 && NCH == 4)   /* ... with half as many 16-byte loads */             \
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<4, false, 4, 2>), gridb, dim3(256), 0, st, a);                  \
    else                                                                                                         \
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<NW, C8, NCH>), gridb, dim3(64 * NW), 0, st, a);                 \
  } while (0)
#define DIM_WGB_NCH(NW, C8) { if (nch == 4) DIM_WGB_LAUNCH(NW, C8, 4); else if (nch == 2) DIM_WGB_LAUNCH(NW, C8, 2); else DIM_WGB_LAUNCH(NW, C8, 1); }
    if (Cin == 8) { if (nw4) DIM_WGB_NCH(4, true) else DIM_WGB_NCH(2, true) }
    else { if (nw4) DIM_WGB_NCH(4, false) else DIM_WGB_NCH(2, false) }
#undef DIM_WGB_NCH
#undef DIM_WGB_LAUNCH
    int rcb = check_launch("conv_wgrad_bf16");
    if (rcb != DIM_OK) return rcb;
    if (keep_slabs) { *keep_slabs = splits; return DIM_OK; }
    if (splits > 1) {
      DIM_REQUIRE(!accumulate, "accumulate with splits > 1 is not supported");
      return dim_splitk_reduce(workspace, nullptr, dw_packed, (long)a.nchunks * Cout * 32 / 4, 4, splits, 1.0f, stream);
    }
    return DIM_OK;
  }
  const bool two = a.nchunks >= 2;  // two K chunks (64 packed columns) per workgroup share one dZ tile
  dim3 grid(two ? (a.nchunks + 1) / 2 : a.nchunks, Cout / (nw4 ? 128 : 64), splits);
#define DIM_WG_LAUNCH(NW, C8, NCH) hipLaunchKernelGGL((conv_wgrad_kernel<NW, C8, NCH>), grid, dim3(64 * NW), 0, st, a)
  if (Cin == 8) {
    if (nw4) { if (two) DIM_WG_LAUNCH(4, true, 2); else DIM_WG_LAUNCH(4, true, 1); }
    else { if (two) DIM_WG_LAUNCH(2, true, 2); else DIM_WG_LAUNCH(2, true, 1); }
  } else {
    if (nw4) { if (two) DIM_WG_LAUNCH(4, false, 2); else DIM_WG_LAUNCH(4, false, 1); }
    else { if (two) DIM_WG_LAUNCH(2, false, 2); else DIM_WG_LAUNCH(2, false, 1); }
  }
#undef DIM_WG_LAUNCH
  int rc = check_launch("conv_wgrad");
  if (rc != DIM_OK) return rc;
  if (keep_slabs) { *keep_slabs = splits; return DIM_OK; }
  if (splits > 1) {
    long n = (long)a.nchunks * Cout * 32;
    if (accumulate) {
      // slabs + existing value: treat dw itself as one more addend by summing into a temp is avoided -- reduce then add
      return set_err(DIM_ERR_ARG, "accumulate with splits > 1 is not supported (reduce first, then accumulate with splits == 1)");
    }
    return dim_splitk_reduce(workspace, nullptr, dw_packed, n / 4, 4, splits, 1.0f, stream);
  }
  return DIM_OK;
}

// The 36 plane products of a Winograd weight gradient, dM_p[k][co] = sum_t V[t][p][k] D[t][p][co], as ONE launch of the wgrad
// kernel: a 1x1 "convolution" over T tile-pixels with planes * K input channels whose dZ operand moves with the plane.
// dM comes out in the packed layout [p * K/32 + k/32][Cout][k%32].
int launch_wgrad_planes(const float* V, const float* D, float* dM_packed, float* slabs, int T, int K, int Cout, int planes, int splits,
                        hipStream_t st) {
  DIM_REQUIRE(K % 64 == 0 && Cout % 64 == 0, "winograd wgrad: K %% 64 == 0 and Cout %% 64 == 0 required");
  DIM_REQUIRE((long)T * planes * K < (1L << 29) && (long)T * planes * Cout < (1L << 29), "winograd wgrad: tensor too large for 32-bit byte offsets");
  WgradArgs a = {};
  a.x = V; a.dz = D;
  a.N = 1; a.H = 1; a.W = T; a.Cin = planes * K; a.in_cstride = planes * K; a.Ho = 1; a.Wo = T; a.Cout = Cout;
  a.dz_cstride = planes * Cout; a.dz_coff = 0; a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0;
  a.M = T;
  a.x_bytes = (unsigned)((long)T * planes * K * 4);
  a.dz_bytes = (unsigned)((long)T * planes * Cout * 4);
  a.div_wo = make_fastdiv((unsigned)T);
  a.div_ho = make_fastdiv(1u);
  a.nchunks = planes * (K / 32);
  a.chunks_per_plane = K / 32;
  a.dz_plane_stride = Cout;
  return wgrad_launch(a, dM_packed, slabs, splits, 0, st);
}

}  // namespace dim

extern "C" {

// rows per partial block: at least 512, and few enough blocks (<= 256 per column tile) that the final fold stays short
static int bias_rows_per_block(int M) {
  int r = ceil_div(M, 256);
  return r < 512 ? 512 : r;
}

long dim_bias_grad_workspace_floats(int M, int C) { return (long)ceil_div(M, bias_rows_per_block(M)) * C; }

int dim_bias_grad(const float* dz, float* db, float* workspace, int M, int C, int dz_cstride, int dz_coff, int accumulate, void* stream) {
  if (M == 0) return DIM_OK;
  DIM_REQUIRE(dz && db && workspace, "null pointer");
  DIM_REQUIRE(C % 4 == 0 && dz_cstride % 4 == 0 && dz_coff % 4 == 0, "bias_grad: channel counts/offsets must be multiples of 4");
  const int rpb = bias_rows_per_block(M);
  int nsplit = ceil_div(M, rpb);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(C, 64), nsplit), dim3(256), 0, st, dz, M, C, dz_cstride, dz_coff, rpb, workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, st, workspace, nsplit, C, db, accumulate);
  return check_launch("bias_grad");
}

int dim_lrelu_bwd_bias_grad(const float* y, int y_cstride, int y_coff, float* dy, int dy_cstride, int dy_coff, float* db, float* workspace,
                            int M, int C, float slope, int accumulate, void* stream) {
  if (M == 0) return DIM_OK;
  DIM_REQUIRE(y && dy && db && workspace, "null pointer");
  DIM_REQUIRE(C % 4 == 0 && y_cstride % 4 == 0 && dy_cstride % 4 == 0 && y_coff % 4 == 0 && dy_coff % 4 == 0, "multiples of 4 required");
  // more, shorter row blocks than dim_bias_grad: this pass also WRITES the map, so it wants every CU busy (<= 512 partial rows)
  int rpb = ceil_div(M, 512);
  if (rpb < 64) rpb = 64;
  const int nsplit = ceil_div(M, rpb);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(lrelu_bwd_colsum_kernel, dim3(ceil_div(C, 64), nsplit), dim3(256), 0, st, y, y_cstride, y_coff, dy, dy_cstride,
                     dy_coff, M, C, slope, rpb, workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, st, workspace, nsplit, C, db, accumulate);
  return check_launch("lrelu_bwd_bias_grad");
}

long dim_lrelu_bwd_bias_grad_workspace_floats(int M, int C) {
  int rpb = ceil_div(M, 512);
  if (rpb < 64) rpb = 64;
  return (long)ceil_div(M, rpb) * C;
}

int dim_lrelu_bwd(const float* y, int y_cstride, int y_coff, float* dy, int dy_cstride, int dy_coff, long M, int C, float slope,
                  void* stream) {
  if (M == 0) return DIM_OK;
  DIM_REQUIRE(y && dy, "null pointer");
  DIM_REQUIRE(C % 4 == 0 && y_cstride % 4 == 0 && dy_cstride % 4 == 0 && y_coff % 4 == 0 && dy_coff % 4 == 0, "multiples of 4 required");
  long total = M * (C / 4);
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), y, y_cstride, y_coff, dy, dy_cstride,
                     dy_coff, M, C, slope);
  return check_launch("lrelu_bwd");
}

}  // extern "C"
