// Shared helpers for libdeepim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/deepim_hip.h"

namespace dim {

constexpr int kWave = 64;

// thread-local message behind dim_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DIM_OK;
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// wgrad.hip: the plane products of a Winograd weight gradient, dM_p[k][co] = sum_t V[t][p][k] D[t][p][co], packed [p*K/32 + k/32][Cout][k%32]
int launch_wgrad_planes(const float* V, const float* D, float* dM_packed, float* slabs, int T, int K, int Cout, int planes, int splits,
                        hipStream_t st);
// wino_gemm.hip: the P plane GEMMs of a Winograd layer as one persistent stream-K launch.  V [T][P][K], U [P][K/32][Cout][32],
// M [T][P][Cout]; tile 4 = 128x128 workgroup tiles, otherwise 64x64.
int launch_wino_gemm(const float* V, const float* U, float* M, int T, int K, int Cout, int P, int tile, hipStream_t st);

// Unsigned division by a launch-time constant in 4 VALU ops (Granlund-Montgomery): q = (t + ((n - t) >> s1)) >> s2,
// t = mulhi(n, mul).  Exact for every 32-bit n and d >= 1.  Kernels that decode a pixel index every K step use it instead
// of the ~20-instruction integer division sequence.
struct FastDiv {
  unsigned mul, s1, s2, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l < 1 ? 0 : l - 1;
  return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
  unsigned t = __umulhi(n, f.mul);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 16-byte buffer load: per-lane byte offset `voff` (range-checked: 0xFFFFFFFF = out of range = zeros), wave-uniform byte
// offset `soff` (NOT range-checked)
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  float4 f;
  f.x = __uint_as_float(v.x);
  f.y = __uint_as_float(v.y);
  f.z = __uint_as_float(v.z);
  f.w = __uint_as_float(v.w);
  return f;
}
// the same with the non-temporal hint: operands that are streamed once (the V planes of a Winograd GEMM)
__device__ __forceinline__ float4 buf_load16_nt(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
  float4 f;
  f.x = __uint_as_float(v.x);
  f.y = __uint_as_float(v.y);
  f.z = __uint_as_float(v.z);
  f.w = __uint_as_float(v.w);
  return f;
}
#endif

// ---- plane GEMMs of the Winograd layers (wino_gemm.hip): launch plan, shared with the input-transform kernels of conv.hip, whose
// spare blocks zero the output tiles that two stream-K workgroups share (saves a launch per layer)
struct WGemmArgs {
  const float* V;
  const float* U;
  float* M;
  int T, K, Cout, P;
  int nch;       // K / 32
  int NTN;       // Cout / BN
  int per, rem;  // chunks per workgroup: per, +1 for the first rem workgroups
  unsigned v_bytes, u_bytes, m_bytes;
  FastDiv d_nch, d_P, d_NTN, d_MT;
  int G, BM, BN, tile;  // workgroups of the GEMM launch, its tile shape and id
  int MT;               // row tiles
  int plane_major;      // item order, see wcur_decode
};
int wino_gemm_plan(WGemmArgs* plan, const float* V, const float* U, float* M, int T, int K, int Cout, int P, int tile);
// zeroed = the shared tiles have been zeroed already (by the transform kernel that ran before): no separate zero launch
int wino_gemm_run(const WGemmArgs& plan, bool zeroed, hipStream_t st);

#ifdef __HIPCC__
// position in the flat chunk list.  Two item orders:
//   plane_major = 0   item = (mt * NTN + nt) * P + p    (first version: a workgroup's consecutive items walk the 36 planes, i.e. 36
//                     different weight sets of K x BN floats -- 19 MB for conv3 -- which no L2 holds: every item re-fetched its
//                     weights from the Infinity Cache, 717 MB per conv3 launch, more than V and M together)
//   plane_major = 1   item = (p * MT + mt) * NTN + nt   consecutive items are consecutive row tiles of ONE plane; together with the
//                     XCD-contiguous workgroup numbering below, the ~32 workgroups that share an L2 work inside 4-5 planes at any
//                     time (2-3 MB of weights), so a plane's weights leave the fabric once per XCD instead of once per item
struct WCur {
  int ch, p, nt, mt;
};
__device__ __forceinline__ WCur wcur_decode(int chunk, const WGemmArgs& a) {
  WCur c;
  const unsigned item = fastdiv((unsigned)chunk, a.d_nch);
  c.ch = chunk - (int)item * a.nch;
  if (a.plane_major) {
    const unsigned t = fastdiv(item, a.d_NTN);
    c.nt = (int)(item - t * a.NTN);
    const unsigned p = fastdiv(t, a.d_MT);
    c.mt = (int)(t - p * a.MT);
    c.p = (int)p;
  } else {
    const unsigned t = fastdiv(item, a.d_P);
    c.p = (int)(item - t * a.P);
    const unsigned mt = fastdiv(t, a.d_NTN);
    c.nt = (int)(t - mt * a.NTN);
    c.mt = (int)mt;
  }
  return c;
}
// workgroups are dealt round-robin to the 8 XCDs (blocks b and b + 8 share one L2): number them so that every XCD owns a CONTIGUOUS
// run of chunk ranges (bijective for any G: the first G % 8 XCDs own one range more)
__device__ __forceinline__ int wg_xcd_contiguous(int b, int G) {
  const int x = b & 7, j = b >> 3, q = G >> 3, r = G & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}
__device__ __forceinline__ int wg_first_chunk(int w, const WGemmArgs& a) { return w * a.per + min(w, a.rem); }
// one 256-thread block zeroes the output tile of the item that the range boundary in front of workgroup w (1 <= w < G) falls into
__device__ __forceinline__ void wino_gemm_zero_tile(const WGemmArgs& a, int w) {
  const WCur c = wcur_decode(wg_first_chunk(w, a), a);
  if (c.ch == 0) return;  // the boundary coincides with an item boundary
  const int ldc = a.P * a.Cout;
  float* base = a.M + (long)c.mt * a.BM * ldc + c.p * a.Cout + c.nt * a.BN;
  const int rows = min(a.BM, a.T - c.mt * a.BM);
  const int q = a.BN / 4;
  for (int idx = threadIdx.x; idx < rows * q; idx += 256) {
    const int r = idx / q, c4 = idx - r * q;
    *reinterpret_cast<float4*>(base + (long)r * ldc + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
#endif

}  // namespace dim

#define DIM_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return dim::set_err(DIM_ERR_ARG, __VA_ARGS__);  \
  } while (0)
