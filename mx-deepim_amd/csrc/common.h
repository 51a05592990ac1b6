// Shared helpers for libdeepim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <utility>

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

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
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

// Weight layout converters (MXNet arrays <-> the kernels' [chunk][n][32] arrays; ~100 MB each way per training iteration plus fc6's
// 84 MB) as LDS-tiled transposes.  A workgroup (bx, by) moves G sub-blocks of 32 (r) x Q (q) elements between
//   "rows" side:   rows[bx * rows_x + by * rows_y + g * sg + r * sr + q]     q contiguous (the taps of one (out, in) channel pair, or
//                                                                            fc6's pixels); g_fast says whether g or r is the next
//                                                                            contiguous index (sg < sr), which orders the load loop
//   "packed" side: nj runs of 32 G contiguous elements, run j at jbase[j] + bx * jx[j] + by * 32 G holding tap jq[j] (-1: zeros)
//                  -- or, table-free (nj == 0): run q at bx * packed_x + by * 32 G + q * dq for every q < Q
// through an LDS image with an odd lane stride on both sides.  Sub-blocks g >= gmax - by G and rows r >= rmax - 32 bx are padding:
// zeros on the packed side, untouched on the rows side.  The element-per-thread versions these replace gathered one side with a
// 36..320-byte lane stride and ran at 0.6-1.3 TB/s.  TO_PACKED: rows -> packed (PT = float or __bf16); else packed -> rows with
// rows = (accumulate ? rows : 0) + scale * packed.
struct WTileArgs {
  const void* src;
  void* dst;
  int G, Q, gmax, rmax, g_fast, nj;
  long sg, sr, rows_x, rows_y, dq, packed_x;
  float scale;
  int accumulate;
  int nslab;         // packed -> rows only: the packed side is the sum of nslab (0 = 1) arrays slab_stride floats apart, added in
  long slab_stride;  // slab order (the pixel-split slabs of a weight gradient: the separate reduce pass and its round trip folded in)
  int jq[32];
  long jbase[32], jx[32];
};
template <bool TO_PACKED, typename PT>
__global__ __launch_bounds__(256) void weight_tile_kernel(WTileArgs a) {
  extern __shared__ float wt_tile[];
  __shared__ int s_jq[32];
  __shared__ long s_jb[32];
  const int G = a.G, Q = a.Q, P = Q | 1, lg = 31 - __builtin_clz(G);  // G is a power of two
  const int bx = blockIdx.x, by = blockIdx.y;
  const bool table = a.nj > 0;
  const int nj = table ? a.nj : Q;
  if (table && threadIdx.x < a.nj) {
    s_jq[threadIdx.x] = a.jq[threadIdx.x];
    s_jb[threadIdx.x] = a.jbase[threadIdx.x] + bx * a.jx[threadIdx.x];
  }
  const int gvalid = min(G, a.gmax - by * G), rvalid = min(32, a.rmax - bx * 32);
  const long rbase = bx * a.rows_x + by * a.rows_y, pbase = (long)by * 32 * G;
  const int n_rows = G * 32 * Q, n_packed = nj * 32 * G;
  // LDS address of (g, r, q): the load order's (g, r) index times P, skewed by r when g is the fast one (lane stride G P + 1: odd)
#define WT_ADDR(g, r, q) (a.g_fast ? (((r) << lg) + (g)) * P + (q) + (r) : (((g) << 5) + (r)) * P + (q))
  // rows side: element i = u Q + q walked in steps of 256 without divisions, four elements per thread and pass so that four loads
  // are in flight (one load per pass left every workgroup at ~25 us whatever its size)
  const int step_q = 256 % Q, step_u = 256 / Q;
  int wq = threadIdx.x % Q, wu = threadIdx.x / Q;
#define WT_ROWS_BATCH                                                             \
    bool ok[4];                                                                    \
    long roff[4];                                                                  \
    int laddr[4];                                                                  \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                \
      const int g = a.g_fast ? wu & (G - 1) : wu >> 5, r = a.g_fast ? wu >> lg : wu & 31; \
      ok[e] = i0 + 256 * e < n_rows && g < gvalid && r < rvalid;                   \
      roff[e] = g * a.sg + r * a.sr + wq;                                          \
      laddr[e] = WT_ADDR(g, r, wq);                                                \
      wq += step_q; wu += step_u;                                                  \
      if (wq >= Q) { wq -= Q; ++wu; }                                              \
    }
  if (TO_PACKED) {
    const float* rows = reinterpret_cast<const float*>(a.src) + rbase;
    PT* packed = reinterpret_cast<PT*>(a.dst) + pbase;
    for (int i0 = threadIdx.x; i0 < n_rows; i0 += 1024) {
      WT_ROWS_BATCH
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = *(ok[e] ? rows + roff[e] : reinterpret_cast<const float*>(a.src));
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok[e]) wt_tile[laddr[e]] = v[e];
    }
    __syncthreads();
#pragma unroll 4
    for (int i = threadIdx.x; i < n_packed; i += 256) {
      const int t = i & (32 * G - 1), j = i >> (5 + lg), g = t >> 5, r = t & 31;
      const int q = table ? s_jq[j] : j;
      const long off = table ? s_jb[j] : bx * a.packed_x + j * a.dq;
      const bool okp = q >= 0 && g < gvalid && r < rvalid;
      const float v = wt_tile[okp ? WT_ADDR(g, r, q) : 0];
      packed[off + t] = (PT)(okp ? v : 0.f);
    }
  } else {
    float* rows = reinterpret_cast<float*>(a.dst) + rbase;
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < n_packed; i0 += 1024) {
      bool okp[4];
      int laddr[4];
      long poff[4];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + 256 * e;
        const int t = i & (32 * G - 1), j = min(i >> (5 + lg), nj - 1), g = t >> 5, r = t & 31;
        const int q = table ? s_jq[j] : j;
        const long off = table ? s_jb[j] : bx * a.packed_x + j * a.dq;
        okp[e] = i < n_packed && q >= 0 && g < gvalid && r < rvalid;
        laddr[e] = WT_ADDR(g, r, max(q, 0));
        poff[e] = okp[e] ? pbase + off + t : 0;
        v[e] = reinterpret_cast<const float*>(a.src)[poff[e]];
      }
      for (int sl = 1; sl < a.nslab; ++sl) {  // same order as splitk_reduce_kernel: ((s0 + s1) + s2) + ...
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = reinterpret_cast<const float*>(a.src)[poff[e] + (okp[e] ? sl * a.slab_stride : 0)];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (okp[e]) wt_tile[laddr[e]] = v[e];
    }
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < n_rows; i0 += 1024) {
      WT_ROWS_BATCH
      float old[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) old[e] = a.accumulate ? *(ok[e] ? rows + roff[e] : reinterpret_cast<float*>(a.dst)) : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok[e]) rows[roff[e]] = old[e] + a.scale * wt_tile[laddr[e]];
    }
  }
#undef WT_ROWS_BATCH
#undef WT_ADDR
}
inline size_t wtile_lds_bytes(int G, int Q) { return ((size_t)G * 32 * (Q | 1) + 32) * 4; }
// sub-blocks per workgroup: a power of two <= 8 that divides `count` and keeps the LDS image within 64 KB -- the largest one that
// still leaves >= 2048 workgroups (`outer` = the other grid dimension), else the smallest; 0 = none fits
inline int wtile_group(int count, int Q, int outer = 1) {
  int best = 0;
  for (int g = 1; g <= 8; g <<= 1) {
    if (count % g || wtile_lds_bytes(g, Q) > 65536) break;
    if (best == 0 || (long)outer * (count / g) >= 2048 || count / best > 65535) best = g;
  }
  return (best && count / best <= 65535 && outer <= 65535) ? best : 0;  // grid.y = count / G
}
template <bool TO_PACKED, typename PT>
inline void wtile_launch(const WTileArgs& a, int gx, int gy, hipStream_t st) {
  hipLaunchKernelGGL((weight_tile_kernel<TO_PACKED, PT>), dim3(gx, gy), dim3(256), wtile_lds_bytes(a.G, a.Q), st, a);
}
#endif

// train.hip: packed weight-gradient layout -> (Cout, Cin, KH, KW), the packed side given as nslab slabs to be summed on the way
int conv2d_unpack_weight_slabs(const float* w_packed, int nslab, long slab_stride, float* w_oihw, int Cout, int CoutPad, int Cin, int KH, int KW,
                               float scale, int accumulate, void* stream);

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
  int dbg_plain;        // timing-only experiment: every flush a plain store (wrong sums for shared items)
  int wide_flush;       // whole items leave through LDS as 16-byte stores (0: one-dword stores, A/B timing)
  const void* U3;       // the three-term bf16 image of U behind it in the same buffer (wino_gemm_split.hip)
  unsigned u3_bytes;
  int split;            // run wino_gemm_split_kernel (f32 operands as three bf16 terms, six MFMA products)
};
// wino_gemm_split.hip.  Every packed Winograd weight buffer is [U f32][U3]: floats -> floats * 5 / 2
inline long wino_packed_with_split(long u_floats) { return u_floats + (u_floats * 3 + 1) / 2; }
int wino_split_weights(float* U, long chunks, int Cout, hipStream_t st);
void wino_set_split(int on);
int wino_get_split();
bool wino_gemm_split_has(int tile);
int wino_gemm_split_slots(int tile);
int wino_gemm_split_run(const WGemmArgs& a, hipStream_t st);
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
