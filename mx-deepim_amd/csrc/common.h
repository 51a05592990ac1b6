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
#endif

}  // namespace dim

#define DIM_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return dim::set_err(DIM_ERR_ARG, __VA_ARGS__);  \
  } while (0)
