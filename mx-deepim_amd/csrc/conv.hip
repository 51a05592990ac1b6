// im2col-free direct convolution on the CDNA4 f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN Convolution / FullyConnected calls of the reference's FlowNetS encoder
// (/root/reference/deepim/symbols/deepIM_flownet.py:67-208).  Activations are NHWC fp32 in HBM,
// weights are pre-packed once into [K/32 chunks][Cout][32] (chunk order: 32-channel slice outer, taps inner); implicit GEMM
//     Y[m = (n,ho,wo)][co] = sum_k X[n, ho*s-p+kh, wo*s-p+kw, c] * Wp[k][co]
// with a 32-deep K chunk that is one tap x 32 channels (Cin % 32 == 0) or, for the 8-channel
// first layer, four consecutive taps (flat, row-major over the kernel window) x 8 channels.
// The 3x3 / stride-1 and 5x5 / stride-2 layers normally run in the Winograd domain instead (second half of this file + wino_gemm.hip).
//
// Block = 4 (or 8) waves; wave tile = (BM/WM) x (BN/WN) in 32x32 MFMA tiles.
// LDS: the A (pixel) chunk [BM][32+4], k-contiguous: staged with one ds_write_b128 per float4 and read back as ds_read_b128 =
//      four k-steps of MFMA operands per LDS instruction; two buffers, software-pipelined (see the loop).  The B operand
//      (weights) never passes through LDS: the packed layout is the fragment layout, every wave loads its own fragments from
//      L2 one chunk ahead (the k order inside a chunk is permuted identically for A and B).
// Epilogue: bias + LeakyReLU fused; with gridDim.z > 1 (split-K) raw partials go to a slab
// and dim_splitk_reduce finishes (deterministic, no atomics).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace dim {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;        // final output (splits == 1) or slab base (splits > 1)
  int N, H, W, Cin;
  int Ho, Wo, Cout;
  int KH, KW, stride, pad_h, pad_w;
  int M;           // N*Ho*Wo
  int nchunks;     // total K chunks of 32
  int chunks_per_split;
  float slope;     // LeakyReLU slope (1 = linear)
  int has_bias;
  // generalised addressing (decoder): input pixel stride, output row stride / channel offset (write into a concat buffer),
  // and an output scatter (oy,ox) = (ho*osy + ooy, wo*osx + oox) clipped to OH x OW (sub-pixel phases of a deconvolution + Crop)
  int in_cstride, out_cstride, out_coff;
  int dense_out, OH, OW, osy, osx, ooy, oox;
  int accumulate;  // out += v (final pass only)
  unsigned x_bytes, w_bytes;  // extents of x / w for the buffer descriptors (loads past them return 0)
  unsigned y_bytes;           // extent of one output problem (kernels that store through a descriptor: tile 9)
  int xcd_chunk;   // > 0: workgroup id -> tile remap that keeps consecutive tiles on one XCD (see conv_fwd_kernel)
  FastDiv div_kw;  // 8-channel layer: flat tap index -> (kh, kw)
  int boy, box;    // batched launch with scattered output: problem b lands at (ooy + (b >> 1) boy, oox + (b & 1) box) (deconv phases)
  long bx, bw, by; // batched launch (gridDim.y > 1): element strides of x / w / y between the problems (Winograd: 16 GEMMs)
  int tile_off;    // first tile of this launch (tail launch of an "auto" workload)
  int slab_row0;   // split-K slabs hold rows [slab_row0, M)
  long slab_stride;  // elements between the slabs of consecutive splits
  int bf16;          // weights are packed bf16, products on v_mfma_f32_32x32x16_bf16 (conv_bf16_kernel)
  int slab_full;     // split-K slabs are whole copies of the OUTPUT tensor (its channel stride, offset and scatter): the partial
                     // results of a strided / scattered launch (input-gradient phases) land where the final values go, slab by slab
  // tile 9 as the input gradient of a layer whose INPUT went through LeakyReLU: out = v * (mask > 0 ? 1 : mask_slope) with `mask` laid
  // out like the output (the stored activation), and every wave's column sums over its pixels -> colsum[(colsum_row0 + 2 block + wm)]
  // [Cout] (the bias gradient of that layer after one small reduce): the separate LeakyReLU' + bias-gradient pass folded in
  const float* mask;
  float mask_slope;
  float* colsum;
  int colsum_row0;
};

// Epilogue of the gathered-tap kernels (f32 and bf16): bias + LeakyReLU (+ accumulate) and the store of a wave's TM x TN accumulator
// tiles.  D layout: col = lane & 31 -> output channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> GEMM row of the tile.
// Every store goes through a buffer descriptor and a row outside the output gets byte offset 0xFFFFFFFF, which the range check drops:
// no branches.  With a per-row `if` hipcc opens each block with `s_waitcnt vmcnt(0)` (the bias load is still "pending" across the
// block boundary), and on gfx950 vmcnt also counts the stores -- the wave's 32 .. 128 stores then leave one round trip (~0.2 us) at a
// time.  Round 1 gave the f32 full-tile path its own branch-free loop for that reason; the partial tiles, the scattered output of the
// deconvolution / strided-gradient phases, the accumulate path and the whole bf16 twin still paid it (6 us per workgroup of a bf16
// layer whose main loop is 7 us).
template <int TM, int TN>
__device__ __forceinline__ void conv_store_tiles(const ConvArgs& a, const f32x16 (&acc)[TM][TN], float* yb, int mrow, int ncol, int split) {
  const bool final = gridDim.z == 1;
  const bool shaped = final || a.slab_full;   // addressed like the output tensor itself
  const int ldc = shaped ? a.out_cstride : a.Cout;
  float* base = final ? yb : yb + (long)split * a.slab_stride;
  const unsigned extent = shaped ? a.y_bytes : (unsigned)((long)(a.M - a.slab_row0) * a.Cout * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(base, 0, extent, 0x00020000);
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = (final && a.has_bias) ? a.bias[ncol + 32 * j] : 0.f;
  const float slope = final ? a.slope : 1.0f;
  int voff[TM][16];   // byte offset of the row's channel ncol (tile j: + 128 j bytes), -1 = not stored
  if (!shaped || a.dense_out) {
    const int col_b = ((shaped ? a.out_coff : 0) + ncol) * 4, row0 = shaped ? 0 : a.slab_row0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow + 32 * i + (r & 3) + 8 * (r >> 2);
        voff[i][r] = m < a.M ? (m - row0) * (ldc * 4) + col_b : -1;
      }
  } else {
    // scattered output (deconvolution phase + Crop, strided-gradient phase): row m = (n, ho, wo) lands at (n, ho*osy+ooy, wo*osx+oox)
    // if that is inside OH x OW
    const int oyb = a.ooy + (int)(blockIdx.y >> 1) * a.boy, oxb = a.oox + (int)(blockIdx.y & 1) * a.box;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow + 32 * i + (r & 3) + 8 * (r >> 2);
        const int mm = m < a.M ? m : 0;
        const int wo = mm % a.Wo, t = mm / a.Wo;
        const int ho = t % a.Ho, n = t / a.Ho;
        const int oy = ho * a.osy + oyb, ox = wo * a.osx + oxb;
        const bool ok = m < a.M && (unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW;
        voff[i][r] = ok ? (((n * a.OH + oy) * a.OW + ox) * ldc + a.out_coff + ncol) * 4 : -1;
      }
  }
  if (final && a.accumulate) {   // wave-uniform: out += result
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, voff[i][r], 128 * j, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[j];
          v = (v > 0.f ? v : v * slope) + old[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff[i][r], 128 * j, 0);
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[j];
          v = v > 0.f ? v : v * slope;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff[i][r], 128 * j, 0);
        }
  }
}

template <int BM, int BN, int WM, int WN, bool CIN8>
__global__ __launch_bounds__(WM * WN * 64) void conv_fwd_kernel(ConvArgs a) {
  constexpr int BK = 32;
  constexpr int NT = WM * WN * 64;  // 4 or 8 waves
  constexpr int RP = NT / 8;        // rows staged per pass (8 threads x float4 = one 32-float row)
  constexpr int LDK = BK + 4;       // row stride (floats): 16 rows x 4 dwords hit 16 distinct 4-bank slots for ds_read_b128
  constexpr int TM = BM / WM / 32;  // MFMA tiles per wave along M
  constexpr int TN = BN / WN / 32;
  constexpr int A_PER_T = BM / RP;  // float4 loads per thread for the A chunk
  static_assert((WM * WN == 4 || WM * WN == 8) && A_PER_T >= 1 && A_PER_T <= 4, "staging plan");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                       // [2][BM][LDK]   pixel-major, k contiguous

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // 1-D tile grid, N tiles fastest: the Cout/BN workgroups that share one A (pixel) tile are adjacent.  Workgroups are
  // dealt round-robin to the 8 XCDs (each with its own L2), so with xcd_chunk = tiles/8 the id is remapped such that XCD k
  // walks tiles [k*chunk, (k+1)*chunk) in order: the A tile is fetched into ONE L2 and re-used there by its N tiles, and
  // neighbouring pixel tiles (which share the 3x3 halo rows) follow on the same XCD.
  int id = blockIdx.x + a.tile_off;
  if (a.xcd_chunk > 0) id = (id & 7) * a.xcd_chunk + (id >> 3);
  const int ntiles_n = a.Cout / BN;
  const int mtile = id / ntiles_n;
  const int m0 = mtile * BM;
  const int n0 = (id - mtile * ntiles_n) * BN;
  const int split = blockIdx.z;
  const int kc_begin = split * a.chunks_per_split;
  const int kc_end = min(a.nchunks, kc_begin + a.chunks_per_split);

  // ---- per-thread staging descriptors: thread (q, srow) moves float4 #q of row srow (+32 per pass) for A and for B
  const int q = tid & 7;
  const int srow = tid >> 3;
  int a_hi0[A_PER_T], a_wi0[A_PER_T], a_pix[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    int m = m0 + srow + RP * i;
    bool ok = m < a.M;
    int mm = ok ? m : 0;
    int wo = mm % a.Wo;
    int t = mm / a.Wo;
    int ho = t % a.Ho;
    int n = t / a.Ho;
    a_hi0[i] = ok ? ho * a.stride - a.pad_h : -(1 << 28);  // rows past M: every tap fails the bounds test
    a_wi0[i] = wo * a.stride - a.pad_w;
    // BYTE offset of this thread's float4 at tap (0,0), channel 0 (may be negative in the padding; 32 bits, host-checked).
    // 8-channel layer: a chunk is 4 taps x 8 channels, thread q holds channel half q & 1 of tap q >> 1 (tap offset added per chunk).
    a_pix[i] = ((n * a.H + (ok ? a_hi0[i] : 0)) * a.W * a.in_cstride + (wo * a.stride - a.pad_w) * a.in_cstride + (CIN8 ? (q & 1) * 4 : q * 4)) * 4;
  }
  // Both operands go through buffer descriptors: a padding tap is a load at offset 0xFFFFFFFF (the range check returns
  // zeros: one v_cndmask on a 32-bit offset, no pointer select, no exec juggling, and -- unlike the flat loads a pointer
  // select compiles to -- nothing that counts on lgkmcnt next to the LDS fragment reads); the weights take the chunk as
  // a scalar offset, so their per-lane offset is loop-invariant.
  const float* xb = a.x + (long)blockIdx.y * a.bx;  // blockIdx.y = problem of a batched launch (0 otherwise)
  const float* wb = a.w + (long)blockIdx.y * a.bw;
  float* yb = a.y + (long)blockIdx.y * a.by;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb), 0, a.w_bytes, 0x00020000);
  const int wchunk_bytes = a.Cout * BK * 4;  // packed [chunk][Cout][32]

  // chunk -> (kh, kw, c0) counters
  int kh, kw, c0;
  if (CIN8) {
    // 8-channel layer: K = (tap, channel) flattened, 4 taps per chunk, taps numbered row-major over KH x KW with NO padding per
    // kernel row (7x7: 49 taps = 13 chunks instead of the 14 that "two chunks per row" needed).  `kw` holds this thread's flat tap.
    kw = 4 * kc_begin + (q >> 1);
    kh = 0;
    c0 = 0;
  } else {
    // K order for Cin % 32 == 0: 32-channel slice OUTER, taps INNER -- consecutive chunks read the same channels at the
    // (kh,kw)-shifted pixels, i.e. mostly the same cache lines (reuse distance 1 chunk instead of Cin/32 chunks).
    // With taps outer the L2 hit rate of conv3_1 was 50 % (rocprofv3 TCC_HIT/TCC_MISS): every tap re-fetched its
    // activations from beyond L2.
    int taps = a.KH * a.KW;
    int cc = kc_begin / taps;
    int tap = kc_begin - cc * taps;
    c0 = cc << 5;
    kh = tap / a.KW;
    kw = tap - kh * a.KW;
  }

  float4 ra0, ra1, ra2, ra3;  // staging registers of the A chunk, named (arrays + lambdas ended up in scratch)

  // tap_off is wave-uniform (scalar): one vector add per load.  (It cannot ride in the instruction's scalar offset: that
  // one is excluded from the range check, and a_pix alone is negative = out of range in the top/left padding.)
#define DIM_LOAD_A(REG, I)                                                                                         \
  if (I < A_PER_T) {                                                                                                \
    bool ok = pf_ok && (unsigned)(a_hi0[I] + tkh) < (unsigned)a.H && (unsigned)(a_wi0[I] + tkw) < (unsigned)a.W;    \
    REG = buf_load16(rx, ok ? a_pix[I] + tap_off : -1, 0);                                                          \
  }
  // PF_OK = false on the one prefetch past the last chunk: its (kh,kw,c0) counters already point one channel slice beyond
  // the tensor, so the (unused) activation read is dropped like a padding tap
#define DIM_LOAD_CHUNK(PF_OK)                                      \
  {                                                                \
    /* (tkh, tkw) = the tap this thread loads: wave-uniform for the 32-channel layers, per thread (from its flat tap) for the */ \
    /* 8-channel one, where a tap past KH*KW lands on a row >= KH only if the bounds test below rejects it explicitly */ \
    const int tkh = CIN8 ? (int)fastdiv((unsigned)kw, a.div_kw) : kh;                                   \
    const int tkw = CIN8 ? kw - tkh * a.KW : kw;                                                        \
    const bool pf_ok = (PF_OK) && (!CIN8 || tkh < a.KH);           \
    const int tap_off = ((tkh * a.W + tkw) * a.in_cstride + c0) * 4; \
    DIM_LOAD_A(ra0, 0) DIM_LOAD_A(ra1, 1) DIM_LOAD_A(ra2, 2) DIM_LOAD_A(ra3, 3) \
  }
#define DIM_ADVANCE()                        \
  if (CIN8) {                                \
    kw += 4;                                 \
  } else {                                   \
    if (++kw == a.KW) {                      \
      kw = 0;                                \
      if (++kh == a.KH) { kh = 0; c0 += 32; } \
    }                                        \
  }
#define DIM_STORE_A(REG, I) \
  if (I < A_PER_T) *reinterpret_cast<float4*>(dA + (srow + RP * I) * LDK + q * 4) = REG;
#define DIM_STORE_CHUNK(BUF)                          \
  {                                                   \
    float* dA = sA + (BUF) * BM * LDK;                \
    DIM_STORE_A(ra0, 0) DIM_STORE_A(ra1, 1) DIM_STORE_A(ra2, 2) DIM_STORE_A(ra3, 3) \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: lane (half h, row r) reads 4 consecutive k = 8s + 4h + {0..3} of its row with one ds_read_b128;
  // MFMA #j of group s then contracts k = 8s + j (lanes 0-31) and k = 8s + 4 + j (lanes 32-63): every k of the chunk
  // is used exactly once, identically for A and B.
  const int frow = lane & 31;
  const int khalf = lane >> 5;
  const int a_off = (wm * (BM / WM) + frow) * LDK + 4 * khalf;
  float4 fa[2][TM];
#define DIM_FRAG_READ(IDX, PA, PB, S)                                                                  \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[IDX][i] = *reinterpret_cast<const float4*>((PA) + 32 * i * LDK + 8 * (S)); \
  }
  // the weights never pass through LDS: every wave fetches its own B fragments (lane (row n, k half) = 16 contiguous bytes of the
  // packed [chunk][Cout][32] array) one whole chunk ahead; fbq[set][group][tile]
  float4 fbq[2][4][TN];
  const int bf_voff = ((n0 + wn * (BN / WN) + frow) * BK + 4 * khalf) * 4;
#define DIM_LOAD_BFRAG(SET, KC)                                                                         \
  {                                                                                                    \
    const int bsoff = (KC) * wchunk_bytes;                                                             \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int j = 0; j < TN; ++j)        \
      fbq[SET][g][j] = buf_load16(rw, bf_voff + (32 * j * BK + 8 * g) * 4, bsoff);                      \
  }
#define DIM_MFMA_GROUP(IDX, SET, G)                                                                     \
  {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) {     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].x, fbq[SET][G][j].x, acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].y, fbq[SET][G][j].y, acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].z, fbq[SET][G][j].z, acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].w, fbq[SET][G][j].w, acc[i][j], 0, 0, 0); \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }

  // ---- software pipeline (per K chunk of 32 = four MFMA groups g0..g3):
  //   g0 | g1 | [registers -> LDS for chunk k+1, then global loads for chunk k+2] | g2 | barrier | [fragments g0 of chunk k+1] | g3
  // The LDS stores and the barrier sit INSIDE the MFMA sequence and the next chunk's first fragments are in flight during g3, so
  // no wave ever reaches a point with nothing to feed the MFMA pipe.  (With stores + barrier + first fragment read at the chunk
  // boundary the four workgroups of a CU ran in lock step and the pipe idled ~20 % of the time: PMC 74-78 % MFMA-busy.)
  // The barrier only orders LDS traffic (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait for the global
  // prefetch that has just been issued.  Hazards: buffer b^1 is written in the middle of chunk k; its last readers were the g3
  // fragments of chunk k-1, which every wave has in registers before it passes that chunk's barrier.
  if (kc_begin < kc_end) {
    DIM_LOAD_CHUNK(true)
    DIM_ADVANCE()
    DIM_STORE_CHUNK(0)
    DIM_LOAD_BFRAG(0, kc_begin)
  }
  __syncthreads();
  DIM_LOAD_CHUNK(kc_begin + 1 < kc_end)
  DIM_ADVANCE()
  DIM_FRAG_READ(0, sA + a_off, 0, 0)

#define DIM_CHUNK_BODY(SET, KCUR)                                                    \
  {                                                                                  \
    const float* cA = sA + buf * BM * LDK + a_off;                                   \
    const float* nA = sA + (buf ^ 1) * BM * LDK + a_off;                             \
    DIM_LOAD_BFRAG(1 - SET, min((KCUR) + 1, a.nchunks - 1))                          \
    DIM_FRAG_READ(1, cA, 0, 1)                                                      \
    DIM_MFMA_GROUP(0, SET, 0)                                                        \
    DIM_FRAG_READ(0, cA, 0, 2)                                                      \
    DIM_MFMA_GROUP(1, SET, 1)                                                        \
    DIM_STORE_CHUNK(buf ^ 1)                                                         \
    DIM_LOAD_CHUNK((KCUR) + 2 < kc_end)                                              \
    DIM_ADVANCE()                                                                    \
    DIM_FRAG_READ(1, cA, 0, 3)                                                      \
    DIM_MFMA_GROUP(0, SET, 2)                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                  \
    DIM_FRAG_READ(0, nA, 0, 0)                                                      \
    DIM_MFMA_GROUP(1, SET, 3)                                                        \
    buf ^= 1;                                                                        \
  }
  int buf = 0;
  for (int kc = kc_begin; kc < kc_end; kc += 2) {
    DIM_CHUNK_BODY(0, kc)
    if (kc + 1 < kc_end) DIM_CHUNK_BODY(1, kc + 1)
  }
#undef DIM_CHUNK_BODY
#undef DIM_LOAD_BFRAG
#undef DIM_FRAG_READ
#undef DIM_MFMA_GROUP
#undef DIM_LOAD_A
#undef DIM_LOAD_CHUNK
#undef DIM_ADVANCE
#undef DIM_STORE_A
#undef DIM_STORE_CHUNK

  // ---- epilogue (conv_store_tiles above: branch-free buffer stores)
  conv_store_tiles<TM, TN>(a, acc, yb, m0 + wm * (BM / WM) + 4 * khalf, n0 + wn * (BN / WN) + frow, split);
}

// ---------------------------------------------------------------------------------------------------------------- first layer, LDS halo
// flow_conv1 (8 channels, 7x7 / stride 2 / pad 3 -> 64 channels; deepIM_flownet.py:67-75) from an LDS-resident input patch.
// In conv_fwd_kernel<.., CIN8> this layer was the furthest below its roof (0.60 ms = 103 TFLOP/s at B = 16): only 13 K chunks per
// workgroup, so the pipeline fill (first gathered loads -> LDS -> barrier) and drain cost ~14 %, another 6 % went into the K padding
// 392 -> 416, and the 49 taps re-gathered the input 3.7x from beyond L2.  Here a workgroup owns an 8 x 16 block of output pixels x
// all 64 output channels: it loads the 21 x 37 x 8 input patch ONCE (zero outside the image = the padding), every wave then reads
// its A fragments for all 49 taps from LDS at shifted addresses (ds_read_b128 with an immediate offset per tap) and runs 392 MFMAs
// without another barrier or global activation load.  Four workgroups fit a CU (37 KB of LDS each), so one workgroup's patch load
// and epilogue hide under the MFMAs of the others.  Weights: the packed [chunk][64][32] array of dim_conv2d_pack_weight as it is
// (a chunk = 4 flat taps x 8 channels, so tap t's 8 channels of an output channel are 32 contiguous bytes), fetched per tap from L2
// one tap ahead.  K is exactly 392.  Same products, same f32 accumulation chain per output as the direct kernel (k order differs).
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv1_halo_kernel(ConvArgs a) {
  constexpr int TH = 8, TW = 16;                 // output pixels per workgroup: 4 waves x (2 rows x 16)
  constexpr int S = 2;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;   // 21 x 37 input pixels
  constexpr int PS = 12;                         // floats per patch pixel: 8 channels + 4 pad (48 B: 2-way instead of 4-way conflicts)
  constexpr int NPIX = PH * PW;
  __shared__ __attribute__((aligned(16))) float patch[NPIX * PS];
  // the bias through LDS: its address depends on the lane half, so `a.bias[...]` in the epilogue is a VECTOR load, and the wait for it
  // (one in-order counter for vector loads and stores on this chip) also waits for the stores issued just before: four store round
  // trips per workgroup in series
  __shared__ __attribute__((aligned(16))) float sbias[64];
  if (threadIdx.x < 64) sbias[threadIdx.x] = a.has_bias ? a.bias[threadIdx.x] : 0.f;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
  int id = wg_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // neighbouring tiles (shared halo) on one XCD
  const int twi = id % tiles_w;
  id /= tiles_w;
  const int thi = id % tiles_h;
  const int n = id / tiles_h;
  const int ho0 = thi * TH, wo0 = twi * TW;
  const int hi0 = ho0 * S - a.pad_h, wi0 = wo0 * S - a.pad_w;

  // ---- patch: 2 float4 per pixel; out-of-image pixels read zeros through the descriptor's range check
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  constexpr int ITEMS = (NPIX * 2 + 255) / 256;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = it * 256 + tid;
    if (item < NPIX * 2) {
      const int pix = item >> 1, half = item & 1;
      const int py = pix / PW, px = pix - py * PW;
      const int hi = hi0 + py, wi = wi0 + px;
      const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const float4 v = buf_load16(rx, ok ? (((n * a.H + hi) * a.W + wi) * a.in_cstride + half * 4) * 4 : -1, 0);
      *reinterpret_cast<float4*>(&patch[pix * PS + half * 4]) = v;
    }
  }
  // ---- fragments.  The WEIGHTS are the MFMA's A operand (rows = output channels) and the pixels its B operand (columns), so a
  // lane ends up with 4 consecutive output channels of ONE pixel per accumulator quad: the epilogue is 8 float4 stores per lane
  // instead of 32 scalar ones
  const int frow = lane & 31, khalf = lane >> 5;
  const int p = wave * 32 + frow;               // output pixel = this lane's B column inside the block
  const int ty = p / TW, tx = p - ty * TW;
  const float* abase = &patch[((ty * S) * PW + tx * S) * PS + 4 * khalf];
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  // weight fragment of tap t, output-channel tile j: 16 bytes at ((t / 4) * 64 + 32 j + frow) * 32 + (t % 4) * 8 + 4 khalf floats
  const int b_voff = (frow * 32 + 4 * khalf) * 4;
  auto load_b = [&](int t, float4& b0, float4& b1) {
    const int soff = ((t >> 2) * 64 * 32 + (t & 3) * 8) * 4;
    b0 = buf_load16(rw, b_voff, soff);
    b1 = buf_load16(rw, b_voff + 32 * 32 * 4, soff);
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float4 b0, b1, nb0, nb1;
  load_b(0, b0, b1);
  __syncthreads();
  float4 fa = *reinterpret_cast<const float4*>(abase);
  for (int kh = 0; kh < KH; ++kh) {
    const float* arow = abase + kh * PW * PS;
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int t = kh * KW + kw;
      // next tap's operands in flight while this tap multiplies (the one past the end re-reads tap 0: in range, unused)
      const int tn = (t + 1 < KH * KW) ? t + 1 : 0;
      load_b(tn, nb0, nb1);
      const float* anext = (kw + 1 < KW) ? arow + (kw + 1) * PS : ((kh + 1 < KH) ? arow + PW * PS : abase);
      const float4 nfa = *reinterpret_cast<const float4*>(anext);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.x, fa.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.x, fa.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.y, fa.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.y, fa.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.z, fa.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.z, fa.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.w, fa.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.w, fa.w, acc1, 0, 0, 0);
      fa = nfa;
      b0 = nb0;
      b1 = nb1;
    }
  }
  // ---- epilogue.  D layout: col = lane & 31 -> pixel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> output channel (+ 32 for acc1)
  const int oy = ho0 + ty, ox = wo0 + tx;
  if (oy < a.Ho && ox < a.Wo) {
    float* o = a.y + a.out_coff + ((long)(n * a.Ho + oy) * a.Wo + ox) * a.out_cstride + 4 * khalf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv0 = *reinterpret_cast<const float4*>(&sbias[8 * g + 4 * khalf]);
      const float4 bv1 = *reinterpret_cast<const float4*>(&sbias[32 + 8 * g + 4 * khalf]);
      float4 v0 = make_float4(acc0[4 * g] + bv0.x, acc0[4 * g + 1] + bv0.y, acc0[4 * g + 2] + bv0.z, acc0[4 * g + 3] + bv0.w);
      float4 v1 = make_float4(acc1[4 * g] + bv1.x, acc1[4 * g + 1] + bv1.y, acc1[4 * g + 2] + bv1.z, acc1[4 * g + 3] + bv1.w);
      v0.x = v0.x > 0.f ? v0.x : v0.x * a.slope; v0.y = v0.y > 0.f ? v0.y : v0.y * a.slope;
      v0.z = v0.z > 0.f ? v0.z : v0.z * a.slope; v0.w = v0.w > 0.f ? v0.w : v0.w * a.slope;
      v1.x = v1.x > 0.f ? v1.x : v1.x * a.slope; v1.y = v1.y > 0.f ? v1.y : v1.y * a.slope;
      v1.z = v1.z > 0.f ? v1.z : v1.z * a.slope; v1.w = v1.w > 0.f ? v1.w : v1.w * a.slope;
      *reinterpret_cast<float4*>(o + 8 * g) = v0;
      *reinterpret_cast<float4*>(o + 32 + 8 * g) = v1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- bf16 MFMA
// The same implicit GEMM on v_mfma_f32_32x32x16_bf16 (16x the f32 matrix rate, f32 accumulate): the training mode of BASELINE
// configs[2].  Activations stay fp32 in HBM (every other kernel of the graph reads them); a thread rounds its float4 to four bf16
// (v_cvt_pk_bf16_f32, round to nearest even) on the way into LDS, so the LDS traffic and the fragment reads halve.  Weights are the
// SAME packed [chunk][Cout][32] arrays converted element-wise to bf16 (dim_f32_to_bf16): a lane's B fragment of k-step s is the 16
// contiguous bytes k = 16 s + 8 h + {0..7} of its output channel -- the operand map of the instruction -- straight from L2.
// With the matrix pipe 16x faster every layer is bound by its operand traffic (L2 -> LDS for the gathered A tile): the loop is a
// plain two-buffer pipeline, and occupancy (<= 64 VGPRs at the 64x32 wave tile) does the latency hiding.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v) {
  bf16x4 p = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return p;
}

// flow_conv1 on the bf16 pipe: PERSISTENT workgroups, the whole weight array and the input patches in LDS.  With 16x the matrix rate
// this layer is pure HBM traffic -- 157 MB of input, 315 MB of output at B = 16 = ~95 us -- and the gathered-tap kernel
// (conv_bf16_kernel<64,64,2,2,true>: 13 K chunks per workgroup, the 49 taps re-gathered 3.7x through the 64 B/clk vector memory path)
// took 0.28 ms.  A one-to-one twin of conv1_halo_kernel (weights per tap from L2, one block per workgroup) took 0.25 ms: without f32
// MFMAs to hide under, every wave streaming the 53 KB of weights from L2 (1.9 GB per launch) is the bound.  So: one 8-wave workgroup per
// CU keeps the bf16 image of the packed [chunk][64][32] weights in LDS (80-byte rows: conflict-free ds_read_b128) and walks a
// contiguous range of 16 x 16 pixel blocks; a block's 37 x 37 x 8 patch is loaded ONCE, rounded to bf16 on the way into LDS (16 B per
// pixel, two buffers), the next block's loads are in flight while this one multiplies, one barrier per block.  Roles as in
// conv1_halo_kernel (weights = the A operand, pixels = B: 8 float4 stores per lane); one v_mfma_f32_32x32x16_bf16 multiplies TWO taps:
// lane half h supplies tap 2 i + h, for the weights the 16 bytes k = 16 s + 8 h + {0..7} of chunk i / 2 (tap 49 = the zero padding
// of chunk 12; its pixel operand re-reads tap 48: finite, multiplied by zero).
template <int KH, int KW>
__global__ __launch_bounds__(512) void conv1_halo_bf16_kernel(ConvArgs a, int tiles, int per_wg) {
  constexpr int TH = 16, TW = 16, S = 2;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;   // 37 x 37 input pixels
  constexpr int NPIX = PH * PW, NT = KH * KW, NPAIR = (NT + 1) / 2, NCH = (NT + 3) / 4;
  constexpr int WROW = 40;                                         // bf16 elements per weight row in LDS (32 + 8 pad)
  constexpr int ITEMS = (NPIX + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* sw = reinterpret_cast<__bf16*>(smem);                    // [NCH * 64][WROW]
  bf16x8* patch = reinterpret_cast<bf16x8*>(sw + NCH * 64 * WROW);  // [2][NPIX]
  // the bias too: a vector load in the epilogue would sit behind the next block's patch loads and this block's stores in the one
  // in-order vector-memory counter (measured: 0.45 ms for the layer, every block waiting for its own stores to land)
  __shared__ __attribute__((aligned(16))) float sbias[64];
  if (threadIdx.x < 64) sbias[threadIdx.x] = a.has_bias ? a.bias[threadIdx.x] : 0.f;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = wg_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // neighbouring block ranges (shared halos) on one XCD
  const int t_begin = wg * per_wg, t_end = min(tiles, t_begin + per_wg);
  if (t_begin >= t_end) return;
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;

  // ---- weights -> LDS, once: rows of 64 bytes, four 16-byte pieces each
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  for (int it = tid; it < NCH * 64 * 4; it += 512) {
    const int row = it >> 2, piece = it & 3;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (row * 32 + piece * 8) * 2, 0, 0);
    *reinterpret_cast<u32x4*>(sw + row * WROW + piece * 8) = v;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  float4 lo[ITEMS], hi[ITEMS];
  auto tile_origin = [&](int t, int& n, int& ho0, int& wo0) {
    const int twi = t % tiles_w;
    const int r = t / tiles_w;
    n = r / tiles_h;
    ho0 = (r - n * tiles_h) * TH;
    wo0 = twi * TW;
  };
  auto patch_load = [&](int t) {   // ITEMS x 2 loads in flight per thread; pixels outside the image read zeros (= the padding)
    int n, ho0, wo0;
    tile_origin(t, n, ho0, wo0);
    const int hi0 = ho0 * S - a.pad_h, wi0 = wo0 * S - a.pad_w;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int pix = it * 512 + tid;
      const int py = pix / PW, px = pix - py * PW;
      const int hy = hi0 + py, wx = wi0 + px;
      const bool ok = pix < NPIX && (unsigned)hy < (unsigned)a.H && (unsigned)wx < (unsigned)a.W;
      const int off = ok ? (((n * a.H + hy) * a.W + wx) * a.in_cstride) * 4 : -1;
      lo[it] = buf_load16(rx, off, 0);
      hi[it] = buf_load16(rx, ok ? off + 16 : -1, 0);
    }
  };
  auto patch_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int pix = it * 512 + tid;
      if (pix < NPIX) {
        const bf16x4 l = to_bf16x4(lo[it]), h = to_bf16x4(hi[it]);
        bf16x8 v = {l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
        patch[buf * NPIX + pix] = v;
      }
    }
  };
  const int frow = lane & 31, khalf = lane >> 5;
  const int p = wave * 32 + frow;               // output pixel = this lane's B column inside the block
  const int ty = p / TW, tx = p - ty * TW;
  const int b_off = (ty * S) * PW + tx * S;
  const __bf16* wbase = sw + frow * WROW + 8 * khalf;

  patch_load(t_begin);
  patch_store(0);
  __syncthreads();
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const bool more = t + 1 < t_end;
    if (more) patch_load(t + 1);
    const bf16x8* pb = patch + buf * NPIX + b_off;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int t0 = 2 * i, t1 = (2 * i + 1 < NT) ? 2 * i + 1 : NT - 1;
      const int o0 = (t0 / KW) * PW + t0 % KW, o1 = (t1 / KW) * PW + t1 % KW;   // constants after unrolling
      const bf16x8 px = pb[khalf ? o1 : o0];
      const __bf16* wr = wbase + ((i >> 1) * 64) * WROW + (i & 1) * 16;
      const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wr);
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wr + 32 * WROW);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, px, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, px, acc1, 0, 0, 0);
    }
    // ---- epilogue: as conv1_halo_kernel.  D layout: col = lane & 31 -> pixel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> channel
    int n, ho0, wo0;
    tile_origin(t, n, ho0, wo0);
    const int oy = ho0 + ty, ox = wo0 + tx;
    if (oy < a.Ho && ox < a.Wo) {
      float* o = a.y + a.out_coff + ((long)(n * a.Ho + oy) * a.Wo + ox) * a.out_cstride + 4 * khalf;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv0 = *reinterpret_cast<const float4*>(&sbias[8 * g + 4 * khalf]);
        const float4 bv1 = *reinterpret_cast<const float4*>(&sbias[32 + 8 * g + 4 * khalf]);
        float4 v0 = make_float4(acc0[4 * g] + bv0.x, acc0[4 * g + 1] + bv0.y, acc0[4 * g + 2] + bv0.z, acc0[4 * g + 3] + bv0.w);
        float4 v1 = make_float4(acc1[4 * g] + bv1.x, acc1[4 * g + 1] + bv1.y, acc1[4 * g + 2] + bv1.z, acc1[4 * g + 3] + bv1.w);
        v0.x = v0.x > 0.f ? v0.x : v0.x * a.slope; v0.y = v0.y > 0.f ? v0.y : v0.y * a.slope;
        v0.z = v0.z > 0.f ? v0.z : v0.z * a.slope; v0.w = v0.w > 0.f ? v0.w : v0.w * a.slope;
        v1.x = v1.x > 0.f ? v1.x : v1.x * a.slope; v1.y = v1.y > 0.f ? v1.y : v1.y * a.slope;
        v1.z = v1.z > 0.f ? v1.z : v1.z * a.slope; v1.w = v1.w > 0.f ? v1.w : v1.w * a.slope;
        // plain stores: a lane's eight 16-byte pieces of a pixel's 256-byte row meet in L2 (non-temporal ones went out as 32-byte
        // fragments: 0.45 ms for the layer)
        *reinterpret_cast<float4*>(o + 8 * g) = v0;
        *reinterpret_cast<float4*>(o + 32 + 8 * g) = v1;
      }
    }
    if (more) patch_store(buf ^ 1);   // the other buffer: its last readers passed the barrier that ended the previous block
    __syncthreads();
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------- first layer, three terms
// flow_conv1 with f32 operands on the bf16 matrix pipe: every weight and every input value is the exact sum of three bf16 terms and a
// product keeps the six largest term products, accumulated in f32 -- the arithmetic of wino_gemm_split.hip (error <= 3 * 2^-27 per
// product, below f32's own rounding of the sum).  On the f32 pipe this layer is bound by its 392 MFMAs of 64 cycles per 32 x 64 block
// (conv1_halo_kernel: 0.60 ms at 16 pairs, 102 TFLOP/s); six MFMAs of 32 cycles per tap pair are 2.2x fewer pipe cycles.
// The three-term weights of all 64 output channels (150 KB) do not fit LDS beside a patch, and streamed per wave from L2 they are the
// bound (conv1_halo_bf16_kernel's note).  So a PERSISTENT 8-wave workgroup owns HALF the output channels: its 76.8 KB of weights stay in
// LDS, [tap pair 25][term 3][k half 2][channel 32][8 bf16] (a wave's A fragment = 1 KB contiguous, conflict-free), and it walks a range
// of 16 x 16 pixel blocks whose 37 x 37 x 8 patch is split on the way into LDS: three images of 16 B per pixel, the even and the odd
// input columns in separate planes with a 24-slot row pitch -- with stride 2 the 16 lanes that ds_read_b128 serves together read one
// tap of 16 consecutive output pixels = 16 consecutive slots of one column parity (two rows apart: 48 slots = a multiple of the 16
// slots the 64 banks hold) -- conflict-free.  The workgroups 2 j and 2 j + 1 (one XCD) walk the same blocks for the two channel halves:
// the second read of a patch comes out of L2.  One wave = 32 pixels x 32 channels, 150 MFMAs per block on two accumulators.
#ifndef DIM_C1_EXP   // timing experiments on conv1_halo_split_kernel (tools/split_exp.sh FILE=conv.hip): 1 no MFMAs, 2 no fragment reads,
#define DIM_C1_EXP 0 // 4 no patch split / store, 8 no output stores -- WRONG results with any bit set
#endif
constexpr int kC1Pairs = 25;
constexpr size_t kC1SplitBytes = 2 * (size_t)kC1Pairs * 3 * 2 * 32 * 16;   // both halves: 153 600 B behind the packed f32 weights

struct C1Split {
  uint4 h, m, l;
};
__device__ __forceinline__ C1Split c1_split8(const float4 lo, const float4 hi) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x8 x = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  const bf16x8 bh = __builtin_convertvector(x, bf16x8);
  const f32x8 r1 = x - __builtin_convertvector(bh, f32x8);
  const bf16x8 bm = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(bm, f32x8);
  const bf16x8 bl = __builtin_convertvector(r2, bf16x8);
  C1Split s;
  s.h = __builtin_bit_cast(uint4, bh);
  s.m = __builtin_bit_cast(uint4, bm);
  s.l = __builtin_bit_cast(uint4, bl);
  return s;
}

// packed f32 weights [13 chunks][64][4 taps x 8 channels] -> the three-term image (layout above); one thread per (channel, tap slot)
__global__ __launch_bounds__(256) void conv1_split_weights_kernel(const float* __restrict__ wp, unsigned char* __restrict__ w3) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 64 * 2 * kC1Pairs) return;
  const int co = t & 63, slot = t >> 6;   // slot = 2 pair + k half = the tap (49 = padding)
  float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
  if (slot < 49) {
    const float* src = wp + ((slot >> 2) * 64 + co) * 32 + (slot & 3) * 8;
    lo = *reinterpret_cast<const float4*>(src);
    hi = *reinterpret_cast<const float4*>(src + 4);
  }
  const C1Split sp = c1_split8(lo, hi);
  const int pair = slot >> 1, kh = slot & 1, half = co >> 5;
  unsigned char* dst = w3 + ((((size_t)(half * kC1Pairs + pair) * 3) * 2 + kh) * 32 + (co & 31)) * 16;
  *reinterpret_cast<uint4*>(dst) = sp.h;
  *reinterpret_cast<uint4*>(dst + 1024) = sp.m;
  *reinterpret_cast<uint4*>(dst + 2048) = sp.l;
}

template <int KH, int KW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1_halo_split_kernel(ConvArgs a, const unsigned char* __restrict__ w3, int tiles, int per_pair) {
  constexpr int TH = 16, TW = 16, S = 2;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;   // 37 x 37 input pixels
  constexpr int NPIX = PH * PW, NT = KH * KW, NPAIR = (NT + 1) / 2;
  constexpr int PITCH = 24;                  // 16-byte slots per patch row of one column parity (19 used)
  constexpr int PLANE = PH * PITCH;          // slots of one parity plane
  constexpr int TERM = 2 * PLANE;            // slots of one term's image
  constexpr int ITEMS = (NPIX + 511) / 512;
  constexpr int WBYTES = NPAIR * 3 * 2 * 32 * 16;
  static_assert(NPAIR == kC1Pairs, "7 x 7 taps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c1[];
  unsigned char* sw = smem_c1;                       // this half's weights
  uint4* sp = reinterpret_cast<uint4*>(smem_c1 + WBYTES);   // [3 terms][2 parities][PH][PITCH]
  float* sbias = reinterpret_cast<float*>(smem_c1 + WBYTES + 3 * TERM * 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = wg_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int half = wg & 1, pr = wg >> 1;
  const int t_begin = pr * per_pair, t_end = min(tiles, t_begin + per_pair);
  if (t_begin >= t_end) return;
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
  if (tid < 32) sbias[tid] = a.has_bias ? a.bias[half * 32 + tid] : 0.f;

  for (int it = tid; it < WBYTES / 16; it += 512)
    *reinterpret_cast<uint4*>(sw + it * 16) = *reinterpret_cast<const uint4*>(w3 + (size_t)half * WBYTES + it * 16);

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  float4 lo[ITEMS], hi[ITEMS];
  auto tile_origin = [&](int t, int& n, int& ho0, int& wo0) {
    const int twi = t % tiles_w;
    const int r = t / tiles_w;
    n = r / tiles_h;
    ho0 = (r - n * tiles_h) * TH;
    wo0 = twi * TW;
  };
  auto patch_load = [&](int t) {   // pixels outside the image read zeros (= the padding)
    int n, ho0, wo0;
    tile_origin(t, n, ho0, wo0);
    const int hi0 = ho0 * S - a.pad_h, wi0 = wo0 * S - a.pad_w;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int pix = it * 512 + tid;
      const int py = pix / PW, px = pix - py * PW;
      const int hy = hi0 + py, wx = wi0 + px;
      const bool ok = pix < NPIX && (unsigned)hy < (unsigned)a.H && (unsigned)wx < (unsigned)a.W;
      const int off = ok ? (((n * a.H + hy) * a.W + wx) * a.in_cstride) * 4 : -1;
      lo[it] = buf_load16(rx, off, 0);
      hi[it] = buf_load16(rx, ok ? off + 16 : -1, 0);
    }
  };
  // the split of the next block's pixels happens in registers while this block multiplies (the VALU work hides under the MFMAs of the
  // SIMD's other wave); after the barrier that ends the block only the LDS stores are left
  C1Split s3[ITEMS];
  auto patch_split = [&]() {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) s3[it] = c1_split8(lo[it], hi[it]);
  };
  auto patch_store = [&]() {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int pix = it * 512 + tid;
      if (pix < NPIX) {
        const int py = pix / PW, px = pix - py * PW;
        const int slot = (px & 1) * PLANE + py * PITCH + (px >> 1);
        sp[slot] = s3[it].h;
        sp[TERM + slot] = s3[it].m;
        sp[2 * TERM + slot] = s3[it].l;
      }
    }
  };
  const int frow = lane & 31, khalf = lane >> 5;
  const int p = wave * 32 + frow;               // output pixel = this lane's B column inside the block
  const int ty = p / TW, tx = p - ty * TW;
  const uint4* pb = sp + (ty * S) * PITCH + tx;
  const unsigned char* wa = sw + (khalf * 32 + frow) * 16;

  // stores through a descriptor: a pixel outside the output gets offset 0xFFFFFFFF, which the range check drops -- no branch around
  // the stores (with one, hipcc waits vmcnt(0) for the next block's patch loads and thereby for these stores: one in-order counter)
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  // (Measured and not kept, same-box A/B: the split dealt in fifteen steps behind the MFMAs of tap pairs 8 .. 22 and the outputs of a
  // block written during the next block's first four pairs -- 379 / 383 us against 368 / 397: inside the noise.)
  // (Measured and not kept: different orders for the two waves of a SIMD -- waves 4 .. 7 splitting late in the tap loop and writing their
  // outputs during the next block's first taps -- 407 us against 363: the wave-uniform branches inside the unrolled tap loop cost more
  // than the overlap returned.  Ablations of this form at 16 pairs: MFMAs 200 us of the 363, fragment reads 75, patch split + store 52,
  // output 46, roughly additive: the eight waves of the one workgroup a CU holds move through a block in step.)
  auto epilogue = [&](const f32x16& sum, int tt) {
    // D layout: col = lane & 31 -> pixel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> channel of this half
    int n, ho0, wo0;
    tile_origin(tt, n, ho0, wo0);
    const int oy = ho0 + ty, ox = wo0 + tx;
    const int o_off = (oy < a.Ho && ox < a.Wo) ? (a.out_coff + ((n * a.Ho + oy) * a.Wo + ox) * a.out_cstride + half * 32 + 4 * khalf) * 4 : -1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(&sbias[8 * g + 4 * khalf]);
      float4 v = make_float4(sum[4 * g] + bv.x, sum[4 * g + 1] + bv.y, sum[4 * g + 2] + bv.z, sum[4 * g + 3] + bv.w);
      v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
      v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
      u32x4 u;
      u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
      if constexpr (DIM_C1_EXP & 8) asm volatile("" ::"v"(u)); else
      __builtin_amdgcn_raw_buffer_store_b128(u, ry, o_off == -1 ? -1 : o_off + 32 * g, 0, 0);
    }
  };
  patch_load(t_begin);
  patch_split();
  patch_store();
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const bool more = t + 1 < t_end;
    patch_load(more ? t + 1 : t);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    // fragments of tap pair i + 1 are requested before the six MFMAs of pair i are issued (two register sets; fenced, or hipcc sinks
    // every read to its first use; a third set changed nothing)
    bf16x8 fx[2][3], fw[2][3];
    auto frags = [&](auto I_, auto SET_) {
      constexpr int i = decltype(I_)::value, set = decltype(SET_)::value;
      constexpr int t0 = 2 * i, t1 = (2 * i + 1 < NT) ? 2 * i + 1 : NT - 1;   // tap 49: zero weights, its pixel operand re-reads tap 48
      constexpr int o0 = ((t0 % KW) & 1) * PLANE + (t0 / KW) * PITCH + ((t0 % KW) >> 1);
      constexpr int o1 = ((t1 % KW) & 1) * PLANE + (t1 / KW) * PITCH + ((t1 % KW) >> 1);
      const uint4* ppx = pb + (khalf ? o1 : o0);
      const unsigned char* wr = wa + i * 3 * 1024;
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) {
        fx[set][tm] = __builtin_bit_cast(bf16x8, ppx[tm * TERM]);
        fw[set][tm] = *reinterpret_cast<const bf16x8*>(wr + tm * 1024);
      }
    };
    frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    static_for<NPAIR>([&](auto I_) {
      constexpr int i = decltype(I_)::value, set = i & 1;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i + 1 < NPAIR && !(DIM_C1_EXP & 2)) frags(std::integral_constant<int, i + 1>{}, std::integral_constant<int, 1 - set>{});
      __builtin_amdgcn_sched_barrier(0);
#if DIM_C1_EXP & 1   // timing experiment: no MFMAs, the operands stay loaded
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) asm volatile("" ::"v"(fw[set][tm]), "v"(fx[set][tm]));
#else
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][2], fx[set][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][0], fx[set][2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][1], fx[set][1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][1], fx[set][0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][0], fx[set][1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][0], fx[set][0], acc1, 0, 0, 0);
#endif
      if constexpr (i == NPAIR / 2 && !(DIM_C1_EXP & 4)) patch_split();   // the loads were issued a dozen tap pairs ago
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
    epilogue(acc0, t);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave has read this block's patch (LDS-only barrier)
    if constexpr (!(DIM_C1_EXP & 4)) patch_store();             // (after the last block: the same block again, unused)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

template <int BM, int BN, int WM, int WN, bool CIN8>
__global__ __launch_bounds__(WM * WN * 64) void conv_bf16_kernel(ConvArgs a) {
  constexpr int BK = 32;
  constexpr int NT = WM * WN * 64;
  constexpr int RP = NT / 8;        // rows staged per pass (8 threads x float4 = one 32-value row)
  constexpr int LDH = BK + 8;       // row stride in bf16 elements (80 B): 16 rows x 16 B land on 16 distinct 4-bank slots (ds_read_b128)
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_PER_T = BM / RP;
  static_assert((WM * WN == 4 || WM * WN == 8) && A_PER_T >= 1 && A_PER_T <= 4, "staging plan");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* sA = reinterpret_cast<__bf16*>(smem);  // [2][BM][LDH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int id = blockIdx.x + a.tile_off;
  if (a.xcd_chunk > 0) id = (id & 7) * a.xcd_chunk + (id >> 3);
  const int ntiles_n = a.Cout / BN;
  const int mtile = id / ntiles_n;
  const int m0 = mtile * BM;
  const int n0 = (id - mtile * ntiles_n) * BN;
  const int split = blockIdx.z;
  const int kc_begin = split * a.chunks_per_split;
  const int kc_end = min(a.nchunks, kc_begin + a.chunks_per_split);

  const int q = tid & 7;
  const int srow = tid >> 3;
  int a_hi0[A_PER_T], a_wi0[A_PER_T], a_pix[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    int m = m0 + srow + RP * i;
    bool ok = m < a.M;
    int mm = ok ? m : 0;
    int wo = mm % a.Wo;
    int t = mm / a.Wo;
    int ho = t % a.Ho;
    int n = t / a.Ho;
    a_hi0[i] = ok ? ho * a.stride - a.pad_h : -(1 << 28);
    a_wi0[i] = wo * a.stride - a.pad_w;
    a_pix[i] = ((n * a.H + (ok ? a_hi0[i] : 0)) * a.W * a.in_cstride + (wo * a.stride - a.pad_w) * a.in_cstride + (CIN8 ? (q & 1) * 4 : q * 4)) * 4;
  }
  const float* xb = a.x + (long)blockIdx.y * a.bx;
  const char* wb = reinterpret_cast<const char*>(a.w) + (long)blockIdx.y * a.bw * 2;  // bw counts elements; bf16 = 2 bytes
  float* yb = a.y + (long)blockIdx.y * a.by;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, a.w_bytes, 0x00020000);
  const int wchunk_bytes = a.Cout * BK * 2;

  int kh, kw, c0;
  if (CIN8) {
    kw = 4 * kc_begin + (q >> 1);
    kh = 0;
    c0 = 0;
  } else {
    int taps = a.KH * a.KW;
    int cc = kc_begin / taps;
    int tap = kc_begin - cc * taps;
    c0 = cc << 5;
    kh = tap / a.KW;
    kw = tap - kh * a.KW;
  }
  float4 ra[A_PER_T];
  auto load_chunk = [&](bool pf) {
    const int tkh = CIN8 ? (int)fastdiv((unsigned)kw, a.div_kw) : kh;
    const int tkw = CIN8 ? kw - tkh * a.KW : kw;
    const bool pf_ok = pf && (!CIN8 || tkh < a.KH);
    const int tap_off = ((tkh * a.W + tkw) * a.in_cstride + c0) * 4;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      bool ok = pf_ok && (unsigned)(a_hi0[i] + tkh) < (unsigned)a.H && (unsigned)(a_wi0[i] + tkw) < (unsigned)a.W;
      ra[i] = buf_load16(rx, ok ? a_pix[i] + tap_off : -1, 0);
    }
    if (CIN8) {
      kw += 4;
    } else if (++kw == a.KW) {
      kw = 0;
      if (++kh == a.KH) { kh = 0; c0 += 32; }
    }
  };
  auto store_chunk = [&](int buf) {
    __bf16* dA = sA + buf * BM * LDH;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) *reinterpret_cast<bf16x4*>(dA + (srow + RP * i) * LDH + q * 4) = to_bf16x4(ra[i]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int khalf = lane >> 5;
  const int a_off = (wm * (BM / WM) + frow) * LDH + 8 * khalf;
  const int bf_voff = ((n0 + wn * (BN / WN) + frow) * BK + 8 * khalf) * 2;
  bf16x8 fb[2][2][TN];  // [set][k-step][tile]
  auto load_bfrag = [&](const int set, int kc) {  // always called with a literal / constexpr set: inlined, indices fold
    const int bsoff = kc * wchunk_bytes;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, bf_voff + (32 * j * BK + 16 * s) * 2, bsoff, 0);
        fb[set][s][j] = *reinterpret_cast<bf16x8*>(&v);
      }
  };

  if (kc_begin < kc_end) {
    load_chunk(true);
    store_chunk(0);
    load_bfrag(0, kc_begin);
  }
  __syncthreads();
  load_chunk(kc_begin + 1 < kc_end);
  int buf = 0;
  // two chunks per trip so that the B-fragment set (a register array) is indexed by a compile-time constant
  auto chunk_body = [&](auto SET, int kc) {
    constexpr int set = decltype(SET)::value;
    const __bf16* cA = sA + buf * BM * LDH + a_off;
    load_bfrag(set ^ 1, min(kc + 1, a.nchunks - 1));
    bf16x8 fa[2][TM];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(cA + 32 * i * LDH + 16 * s);
    // the staged registers of chunk kc+1 go to the other buffer (its readers finished before the previous barrier), then the
    // loads of chunk kc+2 are issued: they fly over the MFMAs below and the next chunk's
    store_chunk(buf ^ 1);
    load_chunk(kc + 2 < kc_end);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[set][s][j], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  };
  for (int kc = kc_begin; kc < kc_end; kc += 2) {
    chunk_body(std::integral_constant<int, 0>{}, kc);
    if (kc + 1 < kc_end) chunk_body(std::integral_constant<int, 1>{}, kc + 1);
  }

  // ---- epilogue (conv_store_tiles above: branch-free buffer stores)
  conv_store_tiles<TM, TN>(a, acc, yb, m0 + wm * (BM / WM) + 4 * khalf, n0 + wn * (BN / WN) + frow, split);
}

// ---------------------------------------------------------------------------------------------------------------- bf16, LDS halo
// The bf16 form of the large-map layers from an LDS-resident input patch (tile 7).  conv_bf16_kernel re-gathers its A tile from L2
// for EVERY tap (10.7 GB of L2 -> LDS traffic per forward at B = 16) and has 4 MFMAs of work per barrier, which leaves every layer
// at 0.11-0.24 of its bound once the matrix pipe is 16x faster.  Here a workgroup (4 waves, 64 pixels x 64 channels each) owns an
// 8 x 16 block of output pixels x 128 output channels; per 32-channel slice it stages the ((8-1) S + K) x ((16-1) S + K) input
// patch once (f32 -> bf16 on the way in, zeros outside the image = the padding) and then walks the K x K taps: the A fragments of a
// tap are ds_read_b128 at an immediate offset into the patch, the B tile of a tap (128 channels x 32 k, 8 KB of the packed bf16
// weights: one contiguous block) is staged through LDS two taps at a time, double buffered, so a barrier pair frames 16 MFMAs per
// wave and the activations leave L2 once per slice instead of once per tap.  Dense output only (forward layers and the stride-1
// input gradients); everything else stays on conv_bf16_kernel.
template <int KH, int S>
__global__ __launch_bounds__(256) void conv_bf16_halo_kernel(ConvArgs a) {
  constexpr int KW = KH, TAPS = KH * KW;
  constexpr int TH = 8, TW = 16, BN = 128;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, NPIX = PH * PW;
  constexpr int PSE = 40;                        // bf16 elements per patch pixel / per weight row: 32 + 8 pad (80 B)
  constexpr int STEPS = (TAPS + 1) / 2;          // taps are processed two per barrier pair (the last step of a slice holds one)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* patch = reinterpret_cast<__bf16*>(smem);             // [NPIX][PSE]
  __bf16* sB = patch + ((NPIX * PSE + 7) / 8) * 8;             // [2 buffers][2 taps][BN][PSE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
  const int ntn = a.Cout / BN;
  int id = blockIdx.x;
  const int n0 = (id % ntn) * BN;                // output-channel tiles fastest: the workgroups sharing a patch are adjacent
  id /= ntn;
  const int twi = id % tiles_w;
  id /= tiles_w;
  const int thi = id % tiles_h;
  const int n = id / tiles_h;
  const int ho0 = thi * TH, wo0 = twi * TW;
  const int hi0 = ho0 * S - a.pad_h, wi0 = wo0 * S - a.pad_w;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

  // ---- weights: tap t of slice cc is packed chunk cc * TAPS + t; this workgroup's 128 rows of it are 8 KB contiguous.
  // thread -> 2 x 16 B of a tap (row = idx / 4, 16-byte segment = idx % 4)
  const int wrow0 = tid >> 2, wseg = tid & 3;
  const int w_voff0 = ((n0 + wrow0) * 32 + wseg * 8) * 2, w_voff1 = w_voff0 + 64 * 32 * 2;
  const int w_lds0 = wrow0 * PSE + wseg * 8, w_lds1 = w_lds0 + 64 * PSE;
  u32x4 wr[2][2];  // [tap of the step][half]
  const int chunk_bytes = a.Cout * 32 * 2;
  auto load_w = [&](int chunk_first, int ntaps, bool ok) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bool live = ok && t < ntaps;
      const int soff = live ? (chunk_first + t) * chunk_bytes : 0;
      wr[t][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, live ? w_voff0 : -1, soff, 0);
      wr[t][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, live ? w_voff1 : -1, soff, 0);
    }
  };
  auto store_w = [&](int buf) {
    __bf16* d = sB + buf * 2 * BN * PSE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *reinterpret_cast<u32x4*>(d + t * BN * PSE + w_lds0) = wr[t][0];
      *reinterpret_cast<u32x4*>(d + t * BN * PSE + w_lds1) = wr[t][1];
    }
  };

  // ---- fragments
  const int frow = lane & 31, khalf = lane >> 5;
  const int ty_l = frow >> 4, tx = frow & 15;
  const int a_el = (((4 * wm + ty_l) * S) * PW + tx * S) * PSE + 8 * khalf;   // + (2 i S PW) PSE for MFMA tile i, + tap, + 16 ks
  const int b_el = (64 * wn + frow) * PSE + 8 * khalf;                        // + 32 j PSE, + 16 ks
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nslices = a.Cin / 32;
  constexpr int PITEMS = (NPIX * 8 + 255) / 256;   // float4 per thread and patch slice
  int wbuf = 0;
  load_w(0, TAPS >= 2 ? 2 : 1, true);
  for (int cc = 0; cc < nslices; ++cc) {
    // ---- stage the patch of this channel slice (the previous slice's readers passed the barrier at the end of its last step)
#pragma unroll
    for (int it0 = 0; it0 < PITEMS; it0 += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int item = (it0 + u) * 256 + tid;
        const int pix = item >> 3, q = item & 7;
        const int py = pix / PW, px = pix - py * PW;
        const int hi = hi0 + py, wi = wi0 + px;
        const bool ok = it0 + u < PITEMS && item < NPIX * 8 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        v[u] = buf_load16(rx, ok ? (((n * a.H + hi) * a.W + wi) * a.in_cstride + cc * 32 + q * 4) * 4 : -1, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int item = (it0 + u) * 256 + tid;
        if (it0 + u < PITEMS && item < NPIX * 8) *reinterpret_cast<bf16x4*>(patch + (item >> 3) * PSE + (item & 7) * 4) = to_bf16x4(v[u]);
      }
    }
    store_w(wbuf);          // the first step's weights (loaded during the previous slice / before the loop)
    __syncthreads();
    for (int st = 0; st < STEPS; ++st) {
      const int ntaps = (2 * st + 2 <= TAPS) ? 2 : 1;
      // next step's weights in flight under this step's MFMAs (next slice's first step after the last one)
      {
        const int nst = st + 1 < STEPS ? st + 1 : 0;
        const int ncc = st + 1 < STEPS ? cc : cc + 1;
        load_w(ncc * TAPS + 2 * nst, (2 * nst + 2 <= TAPS) ? 2 : 1, ncc < nslices);
      }
      const __bf16* cB = sB + wbuf * 2 * BN * PSE + b_el;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t < ntaps) {
          const int tap = 2 * st + t;
          const int kh = tap / KW, kw = tap - kh * KW;
          const __bf16* pa = patch + a_el + (kh * PW + kw) * PSE;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa + (2 * i * S * PW) * PSE + 16 * ks);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cB + t * BN * PSE + 32 * j * PSE + 16 * ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
          }
        }
      }
      if (st + 1 < STEPS) store_w(wbuf ^ 1);   // (the next slice's first step is stored after its patch, above)
      __syncthreads();
      if (st + 1 < STEPS) wbuf ^= 1;
    }
    wbuf ^= 1;
  }

  // ---- epilogue.  D layout: col = lane & 31 -> output channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> pixel of the tile.
  // Branch-free buffer stores (see conv_store_tiles): a pixel outside the map gets offset 0xFFFFFFFF and is dropped.
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  int voff[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pp = (r & 3) + 8 * (r >> 2) + 4 * khalf;           // 0..31 inside MFMA tile i
      const int oy = ho0 + 4 * wm + 2 * i + (pp >> 4), ox = wo0 + (pp & 15);
      voff[i][r] = (oy < a.Ho && ox < a.Wo) ? (((n * a.Ho + oy) * a.Wo + ox) * a.out_cstride + a.out_coff + n0 + 64 * wn + frow) * 4 : -1;
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float bv = a.has_bias ? a.bias[n0 + 64 * wn + 32 * j + frow] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        old[r] = a.accumulate ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, voff[i][r], 128 * j, 0)) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        v = (v > 0.f ? v : v * a.slope) + old[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff[i][r], 128 * j, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- bf16, stride-1 patch
// Tile 9: every stride-1 bf16 convolution with a small rectangular tap set (KH, KW <= 3) on a large map -- the 3x3 forward layers,
// their input gradients, and the stride-1 phase convolutions a stride-2 input gradient or a 4x4 / stride-2 deconvolution splits into
// (2x2, 2x3, 3x2, 3x3 taps, scattered output).  PMC on the kernels above (bf16 training iteration): waves parked on s_waitcnt /
// barriers 60 % of their life, matrix pipe busy 12-26 % -- conv_bf16_kernel moves 12.7 TB/s out of L2 (the gather repeats per
// tap), conv_bf16_halo_kernel awaits its weight tiles and its patch with one step of flight time.  Here
//  * a workgroup (4 waves, 2 x 2) owns a 16 x 16 block of output pixels x 128 output channels, a wave 128 pixels x 64 channels
//    (8 accumulator tiles): 16 MFMAs per tap and k-slice against 8 LDS fragment reads and 4 weight-fragment loads;
//  * the (16+KH-1) x (16+KW-1) input patch of a 32-channel slice lives in LDS (f32 -> bf16 on the way in, zeros outside the image
//    = the padding), double buffered: the next slice's patch is fetched in batches at the first taps of the current slice, each
//    batch converted and stored one tap after the next one was issued -- two taps of flight time, ONE barrier per slice;
//  * the weights never touch LDS: a lane's B fragment is 16 contiguous bytes of the packed bf16 array (chunk = slice * taps + tap),
//    loaded NSETS-1 taps ahead into a rotating register set;
//  * patch rows are 1536 B apart (a multiple of 256 B) and pixels 80 B: a ds_read_b128 lane group ({0-3,12-15,20-27}: two pixel
//    rows of an MFMA tile) then covers all 64 banks exactly once.
template <int KH, int KW, int BN, int S>
__global__ __launch_bounds__(256) void conv_bf16_patch_kernel(ConvArgs a) {
  constexpr int NT = KH * KW;
  // S = 1: 16 x 16 output pixels per workgroup; S = 2 (the stride-2 forward layers): 8 x 16, and the patch keeps the even and the odd
  // input columns of a row in two halves (1536 B apart), so that the 16 pixels of a fragment row -- every second input column --
  // are 80 B apart again and the bank argument below holds for both strides
  constexpr int TH = S == 1 ? 16 : 8, TW = 16;
  constexpr int TMW = TH / 4;                         // 32-pixel MFMA tiles (2 rows x 16) per wave: the wave's TH / 2 rows
  constexpr int TN = BN / 64;                         // 32-channel tiles per wave: BN = 128 (2 x 64 per wave column) or 64 (2 x 32)
  static_assert((BN == 128 || BN == 64) && (S == 1 || S == 2), "channel tile / stride");
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;   // input rows / columns under the block
  constexpr int PXB = 80, HALF = 1536, PITCH = S * HALF, PBUF = PH * PITCH;
  constexpr int NITEM = PH * PW * 8;                 // float4 pieces of one patch slice
  constexpr int PITEMS = (NITEM + 255) / 256;        // per thread
  constexpr int NSETS = NT % 3 == 0 ? 3 : (NT % 4 == 0 ? 4 : (NT % 5 == 0 ? 5 : 2));
  constexpr int PF = NSETS - 1;                      // weight prefetch distance in taps
  constexpr int IPT = (PITEMS + (NT > 1 ? NT - 2 : 0)) / (NT > 1 ? NT - 1 : 1);   // patch pieces fetched per tap (taps 0 .. NT-2)
  static_assert(NT >= 2 && NT % NSETS == 0 && ((PW + S - 1) / S) * PXB <= HALF && IPT * (NT - 1) >= PITEMS, "tap plan");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* patch = reinterpret_cast<char*>(smem);       // [2][PH][PITCH] bytes + 256 B dump slot

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
  const int ntn = a.Cout / BN;
  int id = blockIdx.x;
  const int n0 = (id % ntn) * BN;   // channel tiles fastest: the workgroups sharing a patch are neighbours in launch order
  id /= ntn;
  const int twi = id % tiles_w;
  id /= tiles_w;
  const int thi = id % tiles_h;
  const int n = id / tiles_h;
  const int ho0 = thi * TH, wo0 = twi * TW;
  const int hi0 = ho0 * S - a.pad_h, wi0 = wo0 * S - a.pad_w;
  const float* xb = a.x + (long)blockIdx.y * a.bx;
  const char* wb = reinterpret_cast<const char*>(a.w) + (long)blockIdx.y * a.bw * 2;
  float* yb = a.y + (long)blockIdx.y * a.by;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, a.w_bytes, 0x00020000);
  const int nslices = a.Cin / 32;
  const int chunk_bytes = a.Cout * 64;

  // ---- patch pieces of this thread: global byte offset of slice 0 (-1: outside the image / past the patch), LDS byte offset
  int p_goff[PITEMS], p_loff[PITEMS];
#pragma unroll
  for (int u = 0; u < PITEMS; ++u) {
    const int item = u * 256 + tid;
    const int pix = item >> 3, q = item & 7;
    const int py = pix / PW, px = pix - py * PW;
    const int hi = hi0 + py, wi = wi0 + px;
    const bool ok = item < NITEM && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
    p_goff[u] = ok ? (((n * a.H + hi) * a.W + wi) * a.in_cstride + q * 4) * 4 : -1;
    p_loff[u] = item < NITEM ? py * PITCH + (S == 2 ? (px & 1) * HALF + (px >> 1) * PXB : px * PXB) + q * 8 : -1;
  }

  // ---- fragments (wave (wm, wn): output rows (TH / 2) wm .. + TH / 2, channels (BN / 2) wn .. + BN / 2)
  const int frow = lane & 31, khalf = lane >> 5;
  const int a_off = ((TH / 2) * wm + (frow >> 4)) * S * PITCH + (frow & 15) * PXB + 16 * khalf;   // + 2 i S PITCH, + tap offset, + 32 ks
  const int b_voff = ((n0 + (BN / 2) * wn + frow) * 32 + 8 * khalf) * 2;                      // + 32 j rows, + 16 ks elements
  bf16x8 fb[NSETS][2][TN];  // [set][k-step][channel tile]
  auto load_b = [&](auto SET, int chunk) {
    constexpr int set = decltype(SET)::value;
    const int soff = chunk * chunk_bytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff + (32 * j * 32 + 16 * ks) * 2, soff, 0);
        fb[set][ks][j] = *reinterpret_cast<bf16x8*>(&v);
      }
  };
  f32x16 acc[TMW][TN];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: patch of slice 0 (exposed once), weight sets of taps 0 .. PF-1
  {
    float4 v[PITEMS];
#pragma unroll
    for (int u = 0; u < PITEMS; ++u) v[u] = buf_load16(rx, p_goff[u], 0);
#pragma unroll
    for (int u = 0; u < PITEMS; ++u)
      if (p_loff[u] >= 0) *reinterpret_cast<bf16x4*>(patch + p_loff[u]) = to_bf16x4(v[u]);
  }
  const int last_chunk = a.nchunks - 1;
  if constexpr (PF >= 1) load_b(std::integral_constant<int, 0>{}, 0);
  if constexpr (PF >= 2) load_b(std::integral_constant<int, 1>{}, min(1, last_chunk));
  if constexpr (PF >= 3) load_b(std::integral_constant<int, 2>{}, min(2, last_chunk));
  if constexpr (PF >= 4) load_b(std::integral_constant<int, 3>{}, min(3, last_chunk));
  __syncthreads();

  int buf = 0;
  float4 st[2][IPT];   // two batches of patch pieces in flight
  bf16x8 fa[2][2][TMW];  // [tap parity][k-step][pixel tile]
  auto load_a = [&](auto SET, const char* pc, auto TAP) {
    constexpr int set = decltype(SET)::value, tap = decltype(TAP)::value;
    constexpr int kh = tap / KW, kw = tap - kh * KW;
    constexpr int tap_off = kh * PITCH + (S == 2 ? (kw & 1) * HALF + (kw >> 1) * PXB : kw * PXB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TMW; ++i)
        fa[set][ks][i] = *reinterpret_cast<const bf16x8*>(pc + tap_off + 2 * i * S * PITCH + 32 * ks);
  };
  for (int cc = 0; cc < nslices; ++cc) {
    const char* pcur = patch + buf * PBUF + a_off;
    load_a(std::integral_constant<int, 0>{}, pcur, std::integral_constant<int, 0>{});   // tap 0: after the barrier that published this patch
    char* pnext = patch + (buf ^ 1) * PBUF;
    const int next_soff = (cc + 1) * 128;           // byte offset of the next slice's channels
    const bool have_next = cc + 1 < nslices;
    const int g0 = cc * NT;
    static_for<NT>([&](auto T) {
      constexpr int t = decltype(T)::value;
      // One scheduling region per tap.  Program order: patch pieces of batch t (next slice) and the weights of tap t + PF, the A
      // fragments of tap t + 1, the 16 MFMAs of tap t, rounding + LDS stores of batch t - 1.  With one wave per SIMD nothing else
      // fills the matrix pipe while the wave issues loads / LDS traffic / VALU, so the sched_group_barrier sequence below deals
      // them out one small group behind each MFMA (an MFMA holds the issue port 8 of its 32 cycles); without it hipcc either sinks
      // the loads to their first use (prefetch distance gone, one load even inside a branch followed by vmcnt(0)) or, fenced into
      // blocks, leaves the pipe idle during every non-MFMA block (measured: 43 % MFMA-busy inside a wave's life).
      constexpr int NLD = (t < NT - 1 ? (IPT < PITEMS - t * IPT ? IPT : (PITEMS - t * IPT > 0 ? PITEMS - t * IPT : 0)) : 0) + 2 * TN;
      constexpr int NST = t >= 1 ? (IPT < PITEMS - (t - 1) * IPT ? IPT : (PITEMS - (t - 1) * IPT > 0 ? PITEMS - (t - 1) * IPT : 0)) : 0;
      if (t < NT - 1) {
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
          const int u = t * IPT + e;
          if (u < PITEMS) st[t & 1][e] = buf_load16(rx, (have_next && p_goff[u] >= 0) ? p_goff[u] + next_soff : -1, 0);
        }
      }
      load_b(std::integral_constant<int, (t + PF) % NSETS>{}, min(g0 + t + PF, last_chunk));
      if constexpr (t + 1 < NT) load_a(std::integral_constant<int, (t + 1) & 1>{}, pcur, std::integral_constant<int, t + 1>{});
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][ks][i], fb[t % NSETS][ks][j], acc[i][j], 0, 0, 0);
      if (t >= 1) {
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
          const int u = (t - 1) * IPT + e;
          if (u < PITEMS) {
            char* dst = p_loff[u] >= 0 ? pnext + p_loff[u] : patch + 2 * PBUF + (tid & 31) * 8;
            *reinterpret_cast<bf16x4*>(dst) = to_bf16x4(st[(t - 1) & 1][e]);
          }
        }
      }
      constexpr int NM = 2 * TMW * TN;   // MFMAs of the tap
      static_for<NM>([&](auto Mi) {
        constexpr int m = decltype(Mi)::value;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
        if constexpr (m < NLD) {
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                     // address arithmetic of ...
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     // ... one global load
        }
        if constexpr (m < 2 * TMW && t + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one A-fragment read
        if constexpr (m >= NM - NST) {
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                     // round one piece ...
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                     // ... and store it
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    // LDS-only barrier: the weight loads of the next taps stay in flight across it (__syncthreads would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  }

  // ---- epilogue.  D layout: col = lane & 31 -> output channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> pixel of the MFMA tile.
  // Branch-free: a pixel outside the output is a buffer store at offset 0xFFFFFFFF, which the range check drops.  (Stores inside
  // per-element `if` blocks cost 26 us per workgroup here: hipcc opens every block with s_waitcnt vmcnt(0) -- the bias load is
  // still "pending" at the block boundary -- and on gfx950 vmcnt also counts the stores, so the 128 stores went out one round
  // trip at a time.)
  const int by = blockIdx.y;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, a.y_bytes, 0x00020000);
  const int oy_base = a.ooy + (by >> 1) * a.boy, ox_base = a.oox + (by & 1) * a.box;
  // this lane's 64 pixel slots: row 8 wm + 2 i + (r >> 3), column 4 khalf + (r & 3) + 8 ((r >> 2) & 1) of the 16 x 16 block; the byte
  // offset is affine in both (channel co of tile j = + 128 j bytes), and a block that lies inside the output needs no per-pixel test
  const int hob = ho0 + (TH / 2) * wm, wob = wo0 + 4 * khalf;
  const int row_b = a.osy * a.OW * a.out_cstride * 4, col_b = a.osx * a.out_cstride * 4;
  const int base_b = (((n * a.OH + hob * a.osy + oy_base) * a.OW + wob * a.osx + ox_base) * a.out_cstride + a.out_coff + n0 + (BN / 2) * wn + frow) * 4;
  const bool inside = ho0 + TH <= a.Ho && wo0 + TW <= a.Wo && ho0 * a.osy + oy_base >= 0 && (ho0 + TH - 1) * a.osy + oy_base < a.OH &&
                      wo0 * a.osx + ox_base >= 0 && (wo0 + TW - 1) * a.osx + ox_base < a.OW;   // workgroup-uniform
  int voff[TMW][16];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = 2 * i + (r >> 3), dc = (r & 3) + 8 * ((r >> 2) & 1);
      voff[i][r] = base_b + dr * row_b + dc * col_b;
    }
  if (!inside) {
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ho = hob + 2 * i + (r >> 3), wo = wob + (r & 3) + 8 * ((r >> 2) & 1);
        const int oy = ho * a.osy + oy_base, ox = wo * a.osx + ox_base;
        const bool ok = ho < a.Ho && wo < a.Wo && (unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW;
        voff[i][r] = ok ? voff[i][r] : -1;
      }
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = a.has_bias ? a.bias[n0 + (BN / 2) * wn + 32 * j + frow] : 0.f;
  if (a.mask) {   // wave-uniform: the LeakyReLU' of the layer below and its bias gradient, folded in (see ConvArgs.mask)
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask) + (long)blockIdx.y * a.by, 0, a.y_bytes, 0x00020000);
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      cs[j] = 0.f;
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
        float mk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, voff[i][r], 128 * j, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[j];
          v = v > 0.f ? v : v * a.slope;
          v *= mk[r] > 0.f ? 1.f : a.mask_slope;
          cs[j] += voff[i][r] != -1 ? v : 0.f;   // a pixel slot outside the output is not stored and must not count
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff[i][r], 128 * j, 0);
        }
      }
    }
    // the two lane halves hold different pixels of the same channel; then one plain store per (block, wave row, channel): no atomics,
    // the reduce over blocks sums in a fixed order
    float* crow = a.colsum + ((long)a.colsum_row0 + 2L * (blockIdx.x / ntn) + wm) * a.Cout + n0 + (BN / 2) * wn + frow;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float tot = cs[j] + __shfl_xor(cs[j], 32);
      if (khalf == 0) crow[32 * j] = tot;
    }
  } else if (a.accumulate) {   // wave-uniform: out += result (gradients that meet in one buffer)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, voff[i][r], 128 * j, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[j];
          v = (v > 0.f ? v : v * a.slope) + old[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff[i][r], 128 * j, 0);
        }
      }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[j];
          v = v > 0.f ? v : v * a.slope;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff[i][r], 128 * j, 0);
        }
  }
}

// element-wise f32 -> bf16 (round to nearest even): the packed weight arrays of the bf16 kernels, the flat gradient bucket
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    *reinterpret_cast<bf16x4*>(dst + i) = to_bf16x4(*reinterpret_cast<const float4*>(src + i));
  } else {
    for (long k = i; k < n; ++k) dst[k] = (__bf16)src[k];
  }
}
__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(src + i);
    *reinterpret_cast<float4*>(dst + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  } else {
    for (long k = i; k < n; ++k) dst[k] = (float)src[k];
  }
}

// sum split-K slabs + bias + LeakyReLU.  One float4 per thread.
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ y,
                                     long MC, int Cout, int splits, float slope, int has_bias) {
  long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long i = i4 * 4;
  if (i >= MC) return;
  float4 s = *reinterpret_cast<const float4*>(slabs + i);
  for (int k = 1; k < splits; ++k) {
    float4 p = *reinterpret_cast<const float4*>(slabs + (long)k * MC + i);
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  if (has_bias) {
    int c = (int)(i % Cout);
    s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3];
  }
  s.x = s.x > 0.f ? s.x : s.x * slope;
  s.y = s.y > 0.f ? s.y : s.y * slope;
  s.z = s.z > 0.f ? s.z : s.z * slope;
  s.w = s.w > 0.f ? s.w : s.w * slope;
  *reinterpret_cast<float4*>(y + i) = s;
}

// the same sum for many slabs of a small array (the first layers' weight gradients: 512 slabs of 26 624 floats, where one thread per
// float4 walking all slabs left 26 workgroups with 512 dependent-latency loads each: 181 us): workgroup = 8 float4 columns x 32 slab
// lanes, lane p sums slabs p, p + 32, ..., then a fixed binary tree over the 32 partial sums (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_par_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                                                float* __restrict__ y, long MC, int Cout, int splits, float slope,
                                                                int has_bias) {
  __shared__ float4 red[32][8];
  const int qi = threadIdx.x & 7, part = threadIdx.x >> 3;
  const long i = ((long)blockIdx.x * 8 + qi) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < MC)
    for (int k = part; k < splits; k += 32) {
      const float4 p = *reinterpret_cast<const float4*>(slabs + (long)k * MC + i);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
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
  if (part == 0 && i < MC) {
    s = red[0][qi];
    if (has_bias) {
      int c = (int)(i % Cout);
      s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3];
    }
    s.x = s.x > 0.f ? s.x : s.x * slope;
    s.y = s.y > 0.f ? s.y : s.y * slope;
    s.z = s.z > 0.f ? s.z : s.z * slope;
    s.w = s.w > 0.f ? s.w : s.w * slope;
    *reinterpret_cast<float4*>(y + i) = s;
  }
}

// OIHW (MXNet / reference layout) -> packed [chunk][Cout][32]  (a workgroup's B chunk is one contiguous block)
template <typename PT>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, PT* __restrict__ wp, int Cout, int Cin, int KH, int KW,
                                        int nchunks, int cin8, int CoutValid) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)nchunks * 32 * Cout;
  if (idx >= total) return;
  int kin = (int)(idx % 32);
  long t = idx / 32;
  int co = (int)(t % Cout);
  int kc = (int)(t / Cout);
  int kh, kw, c;
  if (cin8) {
    const int t = kc * 4 + (kin >> 3);  // flat tap, row-major over KH x KW (taps past KH*KW: kh >= KH -> zero weight)
    kh = t / KW;
    kw = t - kh * KW;
    c = kin & 7;
  } else {
    int taps = KH * KW;
    int cc = kc / taps;
    int tap = kc - cc * taps;
    c = cc * 32 + kin;
    kh = tap / KW;
    kw = tap - kh * KW;
  }
  float v = 0.f;
  if (kh < KH && kw < KW && c < Cin && co < CoutValid) v = w[(((long)co * Cin + c) * KH + kh) * KW + kw];
  wp[idx] = (PT)v;
}

// FullyConnected weight (out, in) with `in` flattened (c,h,w) [mx Flatten of NCHW] -> packed
// [chunk][out][32] with chunk = (32-channel slice, h, w) so that fc6 runs through conv_fwd_kernel on the NHWC conv6_1 output.
__global__ void pack_fc_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Out, int C, int H, int W) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Out * C * H * W;
  if (idx >= total) return;
  int kin = (int)(idx % 32);
  long t = idx / 32;
  int o = (int)(t % Out);
  long kc = t / Out;                       // chunk = (channel slice, tap) with the tap (h,w) fastest, as in the conv kernel
  int c = (int)(kc / ((long)H * W)) * 32 + kin;
  long hw = kc % ((long)H * W);
  wp[idx] = w[(long)o * C * H * W + (long)c * H * W + hw];
}

// Deconvolution(k=4, s=2, p=0) weight (Cin, Cout, 4, 4) [MXNet layout] -> four packed 2x2 convolution weights, one per output
// phase (py,px): out[2t+py, 2u+px] = sum_{dy,dx} in[t-1+dy, u-1+dx] * w[ci][co][py+2(1-dy)][px+2(1-dx)]   (pad 1, stride 1).
// Cin is zero-padded to CinPad (multiple of 32).  Layout per phase: [chunk][Cout][32], chunk = (channel slice, dy, dx).
template <typename PT>
__global__ void pack_deconv4x4s2_weight_kernel(const float* __restrict__ w, PT* __restrict__ wp, int Cin, int CinPad, int Cout) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_phase = (long)CinPad * 4 * Cout;
  if (idx >= 4 * per_phase) return;
  int phase = (int)(idx / per_phase);
  long r = idx % per_phase;
  int kin = (int)(r % 32);
  long t = r / 32;
  int co = (int)(t % Cout);
  int kc = (int)(t / Cout);
  int cc = kc / 4, tap = kc % 4, dy = tap / 2, dx = tap % 2;
  int ci = cc * 32 + kin;
  int py = phase / 2, px = phase % 2;
  float v = 0.f;
  if (ci < Cin) v = w[(((long)ci * Cout + co) * 4 + (py + 2 * (1 - dy))) * 4 + (px + 2 * (1 - dx))];
  wp[idx] = (PT)v;
}

// Convolution with a handful of output channels (flow / mask heads, Cout <= 2).  HBM/L2-bound on the activations; weights (Cout,Cin,3,3
// MXNet layout) are re-packed to [Cout][kh][kw][CinPad].  A wave owns PX horizontally adjacent output pixels; a lane strides the channel
// quads (float4), and for every quad and kernel row loads the PX + KW - 1 input pixels of the row and the KW x COUT weight quads ONCE
// for all PX outputs; PX x COUT wave reductions at the end.  History at 16 x 30 x 40 x 770 -> 2 / -> 1: one wave per output pixel (1 + COUT
// float4 loads per 4 COUT multiply-adds: 55 KB of weights + 28 KB of activations per pixel through the vector L1): 36 / 50 us; four
// pixels per wave: 34 / 29 us at 60 % of the chip's vector-memory issue rate; the weights staged in LDS per workgroup with eight pixels
// per wave: 57 / 32 us (60 KB of staging for 32 pixels, two workgroups per CU) -- dropped.
template <int COUT, int KW, int PX>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                              int W, int CinPad, int in_cstride, int KH, int pad, int out_cstride,
                                                              int out_coff) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4 = CinPad >> 2;
  const int segs = (W + PX - 1) / PX;   // pixel groups per row
  // XCD-contiguous numbering: neighbouring groups read the same input rows; in launch order they sit on eight different L2s
  const long grp = (long)wg_xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * 4 + wave;
  if (grp >= (long)N * H * segs) return;
  const int wo0 = (int)(grp % segs) * PX;
  const int ho = (int)((grp / segs) % H);
  const int n = (int)(grp / ((long)segs * H));
  float acc[PX][COUT];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[p][c] = 0.f;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kh = 0; kh < KH; ++kh) {
    const int hi = ho - pad + kh;
    if ((unsigned)hi >= (unsigned)H) continue;   // wave-uniform
    const float* xrow = x + (long)(n * H + hi) * W * in_cstride;
    for (int i = lane; i < c4; i += 64) {
      float4 xv[PX + KW - 1], wv[KW][COUT];
#pragma unroll
      for (int q = 0; q < PX + KW - 1; ++q) {
        const int wi = wo0 - pad + q;
        const bool ok = (unsigned)wi < (unsigned)W;   // wave-uniform; clamped address + select keeps the loads branch-free
        const float4 v = reinterpret_cast<const float4*>(xrow + (long)(ok ? wi : 0) * in_cstride)[i];
        xv[q] = ok ? v : zero;
      }
#pragma unroll
      for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int c = 0; c < COUT; ++c) wv[kw][c] = reinterpret_cast<const float4*>(wp + ((long)(c * KH + kh) * KW + kw) * CinPad)[i];
#pragma unroll
      for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
          for (int c = 0; c < COUT; ++c) {
            const float4 v = xv[p + kw], w4 = wv[kw][c];
            acc[p][c] = fmaf(v.x, w4.x, fmaf(v.y, w4.y, fmaf(v.z, w4.z, fmaf(v.w, w4.w, acc[p][c]))));
          }
    }
  }
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      float v = acc[p][c];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0 && wo0 + p < W) y[((long)(n * H + ho) * W + wo0 + p) * out_cstride + out_coff + c] = v + (bias ? bias[c] : 0.f);
    }
}

__global__ void pack_small_cout_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int CinPad,
                                              int KH, int KW) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cout * KH * KW * CinPad;
  if (idx >= total) return;
  int ci = (int)(idx % CinPad);
  long t = idx / CinPad;
  int kw = (int)(t % KW); t /= KW;
  int kh = (int)(t % KH);
  int co = (int)(t / KH);
  wp[idx] = ci < Cin ? w[(((long)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
}

// Deconvolution(k=4, s=2, p=0) on a tiny channel count (upsample_flow6to5 / 5to4: 2 -> 2) + Crop(offset) written into a
// concat buffer.  x (N,H,W,xstride) NHWC; w (Cin,Cout,4,4) MXNet layout; out pixel (oy,ox) <- full-res (oy+crop, ox+crop).
__global__ void deconv4x4s2_tiny_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                        float* __restrict__ y, int N, int H, int W, int Cin, int xstride, int Cout, int OH, int OW,
                                        int crop, int out_cstride, int out_coff) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * OH * OW * Cout;
  if (idx >= total) return;
  int co = (int)(idx % Cout);
  long t = idx / Cout;
  int ox = (int)(t % OW); t /= OW;
  int oy = (int)(t % OH);
  int n = (int)(t / OH);
  int fy = oy + crop, fx = ox + crop;
  float acc = bias ? bias[co] : 0.f;
  for (int ky = fy & 1; ky < 4; ky += 2) {
    int iy = (fy - ky) >> 1;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = fx & 1; kx < 4; kx += 2) {
      int ix = (fx - kx) >> 1;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float* xs = x + ((long)(n * H + iy) * W + ix) * xstride;
      for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xs[ci], w[(((long)ci * Cout + co) * 4 + ky) * 4 + kx], acc);
    }
  }
  y[((long)(n * OH + oy) * OW + ox) * out_cstride + out_coff + co] = acc;
}

// Deconvolution(k=32, s=16, group = C, no bias) + Crop(offset 8,8): the frozen bilinear x16 upsampling of the flow / mask heads
// (deepIM_flownet.py:326-340, :513-529).  x (N,h,w,C) NHWC; wk (C,1,32,32); y (N,C,OH,OW) NCHW planes.
// mode 0: plain * scale   mode 1: sigmoid (mask probability)
__global__ __launch_bounds__(256) void upsample16_kernel(const float* __restrict__ x, const float* __restrict__ wk, float* __restrict__ y,
                                                         int C, int h, int w, int OH, int OW, int crop, float scale, int mode) {
  const int n = blockIdx.z / C, c = blockIdx.z % C;
  const int oy = blockIdx.y;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  if (ox >= OW) return;
  const int fy = oy + crop, fx = ox + crop;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    int iy = (fy >> 4) - a;
    int ky = fy - 16 * iy;  // in [0,32)
    if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int ix = (fx >> 4) - b;
      int kx = fx - 16 * ix;
      if ((unsigned)ix >= (unsigned)w) continue;
      acc = fmaf(x[((long)(n * h + iy) * w + ix) * C + c], wk[((long)c * 32 + ky) * 32 + kx], acc);
    }
  }
  acc *= scale;
  if (mode == 1) acc = 1.f / (1.f + expf(-acc));
  y[(((long)n * C + c) * OH + oy) * OW + ox] = acc;
}

// the same for four adjacent output pixels per thread (crop % 4 == 0, OW % 4 == 0, 16-byte aligned rows): they share their 2 x 2 input
// pixels, the kernel taps are one float4 per (ky, b), the store is one float4.  One output per thread: 32 + 19 us for the two heads at
// 16 x 480 x 640 (1.2 / 1.0 TB/s of output).
__global__ __launch_bounds__(256) void upsample16_x4_kernel(const float* __restrict__ x, const float* __restrict__ wk, float* __restrict__ y,
                                                            int C, int h, int w, int OH, int OW, int crop, float scale, int mode) {
  const int n = blockIdx.z / C, c = blockIdx.z % C;
  const int oy = blockIdx.y;
  const int ox = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (ox >= OW) return;
  const int fy = oy + crop, fx = ox + crop;   // fx % 4 == 0: fx .. fx + 3 share their 16-block
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int iy = (fy >> 4) - a;
    const int ky = fy - 16 * iy;  // in [0,32)
    if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ix = (fx >> 4) - b;
      const int kx = fx - 16 * ix;   // multiple of 4
      if ((unsigned)ix >= (unsigned)w) continue;
      const float xv = x[((long)(n * h + iy) * w + ix) * C + c];
      const float4 k4 = *reinterpret_cast<const float4*>(wk + ((long)c * 32 + ky) * 32 + kx);
      acc.x = fmaf(xv, k4.x, acc.x);
      acc.y = fmaf(xv, k4.y, acc.y);
      acc.z = fmaf(xv, k4.z, acc.z);
      acc.w = fmaf(xv, k4.w, acc.w);
    }
  }
  acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
  if (mode == 1) {
    acc.x = 1.f / (1.f + expf(-acc.x)); acc.y = 1.f / (1.f + expf(-acc.y));
    acc.z = 1.f / (1.f + expf(-acc.z)); acc.w = 1.f / (1.f + expf(-acc.w));
  }
  *reinterpret_cast<float4*>(y + (((long)n * C + c) * OH + oy) * OW + ox) = acc;
}

// Pose head: fc7 + LeakyReLU + rot (4) + trans (3) + inverse ZoomTrans -> se3 (B,7).
// deepIM_flownet.py:203-208, :956-971; zoom_trans.py:37-41 (b_inv_zoom: dx*wx, dy*wx).
__global__ __launch_bounds__(1024) void pose_head_kernel(const float* __restrict__ fc6, const float* __restrict__ w7,
                                                         const float* __restrict__ b7, const float* __restrict__ wr,
                                                         const float* __restrict__ br, const float* __restrict__ wt,
                                                         const float* __restrict__ bt, const float* __restrict__ zoom_factor,
                                                         float* __restrict__ se3, float* __restrict__ fc7_out) {
  __shared__ float s_h[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  // fc7 (256 x 256): for each output the 64 lanes of a wave read the weight row as one coalesced 1 KB load (a float4 per lane against
  // the lane's own four fc6 values) and fold their partial dots with a fixed shuffle tree.  (The first version gave every thread one
  // output and let it walk its row alone: 64 cache lines per wave-load, 256 loads per thread, 15-16 us for 16 samples.)
  // (16-byte loads when fc6 and w7 are 16-byte aligned; a flat parameter blob may place w7 on any 4-byte boundary: scalar loads then)
  const bool al = ((reinterpret_cast<uintptr_t>(fc6) | reinterpret_cast<uintptr_t>(w7)) & 15) == 0;
  const float* xr = fc6 + (long)b * 256 + 4 * lane;
  const float4 xin = al ? *reinterpret_cast<const float4*>(xr) : make_float4(xr[0], xr[1], xr[2], xr[3]);
  // 16 waves x 16 outputs: every wave issues its 16 row loads before the first use -- one memory round trip for the layer (with 4 rows
  // at a time on 4 waves the kernel still took 16 us: sixteen round trips in series)
  {
    const int o0 = wave * 16;
    float4 wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float* wr_ = w7 + (long)(o0 + u) * 256 + 4 * lane;
      wv[u] = al ? *reinterpret_cast<const float4*>(wr_) : make_float4(wr_[0], wr_[1], wr_[2], wr_[3]);
    }
    float p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) p[u] = fmaf(xin.w, wv[u].w, fmaf(xin.z, wv[u].z, fmaf(xin.y, wv[u].y, xin.x * wv[u].x)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < 16; ++u) p[u] += __shfl_down(p[u], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        float acc = p[u] + b7[o0 + u];
        acc = acc > 0.f ? acc : 0.1f * acc;
        s_h[o0 + u] = acc;
        if (fc7_out) fc7_out[(long)b * 256 + o0 + u] = acc;
      }
    }
  }
  __syncthreads();
  // 7 outputs, one wave each would be overkill: 7 x 64-lane partial dot + shuffle reduce
  for (int o = wave; o < 7; o += 16) {
    const float* wv = (o < 4) ? (wr + o * 256) : (wt + (o - 4) * 256);
    float p = 0.f;
    for (int k = lane; k < 256; k += 64) p = fmaf(s_h[k], wv[k], p);
    for (int off = 32; off > 0; off >>= 1) p += __shfl_down(p, off, 64);
    if (lane == 0) {
      float v = p + ((o < 4) ? br[o] : bt[o - 4]);
      if (o == 4 || o == 5) v = v * zoom_factor[b * 4 + 0];
      se3[b * 7 + o] = v;
    }
  }
}

template <int BM, int BN, int WM, int WN, bool CIN8>
static int launch_conv(const ConvArgs& a, int splits, hipStream_t st, int batch = 1, int tile_begin = 0, int tile_count = -1) {
  // A tiles only: the weights go global -> registers.  f32: [2][BM][36] floats; bf16: [2][BM][40] halves
  const size_t lds = a.bf16 ? (size_t)2 * BM * (32 + 8) * 2 : (size_t)2 * BM * (32 + 4) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<BM, BN, WM, WN, CIN8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * BM * (32 + 4) * sizeof(float)));
    if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int tiles = ceil_div(a.M, BM) * (a.Cout / BN);
  ConvArgs b = a;
  static const int xcd_mode = getenv("DIM_CONV_XCD") ? atoi(getenv("DIM_CONV_XCD")) : 0;  // experiment switch
  const bool whole = tile_begin == 0 && (tile_count < 0 || tile_count == tiles);
  b.xcd_chunk = (whole && xcd_mode > 0 && tiles % 8 == 0 && tiles >= xcd_mode) ? tiles / 8 : 0;
  b.tile_off = tile_begin;
  dim3 grid(tile_count < 0 ? tiles : tile_count, batch, splits);
  if (a.bf16)
    hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, WM, WN, CIN8>), grid, dim3(WM * WN * 64), lds, st, b);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, CIN8>), grid, dim3(WM * WN * 64), lds, st, b);
  return check_launch("conv_fwd");
}

// bf16-only workgroup tiles (no f32 instantiation): 8 = 128 rows x 256 channels on 8 waves of 64 x 64
template <int BM, int BN, int WM, int WN>
static int launch_conv_bf16(const ConvArgs& a, int splits, hipStream_t st, int batch, int tile_begin, int tile_count) {
  const size_t lds = (size_t)2 * BM * (32 + 8) * 2;
  const int tiles = ceil_div(a.M, BM) * (a.Cout / BN);
  ConvArgs b = a;
  b.xcd_chunk = 0;
  b.tile_off = tile_begin;
  dim3 grid(tile_count < 0 ? tiles : tile_count, batch, splits);
  hipLaunchKernelGGL((conv_bf16_kernel<BM, BN, WM, WN, false>), grid, dim3(WM * WN * 64), lds, st, b);
  return check_launch("conv_bf16");
}

}  // namespace dim

using namespace dim;

extern "C" {

long dim_conv2d_packed_weight_floats(int Cout, int Cin, int KH, int KW) {
  if (Cin == 8)  // 4 taps x 8 channels per chunk, taps flat over KH x KW; the first layer's three-term image behind it
    return (long)((KH * KW + 3) / 4) * 32 * Cout + (KH == 7 && KW == 7 && Cout == 64 ? (long)(kC1SplitBytes / 4) : 0);
  return (long)KH * KW * Cin * Cout;
}

// OIHW -> [chunk][CoutPad][32], rows >= Cout zero.  Cin % 32 == 0: tiled (workgroup = (32-channel slice, G output channels):
// rows = w[co][cc * 32 + r][tap], packed run tap at ((cc * T + tap) * CoutPad + co) * 32); the 8-channel first layer: per element
extern "C++" template <typename PT>
int pack_conv_weight_any(const float* w_oihw, PT* w_packed, int Cout, int CoutPad, int Cin, int KH, int KW, void* stream) {
  const int cin8 = Cin == 8, T = KH * KW;
  const int nchunks = cin8 ? (T + 3) / 4 : T * (Cin / 32);
  const int G = cin8 ? 0 : wtile_group(CoutPad, T, Cin / 32);
  if (G) {
    WTileArgs a = {};
    a.src = w_oihw; a.dst = w_packed;
    a.G = G; a.Q = T; a.gmax = Cout; a.rmax = Cin; a.g_fast = 0; a.nj = 0;
    a.sg = (long)Cin * T; a.sr = T; a.rows_x = 32L * T; a.rows_y = (long)G * Cin * T;
    a.dq = (long)CoutPad * 32; a.packed_x = (long)T * CoutPad * 32;
    wtile_launch<true, PT>(a, Cin / 32, CoutPad / G, as_stream(stream));
  } else {
    long total = (long)nchunks * 32 * CoutPad;
    hipLaunchKernelGGL((pack_conv_weight_kernel<PT>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, CoutPad,
                       Cin, KH, KW, nchunks, cin8, Cout);
  }
  return check_launch("pack_conv_weight");
}

int dim_conv2d_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null weight pointer");
  DIM_REQUIRE(Cin == 8 || Cin % 32 == 0, "Cin must be 8 or a multiple of 32 (got %d)", Cin);
  DIM_REQUIRE(Cin != 8 || KW <= 8, "Cin==8 path needs KW<=8 (got %d)", KW);
  int rc = pack_conv_weight_any(w_oihw, w_packed, Cout, Cout, Cin, KH, KW, stream);
  if (rc == DIM_OK && Cin == 8 && KH == 7 && KW == 7 && Cout == 64) {   // flow_conv1: + the image conv1_halo_split_kernel reads
    hipLaunchKernelGGL(conv1_split_weights_kernel, dim3((64 * 2 * kC1Pairs + 255) / 256), dim3(256), 0, as_stream(stream), w_packed,
                       reinterpret_cast<unsigned char*>(w_packed + 13 * 32 * 64));
    rc = check_launch("conv1_split_weights");
  }
  return rc;
}

// same, with the output-channel count padded with zero rows up to CoutPad (a multiple of 64): w_oihw has Cout rows
int dim_conv2d_pack_weight_padded(const float* w_oihw, float* w_packed, int Cout, int CoutPad, int Cin, int KH, int KW, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null weight pointer");
  DIM_REQUIRE(Cin % 32 == 0 && CoutPad >= Cout && CoutPad % 64 == 0, "Cin %% 32 == 0 and CoutPad a multiple of 64 >= Cout required");
  return pack_conv_weight_any(w_oihw, w_packed, Cout, CoutPad, Cin, KH, KW, stream);
}

// the bf16 image of the same packed array in one pass (== dim_f32_to_bf16 of dim_conv2d_pack_weight_padded's output); CoutPad == Cout
// for an unpadded layer
int dim_conv2d_pack_weight_bf16(const float* w_oihw, void* w_packed_bf16, int Cout, int CoutPad, int Cin, int KH, int KW, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed_bf16, "null weight pointer");
  DIM_REQUIRE(Cin == 8 || Cin % 32 == 0, "Cin must be 8 or a multiple of 32 (got %d)", Cin);
  DIM_REQUIRE(Cin != 8 || KW <= 8, "Cin==8 path needs KW<=8 (got %d)", KW);
  DIM_REQUIRE(CoutPad >= Cout, "CoutPad < Cout");
  return pack_conv_weight_any(w_oihw, reinterpret_cast<__bf16*>(w_packed_bf16), Cout, CoutPad, Cin, KH, KW, stream);
}

int dim_fc_pack_weight(const float* w_out_in, float* w_packed, int Out, int C, int H, int W, void* stream) {
  DIM_REQUIRE(w_out_in && w_packed, "null weight pointer");
  long total = (long)Out * C * H * W;
  const int HW = H * W, G = C % 32 == 0 ? wtile_group(Out, HW, C / 32) : 0;
  if (G) {  // workgroup (channel slice cb, G outputs): rows = w[o][cb * 32 + r][q], packed run q at ((cb * HW + q) * Out + o) * 32
    WTileArgs a = {};
    a.src = w_out_in; a.dst = w_packed;
    a.G = G; a.Q = HW; a.gmax = Out; a.rmax = C; a.g_fast = 0; a.nj = 0;
    a.sg = (long)C * HW; a.sr = HW; a.rows_x = 32L * HW; a.rows_y = (long)G * C * HW;
    a.dq = (long)Out * 32; a.packed_x = (long)HW * Out * 32;
    wtile_launch<true, float>(a, C / 32, Out / G, as_stream(stream));
  } else {
    hipLaunchKernelGGL(pack_fc_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_out_in, w_packed, Out,
                       C, H, W);
  }
  return check_launch("pack_fc_weight");
}

// splits == 0 ("auto", see dim_conv2d_fwd): room for the split tail (fewer than one tile per CU on a 304-CU part at most, <= 8 slabs)
static const long kTailWorkspaceFloats = 8L * 304 * 128 * 128;

long dim_conv2d_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int splits) {
  int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (splits == 0) {
    long full = 8L * N * Ho * Wo * Cout;
    return full < kTailWorkspaceFloats ? full : kTailWorkspaceFloats;
  }
  if (splits <= 1) return 0;
  return (long)splits * N * Ho * Wo * Cout;
}

// Plan of the "auto" mode.  All workgroups of a launch are equal, so the launch takes ceil(tiles / CUs) tile-times on the busiest
// CU while the average CU has tiles / CUs of work: conv3 at batch 16 = 1200 tiles on 256 CUs = 4.69 -> 5, i.e. 6 % of the machine
// idles.  The plan runs k*CUs tiles (k whole tiles per CU) as one launch and the remaining `tail` tiles as a second, split-K
// launch of tail*s workgroups of 1/s the length, s chosen to minimise ceil(tail*s / CUs) / s; the slabs are summed by
// dim_splitk_reduce.  -> first tile of the tail (a multiple of nt, so the tail is a row range) and s (1 = single launch).
static void conv_tail_plan(int M, int Cout, int nchunks, int tile, int* tail_begin, int* tail_splits) {
  const int BM = tile == 3 ? 64 : 128, BN = (tile == 3 || tile == 2) ? 64 : 128;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      n_cu = 256;
  }
  const int mt = ceil_div(M, BM), nt = Cout / BN, tiles = mt * nt;
  *tail_begin = tiles;
  *tail_splits = 1;
  const int k = tiles / n_cu;
  if (k < 1) return;                       // small layers: plain split-K chosen by the caller
  const int full = (k * n_cu) / nt * nt;
  const int tail = tiles - full;
  if (tail == 0) return;
  const double single = (double)ceil_div(tiles, n_cu);
  double best = 1e30;
  int best_s = 1;
  for (int s = 2; s <= 8 && s * 4 <= nchunks; ++s) {
    if ((long)s * (M - full / nt * BM) * Cout > kTailWorkspaceFloats) break;
    double t = (double)ceil_div((long)tail * s, n_cu) / s + 0.02 * s;  // 2 % of a tile-time per slab: extra prologues, slab traffic, reduce
    if (t < best) { best = t; best_s = s; }
  }
  if (k + best < single * 0.98) {
    *tail_begin = full;
    *tail_splits = best_s;
  }
}

// ONE copy of the default launch plans of the f32 forward path: the Python executor (FlowNetHip) and the C resident loop
// (dim_refiner_create) both call these, so the two cannot drift apart.
// (tile, splits) of a direct layer (tools/tune_conv.py, batch 16, MI355X): 128 x 128 tiles on 8 waves (tile 4) wherever Cout allows,
// 64 x 64 (tile 3) for Cout = 64 and the 8-channel first layer; splits 0 = "auto" (whole tiles per CU + split-K tail inside the
// library) when every CU gets at least one tile, otherwise the split-K count that minimises the busiest CU's share: all workgroups
// of a launch are equal, the busiest CU gets ceil(tiles s / CUs) of them, each 1 / s of a tile-time long, + 2 % of a tile-time per
// slab for the extra prologues, slab traffic and the reduce.
int dim_conv_auto_plan(long M, int Cout, int nchunks, int cin, int* tile, int* splits) {
  DIM_REQUIRE(tile && splits && M > 0 && Cout > 0 && nchunks > 0, "bad arguments");
  const int n_cu = 256;
  auto best_split = [&](long tiles, int smax) {
    double best = 1e30;
    int bs = 1;
    for (int s = 1; s <= smax; ++s) {
      if (!(s * 4 <= nchunks || s == 1)) continue;
      const double t = (double)((tiles * s + n_cu - 1) / n_cu) / s + 0.02 * s;
      if (t < best) { best = t; bs = s; }
    }
    return bs;
  };
  const bool wide = Cout % 128 == 0 && cin != 8;
  const long tiles = wide ? (M + 127) / 128 * (Cout / 128) : (M + 63) / 64 * (Cout / 64);
  *tile = wide ? 4 : 3;
  *splits = tiles >= n_cu ? 0 : best_split(tiles, 8);
  return DIM_OK;
}

// Plane GEMMs with every f32 operand as three bf16 terms and six MFMA products (wino_gemm_split.hip; default on, DIM_WINO_SPLIT=0 or
// dim_set_winograd_split(0) = the f32 matrix pipe).  Read when a layer is planned, i.e. at the next launch / graph capture.
int dim_set_winograd_split(int on) {
  wino_set_split(on ? 1 : 0);
  return DIM_OK;
}
int dim_get_winograd_split(void) { return wino_get_split(); }

// workgroup tile of a Winograd layer's plane GEMMs (wino_gemm.hip): 5 = 128 rows x 256 output channels (V is streamed once per 256
// channels: conv3, conv3_1, conv4_1), 4 = 128 x 128 (conv2: Cout = 128); for the few-row layers (under 1024 tile rows: conv5 .. conv6_1)
// the tile that wastes the fewest MFMA cycles on padded rows -- 6 = 160 x 128 (conv5, conv5_1: 320 rows at 16 pairs), 7 = 96 x 128
// (conv6_1: 96 rows), 3 = 64 x 64 -- weighted by what each reaches of the matrix peak (measured: 0.54 / 0.70 / 0.74).
// DIM_WINO_BN256=0 keeps the 128 x 128 tile everywhere, DIM_WINO_FEWROW=0 the 64 x 64 tile on the few-row layers (A/B timing).
int dim_winograd_gemm_tile_planes(int Cout, long tiles, int planes) {
  static const int bn256 = [] { const char* e = getenv("DIM_WINO_BN256"); return e ? atoi(e) : 1; }();
  static const int fewrow = [] { const char* e = getenv("DIM_WINO_FEWROW"); return e ? atoi(e) : 1; }();
  if (Cout % 128) return 3;
  if (tiles < 1024) {
    if (!fewrow) return 3;
    if (wino_get_split()) {
      // three-term kernels (wino_gemm_split.hip): 128 x 128 (tile 4) or 96 x 128 (tile 7), two 4-wave workgroups per CU.  Cost = rows a
      // workgroup multiplies: items per workgroup (whole workgroups per CU when there are fewer items than slots, as wino_gemm_plan
      // deals them) x tile rows / what the tile reaches (measured at 16 pairs: conv5_1 51.8 us on 7 against 57.2 on 4 and 73.7 on the
      // f32 pipe; conv5, 81 planes, 93.6 on 4 against 110.0 on 7 and 140.8 on the f32 pipe)
      const struct { int tile, bm; double eff; } cand[2] = {{4, 128, 1.0}, {7, 96, planes > 36 ? 0.75 : 0.85}};   // (conv6, 81 planes x 96 rows: 102 us on 4, 116 on 7)
      const int cus = 256, slots = 512;
      int best = 7;
      double best_cost = 1e30;
      for (const auto& c : cand) {
        const long items = (tiles + c.bm - 1) / c.bm * (Cout / 128) * planes;
        long G = items < slots ? items : slots;
        if (G > cus && G < slots) G = G / cus * cus;
        const double cost = (double)items / (double)G * c.bm / c.eff;
        if (cost < best_cost) { best_cost = cost; best = c.tile; }
      }
      return best;
    }
    const struct { int tile, bm; double eff; } cand[3] = {{3, 64, 0.54}, {7, 96, 0.70}, {6, 160, 0.74}};
    int best = 3;
    double best_cost = 1e30;
    for (const auto& c : cand) {
      const double cost = (double)((tiles + c.bm - 1) / c.bm * c.bm) / c.eff;
      if (cost < best_cost) { best_cost = cost; best = c.tile; }
    }
    return best;
  }
  static const int big = [] { const char* e = getenv("DIM_WINO_BIG_TILE"); return e ? atoi(e) : 0; }();   // A/B timing: 6 or 7 on the big layers
  if (big == 6 || big == 7) return big;
  return (Cout % 256 == 0 && bn256) ? 5 : 4;
}
int dim_winograd_gemm_tile(int Cout, long tiles) { return dim_winograd_gemm_tile_planes(Cout, tiles, 36); }

int dim_conv2d_tail_plan(int M, int Cout, int Cin, int KH, int KW, int tile, int* tail_begin_tile, int* tail_splits) {
  DIM_REQUIRE(tail_begin_tile && tail_splits, "null pointer");
  if (tile == 0) tile = (Cout % 128 == 0 && Cin != 8 && M >= 128) ? 4 : 3;
  conv_tail_plan(M, Cout, (Cin == 8) ? (KH * KW + 3) / 4 : KH * KW * (Cin / 32), tile, tail_begin_tile, tail_splits);
  return DIM_OK;
}

int dim_splitk_reduce(const float* slabs, const float* bias, float* y, long M, int Cout, int splits, float slope, void* stream) {
  if (M == 0) return DIM_OK;
  DIM_REQUIRE(slabs && y, "null pointer");
  DIM_REQUIRE(Cout % 4 == 0 && splits >= 1, "bad geometry");
  long MC = M * Cout;
  if (MC / 4 < 65536 && splits >= 64)  // fewer than 256 workgroups of serial sums: spread the slabs over lanes instead
    hipLaunchKernelGGL(splitk_reduce_par_kernel, dim3(ceil_div(MC / 4, 8)), dim3(256), 0, as_stream(stream), slabs, bias, y, MC, Cout,
                       splits, slope, bias != nullptr);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div(MC / 4, 256)), dim3(256), 0, as_stream(stream), slabs, bias, y, MC, Cout,
                       splits, slope, bias != nullptr);
  return check_launch("splitk_reduce");
}

// tile: 0 = auto, 1 = 128x128 (4 waves), 2 = 128x64, 3 = 64x64, 4 = 128x128 with 8 waves (64x32 per wave)
struct ConvEx {
  int in_cstride, out_cstride, out_coff, OH, OW, osy, osx, ooy, oox;  // 0 / 0 / 0 / 0.. = dense defaults
  int accumulate = 0;    // out += result (skip-connection gradients)
  int pad_w = -1;        // >= 0: horizontal padding differs from `pad` (sub-pixel phases of a strided dgrad)
  int Ho = 0, Wo = 0;    // > 0: explicit output grid instead of floor((H+2p-k)/s)+1 (asymmetric padding)
  int batch = 1;         // > 1: `batch` independent problems, strides below (elements)
  long bx = 0, bw = 0, by = 0;
  int boy = 0, box = 0;  // scattered output: per-problem offset increments (see ConvArgs)
  int bf16 = 0;          // w_packed holds bf16 (dim_f32_to_bf16 of the f32 packed array): run on the bf16 matrix pipe
  int slab_full = 0;     // split-K into output-shaped slabs (ConvArgs.slab_full): the caller sums them (slab_sum_rows)
  const float* mask = nullptr;   // tile 9 only: LeakyReLU' of the layer below + its bias-gradient partial sums (ConvArgs.mask)
  float mask_slope = 1.f;
  float* colsum = nullptr;
  int colsum_row0 = 0;
};

static int conv2d_fwd_impl(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                           int Cin, int Cout, int KH, int KW, int stride, int pad, float slope, int splits, int tile,
                           int partial_only, void* stream, const ConvEx* ex = nullptr) {
  if (N == 0) return DIM_OK;  // empty batch
  DIM_REQUIRE(x && w_packed && y, "null pointer");
  DIM_REQUIRE(Cin == 8 || Cin % 32 == 0, "Cin must be 8 or a multiple of 32 (got %d)", Cin);
  DIM_REQUIRE(Cout % 64 == 0, "Cout must be a multiple of 64 (got %d)", Cout);
  DIM_REQUIRE(stride >= 1 && pad >= 0 && KH >= 1 && KW >= 1, "bad geometry");
  DIM_REQUIRE(Cin != 8 || KW <= 8, "Cin==8 path needs KW<=8");
  ConvArgs a;
  a.y_bytes = 0;
  a.x = x; a.w = w_packed; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride;
  a.pad_h = pad;
  a.pad_w = (ex && ex->pad_w >= 0) ? ex->pad_w : pad;
  a.Ho = (ex && ex->Ho > 0) ? ex->Ho : (H + 2 * pad - KH) / stride + 1;
  a.Wo = (ex && ex->Wo > 0) ? ex->Wo : (W + 2 * a.pad_w - KW) / stride + 1;
  DIM_REQUIRE(a.Ho > 0 && a.Wo > 0, "empty output");
  a.accumulate = ex ? ex->accumulate : 0;
  a.in_cstride = (ex && ex->in_cstride) ? ex->in_cstride : Cin;
  a.out_cstride = (ex && ex->out_cstride) ? ex->out_cstride : Cout;
  a.out_coff = ex ? ex->out_coff : 0;
  a.dense_out = !(ex && ex->osy);
  a.OH = a.dense_out ? a.Ho : ex->OH; a.OW = a.dense_out ? a.Wo : ex->OW;
  a.osy = a.dense_out ? 1 : ex->osy; a.osx = a.dense_out ? 1 : ex->osx;
  a.ooy = a.dense_out ? 0 : ex->ooy; a.oox = a.dense_out ? 0 : ex->oox;
  DIM_REQUIRE(a.in_cstride >= Cin && a.in_cstride % 4 == 0, "in_cstride must be >= Cin and a multiple of 4");
  DIM_REQUIRE(a.out_cstride >= a.out_coff + Cout, "out_cstride < out_coff + Cout");
  DIM_REQUIRE((long)N * H * W * a.in_cstride < (1L << 29), "input too large for 32-bit byte offsets (%ld elements)",
              (long)N * H * W * a.in_cstride);
  a.x_bytes = (unsigned)((long)N * H * W * a.in_cstride * 4);
  a.M = N * a.Ho * a.Wo;
  a.nchunks = (Cin == 8) ? (KH * KW + 3) / 4 : KH * KW * (Cin / 32);
  a.div_kw = make_fastdiv((unsigned)KW);
  DIM_REQUIRE((long)a.nchunks * Cout * 32 * 4 < (1L << 31), "packed weights too large for 32-bit byte offsets");
  a.bf16 = ex ? ex->bf16 : 0;
  a.w_bytes = (unsigned)((long)a.nchunks * Cout * 32 * (a.bf16 ? 2 : 4));
  const bool auto_split = splits == 0;
  if (splits < 1) splits = 1;
  if (splits > a.nchunks) splits = a.nchunks;
  a.chunks_per_split = (a.nchunks + splits - 1) / splits;
  splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
  DIM_REQUIRE(splits == 1 || workspace, "split-K needs a workspace (dim_conv2d_workspace_floats)");
  a.slab_full = (ex && ex->slab_full && splits > 1) ? 1 : 0;
  a.mask = ex ? ex->mask : nullptr;
  a.mask_slope = ex ? ex->mask_slope : 1.f;
  a.colsum = ex ? ex->colsum : nullptr;
  a.colsum_row0 = ex ? ex->colsum_row0 : 0;
  DIM_REQUIRE(!a.mask || (tile == 9 && a.colsum && !a.accumulate && splits == 1), "the LeakyReLU' / bias-gradient fold exists in tile 9 only");
  DIM_REQUIRE(splits == 1 || !ex || a.slab_full || (a.dense_out && a.out_cstride == Cout && a.out_coff == 0),
              "split-K writes a dense [M][Cout] result: not available with a strided / scattered output");
  DIM_REQUIRE(!a.slab_full || (a.bf16 && (tile == 3 || tile == 4) && (!ex || ex->batch == 1)),
              "output-shaped split-K slabs: bf16 gathered-tap kernel (tile 3 / 4), single problem");
  a.y = splits > 1 ? workspace : y;
  // every kernel stores through a buffer descriptor with 32-bit byte offsets (branch-free epilogues)
  DIM_REQUIRE((long)N * a.OH * a.OW * a.out_cstride * 4 < (1L << 31), "output too large for 32-bit byte offsets (%ld bytes)",
              (long)N * a.OH * a.OW * a.out_cstride * 4);
  a.y_bytes = (unsigned)((long)N * a.OH * a.OW * a.out_cstride * 4);
  DIM_REQUIRE(splits == 1 || (long)a.M * Cout * 4 < (1L << 31), "split-K slab too large for 32-bit byte offsets (%ld bytes)", (long)a.M * Cout * 4);
  a.tile_off = 0;
  a.slab_row0 = 0;
  a.slab_stride = a.slab_full ? (long)N * a.OH * a.OW * a.out_cstride : (long)N * a.Ho * a.Wo * Cout;
  const int batch = ex ? ex->batch : 1;
  a.bx = ex ? ex->bx : 0; a.bw = ex ? ex->bw : 0; a.by = ex ? ex->by : 0;
  a.boy = ex ? ex->boy : 0; a.box = ex ? ex->box : 0;
  DIM_REQUIRE(batch == 1 || splits == 1, "batched launch does not combine with split-K");
  a.slope = slope;
  a.has_bias = bias != nullptr;
  hipStream_t st = as_stream(stream);
  if (tile == 0) {
    tile = (Cout % 128 == 0 && Cin != 8 && a.M >= 128) ? 4 : 3;  // same rule as lib/hip/ops.py conv_auto_plan
  }
  if (tile == 7) {
    // the bf16 LDS-halo kernel (conv_bf16_halo_kernel): square 3x3 or 5x5 taps, stride 1 or 2, Cin % 32 == 0, Cout % 128 == 0, dense
    DIM_REQUIRE(a.bf16 && KH == KW && (KH == 3 || KH == 5) && (stride == 1 || stride == 2) && Cin % 32 == 0 && Cout % 128 == 0,
                "tile 7: bf16, 3x3 or 5x5, stride 1 or 2, Cin %% 32 == 0, Cout %% 128 == 0");
    DIM_REQUIRE(!(KH == 5 && stride == 1), "tile 7: 5x5 is built for stride 2");
    DIM_REQUIRE(splits == 1 && batch == 1 && a.dense_out && !partial_only, "tile 7: dense single-launch output only");
    const int blocks = N * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16) * (Cout / 128);
    const int ph = 7 * stride + KH, pw = 15 * stride + KW;
    const size_t lds = (size_t)((ph * pw * 40 + 7) / 8 * 8) * 2 + (size_t)2 * 2 * 128 * 40 * 2;
#define DIM_HALO16(K, S)                                                                                                              \
  {                                                                                                                                   \
    static bool attr_set = false;                                                                                                     \
    if (!attr_set) {                                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_kernel<K, S>),                                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
      if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));             \
      attr_set = true;                                                                                                                \
    }                                                                                                                                 \
    hipLaunchKernelGGL((conv_bf16_halo_kernel<K, S>), dim3(blocks), dim3(256), lds, st, a);                                           \
  }
    if (KH == 3 && stride == 1) DIM_HALO16(3, 1)
    else if (KH == 3) DIM_HALO16(3, 2)
    else DIM_HALO16(5, 2)
#undef DIM_HALO16
    return check_launch("conv_bf16_halo");
  }
  if (tile == 9) {
    // the bf16 patch kernel (conv_bf16_patch_kernel).  Stride 1: 2 .. 9 taps with KH, KW <= 3, 16 x 16 output pixels x 128 or 64
    // channels per workgroup; stride 2: 3x3 or 5x5, 8 x 16 pixels x 128 channels.  Cin % 32 == 0; dense or scattered output, batched
    // launch (deconvolution phases) allowed, no split-K
    DIM_REQUIRE(a.bf16 && Cin % 32 == 0 && splits == 1 && !partial_only, "tile 9: bf16, Cin %% 32 == 0, no split-K");
    if (stride == 1)
      DIM_REQUIRE(KH >= 1 && KH <= 3 && KW >= 1 && KW <= 3 && KH * KW >= 2 && Cout % 64 == 0,
                  "tile 9, stride 1: 2..9 taps (KH, KW <= 3), Cout %% 64 == 0");
    else
      DIM_REQUIRE(stride == 2 && KH == KW && (KH == 3 || KH == 5) && Cout % 128 == 0, "tile 9, stride 2: 3x3 or 5x5, Cout %% 128 == 0");
    const int bn = Cout % 128 == 0 ? 128 : 64;   // 64: the 64-channel layers (input gradient of flow_conv2)
    const int th = stride == 1 ? 16 : 8;
    const int blocks = N * ((a.Ho + th - 1) / th) * ((a.Wo + 15) / 16) * (Cout / bn);
    const size_t lds = (size_t)2 * ((th - 1) * stride + KH) * (1536 * stride) + 256;   // two patch buffers + the dump slot
#define DIM_PATCH16_BN(KHc, KWc, BNc, Sc)                                                                                             \
  {                                                                                                                                   \
    static bool attr_set = false;                                                                                                     \
    if (!attr_set) {                                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_patch_kernel<KHc, KWc, BNc, Sc>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
      if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));             \
      attr_set = true;                                                                                                                \
    }                                                                                                                                 \
    hipLaunchKernelGGL((conv_bf16_patch_kernel<KHc, KWc, BNc, Sc>), dim3(blocks, batch), dim3(256), lds, st, a);                      \
  }
#define DIM_PATCH16(KHc, KWc)                                                                                                         \
  {                                                                                                                                   \
    if (bn == 128) DIM_PATCH16_BN(KHc, KWc, 128, 1) else DIM_PATCH16_BN(KHc, KWc, 64, 1)                                              \
  }
    if (stride == 2) {
      if (KH == 5) DIM_PATCH16_BN(5, 5, 128, 2) else DIM_PATCH16_BN(3, 3, 128, 2)
    } else {
      switch (KH * 4 + KW) {
        case 1 * 4 + 2: DIM_PATCH16(1, 2) break;
        case 2 * 4 + 1: DIM_PATCH16(2, 1) break;
        case 1 * 4 + 3: DIM_PATCH16(1, 3) break;
        case 3 * 4 + 1: DIM_PATCH16(3, 1) break;
        case 2 * 4 + 2: DIM_PATCH16(2, 2) break;
        case 2 * 4 + 3: DIM_PATCH16(2, 3) break;
        case 3 * 4 + 2: DIM_PATCH16(3, 2) break;
        default: DIM_PATCH16(3, 3) break;
      }
    }
#undef DIM_PATCH16_BN
#undef DIM_PATCH16
    return check_launch("conv_bf16_patch");
  }
  if (tile == 6) {
    // the LDS-halo first-layer kernel (conv1_halo_kernel): 8 channels, 7x7 / stride 2, 64 output channels, dense output, no split-K
    DIM_REQUIRE(Cin == 8 && KH == 7 && KW == 7 && stride == 2 && Cout == 64, "tile 6 is the 8-channel 7x7 / stride-2 / 64-filter first layer");
    DIM_REQUIRE(splits == 1 && batch == 1 && a.dense_out && !a.accumulate && !partial_only, "tile 6: dense single-launch output only");
    const int tiles = N * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    DIM_REQUIRE(a.out_cstride % 4 == 0 && a.out_coff % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                "tile 6 stores float4: output channel stride / offset must be multiples of 4 and y 16-byte aligned");
    if (a.bf16) {  // persistent: one 8-wave workgroup per CU walks a contiguous range of 16 x 16 blocks
      const int tiles16 = N * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
      static const int n_cu = [] {
        hipDeviceProp_t prop;
        int dev = 0;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
      }();
      const int wgs = tiles16 < n_cu ? tiles16 : n_cu;
      const int per_wg = (tiles16 + wgs - 1) / wgs;
      constexpr size_t lds = (size_t)13 * 64 * 40 * 2 + 2 * (size_t)37 * 37 * 16;
      static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_halo_bf16_kernel<7, 7>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      DIM_REQUIRE(attr_ok, "cannot reserve %zu bytes of LDS for the first-layer kernel", lds);
      hipLaunchKernelGGL((conv1_halo_bf16_kernel<7, 7>), dim3((tiles16 + per_wg - 1) / per_wg), dim3(512), lds, st, a, tiles16, per_wg);
    } else if (wino_get_split()) {  // three-term arithmetic: persistent, one 8-wave workgroup per CU, channel halves in pairs
      const int tiles16 = N * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
      static const int n_cu2 = [] {
        int dev = 0, cus = 0;
        return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 1) ? cus : 256;
      }();
      const int pairs = tiles16 < n_cu2 / 2 ? tiles16 : n_cu2 / 2;
      const int per_pair = (tiles16 + pairs - 1) / pairs;
      constexpr size_t lds = (size_t)kC1Pairs * 3 * 2 * 32 * 16 + 3 * 2 * (size_t)37 * 24 * 16 + 32 * sizeof(float);
      static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_halo_split_kernel<7, 7>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      DIM_REQUIRE(attr_ok, "cannot reserve %zu bytes of LDS for the first-layer kernel", lds);
      hipLaunchKernelGGL((conv1_halo_split_kernel<7, 7>), dim3(2 * ((tiles16 + per_pair - 1) / per_pair)), dim3(512), lds, st, a,
                         reinterpret_cast<const unsigned char*>(a.w + 13 * 32 * 64), tiles16, per_pair);
    } else {
      hipLaunchKernelGGL((conv1_halo_kernel<7, 7>), dim3(tiles), dim3(256), 0, st, a);
    }
    return check_launch("conv1_halo");
  }
  DIM_REQUIRE((tile != 1 && tile != 4) || Cout % 128 == 0, "tile 128x128 needs Cout %% 128 == 0");
  DIM_REQUIRE(tile != 8 || (a.bf16 && Cout % 256 == 0 && Cin != 8 && !auto_split), "tile 8 (128x256): bf16, Cout %% 256 == 0, explicit splits");
  DIM_REQUIRE(Cin != 8 || tile != 4, "tile 4 (128x128, 8 waves) is not built for the 8-channel layer");
  DIM_REQUIRE(Cin != 8 || batch == 1, "batched launch is not built for the 8-channel layer");
  auto launch = [&](const ConvArgs& args, int nsplit, int t0, int tn) -> int {
    if (Cin == 8) {
      if (tile == 1) return launch_conv<128, 128, 2, 2, true>(args, nsplit, st, 1, t0, tn);
      if (tile == 2) return launch_conv<128, 64, 2, 2, true>(args, nsplit, st, 1, t0, tn);
      return launch_conv<64, 64, 2, 2, true>(args, nsplit, st, 1, t0, tn);
    }
    if (tile == 8) return launch_conv_bf16<128, 256, 2, 4>(args, nsplit, st, batch, t0, tn);
    if (tile == 4) return launch_conv<128, 128, 2, 4, false>(args, nsplit, st, batch, t0, tn);
    if (tile == 1) return launch_conv<128, 128, 2, 2, false>(args, nsplit, st, batch, t0, tn);
    if (tile == 2) return launch_conv<128, 64, 2, 2, false>(args, nsplit, st, batch, t0, tn);
    return launch_conv<64, 64, 2, 2, false>(args, nsplit, st, batch, t0, tn);
  };
  // "auto" (splits == 0): whole tiles per CU in one launch, the remainder as a split-K launch + reduce (conv_tail_plan)
  if (auto_split && batch == 1 && workspace && a.dense_out && a.out_cstride == Cout && a.out_coff == 0 && !a.accumulate && !partial_only) {
    int tail_begin = 0, ts = 1;
    conv_tail_plan(a.M, Cout, a.nchunks, tile, &tail_begin, &ts);
    if (ts >= 2) {
      const int BMt = tile == 3 ? 64 : 128, BNt = (tile == 3 || tile == 2) ? 64 : 128;
      const int nt = Cout / BNt, tiles = ceil_div(a.M, BMt) * nt;
      const int row0 = tail_begin / nt * BMt;
      int rc = launch(a, 1, 0, tail_begin);
      if (rc != DIM_OK) return rc;
      ConvArgs t = a;
      t.y = workspace;
      t.slab_row0 = row0;
      t.slab_stride = (long)(a.M - row0) * Cout;
      t.chunks_per_split = (a.nchunks + ts - 1) / ts;
      const int nsplit = (a.nchunks + t.chunks_per_split - 1) / t.chunks_per_split;
      DIM_REQUIRE(nsplit * t.slab_stride <= kTailWorkspaceFloats, "tail workspace bound exceeded");
      rc = launch(t, nsplit, tail_begin, tiles - tail_begin);
      if (rc != DIM_OK) return rc;
      return dim_splitk_reduce(workspace, bias, y + (long)row0 * Cout, (long)(a.M - row0), Cout, nsplit, slope, stream);
    }
  }
  int rc = launch(a, splits, 0, -1);
  if (rc != DIM_OK) return rc;
  if (a.slab_full) return DIM_OK;   // the caller sums the output-shaped slabs once all its launches are in
  if (splits > 1 && !partial_only) return dim_splitk_reduce(workspace, bias, y, (long)a.M, Cout, splits, slope, stream);
  return DIM_OK;
}

int dim_conv2d_fwd(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                   int Cin, int Cout, int KH, int KW, int stride, int pad, float slope, int splits, int tile, void* stream) {
  return conv2d_fwd_impl(x, w_packed, bias, y, workspace, N, H, W, Cin, Cout, KH, KW, stride, pad, slope, splits, tile, 0, stream);
}

// ---- bf16 twins: identical arguments, `w_packed` is the bf16 image (dim_f32_to_bf16) of the f32 packed array
int dim_f32_to_bf16(const float* src, void* dst_bf16, long n, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(src && dst_bf16 && n > 0, "null pointer");
  DIM_REQUIRE(reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst_bf16) % 8 == 0,
              "dim_f32_to_bf16: src must be 16-byte and dst 8-byte aligned (vector accesses)");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ceil_div((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), src,
                     reinterpret_cast<__bf16*>(dst_bf16), n);
  return check_launch("f32_to_bf16");
}

int dim_bf16_to_f32(const void* src_bf16, float* dst, long n, void* stream) {
  if (n == 0) return DIM_OK;
  DIM_REQUIRE(src_bf16 && dst && n > 0, "null pointer");
  DIM_REQUIRE(reinterpret_cast<uintptr_t>(src_bf16) % 8 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0,
              "dim_bf16_to_f32: src must be 8-byte and dst 16-byte aligned (vector accesses)");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ceil_div((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const __bf16*>(src_bf16), dst, n);
  return check_launch("bf16_to_f32");
}

int dim_conv2d_fwd_bf16(const float* x, const void* w_packed_bf16, const float* bias, float* y, float* workspace, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, float slope, int splits, int tile, void* stream) {
  ConvEx ex = {};
  ex.bf16 = 1;
  DIM_REQUIRE(splits >= 1, "the bf16 path takes an explicit split-K count (>= 1)");
  return conv2d_fwd_impl(x, reinterpret_cast<const float*>(w_packed_bf16), bias, y, workspace, N, H, W, Cin, Cout, KH, KW, stride, pad,
                         slope, splits, tile, 0, stream, &ex);
}

int dim_conv2d_fwd_ex_bf16(const float* x, const void* w_packed_bf16, const float* bias, float* y, int N, int H, int W, int Cin,
                           int in_cstride, int Cout, int KH, int KW, int stride, int pad, float slope, int tile, int out_cstride,
                           int out_coff, int OH, int OW, int osy, int osx, int ooy, int oox, int Ho, int Wo, int pad_w, int accumulate,
                           void* stream) {
  ConvEx ex = {in_cstride, out_cstride, out_coff, OH, OW, osy, osx, ooy, oox};
  ex.Ho = Ho;
  ex.Wo = Wo;
  ex.pad_w = pad_w;
  ex.accumulate = accumulate;
  ex.bf16 = 1;
  return conv2d_fwd_impl(x, reinterpret_cast<const float*>(w_packed_bf16), bias, y, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, slope,
                         1, tile, 0, stream, &ex);
}

int dim_conv2d_fwd_ex(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int in_cstride,
                      int Cout, int KH, int KW, int stride, int pad, float slope, int tile, int out_cstride, int out_coff, int OH,
                      int OW, int osy, int osx, int ooy, int oox, int Ho, int Wo, int pad_w, int accumulate, void* stream) {
  ConvEx ex = {in_cstride, out_cstride, out_coff, OH, OW, osy, osx, ooy, oox};
  ex.Ho = Ho;
  ex.Wo = Wo;
  ex.pad_w = pad_w;
  ex.accumulate = accumulate;
  return conv2d_fwd_impl(x, w_packed, bias, y, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, slope, 1, tile, 0, stream, &ex);
}

// ---------------------------------------------------------------------------------------------------------------- dgrad
// Gradient w.r.t. the input of y = conv(x, W (Cout,Cin,KH,KW), stride s in {1,2}, pad p) on the SAME MFMA kernel with re-packed
// weights (channel roles swapped).
//   s = 1:  dX[i] = sum_j dY[i - (K-1-p) + j] * W[K-1-j]                      one stride-1 convolution over dY
//   s = 2:  input rows iy = 2t + ph (phase ph):  dX[2t+ph] = sum_e dY[t + e] * W[ph + p - 2e],  e in [emin, emax]
//           = stride-1 convolution over dY with KH' = emax-emin+1 taps and pad' = -emin, scattered to rows 2t+ph.
// Packed layout: phases (py,px) one after the other, each [chunk][CinPad][32] with chunk = (32-slice of Cout, jy, jx) and
// CinPad = Cin rounded up to 64 (the kernel's channel tile; padded outputs are zero).
struct DgAxis {
  int ntaps, emin;
};
static inline DgAxis dg_axis(int K, int stride, int p, int ph) {
  if (stride == 1) return {K, -(K - 1 - p)};
  int emin = 1000, emax = -1000;
  for (int e = -K; e <= K; ++e) {
    int k = ph + p - 2 * e;
    if (k >= 0 && k < K) { emin = min(emin, e); emax = max(emax, e); }
  }
  if (emin > emax) return {0, 0};
  return {emax - emin + 1, emin};
}

extern "C++" template <typename PT>
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, PT* __restrict__ wp, int Cout, int Cin, int CinPad, int KH, int KW,
                                         int stride, int pad, int py, int px, int nth, int ntw, int eminh, int eminw, int deconv_layout) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)(Cout / 32) * nth * ntw * CinPad * 32;
  if (idx >= total) return;
  int kin = (int)(idx % 32);
  long t = idx / 32;
  int ci = (int)(t % CinPad);
  int kc = (int)(t / CinPad);
  int taps = nth * ntw;
  int cc = kc / taps, tap = kc % taps, jy = tap / ntw, jx = tap % ntw;
  int co = cc * 32 + kin;
  int ky = (stride == 1) ? KH - 1 - jy : py + pad - 2 * (jy + eminh);
  int kx = (stride == 1) ? KW - 1 - jx : px + pad - 2 * (jx + eminw);
  float v = 0.f;
  if (ci < Cin && ky >= 0 && ky < KH && kx >= 0 && kx < KW)
    v = deconv_layout ? w[(((long)ci * Cout + co) * KH + ky) * KW + kx]   // never used (deconv dgrad is a plain forward conv)
                      : w[(((long)co * Cin + ci) * KH + ky) * KW + kx];
  wp[idx] = (PT)v;
}

long dim_conv2d_dgrad_packed_weight_floats(int Cout, int Cin, int KH, int KW, int stride, int pad) {
  int CinPad = (Cin + 63) / 64 * 64;
  long total = 0;
  int nph = stride == 1 ? 1 : 2;
  for (int py = 0; py < nph; ++py)
    for (int px = 0; px < nph; ++px)
      total += (long)(Cout / 32) * dg_axis(KH, stride, pad, py).ntaps * dg_axis(KW, stride, pad, px).ntaps * CinPad * 32;
  return total;
}

// tiled: workgroup = (32-slice cc of Cout, G input channels): rows = w[cc * 32 + r][ci][ky][kx], packed run (phase, jy, jx) at
// phase offset + ((cc * taps + jtap) * CinPad + ci) * 32 -- all phases in one launch through the tap table (<= 32 runs)
extern "C++" template <typename PT>
int dgrad_pack_weight_any(const float* w_oihw, PT* w_packed, int Cout, int Cin, int KH, int KW, int stride, int pad, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null pointer");
  DIM_REQUIRE(stride == 1 || stride == 2, "dgrad supports stride 1 or 2");
  DIM_REQUIRE(Cout % 32 == 0, "Cout must be a multiple of 32 (it is the contraction dimension of dgrad)");
  int CinPad = (Cin + 63) / 64 * 64;
  int nph = stride == 1 ? 1 : 2;
  int runs = 0;
  for (int py = 0; py < nph; ++py)
    for (int px = 0; px < nph; ++px) runs += dg_axis(KH, stride, pad, py).ntaps * dg_axis(KW, stride, pad, px).ntaps;
  const int T = KH * KW, G = runs <= 32 ? wtile_group(CinPad, T, Cout / 32) : 0;
  WTileArgs t = {};
  long off = 0;
  for (int py = 0; py < nph; ++py)
    for (int px = 0; px < nph; ++px) {
      DgAxis ah = dg_axis(KH, stride, pad, py), aw = dg_axis(KW, stride, pad, px);
      long total = (long)(Cout / 32) * ah.ntaps * aw.ntaps * CinPad * 32;
      if (total > 0 && !G) {
        hipLaunchKernelGGL((pack_dgrad_weight_kernel<PT>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw,
                           w_packed + off, Cout, Cin, CinPad, KH, KW, stride, pad, py, px, ah.ntaps, aw.ntaps, ah.emin, aw.emin, 0);
      } else if (total > 0) {
        for (int jy = 0; jy < ah.ntaps; ++jy)
          for (int jx = 0; jx < aw.ntaps; ++jx) {
            const int ky = (stride == 1) ? KH - 1 - jy : py + pad - 2 * (jy + ah.emin);
            const int kx = (stride == 1) ? KW - 1 - jx : px + pad - 2 * (jx + aw.emin);
            t.jq[t.nj] = (ky >= 0 && ky < KH && kx >= 0 && kx < KW) ? ky * KW + kx : -1;
            t.jbase[t.nj] = off + (long)(jy * aw.ntaps + jx) * CinPad * 32;
            t.jx[t.nj] = (long)ah.ntaps * aw.ntaps * CinPad * 32;
            ++t.nj;
          }
      }
      off += total;
    }
  if (G && t.nj > 0) {
    t.src = w_oihw; t.dst = w_packed;
    t.G = G; t.Q = T; t.gmax = Cin; t.rmax = Cout; t.g_fast = 1;
    t.sg = T; t.sr = (long)Cin * T; t.rows_x = 32L * Cin * T; t.rows_y = (long)G * T;
    wtile_launch<true, PT>(t, Cout / 32, CinPad / G, as_stream(stream));
  }
  return check_launch("pack_dgrad_weight");
}

int dim_conv2d_dgrad_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, int stride, int pad,
                                 void* stream) {
  return dgrad_pack_weight_any(w_oihw, w_packed, Cout, Cin, KH, KW, stride, pad, stream);
}

int dim_conv2d_dgrad_pack_weight_bf16(const float* w_oihw, void* w_packed_bf16, int Cout, int Cin, int KH, int KW, int stride, int pad,
                                      void* stream) {
  return dgrad_pack_weight_any(w_oihw, reinterpret_cast<__bf16*>(w_packed_bf16), Cout, Cin, KH, KW, stride, pad, stream);
}

// dx (N,H,W,dx_cstride)[..., :Cin] (+)= dgrad(dy (N,Ho,Wo,dy_cstride)[..., :Cout]).  accumulate != 0 adds to dx (skip connections).
// dst[row][0:width] (+)= sum over the slabs of slab[row][0:width]; rows are `pitch` floats apart in dst and in every slab
__global__ __launch_bounds__(256) void slab_sum_rows_kernel(const float* __restrict__ slabs, long slab_stride, int nslabs,
                                                            float* __restrict__ dst, long rows, int width4, int pitch, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * width4) return;
  const long row = i / width4;
  const int c = (int)(i - row * width4) * 4;
  const long o = row * pitch + c;
  float4 s = accumulate ? *reinterpret_cast<const float4*>(dst + o) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < nslabs; ++k) {   // fixed order: deterministic
    const float4 v = *reinterpret_cast<const float4*>(slabs + (long)k * slab_stride + o);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(dst + o) = s;
}

// rows of the bias-gradient partial sums one folded input gradient writes: 2 per 16 x 16 block of every phase image
static long dgrad_fold_rows(int N, int H, int W, int stride) {
  long rows = 0;
  const int nph = stride == 1 ? 1 : 2;
  for (int py = 0; py < nph; ++py)
    for (int px = 0; px < nph; ++px) {
      const int Th = stride == 1 ? H : (H - py + 1) / 2, Tw = stride == 1 ? W : (W - px + 1) / 2;
      if (Th > 0 && Tw > 0) rows += 2L * N * ((Th + 15) / 16) * ((Tw + 15) / 16);
    }
  return rows;
}

static int conv2d_dgrad_impl(const float* dy, const float* w_dgrad_packed, float* dx, int N, int H, int W, int Cin, int dx_cstride, int Ho,
                             int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad, int accumulate, int tile, int bf16,
                             void* stream, float* workspace = nullptr, int splits = 1, const float* mask = nullptr, float mask_slope = 1.f,
                             float* colsum = nullptr) {
  if (N == 0) return DIM_OK;
  long fold_row = 0;
  // split-K (bf16, gathered-tap tiles): the small maps give 80-600 workgroups to 1280 resident slots and every workgroup walks
  // K = 4608 .. 9216 alone on its CU; `splits` K ranges write output-shaped slabs (every phase of a strided gradient into the same
  // slabs, each to its own pixels) and ONE pass sums them into dx
  const bool split = splits > 1;
  DIM_REQUIRE(!split || (bf16 && workspace && (tile == 3 || tile == 4)), "split-K input gradient: bf16, tile 3 or 4, workspace required");
  DIM_REQUIRE(stride == 1 || stride == 2, "dgrad supports stride 1 or 2");
  int CinPad = (Cin + 63) / 64 * 64;
  DIM_REQUIRE(dx_cstride >= CinPad, "dx channel stride (%d) must be >= Cin rounded up to 64 (%d)", dx_cstride, CinPad);
  int nph = stride == 1 ? 1 : 2;
  long off = 0;
  for (int py = 0; py < nph; ++py)
    for (int px = 0; px < nph; ++px) {
      DgAxis ah = dg_axis(KH, stride, pad, py), aw = dg_axis(KW, stride, pad, px);
      long total = (long)(Cout / 32) * ah.ntaps * aw.ntaps * CinPad * 32;
      int Th = stride == 1 ? H : (H - py + 1) / 2, Tw = stride == 1 ? W : (W - px + 1) / 2;
      if (Th > 0 && Tw > 0) {
        ConvEx ex = {dy_cstride, dx_cstride, 0, H, W, stride == 1 ? 0 : 2, stride == 1 ? 0 : 2, py, px};
        ex.pad_w = -aw.emin;
        ex.Ho = Th;
        ex.Wo = Tw;
        ex.accumulate = split ? 0 : accumulate;
        ex.bf16 = bf16;
        ex.slab_full = split ? 1 : 0;
        ex.mask = mask;
        ex.mask_slope = mask_slope;
        ex.colsum = colsum;
        ex.colsum_row0 = (int)fold_row;
        fold_row += 2L * N * ((Th + 15) / 16) * ((Tw + 15) / 16);
        if (total > 0) {
          // tile 9 (bf16 patch kernel) takes the phases with 2 .. 9 taps; a single-tap phase is a 1x1 convolution: gathered-tap kernel
          int ptile = tile;
          if (tile == 9 && !(bf16 && ah.ntaps <= 3 && aw.ntaps <= 3 && ah.ntaps * aw.ntaps >= 2))
            ptile = CinPad % 128 == 0 ? 4 : 3;
          DIM_REQUIRE(!mask || ptile == 9, "folded input gradient: phase (%d,%d) has %d x %d taps, which the patch kernel does not take", py, px,
                      ah.ntaps, aw.ntaps);
          // every slab of every phase must be written: the phase's K chunks have to make exactly `splits` non-empty ranges
          const int nch = ah.ntaps * aw.ntaps * (Cout / 32);
          const int psplits = split ? splits : 1;
          DIM_REQUIRE(!split || (nch >= splits && (nch + (nch + splits - 1) / splits - 1) / ((nch + splits - 1) / splits) == splits),
                      "split-K input gradient: a phase has %d K chunks, which do not make %d non-empty splits", nch, splits);
          int rc = conv2d_fwd_impl(dy, bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(w_dgrad_packed) + 2 * off)
                                            : w_dgrad_packed + off, nullptr, dx, split ? workspace : nullptr, N, Ho, Wo, Cout, CinPad,
                                   ah.ntaps, aw.ntaps, 1, -ah.emin, 1.0f, psplits, ptile, 0, stream, &ex);
          if (rc != DIM_OK) return rc;
        } else if (!accumulate || mask || split) {
          // (split: the slab rows of a tap-less phase would never be written, and slab_sum_rows_kernel adds every row of every slab)
          return set_err(DIM_ERR_ARG, "phase (%d,%d) has no taps: dX rows of that phase would stay unwritten", py, px);
        }
      }
      off += total;
    }
  if (split) {
    const long rows = (long)N * H * W;
    hipLaunchKernelGGL(slab_sum_rows_kernel, dim3(ceil_div(rows * (CinPad / 4), 256)), dim3(256), 0, as_stream(stream), workspace,
                       rows * dx_cstride, splits, dx, rows, CinPad / 4, dx_cstride, accumulate);
    return check_launch("slab_sum_rows");
  }
  return DIM_OK;
}

int dim_conv2d_dgrad(const float* dy, const float* w_dgrad_packed, float* dx, int N, int H, int W, int Cin, int dx_cstride, int Ho,
                     int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad, int accumulate, int tile, void* stream) {
  return conv2d_dgrad_impl(dy, w_dgrad_packed, dx, N, H, W, Cin, dx_cstride, Ho, Wo, Cout, dy_cstride, KH, KW, stride, pad, accumulate, tile,
                           0, stream);
}

int dim_conv2d_dgrad_bf16(const float* dy, const void* w_dgrad_packed_bf16, float* dx, int N, int H, int W, int Cin, int dx_cstride,
                          int Ho, int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad, int accumulate, int tile,
                          void* stream) {
  return conv2d_dgrad_impl(dy, reinterpret_cast<const float*>(w_dgrad_packed_bf16), dx, N, H, W, Cin, dx_cstride, Ho, Wo, Cout, dy_cstride,
                           KH, KW, stride, pad, accumulate, tile, 1, stream);
}

long dim_conv2d_dgrad_splitk_workspace_floats(int N, int H, int W, int dx_cstride, int splits) {
  return splits > 1 ? (long)splits * N * H * W * dx_cstride : 0;
}

int dim_conv2d_dgrad_bf16_splitk(const float* dy, const void* w_dgrad_packed_bf16, float* dx, float* workspace, int N, int H, int W, int Cin,
                                 int dx_cstride, int Ho, int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad,
                                 int accumulate, int tile, int splits, void* stream) {
  DIM_REQUIRE(splits >= 1, "splits >= 1");
  DIM_REQUIRE(dx_cstride % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "split-K input gradient: dx / workspace 16-byte aligned, channel stride a multiple of 4");
  return conv2d_dgrad_impl(dy, reinterpret_cast<const float*>(w_dgrad_packed_bf16), dx, N, H, W, Cin, dx_cstride, Ho, Wo, Cout, dy_cstride,
                           KH, KW, stride, pad, accumulate, tile, 1, stream, workspace, splits);
}

long dim_conv2d_dgrad_lrelu_workspace_floats(int N, int H, int W, int Cin, int stride) {
  const long rows = dgrad_fold_rows(N, H, W, stride);
  const int CinPad = (Cin + 63) / 64 * 64;
  return rows * CinPad + dim_bias_grad_workspace_floats(rows, CinPad);
}

// dz = dX * LeakyReLU'(y_act) and db = column sums of dz, both inside the input gradient's epilogue (tile 9) + one small reduce
int dim_conv2d_dgrad_bf16_lrelu(const float* dy, const void* w_dgrad_packed_bf16, float* dz, const float* y_act, float slope, float* db,
                                float* workspace, int N, int H, int W, int Cin, int dx_cstride, int Ho, int Wo, int Cout, int dy_cstride,
                                int KH, int KW, int stride, int pad, int accumulate_db, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(y_act && db && workspace, "null pointer");
  const int CinPad = (Cin + 63) / 64 * 64;
  DIM_REQUIRE(Cin == CinPad && dx_cstride == Cin, "folded input gradient: dz and the stored activation must both be dense (N, H, W, Cin %% 64 == 0)");
  const long rows = dgrad_fold_rows(N, H, W, stride);
  int rc = conv2d_dgrad_impl(dy, reinterpret_cast<const float*>(w_dgrad_packed_bf16), dz, N, H, W, Cin, dx_cstride, Ho, Wo, Cout, dy_cstride, KH,
                             KW, stride, pad, 0, 9, 1, stream, nullptr, 1, y_act, slope, workspace);
  if (rc != DIM_OK) return rc;
  return dim_bias_grad(workspace, db, workspace + rows * CinPad, rows, CinPad, CinPad, 0, accumulate_db, stream);
}

int dim_conv2d_fwd_partial(const float* x, const float* w_packed, float* workspace, int N, int H, int W, int Cin, int Cout, int KH,
                           int KW, int stride, int pad, int splits, int tile, void* stream) {
  DIM_REQUIRE(splits > 1, "dim_conv2d_fwd_partial is the split-K first phase: splits must be > 1");
  return conv2d_fwd_impl(x, w_packed, nullptr, workspace, workspace, N, H, W, Cin, Cout, KH, KW, stride, pad, 1.0f, splits, tile, 1,
                         stream);
}

long dim_deconv4x4s2_packed_weight_floats(int Cin, int Cout) {
  int CinPad = (Cin + 31) / 32 * 32;
  return 4L * CinPad * 4 * Cout;
}

// tiled: workgroup = (32-slice cc of Cin, G output channels): rows = w[cc * 32 + r][co][ky][kx], packed run (phase, dy, dx) at
// phase * per_phase + ((cc * 4 + tap) * Cout + co) * 32
extern "C++" template <typename PT>
int deconv_pack_weight_any(const float* w_iohw, PT* w_packed, int Cin, int Cout, void* stream) {
  DIM_REQUIRE(w_iohw && w_packed, "null pointer");
  int CinPad = (Cin + 31) / 32 * 32;
  long total = 4L * CinPad * 4 * Cout;
  const int G = wtile_group(Cout, 16, CinPad / 32);
  if (G) {
    WTileArgs t = {};
    const long per_phase = (long)CinPad * 4 * Cout;
    for (int phase = 0; phase < 4; ++phase)
      for (int tap = 0; tap < 4; ++tap) {
        const int py = phase / 2, px = phase % 2, dy = tap / 2, dx = tap % 2;
        t.jq[t.nj] = (py + 2 * (1 - dy)) * 4 + (px + 2 * (1 - dx));
        t.jbase[t.nj] = phase * per_phase + (long)tap * Cout * 32;
        t.jx[t.nj] = 4L * Cout * 32;
        ++t.nj;
      }
    t.src = w_iohw; t.dst = w_packed;
    t.G = G; t.Q = 16; t.gmax = Cout; t.rmax = Cin; t.g_fast = 1;
    t.sg = 16; t.sr = (long)Cout * 16; t.rows_x = 32L * Cout * 16; t.rows_y = (long)G * 16;
    wtile_launch<true, PT>(t, CinPad / 32, Cout / G, as_stream(stream));
  } else {
    hipLaunchKernelGGL((pack_deconv4x4s2_weight_kernel<PT>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_iohw, w_packed,
                       Cin, CinPad, Cout);
  }
  return check_launch("pack_deconv_weight");
}

int dim_deconv4x4s2_pack_weight(const float* w_iohw, float* w_packed, int Cin, int Cout, void* stream) {
  return deconv_pack_weight_any(w_iohw, w_packed, Cin, Cout, stream);
}

int dim_deconv4x4s2_pack_weight_bf16(const float* w_iohw, void* w_packed_bf16, int Cin, int Cout, void* stream) {
  return deconv_pack_weight_any(w_iohw, reinterpret_cast<__bf16*>(w_packed_bf16), Cin, Cout, stream);
}

// y[:, oy, ox, out_coff : out_coff+Cout] = LeakyReLU(Crop(Deconvolution(x, k=4, s=2, p=0) + bias, offset=(crop,crop)))   (NHWC)
// x (N,H,W,in_cstride) with Cin valid channels, zero weights for the padding up to a multiple of 32.
static int deconv4x4s2_fwd_impl(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                                int in_cstride, int Cout, int OH, int OW, int crop, float slope, int out_cstride, int out_coff, int tile,
                                int bf16, void* stream) {
  if (N == 0) return DIM_OK;
  int CinPad = (Cin + 31) / 32 * 32;
  DIM_REQUIRE(in_cstride >= CinPad, "in_cstride (%d) must cover the padded channel count %d (pad channels must hold zeros)", in_cstride,
              CinPad);
  DIM_REQUIRE(OH + crop <= 2 * H + 2 && OW + crop <= 2 * W + 2, "crop window outside the deconvolution output");
  // the four output phases read the same input windows and differ in weights and in where they land: ONE batched launch
  // (blockIdx.y = phase), 4x the workgroups of a per-phase launch on maps as small as 15x20
  const long per_phase = (long)CinPad * 4 * Cout;
  ConvEx ex = {in_cstride, out_cstride, out_coff, OH, OW, 2, 2, -crop, -crop};
  ex.batch = 4;
  ex.bw = per_phase;
  ex.boy = 1;
  ex.box = 1;
  ex.bf16 = bf16;
  return conv2d_fwd_impl(x, w_packed, bias, y, nullptr, N, H, W, CinPad, Cout, 2, 2, 1, 1, slope, 1, tile, 0, stream, &ex);
}

int dim_deconv4x4s2_fwd(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                        int in_cstride, int Cout, int OH, int OW, int crop, float slope, int out_cstride, int out_coff, int tile,
                        void* stream) {
  return deconv4x4s2_fwd_impl(x, w_packed, bias, y, N, H, W, Cin, in_cstride, Cout, OH, OW, crop, slope, out_cstride, out_coff, tile, 0,
                              stream);
}

int dim_deconv4x4s2_fwd_bf16(const float* x, const void* w_packed_bf16, const float* bias, float* y, int N, int H, int W, int Cin,
                             int in_cstride, int Cout, int OH, int OW, int crop, float slope, int out_cstride, int out_coff, int tile,
                             void* stream) {
  return deconv4x4s2_fwd_impl(x, reinterpret_cast<const float*>(w_packed_bf16), bias, y, N, H, W, Cin, in_cstride, Cout, OH, OW, crop,
                              slope, out_cstride, out_coff, tile, 1, stream);
}

int dim_conv_small_cout_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null pointer");
  int CinPad = (Cin + 31) / 32 * 32;
  long total = (long)Cout * KH * KW * CinPad;
  hipLaunchKernelGGL(pack_small_cout_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed,
                     Cout, Cin, CinPad, KH, KW);
  return check_launch("pack_small_cout_weight");
}

int dim_conv_small_cout_fwd(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                            int in_cstride, int Cout, int KH, int KW, int pad, int out_cstride, int out_coff, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && w_packed && y, "null pointer");
  DIM_REQUIRE(Cout == 1 || Cout == 2, "small-Cout kernel handles Cout 1 or 2 (got %d)", Cout);
  int CinPad = (Cin + 31) / 32 * 32;
  DIM_REQUIRE(in_cstride >= CinPad && in_cstride % 4 == 0, "in_cstride must cover the padded channel count");
  DIM_REQUIRE(KW == 3 || KW == 1, "small-Cout kernel is built for 3- and 1-wide kernels (got KW = %d)", KW);
  constexpr int PX = 4;   // eight pixels per wave halve the wave count of these small maps: 42 / 31 / 22 / 18 us against 34 / 28 / 17 / 11
  const long groups = (long)N * H * ((W + PX - 1) / PX);
  dim3 grid(ceil_div(groups, 4)), block(256);
#define DIM_SMALL_COUT(CO, KWc)                                                                                                           \
  hipLaunchKernelGGL((conv_small_cout_kernel<CO, KWc, PX>), grid, block, 0, as_stream(stream), x, w_packed, bias, y, N, H, W, CinPad,     \
                     in_cstride, KH, pad, out_cstride, out_coff)
  if (Cout == 1) { if (KW == 3) DIM_SMALL_COUT(1, 3); else DIM_SMALL_COUT(1, 1); }
  else { if (KW == 3) DIM_SMALL_COUT(2, 3); else DIM_SMALL_COUT(2, 1); }
#undef DIM_SMALL_COUT
  return check_launch("conv_small_cout");
}

int dim_deconv4x4s2_tiny_fwd(const float* x, const float* w_iohw, const float* bias, float* y, int N, int H, int W, int Cin,
                             int in_cstride, int Cout, int OH, int OW, int crop, int out_cstride, int out_coff, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && w_iohw && y, "null pointer");
  long total = (long)N * OH * OW * Cout;
  hipLaunchKernelGGL(deconv4x4s2_tiny_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), x, w_iohw, bias, y, N, H,
                     W, Cin, in_cstride, Cout, OH, OW, crop, out_cstride, out_coff);
  return check_launch("deconv_tiny");
}

int dim_upsample16_fwd(const float* x_nhwc, const float* w_c1_32_32, float* y_nchw, int N, int C, int h, int w, int OH, int OW,
                       int crop, float scale, int mode, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x_nhwc && w_c1_32_32 && y_nchw, "null pointer");
  DIM_REQUIRE(mode == 0 || mode == 1, "mode 0 (linear) or 1 (sigmoid)");
  DIM_REQUIRE(OH + crop <= 16 * h + 16 && OW + crop <= 16 * w + 16, "crop window outside the deconvolution output");
  if (crop % 4 == 0 && OW % 4 == 0 && (reinterpret_cast<uintptr_t>(y_nchw) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_c1_32_32) & 15) == 0)
    hipLaunchKernelGGL(upsample16_x4_kernel, dim3(ceil_div(OW / 4, 256), OH, N * C), dim3(256), 0, as_stream(stream), x_nhwc, w_c1_32_32,
                       y_nchw, C, h, w, OH, OW, crop, scale, mode);
  else
    hipLaunchKernelGGL(upsample16_kernel, dim3(ceil_div(OW, 256), OH, N * C), dim3(256), 0, as_stream(stream), x_nhwc, w_c1_32_32,
                       y_nchw, C, h, w, OH, OW, crop, scale, mode);
  return check_launch("upsample16");
}

int dim_pose_head_fwd(const float* fc6, const float* fc7_w, const float* fc7_b, const float* rot_w, const float* rot_b,
                      const float* trans_w, const float* trans_b, const float* zoom_factor, float* se3, float* fc7_out, int B,
                      void* stream) {
  DIM_REQUIRE(fc6 && fc7_w && fc7_b && rot_w && rot_b && trans_w && trans_b && zoom_factor && se3, "null pointer");
  if (B == 0) return DIM_OK;
  hipLaunchKernelGGL(pose_head_kernel, dim3(B), dim3(1024), 0, as_stream(stream), fc6, fc7_w, fc7_b, rot_w, rot_b, trans_w,
                     trans_b, zoom_factor, se3, fc7_out);
  return check_launch("pose_head");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ Winograd F(2x2, 3x3)
// 3x3 / stride 1 / pad 1 layers (conv3_1, conv4_1, conv5_1, conv6_1 of deepIM_flownet.py:103-191) as
//   V = B^T d B  (input tiles 4x4, stride 2)  ->  16 independent GEMMs  M_k = V_k (T x Cin) * U_k (Cin x Cout)  ->  Y = A^T M A
// 2.25x fewer multiply-adds than the direct form; the GEMMs run as ONE persistent stream-K launch (wino_gemm.hip).  Transforms are exact in the sense of using only +,- on the data (B, A have entries
// 0, +-1); the weight transform G (entries 1, 1/2) is applied once at pack time.  f32 throughout.
namespace dim {

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// V[t][k][c]: thread = (tile t, channel quad)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W, int C,
                                                         int in_cstride, int th, int tw, FastDiv div_cq, FastDiv div_tw, FastDiv div_th) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const unsigned CQ = C >> 2;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const int y0 = 2 * (int)ty - 1, x0 = 2 * (int)tx - 1;
  const float* base = x + (long)n * H * W * in_cstride + cq * 4;
  float4 d[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int yy = y0 + a, xx = x0 + b;
      d[a][b] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    ? *reinterpret_cast<const float4*>(base + ((long)yy * W + xx) * in_cstride)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float4 tmp[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {  // B^T d
    tmp[0][b] = f4sub(d[0][b], d[2][b]);
    tmp[1][b] = f4add(d[1][b], d[2][b]);
    tmp[2][b] = f4sub(d[2][b], d[1][b]);
    tmp[3][b] = f4sub(d[1][b], d[3][b]);
  }
  const long plane = C;  // V [t][k][c]: the 16 planes of a tile are consecutive K columns of its row (wino_gemm.hip)
  float* out = V + (long)t * 16 * C + cq * 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {  // (.) B
    *reinterpret_cast<float4*>(out + (a * 4 + 0) * plane) = f4sub(tmp[a][0], tmp[a][2]);
    *reinterpret_cast<float4*>(out + (a * 4 + 1) * plane) = f4add(tmp[a][1], tmp[a][2]);
    *reinterpret_cast<float4*>(out + (a * 4 + 2) * plane) = f4sub(tmp[a][2], tmp[a][1]);
    *reinterpret_cast<float4*>(out + (a * 4 + 3) * plane) = f4sub(tmp[a][1], tmp[a][3]);
  }
}

// Y = A^T M A + bias, LeakyReLU; thread = (tile t, output-channel quad); writes the 2x2 outputs that fall inside H x W
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y,
                                                          int N, int H, int W, int C, int out_cstride, int out_coff, int th, int tw,
                                                          float slope, FastDiv div_cq, FastDiv div_tw, FastDiv div_th) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const unsigned CQ = C >> 2;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const long plane = C;  // M [t][k][c]
  const float* in = M + (long)t * 16 * C + cq * 4;
  float4 m[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const float4*>(in + (a * 4 + b) * plane);
  float4 r0[4], r1[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {  // A^T m
    r0[b] = f4add(f4add(m[0][b], m[1][b]), m[2][b]);
    r1[b] = f4sub(f4sub(m[1][b], m[2][b]), m[3][b]);
  }
  float4 o[2][2];
  o[0][0] = f4add(f4add(r0[0], r0[1]), r0[2]);
  o[0][1] = f4sub(f4sub(r0[1], r0[2]), r0[3]);
  o[1][0] = f4add(f4add(r1[0], r1[1]), r1[2]);
  o[1][1] = f4sub(f4sub(r1[1], r1[2]), r1[3]);
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oy = 2 * (int)ty + a, ox = 2 * (int)tx + b;
      if (oy < H && ox < W) {
        float4 v = f4add(o[a][b], bv);
        v.x = v.x > 0.f ? v.x : v.x * slope;
        v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope;
        v.w = v.w > 0.f ? v.w : v.w * slope;
        *reinterpret_cast<float4*>(y + (((long)n * H + oy) * W + ox) * out_cstride + out_coff + cq * 4) = v;
      }
    }
}

// U_k = G g G^T per (co, ci), written in the 1x1 packed layout of each of the 16 GEMMs: [k][ci/32][co][ci%32]
// DGRAD: the kernel of the input gradient, g'[ci -> co][kh][kw] = w[ci][co][2 - kh][2 - kw] read from the forward's (O, I, 3, 3) array
// (Cout / Cin are the GEMM's: dX channels / dY channels), instead of a flipped + transposed copy made by the caller
template <bool DGRAD>
__global__ void wino_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cout * Cin) return;
  const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
  float g[9];
  {
    const float* gp = w + (DGRAD ? (long)ci * Cout + co : (long)co * Cin + ci) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] = gp[DGRAD ? 8 - i : i];
  }
  float Gg[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Gg[0][j] = g[j];
    Gg[1][j] = 0.5f * (g[j] + g[3 + j] + g[6 + j]);
    Gg[2][j] = 0.5f * (g[j] - g[3 + j] + g[6 + j]);
    Gg[3][j] = g[6 + j];
  }
  const long per_k = (long)Cin * Cout;
  float* o = wp + ((long)(ci >> 5) * Cout + co) * 32 + (ci & 31);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[(i * 4 + 0) * per_k] = Gg[i][0];
    o[(i * 4 + 1) * per_k] = 0.5f * (Gg[i][0] + Gg[i][1] + Gg[i][2]);
    o[(i * 4 + 2) * per_k] = 0.5f * (Gg[i][0] - Gg[i][1] + Gg[i][2]);
    o[(i * 4 + 3) * per_k] = Gg[i][2];
  }
}

// ------------------------------------------------------------------------------------------------ Winograd F(4x4, 3x3)
// Same scheme with 6x6 input tiles at stride 4 and 36 GEMMs: 4x fewer multiply-adds than the direct form (F(2x2): 2.25x) and
// 2.25 T-tile planes per input pixel instead of 4, i.e. less transform traffic as well.  Cook-Toom points {0, 1, -1, 2, -1/2, inf}:
// mixing a large and a small point keeps the f32 error at ~2e-6 rms / 2e-5 max of the output scale (the usual {0,+-1,+-2}: 4e-5 max).
//   B^T = [1 3/2 -2 -3/2 1 0; 0 -1 -5/2 -1/2 1 0; 0 1 1/2 -5/2 1 0; 0 -1/2 -1 1/2 1 0; 0 2 -1 -2 1 0; 0 1 3/2 -2 -3/2 1]
//   G   = [1 0 0; -1/3 -1/3 -1/3; 1/3 -1/3 1/3; 1/15 2/15 4/15; -16/15 8/15 -4/15; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -1/2 0; 0 1 1 4 1/4 0; 0 1 -1 8 -1/8 1]
constexpr int kWino4Vec = 2;  // channels per thread of the F(4x4) transforms

template <int VEC>
struct WinoVec {
  typedef float type __attribute__((ext_vector_type(VEC)));
};

#define DIM_WINO4_BT(O, D, S)                                                          \
  {                                                                                    \
    O[0 * S] = D[0] + 1.5f * D[1] - 2.f * D[2] - 1.5f * D[3] + D[4];                    \
    O[1 * S] = D[4] - D[1] - 2.5f * D[2] - 0.5f * D[3];                                 \
    O[2 * S] = D[4] + D[1] + 0.5f * D[2] - 2.5f * D[3];                                 \
    O[3 * S] = D[4] - 0.5f * D[1] - D[2] + 0.5f * D[3];                                 \
    O[4 * S] = D[4] + 2.f * D[1] - D[2] - 2.f * D[3];                                   \
    O[5 * S] = D[1] + 1.5f * D[2] - 2.f * D[3] - 1.5f * D[4] + D[5];                    \
  }

// V[t][k][c], k = 6a + b: thread = (tile t, VEC channels).
// S = 1: a 3x3 / stride-1 / pad-1 layer.  S = 2: a 5x5 / stride-2 / pad-2 layer as the sum of four 3x3 / stride-1 / pad-1
// convolutions of its phase images X^(py,px)[r][q] = x[2r + py][2q + px] (sub-kernels g[u][v] = w[2u + py][2v + px], zero beyond
// the 5 taps): the four transformed phase tiles are concatenated along the channels, V has 4C of them (phase-major), so that ONE
// GEMM per Winograd plane contracts over phases and channels and the output transform is that of the stride-1 layer.
template <int VEC, int S>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W, int C,
                                                          int in_cstride, int th, int tw, FastDiv div_cq, FastDiv div_tw, FastDiv div_th,
                                                          unsigned nblk, WGemmArgs plan) {
#pragma clang fp contract(fast)
  typedef typename WinoVec<VEC>::type vf;
  if (blockIdx.x >= nblk) {  // spare blocks: zero the M tiles that two workgroups of the following stream-K GEMM share
    wino_gemm_zero_tile(plan, (int)(blockIdx.x - nblk) + 1);
    return;
  }
  // XCD-contiguous numbering: neighbouring tiles share two of their six input rows / columns, and block ids go round-robin to the 8
  // XCDs -- in launch order every overlap is fetched into a second L2 (PMC: 1.4x the algorithmic bytes on the fabric)
  const unsigned idx = (unsigned)wg_xcd_contiguous((int)blockIdx.x, (int)nblk) * 256u + threadIdx.x;
  const unsigned CT = C * S * S;  // channels of V
  const unsigned CQ = CT / VEC;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const int y0 = 4 * (int)ty - 1, x0 = 4 * (int)tx - 1;  // tile origin in the (phase) image
  const unsigned cc = cq * VEC;
  const unsigned ph = S == 1 ? 0u : (unsigned)(cc >= (unsigned)C) + (unsigned)(cc >= 2u * C) + (unsigned)(cc >= 3u * C);
  const int py = ph >> 1, px = ph & 1;
  const float* base = x + (long)n * H * W * in_cstride + (cc - ph * C);
  vf tmp[36];
#pragma unroll
  for (int b = 0; b < 6; ++b) {  // B^T d, one tile column at a time
    const int xx = S * (x0 + b) + px;
    const bool okx = (unsigned)xx < (unsigned)W;
    const int xc = okx ? xx : 0;
    vf d[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      // load from a clamped address, then select: a conditional load would compile to a branch with a wait per load
      const int yy = S * (y0 + a) + py;
      const bool ok = okx && (unsigned)yy < (unsigned)H;
      const int yc = (unsigned)yy < (unsigned)H ? yy : 0;
      vf v = *reinterpret_cast<const vf*>(base + ((long)yc * W + xc) * in_cstride);
      d[a] = ok ? v : (vf)(0.f);
    }
    vf* o = tmp + b;
    DIM_WINO4_BT(o, d, 6)
  }
  const long plane = CT;  // V [t][k][c]
  float* out = V + (long)t * 36 * CT + cc;
#pragma unroll
  for (int a = 0; a < 6; ++a) {  // (.) B
    vf o[6];
    const vf* d = tmp + 6 * a;
    DIM_WINO4_BT(o, d, 1)
#pragma unroll
    for (int b = 0; b < 6; ++b) __builtin_nontemporal_store(o[b], reinterpret_cast<vf*>(out + (a * 6 + b) * plane));
  }
}

#define DIM_WINO4_AT(O, M, S)                                       \
  {                                                                 \
    const vf s1 = M[1] + M[2], d1 = M[1] - M[2];                    \
    O[0 * S] = M[0] + s1 + M[3] + M[4];                             \
    O[1 * S] = d1 + 2.f * M[3] - 0.5f * M[4];                       \
    O[2 * S] = s1 + 4.f * M[3] + 0.25f * M[4];                      \
    O[3 * S] = d1 + 8.f * M[3] - 0.125f * M[4] + M[5];              \
  }

// Y = A^T M A + bias, LeakyReLU; thread = (tile t, VEC output channels); writes the 4x4 outputs that fall inside H x W.
// S = 2 (input gradient of a 5x5 / stride-2 layer): M carries 4 C channels, phase-major; the 4x4 block of phase (py,px) is one
// of the four stride-2 phase images of the H x W output: pixel (2 (4 ty + a) + py, 2 (4 tx + b) + px), channel c.
template <int VEC, int S>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y,
                                                           int N, int H, int W, int C, int out_cstride, int out_coff, int th, int tw,
                                                           float slope, FastDiv div_cq, FastDiv div_tw, FastDiv div_th) {
#pragma clang fp contract(fast)
  typedef typename WinoVec<VEC>::type vf;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const unsigned CT = C * S * S;  // channels of M
  const unsigned CQ = CT / VEC;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const long plane = CT;  // M [t][k][c]
  const unsigned cc = cq * VEC;
  const unsigned ph = S == 1 ? 0u : (unsigned)(cc >= (unsigned)C) + (unsigned)(cc >= 2u * C) + (unsigned)(cc >= 3u * C);
  const int py = ph >> 1, px = ph & 1;
  const unsigned co = cc - ph * C;  // output channel
  const float* in = M + (long)t * 36 * CT + cc;
  vf rr[24];  // A^T m: rr[4 rows][6 columns]
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    vf m[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) m[a] = __builtin_nontemporal_load(reinterpret_cast<const vf*>(in + (a * 6 + b) * plane));
    vf* o = rr + b;
    DIM_WINO4_AT(o, m, 6)
  }
  vf bv = (vf)(0.f);
  if (bias) bv = *reinterpret_cast<const vf*>(bias + co);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    vf o[4];
    const vf* m = rr + 6 * a;
    DIM_WINO4_AT(o, m, 1)
    const int oy = S * (4 * (int)ty + a) + py;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ox = S * (4 * (int)tx + b) + px;
      if (oy < H && ox < W) {
        vf v = o[b] + bv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        *reinterpret_cast<vf*>(y + (((long)n * H + oy) * W + ox) * out_cstride + out_coff + co) = v;
      }
    }
  }
}

// U_k = G g G^T for one 3x3 kernel g, scattered with stride per_k over the 36 planes; f64 inside (runs once per weight update)
__device__ __forceinline__ void wino4_transform_weight(const float g[9], float* __restrict__ o, long per_k) {
  const double G[6][3] = {{1., 0., 0.},
                          {-1. / 3, -1. / 3, -1. / 3},
                          {1. / 3, -1. / 3, 1. / 3},
                          {1. / 15, 2. / 15, 4. / 15},
                          {-16. / 15, 8. / 15, -4. / 15},
                          {0., 0., 1.}};
  double Gg[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Gg[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) o[(i * 6 + j) * per_k] = (float)(Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2]);
}

// (Cout,Cin,3,3) -> the 1x1 packed layout of each of the 36 GEMMs: [k][ci/32][co][ci%32]
template <bool DGRAD>
__global__ void wino4_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cout * Cin) return;
  const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
  const float* gp = w + (DGRAD ? (long)ci * Cout + co : (long)co * Cin + ci) * 9;
  float g[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) g[i] = gp[DGRAD ? 8 - i : i];
  wino4_transform_weight(g, wp + ((long)(ci >> 5) * Cout + co) * 32 + (ci & 31), (long)Cin * Cout);
}

// (Cout,Cin,5,5) of a stride-2 layer -> 36 GEMMs over K = 4 Cin (phase-major: kk = (2 py + px) Cin + ci), sub-kernel of phase
// (py,px): g[u][v] = w[2u + py][2v + px], zero where 2u + py or 2v + px > 4
__global__ void wino4_pack_weight_5x5s2_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cout * Cin * 4) return;
  const int kk = (int)(idx % (4 * Cin)), co = (int)(idx / (4 * Cin));
  const int ph = kk / Cin, ci = kk - ph * Cin;
  const int py = ph >> 1, px = ph & 1;
  const float* gp = w + ((long)co * Cin + ci) * 25;
  float g[9];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int i = 2 * u + py, j = 2 * v + px;
      g[u * 3 + v] = (i < 5 && j < 5) ? gp[i * 5 + j] : 0.f;
    }
  wino4_transform_weight(g, wp + ((long)(kk >> 5) * Cout + co) * 32 + (kk & 31), 4L * Cin * Cout);
}

// Input gradient of the 5x5 / stride-2 layer: dX^(py,px)[r][q] = sum_{u,v} g_ph[2-u][2-v] dY[r+u-1][q+v-1]  (the forward's sub-kernels,
// flipped), contracted over the OUTPUT channels: one F(4x4,3x3) transform of dY, 36 GEMMs with K = Cout and N = 4 Cin (phase-major
// n = (2 py + px) Cin + ci), phase-scattering output transform.  Packed [k][co/32][4 Cin][co%32].
__global__ void wino4_pack_weight_5x5s2_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cout * Cin * 4) return;
  const int co = (int)(idx % Cout), nn = (int)(idx / Cout);
  const int ph = nn / Cin, ci = nn - ph * Cin;
  const int py = ph >> 1, px = ph & 1;
  const float* gp = w + ((long)co * Cin + ci) * 25;
  float g[9];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int i = 2 * (2 - u) + py, j = 2 * (2 - v) + px;
      g[u * 3 + v] = (i < 5 && j < 5) ? gp[i * 5 + j] : 0.f;
    }
  wino4_transform_weight(g, wp + ((long)(co >> 5) * (4 * Cin) + nn) * 32 + (co & 31), 4L * Cin * Cout);
}

// ------------------------------------------------------------------------------------------------ Winograd weight gradient
// dW[u][v] = sum_{n,y,x} X[y+u-1][x+v-1] dY[y][x] per 4x4 tile of dY is the correlation F(3x3, 4x4): input = the 6x6 tile of X (the
// forward's V = B^T d B, same tiles, same kernel), "filter" = the 4x4 tile of dY (D = G4 g G4^T), three outputs per axis:
//   dW = A3^T [ sum_tiles V (.) D ] A3      -- the sum over tiles and batch is a GEMM per plane (contraction over T), 36 instead of 144
// multiply-adds per tile and (ci, co).  Same six points {0, 1, -1, 2, -1/2, inf}:
//   G4  = [1 0 0 0; -1/3 -1/3 -1/3 -1/3; 1/3 -1/3 1/3 -1/3; 1/15 2/15 4/15 8/15; -16/15 8/15 -4/15 2/15; 0 0 0 1]
//   A3^T = [1 1 1 1 1 0; 0 1 -1 2 -1/2 0; 0 1 1 4 1/4 1]
#define DIM_WINO4_G4(O, D, S)                                                                        \
  {                                                                                                  \
    const vf ev_ = D[0] + D[2], od_ = D[1] + D[3];                                                   \
    O[0 * S] = D[0];                                                                                 \
    O[1 * S] = (-1.f / 3) * (ev_ + od_);                                                             \
    O[2 * S] = (1.f / 3) * (ev_ - od_);                                                              \
    O[3 * S] = (1.f / 15) * D[0] + (2.f / 15) * D[1] + (4.f / 15) * D[2] + (8.f / 15) * D[3];        \
    O[4 * S] = (-16.f / 15) * D[0] + (8.f / 15) * D[1] - (4.f / 15) * D[2] + (2.f / 15) * D[3];      \
    O[5 * S] = D[3];                                                                                 \
  }

// D[t][k][c] = (G4 g G4^T)[k] for the 4x4 tile g of dY at (4 ty, 4 tx); thread = (tile, VEC channels)
template <int VEC>
__global__ __launch_bounds__(256) void wino4_dy_kernel(const float* __restrict__ dy, float* __restrict__ D, int N, int H, int W, int C,
                                                       int cstride, int th, int tw, FastDiv div_cq, FastDiv div_tw, FastDiv div_th) {
#pragma clang fp contract(fast)
  typedef typename WinoVec<VEC>::type vf;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const unsigned CQ = C / VEC;
  const unsigned T = (unsigned)N * th * tw;
  const unsigned t = fastdiv(idx, div_cq);
  if (t >= T) return;
  const unsigned cq = idx - t * CQ;
  const unsigned r = fastdiv(t, div_tw);
  const unsigned tx = t - r * tw;
  const unsigned n = fastdiv(r, div_th);
  const unsigned ty = r - n * th;
  const float* base = dy + (long)n * H * W * cstride + cq * VEC;
  vf tmp[24];  // G4 g: [6][4]
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int xx = 4 * (int)tx + b;
    const bool okx = xx < W;
    const int xc = okx ? xx : 0;
    vf g[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = 4 * (int)ty + a;
      const bool ok = okx && yy < H;
      vf v = *reinterpret_cast<const vf*>(base + ((long)(yy < H ? yy : 0) * W + xc) * cstride);
      g[a] = ok ? v : (vf)(0.f);
    }
    vf* o = tmp + b;
    DIM_WINO4_G4(o, g, 4)
  }
  float* out = D + (long)t * 36 * C + cq * VEC;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    vf o[6];
    const vf* g = tmp + 4 * a;
    DIM_WINO4_G4(o, g, 1)
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<vf*>(out + (a * 6 + b) * (long)C) = o[b];
  }
}

// dW = A3^T dM A3 per (co, kk): dM packed [p * K/32 + kk/32][Cout][kk%32] -> MXNet layout.  S = 1: 3x3 / stride-1 layer, K = Cin, dW is
// the (Cout,Cin,3,3) gradient.  S = 2: 5x5 / stride-2 layer, kk = (2 py + px) Cin + ci, and the 3x3 result of phase (py,px) holds the
// taps w[2u + py][2v + px] of the (Cout,Cin,5,5) gradient (u or v = 2 does not exist for an odd phase: dropped).
template <int S>
__global__ __launch_bounds__(256) void wino4_wgrad_output_kernel(const float* __restrict__ dM, float* __restrict__ dw, int Cout, int Cin,
                                                                 float scale, int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int K = Cin * S * S;
  if (idx >= (long)Cout * K) return;
  const int kk = (int)(idx % K), co = (int)(idx / K);  // kk fastest: the 32 k of a packed row are contiguous
  const long per_k = (long)K * Cout;
  const float* in = dM + ((long)(kk >> 5) * Cout + co) * 32 + (kk & 31);
  const float AT[3][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -0.5f, 0.f}, {0.f, 1.f, 1.f, 4.f, 0.25f, 1.f}};
  float r[3][6];  // A3^T dM
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    float m[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) m[a] = in[(a * 6 + b) * per_k];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 6; ++a) acc += AT[u][a] * m[a];
      r[u][b] = acc;
    }
  }
  const int ph = S == 1 ? 0 : kk / Cin, ci = kk - ph * Cin;
  const int py = ph >> 1, px = ph & 1;
  constexpr int KS = S == 1 ? 3 : 5;
  float* o = dw + ((long)co * Cin + ci) * (KS * KS);
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      float acc = 0.f;
#pragma unroll
      for (int b = 0; b < 6; ++b) acc += r[u][b] * AT[v][b];
      const int i = S == 1 ? u : 2 * u + py, j = S == 1 ? v : 2 * v + px;
      if (i < KS && j < KS) {
        const float val = acc * scale;
        o[i * KS + j] = accumulate ? o[i * KS + j] + val : val;
      }
    }
}

}  // namespace dim

extern "C" {

long dim_winograd_packed_weight_floats(int Cout, int Cin, int m) { return wino_packed_with_split((long)(m + 2) * (m + 2) * Cout * Cin); }

// images per slice: tiles * planes * max(K, Cout) floats of one slice stay below 2^32 bytes (32-bit buffer offsets in the plane GEMMs)
static long wino_slice_images(long tiles_per_image, int planes, long K, long Cout) {
  const long per_image = tiles_per_image * planes * (K > Cout ? K : Cout) * 4;
  return per_image < (1L << 32) ? ((1L << 32) - 1) / per_image : 0;
}

long dim_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int m) {
  if (m != 2 && m != 4) return 0;
  const long per = (long)((H + m - 1) / m) * ((W + m - 1) / m);
  const long ns = wino_slice_images(per, (m + 2) * (m + 2), Cin, Cout);
  long T = (N < ns || ns == 0 ? (long)N : ns) * per;
  return (long)(m + 2) * (m + 2) * T * ((long)Cin + Cout);
}

int dim_winograd_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int m, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null weight pointer");
  DIM_REQUIRE(m == 2 || m == 4, "output tile m must be 2 or 4");
  DIM_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "Cin %% 32 == 0 and Cout %% 64 == 0 required");
  long total = (long)Cout * Cin;
  if (m == 2)
    hipLaunchKernelGGL(wino_pack_weight_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cout, Cin);
  else
    hipLaunchKernelGGL(wino4_pack_weight_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cout, Cin);
  int rc = check_launch("winograd_pack_weight");
  return rc != DIM_OK ? rc : wino_split_weights(w_packed, (long)(m + 2) * (m + 2) * (Cin / 32), Cout, as_stream(stream));
}

// transformed weights of the INPUT gradient of a 3x3 / stride-1 / pad-1 layer, straight from its forward (Cout, Cin, 3, 3) array:
// == dim_winograd_pack_weight of w.flip(2, 3).transpose(0, 1), i.e. a Winograd layer with Cin output and Cout input channels
int dim_winograd_dgrad_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int m, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null weight pointer");
  DIM_REQUIRE(m == 2 || m == 4, "output tile m must be 2 or 4");
  DIM_REQUIRE(Cout % 32 == 0 && Cin % 64 == 0, "Cout %% 32 == 0 and Cin %% 64 == 0 required");
  long total = (long)Cout * Cin;
  if (m == 2)
    hipLaunchKernelGGL(wino_pack_weight_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cin, Cout);
  else
    hipLaunchKernelGGL(wino4_pack_weight_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cin, Cout);
  int rc = check_launch("winograd_dgrad_pack_weight");
  return rc != DIM_OK ? rc : wino_split_weights(w_packed, (long)(m + 2) * (m + 2) * (Cout / 32), Cin, as_stream(stream));
}

// one slice of the batch: T * planes * max(K, Cout) floats must stay below 2^32 bytes (32-bit buffer offsets in the GEMM)
static int winograd_slice(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W, int Cin,
                          int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, int m, int S, void** events4,
                          void* stream) {
  const int Ho = S == 1 ? H : (H + 1) / 2, Wo = S == 1 ? W : (W + 1) / 2;  // 5x5 / s2 / p2: floor((H - 1) / 2) + 1
  const int CT = Cin * S * S;                                                  // contraction length of the GEMMs
  const int th = (Ho + m - 1) / m, tw = (Wo + m - 1) / m;
  const int nk = (m + 2) * (m + 2);
  const long T = (long)N * th * tw;
  float* V = workspace;
  float* M = workspace + nk * T * CT;
  hipStream_t st = as_stream(stream);
  const FastDiv dtw = make_fastdiv((unsigned)tw), dth = make_fastdiv((unsigned)th);
#define DIM_WINO_EVENT(I)                                                                  \
  if (events4 && events4[I]) {                                                             \
    hipError_t e = hipEventRecord(reinterpret_cast<hipEvent_t>(events4[I]), st);           \
    if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipEventRecord: %s", hipGetErrorString(e)); \
  }
  if (tile == 0) tile = (Cout % 128 == 0 && T >= 1024) ? 4 : 3;
  WGemmArgs plan;
  int rc = wino_gemm_plan(&plan, V, w_packed, M, (int)T, CT, Cout, nk, tile);
  if (rc != DIM_OK) return rc;
  DIM_WINO_EVENT(0)
  const unsigned nblk = (unsigned)ceil_div(T * (CT / kWino4Vec), 256);
  if (m == 2)
    hipLaunchKernelGGL(wino_input_kernel, dim3(ceil_div(T * (Cin / 4), 256)), dim3(256), 0, st, x, V, N, H, W, Cin, in_cstride, th, tw,
                       make_fastdiv((unsigned)(Cin / 4)), dtw, dth);
  else if (S == 1)  // + G - 1 spare blocks that zero the M tiles shared by two GEMM workgroups
    hipLaunchKernelGGL((wino4_input_kernel<kWino4Vec, 1>), dim3(nblk + plan.G - 1), dim3(256), 0, st, x, V, N, H, W, Cin, in_cstride, th, tw,
                       make_fastdiv((unsigned)(CT / kWino4Vec)), dtw, dth, nblk, plan);
  else
    hipLaunchKernelGGL((wino4_input_kernel<kWino4Vec, 2>), dim3(nblk + plan.G - 1), dim3(256), 0, st, x, V, N, H, W, Cin, in_cstride, th, tw,
                       make_fastdiv((unsigned)(CT / kWino4Vec)), dtw, dth, nblk, plan);
  rc = check_launch("winograd_input");
  if (rc != DIM_OK) return rc;
  DIM_WINO_EVENT(1)
  rc = wino_gemm_run(plan, m == 4, st);
  if (rc != DIM_OK) return rc;
  DIM_WINO_EVENT(2)
  if (m == 2)
    hipLaunchKernelGGL(wino_output_kernel, dim3(ceil_div(T * (Cout / 4), 256)), dim3(256), 0, st, M, bias, y, N, Ho, Wo, Cout, out_cstride,
                       out_coff, th, tw, slope, make_fastdiv((unsigned)(Cout / 4)), dtw, dth);
  else
    hipLaunchKernelGGL((wino4_output_kernel<kWino4Vec, 1>), dim3(ceil_div(T * (Cout / kWino4Vec), 256)), dim3(256), 0, st, M, bias, y, N, Ho, Wo,
                       Cout, out_cstride, out_coff, th, tw, slope, make_fastdiv((unsigned)(Cout / kWino4Vec)), dtw, dth);
  rc = check_launch("winograd_output");
  DIM_WINO_EVENT(3)
#undef DIM_WINO_EVENT
  return rc;
}

// S = 1: 3x3 / stride 1 / pad 1 with output tile m; S = 2: 5x5 / stride 2 / pad 2 through its four phase images (m = 4).
// Large batches run as several slices of whole images through the same workspace (stream order keeps them apart).
static int winograd_impl(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W, int Cin,
                         int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, int m, int S, void** events4,
                         void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && w_packed && y && workspace, "null pointer");
  DIM_REQUIRE(m == 2 || m == 4, "output tile m must be 2 or 4");
  DIM_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "Cin %% 32 == 0 and Cout %% 64 == 0 required");
  if (in_cstride == 0) in_cstride = Cin;
  if (out_cstride == 0) out_cstride = Cout;
  DIM_REQUIRE(in_cstride >= Cin && in_cstride % 4 == 0 && out_cstride >= out_coff + Cout && out_cstride % 4 == 0 && out_coff % 4 == 0,
              "channel strides / offsets must be multiples of 4 and cover the channels");
  const int Ho = S == 1 ? H : (H + 1) / 2, Wo = S == 1 ? W : (W + 1) / 2;
  const long ns = wino_slice_images((long)((Ho + m - 1) / m) * ((Wo + m - 1) / m), (m + 2) * (m + 2), (long)Cin * S * S, Cout);
  DIM_REQUIRE(ns > 0, "one image alone exceeds the 32-bit offsets of the plane GEMMs");
  int n_slice = ns < N ? (int)ns : N;
  if (const char* e = getenv("DIM_WINO_MAX_SLICE")) {  // test hook: force the slicing path at sizes a unit test can check
    const int cap = atoi(e);
    if (cap > 0 && cap < n_slice) n_slice = cap;
  }
  for (int n0 = 0; n0 < N; n0 += n_slice) {
    const int n = N - n0 < n_slice ? N - n0 : n_slice;
    int rc = winograd_slice(x + (long)n0 * H * W * in_cstride, w_packed, bias, y + (long)n0 * Ho * Wo * out_cstride, workspace, n, H, W, Cin,
                            in_cstride, Cout, out_cstride, out_coff, slope, tile, m, S, n0 == 0 ? events4 : nullptr, stream);
    if (rc != DIM_OK) return rc;
  }
  return DIM_OK;
}

int dim_conv2d_fwd_winograd(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                            int Cin, int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, int m,
                            void** events4, void* stream) {
  return winograd_impl(x, w_packed, bias, y, workspace, N, H, W, Cin, in_cstride, Cout, out_cstride, out_coff, slope, tile, m, 1, events4,
                       stream);
}

long dim_winograd5x5s2_packed_weight_floats(int Cout, int Cin) { return wino_packed_with_split(36L * Cout * 4 * Cin); }

long dim_winograd5x5s2_workspace_floats(int N, int H, int W, int Cin, int Cout) {
  const long per = (long)(((H + 1) / 2 + 3) / 4) * (((W + 1) / 2 + 3) / 4);
  const long ns = wino_slice_images(per, 36, 4L * Cin, Cout);
  long T = (N < ns || ns == 0 ? (long)N : ns) * per;
  return 36 * T * (4L * Cin + Cout);
}

int dim_winograd5x5s2_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null weight pointer");
  DIM_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "Cin %% 32 == 0 and Cout %% 64 == 0 required");
  long total = 4L * Cout * Cin;
  hipLaunchKernelGGL(wino4_pack_weight_5x5s2_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cout,
                     Cin);
  int rc = check_launch("winograd5x5s2_pack_weight");
  return rc != DIM_OK ? rc : wino_split_weights(w_packed, 36L * (4 * Cin / 32), Cout, as_stream(stream));
}

int dim_winograd5x5s2_dgrad_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream) {
  DIM_REQUIRE(w_oihw && w_packed, "null weight pointer");
  DIM_REQUIRE(Cout % 32 == 0 && (4 * Cin) % 64 == 0, "Cout %% 32 == 0 and Cin %% 16 == 0 required");
  long total = 4L * Cout * Cin;
  hipLaunchKernelGGL(wino4_pack_weight_5x5s2_dgrad_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w_oihw, w_packed,
                     Cout, Cin);
  int rc = check_launch("winograd5x5s2_dgrad_pack_weight");
  return rc != DIM_OK ? rc : wino_split_weights(w_packed, 36L * (Cout / 32), 4 * Cin, as_stream(stream));
}

int dim_conv2d_dgrad_winograd5x5s2(const float* dy, const float* w_packed, float* dx, float* workspace, int N, int H, int W, int Cin,
                                   int dx_cstride, int Cout, int dy_cstride, int tile, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(dy && w_packed && dx && workspace, "null pointer");
  DIM_REQUIRE(Cout % 32 == 0 && (4 * Cin) % 64 == 0 && Cin % 2 == 0, "Cout %% 32 == 0 and Cin %% 16 == 0 required");
  if (dx_cstride == 0) dx_cstride = Cin;
  if (dy_cstride == 0) dy_cstride = Cout;
  DIM_REQUIRE(dx_cstride >= Cin && dx_cstride % 2 == 0 && dy_cstride >= Cout && dy_cstride % 2 == 0, "channel strides must cover the channels");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int th = (Ho + 3) / 4, tw = (Wo + 3) / 4;
  const int CT = 4 * Cin;
  const long ns = wino_slice_images((long)th * tw, 36, Cout, CT);
  DIM_REQUIRE(ns > 0, "one image alone exceeds the 32-bit offsets of the plane GEMMs");
  hipStream_t st = as_stream(stream);
  const FastDiv dtw = make_fastdiv((unsigned)tw), dth = make_fastdiv((unsigned)th);
  for (int n0 = 0; n0 < N; n0 += (int)ns) {
    const int n = N - n0 < ns ? N - n0 : (int)ns;
    const long T = (long)n * th * tw;
    float* V = workspace;
    float* M = workspace + 36 * T * Cout;
    WGemmArgs plan;
    int rc = wino_gemm_plan(&plan, V, w_packed, M, (int)T, Cout, CT, 36, tile == 0 ? ((CT % 128 == 0 && T >= 1024) ? 4 : 3) : tile);
    if (rc != DIM_OK) return rc;
    const unsigned nblk = (unsigned)ceil_div(T * (Cout / kWino4Vec), 256);
    hipLaunchKernelGGL((wino4_input_kernel<kWino4Vec, 1>), dim3(nblk + plan.G - 1), dim3(256), 0, st, dy + (long)n0 * Ho * Wo * dy_cstride, V, n,
                       Ho, Wo, Cout, dy_cstride, th, tw, make_fastdiv((unsigned)(Cout / kWino4Vec)), dtw, dth, nblk, plan);
    rc = check_launch("winograd_dgrad_input");
    if (rc != DIM_OK) return rc;
    rc = wino_gemm_run(plan, true, st);
    if (rc != DIM_OK) return rc;
    hipLaunchKernelGGL((wino4_output_kernel<kWino4Vec, 2>), dim3(ceil_div(T * (CT / kWino4Vec), 256)), dim3(256), 0, st, M, nullptr,
                       dx + (long)n0 * H * W * dx_cstride, n, H, W, Cin, dx_cstride, 0, th, tw, 1.0f, make_fastdiv((unsigned)(CT / kWino4Vec)),
                       dtw, dth);
    rc = check_launch("winograd_dgrad_output");
    if (rc != DIM_OK) return rc;
  }
  return DIM_OK;
}

// Weight gradient through Winograd.  S = 1: 3x3 / stride 1 / pad 1; S = 2: 5x5 / stride 2 / pad 2 (phase images of x).
long dim_conv2d_wgrad_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int S, int splits) {
  const int Ho = S == 1 ? H : (H + 1) / 2, Wo = S == 1 ? W : (W + 1) / 2;
  const long T = (long)N * ((Ho + 3) / 4) * ((Wo + 3) / 4);
  const long K = (long)Cin * S * S;
  if (splits < 1) splits = 1;
  return 36 * T * (K + Cout) + 36 * K * Cout * (long)(splits + 1);
}

int dim_conv2d_wgrad_winograd(const float* x, const float* dy, float* dw_oihw, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                              int Cout, int dy_cstride, int S, int splits, float scale, int accumulate, void* stream) {
  if (N == 0) return DIM_OK;
  DIM_REQUIRE(x && dy && dw_oihw && workspace, "null pointer");
  DIM_REQUIRE(S == 1 || S == 2, "S must be 1 (3x3 / stride 1) or 2 (5x5 / stride 2)");
  DIM_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && (Cin * S * S) % 64 == 0, "Cin %% 32 == 0 (%% 64 for S = 1) and Cout %% 64 == 0 required");
  if (in_cstride == 0) in_cstride = Cin;
  if (dy_cstride == 0) dy_cstride = Cout;
  DIM_REQUIRE(in_cstride >= Cin && in_cstride % 2 == 0 && dy_cstride >= Cout && dy_cstride % 2 == 0, "channel strides must cover the channels");
  const int Ho = S == 1 ? H : (H + 1) / 2, Wo = S == 1 ? W : (W + 1) / 2;
  const int th = (Ho + 3) / 4, tw = (Wo + 3) / 4;
  const long T = (long)N * th * tw;
  const int K = Cin * S * S;
  DIM_REQUIRE(T * 36 * (K > Cout ? K : Cout) < (1L << 29), "winograd wgrad: batch too large for 32-bit byte offsets");
  if (splits < 1) splits = 1;
  float* V = workspace;
  float* D = V + 36 * T * K;
  float* dM = D + 36 * T * Cout;
  float* slabs = dM + 36L * K * Cout;
  hipStream_t st = as_stream(stream);
  const FastDiv dtw = make_fastdiv((unsigned)tw), dth = make_fastdiv((unsigned)th);
  const unsigned nblk = (unsigned)ceil_div(T * (K / kWino4Vec), 256);  // no stream-K GEMM follows: no spare blocks
  const WGemmArgs none = {};
  if (S == 1)
    hipLaunchKernelGGL((wino4_input_kernel<kWino4Vec, 1>), dim3(nblk), dim3(256), 0, st, x, V, N, H, W, Cin, in_cstride, th, tw,
                       make_fastdiv((unsigned)(K / kWino4Vec)), dtw, dth, nblk, none);
  else
    hipLaunchKernelGGL((wino4_input_kernel<kWino4Vec, 2>), dim3(nblk), dim3(256), 0, st, x, V, N, H, W, Cin, in_cstride, th, tw,
                       make_fastdiv((unsigned)(K / kWino4Vec)), dtw, dth, nblk, none);
  hipLaunchKernelGGL(wino4_dy_kernel<kWino4Vec>, dim3(ceil_div(T * (Cout / kWino4Vec), 256)), dim3(256), 0, st, dy, D, N, Ho, Wo, Cout,
                     dy_cstride, th, tw, make_fastdiv((unsigned)(Cout / kWino4Vec)), dtw, dth);
  int rc = check_launch("winograd_wgrad_transforms");
  if (rc != DIM_OK) return rc;
  rc = launch_wgrad_planes(V, D, dM, slabs, (int)T, K, Cout, 36, splits, st);
  if (rc != DIM_OK) return rc;
  if (S == 1)
    hipLaunchKernelGGL(wino4_wgrad_output_kernel<1>, dim3(ceil_div((long)Cout * K, 256)), dim3(256), 0, st, dM, dw_oihw, Cout, Cin, scale,
                       accumulate);
  else
    hipLaunchKernelGGL(wino4_wgrad_output_kernel<2>, dim3(ceil_div((long)Cout * K, 256)), dim3(256), 0, st, dM, dw_oihw, Cout, Cin, scale,
                       accumulate);
  return check_launch("winograd_wgrad_output");
}

int dim_conv2d_fwd_winograd5x5s2(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                                 int Cin, int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, void** events4,
                                 void* stream) {
  return winograd_impl(x, w_packed, bias, y, workspace, N, H, W, Cin, in_cstride, Cout, out_cstride, out_coff, slope, tile, 4, 2, events4,
                       stream);
}

}  // extern "C"
