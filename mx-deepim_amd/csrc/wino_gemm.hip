// The P independent GEMMs of a Winograd layer (P = 16 or 36 planes:  M_p[T x Cout] = V_p[T x K] . U_p[K x Cout]) as ONE persistent
// launch.  (deepim/symbols/deepIM_flownet.py:95-191 are the layers; see conv.hip for the transforms around this kernel.)
//
// Why not blockIdx.y = plane on conv_fwd_kernel (the first version): a plane's K loop is only K / 32 = 8 or 16 chunks long, so
// every workgroup paid its pipeline fill (first global loads -> LDS -> barrier) and its drain (epilogue stores, exit, dispatch of
// the successor) for that little work -- measured 2.1 chunk-times per (plane, tile), 103-115 TFLOP/s against 127-134 for the long
// loops of the direct layers -- and tiles x planes workgroups of equal length quantise badly on 512 resident slots (conv3: 2736
// workgroups = 5.34 rounds -> 6).
//
// Here the (row tile, Cout tile, plane, K chunk) space is ONE flat list of chunks, dealt in equal contiguous ranges to exactly as
// many workgroups as are resident at once (stream-K).  A workgroup walks its range with the software pipeline of conv_fwd_kernel
// never draining: the prefetch of the next (plane, tile)'s first chunks is in flight while the last chunk of the current one is
// multiplied; at the end of an item the accumulators are stored and cleared between two chunks.  Ranges start and end anywhere,
// so at most one item per workgroup boundary is shared by two workgroups; its output tile is zeroed beforehand
// (by spare blocks of the input-transform kernel that runs in front: common.h wino_gemm_zero_tile) and both add their partial sums with float atomics -- two summands on a zero: the result does not
// depend on their order, the launch stays deterministic.
//
// Layouts (plane-minor, so that "next plane" is just "next K columns" for the loader):
//   V [T][P][K]      row stride P*K floats
//   U [P][K/32][Cout][32]   = the packed 1x1 weights of conv_fwd_kernel, plane-major: flat chunk index p*K/32 + c
//   M [T][P][Cout]   row stride P*Cout floats
#include <cstdlib>

#include "common.h"

namespace dim {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void wcur_advance(WCur& c, const WGemmArgs& a) {
  if (++c.ch == a.nch) {
    c.ch = 0;
    if (a.plane_major) {
      if (++c.nt == a.NTN) {
        c.nt = 0;
        if (++c.mt == a.MT) {
          c.mt = 0;
          ++c.p;
        }
      }
    } else if (++c.p == a.P) {
      c.p = 0;
      if (++c.nt == a.NTN) {
        c.nt = 0;
        ++c.mt;
      }
    }
  }
}
// BN = 256 (tile 5): every wave owns a 64 x 64 block = four accumulators (64 registers) and twice the B fragments, so it is built
// for two waves per SIMD = ONE 8-wave workgroup per CU; V is then read once for 256 output channels instead of once per 128.
// Few-row layers (round 4: conv5 / conv5_1 with T = 320 tile rows at 16 pairs, conv6_1 with T = 96): BM = 160 or 96 rows x 128 columns on
// FOUR waves, every wave one 32-column strip of ALL the rows (TM = 5 or 3 accumulators).  The 64 x 64 tile they ran on does 16 MFMAs
// per wave between two barriers (0.54 of the peak: the staging, the barrier and the fragment reads of a chunk are the same whatever the
// tile holds) and pads 96 rows to 128; these do 80 / 48, with no row padding at 320 / 96 rows.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(BN == 256 || BM == 160 ? 2 : (BM == 96 ? 3 : 4), 8))) void wino_gemm_kernel(WGemmArgs a) {
  constexpr int BK = 32;
  constexpr int NT = WM * WN * 64;
  constexpr int RP = NT / 8;   // rows staged per pass (8 threads x float4 = one 32-float row)
  constexpr int LDK = BK + 4;  // conflict-free for ds_write_b128 staging and ds_read_b128 fragments (see conv.hip)
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int NSTG = BM / RP;  // staging loads per thread and chunk
  constexpr int kTrLd = 32, kTrFloats = 16 * kTrLd;  // row stride / size of a wave's transposing buffer: 16 rows x 32 floats = 2 KB
  constexpr bool kWide = !(BM == 128 && BN == 128);  // wide flush (below): not on the 128 x 128 tile, whose 126 registers have no room for it (it spills)
  static_assert(BM % RP == 0 && NSTG >= 2 && NSTG <= 5, "two to five staging loads per thread");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;  // [2][BM][LDK]
  float* sTr = smem + 2 * BM * LDK;  // [waves][32][32]: every wave's own transposing buffer for the flush of a whole item

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int q = tid & 7;
  const int srow = tid >> 3;
  const int frow = lane & 31;
  const int khalf = lane >> 5;

  const int wg = a.plane_major ? wg_xcd_contiguous(blockIdx.x, a.G) : (int)blockIdx.x;
  const int c_begin = wg_first_chunk(wg, a);
  const int c_end = wg_first_chunk(wg + 1, a);

  const int RS = a.P * a.K;  // V row stride (floats)
  const int a_voff0 = (srow * RS + q * 4) * 4;
  const int a_vstep = RP * RS * 4;  // next staging pass: RP rows further down
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V), 0, a.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U), 0, a.u_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, a.m_bytes, 0x00020000);
  const int wchunk_bytes = a.Cout * BK * 4;

  // three cursors into the chunk list: L = next A chunk to load (runs two ahead), Bc = next B fragments (one ahead), C = compute
  WCur L = wcur_decode(c_begin, a), Bc = L, C = L;

  float4 ra0, ra1, ra2, ra3, ra4;  // staging registers of the A chunk (named: arrays ended up in scratch); NSTG of them are live
  // rows past T read as zeros through the descriptor's range check (offset 0xFFFFFFFF); so does every load past the range's end.
  // The scalar offset is NOT range-checked, which is fine: it is only ever added to an in-range or a rejected vector offset.
#define W_LOAD_CHUNK(PF_OK)                                                            \
  {                                                                                    \
    const bool pf = (PF_OK);                                                           \
    const int mb = L.mt * BM;                                                          \
    const int soff = (int)((unsigned)(mb * RS + L.p * a.K + L.ch * BK) * 4u);          \
    ra0 = buf_load16_nt(rv, (pf && mb + srow < a.T) ? a_voff0 : -1, soff);             \
    ra1 = buf_load16_nt(rv, (pf && mb + srow + RP < a.T) ? a_voff0 + a_vstep : -1, soff);        \
    if constexpr (NSTG > 2) ra2 = buf_load16_nt(rv, (pf && mb + srow + 2 * RP < a.T) ? a_voff0 + 2 * a_vstep : -1, soff); \
    if constexpr (NSTG > 3) ra3 = buf_load16_nt(rv, (pf && mb + srow + 3 * RP < a.T) ? a_voff0 + 3 * a_vstep : -1, soff); \
    if constexpr (NSTG > 4) ra4 = buf_load16_nt(rv, (pf && mb + srow + 4 * RP < a.T) ? a_voff0 + 4 * a_vstep : -1, soff); \
    wcur_advance(L, a);                                                                \
  }
#define W_STORE_CHUNK(BUF)                                                             \
  {                                                                                    \
    float* dA = sA + (BUF) * BM * LDK;                                                 \
    *reinterpret_cast<float4*>(dA + srow * LDK + q * 4) = ra0;                         \
    *reinterpret_cast<float4*>(dA + (srow + RP) * LDK + q * 4) = ra1;                  \
    if constexpr (NSTG > 2) *reinterpret_cast<float4*>(dA + (srow + 2 * RP) * LDK + q * 4) = ra2; \
    if constexpr (NSTG > 3) *reinterpret_cast<float4*>(dA + (srow + 3 * RP) * LDK + q * 4) = ra3; \
    if constexpr (NSTG > 4) *reinterpret_cast<float4*>(dA + (srow + 4 * RP) * LDK + q * 4) = ra4; \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_off = (wm * (BM / WM) + frow) * LDK + 4 * khalf;
  float4 fa[2][TM];
#define W_FRAG_READ(IDX, PA, S)                                                                                        \
  {                                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[IDX][i] = *reinterpret_cast<const float4*>((PA) + 32 * i * LDK + 8 * (S)); \
  }
  // B fragments straight from L2, one chunk ahead (fbq[set][group][tile]); past the end of the range: chunk 0 (unused values)
  float4 fbq[2][4][TN];
  const int bf_voff = ((wn * (BN / WN) + frow) * BK + 4 * khalf) * 4;
#define W_LOAD_BFRAG(SET, VALID)                                                                                       \
  {                                                                                                                    \
    const int bsoff = (VALID) ? (Bc.p * a.nch + Bc.ch) * wchunk_bytes + Bc.nt * BN * BK * 4 : 0;                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int j = 0; j < TN; ++j)                        \
      fbq[SET][g][j] = buf_load16(ru, bf_voff + (32 * j * BK + 8 * g) * 4, bsoff);                                      \
    wcur_advance(Bc, a);                                                                                               \
  }
#define W_MFMA_GROUP(IDX, SET, G)                                                                                      \
  {                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) {                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].x, fbq[SET][G][j].x, acc[i][j], 0, 0, 0);             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].y, fbq[SET][G][j].y, acc[i][j], 0, 0, 0);             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].z, fbq[SET][G][j].z, acc[i][j], 0, 0, 0);             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[IDX][i].w, fbq[SET][G][j].w, acc[i][j], 0, 0, 0);             \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }

  // ---- output: D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int ldc = a.P * a.Cout;
  const int o_row = wm * (BM / WM) + 4 * khalf;  // + 32 i + (r & 3) + 8 (r >> 2)
  const int o_col = wn * (BN / WN) + frow;       // + 32 j
  const int o_voff = (o_row * ldc + o_col) * 4;
  // `whole`: this workgroup has multiplied every chunk of the current item -> plain stores; otherwise the item is shared with the
  // neighbouring workgroup, its tile was zeroed before the launch, and both add
  bool whole = C.ch == 0;
#define W_FLUSH_CHECK(KCUR)                                                                                            \
  {                                                                                                                    \
    const bool item_end = C.ch == a.nch - 1;                                                                           \
    if (item_end || (KCUR) == c_end - 1) {                                                                             \
      const int mb = C.mt * BM;                                                                                        \
      const int cb = C.p * a.Cout + C.nt * BN;                                                                         \
      const int lim = a.T - mb;                                                                                        \
      int ov = o_voff, orow = o_row; /* opaque copies: keeps the 32 store offsets from being hoisted out of the K loop */ \
      asm volatile("" : "+v"(ov), "+v"(orow));                                                                         \
      const int soff = (int)((unsigned)(mb * ldc + cb) * 4u);                                                          \
      const bool plain = (whole && item_end) || a.dbg_plain;                                                           \
      if (kWide && plain && a.wide_flush) {                                                                            \
        /* whole item: every 32 x 32 accumulator tile goes, 16 rows at a time, through the wave's own 2 KB of LDS (the MFMA result   \
           has a column per lane and rows in the registers: 16 one-dword stores per tile, 64 on the 128 x 256 tile -- in-kernel     \
           stamps: 7.5 k cycles per flush there, 10.7 % of a workgroup's life) and leaves as 16-byte stores of 8 rows x 128 B each  \
           (same box A/B: dominant kernel 110.9 -> 113.7 TFLOP/s, loop +0.6 %) */                                                \
        float* sT = sTr + wave * kTrFloats;                                                                            \
        const int trow = lane >> 3, tc4 = (lane & 7) * 4;                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                   \
          _Pragma("unroll") for (int h = 0; h < 2; ++h) { /* rows 16 h .. 16 h + 15 of the tile = registers 8 h .. 8 h + 7 */ \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * khalf) * kTrLd + frow] = acc[i][j][8 * h + r]; \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            /* both reads first, then the two stores (8 LDS writes and a wait lie between a store and the next read into its registers) */ \
            const float4 t0 = *reinterpret_cast<const float4*>(sT + trow * kTrLd + tc4);                               \
            const float4 t1 = *reinterpret_cast<const float4*>(sT + (trow + 8) * kTrLd + tc4);                         \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            const int gr = wm * (BM / WM) + 32 * i + 16 * h + trow;                                                    \
            const int vo0 = (gr < lim) ? (gr * ldc + wn * (BN / WN) + 32 * j + tc4) * 4 : -1;                          \
            const int vo1 = (gr + 8 < lim) ? ((gr + 8) * ldc + wn * (BN / WN) + 32 * j + tc4) * 4 : -1;                \
            u32x4 u0, u1;                                                                                              \
            u0.x = __float_as_uint(t0.x); u0.y = __float_as_uint(t0.y); u0.z = __float_as_uint(t0.z); u0.w = __float_as_uint(t0.w); \
            u1.x = __float_as_uint(t1.x); u1.y = __float_as_uint(t1.y); u1.z = __float_as_uint(t1.z); u1.w = __float_as_uint(t1.w); \
            __builtin_amdgcn_raw_buffer_store_b128(u0, rm, vo0, soff, 2);                                              \
            __builtin_amdgcn_raw_buffer_store_b128(u1, rm, vo1, soff, 2);                                              \
            asm volatile("" ::: "memory");                                                                             \
          }                                                                                                            \
      } else                                                                                                           \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
          const int dr = 32 * i + (r & 3) + 8 * (r >> 2);                                                              \
          const int vo = (orow + dr < lim) ? ov + (dr * ldc + 32 * j) * 4 : -1; /* rows past T: rejected by the range check */ \
          if (plain)                                                                                                   \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rm, vo, soff, 2);                     \
          else if (vo != -1) /* exec-masked rather than rejected: an out-of-range buffer ATOMIC faulted on this part */  \
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][j][r], rm, vo, soff, 0);                            \
        }                                                                                                              \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;                                             \
      whole = true;                                                                                                    \
    }                                                                                                                  \
    wcur_advance(C, a);                                                                                                \
  }

  // ---- software pipeline per chunk of 32 (four MFMA groups), as in conv_fwd_kernel:
  //   g0 | g1 | [registers -> LDS for chunk k+1, global loads for chunk k+2] | g2 | LDS barrier | [fragments g0 of chunk k+1] | g3
  W_LOAD_CHUNK(true)
  W_STORE_CHUNK(0)
  W_LOAD_BFRAG(0, true)
  __syncthreads();
  W_LOAD_CHUNK(c_begin + 1 < c_end)
  W_FRAG_READ(0, sA + a_off, 0)

#define W_CHUNK_BODY(SET, KCUR)                                                        \
  {                                                                                    \
    const float* cA = sA + buf * BM * LDK + a_off;                                     \
    const float* nA = sA + (buf ^ 1) * BM * LDK + a_off;                               \
    W_LOAD_BFRAG(1 - SET, (KCUR) + 1 < c_end)                                          \
    W_FRAG_READ(1, cA, 1)                                                              \
    W_MFMA_GROUP(0, SET, 0)                                                            \
    W_FRAG_READ(0, cA, 2)                                                              \
    W_MFMA_GROUP(1, SET, 1)                                                            \
    W_STORE_CHUNK(buf ^ 1)                                                             \
    W_LOAD_CHUNK((KCUR) + 2 < c_end)                                                   \
    W_FRAG_READ(1, cA, 3)                                                              \
    W_MFMA_GROUP(0, SET, 2)                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                    \
    W_FRAG_READ(0, nA, 0)                                                              \
    W_MFMA_GROUP(1, SET, 3)                                                            \
    buf ^= 1;                                                                          \
    W_FLUSH_CHECK(KCUR)                                                                \
  }
  int buf = 0;
  for (int kc = c_begin; kc < c_end; kc += 2) {
    W_CHUNK_BODY(0, kc)
    if (kc + 1 < c_end) W_CHUNK_BODY(1, kc + 1)
  }
#undef W_CHUNK_BODY
#undef W_FLUSH_CHECK
#undef W_MFMA_GROUP
#undef W_LOAD_BFRAG
#undef W_FRAG_READ
#undef W_STORE_CHUNK
#undef W_LOAD_CHUNK
}

// zero the output tile of every item that two workgroups share (only when the transform kernel in front did not do it)
__global__ __launch_bounds__(256) void wino_gemm_zero_kernel(WGemmArgs a) { wino_gemm_zero_tile(a, blockIdx.x + 1); }

// DIM_WINO_SPLIT=0 / dim_set_winograd_split(0): every plane GEMM on the f32 matrix pipe (wino_gemm_kernel)
static int g_wino_split = getenv("DIM_WINO_SPLIT") ? atoi(getenv("DIM_WINO_SPLIT")) : 1;
void wino_set_split(int on) { g_wino_split = on; }
int wino_get_split() { return g_wino_split; }
static int g_wino_cus = 0;  // compute units of the device (set with the first occupancy query)
template <int BM, int BN, int WM, int WN>
static int wino_gemm_slots() {
  // as many workgroups as are resident at once (occupancy x CUs)
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0, occ = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&wino_gemm_kernel<BM, BN, WM, WN>),
                                                       WM * WN * 64, (2 * BM * 36 + WM * WN * 512) * sizeof(float));
    if (e != hipSuccess || cus <= 0 || occ <= 0) return set_err(DIM_ERR_LAUNCH, "winograd gemm occupancy query: %s", hipGetErrorString(e));
    slots = cus * occ;
    g_wino_cus = cus;
  }
  return slots;
}

// tile: 5 = 128x256 (8 waves, 64x64 per wave), 4 = 128x128 (8 waves), 6 = 160x128 and 7 = 96x128 (4 waves, every wave a 32-column strip
// of all the rows: the few-row layers), anything else = 64x64 (4 waves)
int wino_gemm_plan(WGemmArgs* plan, const float* V, const float* U, float* M, int T, int K, int Cout, int P, int tile) {
  if (tile == 5 && Cout % 256 != 0) tile = 4;
  if ((tile < 4 || tile > 7) || Cout % 128 != 0) tile = 3;
  const int BM = tile == 3 ? 64 : tile == 6 ? 160 : tile == 7 ? 96 : 128, BN = tile == 5 ? 256 : tile == 3 ? 64 : 128;
  DIM_REQUIRE(K % 32 == 0 && Cout % BN == 0 && T > 0, "winograd gemm: K %% 32 == 0 and Cout %% 64 == 0 required");
  const long MT = (T + BM - 1) / BM;
  const long items = MT * (Cout / BN) * P;
  DIM_REQUIRE(items * (K / 32) < (1L << 31) && (long)T * P * K * 4 < (1L << 32) && (long)T * P * Cout * 4 < (1L << 32) &&
                  (long)P * K * Cout * 4 < (1L << 31) && (long)BM * P * (K > Cout ? K : Cout) * 4 < (1L << 31),
              "winograd gemm: problem too large for 32-bit byte offsets (split the batch)");
  WGemmArgs a = {};
  a.V = V;
  a.U = U;
  a.M = M;
  a.T = T;
  a.K = K;
  a.Cout = Cout;
  a.P = P;
  a.nch = K / 32;
  a.NTN = Cout / BN;
  a.v_bytes = (unsigned)((long)T * P * K * 4);
  a.u_bytes = (unsigned)((long)P * K * Cout * 4);
  a.m_bytes = (unsigned)((long)T * P * Cout * 4);
  a.d_nch = make_fastdiv((unsigned)a.nch);
  a.d_P = make_fastdiv((unsigned)P);
  a.d_NTN = make_fastdiv((unsigned)a.NTN);
  a.MT = (int)MT;
  a.d_MT = make_fastdiv((unsigned)MT);
  static const int order = getenv("DIM_WINO_PLANE_MAJOR") ? atoi(getenv("DIM_WINO_PLANE_MAJOR")) : 1;  // A/B switch (0 = first version)
  a.plane_major = order;
  static const int dbg_plain = getenv("DIM_WINO_DBG_PLAIN") ? atoi(getenv("DIM_WINO_DBG_PLAIN")) : 0;  // timing only: WRONG sums
  a.dbg_plain = dbg_plain;
  static const int wide = getenv("DIM_WINO_WIDE_FLUSH") ? atoi(getenv("DIM_WINO_WIDE_FLUSH")) : 1;   // A/B timing: 0 = one-dword stores
  a.wide_flush = wide;
  a.BM = BM;
  a.BN = BN;
  a.tile = tile;
  // f32 operands as three bf16 terms on the bf16 matrix pipe (wino_gemm_split.hip) where a split kernel of this tile shape exists
  a.split = g_wino_split && wino_gemm_split_has(tile) && (long)P * K * Cout * 6 < (1L << 31);
  a.U3 = U + (long)P * K * Cout;
  a.u3_bytes = (unsigned)((long)P * K * Cout * 6);
  if (a.split) a.plane_major = 1;
  // never more workgroups than items: every range is then at least one item long and an item is shared by at most two workgroups
  const int slots = a.split ? wino_gemm_split_slots(tile) : tile == 5 ? wino_gemm_slots<128, 256, 2, 4>() : tile == 4 ? wino_gemm_slots<128, 128, 2, 4>()
                  : tile == 6 ? wino_gemm_slots<160, 128, 1, 4>() : tile == 7 ? wino_gemm_slots<96, 128, 1, 4>() : wino_gemm_slots<64, 64, 2, 2>();
  if (slots <= 0) return slots;
  a.G = items < slots ? (int)items : slots;
  // fewer items than slots: the hardware deals workgroups to the CUs one by one, so 288 workgroups on 256 CUs leave 32 CUs with twice the
  // work of the others (conv5_1 / conv6_1 on the few-row tiles: 90 / 103 us against 70 / 107 on the 64 x 64 tile before this rule).  A
  // whole number of workgroups per CU, each with a slightly longer range, balances them.  DIM_WINO_G_ROUND=0: off (A/B timing).
  static const int g_round = getenv("DIM_WINO_G_ROUND") ? atoi(getenv("DIM_WINO_G_ROUND")) : 1;
  if (g_wino_cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_wino_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) g_wino_cus = 0;
  }
  if (g_round && g_wino_cus > 0 && a.G > g_wino_cus && a.G < slots) a.G = a.G / g_wino_cus * g_wino_cus;
  const long total = items * a.nch;
  a.per = (int)(total / a.G);
  a.rem = (int)(total % a.G);
  *plan = a;
  return DIM_OK;
}

template <int BM, int BN, int WM, int WN>
static int wino_gemm_launch(const WGemmArgs& a, hipStream_t st) {
  constexpr size_t lds = (2 * BM * 36 + WM * WN * 512) * sizeof(float);   // A double buffer + every wave's transposing buffer
  // measured this round: with 69.6 KB of dynamic LDS (4 KB transposing buffers on the 128 x 256 tile) the launch succeeded after
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the addresses above 64 KB read back other waves' data -- stay below it
  static_assert(lds <= 65536, "dynamic LDS of a plane-GEMM workgroup stays within 64 KB");
  hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN>), dim3(a.G), dim3(WM * WN * 64), lds, st, a);
  return check_launch("winograd_gemm");
}

int wino_gemm_run(const WGemmArgs& a, bool zeroed, hipStream_t st) {
  if (!zeroed && a.G > 1) hipLaunchKernelGGL(wino_gemm_zero_kernel, dim3(a.G - 1), dim3(256), 0, st, a);
  if (a.split) return wino_gemm_split_run(a, st);
  if (a.tile == 5) return wino_gemm_launch<128, 256, 2, 4>(a, st);
  if (a.tile == 4) return wino_gemm_launch<128, 128, 2, 4>(a, st);
  if (a.tile == 6) return wino_gemm_launch<160, 128, 1, 4>(a, st);
  if (a.tile == 7) return wino_gemm_launch<96, 128, 1, 4>(a, st);
  return wino_gemm_launch<64, 64, 2, 2>(a, st);
}

int launch_wino_gemm(const float* V, const float* U, float* M, int T, int K, int Cout, int P, int tile, hipStream_t st) {
  WGemmArgs a;
  int rc = wino_gemm_plan(&a, V, U, M, T, K, Cout, P, tile);
  if (rc != DIM_OK) return rc;
  return wino_gemm_run(a, false, st);
}

}  // namespace dim
