// The plane GEMMs of wino_gemm.hip on the bf16 matrix pipe WITHOUT giving up f32 operands: every f32 operand is the exact sum of
// three bf16 terms  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): round to nearest even, the subtractions are
// exact; what is left of x is below 2^-27 |x|), and a product keeps the six term products down to 2^-16 of it,
//      x y  ~  h h' + (h m' + m h') + (h l' + m m' + l h')          dropped: m l' + l m' + l l'  <=  3 * 2^-27 |x y|,
// each exact in the f32 accumulator's input (8 x 8 significant bits), accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Six MFMAs
// of 32 cycles replace the sixteen 64-cycle v_mfma_f32_32x32x2_f32 of a 32 x 32 x 32 block: 2.67x the f32 matrix rate at an error
// below f32's own rounding of the sum -- the parity bars of tests/ (2e-5 on a pose step against the f64-accumulating oracle) hold
// unchanged; every Winograd test of tests/test_gpu_ops.py runs both arithmetics (`wino_split` fixture), tests/test_split_terms.py
// asserts the bounds on the CPU.  (deepim/symbols/deepIM_flownet.py:95-191 are the layers.)
//
// Operands:
//   V [T][P][K] f32  as in wino_gemm.hip; split by the staging threads on the way into LDS (11 VALU ops per pair of floats)
//   U3               the split image of U [P][K/32][Cout][32], written once per weight update by wino_split_weights:
//                    [P * K/32 chunks][Cout/32 column tiles][3 terms][2 k-steps][32 columns][2 k-halves][8 bf16]   = 6 KB per
//                    (chunk, column tile), and the 1 KB of one (term, k-step) is exactly the B operand of one MFMA in lane order
//   M [T][P][Cout]   f32, stream-K partition, shared-item atomics and wide flush as in wino_gemm.hip
// Workgroup = WN waves, wave w = ALL BM rows x columns 32 w .. 32 w + 31 (TM = BM / 32 accumulators): every B fragment is fetched
// from L2 by exactly one wave of the workgroup (an XCD's L2 feeds ~29 B/clk/CU; this shape needs 16 at the matrix pipe's full rate,
// a 64 x 64 wave tile would need 32), the A image in LDS is read by all of them (62 of LDS's 256 B/clk/CU).
// LDS: A [2 buffers][3 terms][BM rows][32 bf16 = 64 B], the four 16-byte slots of a row XOR-swizzled with bits 2-3 of the row
// number: the 16 lanes that ds_read_b128 serves per cycle (16 rows distinct mod 16) then cover all 64 banks once, without padding --
// 48 KB at BM = 128, which leaves 16 KB of the 64 KB a workgroup may use here for the waves' transposing buffers (wide flush).
#include <cstdlib>

#include "common.h"

// timing experiments (tools/split_exp.sh builds one library per value; results are WRONG with any bit set):
// 1 no MFMAs, 2 no V loads, 4 no U3 loads, 8 no split + LDS stores, 16 no fragment reads, 32 no barrier, 64 no flush
#ifndef DIM_SPLIT_EXP
#define DIM_SPLIT_EXP 0
#endif
// timing / cache-policy options of the flush (bit set = variant): 2 plain (not non-temporal) M stores on every tile, 8 no global stores,
// 16 no LDS round trip, 32 accumulators not cleared, 64 results not kept alive -- 8 .. 64 give WRONG results (tools/split_exp.sh oN)
#ifndef DIM_SPLIT_OPT
#define DIM_SPLIT_OPT 0
#endif

namespace dim {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Split4 {
  uint2 h, m, l;
};
__device__ __forceinline__ Split4 split3(const float4 v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  const bf16x4 bh = __builtin_convertvector(x, bf16x4);
  const f32x4 r1 = x - __builtin_convertvector(bh, f32x4);
  const bf16x4 bm = __builtin_convertvector(r1, bf16x4);
  const f32x4 r2 = r1 - __builtin_convertvector(bm, f32x4);
  const bf16x4 bl = __builtin_convertvector(r2, bf16x4);
  Split4 s;
  s.h = __builtin_bit_cast(uint2, bh);
  s.m = __builtin_bit_cast(uint2, bm);
  s.l = __builtin_bit_cast(uint2, bl);
  return s;
}

// U [chunks][Cout][32] f32 -> U3 (layout above).  One thread = 8 consecutive k of one column.
__global__ __launch_bounds__(256) void wino_split_weights_kernel(const float* __restrict__ U, unsigned char* __restrict__ U3, long total, int Cout,
                                                                 FastDiv d_cout) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t & 3);
  const unsigned cn = (unsigned)(t >> 2);  // chunk * Cout + column
  const unsigned c = fastdiv(cn, d_cout);
  const unsigned n = cn - c * (unsigned)Cout;
  const float4* src = reinterpret_cast<const float4*>(U + (long)cn * 32 + 8 * g);
  const Split4 a = split3(src[0]), b = split3(src[1]);
  unsigned char* dst = U3 + (long)c * Cout * 192 + (long)(n >> 5) * 6144 + ((g >> 1) * 32 + (n & 31)) * 32 + (g & 1) * 16;
  *reinterpret_cast<uint4*>(dst) = make_uint4(a.h.x, a.h.y, b.h.x, b.h.y);
  *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(a.m.x, a.m.y, b.m.x, b.m.y);
  *reinterpret_cast<uint4*>(dst + 4096) = make_uint4(a.l.x, a.l.y, b.l.x, b.l.y);
}

int wino_split_weights(float* U, long chunks, int Cout, hipStream_t st) {
  if (Cout % 32 != 0) return DIM_OK;  // no split kernel runs on such a layer
  const long total = chunks * Cout * 4;
  unsigned char* U3 = reinterpret_cast<unsigned char*>(U + chunks * Cout * 32);
  hipLaunchKernelGGL(wino_split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, U, U3, total, Cout,
                     make_fastdiv((unsigned)Cout));
  return check_launch("winograd_split_weights");
}

#if DIM_SPLIT_EXP & 128
// diagnostic build only: per workgroup (wave 0), cycles per segment of the chunk body summed over its chunks
__device__ unsigned long long g_split_stamps[1024 * 8 * 11];
#endif
#if (DIM_SPLIT_EXP & 128) && defined(__HIP_DEVICE_COMPILE__)
#define S_STAMP(K)                                                                     \
  {                                                                                    \
    unsigned long long t_;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    seg[K] += t_ - tprev;                                                              \
    tprev = t_;                                                                        \
  }
#else
#define S_STAMP(K)
#endif

__device__ __forceinline__ void wscur_advance(WCur& c, const WGemmArgs& a) {
  if (++c.ch == a.nch) {
    c.ch = 0;
    if (++c.nt == a.NTN) {
      c.nt = 0;
      if (++c.mt == a.MT) {
        c.mt = 0;
        ++c.p;
      }
    }
  }
}

template <int BM, int WN>
__global__ __launch_bounds__(WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_gemm_split_kernel(WGemmArgs a) {
  constexpr int BK = 32;
  constexpr int NT = WN * 64;
  constexpr int BN = 32 * WN;
  constexpr int TM = BM / 32;
  constexpr int RP = NT / 8;     // rows staged per pass (8 threads x float4 = one 32-float row)
  constexpr int NSTG = BM / RP;  // staging loads per thread and chunk
  constexpr int PL = BM * 64;    // bytes of one term's image of a chunk
  constexpr int kTrLd = 32, kTrFloats = 16 * kTrLd;
  // cache policy of the wide M stores: non-temporal on the big layers (M = 94-354 MB streams past the caches); plain on the 96-row tile
  // (conv5_1, conv6_1: M = 14-24 MB stays cached for the output transform -- conv5_1's GEMM 66.8 -> 55.5 us, same-box A/B; on the big
  // layers plain stores make the output transform 7 % faster and the next input transform 15 % slower: a wash)
  constexpr int kMAux = ((DIM_SPLIT_OPT & 2) || BM == 96) ? 0 : 2;
  static_assert(BM % 32 == 0 && BM % RP == 0 && RP % 32 == 0 && NSTG >= 2 && NSTG <= 4, "two to four staging loads per thread");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
  unsigned char* sA = smem_split;                                     // [2][3][BM][64 B]
  float* sTr = reinterpret_cast<float*>(smem_split + 2 * 3 * PL);     // [waves][16][32] f32

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = tid & 7;
  const int srow = tid >> 3;
  const int frow = lane & 31;
  const int khalf = lane >> 5;

  const int wg = wg_xcd_contiguous(blockIdx.x, a.G);
  const int c_begin = wg_first_chunk(wg, a);
  const int c_end = wg_first_chunk(wg + 1, a);

  const int RS = a.P * a.K;
  const int a_voff0 = (srow * RS + q * 4) * 4;
  const int a_vstep = RP * RS * 4;
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V), 0, a.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.U3), 0, a.u3_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, a.m_bytes, 0x00020000);
  const int wchunk_bytes = a.Cout * 192;

  WCur L = wcur_decode(c_begin, a), Bc = L, C = L;

  // staging registers of TWO chunks: a chunk's V loads are issued two iterations before its split (HBM latency under load is of the
  // order of one iteration of this kernel)
  float4 rav[2][NSTG];
  // scalar part of a V chunk's loads (cursor L): row-tile base, byte offset, "inside the range"
#define S_LOAD_PREP(PF_OK)                                                             \
  const bool l_pf = (PF_OK);                                                           \
  const int l_mb = L.mt * BM;                                                          \
  const int l_soff = (int)((unsigned)(l_mb * RS + L.p * a.K + L.ch * BK) * 4u);        \
  wscur_advance(L, a);
#define S_LOAD_ISSUE(RSET)                                                             \
  _Pragma("unroll") for (int ps = 0; ps < NSTG; ++ps)                                   \
    rav[RSET][ps] = buf_load16_nt(rv, (l_pf && l_mb + srow + ps * RP < a.T) ? a_voff0 + ps * a_vstep : -1, l_soff);
  // the staging thread's 8 bytes of a row: slot q / 2 swizzled with bits 2-3 of the row (RP is a multiple of 32: the same for every pass)
  const int st_off = srow * 64 + ((((q >> 1) ^ (srow >> 2)) & 3) << 4) + (q & 1) * 8;
#define S_STORE_ONE(DA, R, PASS)                                                       \
  {                                                                                    \
    const Split4 s = split3(R);                                                        \
    *reinterpret_cast<uint2*>((DA) + (PASS) * RP * 64) = s.h;                          \
    *reinterpret_cast<uint2*>((DA) + PL + (PASS) * RP * 64) = s.m;                     \
    *reinterpret_cast<uint2*>((DA) + 2 * PL + (PASS) * RP * 64) = s.l;                 \
  }
#define S_STORE_CHUNK(BUF, RSET)                                                       \
  {                                                                                    \
    unsigned char* dA = sA + (BUF) * 3 * PL + st_off;                                  \
    _Pragma("unroll") for (int ps = 0; ps < NSTG; ++ps) S_STORE_ONE(dA, rav[RSET][ps], ps) \
  }

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // A fragment of k-step s: row frow of tile i, k = 16 s + 8 khalf .. + 7  = slot 2 s + khalf, swizzled
  const int fr_off = frow * 64 + (((khalf ^ (frow >> 2)) & 3) << 4);
  bf16x8 fa[TM][3];  // ONE set: the reads of step 1 go into the registers of step 0 as its MFMAs release them (see the iteration below)
#define S_FRAG_READ(T, PA, S)                                                                                          \
  {                                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                      \
      fa[i][T] = *reinterpret_cast<const bf16x8*>((PA) + (T) * PL + i * 2048 + (fr_off ^ (32 * (S))));                 \
  }
  // B fragments straight from L2, one chunk ahead: fb[set][term][k-step]
  bf16x8 fb[2][3][2];
  const int bf_voff = wave * 6144 + frow * 32 + khalf * 16;
#define S_BFRAG_PREP(VALID)                                                                                            \
  const int b_soff = (VALID) ? (Bc.p * a.nch + Bc.ch) * wchunk_bytes + Bc.nt * BN * 192 : 0;                            \
  wscur_advance(Bc, a);
#define S_BFRAG_ISSUE(SET)                                                                                             \
  _Pragma("unroll") for (int t = 0; t < 3; ++t) _Pragma("unroll") for (int s = 0; s < 2; ++s) {                         \
    const float4 v = buf_load16(ru, bf_voff + (t * 2 + s) * 1024, b_soff);                                             \
    fb[SET][t][s] = __builtin_bit_cast(bf16x8, v);                                                                     \
  }
  // terms of a k-step in issue order (A's term, U3's term; 0 = h, 1 = m, 2 = l): smallest products first; step 0 ends with the
  // terms that read A's h, step 1 begins with those that do not (the step-1 fragments are read into the registers step 0 releases)
  static constexpr int kTA[2][6] = {{2, 0, 1, 1, 0, 0}, {2, 1, 1, 0, 0, 0}};
  static constexpr int kTB[2][6] = {{0, 2, 1, 0, 1, 0}, {0, 1, 0, 2, 1, 0}};
  // the split of one staged float4 in six small steps (dealt one per MFMA below)
  f32x4 sp_x, sp_r;
  bf16x4 sp_h, sp_m, sp_l;

  // ---- output (as in wino_gemm_kernel): D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int ldc = a.P * a.Cout;
  const int o_row = 4 * khalf;
  const int o_col = wave * 32 + frow;
  const int o_voff = (o_row * ldc + o_col) * 4;
  bool whole = C.ch == 0;
#define S_FLUSH_CHECK(KCUR)                                                                                            \
  {                                                                                                                    \
    const bool item_end = C.ch == a.nch - 1;                                                                           \
    if (item_end || (KCUR) == c_end - 1) {                                                                             \
      const int mb = C.mt * BM;                                                                                        \
      const int cb = C.p * a.Cout + C.nt * BN;                                                                         \
      const int lim = a.T - mb;                                                                                        \
      int ov = o_voff, orow = o_row;                                                                                   \
      asm volatile("" : "+v"(ov), "+v"(orow));                                                                         \
      const int soff = (int)((unsigned)(mb * ldc + cb) * 4u);                                                          \
      const bool plain = (whole && item_end) || a.dbg_plain;                                                           \
      if (plain && a.wide_flush) {                                                                                     \
        float* sT = sTr + wave * kTrFloats;                                                                            \
        const int trow = lane >> 3, tc4 = (lane & 7) * 4;                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int h = 0; h < 2; ++h) {                  \
          /* a wave's LDS operations execute in order: the reads see the writes in front of them without a wait */      \
          if constexpr (!(DIM_SPLIT_OPT & 16))                                                                         \
          _Pragma("unroll") for (int r = 0; r < 8; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * khalf) * kTrLd + frow] = acc[i][8 * h + r]; \
          float4 t0, t1;                                                                                               \
          if constexpr (DIM_SPLIT_OPT & 16) { /* timing: no LDS round trip (wrong layout) */                           \
            t0 = make_float4(acc[i][8 * h], acc[i][8 * h + 1], acc[i][8 * h + 2], acc[i][8 * h + 3]);                  \
            t1 = make_float4(acc[i][8 * h + 4], acc[i][8 * h + 5], acc[i][8 * h + 6], acc[i][8 * h + 7]);              \
          } else {                                                                                                     \
            t0 = *reinterpret_cast<const float4*>(sT + trow * kTrLd + tc4);                                            \
            t1 = *reinterpret_cast<const float4*>(sT + (trow + 8) * kTrLd + tc4);                                      \
          }                                                                                                            \
          const int gr = 32 * i + 16 * h + trow;                                                                       \
          const int vo0 = (gr < lim) ? (gr * ldc + wave * 32 + tc4) * 4 : -1;                                          \
          const int vo1 = (gr + 8 < lim) ? ((gr + 8) * ldc + wave * 32 + tc4) * 4 : -1;                                \
          u32x4 u0, u1;                                                                                                \
          u0.x = __float_as_uint(t0.x); u0.y = __float_as_uint(t0.y); u0.z = __float_as_uint(t0.z); u0.w = __float_as_uint(t0.w); \
          u1.x = __float_as_uint(t1.x); u1.y = __float_as_uint(t1.y); u1.z = __float_as_uint(t1.z); u1.w = __float_as_uint(t1.w); \
          if constexpr (DIM_SPLIT_OPT & 8) { /* timing: no global stores */                                            \
            if constexpr (!(DIM_SPLIT_OPT & 64)) asm volatile("" ::"v"(u0), "v"(u1));                                  \
          } else {                                                                                                     \
            __builtin_amdgcn_raw_buffer_store_b128(u0, rm, vo0, soff, kMAux);                                          \
            __builtin_amdgcn_raw_buffer_store_b128(u1, rm, vo1, soff, kMAux);                                          \
          }                                                                                                            \
        }                                                                                                              \
      } else                                                                                                           \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) {                   \
        const int dr = 32 * i + (r & 3) + 8 * (r >> 2);                                                                \
        const int vo = (orow + dr < lim) ? ov + dr * ldc * 4 : -1;                                                     \
        if (plain)                                                                                                     \
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][r]), rm, vo, soff, 2);                          \
        else if (vo != -1)                                                                                             \
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][r], rm, vo, soff, 0);                                 \
      }                                                                                                                \
      if constexpr (!(DIM_SPLIT_OPT & 32))                                                                             \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;    \
      whole = true;                                                                                                    \
    }                                                                                                                  \
    wscur_advance(C, a);                                                                                               \
  }

  // ---- one iteration per chunk of 32 = two k-steps of six MFMA terms, between two workgroup barriers, as ONE scheduling region:
  //   U3 fragments of chunk k+1 and the step-0 A fragments of chunk k are requested, then the 48 MFMAs are issued with the rest dealt
  //   into the gaps between them (an MFMA occupies the matrix pipe for 32 cycles, its issue takes 4):
  //     MFMAs  1-16   the split of chunk k+1 (VALU) and its stores into the other LDS buffer (free since the last barrier)
  //     after 16/24   the step-1 A fragments, into the registers the MFMAs of step 0 have released
  //     MFMAs 25-48   the V loads of chunk k+3 into the registers the split has released
  //   then, at the end of an item, the flush; barrier.
  // In-kernel stamps of the first version (stage and fragment reads between MFMA groups, fenced): a wave spent 1 540 cycles of a chunk
  // issuing MFMAs and 2 400 on everything else, one after the other, two waves per SIMD: 5 800 cycles per chunk against the 3 072 its
  // 96 MFMAs occupy the pipe.
  {
    S_LOAD_PREP(true)
    S_LOAD_ISSUE(0)
  }
  S_STORE_CHUNK(0, 0)
  {
    S_BFRAG_PREP(true)
    S_BFRAG_ISSUE(0)
  }
  {
    S_LOAD_PREP(c_begin + 1 < c_end)
    S_LOAD_ISSUE(1)
  }
  {
    S_LOAD_PREP(c_begin + 2 < c_end)
    S_LOAD_ISSUE(0)
  }
  __syncthreads();

#define S_CHUNK_BODY(SET, KCUR)                                                        \
  {                                                                                    \
    const unsigned char* cA = sA + buf * 3 * PL;                                       \
    unsigned char* dA = sA + (buf ^ 1) * 3 * PL + st_off;                              \
    S_BFRAG_PREP((KCUR) + 1 < c_end)                                                   \
    S_LOAD_PREP((KCUR) + 3 < c_end)                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if constexpr (!(DIM_SPLIT_EXP & 16)) {                                             \
      S_FRAG_READ(2, cA, 0)                                                            \
      S_FRAG_READ(0, cA, 0)                                                            \
      S_FRAG_READ(1, cA, 0)                                                            \
    }                                                                                  \
    if constexpr (!(DIM_SPLIT_EXP & 4)) S_BFRAG_ISSUE(1 - SET)                         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    static_for<12 * TM>([&](auto N_) {                                                 \
      constexpr int n = decltype(N_)::value;                                           \
      constexpr int ks = n / (6 * TM), ti = (n % (6 * TM)) / TM, i = n % TM;           \
      constexpr int TA = kTA[ks][ti], TB = kTB[ks][ti];                                \
      if constexpr (!(DIM_SPLIT_EXP & 1))                                              \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA], fb[SET][TB][ks], acc[i], 0, 0, 0); \
      else {                                                                           \
        asm volatile("" ::"v"(fa[i][TA]));                                             \
        asm volatile("" ::"v"(fb[SET][TB][ks]));                                       \
      }                                                                                \
      /* behind the MFMA: one step of the split of chunk k+1 ... */                    \
      if constexpr (n < 6 * NSTG && !(DIM_SPLIT_EXP & 8)) {                            \
        constexpr int ps = n / 6, sub = n % 6;                                         \
        if constexpr (sub == 0) {                                                      \
          const float4 v = rav[1 - SET][ps];                                           \
          sp_x = f32x4{v.x, v.y, v.z, v.w};                                            \
          sp_h = __builtin_convertvector(sp_x, bf16x4);                                \
        } else if constexpr (sub == 1) {                                               \
          sp_r = sp_x - __builtin_convertvector(sp_h, f32x4);                          \
        } else if constexpr (sub == 2) {                                               \
          sp_m = __builtin_convertvector(sp_r, bf16x4);                                \
        } else if constexpr (sub == 3) {                                               \
          sp_r = sp_r - __builtin_convertvector(sp_m, f32x4);                          \
        } else if constexpr (sub == 4) {                                               \
          sp_l = __builtin_convertvector(sp_r, bf16x4);                                \
        } else {                                                                       \
          *reinterpret_cast<uint2*>(dA + ps * RP * 64) = __builtin_bit_cast(uint2, sp_h);          \
          *reinterpret_cast<uint2*>(dA + PL + ps * RP * 64) = __builtin_bit_cast(uint2, sp_m);     \
          *reinterpret_cast<uint2*>(dA + 2 * PL + ps * RP * 64) = __builtin_bit_cast(uint2, sp_l); \
        }                                                                              \
      }                                                                                \
      /* ... the step-1 fragment of a tile into the registers step 0 has just released ... */ \
      if constexpr (!(DIM_SPLIT_EXP & 16)) {                                           \
        if constexpr (n >= 2 * TM && n < 3 * TM)                                       \
          fa[n - 2 * TM][2] = *reinterpret_cast<const bf16x8*>(cA + 2 * PL + (n - 2 * TM) * 2048 + (fr_off ^ 32)); \
        if constexpr (n >= 3 * TM && n < 4 * TM)                                       \
          fa[n - 3 * TM][1] = *reinterpret_cast<const bf16x8*>(cA + PL + (n - 3 * TM) * 2048 + (fr_off ^ 32));     \
        if constexpr (n >= 5 * TM && n < 6 * TM)                                       \
          fa[n - 5 * TM][0] = *reinterpret_cast<const bf16x8*>(cA + (n - 5 * TM) * 2048 + (fr_off ^ 32));          \
      }                                                                                \
      /* ... one V load of chunk k+3 into the staging registers the split has read */  \
      if constexpr (n >= 6 * TM && (n - 6 * TM) % 2 == 0 && (n - 6 * TM) / 2 < NSTG && !(DIM_SPLIT_EXP & 2)) { \
        constexpr int ps = (n - 6 * TM) / 2;                                           \
        rav[1 - SET][ps] = buf_load16_nt(rv, (l_pf && l_mb + srow + ps * RP < a.T) ? a_voff0 + ps * a_vstep : -1, l_soff); \
      }                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                               \
    });                                                                                \
    S_STAMP(0)                                                                         \
    if constexpr (!(DIM_SPLIT_EXP & 64)) S_FLUSH_CHECK(KCUR) else wscur_advance(C, a); \
    S_STAMP(1)                                                                         \
    if constexpr (!(DIM_SPLIT_EXP & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    S_STAMP(2)                                                                         \
    buf ^= 1;                                                                          \
  }
  int buf = 0;
#if (DIM_SPLIT_EXP & 128) && defined(__HIP_DEVICE_COMPILE__)
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev, tstart;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
  tstart = tprev;
#endif
  for (int kc = c_begin; kc < c_end; kc += 2) {
    S_CHUNK_BODY(0, kc)
    if (kc + 1 < c_end) S_CHUNK_BODY(1, kc + 1)
  }
#if (DIM_SPLIT_EXP & 64) && defined(__HIP_DEVICE_COMPILE__)
  _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(acc[i]));   // the accumulators stay live without the flush
#endif
#if (DIM_SPLIT_EXP & 128) && defined(__HIP_DEVICE_COMPILE__)
  if (lane == 0 && blockIdx.x < 1024 && wave < 8) {
    unsigned long long* o = g_split_stamps + (blockIdx.x * 8 + wave) * 11;
    for (int k = 0; k < 8; ++k) o[k] = seg[k];
    o[8] = tprev - tstart;
    o[9] = (unsigned long long)(c_end - c_begin);
    o[10] = 1;
  }
#endif
#undef S_CHUNK_BODY
#undef S_FLUSH_CHECK
#undef S_BFRAG_ISSUE
#undef S_BFRAG_PREP
#undef S_FRAG_READ
#undef S_STORE_CHUNK
#undef S_STORE_ONE
#undef S_LOAD_ISSUE
#undef S_LOAD_PREP
}

template <int BM, int WN>
static constexpr size_t split_lds_bytes() {
  return 2 * 3 * BM * 64 + WN * 16 * 32 * sizeof(float);
}

template <int BM, int WN>
static int wino_gemm_split_slots_t() {
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0, occ = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&wino_gemm_split_kernel<BM, WN>), WN * 64,
                                                       split_lds_bytes<BM, WN>());
    if (e != hipSuccess || cus <= 0 || occ <= 0) return set_err(DIM_ERR_LAUNCH, "winograd split gemm occupancy query: %s", hipGetErrorString(e));
    slots = cus * occ;
  }
  return slots;
}

// the split kernel that stands in for tile `tile` of wino_gemm_plan (same BM x BN, so the zeroing of shared tiles is unchanged); 0 = none
bool wino_gemm_split_has(int tile) { return tile == 5 || tile == 4 || tile == 7; }

int wino_gemm_split_slots(int tile) {
  return tile == 5 ? wino_gemm_split_slots_t<128, 8>() : tile == 4 ? wino_gemm_split_slots_t<128, 4>() : wino_gemm_split_slots_t<96, 4>();
}

template <int BM, int WN>
static int wino_gemm_split_launch(const WGemmArgs& a, hipStream_t st) {
  constexpr size_t lds = split_lds_bytes<BM, WN>();
  static_assert(lds <= 65536, "a workgroup's dynamic LDS stays within 64 KB (see wino_gemm.hip)");
  hipLaunchKernelGGL((wino_gemm_split_kernel<BM, WN>), dim3(a.G), dim3(WN * 64), lds, st, a);
  return check_launch("winograd_gemm_split");
}

#if DIM_SPLIT_EXP & 128
extern "C" int dim_debug_split_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_split_stamps), sizeof(unsigned long long) * 11 * 8 * (n < 1024 ? n : 1024));
}
#endif

int wino_gemm_split_run(const WGemmArgs& a, hipStream_t st) {
  if (a.tile == 5) return wino_gemm_split_launch<128, 8>(a, st);
  if (a.tile == 4) return wino_gemm_split_launch<128, 4>(a, st);
  return wino_gemm_split_launch<96, 4>(a, st);
}

}  // namespace dim
