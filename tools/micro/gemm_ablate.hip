// ablation of the conv main loop: which stage costs the MFMA pipe its idle time?
// flags: bit0 = LDS fragment reads, bit1 = LDS stores + barrier per chunk, bit2 = global loads per chunk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FLAGS, int TM, int TN>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, const float* __restrict__ B, float* out, int nchunks, int lda) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDK = 36;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + 2 * BM * LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int q = tid & 7, srow = tid >> 3;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[2 * TM], rb[2 * TN], ra2[2 * TM], rb2[2 * TN];
  for (int i = 0; i < 2 * TM; ++i) ra[i] = ra2[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int i = 0; i < 2 * TN; ++i) rb[i] = rb2[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  // init LDS
  for (int i = tid; i < 2 * (BM + BN) * LDK; i += 256) smem[i] = 1.0f + (i & 7);
  __syncthreads();
  const float* ga = A + ((long)blockIdx.x * BM + srow) * lda + q * 4;
  const float* gb = B + ((long)blockIdx.y * BN + srow) * 32 + q * 4;
  const int a_off = (wm * 32 * TM + (lane & 31)) * LDK + 4 * (lane >> 5);
  const int b_off = (wn * 32 * TN + (lane & 31)) * LDK + 4 * (lane >> 5);
  // FLAGS & 16: B fragments straight from global memory (packed [chunk][N][32]) into registers, no LDS staging for B
  const float* gbf = B + ((long)blockIdx.y * BN + wn * 32 * TN + (lane & 31)) * 32 + 4 * (lane >> 5);
  const long bchunk = (long)32 * 64 * TN * gridDim.y;
  float4 fbd[4][TN], fbn[4][TN];
  for (int s4 = 0; s4 < 4; ++s4)
    for (int j = 0; j < TN; ++j) fbd[s4][j] = fbn[s4][j] = make_float4(1.f, 1.f, 1.f, 1.f);
  int buf = 0;
  float4 fa[2][TM], fb[2][TN];
  for (int i = 0; i < TM; ++i) fa[0][i] = fa[1][i] = make_float4(1.f, 1.f, 1.f, 1.f);
  for (int j = 0; j < TN; ++j) fb[0][j] = fb[1][j] = make_float4(1.f, 1.f, 1.f, 1.f);
  for (int kc = 0; kc < nchunks; ++kc) {
    const float* cA = sA + buf * BM * LDK + a_off;
    const float* cB = sB + buf * BN * LDK + b_off;
    if (FLAGS & 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const float4*>(cA + 32 * i * LDK);
      if (!(FLAGS & 16)) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const float4*>(cB + 32 * j * LDK);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (FLAGS & 16) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < TN; ++j) fbn[s4][j] = *reinterpret_cast<const float4*>(gbf + (long)(kc + 1 < nchunks ? kc + 1 : kc) * bchunk + (long)j * 32 * 32 + 8 * s4);
    }
    if (FLAGS & 8) {  // two-deep: what was loaded one chunk ago moves to the store registers, new loads go out now
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) ra[i] = ra2[i];
#pragma unroll
      for (int i = 0; i < 2 * TN; ++i) rb[i] = rb2[i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) ra2[i] = *reinterpret_cast<const float4*>(ga + (long)i * 32 * lda + kc * 32);
#pragma unroll
      for (int i = 0; i < 2 * TN; ++i) rb2[i] = *reinterpret_cast<const float4*>(gb + (long)i * 32 * 32 + (long)kc * 32 * 64 * TN * gridDim.y);
    } else if (FLAGS & 4) {
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) ra[i] = *reinterpret_cast<const float4*>(ga + (long)i * 32 * lda + kc * 32);
#pragma unroll
      for (int i = 0; i < 2 * TN; ++i) rb[i] = *reinterpret_cast<const float4*>(gb + (long)i * 32 * 32 + (long)kc * 32 * 64 * TN * gridDim.y);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if ((FLAGS & 1) && s + 1 < 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const float4*>(cA + 32 * i * LDK + 8 * (s + 1));
        if (!(FLAGS & 16)) {
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const float4*>(cB + 32 * j * LDK + 8 * (s + 1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float4 bq = (FLAGS & 16) ? fbd[s][j] : fb[cur][j];
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, bq.x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, bq.y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, bq.z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, bq.w, acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if ((FLAGS & 2) && s == 2) {
        float* dA = sA + (buf ^ 1) * BM * LDK;
        float* dB = sB + (buf ^ 1) * BN * LDK;
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) *reinterpret_cast<float4*>(dA + (srow + 32 * i) * LDK + q * 4) = ra[i];
        if (!(FLAGS & 16)) {
#pragma unroll
          for (int i = 0; i < 2 * TN; ++i) *reinterpret_cast<float4*>(dB + (srow + 32 * i) * LDK + q * 4) = rb[i];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (FLAGS & 2) {
      __syncthreads();
      buf ^= 1;
    }
    if (FLAGS & 16) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < TN; ++j) fbd[s4][j] = fbn[s4][j];
    }
  }
  float sacc = 0.f;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
  out[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = sacc;
}

template <int FLAGS, int TM, int TN>
void run(int mt, int nt, int nchunks) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  size_t lds = (size_t)2 * (BM + BN) * 36 * 4;
  int lda = nchunks * 32;
  float *A, *B, *out;
  (void)hipMalloc(&A, (size_t)mt * BM * lda * 4);
  (void)hipMalloc(&B, (size_t)nt * BN * lda * 4);
  (void)hipMalloc(&out, (size_t)mt * nt * 256 * 4);
  (void)hipMemset(A, 0, (size_t)mt * BM * lda * 4);
  (void)hipMemset(B, 0, (size_t)nt * BN * lda * 4);
  (void)hipFuncSetAttribute((const void*)k<FLAGS, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<FLAGS, TM, TN><<<dim3(mt, nt), 256, lds>>>(A, B, out, nchunks, lda);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<FLAGS, TM, TN><<<dim3(mt, nt), 256, lds>>>(A, B, out, nchunks, lda);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = 2.0 * mt * BM * nt * BN * nchunks * 32;
  printf("tile %dx%d flags=%d (lds_read=%d store+barrier=%d global=%d) grid %dx%d chunks %d: %.3f ms %.1f TF\n", BM, BN, FLAGS, FLAGS & 1,
         (FLAGS >> 1) & 1, (FLAGS >> 2) & 1, mt, nt, nchunks, ms, flops / ms / 1e9);
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(out);
}

int main() {
  // conv3_1-like: M=76800, N=256, K=2304
  run<0, 1, 1>(1200, 4, 72); run<1, 1, 1>(1200, 4, 72); run<3, 1, 1>(1200, 4, 72); run<7, 1, 1>(1200, 4, 72); run<15, 1, 1>(1200, 4, 72); run<23, 1, 1>(1200, 4, 72); run<31, 1, 1>(1200, 4, 72);
  run<0, 2, 2>(600, 2, 72); run<1, 2, 2>(600, 2, 72); run<3, 2, 2>(600, 2, 72); run<7, 2, 2>(600, 2, 72); run<15, 2, 2>(600, 2, 72); run<23, 2, 2>(600, 2, 72); run<31, 2, 2>(600, 2, 72);
  return 0;
}
