// Micro-benchmark: what one "group" of the attention kernel's first half costs on gfx950 -- 4 x v_mfma_f32_16x16x32_bf16 on two
// accumulators (two dependent pairs, the PV product of one d tile) with, per variant, the VALU / LDS work the kernel puts beside them.
// One workgroup per CU, 256 or 512 threads (one or two waves per SIMD); clocks per group from s_memtime of wave 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int VAR>
__global__ __launch_bounds__(512) void group_kernel(int iters, float* sink, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x4 acc[8][2];
  for (int d = 0; d < 8; ++d) for (int q = 0; q < 2; ++q) for (int r = 0; r < 4; ++r) acc[d][q][r] = 0.f;
  bf16x8 a[2], b[4];
  for (int j = 0; j < 8; ++j) { a[0][j] = (__bf16)(0.01f * (lane + j)); a[1][j] = (__bf16)(0.02f * (lane - j)); }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.003f * (lane + i + j));
  float s[32];
  for (int i = 0; i < 32; ++i) s[i] = -0.01f * (lane + i);
  float psum[2] = {0.f, 0.f};
  const char* rd = lds + (lane & 15) * 256 + ((lane >> 4) ^ (lane & 15)) * 16;
  unsigned long long c0 = 0, c1 = 0;
  if (blockIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int G = 0; G < 8; ++G) {
      if constexpr ((VAR & 8) != 0) {   // 2 fragment reads per group, as in the kernel (one group ahead)
        a[0] = *(const bf16x8*)(rd + G * 4096);
        a[1] = *(const bf16x8*)(rd + G * 4096 + 2048);
      }
      if constexpr ((VAR & 16) == 0) {   // (16: no MFMAs -- what the VALU work costs alone)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[G][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[bb], b[bb * 2 + q], acc[G][q], 0, 0, 0);
      }
      if constexpr ((VAR & 1) != 0) {   // 4 exp2
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(s[G * 4 + j]);
        asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[G * 4 + j] = e[j];
          if constexpr ((VAR & 2) != 0) psum[G & 1] += e[j];   // + 4 dependent adds
        }
      }
      if constexpr ((VAR & 4) != 0) {   // 2 cvt_pk (the pack of the second half)
        uint32_t w0, w1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=v"(w0), "=v"(w1) : "v"(s[G * 4]), "v"(s[G * 4 + 1]), "v"(s[G * 4 + 2]), "v"(s[G * 4 + 3]));
        b[G & 3][0] = __builtin_bit_cast(__bf16, (uint16_t)(w0 ^ w1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((VAR & 1) != 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) s[i] = s[i] * 0.5f - 1.0f;   // keep the exp inputs bounded (8 x amortised)
    }
  }
  if (blockIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
  float t = psum[0] + psum[1];
  for (int d = 0; d < 8; ++d) for (int q = 0; q < 2; ++q) t += acc[d][q][0] + acc[d][q][3];
  for (int i = 0; i < 32; ++i) t += s[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; }
}

extern "C" int attn_group_run(int var, int threads, int blocks, int iters, float* sink, unsigned long long* clk, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define RUN(V) case V: hipLaunchKernelGGL(group_kernel<V>, dim3(blocks), dim3(threads), 0, st, iters, sink, clk); break;
  switch (var) {
    RUN(0) RUN(1) RUN(3) RUN(4) RUN(8) RUN(9) RUN(11) RUN(12) RUN(2) RUN(17) RUN(19) RUN(20)
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
