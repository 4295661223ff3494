// Sustained MFMA rate under the power cap with NON-ZERO operands: v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16
// (same FLOPs per cycle on paper; different operand / accumulator register traffic).  No memory traffic at all.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_power.hip -o libmfmapower.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ bf16x8 rnd_frag(uint32_t seed) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    seed = seed * 1664525u + 1013904223u;
    v[j] = (__bf16)(((int)(seed >> 9) % 2048 - 1024) * (1.0f / 1024.0f));
  }
  return v;
}

template <int KIND>
__global__ __launch_bounds__(512) void mfma_power_kernel(int iters, float* sink, unsigned long long* clk) {
  const int tid = threadIdx.x + blockIdx.x * 512;
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = rnd_frag(tid * 8 + i); b[i] = rnd_frag(tid * 8 + 4 + i); }
  float s = 0.f;
  if (KIND >= 2) {
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    i32x8 a8[4], b8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // random e4m3 bytes with the exponent's top bit clear (|x| < 2: finite, no NaN pattern)
        uint32_t x = (uint32_t)(tid * 64 + i * 8 + j) * 2654435761u, y = x * 1664525u + 1013904223u;
        a8[i][j] = (int)(x & 0xbfbfbfbfu);
        b8[i][j] = (int)(y & 0xbfbfbfbfu);
      }
    if (KIND == 2) {
      f32x16 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 3], b8[(i >> 1) & 3], acc[i], 0, 0, 0, 127, 0, 127);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += acc[i][tid & 15];
    } else {
      f32x4 acc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 3], b8[(i >> 2) & 3], acc[i], 0, 0, 0, 127, 0, 127);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[i][tid & 3];
    }
  } else if (KIND == 0) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][tid & 15];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][tid & 3];
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  if (s == 123.456f) sink[tid & 511] = s;
  if (tid == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

extern "C" int mfma_power_run(int kind, int blocks, int iters, float* sink, unsigned long long* clk, void* stream) {
  if (kind == 0) hipLaunchKernelGGL(mfma_power_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink, clk);
  else if (kind == 1) hipLaunchKernelGGL(mfma_power_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink, clk);
  else if (kind == 2) hipLaunchKernelGGL(mfma_power_kernel<2>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink, clk);
  else hipLaunchKernelGGL(mfma_power_kernel<3>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink, clk);
  return (int)hipGetLastError();
}
