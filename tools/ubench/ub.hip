// Issue-cost micro-benchmarks for gfx950 (one workgroup of 8 waves on one CU, s_memtime per wave).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC ub.hip -o libub.so     Run: python tools/ubench/run.py
// Numbers feed the hand schedule of the GEMM K loop: cycles per MFMA alone / with a second wave
// on the SIMD, per s_barrier, per buffer_load ... lds issue (alone and between MFMAs), ds_read_b128 round trip.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// test ids: 0 mfma (waves < nw active), 1 barrier, 2 dma issue, 3 dma issue between mfma, 4 ds_read_b128 x8 round trip,
//           5 ping-pong skeleton (8 mfma | barrier | barrier per half phase, two groups)
__global__ __launch_bounds__(512) void ub_kernel(int test, int nw, const char* gsrc, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, 0x7fffffff, 0x00020000);
  const uint32_t voff = (uint32_t)lane * 16u;
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  constexpr int REP = 64;
  __syncthreads();
  if (test == 0) {
    if (w < nw) {
      t0 = now();
      for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
      t1 = now();
    }
  } else if (test == 1) {
    t0 = now();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_barrier();
    }
    t1 = now();
  } else if (test == 2) {
    if (w < nw) {
      t0 = now();
      for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (w * 8 + i) * 1024), 16, voff, ((r * 8 + i) & 63) * 1024 + w * 65536, 0, 0);
      }
      t1 = now();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t2 = now();
    }
  } else if (test == 3) {
    if (w < nw) {
      t0 = now();
      for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
          if ((i & 3) == 3)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (w * 8 + i) * 1024), 16, voff, ((r * 8 + i) & 63) * 1024 + w * 65536, 0, 0);
        }
      }
      t1 = now();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t2 = now();
    }
  } else if (test == 4) {
    if (w < nw) {
      bf16x8 f[8];
      t0 = now();
      for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = *(const bf16x8*)(smem + ((i * 32 + (lane & 31)) * 128 + ((((lane >> 5) + 2 * (r & 3)) ^ ((lane >> 1) & 7)) << 4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) a[0] = (__bf16)((float)a[0] + (float)f[i][0]);
      }
      t1 = now();
    }
  } else if (test == 5) {
    const int grp = w >> 2;
    if (grp == 1) __builtin_amdgcn_s_barrier();
    t0 = now();
    for (int r = 0; r < REP; ++r) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    t1 = now();
    if (grp == 0) __builtin_amdgcn_s_barrier();
  }
  else if (test >= 6 && test <= 10) {
    // LDS port sharing: waves 0..3 stream conflict-free ds_read_b128 (tests 6, 8, 10); waves 4..7 stream LDS-DMA pieces
    // (tests 7, 8) or ds_write_b128 (tests 9, 10).  Alone vs together tells whether DMA / ds_write and ds_read share the port.
    const bool reader = w < 4 && (test == 6 || test == 8 || test == 10);
    const bool dma = w >= 4 && (test == 7 || test == 8);
    const bool writer = w >= 4 && (test == 9 || test == 10);
    const char* rbase = smem + ((lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4)) + (w & 3) * 4096;
    bf16x8 f[8];
    __syncthreads();
    t0 = now();
    if (reader) {
      for (int r = 0; r < REP * 4; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = *(const bf16x8*)(rbase + i * 16384 + (((r & 3) * 2) << 4) * 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) a[0] = (__bf16)((float)a[0] + (float)f[i][0]);
      }
    } else if (dma) {
      for (int r = 0; r < REP * 4; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 65536 + ((w - 4) * 8 + i) * 1024), 16, voff, ((r * 8 + i) & 63) * 1024 + w * 65536, 0, 0);
        if ((r & 1) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (writer) {
      for (int r = 0; r < REP * 4; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(bf16x8*)(smem + 65536 + ((w - 4) * 8 + i) * 1024 + lane * 16) = a;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    t1 = now();
  }
  float s = (float)a[0];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
  if (s == 123.456f) sink[tid] = s;
  if (lane == 0 && blockIdx.x == 0) {
    out[w * 2] = t1 - t0;
    out[w * 2 + 1] = t2 > t1 ? t2 - t1 : 0;
  }
}

extern "C" int ub_run(int test, int nw, int blocks, const void* gsrc, unsigned long long* out, float* sink, void* stream) {
  hipFuncSetAttribute((const void*)ub_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL(ub_kernel, dim3(blocks), dim3(512), 128 * 1024, (hipStream_t)stream, test, nw, (const char*)gsrc, out, sink);
  return (int)hipGetLastError();
}
