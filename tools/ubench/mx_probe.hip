// Operand-layout probe for the gfx950 block-scaled MFMAs (v_mfma_scale_f32_{32x32x64,16x16x128}_f8f6f4, fp8 e4m3 x
// fp8 e4m3).  One wave; every lane's 8 operand VGPRs and its scale VGPRs come from memory, D goes back to memory;
// tools/ubench/mx_probe.py tries candidate (lane, byte) -> (row, k) maps against a float reference.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC mx_probe.hip -o libmxprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OPA, int OPB>
__global__ void mx_probe_kernel(const i32x8* a, const i32x8* b, float* d32, float* d16, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 0, 0, OPA, sa[l], OPB, sb[l]);
#pragma unroll
  for (int r = 0; r < 16; ++r) d32[l * 16 + r] = c[r];
  f32x4 c2 = {};
  c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c2, 0, 0, OPA, sa[l], OPB, sb[l]);
#pragma unroll
  for (int r = 0; r < 4; ++r) d16[l * 4 + r] = c2[r];
}

extern "C" int mx_probe(const void* a, const void* b, float* d32, float* d16, const int* sa, const int* sb, int opa, int opb) {
  if (opa == 0 && opb == 0) hipLaunchKernelGGL((mx_probe_kernel<0, 0>), dim3(1), dim3(64), 0, 0, (const i32x8*)a, (const i32x8*)b, d32, d16, sa, sb);
  else if (opa == 1 && opb == 2) hipLaunchKernelGGL((mx_probe_kernel<1, 2>), dim3(1), dim3(64), 0, 0, (const i32x8*)a, (const i32x8*)b, d32, d16, sa, sb);
  else return -1;
  return (int)hipDeviceSynchronize();
}
