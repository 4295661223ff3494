// Shared device/host helpers for librf_flux (gfx950 only -- no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "rf_flux_debug.h"   // includes rf_flux.h (the product ABI); the profiling classes live in the debug header

namespace rf {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// buffer-resource LDS-DMA (buffer_load_dwordx4 ... lds).  The resource type and builtins only exist in the device
// pass; the host pass (which still instantiates the kernel templates to take their addresses) sees inert stand-ins.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define RF_MAKE_RSRC(p) __builtin_amdgcn_make_buffer_rsrc((void*)(p), 0, 0x7fffffff, 0x00020000)
#define RF_BUF_LOAD_LDS(r, lds, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, voff, soff, 0, 0)
#else
typedef int rsrc_t;
#define RF_MAKE_RSRC(p) 0
#define RF_BUF_LOAD_LDS(r, lds, voff, soff) ((void)0)
#endif


// ---- error plumbing (thread-local message, never throws across the C ABI) -------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define RF_CHECK_HIP(expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return ::rf::hip_fail(_e, #expr);  \
  } while (0)

#define RF_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      ::rf::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

#define RF_LAUNCH_CHECK() RF_CHECK_HIP(hipGetLastError())

// kernel-class timing hook (rf_profile_begin / rf_profile_end): a no-op unless a profile is open
struct ProfScope {
  ProfScope(int cls, double work, hipStream_t s);
  ~ProfScope();
  void reclass(int cls);   // the launch turned out to belong to another class (the event stays where it was recorded)
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
 private:
  int idx_;
  hipStream_t s_;
};

bool prof_open();   // capi.hip: a kernel-class profile (rf_profile_begin) is open -> kernels also store their clock probes

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---- bf16 <-> f32 (round-to-nearest-even, the same rounding torch uses) ---------------
__device__ __forceinline__ float bf2f(bf16_t x) { return static_cast<float>(x); }
__device__ __forceinline__ bf16_t f2bf(float x) { return static_cast<bf16_t>(x); }

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

// unpack 8 bf16 (held as 4 x u32) to 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  bf16x2 t;
  t[0] = static_cast<bf16_t>(lo);
  t[1] = static_cast<bf16_t>(hi);
  return __builtin_bit_cast(uint32_t, t);
}

__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // F.gelu(approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  With 0.5 (1 + tanh(u)) =
  // sigmoid(2u) this is x / (1 + exp(-2u)): one v_exp_f32 and one v_rcp_f32 (1 ulp) per element -- an IEEE division
  // here compiled to a 10-instruction v_div_scale/fmas/fixup sequence in every GELU epilogue lane.
  // exp overflow -> inf -> rcp = 0 -> x * 0 (the correct limit for x -> -inf).
  const float z = x * (-2.3022082f - 0.1029432f * x * x);  // -2u * log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}

// XCD-aware, bijective block-id remap (8 XCDs, block b runs on XCD b % 8): give each XCD a
// contiguous chunk of the logical tile space so neighbouring tiles share its private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  const int xcd = bid % nx, idx = bid / nx;
  const int q = nwg / nx, r = nwg % nx;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// Shader-clock probe: block 0 brackets its work with {s_memtime (shader clocks), s_memrealtime (100 MHz)} and stores the four
// counters; rf_debug_clock_probe() turns them into the clock the kernel actually ran at.  MI355X is power-limited under
// dense MFMA + LDS + L2 traffic (profiles/r02_kb_ppx_v4_clock.log): the sustained clock, not 2.4 GHz, prices a kernel.
struct ClkProbe {
  unsigned long long c0 = 0, r0 = 0;
  __device__ __forceinline__ void begin() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (blockIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
#endif
  }
  __device__ __forceinline__ void end(unsigned long long* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      unsigned long long c1, r1;
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
      dst[0] = c0; dst[1] = r0; dst[2] = c1; dst[3] = r1;
    }
#endif
  }
};
int read_clk_probe_gemm(unsigned long long* h);   // gemm_bf16.hip
int read_clk_probe_attn(unsigned long long* h);   // attention.hip

// LayerNorm + modulate of up to three token streams (text / image / condition) in ONE launch (rowops.hip; the engine's AdaLN stages).
// Row for row the arithmetic of rf_layernorm_modulate: bit-identical outputs.
struct LnModStream {
  const bf16_t* x; bf16_t* out; const bf16_t* scale; const bf16_t* shift; int rows;
};
int ln_mod_grouped(const LnModStream* streams, int n, int64_t ldx, int64_t ldo, int D, float eps, hipStream_t stream);

}  // namespace rf
