// Bandwidth-bound row kernels for gfx950: 16-byte vector loads/stores, one 64-lane wave per
// row (or per 4 head-rows), fp32 math, a single bf16 rounding on the way out.
//   rf_layernorm_modulate : LayerNorm(no affine) + (1+scale)*x + shift   (AdaLN-Zero family)
//   rf_qk_rmsnorm_rope    : per-head RMSNorm(q,k) + interleaved-pair RoPE, in place
//   rf_euler_step / rf_silu / rf_add_inplace : element-wise
#include "common.hpp"

namespace rf {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- LayerNorm + modulate ---------------------------------------------------------------------
// One wave per row; the row lives in registers (NCH chunks of 8 bf16 per lane, D <= NCH*512).
template <int NCH>
__global__ __launch_bounds__(256) void ln_mod_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                     bf16_t* __restrict__ out, int64_t ldo, int rows, int D,
                                                     const bf16_t* __restrict__ scale,
                                                     const bf16_t* __restrict__ shift, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      const u32x4 raw = *(const u32x4*)(xr + col);
      unpack8(raw, v[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dlt = v[c][j] - mean;
        q += dlt * dlt;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  bf16_t* orow = out + (int64_t)row * ldo;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      float sc[8], sh[8], o[8];
      unpack8(*(const u32x4*)(scale + col), sc);
      unpack8(*(const u32x4*)(shift + col), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * (1.0f + sc[j]) + sh[j];
      *(u32x4*)(orow + col) = pack8(o);
    }
  }
}

// ---- per-head RMSNorm(q,k) + RoPE, in place on [heads][s_pad][128] ------------------------------
// 16 lanes x 8 elements cover one 128-wide head row; a wave handles 4 rows per iteration.
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ q, bf16_t* __restrict__ k,
                                                           int heads, int S, int s_pad, int n_added,
                                                           const bf16_t* __restrict__ wq,
                                                           const bf16_t* __restrict__ wk,
                                                           const bf16_t* __restrict__ waq,
                                                           const bf16_t* __restrict__ wak,
                                                           const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, float eps) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4;          // which of the 4 rows of this wave
  const int c8 = (lane & 15) * 8;     // first of this lane's 8 columns
  const int64_t total = (int64_t)2 * heads * S;  // q rows then k rows
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r4 = wave_id * 4; r4 < total; r4 += nwaves * 4) {
    const int64_t r = r4 + sub;
    if (r < total) {
      const int is_k = r >= (int64_t)heads * S;
      const int64_t rr = is_k ? r - (int64_t)heads * S : r;
      const int head = (int)(rr / S);
      const int tok = (int)(rr - (int64_t)head * S);
      bf16_t* ptr = (is_k ? k : q) + ((int64_t)head * s_pad + tok) * 128 + c8;
      const bf16_t* wsel = tok < n_added ? (is_k ? wak : waq) : (is_k ? wk : wq);
      float v[8], wv[8];
      unpack8(*(const u32x4*)ptr, v);
      unpack8(*(const u32x4*)(wsel + c8), wv);
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);   // within the 16-lane row group
      const float rs = rsqrtf(ss * (1.0f / 128.0f) + eps);
      const f32x4 c0 = *(const f32x4*)(cos_tab + (int64_t)tok * 128 + c8);
      const f32x4 c1 = *(const f32x4*)(cos_tab + (int64_t)tok * 128 + c8 + 4);
      const f32x4 s0 = *(const f32x4*)(sin_tab + (int64_t)tok * 128 + c8);
      const f32x4 s1 = *(const f32x4*)(sin_tab + (int64_t)tok * 128 + c8 + 4);
      float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float a = v[j] * rs * wv[j];
        const float b = v[j + 1] * rs * wv[j + 1];
        // apply_rotary_emb: out = x*cos + rot(x)*sin, rot(x)[2i] = -x[2i+1], rot(x)[2i+1] = x[2i]
        o[j] = a * cs[j] - b * sn[j];
        o[j + 1] = b * cs[j + 1] + a * sn[j + 1];
      }
      *(u32x4*)ptr = pack8(o);
    }
  }
}

__global__ __launch_bounds__(256) void euler_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ v,
                                                    int64_t n, float dt) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float a[8], b[8];
    unpack8(*(const u32x4*)(x + i), a);
    unpack8(*(const u32x4*)(v + i), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = a[j] + dt * b[j];
    *(u32x4*)(x + i) = pack8(a);
  } else {
    for (int64_t j = i; j < n; ++j) x[j] = f2bf(bf2f(x[j]) + dt * bf2f(v[j]));
  }
}

__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float a = bf2f(x[i]);
    out[i] = f2bf(a / (1.0f + __expf(-a)));
  }
}

__global__ __launch_bounds__(256) void add_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ x, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float a[8], b[8];
    unpack8(*(const u32x4*)(out + i), a);
    unpack8(*(const u32x4*)(x + i), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    *(u32x4*)(out + i) = pack8(a);
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = f2bf(bf2f(out[j]) + bf2f(x[j]));
  }
}

}  // namespace rf

extern "C" int rf_layernorm_modulate(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t rows, int32_t D,
                                     const void* scale, const void* shift, float eps, void* stream) {
  using namespace rf;
  if (rows <= 0) return RF_OK;
  RF_REQUIRE(x && out && scale && shift, RF_ERR_NULL, "rf_layernorm_modulate: NULL pointer");
  RF_REQUIRE(D > 0 && D % 8 == 0 && D <= 8 * 512, RF_ERR_SHAPE, "rf_layernorm_modulate: D=%d (need D%%8==0, D<=4096)", D);
  RF_REQUIRE(aligned16(x) && aligned16(out) && aligned16(scale) && aligned16(shift) && ldx % 8 == 0 && ldo % 8 == 0,
             RF_ERR_ALIGN, "rf_layernorm_modulate: operands must be 16-byte aligned");
  const dim3 grid(cdiv(rows, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, 4.0 * rows * (double)D, s);  // bytes: one bf16 row in, one out
  const int nch = cdiv(D, 512);
#define RF_LN_CASE(N)                                                                                          \
  case N:                                                                                                      \
    hipLaunchKernelGGL(ln_mod_kernel<N>, grid, block, 0, s, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, rows, D, \
                       (const bf16_t*)scale, (const bf16_t*)shift, eps);                                       \
    break;
  switch (nch) {
    RF_LN_CASE(1) RF_LN_CASE(2) RF_LN_CASE(3) RF_LN_CASE(4) RF_LN_CASE(5) RF_LN_CASE(6) RF_LN_CASE(7) RF_LN_CASE(8)
    default: RF_REQUIRE(false, RF_ERR_SHAPE, "rf_layernorm_modulate: D too large");
  }
#undef RF_LN_CASE
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_qk_rmsnorm_rope(void* q, void* k, int32_t heads, int32_t S, int32_t s_pad, int32_t n_added,
                                  const void* w_q, const void* w_k, const void* w_added_q, const void* w_added_k,
                                  const float* cos_tab, const float* sin_tab, float eps, void* stream) {
  using namespace rf;
  RF_REQUIRE(q && k && w_q && w_k && cos_tab && sin_tab, RF_ERR_NULL, "rf_qk_rmsnorm_rope: NULL pointer");
  RF_REQUIRE(heads > 0 && S > 0 && s_pad >= S && n_added >= 0 && n_added <= S, RF_ERR_SHAPE, "rf_qk_rmsnorm_rope: bad shape");
  RF_REQUIRE(n_added == 0 || (w_added_q && w_added_k), RF_ERR_NULL, "rf_qk_rmsnorm_rope: added-norm weights NULL");
  RF_REQUIRE(aligned16(q) && aligned16(k) && aligned16(w_q) && aligned16(w_k) && aligned16(cos_tab) && aligned16(sin_tab),
             RF_ERR_ALIGN, "rf_qk_rmsnorm_rope: operands must be 16-byte aligned");
  if (n_added == 0) { w_added_q = w_q; w_added_k = w_k; }
  const int64_t rows = (int64_t)2 * heads * S;
  int64_t blocks = (rows + 15) / 16;
  if (blocks > 256 * 16) blocks = 256 * 16;
  ProfScope prof(RF_KC_ROWOP, (double)rows * 128.0 * 4.0, (hipStream_t)stream);
  hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)q, (bf16_t*)k,
                     heads, S, s_pad, n_added, (const bf16_t*)w_q, (const bf16_t*)w_k, (const bf16_t*)w_added_q,
                     (const bf16_t*)w_added_k, cos_tab, sin_tab, eps);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_euler_step(void* x, const void* v, int64_t n, float dt, void* stream) {
  using namespace rf;
  if (n <= 0) return RF_OK;
  RF_REQUIRE(x && v, RF_ERR_NULL, "rf_euler_step: NULL pointer");
  RF_REQUIRE(aligned16(x) && aligned16(v), RF_ERR_ALIGN, "rf_euler_step: operands must be 16-byte aligned");
  ProfScope prof(RF_KC_ROWOP, 6.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(euler_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x,
                     (const bf16_t*)v, n, dt);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_silu(const void* x, void* out, int64_t n, void* stream) {
  using namespace rf;
  if (n <= 0) return RF_OK;
  RF_REQUIRE(x && out, RF_ERR_NULL, "rf_silu: NULL pointer");
  ProfScope prof(RF_KC_ROWOP, 4.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)out, n);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_add_inplace(void* out, const void* x, int64_t n, void* stream) {
  using namespace rf;
  if (n <= 0) return RF_OK;
  RF_REQUIRE(x && out, RF_ERR_NULL, "rf_add_inplace: NULL pointer");
  RF_REQUIRE(aligned16(x) && aligned16(out), RF_ERR_ALIGN, "rf_add_inplace: operands must be 16-byte aligned");
  ProfScope prof(RF_KC_ROWOP, 6.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out,
                     (const bf16_t*)x, n);
  RF_LAUNCH_CHECK();
  return RF_OK;
}
