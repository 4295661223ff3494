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

// The same row kernel over up to three streams in one launch: block b serves the stream whose 4-row-block prefix covers it (a text
// stream of 512 rows is a 4 us launch of its own otherwise, plus the launch boundary: 38 of them per cfg2 forward).
struct LnModGroup {
  LnModStream s[3];
  int blk_end[3];   // exclusive prefix of 4-row blocks
  int n;
};
template <int NCH>
__global__ __launch_bounds__(256) void ln_mod_grouped_kernel(const LnModGroup g, int64_t ldx, int64_t ldo, int D, float eps) {
  int gi = 0;
  if (g.n > 1 && (int)blockIdx.x >= g.blk_end[0]) gi = 1;
  if (g.n > 2 && (int)blockIdx.x >= g.blk_end[1]) gi = 2;
  const LnModStream& S = g.s[gi];
  const int b0 = gi == 0 ? 0 : g.blk_end[gi - 1];
  const int lane = threadIdx.x & 63;
  const int row = ((int)blockIdx.x - b0) * 4 + (threadIdx.x >> 6);
  if (row >= S.rows) return;
  const bf16_t* xr = S.x + (int64_t)row * ldx;
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
  bf16_t* orow = S.out + (int64_t)row * ldo;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      float sc[8], sh[8], o[8];
      unpack8(*(const u32x4*)(S.scale + col), sc);
      unpack8(*(const u32x4*)(S.shift + col), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * (1.0f + sc[j]) + sh[j];
      *(u32x4*)(orow + col) = pack8(o);
    }
  }
}

int ln_mod_grouped(const LnModStream* streams, int n, int64_t ldx, int64_t ldo, int D, float eps, hipStream_t stream) {
  LnModGroup g;
  g.n = 0;
  int blocks = 0;
  double bytes = 0.0;
  for (int i = 0; i < n && g.n < 3; ++i) {
    if (streams[i].rows <= 0) continue;
    RF_REQUIRE(streams[i].x && streams[i].out && streams[i].scale && streams[i].shift, RF_ERR_NULL, "ln_mod_grouped: NULL pointer");
    RF_REQUIRE(aligned16(streams[i].x) && aligned16(streams[i].out) && aligned16(streams[i].scale) && aligned16(streams[i].shift),
               RF_ERR_ALIGN, "ln_mod_grouped: operands must be 16-byte aligned");
    g.s[g.n] = streams[i];
    blocks += cdiv(streams[i].rows, 4);
    g.blk_end[g.n] = blocks;
    bytes += 4.0 * streams[i].rows * (double)D;
    ++g.n;
  }
  if (g.n == 0) return RF_OK;
  RF_REQUIRE(D > 0 && D % 8 == 0 && D <= 8 * 512 && ldx % 8 == 0 && ldo % 8 == 0, RF_ERR_SHAPE, "ln_mod_grouped: D=%d", D);
  for (int i = g.n; i < 3; ++i) g.s[i] = g.s[0], g.blk_end[i] = blocks;
  ProfScope prof(RF_KC_ROWOP, bytes, stream);
  const int nch = cdiv(D, 512);
#define RF_LNG_CASE(N) \
  case N: hipLaunchKernelGGL(ln_mod_grouped_kernel<N>, dim3(blocks), dim3(256), 0, stream, g, ldx, ldo, D, eps); break;
  switch (nch) {
    RF_LNG_CASE(1) RF_LNG_CASE(2) RF_LNG_CASE(3) RF_LNG_CASE(4) RF_LNG_CASE(5) RF_LNG_CASE(6) RF_LNG_CASE(7) RF_LNG_CASE(8)
    default: RF_REQUIRE(false, RF_ERR_SHAPE, "ln_mod_grouped: D too large");
  }
#undef RF_LNG_CASE
  RF_LAUNCH_CHECK();
  return RF_OK;
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


// ---- fp8 activation quantisation (rf_gemm_w8a8's A operand) --------------------------------------------------
// Symmetric per-row absmax: scale = amax / 448 (1 for an all-zero row), q = e4m3fn(x / scale).  v_cvt_pk_fp8_f32 on
// gfx950 converts to OCP e4m3fn with round-to-nearest-even; inputs are clamped to +-448 so the result never
// depends on the instruction's overflow behaviour.
constexpr float FP8_MAX = 448.0f;

__device__ __forceinline__ uint32_t f32x4_to_fp8(float a, float b, float c, float d) {
  int v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
#endif
  return (uint32_t)v;
}

__device__ __forceinline__ u32x2 quant8(const float (&f)[8], float inv) {
  float t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = fminf(fmaxf(f[j] * inv, -FP8_MAX), FP8_MAX);
  u32x2 r;
  r[0] = f32x4_to_fp8(t[0], t[1], t[2], t[3]);
  r[1] = f32x4_to_fp8(t[4], t[5], t[6], t[7]);
  return r;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// LayerNorm(no affine) + (1+scale) x + shift -> fp8 row + its dequantisation scale.  One wave per row (as ln_mod_kernel).
template <int NCH>
__global__ __launch_bounds__(256) void ln_mod_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out,
                                                         int64_t ldo, float* __restrict__ row_scale, int rows, int D,
                                                         const bf16_t* __restrict__ scale, const bf16_t* __restrict__ shift,
                                                         float eps) {
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
      unpack8(*(const u32x4*)(xr + col), v[c]);
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
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      float sc[8], sh[8];
      unpack8(*(const u32x4*)(scale + col), sc);
      unpack8(*(const u32x4*)(shift + col), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[c][j] = (v[c][j] - mean) * rstd * (1.0f + sc[j]) + sh[j];
        amax = fmaxf(amax, fabsf(v[c][j]));
      }
    }
  }
  amax = wave_max(amax);
  const float sc_row = amax > 0.f ? amax * (1.0f / FP8_MAX) : 1.0f;
  const float inv = 1.0f / sc_row;
  if (lane == 0) row_scale[row] = sc_row;
  uint8_t* orow = out + (int64_t)row * ldo;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) *(u32x2*)(orow + col) = quant8(v[c], inv);
  }
}

// bf16 rows (one or two column segments, e.g. the single block's [attn | mlp] input) -> fp8 rows with ONE common scale
// per row.  One 256-thread block per row, the row stays in registers (<= NCH x 2048 columns).
template <int NCH>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x0, int64_t ld0, int K0,
                                                             const bf16_t* __restrict__ x1, int64_t ld1, int K1,
                                                             uint8_t* __restrict__ out, int64_t ldo, float* __restrict__ row_scale) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t row = blockIdx.x;
  const int K = K0 + K1;
  float v[NCH][8];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 256 + tid) * 8;
    if (col < K) {
      const bf16_t* src = col < K0 ? x0 + row * ld0 + col : x1 + row * ld1 + (col - K0);
      unpack8(*(const u32x4*)src, v[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[c][j]));
    }
  }
  amax = wave_max(amax);
  if (lane == 0) red[w] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc_row = amax > 0.f ? amax * (1.0f / FP8_MAX) : 1.0f;
  const float inv = 1.0f / sc_row;
  if (tid == 0) row_scale[row] = sc_row;
  uint8_t* orow = out + row * ldo;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 256 + tid) * 8;
    if (col < K) *(u32x2*)(orow + col) = quant8(v[c], inv);
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


extern "C" int rf_layernorm_modulate_fp8(const void* x, int64_t ldx, void* out8, int64_t ldo, float* row_scale, int32_t rows,
                                         int32_t D, const void* scale, const void* shift, float eps, void* stream) {
  using namespace rf;
  if (rows <= 0) return RF_OK;
  RF_REQUIRE(x && out8 && row_scale && scale && shift, RF_ERR_NULL, "rf_layernorm_modulate_fp8: NULL pointer");
  RF_REQUIRE(D > 0 && D % 8 == 0 && D <= 8 * 512, RF_ERR_SHAPE, "rf_layernorm_modulate_fp8: D=%d (need D%%8==0, D<=4096)", D);
  RF_REQUIRE(aligned16(x) && aligned16(scale) && aligned16(shift) && ldx % 8 == 0 && ((uintptr_t)out8 & 7u) == 0 && ldo % 8 == 0,
             RF_ERR_ALIGN, "rf_layernorm_modulate_fp8: operands must be 16-byte (fp8 output: 8-byte) aligned");
  const dim3 grid(cdiv(rows, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(RF_KC_QUANT, 3.0 * rows * (double)D, s);  // bytes: bf16 row in, fp8 row out
  const int nch = cdiv(D, 512);
#define RF_LN8_CASE(N)                                                                                              \
  case N:                                                                                                           \
    hipLaunchKernelGGL(ln_mod_fp8_kernel<N>, grid, block, 0, s, (const bf16_t*)x, ldx, (uint8_t*)out8, ldo, row_scale, \
                       rows, D, (const bf16_t*)scale, (const bf16_t*)shift, eps);                                  \
    break;
  switch (nch) {
    RF_LN8_CASE(1) RF_LN8_CASE(2) RF_LN8_CASE(3) RF_LN8_CASE(4) RF_LN8_CASE(5) RF_LN8_CASE(6) RF_LN8_CASE(7) RF_LN8_CASE(8)
    default: RF_REQUIRE(false, RF_ERR_SHAPE, "rf_layernorm_modulate_fp8: D too large");
  }
#undef RF_LN8_CASE
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_quant_rows_fp8(const void* x0, int64_t ld0, int32_t K0, const void* x1, int64_t ld1, int32_t K1, void* out8,
                                 int64_t ldo, float* row_scale, int32_t rows, void* stream) {
  using namespace rf;
  if (rows <= 0) return RF_OK;
  RF_REQUIRE(x0 && out8 && row_scale && K0 > 0 && K1 >= 0 && (K1 == 0 || x1), RF_ERR_NULL, "rf_quant_rows_fp8: NULL pointer / bad K");
  const int K = K0 + K1;
  RF_REQUIRE(K0 % 8 == 0 && K1 % 8 == 0 && K <= 8 * 2048, RF_ERR_SHAPE, "rf_quant_rows_fp8: K0=%d K1=%d (multiples of 8, sum <= 16384)", K0, K1);
  RF_REQUIRE(aligned16(x0) && ld0 % 8 == 0 && (K1 == 0 || (aligned16(x1) && ld1 % 8 == 0)) && ((uintptr_t)out8 & 7u) == 0 && ldo % 8 == 0,
             RF_ERR_ALIGN, "rf_quant_rows_fp8: inputs must be 16-byte, the fp8 output 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(RF_KC_QUANT, 3.0 * rows * (double)K, s);
  const dim3 grid(rows), block(256);
  const int nch = cdiv(K, 2048);
#define RF_Q8_CASE(N)                                                                                                   \
  case N:                                                                                                               \
    hipLaunchKernelGGL(quant_rows_fp8_kernel<N>, grid, block, 0, s, (const bf16_t*)x0, ld0, K0, (const bf16_t*)x1, ld1, K1, \
                       (uint8_t*)out8, ldo, row_scale);                                                                 \
    break;
  switch (nch) {
    RF_Q8_CASE(1) RF_Q8_CASE(2) RF_Q8_CASE(3) RF_Q8_CASE(4) RF_Q8_CASE(5) RF_Q8_CASE(6) RF_Q8_CASE(7) RF_Q8_CASE(8)
    default: RF_REQUIRE(false, RF_ERR_SHAPE, "rf_quant_rows_fp8: row too long");
  }
#undef RF_Q8_CASE
  RF_LAUNCH_CHECK();
  return RF_OK;
}
