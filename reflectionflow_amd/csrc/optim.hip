// Optimizer updates of the LoRA factors as kernels over ONE flat bucket (train_flux/train/model.py:105-117 builds
// torch.optim.AdamW or prodigyopt.Prodigy over the LoRA parameters, config.yaml:55-61 ships Prodigy lr 1, use_bias_correction,
// safeguard_warmup, weight_decay 0.01).
//
// The training binding keeps every LoRA factor as a view into one flat bf16 buffer and every gradient as a view into a second
// one (reflectionflow_amd/train/optim.py): the data-parallel all-reduce runs on the gradient buffer as it lies, its 1 / world_size
// rides into the update as `grad_scale`, and the whole update is one launch (AdamW) or three (Prodigy) instead of a multi-tensor
// foreach over 354 tensors.  All of it is HBM-bound streaming: 16-byte accesses, fp32 arithmetic, one rounding per stored value.
//
//   rf_lora_adamw    torch.optim.AdamW's update (decoupled decay, bias correction, no amsgrad), element for element
//   rf_lora_prodigy  Prodigy (Mishchenko & Defazio 2023, "Prodigy: An Expeditiously Adaptive Parameter-Free Learner", Algorithm 4 =
//                    the Adam form; option names and defaults of the `prodigyopt` package, which is NOT in this image: parity
//                    unpinned, see oracle/optim_oracle.py).  The distance estimate d lives on the DEVICE (`dstate`): the two global
//                    sums are fixed-order two-stage reductions (bit-reproducible) and no value crosses to the host inside a step,
//                    where the package reads one .item() per parameter tensor.
#include "common.hpp"
#include <math.h>

namespace rf {

constexpr int OPT_THREADS = 256;
constexpr int OPT_VEC = 8;                          // bf16 elements per lane and access (16 bytes)
constexpr int OPT_ITEMS = 4;                        // accesses per lane
constexpr int OPT_BLOCK_ELEMS = OPT_THREADS * OPT_VEC * OPT_ITEMS;   // 8192

template <bool F32>
struct StateIO {
  // eight consecutive state values at element index i (i % 8 == 0)
  static __device__ __forceinline__ void load(const void* base, int64_t i, float (&f)[8]) {
    if constexpr (F32) {
      const f32x4* p = (const f32x4*)((const float*)base + i);
      const f32x4 a = p[0], b = p[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = a[j], f[4 + j] = b[j];
    } else {
      unpack8(*(const u32x4*)((const bf16_t*)base + i), f);
    }
  }
  static __device__ __forceinline__ void store(void* base, int64_t i, const float (&f)[8]) {
    if constexpr (F32) {
      f32x4 a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = f[j], b[j] = f[4 + j];
      f32x4* p = (f32x4*)((float*)base + i);
      p[0] = a, p[1] = b;
    } else {
      *(u32x4*)((bf16_t*)base + i) = pack8(f);
    }
  }
};

__device__ __forceinline__ void load_bf8(const bf16_t* base, int64_t i, float (&f)[8]) { unpack8(*(const u32x4*)(base + i), f); }
__device__ __forceinline__ void store_bf8(bf16_t* base, int64_t i, const float (&f)[8]) { *(u32x4*)(base + i) = pack8(f); }

// ---- AdamW -------------------------------------------------------------------------------------------------------------------
// torch.optim.AdamW (torch/optim/adam.py::_single_tensor_adam with decoupled_weight_decay):
//   p *= 1 - lr wd;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
struct AdamwScalars {
  float decay_mul, w1, beta2, one_minus_beta2, step_size, inv_bc2_sqrt, eps, grad_scale;
};

template <bool F32>
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(bf16_t* __restrict__ param, const bf16_t* __restrict__ grad, void* __restrict__ exp_avg,
                                                            void* __restrict__ exp_avg_sq, int64_t n, AdamwScalars c) {
  const int64_t base = (int64_t)blockIdx.x * OPT_BLOCK_ELEMS + threadIdx.x * OPT_VEC;
#pragma unroll
  for (int it = 0; it < OPT_ITEMS; ++it) {
    const int64_t i = base + (int64_t)it * OPT_THREADS * OPT_VEC;
    if (i >= n) break;                                    // n % 8 == 0: an access is all in or all out
    float p[8], g[8], m[8], v[8];
    load_bf8(param, i, p);
    load_bf8(grad, i, g);
    StateIO<F32>::load(exp_avg, i, m);
    StateIO<F32>::load(exp_avg_sq, i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gj = g[j] * c.grad_scale;
      float pj = p[j] * c.decay_mul;
      // torch's lerp: a + w (b - a) for w < 0.5, b - (b - a)(1 - w) otherwise
      const float diff = gj - m[j];
      m[j] = c.w1 < 0.5f ? m[j] + c.w1 * diff : gj - diff * (1.0f - c.w1);
      v[j] = c.beta2 * v[j] + c.one_minus_beta2 * gj * gj;
      const float denom = sqrtf(v[j]) * c.inv_bc2_sqrt + c.eps;
      pj -= c.step_size * (m[j] / denom);
      p[j] = pj;
    }
    store_bf8(param, i, p);
    StateIO<F32>::store(exp_avg, i, m);
    StateIO<F32>::store(exp_avg_sq, i, v);
  }
}

// ---- Prodigy -----------------------------------------------------------------------------------------------------------------
// dstate (device, 8 doubles): [0] d  [1] d_max  [2] d_numerator  [3] d_denom  [4] d_hat  [5] k (steps taken)  [6] dlr of the step in
// flight  [7] d0.   One step = moments (per element + two block partial sums) -> finalize (one block: the sums in block order, the
// new d) -> apply.
struct ProdigyCfg {
  float lr, beta1, beta2, beta3, eps, decay, d_coef, growth_rate, grad_scale;
  int decouple, use_bias_correction, safeguard_warmup;
};

__device__ __forceinline__ double prodigy_dlr(const double* dstate, const ProdigyCfg& c) {
  const double d = dstate[0], k = dstate[5];
  double bc = 1.0;
  if (c.use_bias_correction) bc = sqrt(1.0 - pow((double)c.beta2, k + 1.0)) / (1.0 - pow((double)c.beta1, k + 1.0));
  return d * (double)c.lr * bc;
}

template <bool F32>
__global__ __launch_bounds__(OPT_THREADS) void prodigy_moments_kernel(const bf16_t* __restrict__ param, const bf16_t* __restrict__ grad,
                                                                      void* __restrict__ exp_avg, void* __restrict__ exp_avg_sq, void* __restrict__ s_,
                                                                      const bf16_t* __restrict__ p0, int64_t n, const double* __restrict__ dstate,
                                                                      ProdigyCfg c, float* __restrict__ partials) {
  __shared__ float sc[3];
  __shared__ float red[2][OPT_THREADS / WAVE];
  if (threadIdx.x == 0) {
    const double d = dstate[0], d0 = dstate[7], dlr = prodigy_dlr(dstate, c);
    sc[0] = (float)(d * (1.0 - (double)c.beta1));
    sc[1] = (float)(d * d * (1.0 - (double)c.beta2));
    sc[2] = (float)((d / d0) * (c.safeguard_warmup ? d : dlr));
  }
  __syncthreads();
  const float a_m = sc[0], a_v = sc[1], a_s = sc[2];
  const bool coupled = c.decay != 0.f && !c.decouple;
  float num = 0.f, den = 0.f;
  const int64_t base = (int64_t)blockIdx.x * OPT_BLOCK_ELEMS + threadIdx.x * OPT_VEC;
#pragma unroll
  for (int it = 0; it < OPT_ITEMS; ++it) {
    const int64_t i = base + (int64_t)it * OPT_THREADS * OPT_VEC;
    if (i >= n) break;
    float p[8], g[8], q[8], m[8], v[8], s[8];
    load_bf8(param, i, p);
    load_bf8(grad, i, g);
    load_bf8(p0, i, q);
    StateIO<F32>::load(exp_avg, i, m);
    StateIO<F32>::load(exp_avg_sq, i, v);
    StateIO<F32>::load(s_, i, s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gj = g[j] * c.grad_scale;
      if (coupled) gj += c.decay * p[j];
      num += gj * (q[j] - p[j]);
      m[j] = c.beta1 * m[j] + a_m * gj;
      v[j] = c.beta2 * v[j] + a_v * gj * gj;
      s[j] = c.beta3 * s[j] + a_s * gj;
      den += fabsf(s[j]);
    }
    StateIO<F32>::store(exp_avg, i, m);
    StateIO<F32>::store(exp_avg_sq, i, v);
    StateIO<F32>::store(s_, i, s);
  }
  // block sums in a fixed order: lanes by butterfly, waves 0..3 in order
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) num += __shfl_xor(num, o), den += __shfl_xor(den, o);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = num, red[1][threadIdx.x >> 6] = den;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * (int64_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[2 * (int64_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(OPT_THREADS) void prodigy_finalize_kernel(double* __restrict__ dstate, const float* __restrict__ partials, int nblocks,
                                                                       ProdigyCfg c) {
  __shared__ double red[2][OPT_THREADS];
  double num = 0.0, den = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += OPT_THREADS) num += (double)partials[2 * b], den += (double)partials[2 * b + 1];
  red[0][threadIdx.x] = num, red[1][threadIdx.x] = den;
  __syncthreads();
  for (int o = OPT_THREADS / 2; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o], red[1][threadIdx.x] += red[1][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  double d = dstate[0], d_max = dstate[1];
  const double d0 = dstate[7], dlr = prodigy_dlr(dstate, c);
  const double d_numerator = dstate[2] * (double)c.beta3 + (d / d0) * dlr * red[0][0];
  const double d_denom = red[1][0];
  dstate[6] = dlr;
  if (d_denom == 0.0) {             // no gradient anywhere: the package returns before touching d, the parameters or k
    dstate[6] = 0.0;
    return;
  }
  double d_hat = d;
  if (c.lr > 0.f) {
    d_hat = (double)c.d_coef * d_numerator / d_denom;
    if (d == d0) d = fmax(d, d_hat);
    d_max = fmax(d_max, d_hat);
    d = fmin(d_max, d * (double)c.growth_rate);
  }
  dstate[0] = d, dstate[1] = d_max, dstate[2] = d_numerator, dstate[3] = d_denom, dstate[4] = d_hat;
  dstate[5] += 1.0;
}

template <bool F32>
__global__ __launch_bounds__(OPT_THREADS) void prodigy_apply_kernel(bf16_t* __restrict__ param, const void* __restrict__ exp_avg,
                                                                    const void* __restrict__ exp_avg_sq, int64_t n, const double* __restrict__ dstate,
                                                                    ProdigyCfg c) {
  const float dlr = (float)dstate[6];
  if (dlr == 0.f) return;       // the skipped step (finalize left dlr = 0), or lr = 0: the update is the identity either way
  const float d_eps = (float)(dstate[0] * (double)c.eps);
  const float decay_mul = (c.decay != 0.f && c.decouple) ? -c.decay * dlr : 0.f;
  const int64_t base = (int64_t)blockIdx.x * OPT_BLOCK_ELEMS + threadIdx.x * OPT_VEC;
#pragma unroll
  for (int it = 0; it < OPT_ITEMS; ++it) {
    const int64_t i = base + (int64_t)it * OPT_THREADS * OPT_VEC;
    if (i >= n) break;
    float p[8], m[8], v[8];
    load_bf8(param, i, p);
    StateIO<F32>::load(exp_avg, i, m);
    StateIO<F32>::load(exp_avg_sq, i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float pj = p[j];
      pj += pj * decay_mul;
      pj -= dlr * (m[j] / (sqrtf(v[j]) + d_eps));
      p[j] = pj;
    }
    store_bf8(param, i, p);
  }
}

// ---- gradient clipping by global L2 norm -------------------------------------------------------------------------------------------
// The reference's Lightning Trainer is built with gradient_clip_val = 0.5 (train_flux/train/train.py:165; config.yaml does not override
// it), i.e. torch.nn.utils.clip_grad_norm_(params, 0.5) between the DDP all-reduce and optimizer.step:
//   total_norm = || grad_scale * grad ||_2;  coef = min(1, max_norm / (total_norm + 1e-6));  grad *= coef
// Three launches over the flat gradient bucket: block partial sums of squares (fp32 per lane, fixed butterfly / wave order), one
// workgroup adding the partials in block order in fp64 and forming {total_norm, coef} ON THE DEVICE (`out`, 2 floats: no host sync
// in a step), and the in-place scale (skipped per workgroup when coef == 1: a multiply by 1 is the identity in bf16 as well).
__global__ __launch_bounds__(OPT_THREADS) void sumsq_kernel(const bf16_t* __restrict__ grad, int64_t n, float* __restrict__ partials) {
  __shared__ float red[OPT_THREADS / WAVE];
  float acc = 0.f;
  const int64_t base = (int64_t)blockIdx.x * OPT_BLOCK_ELEMS + threadIdx.x * OPT_VEC;
#pragma unroll
  for (int it = 0; it < OPT_ITEMS; ++it) {
    const int64_t i = base + (int64_t)it * OPT_THREADS * OPT_VEC;
    if (i >= n) break;
    float g[8];
    load_bf8(grad, i, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += g[j] * g[j];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_coef_kernel(const float* __restrict__ partials, int nblocks, float max_norm, float grad_scale,
                                                                float* __restrict__ out) {
  __shared__ double red[OPT_THREADS];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += OPT_THREADS) s += (double)partials[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = OPT_THREADS / 2; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double norm = sqrt(red[0]) * fabs((double)grad_scale);
  const double coef = (double)max_norm / (norm + 1e-6);
  out[0] = (float)norm;
  out[1] = (coef < 1.0 || coef != coef) ? (float)coef : 1.0f;   // torch.clamp(coef, max=1) keeps a NaN coefficient (non-finite norm, error_if_nonfinite=False)
}

__global__ __launch_bounds__(OPT_THREADS) void scale_grad_kernel(bf16_t* __restrict__ grad, int64_t n, const float* __restrict__ out) {
  const float coef = out[1];
  if (coef == 1.0f) return;
  const int64_t base = (int64_t)blockIdx.x * OPT_BLOCK_ELEMS + threadIdx.x * OPT_VEC;
#pragma unroll
  for (int it = 0; it < OPT_ITEMS; ++it) {
    const int64_t i = base + (int64_t)it * OPT_THREADS * OPT_VEC;
    if (i >= n) break;
    float g[8];
    load_bf8(grad, i, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= coef;
    store_bf8(grad, i, g);
  }
}

static int opt_check(const char* who, const void* param, const void* grad, const void* m, const void* v, int64_t n) {
  RF_REQUIRE(param && grad && m && v, RF_ERR_NULL, "%s: NULL pointer", who);
  RF_REQUIRE(n > 0 && n % OPT_VEC == 0 && n / OPT_BLOCK_ELEMS < (1ll << 30), RF_ERR_SHAPE, "%s: n=%lld (need n > 0, n %% 8 == 0)", who, (long long)n);
  RF_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(m) && aligned16(v), RF_ERR_ALIGN, "%s: buffers must be 16-byte aligned", who);
  return RF_OK;
}

}  // namespace rf

extern "C" int rf_lora_adamw(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, int32_t state_fp32, int32_t step,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  using namespace rf;
  if (int rc = opt_check("rf_lora_adamw", param, grad, exp_avg, exp_avg_sq, n)) return rc;
  RF_REQUIRE(step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, RF_ERR_SHAPE, "rf_lora_adamw: step=%d betas=(%g, %g)", step,
             beta1, beta2);
  AdamwScalars c;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  c.decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
  c.w1 = (float)(1.0 - (double)beta1);
  c.beta2 = beta2;
  c.one_minus_beta2 = (float)(1.0 - (double)beta2);
  c.step_size = (float)((double)lr / bc1);
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  c.eps = eps;
  c.grad_scale = grad_scale;
  const unsigned blocks = (unsigned)((n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS);
  const double sb = state_fp32 ? 4.0 : 2.0;
  ProfScope prof(RF_KC_ROWOP, (double)n * (2.0 + 2.0 + 2.0 + 4.0 * sb), (hipStream_t)stream);
  if (state_fp32)
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, (bf16_t*)param, (const bf16_t*)grad, exp_avg,
                       exp_avg_sq, n, c);
  else
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, (bf16_t*)param, (const bf16_t*)grad, exp_avg,
                       exp_avg_sq, n, c);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int64_t rf_lora_prodigy_partials_bytes(int64_t n) {
  return n <= 0 ? 0 : (int64_t)((n + rf::OPT_BLOCK_ELEMS - 1) / rf::OPT_BLOCK_ELEMS) * 2 * (int64_t)sizeof(float);
}

extern "C" int rf_lora_prodigy(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, void* s, const void* p0, int64_t n,
                               int32_t state_fp32, double* dstate, float lr, float beta1, float beta2, float beta3, float eps,
                               float weight_decay, int32_t decouple, int32_t use_bias_correction, int32_t safeguard_warmup, float d_coef,
                               float growth_rate, float grad_scale, float* partials, int64_t partials_bytes, void* stream) {
  using namespace rf;
  if (int rc = opt_check("rf_lora_prodigy", param, grad, exp_avg, exp_avg_sq, n)) return rc;
  RF_REQUIRE(s && p0 && dstate && partials, RF_ERR_NULL, "rf_lora_prodigy: NULL pointer");
  RF_REQUIRE(aligned16(s) && aligned16(p0), RF_ERR_ALIGN, "rf_lora_prodigy: buffers must be 16-byte aligned");
  RF_REQUIRE(partials_bytes >= rf_lora_prodigy_partials_bytes(n), RF_ERR_WORKSPACE, "rf_lora_prodigy: partials %lld B < %lld B",
             (long long)partials_bytes, (long long)rf_lora_prodigy_partials_bytes(n));
  RF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && lr >= 0.f, RF_ERR_SHAPE, "rf_lora_prodigy: lr=%g betas=(%g, %g)", lr,
             beta1, beta2);
  if (lr == 0.f) return RF_OK;   // prodigyopt gates the moment / s / numerator updates on group_lr > 0 and then returns on d_denom == 0: nothing moves, k stays
  ProdigyCfg c;
  c.lr = lr, c.beta1 = beta1, c.beta2 = beta2, c.beta3 = beta3 > 0.f ? beta3 : sqrtf(beta2), c.eps = eps, c.decay = weight_decay;
  c.d_coef = d_coef, c.growth_rate = growth_rate, c.grad_scale = grad_scale;
  c.decouple = decouple, c.use_bias_correction = use_bias_correction, c.safeguard_warmup = safeguard_warmup;
  const unsigned blocks = (unsigned)((n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS);
  const double sb = state_fp32 ? 4.0 : 2.0;
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope prof(RF_KC_ROWOP, (double)n * (3.0 * 2.0 + 6.0 * sb), st);
    if (state_fp32)
      hipLaunchKernelGGL(prodigy_moments_kernel<true>, dim3(blocks), dim3(OPT_THREADS), 0, st, (const bf16_t*)param, (const bf16_t*)grad, exp_avg,
                         exp_avg_sq, s, (const bf16_t*)p0, n, (const double*)dstate, c, partials);
    else
      hipLaunchKernelGGL(prodigy_moments_kernel<false>, dim3(blocks), dim3(OPT_THREADS), 0, st, (const bf16_t*)param, (const bf16_t*)grad, exp_avg,
                         exp_avg_sq, s, (const bf16_t*)p0, n, (const double*)dstate, c, partials);
    RF_LAUNCH_CHECK();
  }
  {
    ProfScope prof(RF_KC_ROWOP, (double)blocks * 8.0, st);
    hipLaunchKernelGGL(prodigy_finalize_kernel, dim3(1), dim3(OPT_THREADS), 0, st, dstate, (const float*)partials, (int)blocks, c);
    RF_LAUNCH_CHECK();
  }
  {
    ProfScope prof(RF_KC_ROWOP, (double)n * (2.0 * 2.0 + 2.0 * sb), st);
    if (state_fp32)
      hipLaunchKernelGGL(prodigy_apply_kernel<true>, dim3(blocks), dim3(OPT_THREADS), 0, st, (bf16_t*)param, (const void*)exp_avg,
                         (const void*)exp_avg_sq, n, (const double*)dstate, c);
    else
      hipLaunchKernelGGL(prodigy_apply_kernel<false>, dim3(blocks), dim3(OPT_THREADS), 0, st, (bf16_t*)param, (const void*)exp_avg,
                         (const void*)exp_avg_sq, n, (const double*)dstate, c);
    RF_LAUNCH_CHECK();
  }
  return RF_OK;
}

extern "C" int rf_lora_clip_grad_norm(void* grad, int64_t n, float max_norm, float grad_scale, float* partials, int64_t partials_bytes,
                                      float* out, void* stream) {
  using namespace rf;
  RF_REQUIRE(grad && partials && out, RF_ERR_NULL, "rf_lora_clip_grad_norm: NULL pointer");
  RF_REQUIRE(n > 0 && n % OPT_VEC == 0 && n / OPT_BLOCK_ELEMS < (1ll << 30), RF_ERR_SHAPE, "rf_lora_clip_grad_norm: n=%lld (need n > 0, n %% 8 == 0)",
             (long long)n);
  RF_REQUIRE(aligned16(grad), RF_ERR_ALIGN, "rf_lora_clip_grad_norm: grad must be 16-byte aligned");
  RF_REQUIRE(max_norm > 0.f, RF_ERR_SHAPE, "rf_lora_clip_grad_norm: max_norm=%g", max_norm);
  const unsigned blocks = (unsigned)((n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS);
  RF_REQUIRE(partials_bytes >= (int64_t)blocks * 4, RF_ERR_WORKSPACE, "rf_lora_clip_grad_norm: partials %lld B < %lld B", (long long)partials_bytes,
             (long long)blocks * 4);
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope prof(RF_KC_ROWOP, (double)n * 2.0, st);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(OPT_THREADS), 0, st, (const bf16_t*)grad, n, partials);
    RF_LAUNCH_CHECK();
  }
  {
    ProfScope prof(RF_KC_ROWOP, (double)blocks * 4.0, st);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(OPT_THREADS), 0, st, (const float*)partials, (int)blocks, max_norm, grad_scale, out);
    RF_LAUNCH_CHECK();
  }
  {
    ProfScope prof(RF_KC_ROWOP, (double)n * 4.0, st);
    hipLaunchKernelGGL(scale_grad_kernel, dim3(blocks), dim3(OPT_THREADS), 0, st, (bf16_t*)grad, n, (const float*)out);
    RF_LAUNCH_CHECK();
  }
  return RF_OK;
}
