// Text encoders of the FLUX pipeline on the HIP path: T5-v1.1 encoder (prompt_embeds) and CLIP text tower
// (pooled_prompt_embeds), reference call site train_flux/flux/generate.py:148-161 -> FluxPipeline.encode_prompt, run per candidate
// and round by tts/tts_reflectionflow.py:286-294.  B sequences of S tokens per call (a rank's candidates of a round: the projections
// then run as ONE GEMM over B * S rows instead of B skinny ones); token ids in, hidden states out; tokenisation is the caller's
// (vocabulary files are host-side data).
//
// Every matrix product is a launch of the bf16 MFMA GEMM that already exists (rf_gemm_bf16): q|k projection, V^T = W_v x^T with the
// operands swapped (the attention kernel wants V transposed), out projection and FFN down-projection with the residual add in the
// epilogue (RF_EPI_GATE_RES, gate of ones), fused wi_0|wi_1 / fc1.  New here:
//   * attn64_kernel: one (head, 64-query block) per workgroup, head dim 64, keys <= 512: the head's K and V^T live in LDS whole
//     (XOR-swizzled K rows, padded + key-permuted V^T rows), S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_16x16x32_bf16 with P kept
//     in registers, full-row softmax in fp32 (no running maximum: the whole row of scores is in registers), additive fp32 bias
//     [heads or 1][S_pad][S_pad] (T5: bucketed relative positions; CLIP: the causal mask; both: -inf on padded keys);
//   * row kernels (HBM-bound, 16-byte accesses): embedding gather (+ position rows), T5 RMS norm, gelu_new(a) * b, quick_gelu.
#include "common.hpp"

namespace rf {

#define RF_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != RF_OK) return _rc; \
  } while (0)

static inline int grid_for(int64_t n, int per_block = 256) { return (int)((n + per_block - 1) / per_block); }

// ---- row kernels -------------------------------------------------------------------------------------------------------------
// out[b * S_pad + i][:] = table[ids[b * S + i]][:] (+ pos[i][:]); rows i >= S of a sequence are zero-filled; D % 8 == 0
__global__ __launch_bounds__(256) void embed_rows_kernel(const bf16_t* __restrict__ table, const bf16_t* __restrict__ pos, const int32_t* __restrict__ ids,
                                                         bf16_t* __restrict__ out, int S, int S_pad, int n_rows, int D, int vocab) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d8 = D >> 3;
  if (idx >= (int64_t)n_rows * d8) return;
  const int grow = (int)(idx / d8), c = (int)(idx % d8) * 8;
  const int b = grow / S_pad, row = grow - b * S_pad;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < S) {
    int id = ids[b * S + row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    v = *(const u32x4*)(table + (int64_t)id * D + c);
    if (pos != nullptr) {
      float a[8], b[8];
      unpack8(v, a);
      unpack8(*(const u32x4*)(pos + (int64_t)row * D + c), b);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += b[i];
      v = pack8(a);
    }
  }
  *(u32x4*)(out + (int64_t)grow * D + c) = v;
}

// T5LayerNorm: y = w * x * rsqrt(mean(x^2) + eps), statistics in fp32; one wave per row, D % 8 == 0, D <= 8 * 64 * 8 = 4096
__global__ __launch_bounds__(256) void rms_norm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int rows,
                                                            int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + (int64_t)row * D;
  float v[8][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < D) {
      unpack8(*(const u32x4*)(xr + c), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  const float r = rsqrtf(ss / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < D) {
      float g[8];
      unpack8(*(const u32x4*)(w + c), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = g[j] * (v[i][j] * r);
      *(u32x4*)(y + (int64_t)row * D + c) = pack8(v[i]);
    }
  }
}

// T5DenseGatedActDense: out[r][c] = gelu_new(h[r][c]) * h[r][F + c]   (h = [rows][2F] from the fused wi_0 | wi_1 GEMM)
__global__ __launch_bounds__(256) void geglu_rows_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, int rows, int F) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int f8 = F >> 3;
  if (idx >= (int64_t)rows * f8) return;
  const int row = (int)(idx / f8), c = (int)(idx % f8) * 8;
  float a[8], b[8];
  unpack8(*(const u32x4*)(h + (int64_t)row * 2 * F + c), a);
  unpack8(*(const u32x4*)(h + (int64_t)row * 2 * F + F + c), b);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = gelu_tanh(a[i]) * b[i];
  *(u32x4*)(out + (int64_t)row * F + c) = pack8(a);
}

// CLIP's activation, in place: x * sigmoid(1.702 x)
__global__ __launch_bounds__(256) void quick_gelu_kernel(bf16_t* __restrict__ x, int64_t n8) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n8) return;
  float a[8];
  unpack8(*(const u32x4*)(x + idx * 8), a);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = a[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * a[i]));
  *(u32x4*)(x + idx * 8) = pack8(a);
}

__global__ __launch_bounds__(256) void fill_bf16_rows_kernel(bf16_t* __restrict__ p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = f2bf(v);
}

// ---- attention, head dim 64, <= 512 keys ----------------------------------------------------------------------------------------
constexpr int TA_MAXS = 512;
constexpr int TA_NT = TA_MAXS / 16;   // score tiles of 16 keys
static inline int ta_lds_bytes(int S_pad) { return S_pad * 128 + 64 * (S_pad * 2 + 16); }

// q [S_pad][ldq], k [S_pad][ldk] (this head's 64 columns start at head * 64), vt [heads * 64][ldvt] (V transposed: row = channel,
// column = key), bias [.][S_pad][S_pad] fp32 with head stride bias_hs (0: shared by the heads) or NULL, out [S][ldo].
// scores = scale * q.k + bias; S_pad % 32 == 0; rows / keys >= S must be masked by the bias (-inf) and are not written.
// blockIdx.z = sequence: its rows of q / k / out start at z * S_pad, its columns of vt at z * S_pad (out: z * S_pad as well).
template <int QI>   // query tiles per wave: a workgroup stages K / V^T once for 64 * QI queries
__global__ __launch_bounds__(256) void attn64_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk,
                                                     const bf16_t* __restrict__ vt, int64_t ldvt, const float* __restrict__ bias, int64_t bias_hs,
                                                     bf16_t* __restrict__ out, int64_t ldo, int S, int S_pad, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const kl = smem;                       // K rows: 128 B each, 16-byte chunk c of row r at chunk c ^ (r & 7)
  char* const vl = smem + S_pad * 128;         // V^T rows: pitch S_pad * 2 + 16 B (bank shift of 4 words per row), key slots permuted
  const int vpitch = S_pad * 2 + 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, qb = blockIdx.x;
  {
    const int64_t r0 = (int64_t)blockIdx.z * S_pad;
    q += r0 * ldq; k += r0 * ldk; vt += r0; out += r0 * ldo;
  }
  // ---- stage the head's K and V^T --------------------------------------------------------------------------------------------
  for (int idx = tid; idx < S_pad * 8; idx += 256) {
    const int r = idx >> 3, c = idx & 7;
    *(u32x4*)(kl + r * 128 + ((c ^ (r & 7)) << 4)) = *(const u32x4*)(k + (int64_t)r * ldk + head * 64 + c * 8);
  }
  // The PV product sums over the 32 keys of a block in ANY order as long as P^T and V^T agree.  A 16x16 score tile leaves keys
  // 4g .. 4g+3 of the tile in lane group g, so the B operand of a 32-key block holds, in its 8 slots of lane group g, keys
  // {4g..4g+3} of the block's first tile and {16+4g .. 16+4g+3} of its second: slot(key) = 8 (key%16 / 4) + 4 (key / 16) + key % 4.
  const int kc = S_pad >> 3;
  for (int idx = tid; idx < 64 * kc; idx += 256) {
    const int d = idx / kc, m = idx - d * kc;
    const u32x4 v = *(const u32x4*)(vt + (int64_t)(head * 64 + d) * ldvt + m * 8);
    const int blk = (m * 8) >> 5, kap = (m * 8) & 31;
    const int tile = kap >> 4, g0 = (kap & 15) >> 2;
    char* dst = vl + d * vpitch + blk * 64;
    u32x2 lo, hi;
    lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3];
    *(u32x2*)(dst + (8 * g0 + 4 * tile) * 2) = lo;
    *(u32x2*)(dst + (8 * (g0 + 1) + 4 * tile) * 2) = hi;
  }
  __syncthreads();
  const int nt = S_pad >> 4;
  const float LOG2E = 1.4426950408889634f;
  const float sl2 = scale * LOG2E;
#pragma unroll 1
  for (int it = 0; it < QI; ++it) {
  // ---- scores: S^T tile t = K[16t .. 16t+15] Q^T: this lane holds keys 16t + 4g + r of query l15 ---------------------------------
  const int q_row = (qb * QI + it) * 64 + w * 16 + l15;
  if ((qb * QI + it) * 64 >= S_pad) break;
  const int q_ld = q_row < S_pad ? q_row : S_pad - 1;
  bf16x8 qf[2];
#pragma unroll
  for (int ds = 0; ds < 2; ++ds) qf[ds] = *(const bf16x8*)(q + (int64_t)q_ld * ldq + head * 64 + ds * 32 + g * 8);
  const float* brow = bias ? bias + head * bias_hs + (int64_t)q_ld * S_pad + 4 * g : nullptr;
  f32x4 s[TA_NT];
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int t = 0; t < TA_NT; ++t) {
    if (t < nt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int r = t * 16 + l15;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const bf16x8 kf = *(const bf16x8*)(kl + r * 128 + (((ds * 4 + g) ^ (r & 7)) << 4));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ds], acc, 0, 0, 0);
      }
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (brow) b = *(const f32x4*)(brow + t * 16);
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) {
        acc[r2] = acc[r2] * sl2 + b[r2] * LOG2E;
        mx = fmaxf(mx, acc[r2]);
      }
      s[t] = acc;
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (!(mx > -__builtin_huge_valf())) mx = 0.f;   // a fully masked (padded) query row: keep the arithmetic finite
  // ---- P = exp2(s - max), row sums, O^T += V^T P^T per 32-key block ------------------------------------------------------------------
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
#pragma unroll
  for (int b = 0; b < TA_NT / 2; ++b) {
    if (2 * b < nt) {
      float p[8];
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) {
        p[r2] = __builtin_amdgcn_exp2f(s[2 * b][r2] - mx);
        p[4 + r2] = __builtin_amdgcn_exp2f(s[2 * b + 1][r2] - mx);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) l += p[i];
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pack8(p));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = *(const bf16x8*)(vl + (dt * 16 + l15) * vpitch + b * 64 + g * 16);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
      }
    }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (q_row < S) {
    const float inv = 1.0f / l;
    bf16_t* orow = out + (int64_t)q_row * ldo + head * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      u32x2 v;
      v[0] = pack2(oacc[dt][0] * inv, oacc[dt][1] * inv);
      v[1] = pack2(oacc[dt][2] * inv, oacc[dt][3] * inv);
      *(u32x2*)(orow + dt * 16) = v;
    }
  }
  }   // it
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct TextCtx {
  hipStream_t st;
  bf16_t *X, *Xn, *QK, *VT, *AO, *HID, *G, *ones;
  void* sk; int64_t sk_bytes;
  int S, S_pad, B;
};

struct TextSizes { int64_t x, qk, vt, ao, hid, g, total; };
constexpr int64_t TEXT_SK = 4096 + (64ll << 20);

static TextSizes text_sizes(int rows, int D, int inner, int F, int hid_cols) {   // rows = B * S_pad
  TextSizes z;
  z.x = round_up((int64_t)rows * D * 2, 256);
  z.qk = round_up((int64_t)rows * 2 * inner * 2, 256);
  z.vt = round_up((int64_t)inner * rows * 2, 256);
  z.ao = round_up((int64_t)rows * inner * 2, 256);
  z.hid = round_up((int64_t)rows * hid_cols * 2, 256);
  z.g = round_up((int64_t)rows * F * 2, 256);
  z.total = 2 * z.x + z.qk + z.vt + z.ao + z.hid + z.g + round_up(16384 * 2, 256) + TEXT_SK;
  return z;
}

static int text_ctx(TextCtx& c, const TextSizes& z, int B, int S, int S_pad, const rf_workspace* ws, hipStream_t st, const char* who) {
  RF_REQUIRE(ws && ws->base && aligned16(ws->base) && ws->bytes >= z.total, RF_ERR_WORKSPACE, "%s: workspace %lld < required %lld bytes", who,
             (long long)(ws ? ws->bytes : 0), (long long)z.total);
  char* p = (char*)ws->base;
  auto take = [&](int64_t bytes) { char* q = p; p += round_up(bytes, 256); return q; };
  c.st = st; c.S = S; c.S_pad = S_pad; c.B = B;
  c.X = (bf16_t*)take(z.x); c.Xn = (bf16_t*)take(z.x); c.QK = (bf16_t*)take(z.qk); c.VT = (bf16_t*)take(z.vt);
  c.AO = (bf16_t*)take(z.ao); c.HID = (bf16_t*)take(z.hid); c.G = (bf16_t*)take(z.g);
  c.ones = (bf16_t*)take(16384 * 2);
  c.sk = take(TEXT_SK); c.sk_bytes = TEXT_SK;
  hipLaunchKernelGGL(fill_bf16_rows_kernel, dim3(64), dim3(256), 0, st, c.ones, (int64_t)16384, 1.0f);
  RF_CHECK_HIP(hipMemsetAsync(c.sk, 0, 4096, st));                       // stream-K flags
  RF_CHECK_HIP(hipMemsetAsync(c.Xn, 0, (size_t)z.x, st));                // padded rows start finite (zero) and stay finite: they run through
  RF_CHECK_HIP(hipMemsetAsync(c.AO, 0, (size_t)z.ao, st));                // the same norms as real rows; no real row ever reads them (masked keys)
  RF_LAUNCH_CHECK();
  return RF_OK;
}

// y[M x N] = epi(x[M x K] . w[N x K]^T + b) (+ residual)
static int text_gemm(TextCtx& c, const bf16_t* x, int64_t ldx, const void* w, const void* b, int M, int N, int K, bf16_t* y, int64_t ldy, int epi,
                     const bf16_t* residual) {
  rf_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.N = N; d.num_groups = 1; d.epilogue = residual ? RF_EPI_GATE_RES : epi;
  d.splitk_ws = c.sk; d.splitk_ws_bytes = c.sk_bytes;
  rf_gemm_group& G = d.g[0];
  G.seg[0].A = x; G.seg[0].lda = ldx; G.seg[0].W = w; G.seg[0].ldw = K; G.seg[0].K = K;
  G.bias = b; G.M = M; G.out = y; G.ldo = ldy;
  if (residual) { G.gate = c.ones; G.residual = residual; G.ldr = ldy; }
  // A 512-token prompt is 2 rows of 256-tiles: 32 .. 160 output tiles with 64 .. 160 K-tiles each neither fill the 256 CUs with whole
  // tiles nor suit the 128-tile kernel AUTO picks below 200 tiles (417 TF): stream-K cuts the K-tile iterations evenly over the CUs
  {
    const int64_t t256 = (int64_t)cdiv(M, 256) * cdiv(N, 256);
    if (t256 >= 16 && t256 < 200 && K / 64 >= 16) d.schedule = RF_SCHED_STREAMK;
  }
  return rf_gemm_bf16(&d, c.st);
}

static int text_attention(TextCtx& c, int heads, int inner, const float* bias, int64_t bias_hs, float scale) {
  const int lds = ta_lds_bytes(c.S_pad);
  static bool attr = false;
  if (!attr) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ta_lds_bytes(TA_MAXS)));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ta_lds_bytes(TA_MAXS)));
    attr = true;
  }
  // a workgroup stages the head's K and V^T (2 x 64 KiB at 512 keys) for its queries: with enough (head, sequence) pairs to fill the
  // CUs it takes 256 queries instead of 64 and stages a quarter as often
  const bool wide = (int64_t)heads * c.B * cdiv(c.S_pad, 256) >= 256;
  ProfScope prof(RF_KC_ATTN, 4.0 * (double)c.S_pad * c.S_pad * 64.0 * heads * c.B, c.st);
  if (wide)
    hipLaunchKernelGGL(attn64_kernel<4>, dim3(cdiv(c.S_pad, 256), heads, c.B), dim3(256), lds, c.st, c.QK, (int64_t)2 * inner, c.QK + inner, (int64_t)2 * inner,
                       c.VT, (int64_t)c.B * c.S_pad, bias, bias_hs, c.AO, (int64_t)inner, c.S, c.S_pad, scale);
  else
    hipLaunchKernelGGL(attn64_kernel<1>, dim3(cdiv(c.S_pad, 64), heads, c.B), dim3(256), lds, c.st, c.QK, (int64_t)2 * inner, c.QK + inner, (int64_t)2 * inner,
                       c.VT, (int64_t)c.B * c.S_pad, bias, bias_hs, c.AO, (int64_t)inner, c.S, c.S_pad, scale);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

}  // namespace rf

using namespace rf;

static int t5_check(const rf_t5_weights* w, int32_t S, const char* who) {
  RF_REQUIRE(w && w->layer && w->embed && w->final_ln, RF_ERR_NULL, "%s: weights NULL", who);
  RF_REQUIRE(w->layers > 0 && w->heads > 0 && w->d_model > 0 && w->d_model % 64 == 0 && w->d_model <= 4096 && w->d_ff > 0 && w->d_ff % 64 == 0,
             RF_ERR_SHAPE, "%s: layers=%d heads=%d d_model=%d d_ff=%d (need d_model %% 64 == 0, <= 4096, d_ff %% 64 == 0)", who, w->layers, w->heads,
             w->d_model, w->d_ff);
  RF_REQUIRE(w->d_kv == 64, RF_ERR_UNSUPPORTED, "%s: the attention kernel is built for d_kv = 64 (got %d)", who, w->d_kv);
  RF_REQUIRE(S > 0 && S <= TA_MAXS, RF_ERR_SHAPE, "%s: S=%d (1 .. %d)", who, S, TA_MAXS);
  return RF_OK;
}

extern "C" int64_t rf_t5_workspace_bytes(const rf_t5_weights* w, int32_t B, int32_t S) {
  if (t5_check(w, S, "rf_t5_workspace_bytes") != RF_OK || B <= 0 || B > 64) return RF_ERR_SHAPE;
  const int S_pad = (int)round_up(S, 32);
  return text_sizes(B * S_pad, w->d_model, w->heads * 64, w->d_ff, 2 * w->d_ff).total;
}

// T5EncoderModel(ids)[0] for B sequences of S tokens: ids [B][S] int32 on the device -> out [B][S][ld_out] bf16 (final RMS norm
// applied).  pos_bias: [heads][S_pad][S_pad] fp32 for S_pad = round_up(S, 32) -- layer 0's bucketed relative-position bias, -inf in
// the columns of padded keys (built once per S by the binding: integer bucketing + one gather).  Internally every sequence owns
// S_pad rows; the projections and the FFN run over all B * S_pad rows at once.
extern "C" int rf_t5_encode(const rf_t5_weights* w, const int32_t* ids, int32_t B, int32_t S, void* out, int64_t ld_out, const rf_workspace* ws,
                            void* stream) {
  RF_TRY(t5_check(w, S, "rf_t5_encode"));
  RF_REQUIRE(B > 0 && B <= 64, RF_ERR_SHAPE, "rf_t5_encode: B=%d (1 .. 64)", B);
  RF_REQUIRE(ids && out && aligned16(out) && ld_out % 8 == 0 && ld_out >= w->d_model, RF_ERR_NULL, "rf_t5_encode: NULL / unaligned tensor");
  const int S_pad = (int)round_up(S, 32), D = w->d_model, inner = w->heads * 64, F = w->d_ff, R = B * S_pad;
  RF_REQUIRE(w->pos_bias && w->bias_S == S_pad, RF_ERR_SHAPE, "rf_t5_encode: pos_bias is built for S_pad = %d, this call needs %d", w->bias_S, S_pad);
  hipStream_t st = (hipStream_t)stream;
  TextCtx c;
  RF_TRY(text_ctx(c, text_sizes(R, D, inner, F, 2 * F), B, S, S_pad, ws, st, "rf_t5_encode"));
  {
    ProfScope prof(RF_KC_ROWOP, (double)R * D * 4, st);
    hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_for((int64_t)R * D / 8)), dim3(256), 0, st, (const bf16_t*)w->embed, (const bf16_t*)nullptr, ids, c.X, S,
                       S_pad, R, D, w->vocab);
    RF_LAUNCH_CHECK();
  }
  auto rms = [&](const void* g, const bf16_t* x, bf16_t* y) {
    ProfScope prof(RF_KC_ROWOP, (double)R * D * 4, st);
    hipLaunchKernelGGL(rms_norm_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, x, (const bf16_t*)g, y, R, D, w->eps);
    return hipGetLastError() == hipSuccess ? RF_OK : RF_ERR_HIP;
  };
  for (int i = 0; i < w->layers; ++i) {
    const rf_t5_layer& L = w->layer[i];
    RF_REQUIRE(L.ln0 && L.w_qk && L.w_v && L.w_o && L.ln1 && L.w_wi && L.w_wo, RF_ERR_NULL, "rf_t5_encode: layer %d has NULL weights", i);
    RF_TRY(rms(L.ln0, c.X, c.Xn));
    RF_TRY(text_gemm(c, c.Xn, D, L.w_qk, nullptr, R, 2 * inner, D, c.QK, 2 * inner, RF_EPI_STORE, nullptr));
    RF_TRY(text_gemm(c, (const bf16_t*)L.w_v, D, c.Xn, nullptr, inner, R, D, c.VT, R, RF_EPI_STORE, nullptr));   // V^T = W_v x^T
    RF_TRY(text_attention(c, w->heads, inner, w->pos_bias, (int64_t)S_pad * S_pad, 1.0f));                       // T5: no 1/sqrt(d)
    RF_TRY(text_gemm(c, c.AO, inner, L.w_o, nullptr, R, D, inner, c.X, D, RF_EPI_STORE, c.X));
    RF_TRY(rms(L.ln1, c.X, c.Xn));
    RF_TRY(text_gemm(c, c.Xn, D, L.w_wi, nullptr, R, 2 * F, D, c.HID, 2 * F, RF_EPI_STORE, nullptr));
    {
      ProfScope prof(RF_KC_ROWOP, (double)R * F * 6, st);
      hipLaunchKernelGGL(geglu_rows_kernel, dim3(grid_for((int64_t)R * F / 8)), dim3(256), 0, st, c.HID, c.G, R, F);
      RF_LAUNCH_CHECK();
    }
    RF_TRY(text_gemm(c, c.G, F, L.w_wo, nullptr, R, D, F, c.X, D, RF_EPI_STORE, c.X));
  }
  RF_TRY(rms(w->final_ln, c.X, c.Xn));
  // the real rows of every sequence -> the caller's [B][S][ld_out]
  for (int b = 0; b < B; ++b)
    RF_CHECK_HIP(hipMemcpy2DAsync((char*)out + (int64_t)b * S * ld_out * 2, (size_t)ld_out * 2, c.Xn + (int64_t)b * S_pad * D, (size_t)D * 2, (size_t)D * 2,
                                  (size_t)S, hipMemcpyDeviceToDevice, st));
  return RF_OK;
}

static int clip_check(const rf_clip_weights* w, int32_t S, const char* who) {
  RF_REQUIRE(w && w->layer && w->tok_embed && w->pos_embed && w->final_ln_scale && w->final_ln_shift, RF_ERR_NULL, "%s: weights NULL", who);
  RF_REQUIRE(w->layers > 0 && w->heads > 0 && w->hidden == w->heads * 64 && w->hidden % 64 == 0 && w->hidden <= 4096 && w->inter > 0 && w->inter % 64 == 0,
             RF_ERR_SHAPE, "%s: layers=%d heads=%d hidden=%d inter=%d (need hidden = 64 * heads <= 4096, inter %% 64 == 0)", who, w->layers, w->heads,
             w->hidden, w->inter);
  RF_REQUIRE(S > 0 && S <= w->max_pos && S <= TA_MAXS, RF_ERR_SHAPE, "%s: S=%d (1 .. min(max_pos = %d, %d))", who, S, w->max_pos, TA_MAXS);
  return RF_OK;
}

extern "C" int64_t rf_clip_text_workspace_bytes(const rf_clip_weights* w, int32_t B, int32_t S) {
  if (clip_check(w, S, "rf_clip_text_workspace_bytes") != RF_OK || B <= 0 || B > 64) return RF_ERR_SHAPE;
  const int S_pad = (int)round_up(S, 32);
  return text_sizes(B * S_pad, w->hidden, w->hidden, w->inter, w->inter).total;
}

// CLIPTextModel(ids) for B sequences of S tokens: last_hidden_state [B][S][hidden] (final LayerNorm applied; may be NULL) and
// pooler_output [B][hidden] = the final-normed row eos_pos[b] of sequence b (HOST array; the binding finds the positions: argmax(ids)
// for the legacy eos_token_id == 2 config, else the first eos_token_id).  mask: [S_pad][S_pad] fp32, 0 on and below the diagonal,
// -inf above it and in the columns of padded keys.  The binding folds 1/sqrt(64) into w_q / b_q, writes LayerNorm weights as
// (weight - 1, bias) pairs for rf_layernorm_modulate and folds v_proj.bias into the out-projection bias (softmax rows sum to 1).
extern "C" int rf_clip_text_encode(const rf_clip_weights* w, const int32_t* ids, int32_t B, int32_t S, const int32_t* eos_pos, void* last_hidden,
                                   void* pooled, const rf_workspace* ws, void* stream) {
  RF_TRY(clip_check(w, S, "rf_clip_text_encode"));
  RF_REQUIRE(B > 0 && B <= 64, RF_ERR_SHAPE, "rf_clip_text_encode: B=%d (1 .. 64)", B);
  RF_REQUIRE(ids && (last_hidden || pooled) && (eos_pos || !pooled), RF_ERR_NULL, "rf_clip_text_encode: NULL tensor");
  if (pooled)
    for (int b = 0; b < B; ++b)
      RF_REQUIRE(eos_pos[b] >= 0 && eos_pos[b] < S, RF_ERR_SHAPE, "rf_clip_text_encode: eos_pos[%d]=%d outside 0 .. %d", b, eos_pos[b], S - 1);
  const int S_pad = (int)round_up(S, 32), D = w->hidden, F = w->inter, R = B * S_pad;
  RF_REQUIRE(w->mask && w->mask_S == S_pad, RF_ERR_SHAPE, "rf_clip_text_encode: mask is built for S_pad = %d, this call needs %d", w->mask_S, S_pad);
  hipStream_t st = (hipStream_t)stream;
  TextCtx c;
  RF_TRY(text_ctx(c, text_sizes(R, D, D, F, F), B, S, S_pad, ws, st, "rf_clip_text_encode"));
  {
    ProfScope prof(RF_KC_ROWOP, (double)R * D * 6, st);
    hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_for((int64_t)R * D / 8)), dim3(256), 0, st, (const bf16_t*)w->tok_embed, (const bf16_t*)w->pos_embed,
                       ids, c.X, S, S_pad, R, D, w->vocab);
    RF_LAUNCH_CHECK();
  }
  for (int i = 0; i < w->layers; ++i) {
    const rf_clip_layer& L = w->layer[i];
    RF_REQUIRE(L.ln1_scale && L.ln1_shift && L.w_qk && L.b_qk && L.w_v && L.w_o && L.b_o && L.ln2_scale && L.ln2_shift && L.w_fc1 && L.b_fc1 && L.w_fc2 &&
                   L.b_fc2, RF_ERR_NULL, "rf_clip_text_encode: layer %d has NULL weights", i);
    RF_TRY(rf_layernorm_modulate(c.X, D, c.Xn, D, R, D, L.ln1_scale, L.ln1_shift, w->eps, st));
    RF_TRY(text_gemm(c, c.Xn, D, L.w_qk, L.b_qk, R, 2 * D, D, c.QK, 2 * D, RF_EPI_STORE, nullptr));
    RF_TRY(text_gemm(c, (const bf16_t*)L.w_v, D, c.Xn, nullptr, D, R, D, c.VT, R, RF_EPI_STORE, nullptr));
    RF_TRY(text_attention(c, w->heads, D, w->mask, 0, 1.0f));
    RF_TRY(text_gemm(c, c.AO, D, L.w_o, L.b_o, R, D, D, c.X, D, RF_EPI_STORE, c.X));
    RF_TRY(rf_layernorm_modulate(c.X, D, c.Xn, D, R, D, L.ln2_scale, L.ln2_shift, w->eps, st));
    RF_TRY(text_gemm(c, c.Xn, D, L.w_fc1, L.b_fc1, R, F, D, c.HID, F, RF_EPI_STORE, nullptr));
    {
      ProfScope prof(RF_KC_ROWOP, (double)R * F * 4, st);
      hipLaunchKernelGGL(quick_gelu_kernel, dim3(grid_for((int64_t)R * F / 8)), dim3(256), 0, st, c.HID, (int64_t)R * F / 8);
      RF_LAUNCH_CHECK();
    }
    RF_TRY(text_gemm(c, c.HID, F, L.w_fc2, L.b_fc2, R, D, F, c.X, D, RF_EPI_STORE, c.X));
  }
  RF_TRY(rf_layernorm_modulate(c.X, D, c.Xn, D, R, D, w->final_ln_scale, w->final_ln_shift, w->eps, st));
  for (int b = 0; b < B; ++b) {
    const bf16_t* src = c.Xn + (int64_t)b * S_pad * D;
    if (last_hidden) RF_CHECK_HIP(hipMemcpyAsync((char*)last_hidden + (int64_t)b * S * D * 2, src, (size_t)S * D * 2, hipMemcpyDeviceToDevice, st));
    if (pooled) RF_CHECK_HIP(hipMemcpyAsync((char*)pooled + (int64_t)b * D * 2, src + (int64_t)eos_pos[b] * D, (size_t)D * 2, hipMemcpyDeviceToDevice, st));
  }
  return RF_OK;
}
