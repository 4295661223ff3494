// Row kernels of the TRAINING path (train_flux/train/model.py:164-238 back-propagates through block.py's
// block_forward / single_block_forward).  HBM-bound: 16-byte accesses, fp32 math, one bf16 rounding on the way out; every
// column reduction (the modulation / gate gradients) is a fixed-order two-stage sum -> bit-reproducible, no atomics.
//
//   rf_qkv_train_fwd          raw q|k|v rows of the joint sequence -> the attention operands of forward AND backward:
//                             per-head RMSNorm + RoPE (block.py:38-41,60-67,74-78,92-99), q~ = q * softmax_scale * log2 e,
//                             head-major rows q~, k, v; the forward kernel's V^T tiles; the backward kernels' q~^T / k^T tiles
//   rf_qkv_train_bwd          (dq~, dk, dv) head-major -> d raw q|k|v token-major (RoPE^T, RMSNorm backward)
//   rf_layernorm_modulate_bwd y = LN(x) (1 + scale) + shift:  dx (+ residual gradient), d scale, d shift
//   rf_gate_bwd               y = res + gate o f:  df = gate o dy,  d gate = sum_rows dy o f
//   rf_gelu / rf_gelu_bwd     h = gelu_tanh(z);  dz = dh gelu_tanh'(z)
//   rf_transpose_bf16         [R][C] -> [C][R_pad] (zero padded): the token-contraction operands of the LoRA gradient GEMMs
#include "common.hpp"

namespace rf {

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sum8(float v) {   // over the 8 consecutive lanes that share a (token, head) row
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}
__device__ __forceinline__ void load16(const bf16_t* p, float (&f)[16]) {
  unpack8(*(const u32x4*)p, *reinterpret_cast<float(*)[8]>(&f[0]));
  unpack8(*(const u32x4*)(p + 8), *reinterpret_cast<float(*)[8]>(&f[8]));
}
__device__ __forceinline__ void store16(bf16_t* p, const float (&f)[16]) {
  *(u32x4*)p = pack8(*reinterpret_cast<const float(*)[8]>(&f[0]));
  *(u32x4*)(p + 8) = pack8(*reinterpret_cast<const float(*)[8]>(&f[8]));
}
__device__ __forceinline__ void load16f(const float* p, float (&f)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = *(const f32x4*)(p + 4 * j);
    f[4 * j] = v[0], f[4 * j + 1] = v[1], f[4 * j + 2] = v[2], f[4 * j + 3] = v[3];
  }
}

// ---- q|k|v -> attention operands -----------------------------------------------------------------------------------------------
// grid (s_pad / 32, heads), 256 threads: thread (token i = tid / 8, 16 channels c = tid % 8) of head blockIdx.y.
__global__ __launch_bounds__(256) void qkv_train_fwd_kernel(const bf16_t* __restrict__ raw, int64_t ld, int heads, int S, int s_pad,
                                                            int n_added, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
                                                            const bf16_t* __restrict__ waq, const bf16_t* __restrict__ wak,
                                                            const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                            float eps, float q_scale, bf16_t* __restrict__ q, bf16_t* __restrict__ k,
                                                            bf16_t* __restrict__ v, bf16_t* __restrict__ vt, bf16_t* __restrict__ qt,
                                                            bf16_t* __restrict__ kt) {
  __shared__ bf16_t tl[3][32][130];
  const int tid = threadIdx.x, head = blockIdx.y, t0 = blockIdx.x * 32;
  const int i = tid >> 3, c = tid & 7, tok = t0 + i, D = heads * 128;
  float o[3][16];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[a][e] = 0.f;
  if (tok < S) {
    const bf16_t* row = raw + (int64_t)tok * ld + head * 128 + c * 16;
    float cs[16], sn[16];
    load16f(cos_tab + (int64_t)tok * 128 + c * 16, cs);
    load16f(sin_tab + (int64_t)tok * 128 + c * 16, sn);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float x[16], w[16];
      load16(row + a * D, x);
      load16((tok < n_added ? (a ? wak : waq) : (a ? wk : wq)) + c * 16, w);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) ss += x[e] * x[e];
      const float rs = rsqrtf(sum8(ss) * (1.0f / 128.0f) + eps);
      const float sc = a == 0 ? q_scale : 1.0f;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float ya = x[e] * rs * w[e], yb = x[e + 1] * rs * w[e + 1];
        o[a][e] = (ya * cs[e] - yb * sn[e]) * sc;
        o[a][e + 1] = (yb * cs[e + 1] + ya * sn[e + 1]) * sc;
      }
    }
    load16(row + 2 * D, o[2]);
  }
  const int64_t hrow = ((int64_t)head * s_pad + tok) * 128 + c * 16;
  const bool bwd_ops = qt != nullptr;     // (v, qt, kt are given or omitted together: a forward that nobody differentiates writes half the bytes)
  store16(q + hrow, o[0]);
  store16(k + hrow, o[1]);
  if (bwd_ops) store16(v + hrow, o[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {          // (a compile-time trip count: o[][] stays in registers)
    if (a == 2 || bwd_ops) {
#pragma unroll
      for (int e = 0; e < 16; ++e) tl[a][i][c * 16 + e] = f2bf(o[a][e]);
    }
  }
  __syncthreads();
  const int blk = blockIdx.x;
  const int64_t tbase = ((int64_t)head * (s_pad >> 5) + blk) * (128 * 32);
  bf16_t* const vtile = vt + ((int64_t)head * (s_pad >> 6) + (blk >> 1)) * (128 * 64) + (blk & 1) * 32;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int d = (tid >> 2) + 64 * p, g = tid & 3;
    float a8[8], b8[8], v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // forward kernels' V^T: position = token with bits 2 and 3 swapped; positions 8 g .. 8 g + 7 of this half tile
      const int nv = 16 * (g >> 1) + 8 * (e >> 2) + 4 * (g & 1) + (e & 3);
      v8[e] = bf2f(tl[2][nv][d]);
    }
    *(u32x4*)(vtile + d * 64 + g * 8) = pack8(v8);
    if (bwd_ops) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = e < 4 ? 4 * g + e : 16 + 4 * g + e - 4;            // token of slot 8 g + e (attention_bwd.hip slot32)
        a8[e] = bf2f(tl[0][n][d]);
        b8[e] = bf2f(tl[1][n][d]);
      }
      *(u32x4*)(qt + tbase + d * 32 + g * 8) = pack8(a8);
      *(u32x4*)(kt + tbase + d * 32 + g * 8) = pack8(b8);
    }
  }
}

// ---- backward of the same map ----------------------------------------------------------------------------------------------------
// grid (ceil(S / 32), heads), 256 threads.
__global__ __launch_bounds__(256) void qkv_train_bwd_kernel(const bf16_t* __restrict__ raw, int64_t ld, int heads, int S, int s_pad,
                                                            int n_added, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
                                                            const bf16_t* __restrict__ waq, const bf16_t* __restrict__ wak,
                                                            const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                            float eps, float q_scale, const bf16_t* __restrict__ dq,
                                                            const bf16_t* __restrict__ dk, const bf16_t* __restrict__ dv,
                                                            bf16_t* __restrict__ draw, int64_t ldd) {
  const int tid = threadIdx.x, head = blockIdx.y;
  const int i = tid >> 3, c = tid & 7, tok = blockIdx.x * 32 + i, D = heads * 128;
  if (tok >= S) return;          // (whole 8-lane groups leave together: the shuffles below stay within a group)
  const bf16_t* row = raw + (int64_t)tok * ld + head * 128 + c * 16;
  bf16_t* drow = draw + (int64_t)tok * ldd + head * 128 + c * 16;
  const int64_t hrow = ((int64_t)head * s_pad + tok) * 128 + c * 16;
  float cs[16], sn[16];
  load16f(cos_tab + (int64_t)tok * 128 + c * 16, cs);
  load16f(sin_tab + (int64_t)tok * 128 + c * 16, sn);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    float x[16], w[16], go[16], dx[16];
    load16(row + a * D, x);
    load16((tok < n_added ? (a ? wak : waq) : (a ? wk : wq)) + c * 16, w);
    load16((a ? dk : dq) + hrow, go);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) ss += x[e] * x[e];
    const float rs = rsqrtf(sum8(ss) * (1.0f / 128.0f) + eps);
    const float sc = a == 0 ? q_scale : 1.0f;
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const float ga = go[e] * sc, gb = go[e + 1] * sc;
      // o[e] = ya cs[e] - yb sn[e],  o[e+1] = yb cs[e+1] + ya sn[e+1]
      const float dya = ga * cs[e] + gb * sn[e + 1], dyb = gb * cs[e + 1] - ga * sn[e];
      dx[e] = dya * w[e];            // d n
      dx[e + 1] = dyb * w[e + 1];
      dot += dx[e] * x[e] * rs + dx[e + 1] * x[e + 1] * rs;
    }
    const float mean = sum8(dot) * (1.0f / 128.0f);
#pragma unroll
    for (int e = 0; e < 16; ++e) dx[e] = rs * (dx[e] - x[e] * rs * mean);
    store16(drow + a * D, dx);
  }
  *(u32x4*)(drow + 2 * D) = *(const u32x4*)(dv + hrow);
  *(u32x4*)(drow + 2 * D + 8) = *(const u32x4*)(dv + hrow + 8);
}

// The four waves of a block add their per-lane column sums through LDS in wave order -- ((w0 + w1) + w2) + w3, bit-reproducible --
// and wave 3 stores the block's sums to dst[0 .. D).  sm: NCH * 512 floats.
template <int NCH>
__device__ __forceinline__ void block_colsum(float (&acc)[NCH][8], float* sm, float* __restrict__ dst, int D, int lane, int w) {
#pragma unroll
  for (int turn = 0; turn < 4; ++turn) {
    if (w == turn) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < D) {
          if (turn > 0) {
            const f32x4 a = *(const f32x4*)(sm + col), b = *(const f32x4*)(sm + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = a[j] + acc[c][j], acc[c][4 + j] = b[j] + acc[c][4 + j];
          }
          f32x4 a, b;
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = acc[c][j], b[j] = acc[c][4 + j];
          float* o = turn < 3 ? sm + col : dst + col;
          *(f32x4*)o = a, *(f32x4*)(o + 4) = b;
        }
      }
    }
    if (turn < 3) __syncthreads();
  }
}

// ---- LayerNorm + modulate, backward --------------------------------------------------------------------------------------------
// One wave per row, grid-stride over the rows; a lane keeps the column sums of ITS columns over ITS rows, the block adds its four
// waves (block_colsum) and writes partials[block][2][D] (d shift | d scale); colsum_reduce_kernel adds the blocks in index order.
// At D = 3072 the column sums alone are 96 registers per lane, so a SIMD holds ONE wave: the loads of a wave's next row (x, dy, dres
// as packed 16-byte registers) are issued before the arithmetic of the current one, and the grid covers every CU (round 4 ran 128
// blocks, each row's loads exposed: 90 us for 4096 x 3072 where the traffic is worth 25).
template <int NCH>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                                         int64_t lddy, const bf16_t* __restrict__ dres, int64_t lddres,
                                                         bf16_t* __restrict__ dx, int64_t lddx, int rows, int D,
                                                         const bf16_t* __restrict__ scale, float eps, float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float a_shift[NCH][8], a_scale[NCH][8], sc1[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0.f;
    if (col < D) unpack8(*(const u32x4*)(scale + col), t);
#pragma unroll
    for (int j = 0; j < 8; ++j) a_shift[c][j] = 0.f, a_scale[c][j] = 0.f, sc1[c][j] = 1.0f + t[j];
  }
  u32x4 xq[NCH], gq[NCH], rq[NCH];      // the row in flight, packed (columns >= D stay zero)
#pragma unroll
  for (int c = 0; c < NCH; ++c) xq[c] = gq[c] = rq[c] = u32x4{0u, 0u, 0u, 0u};
  auto fetch = [&](int row) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
        xq[c] = *(const u32x4*)(x + (int64_t)row * ldx + col);
        gq[c] = *(const u32x4*)(dy + (int64_t)row * lddy + col);
        if (dres != nullptr) rq[c] = *(const u32x4*)(dres + (int64_t)row * lddres + col);
      }
    }
  };
  if (wave < rows) fetch(wave);
  for (int row = wave; row < rows; row += nwaves) {
    float v[NCH][8], gy[NCH][8];
    u32x4 rcur[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      unpack8(xq[c], v[c]);
      unpack8(gq[c], gy[c]);
      rcur[c] = rq[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    }
    if (row + nwaves < rows) fetch(row + nwaves);
    const float mean = wave_sum_t(s) / (float)D;
    float qv = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dlt = v[c][j] - mean;
          qv += dlt * dlt;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum_t(qv) / (float)D + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (v[c][j] - mean) * rstd;
          a_shift[c][j] += gy[c][j];
          a_scale[c][j] += gy[c][j] * xh;
          const float dxh = gy[c][j] * sc1[c][j];
          v[c][j] = xh;
          gy[c][j] = dxh;
          m1 += dxh;
          m2 += dxh * xh;
        }
      }
    }
    m1 = wave_sum_t(m1) / (float)D;
    m2 = wave_sum_t(m2) / (float)D;
    bf16_t* orow = dx + (int64_t)row * lddx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
        float r8[8], o8[8];
        unpack8(rcur[c], r8);                 // zero without a residual gradient
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = r8[j] + rstd * (gy[c][j] - m1 - v[c][j] * m2);
        *(u32x4*)(orow + col) = pack8(o8);
      }
    }
  }
  __shared__ float sm[NCH * 512];
  float* pw = partials + (int64_t)blockIdx.x * 2 * D;
  block_colsum<NCH>(a_shift, sm, pw, D, lane, threadIdx.x >> 6);
  __syncthreads();                            // (wave 3 has read the last LDS image before the next one is written)
  block_colsum<NCH>(a_scale, sm, pw + D, D, lane, threadIdx.x >> 6);
}

// out[c] = sum_w partials[w][c] in a FIXED order: a block owns 64 columns; thread (c, j) adds the partial rows j, j + 4, j + 8, ... with
// four independent accumulators (the loads of a column are independent: the loop is latency-, not dependency-bound), then the four
// j-sums are combined as ((j0 + j1) + (j2 + j3)).  `cols` columns (2 D for the LayerNorm kernel, D for the gate kernel).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partials, int nw, int cols, float* __restrict__ out0,
                                                            float* __restrict__ out1, int split) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), j = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    const float* p = partials + c;
    int w = j;
    for (; w + 12 < nw; w += 16) {
      a0 += p[(int64_t)w * cols];
      a1 += p[(int64_t)(w + 4) * cols];
      a2 += p[(int64_t)(w + 8) * cols];
      a3 += p[(int64_t)(w + 12) * cols];
    }
    for (; w < nw; w += 4) a0 += p[(int64_t)w * cols];
  }
  part[j][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (j == 0 && c < cols) {
    const int t = threadIdx.x;
    const float s = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    if (c < split) out0[c] = s;
    else out1[c - split] = s;
  }
}

// ---- gated residual, backward ------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* __restrict__ dy, int64_t lddy, const bf16_t* __restrict__ f, int64_t ldf,
                                                       const bf16_t* __restrict__ gate, bf16_t* __restrict__ df, int64_t lddf, int rows, int D,
                                                       float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float acc[NCH][8], gt[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[c][j] = 0.f, gt[c][j] = 0.f;
    if (col < D) unpack8(*(const u32x4*)(gate + col), gt[c]);
  }
  for (int row = wave; row < rows; row += nwaves) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
        float g8[8], f8[8], o8[8];
        unpack8(*(const u32x4*)(dy + (int64_t)row * lddy + col), g8);
        unpack8(*(const u32x4*)(f + (int64_t)row * ldf + col), f8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[c][j] += g8[j] * f8[j];
          o8[j] = g8[j] * gt[c][j];
        }
        *(u32x4*)(df + (int64_t)row * lddf + col) = pack8(o8);
      }
    }
  }
  __shared__ float sm[NCH * 512];
  block_colsum<NCH>(acc, sm, partials + (int64_t)blockIdx.x * D, D, lane, threadIdx.x >> 6);
}

// y = res + gate o f (the training path keeps f = the projection's bf16 output for the gate gradient, so the gated residual is its
// own pass here instead of the RF_EPI_GATE_RES epilogue)
__global__ __launch_bounds__(256) void gate_res_kernel(const bf16_t* __restrict__ f, int64_t ldf, const bf16_t* __restrict__ gate,
                                                       const bf16_t* __restrict__ res, int64_t ldr, bf16_t* __restrict__ out, int64_t ldo,
                                                       int rows, int cols8) {
  const int64_t total = (int64_t)rows * cols8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cols8;
    const int c = (int)(idx - r * cols8) * 8;
    float fv[8], gv[8], rv[8], o[8];
    unpack8(*(const u32x4*)(f + r * ldf + c), fv);
    unpack8(*(const u32x4*)(gate + c), gv);
    unpack8(*(const u32x4*)(res + r * ldr + c), rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rv[j] + gv[j] * fv[j];
    *(u32x4*)(out + r * ldo + c) = pack8(o);
  }
}

// ---- GELU (tanh) and its derivative ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_tanh_grad(float z) {
  // gelu(z) = z s(z), s = sigmoid(2u), u = sqrt(2/pi) (z + 0.044715 z^3):  d/dz = s + z s (1 - s) 2 u'(z)
  const float z2 = z * z;
  const float e = __builtin_amdgcn_exp2f(z * (-2.3022082f - 0.1029432f * z2));      // exp(-2u)
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  const float du2 = 1.5957691216f * (1.0f + 0.134145f * z2);                        // 2 u'
  return s + z * s * (1.0f - s) * du2;
}
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const bf16_t* __restrict__ z, int64_t ldz, const bf16_t* __restrict__ dh, int64_t lddh,
                                                   bf16_t* __restrict__ out, int64_t ldo, int rows, int cols8) {
  const int64_t total = (int64_t)rows * cols8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cols8;
    const int c = (int)(idx - r * cols8) * 8;
    float zv[8], o[8];
    unpack8(*(const u32x4*)(z + r * ldz + c), zv);
    if (BWD) {
      float g[8];
      unpack8(*(const u32x4*)(dh + r * lddh + c), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = g[j] * gelu_tanh_grad(zv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = gelu_tanh(zv[j]);
    }
    *(u32x4*)(out + r * ldo + c) = pack8(o);
  }
}

// ---- transpose with zero padding ---------------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < rows_pad.  grid (ceil(rows_pad / 64), ceil(cols / 64)), 256 threads.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, int64_t lds_, int rows, int cols,
                                                        bf16_t* __restrict__ dst, int64_t ldd, int rows_pad) {
  __shared__ bf16_t t[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    t[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(int64_t)(r0 + r) * lds_ + c0 + c] : f2bf(0.f);
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int c = idx >> 6, r = idx & 63;
    if (c0 + c < cols && r0 + r < rows_pad) dst[(int64_t)(c0 + c) * ldd + r0 + r] = t[r][c];
  }
}

// The same map with 16-byte accesses on both sides (pointers 16-byte aligned, pitches / cols / rows_pad multiples of 8: every LoRA
// operand of the training step).  The 64 x 64 tile sits in LDS at a 144-byte row pitch; a thread gathers the 8 rows of one output run.
__global__ __launch_bounds__(256) void transpose_vec_kernel(const bf16_t* __restrict__ src, int64_t lds_, int rows, int cols,
                                                            bf16_t* __restrict__ dst, int64_t ldd, int rows_pad) {
  __shared__ __attribute__((aligned(16))) bf16_t t[64][72];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int idx = tid + 256 * p, r = idx >> 3, c = (idx & 7) * 8;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (r0 + r < rows && c0 + c < cols) v = *(const u32x4*)(src + (int64_t)(r0 + r) * lds_ + c0 + c);
    *(u32x4*)&t[r][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int idx = tid + 256 * p, c = idx >> 3, r = (idx & 7) * 8;
    if (c0 + c < cols && r0 + r < rows_pad) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[e] = (uint32_t)*(const uint16_t*)&t[r + 2 * e][c] | ((uint32_t)*(const uint16_t*)&t[r + 2 * e + 1][c] << 16);
      *(u32x4*)(dst + (int64_t)(c0 + c) * ldd + r0 + r) = u32x4{w[0], w[1], w[2], w[3]};
    }
  }
}

// ---- fused LoRA operands of sibling linears, and their gradients back to the factors -------------------------------------------------
struct LoraFuseTab {
  rf_lora_fuse_entry e[RF_LORA_FUSE_MAX];
  int n;
};
// thread = 8 consecutive elements of A_out [r_pad][K] (first r_pad * K / 8 threads) or of Bs_out [N][r_pad] (the rest)
__global__ __launch_bounds__(256) void lora_fuse_kernel(LoraFuseTab t, int K, int N, int r_pad, bf16_t* __restrict__ A_out,
                                                        bf16_t* __restrict__ B_out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, nA = (int64_t)r_pad * (K >> 3), nB = (int64_t)N * (r_pad >> 3);
  if (idx < nA) {
    const int row = (int)(idx / (K >> 3)), k = (int)(idx % (K >> 3)) * 8;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    for (int i = 0; i < t.n; ++i)
      if (row >= t.e[i].r0 && row < t.e[i].r0 + t.e[i].r) v = *(const u32x4*)((const bf16_t*)t.e[i].A + (int64_t)(row - t.e[i].r0) * K + k);
    *(u32x4*)(A_out + (int64_t)row * K + k) = v;
  } else if (idx < nA + nB) {
    const int64_t j = idx - nA;
    const int nrow = (int)(j / (r_pad >> 3)), c0 = (int)(j % (r_pad >> 3)) * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int i = 0; i < t.n; ++i) {
      const rf_lora_fuse_entry& E = t.e[i];
      if (nrow < E.n0 || nrow >= E.n0 + E.n || c0 + 8 <= E.r0 || c0 >= E.r0 + E.r) continue;
      const bf16_t* src = (const bf16_t*)E.B + (int64_t)(nrow - E.n0) * E.r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e - E.r0;
        if (c >= 0 && c < E.r) o[e] = bf2f(src[c]) * E.scaling;
      }
    }
    *(u32x4*)(B_out + (int64_t)nrow * r_pad + c0) = pack8(o);
  }
}
// thread = 8 consecutive elements of some dA_i row (first threads: r_total * K / 8 of them, rows in entry order) or ONE row segment of a
// dB_i (n_total rows x entries: the r columns of entry i in row n)
__global__ __launch_bounds__(256) void lora_unfuse_kernel(LoraFuseTab t, int K, int N, int r_pad, const bf16_t* __restrict__ dA,
                                                          const bf16_t* __restrict__ dB, int accumulate, int r_total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, nA = (int64_t)r_total * (K >> 3);
  if (idx < nA) {
    int row = (int)(idx / (K >> 3));
    const int k = (int)(idx % (K >> 3)) * 8;
    for (int i = 0; i < t.n; ++i) {
      const rf_lora_fuse_entry& E = t.e[i];
      if (row < E.r) {
        if (E.dA != nullptr) {
          bf16_t* dst = (bf16_t*)E.dA + (int64_t)row * K + k;
          const u32x4 g = *(const u32x4*)(dA + (int64_t)(E.r0 + row) * K + k);
          if (accumulate) {
            float a[8], b[8];
            unpack8(*(const u32x4*)dst, a);
            unpack8(g, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
            *(u32x4*)dst = pack8(a);
          } else {
            *(u32x4*)dst = g;
          }
        }
        return;
      }
      row -= E.r;
    }
    return;
  }
  // dB: item = (entry i, row n of its n rows); the r columns are contiguous in both tensors
  int64_t j = idx - nA;
  for (int i = 0; i < t.n; ++i) {
    const rf_lora_fuse_entry& E = t.e[i];
    if (j < E.n) {
      if (E.dB != nullptr) {
        bf16_t* dst = (bf16_t*)E.dB + j * E.r;
        const bf16_t* src = dB + (int64_t)(E.n0 + j) * r_pad + E.r0;
        for (int c = 0; c < E.r; ++c) {
          const float g = bf2f(f2bf(bf2f(src[c]) * E.scaling));          // the product rounded to bf16, as `dBs_block * scaling` is
          dst[c] = f2bf(accumulate ? bf2f(dst[c]) + g : g);
        }
      }
      return;
    }
    j -= E.n;
  }
}

// blocks of four row-waves; one fp32 partial row per BLOCK, so rf_train_partials_bytes' 512 rows bound the grid
static int row_blocks(int rows, int cap) {
  const int b = cdiv(rows, 4);
  return b < cap ? b : cap;
}


// ---- token-axis ("TN") skinny GEMM: the LoRA factor gradients -----------------------------------------------------------------
//   out[n][j] = sum_s big[s][n] * skinny[s][j]        big [S][ld_big] (N columns), skinny [S][ld_sk] (R <= 64 columns), both row-major bf16
// dB = dY^T T  and  dA^T = x^T dT  contract over the TOKENS, the one axis along which neither operand is contiguous.  Going through
// rf_gemm_bf16 meant writing dY^T and x^T out first (two full-size transposes per LoRA site and step: 6 % of a training step at the
// reference's shape).  Here the transpose happens on the way INTO the LDS: a thread's 16-byte load (8 columns of one token) is
// scattered as eight 2-byte writes into a [column][32 tokens] image, whose rows are then exactly the 16-byte MFMA fragments
// (16x16x32: lane (g, i) <- row i, tokens 8 g .. 8 g + 7; chunk g of row n sits at g ^ {0, 2, 3, 1}[(n >> 2) & 3]: conflict-free reads).
// grid (ceil(N / 128), chunks of TN_CHUNK tokens): fp32 partial tiles [chunk][N][R] -> tn_reduce_kernel sums them in chunk order
// (deterministic) and writes bf16, optionally transposed ([R][N]: dA).
constexpr int TN_COLS = 128, TN_CHUNK = 128;
// chunk g of a 64-byte image row n sits at position g ^ tn_swz((n >> 2) & 3), {0, 2, 3, 1}: conflict-free under ds_read_b128's
// NON-contiguous 16-lane groups (the plain g ^ ((n >> 2) & 3) is 2-way conflicted there; tests/test_lds_layouts_cpu.py holds the model)
__device__ __forceinline__ int tn_swz(int h) { return (0x78 >> (2 * h)) & 3; }
__device__ __forceinline__ f32x4 mfma16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
  return c;
#endif
}
// TR: the output is wanted transposed ([R][N]: dA) -- the MFMA operands swap roles, so a lane holds 4 rows j of ONE column n and the
// partial tiles are written (and later summed) as [chunk][R][N] with n along the lanes: coalesced on both sides.
template <int RT, bool TR>   // RT = R / 16 (1, 2, 4 or 8) column tiles of the skinny operand
__global__ __launch_bounds__(256) void tn_skinny_kernel(const bf16_t* __restrict__ big, int64_t ld_big, const bf16_t* __restrict__ sk,
                                                        int64_t ld_sk, float* __restrict__ ws, int S, int N) {
  constexpr int R = 16 * RT;
  __shared__ __attribute__((aligned(16))) char bigT[2][TN_COLS * 64];   // [128 columns][32 tokens] bf16, chunk-swizzled
  __shared__ __attribute__((aligned(16))) char skT[2][R * 64];          // [R columns][32 tokens]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * TN_COLS, s_begin = blockIdx.y * TN_CHUNK;
  const int s_end = min(S, s_begin + TN_CHUNK);
  const int nsteps = (s_end - s_begin + 31) >> 5;
  // loader roles: big tile = 32 tokens x 16 column-octets = 512 pieces, two per thread; skinny tile = 32 tokens x (R / 8) octets
  constexpr int SKP = (32 * (R / 8) + 255) / 256;   // skinny pieces per thread (2 at R = 128)
  u32x4 rb[2], rs[SKP];
  auto fetch = [&](const int st) {
    const int s0 = s_begin + st * 32;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + 256 * p, row = idx >> 4, c = idx & 15;
      const int n = n0 + c * 8;
      rb[p] = u32x4{0u, 0u, 0u, 0u};
      if (s0 + row < s_end && n < N) rb[p] = *(const u32x4*)(big + (int64_t)(s0 + row) * ld_big + n);   // N % 8 == 0
    }
#pragma unroll
    for (int p = 0; p < SKP; ++p) {
      const int idx = tid + 256 * p, row = idx / (R / 8), c = idx % (R / 8);
      rs[p] = u32x4{0u, 0u, 0u, 0u};
      if (idx < 32 * (R / 8) && s0 + row < s_end) rs[p] = *(const u32x4*)(sk + (int64_t)(s0 + row) * ld_sk + c * 8);
    }
  };
  auto put8 = [&](char* img, const u32x4 v, const int col0, const int tok) {   // columns col0 .. col0 + 7 of token tok
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = col0 + e;
      const uint32_t word = v[e >> 1];
      const uint16_t h = (e & 1) ? (uint16_t)(word >> 16) : (uint16_t)(word & 0xffffu);
      *(uint16_t*)(img + n * 64 + (((tok >> 3) ^ tn_swz((n >> 2) & 3)) << 4) + (tok & 7) * 2) = h;
    }
  };
  auto commit = [&](const int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + 256 * p;
      put8(bigT[buf], rb[p], (idx & 15) * 8, idx >> 4);
    }
#pragma unroll
    for (int p = 0; p < SKP; ++p) {
      const int idx = tid + 256 * p;
      if (idx < 32 * (R / 8)) put8(skT[buf], rs[p], (idx % (R / 8)) * 8, idx / (R / 8));
    }
  };
  f32x4 acc[2][RT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < RT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nsteps > 0) {
    fetch(0);
    commit(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) fetch(st + 1);
    bf16x8 bf[RT];
#pragma unroll
    for (int b = 0; b < RT; ++b) {
      const int j = 16 * b + l15;
      bf[b] = *(const bf16x8*)(skT[buf] + j * 64 + ((g ^ tn_swz((j >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int n = 16 * (2 * w + a) + l15;                    // this wave's column tiles 2 w, 2 w + 1
      const bf16x8 af = *(const bf16x8*)(bigT[buf] + n * 64 + ((g ^ tn_swz((n >> 2) & 3)) << 4));
#pragma unroll
      for (int b = 0; b < RT; ++b) acc[a][b] = TR ? mfma16x16x32(bf[b], af, acc[a][b]) : mfma16x16x32(af, bf[b], acc[a][b]);
    }
    if (st + 1 < nsteps) commit(buf ^ 1);
    __syncthreads();
  }
  if constexpr (TR) {
    // partial tile: lane (g, l15) holds out^T[j = 16 b + 4 g + r][n = 16 (2 w + a) + l15]
    float* dst = ws + (int64_t)blockIdx.y * R * N + n0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int n = 16 * (2 * w + a) + l15;
      if (n0 + n < N) {
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(int64_t)(16 * b + 4 * g + r) * N + n] = acc[a][b][r];
      }
    }
  } else {
    // partial tile: lane (g, j = l15) holds out[n = 16 (2 w + a) + 4 g + r][16 b + j]
    float* dst = ws + ((int64_t)blockIdx.y * N + n0) * R;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * (2 * w + a) + 4 * g + r;
        if (n0 + n < N) {
#pragma unroll
          for (int b = 0; b < RT; ++b) dst[(int64_t)n * R + 16 * b + l15] = acc[a][b][r];
        }
      }
  }
}

// out[n][j] (or out[j][n] when transposed) = bf16(sum over chunks, in chunk order).  The partials are [chunk][N][R], or [chunk][R][N] for a
// transposed output: either way thread idx reads element idx of every chunk and writes element idx of a contiguous output row.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int chunks, int N, int R, bf16_t* __restrict__ out,
                                                        int64_t ld_out, int transposed) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)N * R;
  if (idx >= total) return;
  float v = 0.f;
  for (int c = 0; c < chunks; ++c) v += ws[(int64_t)c * total + idx];
  const int inner = transposed ? N : R;
  out[(idx / inner) * ld_out + idx % inner] = f2bf(v);
}

}  // namespace rf

using namespace rf;

extern "C" int rf_qkv_train_fwd(const void* raw, int64_t ld_raw, int32_t heads, int32_t S, int32_t s_pad, int32_t n_added,
                                const void* w_q, const void* w_k, const void* w_added_q, const void* w_added_k, const float* cos_tab,
                                const float* sin_tab, float eps, float q_scale, void* q, void* k, void* v, void* vt, void* qt, void* kt,
                                void* stream) {
  RF_REQUIRE(raw && w_q && w_k && cos_tab && sin_tab && q && k && vt, RF_ERR_NULL, "rf_qkv_train_fwd: NULL operand");
  RF_REQUIRE((v != nullptr) == (qt != nullptr) && (qt != nullptr) == (kt != nullptr), RF_ERR_NULL,
             "rf_qkv_train_fwd: v, qt, kt (the backward's operands) are given or omitted together");
  RF_REQUIRE(n_added == 0 || (w_added_q && w_added_k), RF_ERR_NULL, "rf_qkv_train_fwd: text rows need norm_added_q / norm_added_k");
  RF_REQUIRE(heads > 0 && S > 0 && s_pad >= S && s_pad % 64 == 0 && ld_raw % 8 == 0 && ld_raw >= 3 * heads * 128, RF_ERR_SHAPE,
             "rf_qkv_train_fwd: heads=%d S=%d s_pad=%d ld=%lld", heads, S, s_pad, (long long)ld_raw);
  RF_REQUIRE(aligned16(raw) && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(vt) && aligned16(qt) && aligned16(kt),
             RF_ERR_ALIGN, "rf_qkv_train_fwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)S * heads * 128.0 * 2.0 * (3.0 + (qt ? 6.0 : 3.0)), st);
  hipLaunchKernelGGL(qkv_train_fwd_kernel, dim3(s_pad / 32, heads), dim3(256), 0, st, (const bf16_t*)raw, ld_raw, heads, S, s_pad, n_added,
                     (const bf16_t*)w_q, (const bf16_t*)w_k, (const bf16_t*)w_added_q, (const bf16_t*)w_added_k, cos_tab, sin_tab, eps,
                     q_scale == 0.f ? 1.0f : q_scale, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, (bf16_t*)vt, (bf16_t*)qt, (bf16_t*)kt);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_qkv_train_bwd(const void* raw, int64_t ld_raw, int32_t heads, int32_t S, int32_t s_pad, int32_t n_added,
                                const void* w_q, const void* w_k, const void* w_added_q, const void* w_added_k, const float* cos_tab,
                                const float* sin_tab, float eps, float q_scale, const void* dq, const void* dk, const void* dv,
                                void* d_raw, int64_t ld_draw, void* stream) {
  RF_REQUIRE(raw && w_q && w_k && cos_tab && sin_tab && dq && dk && dv && d_raw, RF_ERR_NULL, "rf_qkv_train_bwd: NULL operand");
  RF_REQUIRE(n_added == 0 || (w_added_q && w_added_k), RF_ERR_NULL, "rf_qkv_train_bwd: text rows need norm_added_q / norm_added_k");
  RF_REQUIRE(heads > 0 && S > 0 && s_pad >= S && s_pad % 64 == 0 && ld_raw % 8 == 0 && ld_draw % 8 == 0, RF_ERR_SHAPE,
             "rf_qkv_train_bwd: heads=%d S=%d s_pad=%d", heads, S, s_pad);
  RF_REQUIRE(aligned16(raw) && aligned16(dq) && aligned16(dk) && aligned16(dv) && aligned16(d_raw), RF_ERR_ALIGN,
             "rf_qkv_train_bwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)S * heads * 128.0 * 2.0 * (2.0 + 3.0 + 3.0), st);
  hipLaunchKernelGGL(qkv_train_bwd_kernel, dim3(cdiv(S, 32), heads), dim3(256), 0, st, (const bf16_t*)raw, ld_raw, heads, S, s_pad, n_added,
                     (const bf16_t*)w_q, (const bf16_t*)w_k, (const bf16_t*)w_added_q, (const bf16_t*)w_added_k, cos_tab, sin_tab, eps,
                     q_scale == 0.f ? 1.0f : q_scale, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, (bf16_t*)d_raw, ld_draw);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int64_t rf_gemm_tn_skinny_ws_bytes(int32_t S, int32_t N, int32_t R) {
  return (int64_t)((S + rf::TN_CHUNK - 1) / rf::TN_CHUNK) * N * R * 4;
}

extern "C" int rf_gemm_tn_skinny(const void* big, int64_t ld_big, const void* skinny, int64_t ld_sk, void* out, int64_t ld_out,
                                 int32_t S, int32_t N, int32_t R, int32_t transposed, float* ws, int64_t ws_bytes, void* stream) {
  using namespace rf;
  RF_REQUIRE(big && skinny && out && ws, RF_ERR_NULL, "rf_gemm_tn_skinny: NULL operand");
  RF_REQUIRE(S > 0 && N > 0 && N % 8 == 0 && R > 0 && R % 16 == 0 && ld_big % 8 == 0 && ld_sk % 8 == 0 && ld_big >= N && ld_sk >= R &&
                 ld_out >= (transposed ? N : R),
             RF_ERR_SHAPE, "rf_gemm_tn_skinny: S=%d N=%d R=%d (N %% 8 == 0, R %% 16 == 0)", S, N, R);
  RF_REQUIRE(aligned16(big) && aligned16(skinny), RF_ERR_ALIGN, "rf_gemm_tn_skinny: 16-byte alignment");
  RF_REQUIRE(ws_bytes >= rf_gemm_tn_skinny_ws_bytes(S, N, R), RF_ERR_WORKSPACE, "rf_gemm_tn_skinny: scratch needs %lld bytes",
             (long long)rf_gemm_tn_skinny_ws_bytes(S, N, R));
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_GEMM_SMALL, 2.0 * S * (double)N * R, st);
  const int chunks = cdiv(S, TN_CHUNK);
  const dim3 grid(cdiv(N, TN_COLS), chunks);
  // the skinny operand in slices of 128 / 64 / 32 / 16 columns (a fused q | k | v | mlp LoRA has R = 128); one launch pair per slice
  for (int j0 = 0; j0 < R;) {
    const int rem = R - j0, Rs = rem >= 128 ? 128 : rem >= 64 ? 64 : rem >= 32 ? 32 : 16;
    const bf16_t* sks = (const bf16_t*)skinny + j0;
#define RF_TN(RT_)                                                                                                                \
  do {                                                                                                                            \
    if (transposed) hipLaunchKernelGGL((tn_skinny_kernel<RT_, true>), grid, dim3(256), 0, st, (const bf16_t*)big, ld_big, sks, ld_sk, ws, S, N);  \
    else hipLaunchKernelGGL((tn_skinny_kernel<RT_, false>), grid, dim3(256), 0, st, (const bf16_t*)big, ld_big, sks, ld_sk, ws, S, N);            \
  } while (0)
    if (Rs == 128) RF_TN(8);
    else if (Rs == 64) RF_TN(4);
    else if (Rs == 32) RF_TN(2);
    else RF_TN(1);
#undef RF_TN
    RF_LAUNCH_CHECK();
    bf16_t* outs = (bf16_t*)out + (transposed ? (int64_t)j0 * ld_out : (int64_t)j0);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(cdiv((int)((int64_t)N * Rs), 256)), dim3(256), 0, st, (const float*)ws, chunks, N, Rs, outs,
                       ld_out, transposed);
    RF_LAUNCH_CHECK();
    j0 += Rs;
  }
  return RF_OK;
}

extern "C" int64_t rf_train_partials_bytes(int32_t D) { return (int64_t)512 * 2 * D * 4; }

extern "C" int rf_layernorm_modulate_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* dres, int64_t lddres,
                                         void* dx, int64_t lddx, int32_t rows, int32_t D, const void* scale, float eps, float* d_scale,
                                         float* d_shift, float* partials, int64_t partials_bytes, void* stream) {
  RF_REQUIRE(x && dy && dx && scale && partials && ((d_scale == nullptr) == (d_shift == nullptr)), RF_ERR_NULL,
             "rf_layernorm_modulate_bwd: NULL operand (d_scale and d_shift are given or omitted together)");
  RF_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 6 * 512 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && lddres % 8 == 0,
             RF_ERR_SHAPE, "rf_layernorm_modulate_bwd: rows=%d D=%d (D %% 8 == 0, D <= 3072)", rows, D);
  RF_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(scale) && (dres == nullptr || aligned16(dres)), RF_ERR_ALIGN,
             "rf_layernorm_modulate_bwd: 16-byte alignment");
  const int nch = cdiv(D, 512);
  const int blocks = row_blocks(rows, nch > 4 ? 256 : 512), nw = blocks;   // nch > 4: one wave per SIMD, one block per CU
  RF_REQUIRE(partials_bytes >= (int64_t)nw * 2 * D * 4, RF_ERR_WORKSPACE, "rf_layernorm_modulate_bwd: partials need %lld bytes",
             (long long)nw * 2 * D * 4);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)rows * D * 2.0 * (dres ? 4.0 : 3.0), st);
#define RF_LNB(N)                                                                                                                    \
  hipLaunchKernelGGL(ln_mod_bwd_kernel<N>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy,           \
                     (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, rows, D, (const bf16_t*)scale, eps, partials)
  if (nch <= 1) RF_LNB(1);
  else if (nch <= 2) RF_LNB(2);
  else if (nch <= 4) RF_LNB(4);
  else RF_LNB(6);
#undef RF_LNB
  RF_LAUNCH_CHECK();
  if (d_scale != nullptr) {   // (the modulation rows of a stream without LoRA on its AdaLN linear need no gradient)
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(2 * D, 64)), dim3(256), 0, st, (const float*)partials, nw, 2 * D, d_shift, d_scale, D);
    RF_LAUNCH_CHECK();
  }
  return RF_OK;
}

extern "C" int rf_gate_bwd(const void* dy, int64_t lddy, const void* f, int64_t ldf, const void* gate, void* df, int64_t lddf, int32_t rows,
                           int32_t D, float* d_gate, float* partials, int64_t partials_bytes, void* stream) {
  RF_REQUIRE(dy && f && gate && df && partials, RF_ERR_NULL, "rf_gate_bwd: NULL operand");   // d_gate NULL: the column sum is not wanted
  RF_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 6 * 512 && lddy % 8 == 0 && ldf % 8 == 0 && lddf % 8 == 0, RF_ERR_SHAPE,
             "rf_gate_bwd: rows=%d D=%d", rows, D);
  RF_REQUIRE(aligned16(dy) && aligned16(f) && aligned16(gate) && aligned16(df), RF_ERR_ALIGN, "rf_gate_bwd: 16-byte alignment");
  const int blocks = row_blocks(rows, 512), nw = blocks;
  RF_REQUIRE(partials_bytes >= (int64_t)nw * D * 4, RF_ERR_WORKSPACE, "rf_gate_bwd: partials need %lld bytes", (long long)nw * D * 4);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)rows * D * 2.0 * 3.0, st);
  const int nch = cdiv(D, 512);
#define RF_GB(N)                                                                                                                    \
  hipLaunchKernelGGL(gate_bwd_kernel<N>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)f, ldf,            \
                     (const bf16_t*)gate, (bf16_t*)df, lddf, rows, D, partials)
  if (nch <= 1) RF_GB(1);
  else if (nch <= 2) RF_GB(2);
  else if (nch <= 4) RF_GB(4);
  else RF_GB(6);
#undef RF_GB
  RF_LAUNCH_CHECK();
  if (d_gate != nullptr) {
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(D, 64)), dim3(256), 0, st, (const float*)partials, nw, D, d_gate, d_gate, D);
    RF_LAUNCH_CHECK();
  }
  return RF_OK;
}

static int gelu_launch(bool bwd, const void* z, int64_t ldz, const void* dh, int64_t lddh, void* out, int64_t ldo, int32_t rows, int32_t cols,
                       void* stream) {
  RF_REQUIRE(z && out && (!bwd || dh), RF_ERR_NULL, "rf_gelu: NULL operand");
  RF_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ldz % 8 == 0 && ldo % 8 == 0 && lddh % 8 == 0, RF_ERR_SHAPE, "rf_gelu: rows=%d cols=%d",
             rows, cols);
  RF_REQUIRE(aligned16(z) && aligned16(out) && (!bwd || aligned16(dh)), RF_ERR_ALIGN, "rf_gelu: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)rows * cols * 2.0 * (bwd ? 3.0 : 2.0), st);
  const int64_t total = (int64_t)rows * (cols / 8);
  const int blocks = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  if (bwd)
    hipLaunchKernelGGL(gelu_kernel<true>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, ldz, (const bf16_t*)dh, lddh, (bf16_t*)out, ldo,
                       rows, cols / 8);
  else
    hipLaunchKernelGGL(gelu_kernel<false>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, ldz, (const bf16_t*)nullptr, 0, (bf16_t*)out, ldo,
                       rows, cols / 8);
  RF_LAUNCH_CHECK();
  return RF_OK;
}
extern "C" int rf_gelu(const void* z, int64_t ldz, void* h, int64_t ldh, int32_t rows, int32_t cols, void* stream) {
  return gelu_launch(false, z, ldz, nullptr, 0, h, ldh, rows, cols, stream);
}
extern "C" int rf_gelu_bwd(const void* z, int64_t ldz, const void* dh, int64_t lddh, void* dz, int64_t lddz, int32_t rows, int32_t cols,
                           void* stream) {
  return gelu_launch(true, z, ldz, dh, lddh, dz, lddz, rows, cols, stream);
}

extern "C" int rf_gate_residual(const void* f, int64_t ldf, const void* gate, const void* res, int64_t ldr, void* out, int64_t ldo, int32_t rows,
                                int32_t D, void* stream) {
  RF_REQUIRE(f && gate && res && out, RF_ERR_NULL, "rf_gate_residual: NULL operand");
  RF_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && ldf % 8 == 0 && ldr % 8 == 0 && ldo % 8 == 0, RF_ERR_SHAPE, "rf_gate_residual: rows=%d D=%d", rows, D);
  RF_REQUIRE(aligned16(f) && aligned16(gate) && aligned16(res) && aligned16(out), RF_ERR_ALIGN, "rf_gate_residual: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)rows * D * 2.0 * 3.0, st);
  const int64_t total = (int64_t)rows * (D / 8);
  const int blocks = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  hipLaunchKernelGGL(gate_res_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)f, ldf, (const bf16_t*)gate, (const bf16_t*)res, ldr,
                     (bf16_t*)out, ldo, rows, D / 8);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_transpose_bf16(const void* src, int64_t ld_src, int32_t rows, int32_t cols, void* dst, int64_t ld_dst, int32_t rows_pad,
                                 void* stream) {
  RF_REQUIRE(src && dst, RF_ERR_NULL, "rf_transpose_bf16: NULL operand");
  RF_REQUIRE(rows > 0 && cols > 0 && rows_pad >= rows && ld_dst >= rows_pad && ld_src >= cols, RF_ERR_SHAPE,
             "rf_transpose_bf16: rows=%d cols=%d rows_pad=%d", rows, cols, rows_pad);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(RF_KC_ROWOP, (double)rows * cols * 4.0, st);
  const dim3 grid(cdiv(rows_pad, 64), cdiv(cols, 64));
  if (aligned16(src) && aligned16(dst) && ld_src % 8 == 0 && ld_dst % 8 == 0 && cols % 8 == 0 && rows_pad % 8 == 0)
    hipLaunchKernelGGL(transpose_vec_kernel, grid, dim3(256), 0, st, (const bf16_t*)src, ld_src, rows, cols, (bf16_t*)dst, ld_dst, rows_pad);
  else
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, (const bf16_t*)src, ld_src, rows, cols, (bf16_t*)dst, ld_dst, rows_pad);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

static int lora_fuse_tab(const char* who, const rf_lora_fuse_entry* entries, int32_t n_entries, int32_t K, int32_t N, int32_t r_pad,
                         rf::LoraFuseTab& t, bool grads) {
  RF_REQUIRE(entries != nullptr, RF_ERR_NULL, "%s: NULL entries", who);
  RF_REQUIRE(n_entries >= 1 && n_entries <= RF_LORA_FUSE_MAX && K > 0 && K % 8 == 0 && N > 0 && r_pad > 0 && r_pad % 8 == 0, RF_ERR_SHAPE,
             "%s: n_entries=%d K=%d N=%d r_pad=%d (1..%d entries, K %% 8 == 0, r_pad %% 8 == 0)", who, n_entries, K, N, r_pad, RF_LORA_FUSE_MAX);
  t.n = n_entries;
  for (int i = 0; i < n_entries; ++i) {
    const rf_lora_fuse_entry& E = entries[i];
    RF_REQUIRE(E.r > 0 && E.n > 0 && E.r0 >= 0 && E.n0 >= 0 && E.r0 + E.r <= r_pad && E.n0 + E.n <= N, RF_ERR_SHAPE,
               "%s: entry %d (n0=%d n=%d r0=%d r=%d) outside [N=%d][r_pad=%d]", who, i, E.n0, E.n, E.r0, E.r, N, r_pad);
    for (int j = 0; j < i; ++j)
      RF_REQUIRE(E.r0 >= entries[j].r0 + entries[j].r || entries[j].r0 >= E.r0 + E.r, RF_ERR_SHAPE, "%s: entries %d and %d share rank columns", who, j, i);
    if (grads) {
      RF_REQUIRE(E.dA == nullptr || rf::aligned16(E.dA), RF_ERR_ALIGN, "%s: entry %d: dA must be 16-byte aligned", who, i);
    } else {
      RF_REQUIRE(E.A != nullptr && E.B != nullptr, RF_ERR_NULL, "%s: entry %d: NULL factor", who, i);
      RF_REQUIRE(rf::aligned16(E.A), RF_ERR_ALIGN, "%s: entry %d: A must be 16-byte aligned", who, i);
    }
    t.e[i] = E;
  }
  return RF_OK;
}

extern "C" int rf_lora_fuse(const rf_lora_fuse_entry* entries, int32_t n_entries, int32_t K, int32_t N, int32_t r_pad, void* A_out,
                            void* Bs_out, void* stream) {
  rf::LoraFuseTab t;
  if (int rc = lora_fuse_tab("rf_lora_fuse", entries, n_entries, K, N, r_pad, t, false)) return rc;
  RF_REQUIRE(A_out && Bs_out, RF_ERR_NULL, "rf_lora_fuse: NULL output");
  RF_REQUIRE(aligned16(A_out) && aligned16(Bs_out), RF_ERR_ALIGN, "rf_lora_fuse: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = (int64_t)r_pad * (K >> 3) + (int64_t)N * (r_pad >> 3);
  ProfScope prof(RF_KC_ROWOP, 2.0 * 2.0 * ((double)r_pad * K + (double)N * r_pad), st);
  hipLaunchKernelGGL(lora_fuse_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, st, t, K, N, r_pad, (bf16_t*)A_out, (bf16_t*)Bs_out);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

extern "C" int rf_lora_unfuse_grads(const rf_lora_fuse_entry* entries, int32_t n_entries, int32_t K, int32_t N, int32_t r_pad, const void* dA,
                                    const void* dBs, int32_t accumulate, void* stream) {
  rf::LoraFuseTab t;
  if (int rc = lora_fuse_tab("rf_lora_unfuse_grads", entries, n_entries, K, N, r_pad, t, true)) return rc;
  RF_REQUIRE(dA && dBs, RF_ERR_NULL, "rf_lora_unfuse_grads: NULL gradient");
  RF_REQUIRE(aligned16(dA), RF_ERR_ALIGN, "rf_lora_unfuse_grads: 16-byte alignment");
  int r_total = 0;
  int64_t n_total = 0;
  for (int i = 0; i < n_entries; ++i) r_total += entries[i].r, n_total += entries[i].n;
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = (int64_t)r_total * (K >> 3) + n_total;
  ProfScope prof(RF_KC_ROWOP, 2.0 * 3.0 * ((double)r_total * K + (double)n_total * 8), st);
  hipLaunchKernelGGL(lora_unfuse_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, st, t, K, N, r_pad, (const bf16_t*)dA,
                     (const bf16_t*)dBs, accumulate, r_total);
  RF_LAUNCH_CHECK();
  return RF_OK;
}
