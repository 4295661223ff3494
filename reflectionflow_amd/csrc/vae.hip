// FLUX VAE (AutoencoderKL, 16 latent channels, 8x) decode / encode for gfx950 behind the C ABI.
// Replaces the diffusers modules the reference calls either side of the denoise loop:
//   decode: generate.py:302-307 (vae.decode(latents / scaling + shift)), runner hand-off tts_reflectionflow.py:273-279
//   encode: pipeline_tools.py:7-14 (vae.encode(images).latent_dist), condition.py:96-132
//
// Layout: every activation is a ZERO-HALO NHWC image, bf16: rows = the (H+2) x (W+2) padded pixels, C channels each.  With
// that, a 3x3 / stride 1 / pad 1 convolution IS the library's grouped GEMM with three K-segments and no new MFMA code: for
// output pixel p (padded index) the taps (dy, -1..+1) are 3C CONTIGUOUS values starting at pixel p + dy(W+2) - 1, so
//       out[p, :] = sum_dy  A_dy[p, 0:3C] . W[:, dy, 0:3C]^T,      A_dy = X + ((dy+1)(W+2)) C,  lda = C  (rows overlap)
// over the rows p = (W+3) .. last interior pixel: M = (H-1)(W+2) + W rows, weights repacked once to [Cout][3][3][Cin].
// The two halo COLUMNS inside that row range receive finite garbage; nobody reads it: GroupNorm / upsample / im2col skip
// halo pixels and re-write clean zeros into the halo of everything a convolution will read.  Residual adds ride in the GEMM
// epilogue (RF_EPI_GATE_RES with a gate of ones), the 1x1 shortcuts are plain GEMMs.
// Row kernels here (all HBM-bound, 16-byte accesses): GroupNorm statistics (shifted sums per block, fp64 finalisation) and
// apply (+ SiLU; padded or compact output), nearest 2x upsample, stride-2 im2col for the three encoder downsamplers,
// compact -> padded residual add.  The mid-block attention (ONE head of 512 channels) is a flash kernel of its own
// (vae_attn_kernel: 16x16x32 MFMAs, K / V^T tiles through XOR-swizzled / padded LDS, online softmax in fp32).
#include "common.hpp"

namespace rf {

// ---- GroupNorm ------------------------------------------------------------------------------------------------------------
// pass 1: per block, per channel: sum (x - s) and sum (x - s)^2 over its interior pixels; s = the channel's value at the
// first interior pixel (a shift: keeps E[x^2] - mean^2 away from cancellation when |mean| >> std)
template <int TPP>   // threads per pixel = C / 8
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, int H, int W, int C, int pix_per_block,
                                                       float* __restrict__ part /* [grid][C][2] */) {
  constexpr int PPI = 256 / TPP;              // pixels per iteration
  __shared__ float red[2][PPI][TPP * 8 + 1];
  const int cs = threadIdx.x % TPP, pl = threadIdx.x / TPP;
  const int Wp = W + 2;
  const int64_t npix = (int64_t)(H + 2) * Wp;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  float sh[8];
  unpack8(*(const u32x4*)(x + (int64_t)(Wp + 1) * C + cs * 8), sh);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = 0.f, s2[e] = 0.f;
  auto inside = [&](const int64_t p) {
    const int yp = (int)(p / Wp), xp = (int)(p - (int64_t)yp * Wp);
    return p < p1 && yp >= 1 && yp <= H && xp >= 1 && xp <= W;
  };
  auto acc = [&](const u32x4 raw) {
    float v[8];
    unpack8(raw, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[e] - sh[e];
      s1[e] += d;
      s2[e] += d * d;
    }
  };
  for (int64_t p = p0 + pl; p < p1; p += 4 * PPI) {     // four independent 16-byte loads in flight per thread
    u32x4 r[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      in[u] = inside(p + u * PPI);
      if (in[u]) r[u] = *(const u32x4*)(x + (p + u * PPI) * C + cs * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (in[u]) acc(r[u]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][pl][cs * 8 + e] = s1[e];
    red[1][pl][cs * 8 + e] = s2[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < PPI; ++i) a += red[0][i][c], b += red[1][i][c];
    part[((int64_t)blockIdx.x * C + c) * 2 + 0] = a;
    part[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

// pass 2 (one workgroup per GROUP): block partials -> the group's mean / rstd (fp64) -> per-channel affine y = x * a[c] + b[c].
// 256 threads = cpg channel lanes x (256 / cpg) block lanes; fixed-order tree reductions (bit-reproducible).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int nblk, const bf16_t* __restrict__ x_first,
                                                          int C, int groups, double count, float eps,
                                                          const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                          float* __restrict__ ab /* [2][C] */) {
  __shared__ double r1[256], r2[256];
  const int cpg = C / groups;                  // 2, 4, 8 or 16
  const int g = blockIdx.x;
  const int tpc = 256 / cpg;                   // threads per channel
  const int ci = threadIdx.x / tpc, bl = threadIdx.x % tpc;
  const int c = g * cpg + ci;
  double a = 0.0, b = 0.0;
  for (int i = bl; i < nblk; i += tpc) {
    const float* q = part + ((int64_t)i * C + c) * 2;
    a += (double)q[0];
    b += (double)q[1];
  }
  r1[threadIdx.x] = a;
  r2[threadIdx.x] = b;
  __syncthreads();
  for (int o = tpc >> 1; o >= 1; o >>= 1) {    // per-channel segments are power-of-two aligned
    if (bl < o) {
      r1[threadIdx.x] += r1[threadIdx.x + o];
      r2[threadIdx.x] += r2[threadIdx.x + o];
    }
    __syncthreads();
  }
  __shared__ double gs1, gs2;
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    const double n = count / cpg;              // interior pixels
    for (int j = 0; j < cpg; ++j) {
      // un-shift: sum x = a + n s;  sum x^2 = b + 2 s a + n s^2
      const double sj = (double)bf2f(x_first[g * cpg + j]);
      const double aj = r1[j * tpc], bj = r2[j * tpc];
      t1 += aj + n * sj;
      t2 += bj + 2.0 * sj * aj + n * sj * sj;
    }
    const double mean = t1 / count;
    double var = t2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    gs1 = mean;
    gs2 = 1.0 / sqrt(var + (double)eps);
  }
  __syncthreads();
  if (threadIdx.x < cpg) {
    const int cc = g * cpg + threadIdx.x;
    const double ga = (double)bf2f(gamma[cc]), be = (double)bf2f(beta[cc]);
    ab[cc] = (float)(gs2 * ga);
    ab[C + cc] = (float)(be - gs1 * gs2 * ga);
  }
}

// pass 3: y = act(x * a + b) on interior pixels; padded output gets a clean zero halo, compact output drops the halo
template <bool SILU, bool COMPACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W, int C,
                                                       const float* __restrict__ ab) {
  const int tpp = C / 8;
  const int Wp = W + 2;
  const int64_t total = (int64_t)(H + 2) * Wp * tpp;
  const int64_t stride = (int64_t)gridDim.x * 256;          // a multiple of tpp (tpp divides 256)
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cs = (int)(i % tpp);
  float a[8], b[8];
  {
    const f32x4 a0 = *(const f32x4*)(ab + cs * 8), a1 = *(const f32x4*)(ab + cs * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(ab + C + cs * 8), b1 = *(const f32x4*)(ab + C + cs * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = a0[e], a[e + 4] = a1[e], b[e] = b0[e], b[e + 4] = b1[e];
  }
  for (; i < total; i += stride) {
    const int64_t p = i / tpp;
    const int yp = (int)(p / Wp), xp = (int)(p - (int64_t)yp * Wp);
    const bool inside = yp >= 1 && yp <= H && xp >= 1 && xp <= W;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (inside) {
      float v[8];
      unpack8(*(const u32x4*)(x + p * C + cs * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = v[e] * a[e] + b[e];
        if (SILU) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
        v[e] = t;
      }
      o = pack8(v);
    }
    if (COMPACT) {
      if (inside) *(u32x4*)(y + ((int64_t)(yp - 1) * W + (xp - 1)) * C + cs * 8) = o;
    } else {
      *(u32x4*)(y + p * C + cs * 8) = o;
    }
  }
}

// nearest 2x upsample, padded [h+2][w+2][C] -> padded [2h+2][2w+2][C] with a zero halo  (Upsample2D: interpolate(scale 2, nearest))
__global__ __launch_bounds__(256) void upsample2x_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int h, int w, int C) {
  const int tpp = C / 8;
  const int Wo = 2 * w + 2, Wi = w + 2;
  const int64_t total = (int64_t)(2 * h + 2) * Wo * tpp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cs = (int)(i % tpp);
    const int64_t p = i / tpp;
    const int yp = (int)(p / Wo), xp = (int)(p - (int64_t)yp * Wo);
    u32x4 o = {0u, 0u, 0u, 0u};
    if (yp >= 1 && yp <= 2 * h && xp >= 1 && xp <= 2 * w)
      o = *(const u32x4*)(x + ((int64_t)((yp - 1) / 2 + 1) * Wi + ((xp - 1) / 2 + 1)) * C + cs * 8);
    *(u32x4*)(y + p * C + cs * 8) = o;
  }
}

// im2col of a 3x3 / stride 2 convolution over F.pad(x, (0, 1, 0, 1)) (Downsample2D(padding=0)), written over the PADDED
// output index space so the GEMM behind it stores straight into a zero-halo image: row r <-> padded output pixel r + (Wo+3);
// interior rows get the 9 taps in(2 yo + dy, 2 xo + dx) (three runs of 3C contiguous values), halo-column rows get zeros.
__global__ __launch_bounds__(256) void im2col_s2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int Hi, int Wi, int C) {
  const int Ho = Hi / 2, Wo = Wi / 2;
  const int Wop = Wo + 2, Wip = Wi + 2;
  const int cpr = 9 * C / 8;                                  // 16-byte chunks per output row
  const int64_t M = (int64_t)(Ho - 1) * Wop + Wo;
  const int64_t total = M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr);
    const int64_t p = r + Wop + 1;
    const int yp = (int)(p / Wop), xp = (int)(p - (int64_t)yp * Wop);
    u32x4 o = {0u, 0u, 0u, 0u};
    if (xp >= 1 && xp <= Wo) {
      const int dy = c / (3 * C / 8), rem = c - dy * (3 * C / 8);       // rem: chunk inside the 3C run
      const int yi = 2 * (yp - 1) + dy + 1, xi = 2 * (xp - 1) + 1;     // padded input coordinates of tap (dy, 0)
      // the input is a residual stream: its halo is NOT clean (conv outputs leave garbage there) -- the pad (0,1,0,1) taps
      // that fall on the bottom row / right column are zero by construction here, not by reading them
      if (yi <= Hi && xi + (rem * 8) / C <= Wi) o = *(const u32x4*)(x + ((int64_t)yi * Wip + xi) * C + rem * 8);
    }
    *(u32x4*)(col + r * (int64_t)(9 * C) + c * 8) = o;
  }
}

// x_pad[interior] += o_compact   (attention residual: hidden_states + attn(hidden_states))
__global__ __launch_bounds__(256) void add_compact_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ o, int H, int W, int C) {
  const int tpp = C / 8;
  const int64_t total = (int64_t)H * W * tpp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cs = (int)(i % tpp);
    const int64_t s = i / tpp;
    const int y = (int)(s / W), xx = (int)(s - (int64_t)y * W);
    bf16_t* px = x + ((int64_t)(y + 1) * (W + 2) + xx + 1) * C + cs * 8;
    float a[8], b[8];
    unpack8(*(const u32x4*)px, a);
    unpack8(*(const u32x4*)(o + s * C + cs * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    *(u32x4*)px = pack8(a);
  }
}

__global__ __launch_bounds__(256) void fill_bf16_kernel(bf16_t* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = f2bf(v);
}

// ---- mid-block attention: ONE head, head_dim = C = 512, non-causal, S = H W tokens -----------------------------------------
// qk: [S][2C] bf16 (q | k, bias included), vt: [C][S] bf16 (V^T WITHOUT its bias: softmax rows sum to 1, so the bias is added
// after the product -- folded into the out-projection bias by the packer), out: [S][C].
// Workgroup = 64 queries, 8 waves (two per SIMD, 256 registers each): wave w serves the 16 queries of group w & 3 and the HALF
// w >> 2 of the 512 output channels -- 64 O^T accumulator + 64 Q registers per lane, no accumulator-file spilling (a first build
// with 4 waves x all 512 channels needed 128 + 64 + fragments > 256 registers: hipcc moved ~350 registers per key tile between
// the VGPR and AGPR halves and ran 6500 clocks per tile for 1024 clocks of MFMA).  Both waves of a query group compute the same
// scores (+50 % score MFMAs -- the price of keeping every wave inside 256 registers) and hence the same softmax.
// Key tile = 32 keys: K tile [32][C] (1 KB rows, 16-byte chunk c of row r stored at c ^ (r & 15): the 16-lane groups of
// ds_read_b128 hit 16 distinct slots) and V^T tile [C][32] (64-byte rows, chunk c of row d stored at c ^ ((d >> 2) & 3): the two
// ds_read_b64 of a fragment hit 64 distinct banks per half-wave), double buffered, filled by LDS-DMA (buffer_load_dwordx4 ... lds:
// the DMA writes lane-linearly, so both swizzles sit on the per-lane SOURCE offset).
// Scores are computed transposed (S^T = K Q^T) so a lane owns one query: tile T (16 keys) leaves keys 4g..4g+3 of query l15
// in lane group g; the PV product's B operand (P^T) of lane group g is {T0 keys 4g.., T1 keys 16 + 4g..} -- so its A operand
// (V^T fragment) reads exactly those two 4-key runs of a row (2 x ds_read_b64).  Fragment reads run one group of four ahead of
// their MFMAs through a register double buffer fenced by sched_barrier(0) (left alone, hipcc sinks every ds_read to its use).
constexpr int VA_C = 512;
constexpr int VA_KT = 32;                       // keys per tile
constexpr int VA_KBYTES = VA_KT * VA_C * 2;     // 32 KiB
constexpr int VA_VPITCH = 64;
constexpr int VA_VBYTES = VA_C * VA_VPITCH;     // 32 KiB
constexpr int VA_LDS = 2 * (VA_KBYTES + VA_VBYTES);

__global__ __launch_bounds__(512) void vae_attn_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
                                                       int S, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qg = w & 3, hh = w >> 2;                     // query group, channel half
  const int l15 = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 64 + qg * 16;
  const int64_t ldqk = 2 * VA_C;

  // Q B-operand fragments: query q0 + l15, d = 32 ds + 8 g .. +8
  bf16x8 qf[16];
  {
    const bf16_t* qp = qk + (int64_t)(q0 + l15 < S ? q0 + l15 : S - 1) * ldqk + g * 8;
#pragma unroll
    for (int ds = 0; ds < 16; ++ds) qf[ds] = *(const bf16x8*)(qp + ds * 32);
  }
  f32x4 oacc[16];                                        // O^T tiles of this wave's channel half: d = (hh*16 + dt)*16 + 4g + r
#pragma unroll
  for (int dt = 0; dt < 16; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) oacc[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // LDS-DMA pieces of a tile, 4 + 4 per wave: K piece i = key row 8 i + w (1 KB: lane = physical chunk, fetches logical chunk
  // lane ^ (row & 15)); V^T piece i = rows 16 (8 i + w) .. + 15 (lane = (row, physical chunk): fetches logical chunk pc ^ ((row >> 2) & 3))
  const rsrc_t rsK = RF_MAKE_RSRC(qk + VA_C), rsV = RF_MAKE_RSRC(vt);
  uint32_t k_src[4], v_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + w;
    k_src[i] = (uint32_t)(row * (int)ldqk * 2 + ((lane ^ (row & 15)) << 4));
    const int d = 16 * (8 * i + w) + (lane >> 2);
    v_src[i] = (uint32_t)(((int64_t)d * S) * 2 + (((lane & 3) ^ ((d >> 2) & 3)) << 4));
  }
  auto dma_tile = [&](const int t, const int buf) {
    char* kb = smem + buf * (VA_KBYTES + VA_VBYTES);
    char* vb = kb + VA_KBYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      RF_BUF_LOAD_LDS(rsK, (lds_void*)(kb + (8 * i + w) * 1024), k_src[i], t * (VA_KT * (int)ldqk * 2));
      RF_BUF_LOAD_LDS(rsV, (lds_void*)(vb + (8 * i + w) * 1024), v_src[i], t * (VA_KT * 2));
    }
  };

  const int nt = S / VA_KT;
  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // per-lane V^T fragment offsets: row l15 of a d-tile, 8-byte units g and 4 + g of the row -> chunks (g >> 1), 2 + (g >> 1), swizzled
  const int vsw = (l15 >> 2) & 3;
  const int v_off0 = (hh * 256 + l15) * VA_VPITCH + (((g >> 1) ^ vsw) << 4) + (g & 1) * 8;
  const int v_off1 = (hh * 256 + l15) * VA_VPITCH + (((2 + (g >> 1)) ^ vsw) << 4) + (g & 1) * 8;
  for (int t = 0; t < nt; ++t) {
    const char* kb = smem + (t & 1) * (VA_KBYTES + VA_VBYTES);
    const char* vb = kb + VA_KBYTES;
    if (t + 1 < nt) dma_tile(t + 1, (t + 1) & 1);  // lands under this tile's MFMAs; that buffer's readers passed the barrier below
    // ---- S^T = K Q^T: two 16-key tiles x 16 d-steps, fragment reads one group of four ahead
    f32x4 sc[2];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[T][r] = 0.f;
    {
      bf16x8 kfr[2][4];
      auto rd_k = [&](const int G, bf16x8 (&dst)[4]) {       // group G: tile T = G / 4, d-steps 4 (G % 4) .. + 3
        const int row = (G >> 2) * 16 + l15;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ds = (G & 3) * 4 + e;
          dst[e] = *(const bf16x8*)(kb + row * 1024 + (((ds * 4 + g) ^ (row & 15)) << 4));
        }
      };
      rd_k(0, kfr[0]);
#pragma unroll
      for (int G = 0; G < 8; ++G) {
        if (G + 1 < 8) rd_k(G + 1, kfr[(G + 1) & 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          sc[G >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[G & 1][e], qf[(G & 3) * 4 + e], sc[G >> 2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- online softmax for query l15 (its 8 keys of this tile in this lane; the rest in lanes l15 + 16 g')
    float tmax = -1e30f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[T][r] *= scale_log2e;
        tmax = fmaxf(tmax, sc[T][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    float pv[8];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(sc[T][r] - m_new);
        pv[T * 4 + r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
    }
    bf16x8 pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) pb[e] = f2bf(pv[e]);
    // ---- O^T += V^T P^T: this wave's 16 d-tiles, k = this tile's 32 keys in the operand order {4g.., 16 + 4g..}
    {
      u32x4 vfr[2][4];
      auto rd_v = [&](const int G, u32x4 (&dst)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const char* vr = vb + (G * 4 + e) * 16 * VA_VPITCH;
          const u32x2 lo = *(const u32x2*)(vr + v_off0), hi = *(const u32x2*)(vr + v_off1);
          dst[e] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      };
      rd_v(0, vfr[0]);
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        if (G + 1 < 4) rd_v(G + 1, vfr[(G + 1) & 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          oacc[G * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vfr[G & 1][e]), pb, oacc[G * 4 + e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // tile t + 1 has landed for this wave (counted explicitly: never rely on hipcc's bookkeeping for LDS-DMA across a back edge),
    // then for every wave; and every wave is done reading tile t's buffer (its fragment reads were waited for by their MFMAs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ---- normalise and store: lane holds d = (hh*16 + dt)*16 + 4g + r of query q0 + l15
  float l_tot = l_run;
  l_tot += __shfl_xor(l_tot, 16);
  l_tot += __shfl_xor(l_tot, 32);
  const float inv = 1.0f / l_tot;
  if (q0 + l15 < S) {
    bf16_t* orow = out + (int64_t)(q0 + l15) * VA_C + hh * 256 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      u32x2 o;
      o[0] = pack2(oacc[dt][0] * inv, oacc[dt][1] * inv);
      o[1] = pack2(oacc[dt][2] * inv, oacc[dt][3] * inv);
      *(u32x2*)(orow + dt * 16) = o;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static inline int grid_for(int64_t items) {
  const int64_t b = cdiv64(items, 256);
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

struct VaeCtx {
  const rf_vae_weights* w;
  hipStream_t st;
  char* base;
  int64_t cap;                 // bytes per activation buffer
  bf16_t *X, *T1, *T2, *T3;    // padded activation buffers
  bf16_t* ones;                // [1024] bf16 1.0
  float* part;                 // GroupNorm block partials [GN_BLOCKS][C][2]
  float* ab;                   // [2][C]
  bf16_t *aqk, *avt, *ao;      // attention: [S][2C], [C][S], [S][C]
  bf16_t* col;                 // encoder: im2col buffer
  void* sk; int64_t sk_bytes;  // GEMM scratch (stream-K flags + partial tiles)
};
constexpr int GN_BLOCKS = 2048;

static inline int64_t padded_elems(int H, int W, int C) { return (int64_t)(H + 2) * (W + 2) * C; }

#define RF_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != RF_OK) return _rc; \
  } while (0)

static int group_norm(VaeCtx& c, const rf_vae_norm& n, const bf16_t* x, bf16_t* y, int H, int W, int C, bool silu, bool compact) {
  RF_REQUIRE(n.gamma && n.beta, RF_ERR_NULL, "rf_vae: GroupNorm weights NULL");
  const int cpg = c.w->groups > 0 ? C / c.w->groups : 0;
  RF_REQUIRE(C % 8 == 0 && 256 % (C / 8) == 0 && C <= 512 && c.w->groups > 0 && C % c.w->groups == 0 &&
                 (cpg == 2 || cpg == 4 || cpg == 8 || cpg == 16), RF_ERR_SHAPE,
             "rf_vae: GroupNorm over C=%d channels, %d groups is not supported (C in {64,128,256,512}, 2..16 channels per group)", C, c.w->groups);
  const int64_t npix = (int64_t)(H + 2) * (W + 2);
  const int nblk = (int)(npix < GN_BLOCKS * 64 ? cdiv64(npix, 64) : GN_BLOCKS);
  const int ppb = (int)cdiv64(npix, nblk);
  ProfScope prof(RF_KC_ROWOP, 3.0 * (double)H * W * C * 2, c.st);
  switch (C / 8) {
    case 8: hipLaunchKernelGGL(gn_stats_kernel<8>, dim3(nblk), dim3(256), 0, c.st, x, H, W, C, ppb, c.part); break;
    case 16: hipLaunchKernelGGL(gn_stats_kernel<16>, dim3(nblk), dim3(256), 0, c.st, x, H, W, C, ppb, c.part); break;
    case 32: hipLaunchKernelGGL(gn_stats_kernel<32>, dim3(nblk), dim3(256), 0, c.st, x, H, W, C, ppb, c.part); break;
    case 64: hipLaunchKernelGGL(gn_stats_kernel<64>, dim3(nblk), dim3(256), 0, c.st, x, H, W, C, ppb, c.part); break;
    default: RF_REQUIRE(false, RF_ERR_SHAPE, "rf_vae: GroupNorm channel count %d", C);
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(c.w->groups), dim3(256), 0, c.st, c.part, nblk, x + (int64_t)(W + 3) * C, C, c.w->groups,
                     (double)H * W * (C / c.w->groups), 1e-6f, (const bf16_t*)n.gamma, (const bf16_t*)n.beta, c.ab);
  const int grid = grid_for(npix * (C / 8));
  if (compact) {
    if (silu) hipLaunchKernelGGL((gn_apply_kernel<true, true>), dim3(grid), dim3(256), 0, c.st, x, y, H, W, C, c.ab);
    else hipLaunchKernelGGL((gn_apply_kernel<false, true>), dim3(grid), dim3(256), 0, c.st, x, y, H, W, C, c.ab);
  } else {
    if (silu) hipLaunchKernelGGL((gn_apply_kernel<true, false>), dim3(grid), dim3(256), 0, c.st, x, y, H, W, C, c.ab);
    else hipLaunchKernelGGL((gn_apply_kernel<false, false>), dim3(grid), dim3(256), 0, c.st, x, y, H, W, C, c.ab);
  }
  RF_LAUNCH_CHECK();
  return RF_OK;
}

static void set_seg(rf_kseg& s, const void* A, int64_t lda, const void* W, int64_t ldw, int K) {
  s.A = A; s.lda = lda; s.W = W; s.ldw = ldw; s.K = K; s._pad = 0;
}

// 3x3 / stride 1 / pad 1 convolution of a zero-halo image as a 3-segment GEMM (see the file header); residual != NULL: out = residual + conv.
// Narrow outputs run FOLDED when the packer provided the copy (rf_vae_conv.wf) and the image is wide enough: g adjacent output
// pixels are one GEMM row -- A rows of (g+2) Cin taps at a pitch of g Cin, N = g Cout -- so a cout = 128 layer fills the 256-column
// MFMA tile (1.5x fewer MFMAs than half-empty tiles) and the 8-channel conv_out 2.8x fewer.  The last row may run up to g-1
// pixels past the last interior pixel: those land in the bottom halo row (g <= W + 3), whose content nobody reads.
static int conv3x3(VaeCtx& c, const rf_vae_conv& cv, const bf16_t* x, bf16_t* y, int H, int W, const bf16_t* residual) {
  RF_REQUIRE(cv.w && cv.cin % 64 == 0 && cv.cout % 8 == 0 && cv.cout <= 1024, RF_ERR_SHAPE, "rf_vae: conv3x3 %d -> %d (need cin %% 64 == 0, cout %% 8 == 0)",
             cv.cin, cv.cout);
  RF_REQUIRE(padded_elems(H, W, cv.cin > cv.cout ? cv.cin : cv.cout) * 2 <= c.cap, RF_ERR_WORKSPACE, "rf_vae: activation %dx%dx%d exceeds the workspace buffers",
             H, W, cv.cin > cv.cout ? cv.cin : cv.cout);
  const int Wp = W + 2, Ci = cv.cin, Co = cv.cout;
  const int g = (cv.wf != nullptr && cv.fold > 1 && cv.fold <= W + 3) ? cv.fold : 1;
  const int M = (H - 1) * Wp + W;
  const bf16_t* wsel = (const bf16_t*)(g > 1 ? cv.wf : cv.w);
  const int Kseg = (g + 2) * Ci;
  const int64_t lda = (int64_t)g * Ci, ldo = (int64_t)g * Co;
  const int rows = cdiv(M, g);
  // the GEMM addresses an operand through a 2 GiB buffer window (32-bit byte offsets): cut the rows into launches that fit
  // (2048^2 images: 4.2 M rows x 1 KiB).  Chunks are multiples of 256 rows: whole tiles, bit-identical to one launch.
  int64_t max_rows = ((0x7fffffffll - 4096 - (int64_t)Kseg * 2) / (lda * 2)) / 256 * 256;
  if (max_rows < 256) max_rows = 256;
  for (int64_t r0 = 0; r0 < rows; r0 += max_rows) {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = g * Co; d.num_groups = 1; d.epilogue = residual ? RF_EPI_GATE_RES : RF_EPI_STORE;
    d.splitk_ws = c.sk; d.splitk_ws_bytes = c.sk_bytes;
    rf_gemm_group& G = d.g[0];
    G.M = (int)(rows - r0 < max_rows ? rows - r0 : max_rows);
    // the bottleneck-resolution layers (128^2 x 512 channels: 130 tiles of 256^2 with 72 K-tiles each) neither fill the chip with
    // whole 256^2 tiles nor run well on the 128^2-tile kernel AUTO would pick for < 200 tiles: stream-K cuts their K-tile
    // iterations evenly over the CUs
    {
      const int64_t t256 = (int64_t)cdiv(G.M, 256) * cdiv(d.N, 256);
      if (t256 >= 32 && t256 < 200 && 3 * Kseg / 64 >= 16) d.schedule = RF_SCHED_STREAMK;
    }
    for (int dy = 0; dy < 3; ++dy)
      set_seg(G.seg[dy], x + (int64_t)dy * Wp * Ci + r0 * lda, lda, wsel + (int64_t)dy * Kseg, 3 * Kseg, Kseg);
    G.bias = g > 1 ? cv.bf : cv.b;
    G.out = y + (int64_t)(Wp + 1) * Co + r0 * ldo; G.ldo = ldo;
    if (residual) { G.residual = residual + (int64_t)(Wp + 1) * Co + r0 * ldo; G.ldr = ldo; G.gate = c.ones; }
    RF_TRY(rf_gemm_bf16(&d, c.st));
  }
  return RF_OK;
}

// plain GEMM  y[M x N] = x[M x K] . w[N x K]^T + b   (1x1 convolutions over ALL padded pixels, the attention projections)
static int linear(VaeCtx& c, const bf16_t* x, int64_t ldx, const void* w, const void* b, int M, int N, int K, bf16_t* y, int64_t ldy) {
  int64_t max_rows = ((0x7fffffffll - 4096 - (int64_t)K * 2) / (ldx * 2)) / 256 * 256;   // 2 GiB operand window, see conv3x3
  if (max_rows < 256) max_rows = 256;
  for (int64_t r0 = 0; r0 < M; r0 += max_rows) {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = N; d.num_groups = 1; d.epilogue = RF_EPI_STORE;
    d.splitk_ws = c.sk; d.splitk_ws_bytes = c.sk_bytes;
    set_seg(d.g[0].seg[0], x + r0 * ldx, ldx, w, K, K);
    d.g[0].bias = b; d.g[0].M = (int)(M - r0 < max_rows ? M - r0 : max_rows); d.g[0].out = y + r0 * ldy; d.g[0].ldo = ldy;
    RF_TRY(rf_gemm_bf16(&d, c.st));
  }
  return RF_OK;
}

// ResnetBlock2D: x <- shortcut(x) + conv2(silu(gn2(conv1(silu(gn1(x))))))   (in place on c.X; channel count may change)
static int resnet(VaeCtx& c, const rf_vae_resnet& r, int H, int W) {
  const int Ci = r.conv1.cin, Co = r.conv1.cout;
  RF_REQUIRE(r.conv2.cin == Co && r.conv2.cout == Co, RF_ERR_SHAPE, "rf_vae: resnet conv2 %d -> %d after conv1 -> %d", r.conv2.cin, r.conv2.cout, Co);
  RF_TRY(group_norm(c, r.norm1, c.X, c.T1, H, W, Ci, true, false));
  RF_TRY(conv3x3(c, r.conv1, c.T1, c.T2, H, W, nullptr));
  RF_TRY(group_norm(c, r.norm2, c.T2, c.T1, H, W, Co, true, false));
  const bf16_t* res = c.X;
  if (r.shortcut.w) {
    RF_REQUIRE(r.shortcut.cin == Ci && r.shortcut.cout == Co, RF_ERR_SHAPE, "rf_vae: shortcut %d -> %d", r.shortcut.cin, r.shortcut.cout);
    RF_TRY(linear(c, c.X, Ci, r.shortcut.w, r.shortcut.b, (H + 2) * (W + 2), Co, Ci, c.T3, Co));
    res = c.T3;
  } else {
    RF_REQUIRE(Ci == Co, RF_ERR_SHAPE, "rf_vae: resnet %d -> %d without a shortcut convolution", Ci, Co);
  }
  return conv3x3(c, r.conv2, c.T1, c.X, H, W, res);   // out may alias the residual (Ci == Co) -- the epilogue reads before it writes
}

// mid-block attention, in place on c.X: x += to_out(softmax(q k^T / sqrt(C)) v),  q, k, v = linear(group_norm(x))
static int attention(VaeCtx& c, const rf_vae_attn& a, int H, int W) {
  const int C = a.C, S = H * W;
  RF_REQUIRE(C == VA_C, RF_ERR_UNSUPPORTED, "rf_vae: the attention kernel is built for %d channels (got %d)", VA_C, C);
  RF_REQUIRE(S % 64 == 0, RF_ERR_SHAPE, "rf_vae: attention over %d tokens (need a multiple of 64)", S);
  RF_REQUIRE(a.w_qk && a.w_v && a.w_out, RF_ERR_NULL, "rf_vae: attention weights NULL");
  bf16_t* xn = c.T1;                                               // compact [S][C]
  RF_TRY(group_norm(c, a.norm, c.X, xn, H, W, C, false, true));
  RF_TRY(linear(c, xn, C, a.w_qk, a.b_qk, S, 2 * C, C, c.aqk, 2 * C));
  RF_TRY(linear(c, (const bf16_t*)a.w_v, C, xn, nullptr, C, S, C, c.avt, S));   // V^T = W_v . xn^T  (operands swapped)
  {
    static bool attr_set = false;
    if (!attr_set) {
      RF_CHECK_HIP(hipFuncSetAttribute((const void*)vae_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VA_LDS));
      attr_set = true;
    }
    ProfScope prof(RF_KC_ATTN, 4.0 * (double)S * S * C, c.st);
    hipLaunchKernelGGL(vae_attn_kernel, dim3(S / 64), dim3(512), VA_LDS, c.st, c.aqk, c.avt, c.ao, S,
                       1.4426950408889634f / sqrtf((float)C));
    RF_LAUNCH_CHECK();
  }
  RF_TRY(linear(c, c.ao, C, a.w_out, a.b_out, S, C, C, c.T2, C));
  hipLaunchKernelGGL(add_compact_kernel, dim3(grid_for((int64_t)S * (C / 8))), dim3(256), 0, c.st, c.X, c.T2, H, W, C);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

struct VaeSizes { int64_t cap, attn_qk, attn_vt, attn_o, col, total; };

static int max_channels(const rf_vae_weights* w) {
  int m = w->conv_in.cout > w->conv_in.cin ? w->conv_in.cout : w->conv_in.cin;
  for (int i = 0; i < w->levels; ++i)
    for (int j = 0; j < w->res_per_level; ++j) {
      if (w->res[i][j].conv1.cin > m) m = w->res[i][j].conv1.cin;
      if (w->res[i][j].conv1.cout > m) m = w->res[i][j].conv1.cout;
    }
  if (w->conv_out.cin > m) m = w->conv_out.cin;
  if (w->conv_out.cout > m) m = w->conv_out.cout;
  return m;
}

// walk the level structure once to size the buffers: max over layers of (H+2)(W+2) C
static VaeSizes vae_sizes(const rf_vae_weights* w, int encode, int h, int wd) {
  VaeSizes z;
  memset(&z, 0, sizeof(z));
  int64_t cap = 0;
  auto see = [&](int H, int W, int C) { const int64_t b = padded_elems(H, W, C) * 2; if (b > cap) cap = b; };
  int H = h, W = wd;
  see(H, W, w->conv_in.cin); see(H, W, w->conv_in.cout);
  int64_t col = 0;
  for (int i = 0; i < w->levels; ++i) {
    for (int j = 0; j < w->res_per_level; ++j) { see(H, W, w->res[i][j].conv1.cin); see(H, W, w->res[i][j].conv1.cout); }
    if (w->resample[i].w) {
      if (encode) {
        const int64_t cb = (int64_t)(H / 2) * (W / 2 + 2) * 9 * w->resample[i].cin * 2;
        if (cb > col) col = cb;
        H /= 2; W /= 2;
      } else {
        H *= 2; W *= 2;
      }
      see(H, W, w->resample[i].cin); see(H, W, w->resample[i].cout);
    }
  }
  see(H, W, w->conv_out.cin); see(H, W, w->conv_out.cout);
  // the mid block runs at the bottleneck resolution: the input's for decode, the last level's for encode
  const int Hm = encode ? H : h, Wm = encode ? W : wd;
  const int Cm = w->mid0.conv1.cin;
  see(Hm, Wm, Cm);
  z.cap = round_up(cap + 64 * 512 * 2, 256);   // + 64 pixels of slack: a folded convolution's last row reads up to fold - 1 pixels past the image
  if (w->has_attn) {
    const int64_t S = (int64_t)Hm * Wm;
    z.attn_qk = round_up(S * 2 * Cm * 2, 256);
    z.attn_vt = round_up(S * Cm * 2, 256);
    z.attn_o = round_up(S * Cm * 2, 256);
  }
  z.col = round_up(col, 256);
  const int64_t sk = 4096 + (64ll << 20);
  z.total = 4 * z.cap + z.attn_qk + z.attn_vt + z.attn_o + z.col + round_up(1024 * 2, 256) + round_up((int64_t)GN_BLOCKS * 512 * 2 * 4, 256) +
            round_up(2 * 512 * 4, 256) + sk;
  return z;
}

static int vae_ctx(VaeCtx& c, const rf_vae_weights* w, int encode, int h, int wd, const rf_workspace* ws, hipStream_t st) {
  RF_REQUIRE(w && ws && ws->base, RF_ERR_NULL, "rf_vae: weights / workspace NULL");
  RF_REQUIRE(w->levels >= 1 && w->levels <= 4 && w->res_per_level >= 1 && w->res_per_level <= 3, RF_ERR_SHAPE, "rf_vae: %d levels x %d resnets",
             w->levels, w->res_per_level);
  RF_REQUIRE(h > 0 && wd > 0 && max_channels(w) <= 512, RF_ERR_SHAPE, "rf_vae: bad geometry");
  const VaeSizes z = vae_sizes(w, encode, h, wd);
  RF_REQUIRE(aligned16(ws->base) && ws->bytes >= z.total, RF_ERR_WORKSPACE, "rf_vae: workspace %lld < required %lld bytes", (long long)ws->bytes,
             (long long)z.total);
  c.w = w; c.st = st; c.base = (char*)ws->base; c.cap = z.cap;
  char* p = c.base;
  auto take = [&](int64_t bytes) { char* q = p; p += round_up(bytes, 256); return q; };
  c.X = (bf16_t*)take(z.cap); c.T1 = (bf16_t*)take(z.cap); c.T2 = (bf16_t*)take(z.cap); c.T3 = (bf16_t*)take(z.cap);
  c.aqk = (bf16_t*)take(z.attn_qk); c.avt = (bf16_t*)take(z.attn_vt); c.ao = (bf16_t*)take(z.attn_o);
  c.col = (bf16_t*)take(z.col);
  c.ones = (bf16_t*)take(1024 * 2);
  c.part = (float*)take((int64_t)GN_BLOCKS * 512 * 2 * 4);
  c.ab = (float*)take(2 * 512 * 4);
  c.sk = take(4096 + (64ll << 20)); c.sk_bytes = 4096 + (64ll << 20);
  hipLaunchKernelGGL(fill_bf16_kernel, dim3(4), dim3(256), 0, st, c.ones, (int64_t)1024, 1.0f);
  RF_CHECK_HIP(hipMemsetAsync(c.sk, 0, 4096, st));          // stream-K flags: zero before the first launch
  RF_LAUNCH_CHECK();
  return RF_OK;
}

}  // namespace rf

using namespace rf;

extern "C" int64_t rf_vae_workspace_bytes(const rf_vae_weights* w, int32_t encode, int32_t h, int32_t wd) {
  if (!w || h <= 0 || wd <= 0) return RF_ERR_NULL;
  return vae_sizes(w, encode, h, wd).total;
}

// decoder: z (zero-halo NHWC [(h+2)(w+2)][conv_in.cin], latent channels first, the rest zero) -> image (zero-halo NHWC
// [(8h+2)(8w+2)][conv_out.cout], RGB in channels 0..2; halo columns hold garbage -- the caller slices the interior)
extern "C" int rf_vae_decode(const rf_vae_weights* w, const void* z, int32_t h, int32_t wd, void* out, const rf_workspace* ws, void* stream) {
  VaeCtx c;
  hipStream_t st = (hipStream_t)stream;
  RF_TRY(vae_ctx(c, w, 0, h, wd, ws, st));
  RF_REQUIRE(z && out && aligned16(z) && aligned16(out), RF_ERR_NULL, "rf_vae_decode: NULL / unaligned tensor");
  int H = h, W = wd;
  RF_TRY(conv3x3(c, w->conv_in, (const bf16_t*)z, c.X, H, W, nullptr));
  RF_TRY(resnet(c, w->mid0, H, W));
  if (w->has_attn) RF_TRY(attention(c, w->attn, H, W));
  RF_TRY(resnet(c, w->mid1, H, W));
  for (int i = 0; i < w->levels; ++i) {
    for (int j = 0; j < w->res_per_level; ++j) RF_TRY(resnet(c, w->res[i][j], H, W));
    if (w->resample[i].w) {
      const int C = w->resample[i].cin;
      RF_REQUIRE(padded_elems(2 * H, 2 * W, C) * 2 <= c.cap, RF_ERR_WORKSPACE, "rf_vae_decode: upsampled activation exceeds the workspace");
      {
        ProfScope prof(RF_KC_ROWOP, 5.0 * (double)H * W * C * 2, st);
        hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(padded_elems(2 * H, 2 * W, C) / 8)), dim3(256), 0, st, c.X, c.T1, H, W, C);
        RF_LAUNCH_CHECK();
      }
      H *= 2; W *= 2;
      RF_TRY(conv3x3(c, w->resample[i], c.T1, c.X, H, W, nullptr));
    }
  }
  RF_TRY(group_norm(c, w->norm_out, c.X, c.T1, H, W, w->conv_out.cin, true, false));
  // conv_out writes the caller's buffer: its top / bottom halo rows are not touched
  return conv3x3(c, w->conv_out, c.T1, (bf16_t*)out, H, W, nullptr);
}

// encoder: image (zero-halo NHWC [(H+2)(W+2)][conv_in.cin], RGB in channels 0..2) -> moments (zero-halo NHWC
// [(H/8+2)(W/8+2)][conv_out.cout] = mean | logvar)
extern "C" int rf_vae_encode(const rf_vae_weights* w, const void* img, int32_t Hin, int32_t Win, void* out, const rf_workspace* ws, void* stream) {
  VaeCtx c;
  hipStream_t st = (hipStream_t)stream;
  RF_TRY(vae_ctx(c, w, 1, Hin, Win, ws, st));
  RF_REQUIRE(img && out && aligned16(img) && aligned16(out), RF_ERR_NULL, "rf_vae_encode: NULL / unaligned tensor");
  int H = Hin, W = Win;
  RF_TRY(conv3x3(c, w->conv_in, (const bf16_t*)img, c.X, H, W, nullptr));
  for (int i = 0; i < w->levels; ++i) {
    for (int j = 0; j < w->res_per_level; ++j) RF_TRY(resnet(c, w->res[i][j], H, W));
    if (w->resample[i].w) {
      const rf_vae_conv& cv = w->resample[i];
      RF_REQUIRE(H % 2 == 0 && W % 2 == 0 && cv.cin % 64 == 0 && cv.cout % 8 == 0, RF_ERR_SHAPE, "rf_vae_encode: downsample of %dx%dx%d", H, W, cv.cin);
      const int Ho = H / 2, Wo = W / 2, Ci = cv.cin, Co = cv.cout;
      const int64_t M = (int64_t)(Ho - 1) * (Wo + 2) + Wo;
      {
        ProfScope prof(RF_KC_ROWOP, 2.0 * (double)M * 9 * Ci * 2, st);
        hipLaunchKernelGGL(im2col_s2_kernel, dim3(grid_for(M * (9 * Ci / 8))), dim3(256), 0, st, c.X, c.col, H, W, Ci);
        RF_LAUNCH_CHECK();
      }
      RF_TRY(linear(c, c.col, 9 * Ci, cv.w, cv.b, (int)M, Co, 9 * Ci, c.T1 + (int64_t)(Wo + 3) * Co, Co));
      // (the result is a residual stream: nobody reads its halo -- GroupNorm / im2col skip it and write clean halos for what a conv reads)
      bf16_t* t = c.X; c.X = c.T1; c.T1 = t;
      H = Ho; W = Wo;
    }
  }
  RF_TRY(resnet(c, w->mid0, H, W));
  if (w->has_attn) RF_TRY(attention(c, w->attn, H, W));
  RF_TRY(resnet(c, w->mid1, H, W));
  RF_TRY(group_norm(c, w->norm_out, c.X, c.T1, H, W, w->conv_out.cin, true, false));
  return conv3x3(c, w->conv_out, c.T1, (bf16_t*)out, H, W, nullptr);
}
