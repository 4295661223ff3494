// Flash-attention BACKWARD for gfx950 (MI355X), head_dim 128, non-causal joint attention (the training
// step of train_flux/train/model.py:164-238 differentiates F.scaled_dot_product_attention of block.py:123-125).
//
//   forward (recomputed by the caller):  s2 = q~ . k   (q~ carries softmax_scale * log2 e),  P = exp2(s2 - lse2),  O = P V
//   backward:   D  = rowsum(dO o O)                 dV = P^T dO
//               dP = dO V^T                         g  = ln2 * P o (dP - D)          ( = dL/ds2 )
//               dq~ = g K                           dK = g^T q~
//
// Deterministic, no atomics: two kernels, each accumulating its outputs in registers over a loop --
//   attn_bwd_dq_kernel   one workgroup per 64 queries (a wave owns 16): pass 1 streams the keys once for the row statistics
//                        lse2 (online max / sum; the forward kernels do not emit them), pass 2 streams them again for dq~;
//   (both loops double-buffer their LDS stage: the global loads of step s+1 are issued before step s multiplies and committed to
//   the other buffer after it -- one barrier per step)
//   attn_bwd_dkv_kernel  one workgroup per 64 * KT keys (a wave owns KT tiles of 16), streams the queries for dK, dV.
// All five products run on v_mfma_f32_16x16x32_bf16.  A 16x16 score tile leaves a lane with 4 rows of ONE column, so two
// tiles give the 8 contraction slots of the next MFMA's operand without any data movement, provided the other operand is
// laid out in the same slot order:  slot(n) = 8 ((n % 16) / 4) + 4 (n / 16) + n % 4  for index n of a block of 32
// (the trick of the forward kernels' V^T).  The producers write those operands once as TRANSPOSED TILES
//   xT [heads][s_pad / 32][128 (d)][32 (slot)]     for x in {q~, k, dO}
// (rf_qkv_train_fwd for q~ / k, attn_bwd_prep_kernel for dO), so every LDS stage here is a straight 16-byte copy and every
// fragment read a conflict-free ds_read_b128: row-major [32][128] tiles are XOR-swizzled per 16-byte chunk (chunk ^ (row & 15)),
// transposed tiles are read as 1 KiB contiguous per fragment.
#include "common.hpp"

namespace rf {

constexpr int AB_ROWS = 32 * 256;     // bytes of a [32][128] bf16 tile (row-major, swizzled) or of a transposed tile
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int slot32(int n) { return 8 * ((n & 15) >> 2) + 4 * (n >> 4) + (n & 3); }

// NT threads move a [32][128] bf16 row tile (global row pitch ld elements, rows >= rows_valid read as zero) or a transposed tile
// (8 KiB contiguous) in two halves, so that the global loads of step s+1 are in flight while step s multiplies: FETCH into
// registers (512 / NT chunks of 16 bytes per thread) ... and COMMIT them to the LDS image afterwards (row tiles swizzled).
template <int NT>
__device__ __forceinline__ void fetch_rows(u32x4 (&r)[512 / NT], const bf16_t* src, int64_t ld, int rows_valid, int tid) {
#pragma unroll
  for (int p = 0; p < 512 / NT; ++p) {
    const int idx = tid + NT * p, row = idx >> 4, c = idx & 15;
    r[p] = u32x4{0u, 0u, 0u, 0u};
    if (row < rows_valid) r[p] = *(const u32x4*)(src + (int64_t)row * ld + c * 8);
  }
}
template <int NT>
__device__ __forceinline__ void fetch_tile(u32x4 (&r)[512 / NT], const bf16_t* src, int tid) {
#pragma unroll
  for (int p = 0; p < 512 / NT; ++p) r[p] = *(const u32x4*)(src + (tid + NT * p) * 8);
}
template <int NT>
__device__ __forceinline__ void commit_rows(char* dst, const u32x4 (&r)[512 / NT], int tid) {
#pragma unroll
  for (int p = 0; p < 512 / NT; ++p) {
    const int idx = tid + NT * p, row = idx >> 4, c = idx & 15;
    *(u32x4*)(dst + row * 256 + ((c ^ (row & 15)) << 4)) = r[p];
  }
}
template <int NT>
__device__ __forceinline__ void commit_tile(char* dst, const u32x4 (&r)[512 / NT], int tid) {
#pragma unroll
  for (int p = 0; p < 512 / NT; ++p) *(u32x4*)(dst + (tid + NT * p) * 16) = r[p];
}
// A/B fragment of a swizzled row tile: lane (g, i) <- row 16 t + i, elements 32 ks + 8 g .. + 7
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int t, int ks, int l15, int g) {
  const int r = 16 * t + l15;
  return *(const bf16x8*)(tile + r * 256 + (((4 * ks + g) ^ (r & 15)) << 4));
}
// fragment of a transposed tile: lane (g, i) <- d = 16 dt + i, slots 8 g .. 8 g + 7.  A row is 64 bytes, so four rows share a
// 256-byte bank row and ds_read_b128's 16-lane groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- would meet rows i and i + 12 (and i + 4, i + 8 of the neighbouring g) on one 16-byte slot: 2-way
// conflicts on every such read (rocprofv3: 25-31 % of the kernels' LDS cycles, profiles/r04_attention_bwd_pmc.md).  The LDS image
// therefore keeps chunk g of row d at position g ^ TT_SWZ((d >> 2) & 3), TT_SWZ = {0, 2, 3, 1}: within every lane group the four
// (g, d >> 2) classes land on four different positions.  (Applied on the global side of the LDS-DMA, see dma_tile.)
__device__ __forceinline__ int tt_swz(int h) { return (0x78 >> (2 * h)) & 3; }
__device__ __forceinline__ bf16x8 frag_tile(const char* tile, int dt, int l15, int g) {
  return *(const bf16x8*)(tile + (16 * dt + l15) * 64 + ((g ^ tt_swz(l15 >> 2)) << 4));
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
  return c;
#endif
}

// The output accumulators of both kernels (dq~: QT x 8 quads; dK, dV: 2 x KT x 8) live in AGPRs and are accumulated IN PLACE (inline asm pins them there:
// left to the register allocator the loop carried ~100 v_accvgpr_read per step for the k / v fragments it had parked in AGPRs
// instead).  MFMA -> MFMA on one accumulator is interlocked by the hardware; the first read after the loop is fenced by s_nops.
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_ACC_MFMA(C, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(A_), "v"(B_))
#else
#define RF_ACC_MFMA(C, A_, B_) ((void)0)
#endif

// ---- LDS-DMA staging (buffer_load_dwordx4 ... lds) ---------------------------------------------------------------------------------
// A step of either loop multiplies for ~1 us against a staged tile set that takes longer than that to arrive from the L2, so one
// step of lookahead through registers left every step waiting for its loads (2 us per step measured, both kernels).  The stages
// are therefore a RING of AB_RING slots filled by LDS-DMA AB_RING - 1 steps ahead: no staging registers, and the only wait is
// `vmcnt` down to the pieces of the NEWER stages.  A DMA instruction of a wave writes 1 KiB of LDS linearly (lane x 16 bytes), so
// the XOR swizzle of the row tiles is applied on the global side: position (row, pc) receives chunk pc ^ (row & 15).
constexpr int AB_RING = 4;
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_MAKE_RSRC_N(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), 0, (int)(bytes), 0x00020000)
#define RF_BUF_LOAD_LDS4(r, lds, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 4, voff, soff, 0, 0)
#else
#define RF_MAKE_RSRC_N(p, bytes) 0
#define RF_BUF_LOAD_LDS4(r, lds, voff, soff) ((void)0)
#endif
// per-lane byte offsets of a wave's pieces of a row tile (row pitch in bytes) / of a transposed tile; piece i of wave w = 1 KiB
// number i NW + w of the 8 KiB image
template <int NW>
__device__ __forceinline__ void dma_row_offsets(uint32_t (&off)[8 / NW], const uint32_t pitch, const int w, const int lane) {
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) {
    const int row = 4 * (i * NW + w) + (lane >> 4);
    off[i] = (uint32_t)row * pitch + (uint32_t)(((lane & 15) ^ (row & 15)) << 4);
  }
}
template <int NW>
__device__ __forceinline__ void dma_rows(const rsrc_t r, char* tile, const uint32_t (&off)[8 / NW], const uint32_t base, const int w) {
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) RF_BUF_LOAD_LDS(r, (lds_void*)(tile + (i * NW + w) * 1024), off[i], base);
}
template <int NW>
__device__ __forceinline__ void dma_tile(const rsrc_t r, char* tile, const uint32_t base, const int w, const int lane) {
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i)   // LDS position (row = 16 piece + lane / 4, pos = lane % 4) <- chunk pos ^ TT_SWZ((row >> 2) & 3) of that row
    RF_BUF_LOAD_LDS(r, (lds_void*)(tile + (i * NW + w) * 1024),
                    (uint32_t)((i * NW + w) * 1024 + (lane >> 2) * 64 + (((lane & 3) ^ tt_swz((lane >> 4) & 3)) << 4)), base);
}
// wait until at most `later` NEWER stages of PER pieces each are still in flight (later <= AB_RING - 2)
template <int PER>
__device__ __forceinline__ void dma_wait(const int later) {
  if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
  else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- prep: D = rowsum(dO o O), dO^T tiles -----------------------------------------------------------------------------------
// grid (s_pad / 32, heads), 256 threads: 32 tokens x the head's 128 channels.
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16_t* __restrict__ o, int64_t ldo, const bf16_t* __restrict__ dout,
                                                            int64_t lddo, bf16_t* __restrict__ dot, float* __restrict__ dsum, int S,
                                                            int s_pad) {
  __shared__ bf16_t tile[32][130];          // pitch 65 dwords: column reads of 2-byte elements spread over the banks
  const int tid = threadIdx.x, head = blockIdx.y, t0 = blockIdx.x * 32;
  const int i = tid >> 3, c = tid & 7, tok = t0 + i;
  float part = 0.f;
  float dv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dv[e] = 0.f;
  if (tok < S) {
    float ov[16];
    const bf16_t* dp = dout + (int64_t)tok * lddo + head * 128 + c * 16;
    const bf16_t* op = o + (int64_t)tok * ldo + head * 128 + c * 16;
    unpack8(*(const u32x4*)dp, *reinterpret_cast<float(*)[8]>(&dv[0]));
    unpack8(*(const u32x4*)(dp + 8), *reinterpret_cast<float(*)[8]>(&dv[8]));
    unpack8(*(const u32x4*)op, *reinterpret_cast<float(*)[8]>(&ov[0]));
    unpack8(*(const u32x4*)(op + 8), *reinterpret_cast<float(*)[8]>(&ov[8]));
#pragma unroll
    for (int e = 0; e < 16; ++e) part += dv[e] * ov[e];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) tile[i][c * 16 + e] = f2bf(dv[e]);
  part += __shfl_xor(part, 1);
  part += __shfl_xor(part, 2);
  part += __shfl_xor(part, 4);
  if (c == 0) dsum[(int64_t)head * s_pad + tok] = part;
  __syncthreads();
  bf16_t* dst = dot + ((int64_t)head * (s_pad >> 5) + blockIdx.x) * (128 * 32);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int d = (tid >> 2) + 64 * p, g = tid & 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(tile[(e < 4 ? 4 * g + e : 16 + 4 * g + e - 4)][d]);
    *(u32x4*)(dst + d * 32 + g * 8) = pack8(v);
  }
}

// (head, row block) of a workgroup of the (blocks, heads) grids below.  Workgroup L = blockIdx.y * gridDim.x + blockIdx.x is placed on
// XCD L % 8; taken literally that spreads the ~25 workgroups of a head over all eight L2s and each of them fetches the head's
// K / V (Q / dO) stream for itself -- rocprofv3 FETCH_SIZE: 944 MB per dq launch for 104 MB of distinct operands.  With
// heads % 8 == 0 (FLUX: 24) the workgroups of XCD x walk heads x, x + 8, x + 16 block by block instead, as the forward kernels do.
__device__ __forceinline__ void head_and_block(int& head, int& blk) {
  const int nb = gridDim.x, H = gridDim.y;
  if ((H & 7) == 0) {
    const int L = blockIdx.y * nb + blockIdx.x, x = L & 7, j = L >> 3;
    head = (j / nb) * 8 + x;
    blk = j % nb;
  } else {
    head = blockIdx.y;
    blk = blockIdx.x;
  }
}

// ---- dq~ (and the row statistics) -------------------------------------------------------------------------------------------
// grid (ceil(s_pad / (128 QT)), heads), 512 threads: 8 waves, a wave owns QT tiles of 16 queries, so a workgroup multiplies every
// staged key step (K rows | V rows | K^T tile, 24 KiB) against 128 QT queries -- the loop is L2 -> LDS bandwidth bound (each
// workgroup streams the head's whole K, V, K^T), and FLOPs per staged byte scale with the queries per workgroup.
// HAVE_LSE: the forward launch already wrote the row statistics (rf_attn_desc.lse): pass 1 is skipped.
// NW waves (8 or 4) x QT query tiles: 16 NW QT queries per workgroup -- the host picks the pair whose grid wastes least of its last
// round of CUs (S = 5632 x 24 heads: 256-query workgroups are 2.06 rounds run as 3, 192-query ones 2.81).
template <int QT, bool HAVE_LSE, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, const bf16_t* __restrict__ kt,
                                                          const bf16_t* __restrict__ dout, int64_t lddo, const float* __restrict__ dsum,
                                                          float* __restrict__ lse, bf16_t* __restrict__ dq, int S, int s_pad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // AB_RING stages of {K rows | V rows | K^T tile} (98 304 bytes)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, g = lane >> 4;
  int head, blk;
  head_and_block(head, blk);
  const int64_t hb = (int64_t)head * s_pad;
  int qrow[QT];                                               // this lane's queries (COLUMNS of the S^T tiles)
  bf16x8 qf[QT][4], dof[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qrow[t] = blk * (16 * NW * QT) + w * (16 * QT) + 16 * t + l15;
    const int qr = qrow[t] < s_pad ? qrow[t] : s_pad - 1;     // (a partial last workgroup: those lanes are not written)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[t][ks] = *(const bf16x8*)(q + (hb + qr) * 128 + 32 * ks + 8 * g);
      u32x4 z = {0u, 0u, 0u, 0u};
      if (qrow[t] < S) z = *(const u32x4*)(dout + (int64_t)qrow[t] * lddo + head * 128 + 32 * ks + 8 * g);
      dof[t][ks] = __builtin_bit_cast(bf16x8, z);
    }
  }
  const int nsteps = (S + 31) >> 5;
  const float NEG = -__builtin_huge_valf();

  u32x4 rk[8 / NW];
  float my_lse[QT], my_d[QT];
  if constexpr (HAVE_LSE) {
    // padded queries: lse = +huge makes every P of that row exactly 0 here and in the dK / dV kernel (which reads all s_pad rows)
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      my_lse[t] = qrow[t] < S ? lse[hb + qrow[t]] : 1e30f;
      if (g == 0 && qrow[t] >= S && qrow[t] < s_pad) lse[hb + qrow[t]] = 1e30f;
      my_d[t] = qrow[t] < s_pad ? dsum[hb + qrow[t]] : 0.f;
    }
  } else {
  // pass 1: lse2 of this lane's queries over all keys (each lane sees keys 4 g + r (+16) of every step; merged over g at the end)
  float m[QT], l[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) m[t] = NEG, l[t] = 0.f;
  fetch_rows<NW * 64>(rk, k + hb * 128, 128, 32, tid);
  commit_rows<NW * 64>(smem, rk, tid);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const char* kl = smem + (st & 1) * 3 * AB_ROWS;
    const bool more = st + 1 < nsteps;
    if (more) fetch_rows<NW * 64>(rk, k + (hb + (st + 1) * 32) * 128, 128, 32, tid);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kfr[2][4];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kfr[t2][ks] = frag_rows(kl, t2, ks, l15, g);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f32x4 s[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        s[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[t2] = mfma16(kfr[t2][ks], qf[t][ks], s[t2]);
      }
      float mx = m[t];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (st * 32 + 16 * t2 + 4 * g + r >= S) s[t2][r] = NEG;
          mx = fmaxf(mx, s[t2][r]);
        }
      if (mx > NEG) {
        float add = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int r = 0; r < 4; ++r) add += __builtin_amdgcn_exp2f(s[t2][r] - mx);
        l[t] = l[t] * __builtin_amdgcn_exp2f(m[t] - mx) + add;     // m = -inf, l = 0 on the first contribution: exp2(-inf) = 0
        m[t] = mx;
      }
    }
    if (more) commit_rows<NW * 64>(smem + ((st + 1) & 1) * 3 * AB_ROWS, rk, tid);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float m2 = __shfl_xor(m[t], o), l2 = __shfl_xor(l[t], o);
      const float mn = fmaxf(m[t], m2);
      if (mn > NEG) l[t] = l[t] * __builtin_amdgcn_exp2f(m[t] - mn) + l2 * __builtin_amdgcn_exp2f(m2 - mn);
      m[t] = mn;
    }
    // padded queries: lse = +huge makes every P of that row exactly 0 in the dK / dV kernel
    my_lse[t] = (qrow[t] < S && l[t] > 0.f) ? m[t] + __builtin_amdgcn_logf(l[t]) : 1e30f;
    if (g == 0 && qrow[t] < s_pad) lse[hb + qrow[t]] = my_lse[t];
    my_d[t] = qrow[t] < s_pad ? dsum[hb + qrow[t]] : 0.f;
  }
  }   // !HAVE_LSE

  // pass 2: dq~^T[d][q] += K^T[d][key slots] g^T[key slots][q]
  f32x4 acc[QT][8];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // -lse2 and -D ride into the score / dP MFMA chains as their C operand (a lane's four accumulators are four KEYS of one query, so
  // both are lane constants): P = exp2(chain), g / ln2 = P o chain' -- one exp2 and one multiply per element; ln 2 is applied once to
  // the finished dq~.  Keys >= S exist only in the last step (zero K / V rows: they add nothing to dq~, but exp2(-lse2) of a row
  // whose scores are all far below zero must not meet them as inf * 0): that step alone masks.
  f32x4 negl[QT], negd[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) negl[t][r] = -my_lse[t], negd[t][r] = -my_d[t];
  const rsrc_t r_k = RF_MAKE_RSRC_N(k, (int64_t)gridDim.y * s_pad * 256), r_v = RF_MAKE_RSRC_N(v, (int64_t)gridDim.y * s_pad * 256);
  const rsrc_t r_kt = RF_MAKE_RSRC_N(kt, (int64_t)gridDim.y * s_pad * 256);
  uint32_t off_r[8 / NW];
  dma_row_offsets<NW>(off_r, 256u, w, lane);
  constexpr int PER = 3 * 8 / NW;                           // DMA instructions per stage and wave
  auto issue = [&](const int st) {
    char* buf = smem + (st % AB_RING) * 3 * AB_ROWS;
    dma_rows<NW>(r_k, buf, off_r, (uint32_t)((hb + st * 32) * 256), w);
    dma_rows<NW>(r_v, buf + AB_ROWS, off_r, (uint32_t)((hb + st * 32) * 256), w);
    dma_tile<NW>(r_kt, buf + 2 * AB_ROWS, (uint32_t)(((int64_t)head * (s_pad >> 5) + st) * (128 * 32 * 2)), w, lane);
  };
  auto step = [&](const int st, auto last_tag, f32x4 (&accr)[QT][8]) {   // (accr = acc: an asm operand cannot name a capture of a generic lambda)
    constexpr bool LAST = decltype(last_tag)::value;
    dma_wait<PER>(min(AB_RING - 2, nsteps - 1 - st));     // stage st has landed (this wave's pieces) ...
    __syncthreads();                                      // ... and everybody's; every wave is done with step st - 1
    if (st + AB_RING - 1 < nsteps) issue(st + AB_RING - 1);   // into the slot step st - 1 used
    const char* kl = smem + (st % AB_RING) * 3 * AB_ROWS;
    const char* vl = kl + AB_ROWS;
    const char* ktl = kl + 2 * AB_ROWS;
    __builtin_amdgcn_sched_barrier(0);
    uint32_t gw[QT][4];                                  // the 8-slot g operand of each query tile, as packed bf16 pairs
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      bf16x8 kfr[4], vfr[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kfr[ks] = frag_rows(kl, t2, ks, l15, g), vfr[ks] = frag_rows(vl, t2, ks, l15, g);
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        f32x4 s = negl[t], dp = negd[t];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma16(kfr[ks], qf[t][ks], s);
          dp = mfma16(vfr[ks], dof[t][ks], dp);
        }
        float gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(s[r]);
          if (LAST && st * 32 + 16 * t2 + 4 * g + r >= S) p = 0.f;
          gv[r] = p * dp[r];
        }
        gw[t][2 * t2] = pack2(gv[0], gv[1]);            // slots 4 t2 .. 4 t2 + 3 of this lane group
        gw[t][2 * t2 + 1] = pack2(gv[2], gv[3]);
      }
    }
    bf16x8 gf[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) gf[t] = __builtin_bit_cast(bf16x8, u32x4{gw[t][0], gw[t][1], gw[t][2], gw[t][3]});
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16x8 a_k = frag_tile(ktl, dt, l15, g);
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        // one wave per SIMD: 512 registers, the accumulators pinned in the AGPR half; two waves per SIMD: 256 in all, and pinning
        // splits them 128 + 128 with spills across -- there the allocator keeps everything in VGPRs
        if constexpr (NW == 4) RF_ACC_MFMA(accr[t][dt], a_k, gf[t]);
        else accr[t][dt] = mfma16(a_k, gf[t], accr[t][dt]);
      }
    }
  };
  __syncthreads();                                        // (the statistics pass, if it ran, is done with the LDS)
#pragma unroll
  for (int st = 0; st < AB_RING - 1; ++st)
    if (st < nsteps) issue(st);
  for (int st = 0; st + 1 < nsteps; ++st) step(st, std::false_type{}, acc);
  step(nsteps - 1, std::true_type{}, acc);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have left the matrix pipe before the AGPRs are read
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (qrow[t] >= s_pad) continue;
    bf16_t* drow = dq + (hb + qrow[t]) * 128 + 4 * g;      // padded rows are written too (zeros: their dO and D are zero)
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      u32x2 o;
      o[0] = pack2(acc[t][dt][0] * LN2, acc[t][dt][1] * LN2);
      o[1] = pack2(acc[t][dt][2] * LN2, acc[t][dt][3] * LN2);
      *(u32x2*)(drow + 16 * dt) = o;
    }
  }
}

// ---- dK, dV -----------------------------------------------------------------------------------------------------------------
// grid (s_pad / (64 KT), heads), 256 threads; a wave owns KT tiles of 16 keys (their k, v fragments stay in registers).
// LDS per step of 32 queries: q~ rows | dO rows | q~^T tile | dO^T tile | lse[32] | D[32].
// RING = LDS ring slots, WPE = waves per SIMD the register budget must allow (2: two workgroups share a CU)
template <int KT, int RING = AB_RING, int WPE = 1>
__global__ __launch_bounds__(256, WPE) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ qt,
                                                           const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                           const bf16_t* __restrict__ dout, int64_t lddo, const bf16_t* __restrict__ dot,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int S, int s_pad) {
  constexpr int BUF = 4 * AB_ROWS + 512;                 // q~ rows | dO rows | q~^T tile | dO^T tile | lse2[64] | D[64]
  extern __shared__ __attribute__((aligned(16))) char smem[];   // RING * BUF = 133 120 bytes (dynamic)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, g = lane >> 4;
  int head, blk;
  head_and_block(head, blk);
  const int64_t hb = (int64_t)head * s_pad;
  const int kv0 = (blk * 4 + w) * 16 * KT;                    // first key of this wave
  bf16x8 kf[KT][4], vf[KT][4];
#pragma unroll
  for (int kt_ = 0; kt_ < KT; ++kt_) {
    int kr = kv0 + 16 * kt_ + l15;
    kr = kr < s_pad ? kr : s_pad - 1;                         // (a partial last workgroup: results of those lanes are not written)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[kt_][ks] = *(const bf16x8*)(k + (hb + kr) * 128 + 32 * ks + 8 * g);
      vf[kt_][ks] = *(const bf16x8*)(v + (hb + kr) * 128 + 32 * ks + 8 * g);
    }
  }
  f32x4 akv[KT][8], avv[KT][8];
#pragma unroll
  for (int kt_ = 0; kt_ < KT; ++kt_)
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) akv[kt_][dt] = f32x4{0.f, 0.f, 0.f, 0.f}, avv[kt_][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (S + 31) >> 5;
  // stage = q~ rows | dO rows | q~^T tile | dO^T tile | lse2[64] | D[64] (the kernel uses the first 32 of each)
  const rsrc_t r_q = RF_MAKE_RSRC_N(q, (int64_t)gridDim.y * s_pad * 256), r_qt = RF_MAKE_RSRC_N(qt, (int64_t)gridDim.y * s_pad * 256);
  const rsrc_t r_dot = RF_MAKE_RSRC_N(dot, (int64_t)gridDim.y * s_pad * 256);
  const rsrc_t r_do = RF_MAKE_RSRC_N(dout, (int64_t)S * lddo * 2);                 // rows >= S read as zero (range check)
  const rsrc_t r_lse = RF_MAKE_RSRC_N(lse, (int64_t)gridDim.y * s_pad * 4), r_d = RF_MAKE_RSRC_N(dsum, (int64_t)gridDim.y * s_pad * 4);
  uint32_t off_q[2], off_do[2];
  dma_row_offsets<4>(off_q, 256u, w, lane);
  dma_row_offsets<4>(off_do, (uint32_t)(lddo * 2), w, lane);
  constexpr int PER = 10;                                   // DMA instructions per stage and wave
  auto issue = [&](const int st) {
    char* buf = smem + (st % RING) * BUF;
    const int q0 = st * 32;
    dma_rows<4>(r_q, buf, off_q, (uint32_t)((hb + q0) * 256), w);
    dma_rows<4>(r_do, buf + AB_ROWS, off_do, (uint32_t)(((int64_t)q0 * lddo + head * 128) * 2), w);
    dma_tile<4>(r_qt, buf + 2 * AB_ROWS, (uint32_t)(((int64_t)head * (s_pad >> 5) + st) * (128 * 32 * 2)), w, lane);
    dma_tile<4>(r_dot, buf + 3 * AB_ROWS, (uint32_t)(((int64_t)head * (s_pad >> 5) + st) * (128 * 32 * 2)), w, lane);
    // (every wave issues the two statistics pieces: identical bytes, and it keeps the waves' vmcnt arithmetic the same)
    RF_BUF_LOAD_LDS4(r_lse, (lds_void*)(buf + 4 * AB_ROWS), (uint32_t)(lane * 4), (uint32_t)((hb + q0) * 4));
    RF_BUF_LOAD_LDS4(r_d, (lds_void*)(buf + 4 * AB_ROWS + 256), (uint32_t)(lane * 4), (uint32_t)((hb + q0) * 4));
  };
#pragma unroll
  for (int st = 0; st < RING - 1; ++st)
    if (st < nsteps) issue(st);
  for (int st = 0; st < nsteps; ++st) {
    dma_wait<PER>(min(RING - 2, nsteps - 1 - st));     // stage st has landed (this wave's pieces) ...
    __syncthreads();                                      // ... and everybody's; every wave is done with step st - 1
    if (st + RING - 1 < nsteps) issue(st + RING - 1);   // into the slot step st - 1 used
    __builtin_amdgcn_sched_barrier(0);
    const char* ql = smem + (st % RING) * BUF;
    const char* dol = ql + AB_ROWS;
    const char* qtl = ql + 2 * AB_ROWS;
    const char* dotl = ql + 3 * AB_ROWS;
    const float* stat = (const float*)(ql + 4 * AB_ROWS);
    f32x4 lv[2], dvv[2];                                  // -lse2, -D of this lane's queries: the C operands of the chains below
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      lv[t] = -*(const f32x4*)(stat + 16 * t + 4 * g);
      dvv[t] = -*(const f32x4*)(stat + 64 + 16 * t + 4 * g);
    }
    // query tile t outermost: its q~ / dO fragments are read once and serve the KT key tiles (32 registers at a time, not 64)
    uint32_t pw[KT][4], gw[KT][4];                          // P / g operands of the second pair of products, packed bf16 pairs
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 qfr[4], dofr[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qfr[ks] = frag_rows(ql, t, ks, l15, g), dofr[ks] = frag_rows(dol, t, ks, l15, g);
#pragma unroll
      for (int kt_ = 0; kt_ < KT; ++kt_) {
        f32x4 s = lv[t], dp = dvv[t];                                        // chains start from -lse2[q], -D[q] (row = query here)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma16(qfr[ks], kf[kt_][ks], s);                               // S[q = 16 t + 4 g + r][key = l15] - lse2[q]
          dp = mfma16(dofr[ks], vf[kt_][ks], dp);
        }
        float pv[4], gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(s[r]);                              // padded queries: -lse = -1e30 -> 0
          gv[r] = pv[r] * dp[r];                                             // g / ln 2 (ln 2 goes onto the finished dK)
        }
        pw[kt_][2 * t] = pack2(pv[0], pv[1]), pw[kt_][2 * t + 1] = pack2(pv[2], pv[3]);     // slots 4 t .. 4 t + 3 of this lane group
        gw[kt_][2 * t] = pack2(gv[0], gv[1]), gw[kt_][2 * t + 1] = pack2(gv[2], gv[3]);
      }
    }
    bf16x8 pf[KT], gf[KT];
#pragma unroll
    for (int kt_ = 0; kt_ < KT; ++kt_) {
      pf[kt_] = __builtin_bit_cast(bf16x8, u32x4{pw[kt_][0], pw[kt_][1], pw[kt_][2], pw[kt_][3]});
      gf[kt_] = __builtin_bit_cast(bf16x8, u32x4{gw[kt_][0], gw[kt_][1], gw[kt_][2], gw[kt_][3]});
    }
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16x8 a_do = frag_tile(dotl, dt, l15, g), a_q = frag_tile(qtl, dt, l15, g);
#pragma unroll
      for (int kt_ = 0; kt_ < KT; ++kt_) {
        RF_ACC_MFMA(avv[kt_][dt], a_do, pf[kt_]);                            // dV^T[d][key] += dO^T[d][q slots] P[q slots][key]
        RF_ACC_MFMA(akv[kt_][dt], a_q, gf[kt_]);                             // dK^T[d][key] += q~^T[d][q slots] g[q slots][key]
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have left the matrix pipe before the AGPRs are read
#pragma unroll
  for (int kt_ = 0; kt_ < KT; ++kt_) {
    const int kr = kv0 + 16 * kt_ + l15;
    if (kr < s_pad) {
      const bool live = kr < S;                                              // padded keys: zeros (they took part as zero rows)
      bf16_t* krow = dk + (hb + kr) * 128 + 4 * g;
      bf16_t* vrow = dv + (hb + kr) * 128 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        u32x2 a, b;
        a[0] = live ? pack2(akv[kt_][dt][0] * LN2, akv[kt_][dt][1] * LN2) : 0u;
        a[1] = live ? pack2(akv[kt_][dt][2] * LN2, akv[kt_][dt][3] * LN2) : 0u;
        b[0] = live ? pack2(avv[kt_][dt][0], avv[kt_][dt][1]) : 0u;
        b[1] = live ? pack2(avv[kt_][dt][2], avv[kt_][dt][3]) : 0u;
        *(u32x2*)(krow + 16 * dt) = a;
        *(u32x2*)(vrow + 16 * dt) = b;
      }
    }
  }
}

static int g_last_attn_bwd_path = 0;
}  // namespace rf

/* which forms the last rf_attention_bwd launched (rf_attn_bwd_kernel bits): lets a test see what AUTO picked */
extern "C" int rf_debug_last_attn_bwd_path(void) { return rf::g_last_attn_bwd_path; }

extern "C" int rf_attention_bwd(const rf_attn_bwd_desc* d, void* stream) {
  using namespace rf;
  RF_REQUIRE(d != nullptr, RF_ERR_NULL, "rf_attention_bwd: NULL descriptor");
  RF_REQUIRE(d->q && d->k && d->v && d->qt && d->kt && d->o && d->dout && d->dq && d->dk && d->dv && d->dot && d->lse && d->dsum,
             RF_ERR_NULL, "rf_attention_bwd: NULL operand");
  RF_REQUIRE(d->heads > 0 && d->S > 0 && d->s_pad >= d->S && d->s_pad % 64 == 0, RF_ERR_SHAPE,
             "rf_attention_bwd: heads=%d S=%d s_pad=%d (need s_pad %% 64 == 0, s_pad >= S)", d->heads, d->S, d->s_pad);
  RF_REQUIRE(d->mode == 0, RF_ERR_UNSUPPORTED, "rf_attention_bwd: only the plain joint attention (mode 0) has a backward");
  RF_REQUIRE(d->ldo % 8 == 0 && d->lddo % 8 == 0 && aligned16(d->o) && aligned16(d->dout) && aligned16(d->q) && aligned16(d->k) &&
                 aligned16(d->v) && aligned16(d->qt) && aligned16(d->kt) && aligned16(d->dot) && aligned16(d->dq) && aligned16(d->dk) &&
                 aligned16(d->dv),
             RF_ERR_ALIGN, "rf_attention_bwd: operands must be 16-byte aligned with row pitches %% 8 == 0");
  hipStream_t st = (hipStream_t)stream;
  const int S = d->S, sp = d->s_pad, H = d->heads;
  // algorithmic work = the 5 products of a flash backward (S, dP, dV, dK, dq~) x 2 S^2 128 per head = 2.5 x the forward's;
  // this two-kernel, atomics-free form EXECUTES 8 (S three times: statistics, dq~ pass, dK / dV pass; dP twice), 7 when the
  // forward supplied the row statistics (lse_given)
  ProfScope ps(RF_KC_ATTN_BWD, 5.0 * 2.0 * (double)S * (double)S * 128.0 * (double)H, st);
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(sp / 32, H), dim3(256), 0, st, (const bf16_t*)d->o, d->ldo, (const bf16_t*)d->dout,
                     d->lddo, (bf16_t*)d->dot, d->dsum, S, sp);
  RF_LAUNCH_CHECK();
  static int num_cus = 0;
  if (num_cus == 0) {
    int dev = 0;
    RF_CHECK_HIP(hipGetDevice(&dev));
    RF_CHECK_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (num_cus <= 0) num_cus = 256;
  }
  // one workgroup per CU at a time (LDS ring): time ~ rounds of CUs x rows per workgroup.  Pick the shape with the least.
  auto cost = [&](const int rows) { return (int64_t)(((int64_t)((sp + rows - 1) / rows) * H + num_cus - 1) / num_cus) * rows; };
  constexpr int DQ_LDS = AB_RING * 3 * AB_ROWS;
  static bool attr_set = false;
  constexpr int DKV_LDS = AB_RING * (4 * AB_ROWS + 512);
#define RF_DQ_ALL(F) F(2, 8) F(2, 4) F(3, 4)
  if (!attr_set) {
#define RF_DQ_ATTR(QT_, NW_)                                                                                                                 \
  RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<QT_, true, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS)); \
  RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<QT_, false, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS));
    RF_DQ_ALL(RF_DQ_ATTR)
#undef RF_DQ_ATTR
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS / 2));
    attr_set = true;
  }
#define RF_DQ_LAUNCH(QT_, NW_)                                                                                                          \
  {                                                                                                                                     \
    const dim3 grid((sp + 16 * NW_ * QT_ - 1) / (16 * NW_ * QT_), H);                                                                   \
    if (d->lse_given)                                                                                                                   \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<QT_, true, NW_>), grid, dim3(NW_ * 64), DQ_LDS, st, (const bf16_t*)d->q,                  \
                         (const bf16_t*)d->k, (const bf16_t*)d->v, (const bf16_t*)d->kt, (const bf16_t*)d->dout, d->lddo,               \
                         (const float*)d->dsum, d->lse, (bf16_t*)d->dq, S, sp);                                                        \
    else                                                                                                                                \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<QT_, false, NW_>), grid, dim3(NW_ * 64), DQ_LDS, st, (const bf16_t*)d->q,                 \
                         (const bf16_t*)d->k, (const bf16_t*)d->v, (const bf16_t*)d->kt, (const bf16_t*)d->dout, d->lddo,               \
                         (const float*)d->dsum, d->lse, (bf16_t*)d->dq, S, sp);                                                        \
  }
  int dq_form = d->kernel & 0xff, dkv_form = d->kernel & 0xff00;
  RF_REQUIRE((d->kernel & ~0xffff) == 0 && dq_form <= RF_ATTN_BWD_DQ_192 && dkv_form <= RF_ATTN_BWD_DKV_128X2, RF_ERR_UNSUPPORTED,
             "rf_attention_bwd: kernel=0x%x is not an rf_attn_bwd_kernel combination", d->kernel);
  if (dq_form == RF_ATTN_BWD_AUTO) {
    const int64_t c256 = cost(256), c192 = cost(192), c128 = cost(128);
    dq_form = (c192 < c256 && c192 <= c128) ? RF_ATTN_BWD_DQ_192 : c128 < c256 ? RF_ATTN_BWD_DQ_128 : RF_ATTN_BWD_DQ_256;
  }
  if (dq_form == RF_ATTN_BWD_DQ_192) RF_DQ_LAUNCH(3, 4)
  else if (dq_form == RF_ATTN_BWD_DQ_128) RF_DQ_LAUNCH(2, 4)
  else RF_DQ_LAUNCH(2, 8)
#undef RF_DQ_LAUNCH
#undef RF_DQ_ALL
  RF_LAUNCH_CHECK();
#define RF_DKV_LAUNCH(KERNEL, KT_, LDS_)                                                                                                  \
  hipLaunchKernelGGL(KERNEL, dim3((sp + 64 * KT_ - 1) / (64 * KT_), H), dim3(256), LDS_, st, (const bf16_t*)d->q, (const bf16_t*)d->qt,   \
                     (const bf16_t*)d->k, (const bf16_t*)d->v, (const bf16_t*)d->dout, d->lddo, (const bf16_t*)d->dot,                    \
                     (const float*)d->lse, (const float*)d->dsum, (bf16_t*)d->dk, (bf16_t*)d->dv, S, sp)
  {
    // 128-key workgroups also come in a two-per-CU form (2-slot ring, <= 256 registers, 12 dwords of scratch outside the MFMA
    // section): twice the slots per round, each workgroup 1.77x slower for sharing its CU (S = 2560 x 24 heads: 480 workgroups in
    // ONE round of 512 slots, 210 us against 237 us as two rounds of 256)
    if (dkv_form == RF_ATTN_BWD_AUTO) {
      const int64_t wg128 = (int64_t)((sp + 127) / 128) * H;
      const int64_t c2 = ((wg128 + 2 * num_cus - 1) / (2 * num_cus)) * 226;   // 128 rows x 1.77
      dkv_form = (c2 < cost(128) && c2 < cost(192)) ? RF_ATTN_BWD_DKV_128X2 : cost(128) < cost(192) ? RF_ATTN_BWD_DKV_128 : RF_ATTN_BWD_DKV_192;
    }
    if (dkv_form == RF_ATTN_BWD_DKV_128X2) RF_DKV_LAUNCH((attn_bwd_dkv_kernel<2, 2, 2>), 2, DKV_LDS / 2);
    else if (dkv_form == RF_ATTN_BWD_DKV_128) RF_DKV_LAUNCH((attn_bwd_dkv_kernel<2>), 2, DKV_LDS);
    else RF_DKV_LAUNCH((attn_bwd_dkv_kernel<3>), 3, DKV_LDS);
  }
  g_last_attn_bwd_path = dq_form | dkv_form;
#undef RF_DKV_LAUNCH
  RF_LAUNCH_CHECK();
  return RF_OK;
}
