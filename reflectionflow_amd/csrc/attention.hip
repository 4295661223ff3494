// Non-causal flash attention forward for gfx950, head_dim 128, bf16 in / fp32 softmax.
// Replaces F.scaled_dot_product_attention + the head transposes of the reference
// (train_flux/flux/block.py:106-129), including its two condition-token variants:
//   mode 1: additive bias log(c_factor) on the (main <-> condition) blocks  (block.py:115-122)
//   mode 2: (main <-> condition) blocks masked out, union_cond_attn=False   (block.py:106-114)
//
// Design (DESIGN.md "K3"):
//   * one workgroup = 4 waves x 32 query rows; the 64-key K tile [64][128] and V^T tile
//     [128][64] are streamed HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), double buffered,
//     one barrier per tile; both images are XOR-swizzled through the DMA source address so the
//     ds_read_b128 fragment reads are bank-conflict free;
//   * scores are computed TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x16_bf16 with A=K, B=Q):
//     every lane then owns ONE query row (q = lane & 31) and 16 of the 32 keys of a fragment,
//     so the online softmax is lane-local (one cross-lane exchange per tile for the row max);
//   * the output is accumulated transposed as well (O^T = V^T P^T): the softmax rescale factor
//     of a query is a per-lane scalar, and P feeds the MFMA B operand straight from the
//     registers it was exponentiated in.  The key order inside a tile is whatever the S^T
//     accumulator layout yields; the producer GEMM stores V^T with the matching key
//     permutation (bits 2,3 of the key index swapped), so no in-kernel transpose or permute
//     of V or P is needed.
//
// Map of this file (all of it ships in librf_flux.so).  attn_fwd_kernel (v1: the design above, online softmax; masks, bias,
// ragged S) and attn_fwd_kernel_v2 (8 waves x 32 queries, K / VT rings) are the general kernels.  With no mask / bias, a
// prescaled q and S % 256 == 0 the shift-free kernels run instead (no per-tile running maximum): attn5_body =
// attn_fwd_kernel_v5<LAG> (16x16x32 MFMAs) and its split launch attn_fwd_kernel_v5sk<LAG> + attn5_combine_kernel; LAG = false
// needs a proven score bound <= 100 (P = exp2(s)), LAG = true needs nothing (P = exp2(s - m) with a LAGGED row maximum m that
// only moves when a tile overflows 2^30 -- checked on the row sums the kernel forms anyway).  attn_fwd_kernel_v4 is the
// 32x32x16 form of the bounded kernel (shipped above 8192 keys).  The kernel of a launch is picked from the shape or by the
// CALLER per launch (rf_attn_desc.kernel) -- no process-global switch.  The knock-out, ping-pong (v7) and one-wave-per-SIMD
// variants of the round-2/3 studies left the tree in round 5 (git 6cfca97 has them); profiles/r02_attention.md /
// r03_attention.md have the story.
#include "common.hpp"
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

namespace rf {


struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* out;
  int heads, S, s_pad, n_main, mode, nqb;
  int64_t ldo;
  float cross_bias_l2;  // bias * log2(e)
  float sl2;            // softmax scale * log2(e)
  float lag_thresh;     // lagged-max kernels: a lane's 16-key row sum above this re-centres the row (default 2^30)
  int probe;            // block 0 stores its shader-clock probe (only while rf_profile_begin is open)
  float* lse;           // optional [heads][s_pad] fp32: log2-sum-exp2 of each query's (scaled, biased) score row -- the row statistic
                        // rf_attention_bwd needs (rf_attn_desc.lse); rows >= S are not written
};

constexpr int ATT_QBLK = 128;        // query rows per workgroup (4 waves x 32)
constexpr int ATT_KV = 64;           // keys per tile
constexpr int ATT_K_BYTES = ATT_KV * 256;    // 16 KiB
constexpr int ATT_V_BYTES = 128 * 128;       // 16 KiB
constexpr int ATT_STAGE = ATT_K_BYTES + ATT_V_BYTES;

// PRE: q already carries softmax_scale*log2(e) (folded in by the QKV GEMM epilogue): no multiply per score
template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // all query blocks of one head run on one XCD (block b -> XCD b % 8; heads % 8 == 0 in FLUX),
  // so that head's K/V (2.4 MB at S=4608) stay resident in that XCD's L2.
  const int head = blockIdx.x % p.heads;
  const int qb = blockIdx.x / p.heads;
  const int S = p.S;
  const int q_row = qb * ATT_QBLK + w * 32 + l31;
  const int q_ld = q_row < S ? q_row : S - 1;
  const int ntiles = (S + ATT_KV - 1) / ATT_KV;

  const bf16_t* Kh = p.k + (int64_t)head * p.s_pad * 128;
  const bf16_t* Vh = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64);
  // K / V^T tiles are staged with buffer_load_dwordx4 ... lds (SGPR resource + 32-bit per-lane offset): about half the
  // issue cost of global_load_lds with 64-bit per-lane addresses (see gemm_bf16.hip / profiles/r01_gemm_variants.md)
  const rsrc_t rsK = RF_MAKE_RSRC(Kh), rsV = RF_MAKE_RSRC(Vh);

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q, h) holds d = ks*16 + h*8 .. +8 ---
  bf16x8 qf[8];
  {
    const bf16_t* qp = p.q + ((int64_t)head * p.s_pad + q_ld) * 128 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- LDS-DMA source offsets (elements) -------------------------------------------------
  // K tile: instruction i of wave w fills rows (i*4+w)*4 + lane/16, physical chunk lane%16
  int k_row[4], k_chunk[4];
  int v_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + w) * 4 + (lane >> 4);
    k_row[i] = row;
    k_chunk[i] = ((lane & 15) ^ (row & 15)) * 8;
    const int vrow = (i * 4 + w) * 8 + (lane >> 3);
    v_off[i] = vrow * 64 + (((lane & 7) ^ ((vrow >> 1) & 7)) * 8);
  }

  auto stage = [&](int t, int buf) {
    char* base = smem + buf * ATT_STAGE;
    const int kv0 = t * ATT_KV;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kv = kv0 + k_row[i];
      kv = kv < S ? kv : S - 1;
      RF_BUF_LOAD_LDS(rsK, (lds_void*)(base + (i * 4 + w) * 1024), (uint32_t)(kv * 256 + k_chunk[i] * 2), 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RF_BUF_LOAD_LDS(rsV, (lds_void*)(base + ATT_K_BYTES + (i * 4 + w) * 1024), (uint32_t)(v_off[i] * 2), t * (128 * 64 * 2));
  };

  // ---- per-lane bias for the two key regions (log2 domain) -------------------------------
  const float NEG_INF = -__builtin_huge_valf();
  const bool q_is_cond = q_row >= p.n_main;
  float badd_main = 0.f, badd_cond = 0.f;
  if (p.mode == 1) {
    badd_main = q_is_cond ? p.cross_bias_l2 : 0.f;
    badd_cond = q_is_cond ? 0.f : p.cross_bias_l2;
  } else if (p.mode == 2) {
    badd_main = q_is_cond ? NEG_INF : 0.f;
    badd_cond = q_is_cond ? 0.f : NEG_INF;
  }

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1e30f;  // running max (log2 domain), finite so that (-inf) - m is well defined
  float l_run = 0.f;     // this lane's partial row sum (its 32 of the 64 keys per tile)

  const int k_swz = l31 & 15;
  const int v_swz = (l31 >> 1) & 7;

  stage(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    // Tile t's LDS-DMA must have landed for EVERY wave before anyone reads it.  The wait is explicit:
    // hipcc does not reliably track LDS-DMA across a loop back edge (it hoisted its own vmcnt(0)
    // out of this loop, leaving tiles >= 1 unguarded -- a cold-cache race).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ... and all waves have finished reading tile t-1's buffer
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
    const char* kb = smem + (t & 1) * ATT_STAGE;
    const char* vb = kb + ATT_K_BYTES;
    const int kv0 = t * ATT_KV;

    // ---- S^T = K Q^T : 2 key blocks x 8 k-steps -----------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kvb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kb + (kvb * 32 + l31) * 256 + (((ks * 2 + h) ^ k_swz) << 4));
        sacc[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kvb], 0, 0, 0);
      }
    }

    // ---- online softmax (lane-local: this lane's query, 32 of the tile's 64 keys) --------
    const bool special = (kv0 + ATT_KV > S) || (p.mode != 0 && kv0 < p.n_main && kv0 + ATT_KV > p.n_main);
    float tmax = NEG_INF;
    if (!special) {
      const float badd = (p.mode != 0 && kv0 >= p.n_main) ? badd_cond : badd_main;
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = PRE ? sacc[kvb][r] + badd : sacc[kvb][r] * p.sl2 + badd;
          sacc[kvb][r] = x;
          tmax = fmaxf(tmax, x);
        }
    } else {
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kvb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float badd = kv >= S ? NEG_INF : (kv >= p.n_main ? badd_cond : badd_main);
          const float x = PRE ? sacc[kvb][r] + badd : sacc[kvb][r] * p.sl2 + badd;
          sacc[kvb][r] = x;
          tmax = fmaxf(tmax, x);
        }
    }
    {  // the other 32 keys of this query live in lane ^ 32: v_permlane32_swap (VALU, no LDS round trip)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
      tmax = fmaxf(tmax, __uint_as_float(h ? sw[0] : sw[1]));
    }
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 t8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = __builtin_amdgcn_exp2f(sacc[kvb][s2 * 8 + j] - m_new);
          psum += pv;
          t8[j] = f2bf(pv);
        }
        pf[kvb * 2 + s2] = t8;
      }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {  // wave-uniform: after the first tiles the running max rarely moves
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }

    // ---- O^T += V^T P^T : 4 d-blocks x 4 key steps ---------------------------------------
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 vf = *(const bf16x8*)(vb + (db * 32 + l31) * 128 + (((s * 2 + h) ^ v_swz) << 4));
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oacc[db], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane owns query q_row, d = db*32 + 8*rg + 4*h + (0..3) ----------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (p.lse != nullptr && h == 0 && q_row < S) p.lse[(int64_t)head * p.s_pad + q_row] = m_run + __builtin_amdgcn_logf(l_tot);
  if (q_row < S) {
    bf16_t* orow = p.out + (int64_t)q_row * p.ldo + head * 128 + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        u32x2 v;
        v[0] = pack2(oacc[db][rg * 4 + 0] * inv, oacc[db][rg * 4 + 1] * inv);
        v[1] = pack2(oacc[db][rg * 4 + 2] * inv, oacc[db][rg * 4 + 3] * inv);
        *(u32x2*)(orow + db * 32 + rg * 8) = v;
      }
  }
}

// =================================================================================================
// v2: 8 waves x 32 queries per workgroup (one K/V tile feeds 256 queries: half the DMA traffic per
// flop), K in a 4-stage and V^T in a 3-stage LDS ring filled two tiles ahead under a COUNTED vmcnt
// (the queue is never drained in the main loop), and the score MFMAs of tile t+1 are issued next to the
// softmax VALU of tile t inside each wave, so the matrix pipe has work while a wave exponentiates.
//   per iteration t:   wait{K(t+1), V(t)} -> barrier -> DMA{K(t+3), V(t+2)} -> S_next = K(t+1) Q^T
//                      -> softmax(S_cur) -> O^T += V(t)^T P^T -> S_cur = S_next
// Hazards: a ring slot is re-filled >= 1 barrier after every wave finished reading it (K slot of tile
// t-1 is re-filled in iteration t, V slot of tile t-1 in iteration t, both after barrier B_t); a tile is
// read only after every wave's counted vmcnt for it and a barrier.
constexpr int ATT2_NK = 4, ATT2_NV = 3;
constexpr int ATT2_LDS = (ATT2_NK + ATT2_NV) * 16384;

#define RF_ATT_WAIT_BARRIER(allowed)                                             \
  do {                                                                           \
    if ((allowed) >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         \
    else if ((allowed) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    \
    else if ((allowed) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    \
    else if ((allowed) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    __syncthreads();                                                             \
  } while (0)

// GENERIC = false: no condition bias/mask and S % 64 == 0 (every BASELINE config with union attention): the loop
// body is ONE basic block, so the scheduler can interleave the next tile's score MFMAs with the softmax VALU.
// GENERIC = true: per-element key masks/biases (ragged tail, attn.c_factor, union_cond_attn=False).
template <bool GENERIC, bool PRE>
__global__ __launch_bounds__(512) void attn_fwd_kernel_v2(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int qb = blockIdx.x / p.heads;
  const int S = p.S;
  const int q_row = qb * 256 + w * 32 + l31;
  const int q_ld = q_row < S ? q_row : S - 1;
  const int nt = (S + ATT_KV - 1) / ATT_KV;
  const bf16_t* Kh = p.k + (int64_t)head * p.s_pad * 128;
  const bf16_t* Vh = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64);
  // K / V^T tiles are staged with buffer_load_dwordx4 ... lds (SGPR resource + 32-bit per-lane offset): about half the
  // issue cost of global_load_lds with 64-bit per-lane addresses (see gemm_bf16.hip / profiles/r01_gemm_variants.md)
  const rsrc_t rsK = RF_MAKE_RSRC(Kh), rsV = RF_MAKE_RSRC(Vh);
  char* const kring = smem;
  char* const vring = smem + ATT2_NK * 16384;

  bf16x8 qf[8];
  {
    const bf16_t* qp = p.q + ((int64_t)head * p.s_pad + q_ld) * 128 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }
  // DMA pieces: 2 of the 16 x 1 KiB pieces of a K tile (4 rows each) and of a V^T tile (8 rows each) per wave
  int k_row[2], k_chunk[2], v_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 8 + w) * 4 + (lane >> 4);
    k_row[i] = row;
    k_chunk[i] = ((lane & 15) ^ (row & 15)) * 8;
    const int vrow = (i * 8 + w) * 8 + (lane >> 3);
    v_off[i] = vrow * 64 + (((lane & 7) ^ ((vrow >> 1) & 7)) * 8);
  }
  auto issue_k = [&](int t) {
    char* base = kring + (t % ATT2_NK) * 16384;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int kv = t * ATT_KV + k_row[i];
      kv = kv < S ? kv : S - 1;
      RF_BUF_LOAD_LDS(rsK, (lds_void*)(base + (i * 8 + w) * 1024), (uint32_t)(kv * 256 + k_chunk[i] * 2), 0);
    }
  };
  auto issue_v = [&](int t) {
    char* base = vring + (t % ATT2_NV) * 16384;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      RF_BUF_LOAD_LDS(rsV, (lds_void*)(base + (i * 8 + w) * 1024), (uint32_t)(v_off[i] * 2), t * (128 * 64 * 2));
  };

  const float NEG_INF = -__builtin_huge_valf();
  const bool q_is_cond = q_row >= p.n_main;
  float badd_main = 0.f, badd_cond = 0.f;
  if (p.mode == 1) {
    badd_main = q_is_cond ? p.cross_bias_l2 : 0.f;
    badd_cond = q_is_cond ? 0.f : p.cross_bias_l2;
  } else if (p.mode == 2) {
    badd_main = q_is_cond ? NEG_INF : 0.f;
    badd_cond = q_is_cond ? 0.f : NEG_INF;
  }
  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const int k_swz = l31 & 15;
  const int v_swz = (l31 >> 1) & 7;

  auto qk = [&](int t, f32x16 (&sacc)[2]) {
    const char* kb = kring + (t % ATT2_NK) * 16384;
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kvb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kb + (kvb * 32 + l31) * 256 + (((ks * 2 + h) ^ k_swz) << 4));
        sacc[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kvb], 0, 0, 0);
      }
    }
  };

  // ---- prologue: queue order  K0 | K1 V0 | K2 V1 -----------------------------------------------
  issue_k(0);
  if (nt > 1) issue_k(1);
  issue_v(0);
  if (nt > 2) issue_k(2);
  if (nt > 1) issue_v(1);
  {
    const int allowed = 2 * ((nt > 1) + 1 + (nt > 2) + (nt > 1));  // everything younger than K0
    RF_ATT_WAIT_BARRIER(allowed);
  }
  f32x16 s_cur[2], s_nxt[2];
  qk(0, s_cur);

  for (int t = 0; t < nt; ++t) {
    // K(t+1), V(t) must have landed; K(t+2), V(t+1) (issued last iteration) may stay in flight
    {
      const int allowed = 2 * ((t + 2 < nt) + (t + 1 < nt));
      RF_ATT_WAIT_BARRIER(allowed);
    }
    if (t + 3 < nt) issue_k(t + 3);
    if (t + 2 < nt) issue_v(t + 2);
    // scores of the NEXT tile (clamped on the last iteration: that K slot is still resident, the result is
    // unused) -- unconditional so that these MFMAs share a basic block with the softmax VALU below
    qk(t + 1 < nt ? t + 1 : nt - 1, s_nxt);

    const int kv0 = t * ATT_KV;
    float tmax = NEG_INF;
    if (!GENERIC) {
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = PRE ? s_cur[kvb][r] : s_cur[kvb][r] * p.sl2;
          s_cur[kvb][r] = x;
          tmax = fmaxf(tmax, x);
        }
    } else {
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kvb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float badd = kv >= S ? NEG_INF : (kv >= p.n_main ? badd_cond : badd_main);
          const float x = PRE ? s_cur[kvb][r] + badd : s_cur[kvb][r] * p.sl2 + badd;
          s_cur[kvb][r] = x;
          tmax = fmaxf(tmax, x);
        }
    }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
      tmax = fmaxf(tmax, __uint_as_float(h ? sw[0] : sw[1]));
    }
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 t8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = __builtin_amdgcn_exp2f(s_cur[kvb][s2 * 8 + j] - m_new);
          psum += pv;
          t8[j] = f2bf(pv);
        }
        pf[kvb * 2 + s2] = t8;
      }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    const char* vb = vring + (t % ATT2_NV) * 16384;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 vf = *(const bf16x8*)(vb + (db * 32 + l31) * 128 + (((s * 2 + h) ^ v_swz) << 4));
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oacc[db], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb) s_cur[kvb] = s_nxt[kvb];
  }

  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = l_run + __uint_as_float(h ? sw[0] : sw[1]);
  }
  const float inv = 1.0f / l_tot;
  if (p.lse != nullptr && h == 0 && q_row < S) p.lse[(int64_t)head * p.s_pad + q_row] = m_run + __builtin_amdgcn_logf(l_tot);
  if (q_row < S) {
    bf16_t* orow = p.out + (int64_t)q_row * p.ldo + head * 128 + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        u32x2 v;
        v[0] = pack2(oacc[db][rg * 4 + 0] * inv, oacc[db][rg * 4 + 1] * inv);
        v[1] = pack2(oacc[db][rg * 4 + 2] * inv, oacc[db][rg * 4 + 3] * inv);
        *(u32x2*)(orow + db * 32 + rg * 8) = v;
      }
  }
}


// =================================================================================================
// v4 -- bounded-score attention.  The v2 workgroup (8 waves x 32 queries, K / V^T rings filled by LDS-DMA under
// counted waits) with the loop body rebuilt around what the ISA of v2 showed (profiles/r02_attention.md): hipcc
// serialises every MFMA of v2 behind its own `ds_read_b128; s_waitcnt lgkmcnt(0)` (one fragment buffer: ~an LDS
// round trip per MFMA instead of 32 cycles), and its online softmax costs ~154 VALU instructions per tile and wave.
//   * FLUX normalises q and k per head (RMSNorm, block.py:38-41,60-67) before RoPE, so |q.k| <= |q||k| is bounded
//     by the norm weights alone: |s| <= sqrt(128) max|w_q| max|w_k| (x log2 e in the exp2 domain; ~19 for weights
//     near 1).  softmax is shift invariant, so with a PROVEN bound the shift can simply be 0: P = exp2(s), no running
//     maximum, no max exchange, no rescale of O -- and nothing is lost: bf16 P and fp32 O / l keep the same relative
//     precision at every magnitude, |s| <= 100 stays far inside their exponent range even summed over 2^17 keys.
//     The caller passes the bound (rf_attention_fwd's `score_bound`; the engine derives it from the block's norm
//     weights); without one -- or above 100 -- the online-softmax kernels v2 / v1 run.  32 exp + 32 add + 16 cvt_pk
//     per tile and wave are all the VALU work left;
//   * a tile's work is three INDEPENDENT streams:  O^T += V(t-1)^T P(t-1)^T (16 MFMA) | S(t+1)^T = K(t+1) Q^T (16 MFMA) |
//     P(t) = exp2(S(t)) (VALU) -- nothing waits for anything produced in the same tile.  The 32 MFMAs run as 16
//     groups of two, fenced by sched_barrier(0); every group first issues the fragment reads of the NEXT group into
//     the other half of a double buffer, so reads are always one group (64-128 cycles) ahead of their MFMA and the
//     compiler's own counted lgkmcnt keeps them in flight; each group carries a fixed slice of the VALU work;
//   * the ring position is a template parameter (loop unrolled by the ring size 4): every LDS address is one of 12
//     per-lane registers + an immediate -- no address arithmetic in the loop.
// Preconditions (dispatch): mode 0, S % 256 == 0 (whole rounds of the 4-slot ring), q prescaled, 0 < score_bound <= 100.
constexpr int ATT4_RING = 4;

// counted wait for this wave's LDS-DMA + a RAW s_barrier: __syncthreads() would add `s_waitcnt vmcnt(0)` while LDS-DMA
// is in flight and drain the ring every tile (seen in the ISA); every wave's fragment reads of the previous tile
// are already retired by the lgkmcnt waits in front of the MFMAs that consumed them
#define RF_ATT4_WAIT_BARRIER(allowed)                                            \
  do {                                                                           \
    if ((allowed) >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         \
    else if ((allowed) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    \
    else if ((allowed) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    __builtin_amdgcn_s_barrier();                                                \
  } while (0)
// pin values to this program point: hipcc's IR passes otherwise sink the pure exp2 / cvt chains of one tile into the
// next tile's first block (right in front of their MFMA), which undoes the software pipeline (seen in the ISA)
#define RF_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
constexpr int ATT4_LDS = 2 * ATT4_RING * 16384;

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

__device__ unsigned long long g_attn_clk_probe[4];   // see ClkProbe (common.hpp)

__global__ __launch_bounds__(512) void attn_fwd_kernel_v4(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ClkProbe clk;
  if (p.probe) clk.begin();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int qb = blockIdx.x / p.heads;
  const int S = p.S;
  const int q_row = qb * 256 + w * 32 + l31;
  const int q_ld = q_row < S ? q_row : S - 1;
  const int nt = S / ATT_KV;
  const bf16_t* Kh = p.k + (int64_t)head * p.s_pad * 128;
  const bf16_t* Vh = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64);
  const rsrc_t rsK = RF_MAKE_RSRC(Kh), rsV = RF_MAKE_RSRC(Vh);
  char* const kring = smem;
  char* const vring = smem + ATT4_RING * 16384;

  bf16x8 qf[8];
  {
    const bf16_t* qp = p.q + ((int64_t)head * p.s_pad + q_ld) * 128 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }
  // DMA pieces: the 16 x 1 KiB pieces of a K tile (4 rows each) and of a V^T tile (8 rows each).  Waves 0-3 issue ALL of them (4 + 4
  // each), waves 4-7 none: the SIMD's arbiter lets the older wave of a pair run ahead and it then waits ~1000 clocks per tile at the
  // barrier (tile stamps of v5, DESIGN K3M) -- the ~60-clock issue of a piece is free there and on the critical path in its partner.
  const bool dma_owner = w < 4;
  uint32_t k_src[4], v_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pc = i * 4 + (w & 3);
    const int row = pc * 4 + (lane >> 4);
    k_src[i] = (uint32_t)(row * 256 + (((lane & 15) ^ (row & 15)) * 16));
    const int vrow = pc * 8 + (lane >> 3);
    v_src[i] = (uint32_t)((vrow * 64 + (((lane & 7) ^ ((vrow >> 1) & 7)) * 8)) * 2);
  }
  auto issue_k = [&](int t, int slot) {
    if (!dma_owner) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RF_BUF_LOAD_LDS(rsK, (lds_void*)(kring + slot * 16384 + (i * 4 + (w & 3)) * 1024), k_src[i], t * (ATT_KV * 256));
  };
  auto issue_v = [&](int t, int slot) {
    if (!dma_owner) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      RF_BUF_LOAD_LDS(rsV, (lds_void*)(vring + slot * 16384 + (i * 4 + (w & 3)) * 1024), v_src[i], t * (128 * 64 * 2));
  };
  // counted wait + barrier: n issue calls (4 instructions each in an owner wave, none elsewhere) may stay in flight
  auto wait_barrier = [&](const int n) {
    if (n >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // fragment read addresses: 8 (K, per k-step) + 4 (V^T, per key step) per-lane registers; ring slot, key block and
  // d block are immediates
  const char* k_rd[8];
  const char* v_rd[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = kring + l31 * 256 + (((ks * 2 + h) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) v_rd[s2] = vring + l31 * 128 + (((s2 * 2 + h) ^ ((l31 >> 1) & 7)) << 4);

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float l_run = 0.f;
  f32x16 s_cur[2], s_nxt[2];
  bf16x8 pf[4];   // P(t-1): B-operand fragments of the pending PV product

  // ---- prologue: DMA queue order K0 | K1 | K2 V0, then per tile t the pair K(t+3) V(t+1) ------------------------------
  {  // V ring slot 3 stands in for V(-1): tile 0 multiplies it with P(-1) = 0, so it must be finite
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4*)(vring + 3 * 16384 + (i * 8 + w) * 1024 + lane * 16) = z;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[i][j] = (bf16_t)0.f;
  }
  issue_k(0, 0);
  if (nt > 1) issue_k(1, 1);
  if (nt > 2) issue_k(2, 2);
  issue_v(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill above
  wait_barrier((nt > 1) + (nt > 2) + 1);   // K0 landed
#pragma unroll
  for (int kvb = 0; kvb < 2; ++kvb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_cur[kvb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 kf = *(const bf16x8*)(k_rd[ks] + kvb * 32 * 256);
      s_cur[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_cur[kvb], 0, 0, 0);
    }
  }

  // one tile; TS = t % 4 is the ring position.  There is ONE body: tile 0 multiplies the zeroed P(-1) with the
  // zero-filled V slot 3, the last tile computes scores of a stale K slot that nobody reads.
  auto tile = [&](const int t, auto ts_tag) {
    constexpr int TS = decltype(ts_tag)::value;
    constexpr int KSLOT = (TS + 1) % 4, VSLOT = (TS + 3) % 4;
    // needed now: K(t+1) [next scores], V(t-1) [pending PV]; may stay in flight: K(t+2), V(t)
    wait_barrier((t + 2 < nt) + 1);
    if (t + 3 < nt) issue_k(t + 3, (TS + 3) % 4);
    if (t + 1 < nt) issue_v(t + 1, KSLOT);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fr[2][2];   // fragment double buffer: group g multiplies fr[g & 1][0..1]
    float psum = 0.f;
    // fragment pair of group G: G < 8 -> V^T(t-1) [d block G/2, key steps (G%2)*2 + 0,1], else K(t+1) [key block (G-8)/4, k-steps ((G-8)%4)*2 + 0,1]
    auto load_group = [&](auto gtag) {
      constexpr int G = decltype(gtag)::value;
      if constexpr (G < 8) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          fr[G & 1][e] = *(const bf16x8*)(v_rd[(G % 2) * 2 + e] + VSLOT * 16384 + (G / 2) * 32 * 128);
      } else if constexpr (G < 16) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          fr[G & 1][e] = *(const bf16x8*)(k_rd[((G - 8) % 4) * 2 + e] + KSLOT * 16384 + ((G - 8) / 4) * 32 * 256);
      }
    };
    load_group(std::integral_constant<int, 0>{});
    // ---- first half: pending PV product (8 groups of 2 MFMA) | P = exp2(S) in place + row sums (4 per group) ----
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto gtag) {
      constexpr int G = decltype(gtag)::value;
      load_group(std::integral_constant<int, G + 1>{});
#pragma unroll
      for (int e = 0; e < 2; ++e)
        oacc[G / 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[G & 1][e], pf[(G % 2) * 2 + e], oacc[G / 2], 0, 0, 0);
      {
        float e4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e4[j] = __builtin_amdgcn_exp2f(s_cur[(G * 4 + j) / 16][(G * 4 + j) % 16]);
        RF_PIN4(e4[0], e4[1], e4[2], e4[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s_cur[(G * 4 + j) / 16][(G * 4 + j) % 16] = e4[j];
          psum += e4[j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("" : "+v"(psum));
    l_run += psum;
    // ---- second half: next tile's scores (8 groups of 2 MFMA) | pack P into the PV operand (4 per group) ---------
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto gtag) {
      constexpr int G = decltype(gtag)::value + 8;
      load_group(std::integral_constant<int, G + 1>{});
      constexpr int kvb = (G - 8) / 4;
      constexpr int ksb = ((G - 8) % 4) * 2;
      if constexpr (ksb == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_nxt[kvb][r] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
        s_nxt[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[G & 1][e], qf[ksb + e], s_nxt[kvb], 0, 0, 0);
      {
        constexpr int i0 = (G - 8) * 4;
        uint32_t w0 = pack2(s_cur[i0 / 16][i0 % 16], s_cur[i0 / 16][i0 % 16 + 1]);
        uint32_t w1 = pack2(s_cur[i0 / 16][i0 % 16 + 2], s_cur[i0 / 16][i0 % 16 + 3]);
        asm volatile("" : "+v"(w0), "+v"(w1));
        u32x4 t4 = __builtin_bit_cast(u32x4, pf[i0 / 8]);
        t4[(i0 % 8) / 2] = w0;
        t4[(i0 % 8) / 2 + 1] = w1;
        pf[i0 / 8] = __builtin_bit_cast(bf16x8, t4);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb) s_cur[kvb] = s_nxt[kvb];
  };
  for (int t = 0; t < nt; t += 4) {   // unrolled by the ring size (dispatch guarantees nt % 4 == 0): one exit
    tile(t, std::integral_constant<int, 0>{});
    tile(t + 1, std::integral_constant<int, 1>{});
    tile(t + 2, std::integral_constant<int, 2>{});
    tile(t + 3, std::integral_constant<int, 3>{});
  }

  if (p.probe) clk.end(g_attn_clk_probe);
  // ---- epilogue: the last pending product O^T += V(nt-1)^T P(nt-1)^T ------------------------------------------------
  wait_barrier(0);
  {
    const int vslot = (nt - 1) % ATT4_RING;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const bf16x8 vf = *(const bf16x8*)(v_rd[s2] + vslot * 16384 + db * 32 * 128);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], oacc[db], 0, 0, 0);
      }
  }
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = l_run + __uint_as_float(h ? sw[0] : sw[1]);
  }
  const float inv = 1.0f / l_tot;
  if (p.lse != nullptr && h == 0 && q_row < S) p.lse[(int64_t)head * p.s_pad + q_row] = __builtin_amdgcn_logf(l_tot);   // bounded form: m == 0
  if (q_row < S) {
    bf16_t* orow = p.out + (int64_t)q_row * p.ldo + head * 128 + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        u32x2 v;
        v[0] = pack2(oacc[db][rg * 4 + 0] * inv, oacc[db][rg * 4 + 1] * inv);
        v[1] = pack2(oacc[db][rg * 4 + 2] * inv, oacc[db][rg * 4 + 3] * inv);
        *(u32x2*)(orow + db * 32 + rg * 8) = v;
      }
  }
}

// =================================================================================================
// v5 -- v4's schedule on v_mfma_f32_16x16x32_bf16.  Under the 1.4 kW cap the 16x16 shape sustains ~15 % more FLOP/s than
// 32x32x16 (half the accumulator bytes per MAC; tools/ubench/mfma_power.py) and attention runs at the cap (1.8 GHz).
// Same workgroup, rings, DMA queue, counted waits, three independent streams per tile and 16 fenced groups -- a group is now
// four 16-cycle MFMAs on two fragments (each fragment feeds both 16-query tiles of the wave) instead of two 32-cycle ones.
//   * a wave's 32 queries are two q-tiles; S^T, P^T and O^T tiles have q = lane & 15 in BOTH MFMA shapes, so P still goes
//     from the score accumulators into the PV product's B operand inside a lane;
//   * the PV B operand of a 32-key block wants, in lane group g = lane >> 4, the keys VT stores at positions 8g .. 8g+7
//     of the block = {0-3, 8-11}, {4-7, 12-15}, {16-19, 24-27}, {20-23, 28-31} (the permutation the QKV epilogue already
//     writes for the 32x32 kernel).  A 16x16 score tile leaves rows 4g .. 4g+3 in lane group g, so each 32-key block is
//     scored as TWO tiles whose MFMA rows are the keys T0 = {0-7, 16-23} and T1 = T0 + 8: lane group g then holds
//     exactly its eight keys, four from each tile.  The row permutation costs nothing: it is the per-lane row offset of
//     the K fragment read;
//   * K ring swizzle: chunk ^ ((row & 7) | ((row >> 1) & 8)) -- the 16 rows of a T tile get 16 different chunk slots
//     (for every fragment the term is just lane & 15).  VT ring: v4's (row >> 1) & 7.
// normalise a wave's O^T accumulators by its row sums and store its NQT x 16 output rows (v5 accumulator layout); q0w = the
// wave's first query row
template <int NQT, bool NARROW = false>
__device__ __forceinline__ void attn5_finish(const AttnParams& p, f32x4 (&oacc)[8][NQT], float (&l_run)[NQT], const int lane, const int head,
                                             const int q0w, const float (&m_row)[NQT]) {
  const int l15 = lane & 15, g = lane >> 4;
  const int S = p.S;
  const bool wide = !NARROW && (p.ldo & 7) == 0;   // 16-byte stores need 16-byte aligned rows
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    // a query's keys are spread over the four lane groups: lanes l15, l15 + 16, + 32, + 48
    float l_tot = l_run[qt];
    l_tot += __shfl_xor(l_tot, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.0f / l_tot;
    const int q_row = q0w + qt * 16 + l15;
    // the row statistic of the backward: P = exp2(s - m) summed to l  =>  log2-sum-exp2 = m + log2 l  (m = the lagged maximum, 0 in the bounded form)
    if (p.lse != nullptr && g == 0 && q_row < S) p.lse[(int64_t)head * p.s_pad + q_row] = m_row[qt] + __builtin_amdgcn_logf(l_tot);
    if (wide) {
      // The store tail is store-ISSUE-bound (cdna_hip_programming.md T21): 4 x 16 bytes per lane and q-tile instead of 8 x 8.  A lane
      // holds d = 4g .. 4g+3 of every d tile; lane groups g and g ^ 1 trade so that the even group owns d = 8 (g >> 1) .. +7 of
      // the even d tile of a pair and the odd group the same columns of the odd one.
      bf16_t* orow = p.out + (int64_t)(q_row < S ? q_row : 0) * p.ldo + head * 128 + 8 * (g >> 1);
#pragma unroll
      for (int dp = 0; dp < 4; ++dp) {
        u32x2 ev, od;
        ev[0] = pack2(oacc[2 * dp][qt][0] * inv, oacc[2 * dp][qt][1] * inv);
        ev[1] = pack2(oacc[2 * dp][qt][2] * inv, oacc[2 * dp][qt][3] * inv);
        od[0] = pack2(oacc[2 * dp + 1][qt][0] * inv, oacc[2 * dp + 1][qt][1] * inv);
        od[1] = pack2(oacc[2 * dp + 1][qt][2] * inv, oacc[2 * dp + 1][qt][3] * inv);
        const u32x2 send = (g & 1) ? ev : od;
        u32x2 recv;
        recv[0] = (uint32_t)__shfl_xor((int)send[0], 16);
        recv[1] = (uint32_t)__shfl_xor((int)send[1], 16);
        u32x4 v;
        if (g & 1) { v[0] = recv[0]; v[1] = recv[1]; v[2] = od[0]; v[3] = od[1]; }
        else { v[0] = ev[0]; v[1] = ev[1]; v[2] = recv[0]; v[3] = recv[1]; }
        if (q_row < S) *(u32x4*)(orow + (2 * dp + (g & 1)) * 16) = v;
      }
    } else if (q_row < S) {
      bf16_t* orow = p.out + (int64_t)q_row * p.ldo + head * 128 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        u32x2 v;
        v[0] = pack2(oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv);
        v[1] = pack2(oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv);
        *(u32x2*)(orow + dt * 16) = v;
      }
    }
  }
}


// One (head, 256-query block) over the key tiles [t0, t0 + nt), nt % 4 == 0.  partial == nullptr: the range is the whole
// key axis -> normalise and store the output rows.  Otherwise (split launch): leave the raw O^T accumulators and row sums of
// this key range in the 136 KiB slot `partial` ([16 quads][512 threads] x 16 B, then l [2][512] and m [2][512] floats),
// thread-linear: attn5_combine_kernel adds the slots of a block's pieces thread by thread (scaled by exp2(m_piece - max m):
// exactly 1 without LAG, where every m is 0) and finishes it.
//
// LAG = true: the weight-independent form.  Every query row carries a LAGGED maximum m (exp2 domain): P = exp2(s - m), and -m
// rides into the score MFMAs as their C operand (a persistent f32x4 per q-tile: no VALU).  m starts as the exact row maximum
// of the piece's first key tile; afterwards it moves only when a tile overflows: a lane's 16-key row sum (which the kernel
// forms anyway) above lag_thresh (2^30) -- one compare per q-tile and tile.  The wave that trips re-centres ALONE, between the
// two halves of the tile: it recomputes the tile's raw scores from the K ring slot (still resident), takes their exact maximum,
// scales O and l by exp2(m_old - m_new) -- PV(t-1) is complete at that point, P(t) not yet packed, so everything at the old
// scale is scaled exactly once (cdna_hip_programming.md T13 hazard) -- and re-exponentiates the tile against m_new.  fp32 l / O
// and bf16 P hold 2^30 * S * |V| with room to spare, so between re-centrings nothing is lost; on i.i.d. data the slow path
// never runs after the first tile.  With LAG = false, m == 0 throughout (the caller's proven bound |s| <= 100 makes that safe).
//
// NQT = q-tiles (16 queries) of THIS wave: 2 everywhere except in waves 4-7 of the 192-query workgroups of the mixed-size launch
// (attn_fwd_kernel_v5mix), which carry one.  q0w = the wave's first query row.  Every wave of a workgroup executes the same
// barriers whatever its NQT.  ROT: this wave runs the halves of an interval in the order G, F (waves 4-7); DMA: see below.
// VAR = compile-time schedule options (every one computes the same result; the library instantiates ATT5_VAR = 256 | 2048 only):
//   4, 32  fragment reads ONE / THREE groups ahead of their MFMAs instead of two (4 | 32: four)
//   8      8-byte epilogue stores
//   128, 256, 384, ... (bits 7-9)  wave-priority scheme 1..7 (256 = scheme 2: s_setprio 2 in G, 0 in F -- shipped)
//   2048   row sums on the matrix pipe (SUMM; bounded form only -- shipped)
// (The timing knock-outs and s_memtime tile stamps of the round-3 studies in profiles/ are not in this tree: git 6cfca97.)
template <bool PROBE, bool LAG, int VAR = 0, int NQT = 2, bool ROT = false, int DMA = 1>
__device__ __forceinline__ void attn5_body(const AttnParams& p, char* smem, const int tid, const int lane, const int w,
                                           const int head, const int q0w, const int t0, const int nt, float* partial, ClkProbe& clk) {
  const int l15 = lane & 15, g = lane >> 4;
  const int S = p.S;
  constexpr int PRIO = (VAR >> 7) & 7;
  // SUMM: the row sums l = sum_k P ride on the matrix pipe -- O^T gets a ninth "d tile" whose V^T rows are all ones (a constant register
  // fragment, no LDS read): 2 NQT MFMAs per key tile instead of 16 NQT dependent v_add_f32 in the exp2 half (a wave's adds cost ~6 clocks
  // each on its critical path, DESIGN K3M; the matrix pipe is ~60 % busy).  l then sums the bf16-ROUNDED P, the values the numerator uses.
  // Not with LAG, whose overflow test needs the sums before P is packed.
  constexpr bool SUMM = !LAG && (VAR & 2048) != 0;
  const bf16_t* Kh = p.k + (int64_t)head * p.s_pad * 128;
  const bf16_t* Vh = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64);
  const rsrc_t rsK = RF_MAKE_RSRC(Kh), rsV = RF_MAKE_RSRC(Vh);
  char* const kring = smem;
  char* const vring = smem + ATT4_RING * 16384;
  if constexpr (PRIO == 4) __builtin_amdgcn_s_setprio(ROT ? 1 : 0);

  // Q B-operand fragments: [q-tile][d step of 32]: lane -> query qt*16 + l15, d = 32 ds + 8g .. +8
  bf16x8 qf[NQT][4];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    const int q_row = q0w + qt * 16 + l15;
    const bf16_t* qp = p.q + ((int64_t)head * p.s_pad + (q_row < S ? q_row : S - 1)) * 128 + g * 8;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[qt][ds] = *(const bf16x8*)(qp + ds * 32);
  }
  // DMA pieces: the 16 x 1 KiB pieces of a K tile (4 rows each) and of a V^T tile (8 rows each).  DMA = 1: every wave issues 2 + 2 of
  // them; DMA = 2: this wave issues 4 + 4 (waves 0-3 of a workgroup whose waves 4-7 run with DMA = 0 and issue none).  The SIMD's
  // arbiter lets the older wave of a pair run ahead; it then WAITS ~1000 clocks per tile at the barrier for its partner (tile stamps,
  // DESIGN K3M), so the ~60-clock issue of a piece is free there and on the critical path in the partner.
  constexpr int NPC = DMA == 2 ? 4 : (DMA == 1 ? 2 : 0), PSTR = DMA == 2 ? 4 : 8;   // pieces per tile of this wave; piece = i * PSTR + (w % PSTR)
  uint32_t k_src[NPC > 0 ? NPC : 1], v_src[NPC > 0 ? NPC : 1];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int pc = i * PSTR + (w % PSTR);
    const int row = pc * 4 + (lane >> 4);
    const int ksw = (row & 7) | ((row >> 1) & 8);
    k_src[i] = (uint32_t)(row * 256 + (((lane & 15) ^ ksw) * 16));
    const int vrow = pc * 8 + (lane >> 3);
    v_src[i] = (uint32_t)((vrow * 64 + (((lane & 7) ^ ((vrow >> 1) & 7)) * 8)) * 2);
  }
  auto issue_k = [&](int t, int slot) {
#pragma unroll
    for (int i = 0; i < NPC; ++i)
      RF_BUF_LOAD_LDS(rsK, (lds_void*)(kring + slot * 16384 + (i * PSTR + (w % PSTR)) * 1024), k_src[i], (t0 + t) * (ATT_KV * 256));
  };
  auto issue_v = [&](int t, int slot) {
#pragma unroll
    for (int i = 0; i < NPC; ++i)
      RF_BUF_LOAD_LDS(rsV, (lds_void*)(vring + slot * 16384 + (i * PSTR + (w % PSTR)) * 1024), v_src[i], (t0 + t) * (128 * 64 * 2));
  };
  // counted waits: `n` issue calls (K or V^T tiles) may stay in flight = n * NPC instructions of THIS wave
  auto wait_barrier = [&](const int n) {
    if constexpr (NPC == 0) {
      __builtin_amdgcn_s_barrier();
    } else if constexpr (NPC == 2) {
      RF_ATT4_WAIT_BARRIER(2 * n);
    } else {
      if (n >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (n == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  // fragment read addresses: 4 (K, per d step) + 2 (V^T, per 32-key block) per-lane registers; ring slot, key block,
  // T tile (+8 rows) and d tile are immediates.  K row of MFMA row l15 in tile T0: (l15 & 7) | ((l15 & 8) << 1).
  const char* k_rd[4];
  const char* v_rd[2];
  const int krow = (l15 & 7) | ((l15 & 8) << 1);
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) k_rd[ds] = kring + krow * 256 + (((ds * 4 + g) ^ l15) << 4);
#pragma unroll
  for (int b = 0; b < 2; ++b) v_rd[b] = vring + l15 * 128 + (((b * 4 + g) ^ ((l15 >> 1) & 7)) << 4);
  // K fragment (32-key block b, tile T, d step ds): k_rd[ds] + slot*16384 + b*32*256 + T*8*256
  // V fragment (d tile dt, block b):               v_rd[b] + slot*16384 + dt*16*128

  f32x4 oacc[8][NQT];   // O^T tiles [d tile][q tile]: d = dt*16 + 4g + r, q = l15
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][qt][r] = 0.f;
  float l_run[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) l_run[qt] = 0.f;
  [[maybe_unused]] f32x4 lsum[NQT];    // SUMM: every row of this "O^T tile" is the row sum of query l15
  [[maybe_unused]] bf16x8 ones;
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) lsum[qt][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
  // score tiles, index ti = (b*2 + T)*NQT + qt: this lane holds keys b*32 + T*8 + {0-3 | 4-7 | 16-19 | 20-23}[g] of query l15
  f32x4 s_cur[4 * NQT], s_nxt[4 * NQT];
  bf16x8 pf[2 * NQT];   // P(t-1): B-operand fragments [b*NQT + qt] of the pending PV product: words T*2, T*2+1 from tile (b, T, qt)

  {  // V ring slot 3 stands in for V(-1): tile 0 multiplies it with P(-1) = 0, so it must be finite
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4*)(vring + 3 * 16384 + (i * 8 + w) * 1024 + lane * 16) = z;
#pragma unroll
    for (int i = 0; i < 2 * NQT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[i][j] = (bf16_t)0.f;
  }
  issue_k(0, 0);
  if (nt > 1) issue_k(1, 1);
  if (nt > 2) issue_k(2, 2);
  issue_v(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill above
  wait_barrier((nt > 1) + (nt > 2) + 1);   // K0 landed
#pragma unroll
  for (int bt = 0; bt < 4; ++bt) {   // (b, T) = (bt >> 1, bt & 1)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_cur[bt * NQT + qt][r] = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const bf16x8 kf = *(const bf16x8*)(k_rd[ds] + (bt >> 1) * 32 * 256 + (bt & 1) * 8 * 256);
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) s_cur[bt * NQT + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], s_cur[bt * NQT + qt], 0, 0, 0);
    }
  }
  // -m of the two q-tiles, as the C operand of the score MFMAs (LAG); a query's four lanes (l15 + 16 g) hold the same value
  f32x4 negm[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) negm[qt][r] = 0.f;
  if constexpr (LAG) {
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      float mx = s_cur[qt][0];
#pragma unroll
      for (int bt = 0; bt < 4; ++bt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s_cur[bt * NQT + qt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
      for (int r = 0; r < 4; ++r) negm[qt][r] = -mx;
    }
    // S(0) - m exactly as every later tile forms it: the chain starts from -m in the C operand (subtracting afterwards rounds
    // differently, and a row must not depend on which wave of which launch shape computes it)
    if constexpr (!ROT) {
#pragma unroll
      for (int bt = 0; bt < 4; ++bt) {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) s_cur[bt * NQT + qt] = negm[qt];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          const bf16x8 kf = *(const bf16x8*)(k_rd[ds] + (bt >> 1) * 32 * 256 + (bt & 1) * 8 * 256);
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) s_cur[bt * NQT + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], s_cur[bt * NQT + qt], 0, 0, 0);
        }
      }
    }
  }

  if constexpr (ROT) {   // the rotated order computes S(0) itself, as G(-1) of interval 0, which also packs P(-1) = 0 from here
#pragma unroll
    for (int i = 0; i < 4 * NQT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_cur[i][r] = 0.f;
  }

  auto tile = [&](const int t, auto ts_tag) {
    constexpr int TS = decltype(ts_tag)::value;
    // between two barriers a wave runs F(t) = [pending PV(t-1) | P(t) = exp2(S(t)) + row sums] and G = [scores of the next tile |
    // pack P into the PV operand].  ROT = false: F(t), G(t) -- G reads K(t+1) in ring slot (TS+1) % 4.  ROT = true: G(t-1), F(t) --
    // G reads K(t) in slot TS.  Both read V(t-1) in slot (TS+3) % 4 and both leave the rings alone until the next barrier.
    constexpr int KSLOT = ROT ? TS : (TS + 1) % 4, VSLOT = (TS + 3) % 4;
    // needed now: K(t+1) [next scores], V(t-1) [pending PV]; may stay in flight: K(t+2), V(t)
    wait_barrier((t + 2 < nt) + 1);
    if (t + 3 < nt) issue_k(t + 3, (TS + 3) % 4);
    if (t + 1 < nt) issue_v(t + 1, (TS + 1) % 4);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int AHEAD = (VAR & 36) == 36 ? 4 : (VAR & 32) ? 3 : (VAR & 4) ? 1 : 2;   // fragment reads run AHEAD groups in front of their MFMAs (2: -2.3 % vs 1, profiles/r03_kb_attn_mix_v1.log)
    constexpr int NFR = AHEAD + 1;
    // s_setprio schemes, PRIO = (VAR >> 7) & 7.  The SIMD's arbiter favours the OLDER wave of a pair whenever both
    // have an MFMA ready (tools/ubench/attn_group.py: wave 0 runs at its solo speed, its partner gets the gaps); the schemes shift
    // that: 1: F = 2, G = 0   2: F = 0, G = 2   3: plain waves F = 0, G = 2, rotated waves G = 2, F = 1   4: rotated waves 1, plain 0
    // 5: alternating per group, opposite phase in the rotated waves   6: plain F = 1, G = 2, rotated G = 2, F = 0
    auto prio_F = [&]() {
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(2);
      if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(ROT ? 1 : 0);
      if constexpr (PRIO == 6) __builtin_amdgcn_s_setprio(ROT ? 0 : 1);
    };
    auto prio_G = [&]() {
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if constexpr (PRIO == 2 || PRIO == 3 || PRIO == 6) __builtin_amdgcn_s_setprio(2);
    };
    bf16x8 fr[NFR][2];   // fragment ring: group G multiplies fr[G % NFR][0..1]
    float psum[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) psum[qt] = 0.f;
    // The 16 groups of an interval sit at positions P = 0 .. 15 (ROT = false: P = G; ROT = true: the G groups 8 .. 15 first).
    // fragment pair of group G, held in fr[P % NFR]: G < 8 -> V^T(t-1) [d tile G, key blocks 0, 1], else K [(b, T) = (G-8)/2,
    // d steps ((G-8)%2)*2 + 0, 1]
    auto load_pos = [&](auto ptag) {
      constexpr int P = decltype(ptag)::value;
      constexpr int G = ROT ? (P + 8) % 16 : P;
      if constexpr (P >= 16) {
      } else if constexpr (G < 8) {
#pragma unroll
        for (int e = 0; e < 2; ++e) fr[P % NFR][e] = *(const bf16x8*)(v_rd[e] + VSLOT * 16384 + G * 16 * 128);
      } else {
        constexpr int bt = (G - 8) / 2;
#pragma unroll
        for (int e = 0; e < 2; ++e)
          fr[P % NFR][e] = *(const bf16x8*)(k_rd[((G - 8) % 2) * 2 + e] + KSLOT * 16384 + (bt >> 1) * 32 * 256 + (bt & 1) * 8 * 256);
      }
    };
    load_pos(std::integral_constant<int, 0>{});
    if constexpr (AHEAD >= 2) load_pos(std::integral_constant<int, 1>{});
    if constexpr (AHEAD >= 3) load_pos(std::integral_constant<int, 2>{});
    if constexpr (AHEAD >= 4) load_pos(std::integral_constant<int, 3>{});
    // ---- F: pending PV product (8 groups of 4 MFMA: d tile G) | P = exp2(S) in place + row sums (tile G) ----
    auto half_F = [&]() {
    prio_F();
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto gtag) {
      constexpr int G = decltype(gtag)::value;
      constexpr int P = ROT ? G + 8 : G;
      if constexpr (PRIO == 5) __builtin_amdgcn_s_setprio(((G & 1) != 0) == ROT ? 2 : 0);
      load_pos(std::integral_constant<int, P + AHEAD>{});
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
          oacc[G][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[P % NFR][b], pf[b * NQT + qt], oacc[G][qt], 0, 0, 0);
      if constexpr (SUMM && (G == 0 || G == 4)) {   // the ones "d tile": key block b = G / 4 of P(t-1)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) lsum[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[(G / 4) * NQT + qt], lsum[qt], 0, 0, 0);
      }
      {
        if constexpr (NQT == 2) {   // score tile G = (b*2 + T)*2 + qt
          float e4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) e4[j] = __builtin_amdgcn_exp2f(s_cur[G][j]);
          RF_PIN4(e4[0], e4[1], e4[2], e4[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s_cur[G][j] = e4[j];
            if constexpr (!SUMM) psum[G & 1] += e4[j];
          }
        } else {                    // four score tiles over eight groups: half a tile each
          float e0 = __builtin_amdgcn_exp2f(s_cur[G >> 1][(G & 1) * 2]), e1 = __builtin_amdgcn_exp2f(s_cur[G >> 1][(G & 1) * 2 + 1]);
          asm volatile("" : "+v"(e0), "+v"(e1));
          s_cur[G >> 1][(G & 1) * 2] = e0;
          s_cur[G >> 1][(G & 1) * 2 + 1] = e1;
          if constexpr (!SUMM) {
            psum[0] += e0;   // (the summation order of the two-q-tile form: bit-identical row sums)
            psum[0] += e1;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) asm volatile("" : "+v"(psum[qt]));
    if constexpr (LAG) {
      // a P above ~2^30 shows in its lane's row sum (inf included; the negated compare also catches NaN)
      bool hot = false;
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) hot = hot || !(psum[qt] <= p.lag_thresh);
      if (__builtin_expect(__any(hot), 0)) {
        // ---- re-centre (rare, wave-uniform, no barrier): K(t) is still in ring slot TS --------------------------------
        float mx[NQT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) mx[qt] = -__builtin_huge_valf();
        auto raw_tile = [&](const int bt, f32x4 (&a)[NQT]) {   // a += K(t)[(b, T) = bt] Q^T for the wave's q-tiles
#pragma unroll
          for (int ds = 0; ds < 4; ++ds) {
            const bf16x8 kf = *(const bf16x8*)(k_rd[ds] + TS * 16384 + (bt >> 1) * 32 * 256 + (bt & 1) * 8 * 256);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], a[qt], 0, 0, 0);
          }
        };
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
          f32x4 a[NQT];
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[qt][r] = 0.f;
          raw_tile(bt, a);
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[qt] = fmaxf(mx[qt], a[qt][r]);
        }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          float m = mx[qt];
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          const float m_old = -negm[qt][0];
          const float m_new = fmaxf(m_old, m);
          const float f = __builtin_amdgcn_exp2f(m_old - m_new);   // <= 1; exactly 1 for the q-tile that did not trip
#pragma unroll
          for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][qt][r] *= f;
          l_run[qt] *= f;
#pragma unroll
          for (int r = 0; r < 4; ++r) negm[qt][r] = -m_new;
          psum[qt] = 0.f;
        }
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
          f32x4 a[NQT];
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) a[qt] = negm[qt];
          raw_tile(bt, a);
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __builtin_amdgcn_exp2f(a[qt][r]);
              s_cur[bt * NQT + qt][r] = e;
              psum[qt] += e;
            }
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) l_run[qt] += psum[qt];
    };   // half_F
    // ---- G: the next tile's scores (8 groups of 4 MFMA) | pack P tile G-8 into the PV operand ----------------
    auto half_G = [&]() {
    prio_G();
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto gtag) {
      constexpr int G = decltype(gtag)::value + 8;
      constexpr int P = ROT ? G - 8 : G;
      if constexpr (PRIO == 5) __builtin_amdgcn_s_setprio(((G & 1) != 0) == ROT ? 2 : 0);
      load_pos(std::integral_constant<int, P + AHEAD>{});
      constexpr int bt = (G - 8) / 2;
      constexpr int dsb = ((G - 8) % 2) * 2;
      if constexpr (dsb == 0 && !LAG) {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s_nxt[bt * NQT + qt][r] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          // LAG: the chain of a score tile starts from -m (C operand = the persistent negm registers, D = the tile)
          if constexpr (LAG && dsb == 0) {
            if (e == 0) {
              s_nxt[bt * NQT + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[P % NFR][e], qf[qt][dsb + e], negm[qt], 0, 0, 0);
              continue;
            }
          }
          s_nxt[bt * NQT + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[P % NFR][e], qf[qt][dsb + e], s_nxt[bt * NQT + qt], 0, 0, 0);
        }
      if constexpr (NQT == 2 || ((G - 8) & 1) == 0) {
        // NQT == 2: score tile ti = G - 8 = (b*2 + T)*2 + qt; NQT == 1: tile (G - 8) / 2 = b*2 + T on the even groups
        constexpr int ti = NQT == 2 ? G - 8 : (G - 8) / 2;
        constexpr int bT = ti / NQT, qt_ = ti % NQT;
        uint32_t w0 = pack2(s_cur[ti][0], s_cur[ti][1]);
        uint32_t w1 = pack2(s_cur[ti][2], s_cur[ti][3]);
        asm volatile("" : "+v"(w0), "+v"(w1));
        constexpr int pi = (bT >> 1) * NQT + qt_, wi = (bT & 1) * 2;
        u32x4 t4 = __builtin_bit_cast(u32x4, pf[pi]);
        t4[wi] = w0;
        t4[wi + 1] = w1;
        pf[pi] = __builtin_bit_cast(bf16x8, t4);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int i = 0; i < 4 * NQT; ++i) s_cur[i] = s_nxt[i];
    };   // half_G
    if constexpr (ROT) {
      half_G();
      half_F();
    } else {
      half_F();
      half_G();
    }
  };
  for (int t = 0; t < nt; t += 4) {   // unrolled by the ring size (dispatch guarantees nt % 4 == 0): one exit
    tile(t, std::integral_constant<int, 0>{});
    tile(t + 1, std::integral_constant<int, 1>{});
    tile(t + 2, std::integral_constant<int, 2>{});
    tile(t + 3, std::integral_constant<int, 3>{});
  }
  if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(0);

  if (PROBE && p.probe) clk.end(g_attn_clk_probe);
  // ---- epilogue: the last pending product O^T += V(nt-1)^T P(nt-1)^T ------------------------------------------------
  wait_barrier(0);
  if constexpr (ROT) {   // the rotated order has not packed P(nt-1) yet
#pragma unroll
    for (int ti = 0; ti < 4 * NQT; ++ti) {
      const int bT = ti / NQT, qt_ = ti % NQT;
      u32x4 t4 = __builtin_bit_cast(u32x4, pf[(bT >> 1) * NQT + qt_]);
      t4[(bT & 1) * 2] = pack2(s_cur[ti][0], s_cur[ti][1]);
      t4[(bT & 1) * 2 + 1] = pack2(s_cur[ti][2], s_cur[ti][3]);
      pf[(bT >> 1) * NQT + qt_] = __builtin_bit_cast(bf16x8, t4);
    }
  }
  {
    const int vslot = (nt - 1) % ATT4_RING;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bf16x8 vf = *(const bf16x8*)(v_rd[b] + vslot * 16384 + dt * 16 * 128);
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[b * NQT + qt], oacc[dt][qt], 0, 0, 0);
      }
    if constexpr (SUMM) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) lsum[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[b * NQT + qt], lsum[qt], 0, 0, 0);
      // every lane of a query holds the WHOLE row sum; attn5_finish / the combine kernel add a query's four lanes: hand them a quarter each
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) l_run[qt] = lsum[qt][0] * 0.25f;
    }
  }
  if constexpr (NQT == 2)   // (the mixed-size launch never splits the key axis)
  if (partial != nullptr) {
    float* dst = partial + tid * 4;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) *(f32x4*)(dst + (dt * 2 + qt) * 2048) = oacc[dt][qt];
    partial[16 * 2048 + tid] = l_run[0];
    partial[16 * 2048 + 512 + tid] = l_run[1];
    partial[16 * 2048 + 1024 + tid] = -negm[0][0];   // the piece's lagged row maxima (0 without LAG)
    partial[16 * 2048 + 1536 + tid] = -negm[1][0];
    return;
  }
  float m_row[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) m_row[qt] = -negm[qt][0];
  attn5_finish<NQT, (VAR & 8) != 0>(p, oacc, l_run, lane, head, q0w, m_row);
}

// The variant the product kernels run (the VAR bits of attn5_body): 256 = wave priority scheme 2 -- s_setprio 2 in G (scores + packing),
// 0 in F (PV + exp2) -- worth 1.8 % on top of the DMA distribution (208.3 -> 204.6 us at S = 4608, profiles/r03_kb_attn_stamps_v7.log);
// 2048 = the row sums on the matrix pipe (bounded form only; 1.3-2 % in seven interleaved pairs, profiles/r03_kb_attn_stamps_v10.log).
constexpr int ATT5_VAR = 256 | 2048;

template <bool LAG>
__global__ __launch_bounds__(512) void attn_fwd_kernel_v5(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ClkProbe clk;
  if (p.probe) clk.begin();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves w and w + 4 share a SIMD: the second one runs the halves of an interval in the rotated order, so that one of the pair is in
  // its exp2-heavy half while the other is in its MFMA-only half (see attn5_body)
  const int head = blockIdx.x % p.heads, q0w = (int)(blockIdx.x / p.heads) * 256 + w * 32;
  if (w < 4) attn5_body<true, LAG, ATT5_VAR, 2, false, 2>(p, smem, tid, lane, w, head, q0w, 0, p.S / ATT_KV, nullptr, clk);
  else attn5_body<false, LAG, ATT5_VAR, 2, true, 0>(p, smem, tid, lane, w, head, q0w, 0, p.S / ATT_KV, nullptr, clk);
}


// Mixed-size launch.  heads x S / 256 workgroups of 256 queries rarely come out as whole rounds of the 256 CUs (S = 4608: 432 =
// 1.69 rounds run as 2; the split launch above fixes that at the price of ~70 MB of partial (O, l) through HBM and a second
// launch: break-even at 84 % fill).  This launch changes the SIZE of some workgroups instead: `n_big` workgroups of 256 queries
// (dispatched first) and the rest of 192 -- waves 0-3 carry two q-tiles as everywhere, waves 4-7 ONE (waves w and w + 4 share a
// SIMD, so every SIMD does 3/4 of the MFMAs of a full workgroup; K / V^T tiles, DMA pieces and barriers are those of the full
// one).  S = 4608: per head 9 x 256 + 12 x 192 queries -> 216 + 288 workgroups; a CU runs a big and a small one or two small
// ones: 16 + 12 = 28 q-tile units instead of 2 x 16 = 32.  Nothing is split along the key axis: no scratch, no second launch,
// and every output row is computed by exactly the code of the plain launch (bit-identical results).
struct AttnMixParams {
  int n_big;          // workgroups of 256 queries = heads * big_per_head
  int big_per_head;
};

template <bool LAG, int MIX_SMALL_DMA_A = 0>   // 192-query workgroups: 0 = waves 4-7 issue the DMA pieces (shipped), 2 = waves 0-3 (experiments)
__global__ __launch_bounds__(512) void attn_fwd_kernel_v5mix(const AttnParams p, const AttnMixParams mx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ClkProbe clk;
  if (p.probe) clk.begin();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  // (n_big % 8 == 0 when heads % 8 == 0: workgroup b of either kind runs on XCD b % 8 = head % 8, as in the plain launch)
  const bool big = b < mx.n_big;
  const int b2 = big ? b : b - mx.n_big;
  const int head = b2 % p.heads;
  const int q0 = big ? (b2 / p.heads) * 256 : mx.big_per_head * 256 + (b2 / p.heads) * 192;
  // DMA pieces: in a 256-query workgroup waves 0-3 (which wait for their partners) issue them all; in a 192-query one the waves with ONE
  // q-tile (4-7) are the early ones and take them
  if (big) {
    if (w < 4) attn5_body<true, LAG, ATT5_VAR, 2, false, 2>(p, smem, tid, lane, w, head, q0 + w * 32, 0, p.S / ATT_KV, nullptr, clk);
    else attn5_body<false, LAG, ATT5_VAR, 2, true, 0>(p, smem, tid, lane, w, head, q0 + w * 32, 0, p.S / ATT_KV, nullptr, clk);
  } else {
    if (w < 4) attn5_body<true, LAG, ATT5_VAR, 2, false, MIX_SMALL_DMA_A>(p, smem, tid, lane, w, head, q0 + w * 32, 0, p.S / ATT_KV, nullptr, clk);
    else attn5_body<false, LAG, ATT5_VAR, 1, true, 2 - MIX_SMALL_DMA_A>(p, smem, tid, lane, w, head, q0 + 128 + (w - 4) * 16, 0, p.S / ATT_KV, nullptr, clk);
  }
}


// Sizes of the mixed launch for `units` = S / 16 q-tiles per head: units = 16 a + 12 b.  Greedy in-order dispatch of the a * heads
// big workgroups, then the b * heads small ones, onto P CUs, with a small workgroup costing `small_cost` of a big one (measured
// 0.75-0.8); returns the makespan in big-workgroup units and the best (a, b), or a = units / 16, b = 0 if nothing beats the
// plain launch.
constexpr float ATT5_SMALL_COST = 0.9f;   // a 192-query workgroup's time / a 256-query one's: the MFMAs are 3/4, but a tile takes as long as
                                          // its slowest wave (S = 4608: 216 -> 206 us = 1.9 instead of 2 rounds; profiles/r03_kb_attn_mix_v1.log)
static float attn_mix_plan(const int units, const int heads, const int P, const float small_cost, int* a_out, int* b_out) {
  float best = 1e30f;
  int best_a = units / 16, best_b = 0;
  std::vector<float> cu((size_t)P);
  for (int b = 0; 12 * b <= units; b += 4) {
    if ((units - 12 * b) % 16 != 0) continue;
    const int a = (units - 12 * b) / 16;
    // in-order dispatch = always to the CU that frees first: a min-heap over P finish times
    std::fill(cu.begin(), cu.end(), 0.f);
    auto cmp = [](float x, float y) { return x > y; };
    std::make_heap(cu.begin(), cu.end(), cmp);
    auto push = [&](const int n, const float cost) {
      for (int i = 0; i < n; ++i) {
        std::pop_heap(cu.begin(), cu.end(), cmp);
        cu.back() += cost;
        std::push_heap(cu.begin(), cu.end(), cmp);
      }
    };
    push(a * heads, 1.f);
    push(b * heads, small_cost);
    const float span = *std::max_element(cu.begin(), cu.end());
    if (span < best - 1e-4f) { best = span; best_a = a; best_b = b; }
  }
  *a_out = best_a;
  *b_out = best_b;
  return best;
}
// the plan of the last (S, heads, P) of this thread (the simulation costs ~50 us of host time: not per launch)
static float attn_mix_plan_cached(const int units, const int heads, const int P, int* a_out, int* b_out) {
  thread_local int key[3] = {0, 0, 0}, ab[2] = {0, 0};
  thread_local float span = 0.f;
  if (key[0] != units || key[1] != heads || key[2] != P) {
    span = attn_mix_plan(units, heads, P, ATT5_SMALL_COST, &ab[0], &ab[1]);
    key[0] = units; key[1] = heads; key[2] = P;
  }
  *a_out = ab[0];
  *b_out = ab[1];
  return span;
}


// Split launch (rf_attention_fwd_ws with scratch): 432 workgroups at S = 4608 are 1.69 rounds of 256 CUs run as 2, 528 at
// S = 5632 are 2.06 run as 3.  One persistent workgroup per CU takes an equal share of the (block, 4-tile quad) space
// instead -- at most one piece at the head and one at the tail of its range are partial blocks, which go to its two
// scratch slots.  XCD x (workgroups x, x + 8, ...) keeps the heads h % 8 == x, as the one-block-per-workgroup launch does
// (its block b runs on XCD b % 8 = head % 8 when heads % 8 == 0): a head's K / VT stay in one L2.
struct AttnSkParams {
  int nq;        // quads (4 key tiles) per block
  int nqb;       // 256-query blocks
  int hpx;       // heads per XCD (heads / 8)
  int wpx;       // workgroups per XCD
  float* ws;     // [workgroup][2] slots of ATT5_SLOT floats
};
constexpr int ATT5_SLOT = 16 * 2048 + 2048;   // O^T quads, l [2][512], m [2][512]

__device__ __forceinline__ int att5_start(const AttnSkParams& sk, const int wl) {   // first quad of XCD-local workgroup wl
  return (int)((int64_t)wl * (sk.hpx * sk.nqb * sk.nq) / sk.wpx);
}

template <bool LAG>
__global__ __launch_bounds__(512) void attn_fwd_kernel_v5sk(const AttnParams p, const AttnSkParams sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ClkProbe clk;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int start = att5_start(sk, wl), end = att5_start(sk, wl + 1);
  bool first = true;
  for (int cur = start; cur < end;) {
    const int lu = cur / sk.nq, q0 = cur - lu * sk.nq;
    const int len = min(sk.nq - q0, end - cur);
    const int head = x + 8 * (lu / sk.nqb), qb = lu % sk.nqb;
    float* partial = (q0 == 0 && len == sk.nq) ? nullptr : sk.ws + (int64_t)(2 * blockIdx.x + (cur == start ? 0 : 1)) * ATT5_SLOT;
    if (!first) __builtin_amdgcn_s_barrier();   // every wave is done with the previous piece's rings
    first = false;
    // launder the thread id once per piece: keeps the per-lane address math inside the piece (LICM would hoist it
    // across the loop and spill)
    int tid_i = tid;
    asm volatile("" : "+v"(tid_i));
    if (w < 4) attn5_body<false, LAG, ATT5_VAR, 2, false, 2>(p, smem, tid_i, tid_i & 63, w, head, qb * 256 + w * 32, 4 * q0, 4 * len, partial, clk);
    else attn5_body<false, LAG, ATT5_VAR, 2, true, 0>(p, smem, tid_i, tid_i & 63, w, head, qb * 256 + w * 32, 4 * q0, 4 * len, partial, clk);
    cur += len;
  }
}

// one workgroup per (head, query block): nothing to do if one piece covered it, else add its pieces' slots and finish
__global__ __launch_bounds__(512) void attn5_combine_kernel(const AttnParams p, const AttnSkParams sk) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int head = blockIdx.x % p.heads, qb = blockIdx.x / p.heads;
  const int x = head & 7, lu = (head >> 3) * sk.nqb + qb;
  const int u0 = lu * sk.nq, u1 = u0 + sk.nq;
  // the XCD-local workgroup whose range holds quad u0
  int wl = (int)(((int64_t)u0 * sk.wpx) / (sk.hpx * sk.nqb * sk.nq));
  while (wl > 0 && att5_start(sk, wl) > u0) --wl;
  while (att5_start(sk, wl + 1) <= u0) ++wl;
  if (att5_start(sk, wl + 1) >= u1) return;   // one workgroup held the whole block: it stored the rows itself
  f32x4 oacc[8][2];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][qt][r] = 0.f;
  float l_run[2] = {0.f, 0.f};
  auto slot_of = [&](const int wl2) {
    const int s0 = att5_start(sk, wl2);
    const int piece_begin = s0 > u0 ? s0 : u0;
    return sk.ws + (int64_t)(2 * (wl2 * 8 + x) + (piece_begin == s0 ? 0 : 1)) * ATT5_SLOT;
  };
  // the pieces' lagged maxima (all 0 for the bounded kernel): scale every piece to the largest
  float mmax[2] = {-__builtin_huge_valf(), -__builtin_huge_valf()};
  for (int wl2 = wl; att5_start(sk, wl2) < u1; ++wl2) {
    const float* src = slot_of(wl2);
    mmax[0] = fmaxf(mmax[0], src[16 * 2048 + 1024 + tid]);
    mmax[1] = fmaxf(mmax[1], src[16 * 2048 + 1536 + tid]);
  }
  for (; att5_start(sk, wl) < u1; ++wl) {
    const float* src = slot_of(wl);
    const float f[2] = {__builtin_amdgcn_exp2f(src[16 * 2048 + 1024 + tid] - mmax[0]), __builtin_amdgcn_exp2f(src[16 * 2048 + 1536 + tid] - mmax[1])};
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] += *(const f32x4*)(src + (dt * 2 + qt) * 2048 + tid * 4) * f[qt];
    l_run[0] += src[16 * 2048 + tid] * f[0];
    l_run[1] += src[16 * 2048 + 512 + tid] * f[1];
  }
  attn5_finish<2>(p, oacc, l_run, lane, head, qb * 256 + w * 32, mmax);
}

int read_clk_probe_attn(unsigned long long* h) {
  return hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attn_clk_probe), 4 * sizeof(unsigned long long)) == hipSuccess ? RF_OK : RF_ERR_HIP;
}

// Kernel-selection knobs: compile-time constants in librf_flux.so; mutable (rf_debug_*) in the experiments build only.
#define RF_ATT_V7_DEFAULT 0
struct AttnTuning {
  int v2;      // -1 = cost model between v1 and v2, 0 / 1 = forced
  int v4;      // 1 = the shift-free kernels may run, 0 = never
  int v5;      // MFMA shape of the bounded kernel: 1 = 16x16x32 (v5; shipped at every length), 0 = 32x32x16 (v4), -1 = by size: v5 below
               // 8192 keys.  Round 2 shipped -1: v5 ran the chip at 2.0 instead of 1.75 GHz and at S = 17920 that cost the neighbouring
               // GEMMs more than it won (189.5 + 186.6 vs 187.6 + 183.5 ms per forward).  With the DMA pieces issued by the waves
               // that wait (round 3) both kernels got faster, v5 more: in sequence at S = 17920 attention 169.3 vs 175.9 ms and
               // GEMMs 181.5 vs 179.3 ms per forward (profiles/r03_ab_attn_cfg5.log): v5 is 1.2 % ahead overall, 53.2 % of peak.
  int v6;      // experiments: one wave per SIMD form
  int knock;   // experiments: timing knock-outs
  int sk;      // split launch: -1 = heuristic, 0 = never, 1 = whenever possible
  int lag;     // -1 = lagged-max kernel only without a usable bound, 0 = never, 1 = always
  int mix;     // mixed-size launch: -1 = when the dispatch simulation predicts >= 4 %, 0 = never
  int v7;      // 1 = the ping-pong schedule (attn7_body) for the whole-key-axis launches of the 16x16x32 kernels, 0 = v5's
};
static constexpr AttnTuning g_at = {-1, 1, 1, 0, 0, -1, -1, -1, RF_ATT_V7_DEFAULT};
static int g_last_attn_path = 0;

}  // namespace rf

// read-only: the sizes the mixed-size launch would use for S keys x heads on num_cus CUs (pure host arithmetic: CPU tests check its
// invariants without a GPU).  out = {256-query workgroups per head, 192-query workgroups per head, 1000 x simulated makespan in units of
// a 256-query workgroup, plain-grid rounds}
extern "C" int rf_debug_attn_mix_plan(int32_t S, int32_t heads, int32_t num_cus, int32_t* out) {
  RF_REQUIRE(out != nullptr && S > 0 && S % 256 == 0 && heads > 0 && num_cus >= 8, RF_ERR_SHAPE, "rf_debug_attn_mix_plan: bad arguments");
  int a = 0, b = 0;
  const float span = rf::attn_mix_plan(S / 16, heads, num_cus / 8 * 8, rf::ATT5_SMALL_COST, &a, &b);
  out[0] = a; out[1] = b; out[2] = (int32_t)(span * 1000.f + 0.5f); out[3] = rf::cdiv(heads * (S / 256), num_cus / 8 * 8);
  return RF_OK;
}

// read-only introspection: 1 / 2 / 4 / 5 = kernel version, 6 = v5 split launch, 8 = v5 lagged-max, 9 = its split launch, 10 / 11 = the
// mixed-size launch of the bounded / lagged-max kernel (7 = v6, experiments)
extern "C" int rf_debug_last_attn_path(void) { return rf::g_last_attn_path; }


extern "C" int64_t rf_attention_ws_bytes(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  return (int64_t)2 * (cus / 8 * 8) * rf::ATT5_SLOT * 4;
}

extern "C" int rf_attention_fwd(const void* q, const void* k, const void* vt, void* out, int32_t heads,
                                int32_t S, int32_t s_pad, int64_t ldo, int32_t n_main, int32_t mode,
                                float cross_bias, float scale, int32_t q_prescaled, float score_bound, void* stream) {
  return rf_attention_fwd_ws(q, k, vt, out, heads, S, s_pad, ldo, n_main, mode, cross_bias, scale, q_prescaled, score_bound,
                             nullptr, 0, stream);
}

extern "C" int rf_attention_fwd_ws(const void* q, const void* k, const void* vt, void* out, int32_t heads,
                                   int32_t S, int32_t s_pad, int64_t ldo, int32_t n_main, int32_t mode,
                                   float cross_bias, float scale, int32_t q_prescaled, float score_bound,
                                   void* ws, int64_t ws_bytes, void* stream) {
  rf_attn_desc d;
  memset(&d, 0, sizeof(d));
  d.q = q; d.k = k; d.vt = vt; d.out = out; d.heads = heads; d.S = S; d.s_pad = s_pad; d.ldo = ldo; d.n_main = n_main;
  d.mode = mode; d.cross_bias = cross_bias; d.scale = scale; d.q_prescaled = q_prescaled; d.score_bound = score_bound;
  d.ws = ws; d.ws_bytes = ws_bytes; d.kernel = RF_ATTN_AUTO;
  return rf_attention(&d, stream);
}

extern "C" int rf_attention(const rf_attn_desc* d, void* stream) {
  using namespace rf;
  RF_REQUIRE(d != nullptr, RF_ERR_NULL, "rf_attention: desc is NULL");
  const int32_t heads = d->heads, S = d->S, s_pad = d->s_pad, mode = d->mode;
  int32_t n_main = d->n_main;
  const int64_t ldo = d->ldo;
  RF_REQUIRE(d->q && d->k && d->vt && d->out, RF_ERR_NULL, "rf_attention: NULL pointer");
  RF_REQUIRE(heads > 0 && S > 0 && s_pad >= S && s_pad % 64 == 0, RF_ERR_SHAPE,
             "rf_attention: bad shape heads=%d S=%d s_pad=%d", heads, S, s_pad);
  RF_REQUIRE(mode >= 0 && mode <= 2, RF_ERR_SHAPE, "rf_attention: mode=%d", mode);
  RF_REQUIRE(d->score_bound >= 0.f, RF_ERR_SHAPE, "rf_attention: score_bound=%g (0 = unknown)", (double)d->score_bound);
  RF_REQUIRE(d->lag_thresh >= 0.f, RF_ERR_SHAPE, "rf_attention: lag_thresh=%g (0 = default)", (double)d->lag_thresh);
  RF_REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->vt) && aligned16(d->out) && ldo % 4 == 0, RF_ERR_ALIGN,
             "rf_attention: operands must be 16-byte aligned");
  RF_REQUIRE(ldo >= (int64_t)heads * 128, RF_ERR_SHAPE, "rf_attention: ldo < heads*128");
  if (mode == 0) n_main = S;
  RF_REQUIRE(n_main >= 0 && n_main <= S, RF_ERR_SHAPE, "rf_attention: n_main=%d", n_main);
  static bool attr_set = false;
  if (!attr_set) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_STAGE));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_STAGE));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v4, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v5<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v5<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v5sk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v5sk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v5mix<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel_v5mix<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT4_LDS));
    attr_set = true;
  }
  AttnParams p;
  p.q = (const bf16_t*)d->q; p.k = (const bf16_t*)d->k; p.vt = (const bf16_t*)d->vt; p.out = (bf16_t*)d->out;
  p.heads = heads; p.S = S; p.s_pad = s_pad; p.n_main = n_main; p.mode = mode;
  p.nqb = cdiv(S, ATT_QBLK); p.ldo = ldo;
  p.cross_bias_l2 = d->cross_bias * 1.4426950408889634f;
  p.sl2 = d->scale * 1.4426950408889634f;
  p.lag_thresh = d->lag_thresh > 0.f ? d->lag_thresh : 1073741824.0f;   // 2^30
  p.probe = prof_open() ? 1 : 0;
  p.lse = d->lse;
  hipStream_t st = (hipStream_t)stream;
  const bool pre = d->q_prescaled != 0;
  ProfScope prof(RF_KC_ATTN, 4.0 * (double)S * (double)S * 128.0 * heads, st);

  // ---- which kernel -------------------------------------------------------------------------------------------------
  // shift-free kernels (no per-tile running maximum; 25-45 % faster than either online-softmax kernel wherever they apply,
  // profiles/r02_kb_attn*.log): need no mask / bias, whole rounds of the 4-slot rings and a prescaled q.  Bounded form
  // (P = exp2(s)): the caller's proven |s| <= score_bound <= 100.  Lagged-max form: nothing else.
  const bool fast_ok = mode == 0 && S % 256 == 0 && pre;
  const bool bound_ok = d->score_bound > 0.f && d->score_bound <= 100.f;
  int kern = d->kernel;
  void* ws = d->ws;
  const int64_t ws_bytes = d->ws_bytes;
  static int num_cus = 0;
  if (num_cus == 0) {
    int dev = 0;
    RF_CHECK_HIP(hipGetDevice(&dev));
    RF_CHECK_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int P = num_cus / 8 * 8;
  AttnSkParams sk;
  sk.nq = S / 256; sk.nqb = S / 256; sk.hpx = heads / 8; sk.wpx = P / 8; sk.ws = (float*)ws;
  const bool can_split = fast_ok && heads % 8 == 0 && P >= 8 && ws != nullptr && aligned16(ws) &&
                         ws_bytes >= (int64_t)2 * P * ATT5_SLOT * 4 && (int64_t)sk.hpx * sk.nqb * sk.nq >= 2 * sk.wpx;
  if (kern == RF_ATTN_AUTO) {
    bool use_v2 = g_at.v2 == 1;
    if (g_at.v2 < 0) {
      // v2 (256-query workgroups, 1 per CU) is ~8 % more efficient per row but has a coarser tail than v1
      // (128-query workgroups, 2 per CU); estimate both in units of "one CU doing 256 rows x S keys"
      const int nb1 = heads * cdiv(S, 128), nb2 = heads * cdiv(S, 256);
      const int rem1 = nb1 % 512;
      const float t1 = (float)(nb1 / 512) + (rem1 == 0 ? 0.f : (rem1 <= 256 ? 0.55f : 1.f));
      const float t2 = (float)cdiv(nb2, 256) / 1.08f;
      use_v2 = t2 < t1;
    }
    kern = use_v2 ? RF_ATTN_ONLINE256 : RF_ATTN_ONLINE128;
    const bool lag = g_at.lag == 1 || (g_at.lag < 0 && !bound_ok);
    if (fast_ok && g_at.v4 && g_at.v2 != 0 && (bound_ok || lag) && !(lag && g_at.lag == 0)) {
      const bool use5 = lag || g_at.v5 == 1 || (g_at.v5 < 0 && S < 8192);
      // split launch: one persistent workgroup per CU over equal shares of the (block, key quad) space when the plain grid
      // fills less than 80 % of its rounds.  Measured (profiles/r02_kb_attn_split.log): S = 5632 (528 blocks = 2.06 rounds
      // run as 3, 69 %) 399 -> 328 us; S = 4608 (1.69 as 2, 84 %) break-even -- the part is power-limited, idle CUs in the
      // last round let the busy ones clock higher; S = 17920 (6.56 as 7, 94 %) 8 % slower.
      const int blocks = heads * (S / 256), rounds = cdiv(blocks, P);
      // mixed-size launch first: no scratch, no second launch, bit-identical rows.  Taken when the simulated dispatch beats the
      // plain grid's rounds by >= 4 % (S = 4608: 1.9 vs 2 -> 216 -> 206 us; S = 5632: 2.7 vs 3 -> 392 -> 325 us, the split
      // launch's 321-325 us without its scratch traffic; S = 17920: 6.9 vs 7 measured 1-4 % SLOWER, not taken)
      bool mix = false;
      if (use5 && g_at.sk != 1 && g_at.mix != 0) {
        int a = 0, b = 0;
        const float span = attn_mix_plan_cached(S / 16, heads, P, &a, &b);
        mix = b > 0 && span <= 0.96f * (float)rounds;
      }
      const bool split = !mix && use5 && can_split && g_at.sk != 0 && (g_at.sk == 1 || (double)blocks / ((double)rounds * P) < 0.80);
      kern = !use5 ? RF_ATTN_BOUNDED32
             : mix ? (lag ? RF_ATTN_LAGGED16_MIX : RF_ATTN_BOUNDED16_MIX)
             : lag ? (split ? RF_ATTN_LAGGED16_SPLIT : RF_ATTN_LAGGED16)
                   : (split ? RF_ATTN_BOUNDED16_SPLIT : RF_ATTN_BOUNDED16);
    }
  } else {
    // an explicit request must be runnable as asked: fail loudly instead of substituting another kernel
    const bool want_fast = kern == RF_ATTN_BOUNDED32 || kern == RF_ATTN_BOUNDED16 || kern == RF_ATTN_BOUNDED16_SPLIT ||
                           kern == RF_ATTN_LAGGED16 || kern == RF_ATTN_LAGGED16_SPLIT || kern == RF_ATTN_BOUNDED16_MIX ||
                           kern == RF_ATTN_LAGGED16_MIX;
    RF_REQUIRE(kern == RF_ATTN_ONLINE128 || kern == RF_ATTN_ONLINE256 || want_fast, RF_ERR_SHAPE, "rf_attention: kernel=%d", kern);
    if (want_fast) RF_REQUIRE(fast_ok, RF_ERR_UNSUPPORTED, "rf_attention: kernel %d needs mode 0, S %% 256 == 0 and a prescaled q", kern);
    if (kern == RF_ATTN_BOUNDED32 || kern == RF_ATTN_BOUNDED16 || kern == RF_ATTN_BOUNDED16_SPLIT || kern == RF_ATTN_BOUNDED16_MIX)
      RF_REQUIRE(bound_ok, RF_ERR_UNSUPPORTED, "rf_attention: kernel %d needs 0 < score_bound <= 100 (got %g)", kern, (double)d->score_bound);
    if (kern == RF_ATTN_BOUNDED16_MIX || kern == RF_ATTN_LAGGED16_MIX)
      RF_REQUIRE(d->mix_small >= 0 && d->mix_small % 4 == 0 && 12 * d->mix_small <= S / 16, RF_ERR_SHAPE,
                 "rf_attention: mix_small=%d must be a multiple of 4 with 12 * mix_small <= S / 16", d->mix_small);
    if (kern == RF_ATTN_BOUNDED16_SPLIT || kern == RF_ATTN_LAGGED16_SPLIT)
      RF_REQUIRE(can_split, RF_ERR_WORKSPACE, "rf_attention: the split launch needs heads %% 8 == 0 and %lld bytes of 16-byte aligned scratch",
                 (long long)((int64_t)2 * P * ATT5_SLOT * 4));
  }

  const dim3 grid2(heads * cdiv(S, 256)), blk(512);
  switch (kern) {
    case RF_ATTN_BOUNDED16_SPLIT:
      hipLaunchKernelGGL(attn_fwd_kernel_v5sk<false>, dim3(P), blk, ATT4_LDS, st, p, sk);
      hipLaunchKernelGGL(attn5_combine_kernel, grid2, blk, 0, st, p, sk);
      g_last_attn_path = 6;
      break;
    case RF_ATTN_LAGGED16_SPLIT:
      hipLaunchKernelGGL(attn_fwd_kernel_v5sk<true>, dim3(P), blk, ATT4_LDS, st, p, sk);
      hipLaunchKernelGGL(attn5_combine_kernel, grid2, blk, 0, st, p, sk);
      g_last_attn_path = 9;
      break;
    case RF_ATTN_LAGGED16:
      hipLaunchKernelGGL(attn_fwd_kernel_v5<true>, grid2, blk, ATT4_LDS, st, p);
      g_last_attn_path = 8;
      break;
    case RF_ATTN_BOUNDED16_MIX:
    case RF_ATTN_LAGGED16_MIX: {
      // (an explicit request runs the best mixed plan even if it is the plain grid: a = S / 256, b = 0)
      int a = 0, b = d->mix_small;
      if (b == 0) attn_mix_plan_cached(S / 16, heads, P, &a, &b);
      else a = (S / 16 - 12 * b) / 16;
      AttnMixParams mx;
      mx.n_big = a * heads;
      mx.big_per_head = a;
      const dim3 gridm((a + b) * heads);
      if (kern == RF_ATTN_LAGGED16_MIX) hipLaunchKernelGGL(attn_fwd_kernel_v5mix<true>, gridm, blk, ATT4_LDS, st, p, mx);
      else hipLaunchKernelGGL(attn_fwd_kernel_v5mix<false>, gridm, blk, ATT4_LDS, st, p, mx);
      g_last_attn_path = kern == RF_ATTN_LAGGED16_MIX ? 11 : 10;
      break;
    }
    case RF_ATTN_BOUNDED16:
      hipLaunchKernelGGL(attn_fwd_kernel_v5<false>, grid2, blk, ATT4_LDS, st, p);
      g_last_attn_path = 5;
      break;
    case RF_ATTN_BOUNDED32:
      hipLaunchKernelGGL(attn_fwd_kernel_v4, grid2, blk, ATT4_LDS, st, p);
      g_last_attn_path = 4;
      break;
    case RF_ATTN_ONLINE256: {
      const bool generic = !(mode == 0 && S % 64 == 0);
      g_last_attn_path = 2;
      if (generic) {
        if (pre) hipLaunchKernelGGL((attn_fwd_kernel_v2<true, true>), grid2, blk, ATT2_LDS, st, p);
        else hipLaunchKernelGGL((attn_fwd_kernel_v2<true, false>), grid2, blk, ATT2_LDS, st, p);
      } else {
        if (pre) hipLaunchKernelGGL((attn_fwd_kernel_v2<false, true>), grid2, blk, ATT2_LDS, st, p);
        else hipLaunchKernelGGL((attn_fwd_kernel_v2<false, false>), grid2, blk, ATT2_LDS, st, p);
      }
      break;
    }
    default: {
      g_last_attn_path = 1;
      const dim3 grid1(heads * p.nqb), blk1(256);
      if (pre) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid1, blk1, 2 * ATT_STAGE, st, p);
      else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid1, blk1, 2 * ATT_STAGE, st, p);
    }
  }
  RF_LAUNCH_CHECK();
  return RF_OK;
}
