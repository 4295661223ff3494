// bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   out[m,n] = epi( sum_k A[m,k] W[n,k] (+ sum_k2 A2[m,k2] W2[n,k2]) + bias[n] )
//
// Design (CDNA4-first, see DESIGN.md "K1"):
//   * block tile 256 x 256 x 64, 8 waves (4 x 2), each wave owns a 64 x 128 sub-tile of v_mfma_f32_16x16x32_bf16 tiles
//     (the fp8 kernels: v_mfma_scale_f32_16x16x128_f8f6f4); fp32 accumulators stay in registers for the whole K loop;
//   * operands are staged HBM -> LDS with the LDS-DMA path (buffer_load_dwordx4 ... lds, 16 B per
//     lane, no VGPR round trip).  The DMA writes LDS lane-linearly, so the bank-conflict-free
//     XOR swizzle is applied on the per-lane SOURCE address and mirrored on the ds_read_b128
//     side (chunk' = chunk ^ ((row >> 1) & 7) for 128-byte rows);
//   * ping-pong main loop: two wave groups half a phase apart, four phases per K-tile, the staging re-used as four
//     16 KiB half-tiles under counted vmcnt waits;
//   * up to 4 token groups (text / image / condition streams) with their own A, W, bias, gate
//     and output pointers ride in one launch, so the small text stream fills the tail of the
//     grid instead of costing its own under-filled launch;
//   * up to three K segments per group: segment 1/2 carry the second half of a concatenated
//     input (single-block proj_out = [attn | mlp]) and/or LoRA (A = x.lora_A^T,
//     W = scaling*lora_B) without materialising a concat or a merged weight;
//   * blockIdx is remapped so that every XCD works on a contiguous chunk of tiles (private L2).
//
// Map of this file (all of it ships in librf_flux.so): the epilogues (gemm_epilogue / gemm_epilogue_lds_v over the Acc32 / Acc16
// accumulator views), gemm_mainloop (plain loop of the 128 x 128 kernel for small problems and of split-K), the 256 x 256 loops
// gemm_mainloop_pp3_m16 (bf16: gemm_bf16_pp16e_kernel one tile per block, gemm_bf16_sk_kernel<..., EVEN> for the stream-K and
// persistent schedules) and gemm_mainloop_pp2_m16 (fp8 / mixed launches: gemm_w8_pp16_kernel and the W8 stream-K kernel),
// splitk_reduce_kernel, dispatch() and the C entry points.  The schedule of a launch is picked by dispatch() from the shape or
// by the CALLER per launch (rf_gemm_desc.schedule) -- there is no process-global kernel switch in this library.
// The loops of the A/B studies in profiles/r01..r03 (round-1 phases, 32x32x16 MFMA shapes, skinny-N, knock-outs, the s_memtime
// timeline) are not in this tree any more: git 6cfca97 holds the last tree that built them (csrc/experiments/).
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

namespace rf {

struct KSegDev {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  int nk;  // K-tiles (of 64) in this segment
  int _pad;
};

struct GemmGroupDev {
  KSegDev seg[3];
  int w8, _pad8;          // this token group's operands are fp8 e4m3 (rf_gemm_w8a8 group with a_scale); else bf16
  const float* a_scale;   // W8A8: per-row (token) dequantisation scale of this group's activations [M]
  const float* w_scale;   // W8A8: per-output-channel dequantisation scale of this group's weights [N]
  const bf16_t* bias;
  bf16_t* out; int64_t ldo;
  const bf16_t* residual; int64_t ldr;
  const bf16_t* gate;
  const bf16_t* norm_q; const bf16_t* norm_k;
  int M, tok_offset, tile_start, tiles_m;
};

struct GemmParams {
  int N, epi, ngroups, n_split, heads, s_pad, tiles_n, total_tiles;
  int w8;      // 1: at least one token group has fp8 e4m3 operands (1 byte / element; a K-tile is 128 elements = the same
               //    128 bytes): the launch uses the mixed-precision kernels, which pick the multiply per group
  int vec_ok;  // every output/residual/bias/gate pointer is 16-byte aligned and N % 8 == 0: LDS-staged epilogue
  bf16_t* q; bf16_t* k; bf16_t* vt;
  const float* rope_cos; const float* rope_sin; float norm_eps; float q_scale;
  // deterministic split-K (few-tile GEMMs, e.g. the LoRA down-projections): blockIdx.y = K-slice of kchunk
  // K-tiles, fp32 partial tiles go to ws[slice][m][n_pad], splitk_reduce_kernel sums them in slice order
  int ksplit, kchunk; float* ws; int64_t ws_slice; int ws_ld;
  float* scratch; int64_t scratch_bytes;  // host side: the caller's scratch as passed (flags + partials)
  int sched, _pad_s;                      // host side: rf_gemm_desc.schedule (rf_gemm_schedule)
  int probe, _pad_p;                      // block 0 stores its shader-clock probe (rf_gemm_desc.clock_probe, or while a profile is open)
  GemmGroupDev g[4];
};

__device__ unsigned long long g_clk_probe[4];   // see ClkProbe (common.hpp)
__device__ unsigned long long g_clk_probe_epi[2];   // {s_memtime, s_memrealtime} when block 0 / wave 0 has drained its epilogue stores

constexpr int PERSISTENT_ROUNDS = 0;   // > 0: bf16 launches with >= this many rounds run as ONE persistent launch (measured neutral: off)
constexpr int EPI_PARTIAL = 100;  // internal epilogue id: raw fp32 accumulators -> split-K scratch



// ---- shared epilogue ---------------------------------------------------------------------------
// acc[i][j]: 32x32 MFMA accumulators of one wave; fragment (i,j) covers rows wrow0 + i*32 .. and
// columns wcol0 + j*32 .. of the block tile at (m0, n0).
// accumulator layout (32x32 MFMA): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const GemmGroupDev& G, f32x16 (&acc)[FM][FN],
                                              const int m0, const int n0, const int wrow0, const int wcol0,
                                              const int l31, const int h) {
  const int M = G.M, N = p.N;
  int epi = p.epi;
  int ncol_base = 0;  // column offset subtracted for the GELU half of QKV_GELU
  if (epi == RF_EPI_QKV_GELU) {
    if (n0 >= p.n_split) {
      epi = RF_EPI_GELU;
      ncol_base = p.n_split;
    } else {
      epi = RF_EPI_QKV;
    }
  }
  const int DH = p.heads * 128;

#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wcol0 + j * 32 + l31;
    const bool nok = n < N;
    const float bias_v = (G.bias != nullptr && nok) ? bf2f(G.bias[n]) : 0.f;
    float gate_v = 0.f;
    if (epi == RF_EPI_GATE_RES && nok) gate_v = bf2f(G.gate[n]);
    // QKV destination decode for this column
    int which = 0, head = 0, d = 0;
    if (epi == RF_EPI_QKV && nok) {
      which = n / DH;
      const int rem = n - which * DH;
      head = rem >> 7;
      d = rem & 127;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int mrow0 = m0 + wrow0 + i * 32 + 4 * h;
      if (epi == RF_EPI_QKV) {
        if (!nok) continue;
        if (which < 2) {
          bf16_t* dst = (which == 0 ? p.q : p.k) + ((int64_t)head * p.s_pad) * 128 + d;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < M) dst[(int64_t)(G.tok_offset + m) * 128] = f2bf((acc[i][j][r] + bias_v) * (which == 0 ? p.q_scale : 1.0f));
          }
        } else {
          // V^T tiles: [head][tok/64][d][64], key position has bits 2,3 swapped
          bf16_t* dst = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64) + d * 64;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int m = mrow0 + 8 * rg;
            const int tok = G.tok_offset + m;
            if (((tok & 3) == 0) && (m + 3 < M)) {
              const int pos = (tok & 51) | ((tok & 4) << 1) | ((tok & 8) >> 1);  // within tile
              u32x2 v;
              v[0] = pack2(acc[i][j][rg * 4 + 0] + bias_v, acc[i][j][rg * 4 + 1] + bias_v);
              v[1] = pack2(acc[i][j][rg * 4 + 2] + bias_v, acc[i][j][rg * 4 + 3] + bias_v);
              *(u32x2*)(dst + (int64_t)(tok >> 6) * (128 * 64) + pos) = v;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int t2 = tok + e;
                if (m + e < M) {
                  const int pos = (t2 & 51) | ((t2 & 4) << 1) | ((t2 & 8) >> 1);
                  dst[(int64_t)(t2 >> 6) * (128 * 64) + pos] = f2bf(acc[i][j][rg * 4 + e] + bias_v);
                }
              }
            }
          }
        }
      } else {
        if (!nok) continue;
        bf16_t* orow = G.out + (n - ncol_base);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow0 + (r & 3) + 8 * (r >> 2);
          if (m < M) {
            float v = acc[i][j][r] + bias_v;
            if (epi == RF_EPI_GELU) {
              v = gelu_tanh(v);
            } else if (epi == RF_EPI_GATE_RES) {
              v = fmaf(gate_v, v, G.residual != nullptr ? bf2f(G.residual[(int64_t)m * G.ldr + n]) : 0.f);
            }
            orow[(int64_t)m * G.ldo] = f2bf(v);
          }
        }
      }
    }
  }
}

// ---- LDS-staged epilogue (the fast path) ------------------------------------------------------------
// The MFMA accumulator layout gives a lane ONE column and 16 scattered rows: stored directly that is
// 2-byte accesses, and the gated-residual epilogue becomes a chain of dependent 2-byte loads
// (measured: 670 vs 975 TF/s on 4608x3072x3072).  After the K loop the LDS is idle, so every wave
// transposes its accumulators through a private 16.5 KiB region, one 32-row fragment block at a time
// ([32 rows][128 cols] fp32, rows padded to 528 B -> conflict-free ds_write_b32 and ds_read_b128 with
// every offset an immediate off one base register), and then owns 8 CONSECUTIVE columns of a row: bias/gate are loaded once per lane as
// 16 bytes, residual rows are prefetched as 16-byte loads, results leave as 16-byte stores (q/k:
// 16 bytes of one head row).  V^T is written straight from registers (its natural layout already
// gives 8-byte runs along the key axis).
constexpr int EPI_ROW = 528;                 // padded fp32 row: 128 cols * 4 B + 16 B
constexpr int EPI_REGION = 32 * EPI_ROW;     // one wave's staging region

// Accumulator views: the epilogue touches a wave's accumulators in two ways only -- 4-token runs of one column (V^T,
// straight from registers) and "stage the 32-row block i into the wave's LDS region" -- so the two MFMA shapes differ
// only here.  32x32x16: acc[i][j] covers rows i*32.., cols j*32..; col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// 16x16x32: acc[it][jt] covers rows it*16.., cols jt*16..; col = lane&15, row = 4*(lane>>4) + r.
template <int FM>
struct Acc32 {
  f32x16 (&a)[FM][4];
  // f(row in the wave tile of the run's first token, column in the 128-column strip, float (&v)[4])
  template <class F>
  __device__ __forceinline__ void for_each_run4(const int lane, F f) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          float v[4] = {a[i][j][rg * 4 + 0], a[i][j][rg * 4 + 1], a[i][j][rg * 4 + 2], a[i][j][rg * 4 + 3]};
          f(i * 32 + 4 * h + 8 * rg, j * 32 + l31, v);
        }
  }
  __device__ __forceinline__ void stage(const int i, char* region, const int lane) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        *(float*)(region + row * EPI_ROW + (j * 32 + l31) * 4) = a[i][j][r];
      }
  }
};
template <int FM>
struct Acc16 {
  f32x4 (&a)[2 * FM][8];
  template <class F>
  __device__ __forceinline__ void for_each_run4(const int lane, F f) {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
      for (int it = 0; it < 2 * FM; ++it) {
        float v[4] = {a[it][jt][0], a[it][jt][1], a[it][jt][2], a[it][jt][3]};
        f(it * 16 + 4 * g, jt * 16 + l15, v);
      }
  }
  __device__ __forceinline__ void stage(const int i, char* region, const int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    // padded row = 132 dwords: lanes (g, l15) of one register hit banks 16g + l15 + const -- conflict-free
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *(float*)(region + (t * 16 + 4 * g + r) * EPI_ROW + (jt * 16 + l15) * 4) = a[2 * i + t][jt][r];
  }
};

template <int FM, bool W8, class ACC>
__device__ __forceinline__ void gemm_epilogue_lds_v(const GemmParams& p, const GemmGroupDev& G, ACC acc,
                                                    const int m0, const int n0, const int wrow0, const int wcol0,
                                                    const int lane, char* region) {
  const int M = G.M, N = p.N;
  int epi = p.epi;
  int ncol_base = 0;
  if (epi == RF_EPI_QKV_GELU) {
    if (n0 >= p.n_split) {
      epi = RF_EPI_GELU;
      ncol_base = p.n_split;
    } else {
      epi = RF_EPI_QKV;
    }
  }
  if (p.ksplit > 1) epi = EPI_PARTIAL;
  const int ncol0 = n0 + wcol0;  // first column of this wave's 128-column strip
  if (ncol0 >= N) return;
  int which = 0, head = 0;
  if (epi == RF_EPI_QKV) {
    const int DH = p.heads * 128;
    which = ncol0 / DH;
    head = (ncol0 - which * DH) >> 7;
  }

  if (epi == RF_EPI_QKV && which == 2) {
    // V^T tiles: [head][tok/64][d][64], key position has bits 2,3 swapped; a lane holds 4-key runs of one d
    acc.for_each_run4(lane, [&](const int row, const int col, float (&v)[4]) {
      const int n = ncol0 + col;
      const float bias_v = G.bias != nullptr ? bf2f(G.bias[n]) : 0.f;
      bf16_t* dst = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64) + col * 64;
      const int m = m0 + wrow0 + row;
      if constexpr (W8) {  // dequantise: acc * s_act[row] * s_w[col]
        const float swn = G.w_scale[n];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= swn * (m + e < M ? G.a_scale[m + e] : 0.f);
      }
      const int tok = G.tok_offset + m;
      if (((tok & 3) == 0) && (m + 3 < M)) {
        const int pos = (tok & 51) | ((tok & 4) << 1) | ((tok & 8) >> 1);
        u32x2 o;
        o[0] = pack2(v[0] + bias_v, v[1] + bias_v);
        o[1] = pack2(v[2] + bias_v, v[3] + bias_v);
        *(u32x2*)(dst + (int64_t)(tok >> 6) * (128 * 64) + pos) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t2 = tok + e;
          if (m + e < M) {
            const int pos = (t2 & 51) | ((t2 & 4) << 1) | ((t2 & 8) >> 1);
            dst[(int64_t)(t2 >> 6) * (128 * 64) + pos] = f2bf(v[e] + bias_v);
          }
        }
      }
    });
    return;
  }

  // this lane's 8 consecutive columns
  const int q8 = (lane & 15) * 8;
  const int n = ncol0 + q8;
  const bool nok = n < N;  // N % 8 == 0 on this path
  float bias8[8], gate8[8], sw8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f, gate8[e] = 0.f, sw8[e] = 1.f;
  constexpr bool w8 = W8;
  if (w8 && nok) {
    const f32x4 s0 = *(const f32x4*)(G.w_scale + n), s1 = *(const f32x4*)(G.w_scale + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) sw8[e] = s0[e], sw8[e + 4] = s1[e];
  }
  if (nok && G.bias != nullptr && epi != EPI_PARTIAL) unpack8(*(const u32x4*)(G.bias + n), bias8);
  if (nok && epi == RF_EPI_GATE_RES) unpack8(*(const u32x4*)(G.gate + n), gate8);
  const int rsub = lane >> 4;            // row within a 4-row read group
  const int c0 = (lane & 15) * 2;        // first of this lane's two 16-byte chunks
  // fused per-head RMSNorm + RoPE (block.py:38-41,60-67,74-78,92-99): the 16 lanes lane&~15 .. +15 hold
  // one complete 128-wide head row, 8 consecutive d each (rotation pairs stay inside a lane)
  const bool fuse_rope = (epi == RF_EPI_QKV) && (p.rope_cos != nullptr);
  float nw8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) nw8[e] = 1.f;
  if (fuse_rope) unpack8(*(const u32x4*)((which == 0 ? G.norm_q : G.norm_k) + q8), nw8);

#pragma unroll
  for (int i = 0; i < FM; ++i) {
    // registers -> LDS (row-major fp32, padded rows)
    acc.stage(i, region, lane);
    const int mbase = m0 + wrow0 + i * 32;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {  // two batches of 4 row groups: bounds the prefetch registers
      u32x4 resv[4];
      if (epi == RF_EPI_GATE_RES && G.residual != nullptr) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = mbase + (ib * 4 + it) * 4 + rsub;
          if (m < M && nok) resv[it] = *(const u32x4*)(G.residual + (int64_t)m * G.ldr + n);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = (ib * 4 + it) * 4 + rsub;
        const int m = mbase + row;
        const f32x4 lo = *(const f32x4*)(region + row * EPI_ROW + c0 * 16);
        const f32x4 hi = *(const f32x4*)(region + row * EPI_ROW + c0 * 16 + 16);
        if (epi == EPI_PARTIAL) {
          if (m < M && nok) {
            float* dstp = p.ws + (int64_t)blockIdx.y * p.ws_slice + (int64_t)m * p.ws_ld + n;
            *(f32x4*)dstp = lo;
            *(f32x4*)(dstp + 4) = hi;
          }
          continue;
        }
        if (m < M && nok) {
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if constexpr (W8) {  // dequantise: fp8 products were accumulated unscaled; y = acc * s_act[m] * s_w[n]
            const float sa = G.a_scale[m];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= sa * sw8[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias8[e];
          if (epi == RF_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
          } else if (epi == RF_EPI_GATE_RES) {
            float rr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) rr[e] = 0.f;
            if (G.residual != nullptr) unpack8(resv[it], rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(gate8[e], v[e], rr[e]);
          }
          bf16_t* dst;
          if (epi == RF_EPI_QKV) {
            if (fuse_rope) {
              const int64_t trow = (int64_t)(G.tok_offset + m) * 128 + q8;
              const f32x4 ca = *(const f32x4*)(p.rope_cos + trow), cb = *(const f32x4*)(p.rope_cos + trow + 4);
              const f32x4 sa = *(const f32x4*)(p.rope_sin + trow), sb = *(const f32x4*)(p.rope_sin + trow + 4);
              const float cs[8] = {ca[0], ca[1], ca[2], ca[3], cb[0], cb[1], cb[2], cb[3]};
              const float sn[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
              float ss = 0.f;
#pragma unroll
              for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
#pragma unroll
              for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
              const float rs = rsqrtf(fmaf(ss, 1.0f / 128.0f, p.norm_eps));
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const float a = v[e] * rs * nw8[e], b = v[e + 1] * rs * nw8[e + 1];
                v[e] = fmaf(a, cs[e], -(b * sn[e]));      // explicit contraction: every epilogue of the library rounds the same way
                v[e + 1] = fmaf(b, cs[e + 1], a * sn[e + 1]);
              }
            }
            if (which == 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= p.q_scale;   // softmax scale * log2(e) folded into q (fp32, pre-rounding)
            }
            dst = (which == 0 ? p.q : p.k) + ((int64_t)head * p.s_pad + G.tok_offset + m) * 128 + q8;
          } else {
            dst = G.out + (int64_t)m * G.ldo + (n - ncol_base);
          }
          *(u32x4*)dst = pack8(v);   // (non-temporal stores here were measured neutral to -1 %: profiles/r02, tools/kb_nt_store.py)
        }
      }
    }
  }
}

template <int FM, bool W8 = false>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, const GemmGroupDev& G, f32x16 (&acc)[FM][4],
                                                  const int m0, const int n0, const int wrow0, const int wcol0,
                                                  const int lane, char* region) {
  gemm_epilogue_lds_v<FM, W8>(p, G, Acc32<FM>{acc}, m0, n0, wrow0, wcol0, lane, region);
}
template <int FM, bool W8 = false>
__device__ __forceinline__ void gemm_epilogue_lds16(const GemmParams& p, const GemmGroupDev& G, f32x4 (&acc)[2 * FM][8],
                                                    const int m0, const int n0, const int wrow0, const int wcol0,
                                                    const int lane, char* region) {
  gemm_epilogue_lds_v<FM, W8>(p, G, Acc16<FM>{acc}, m0, n0, wrow0, wcol0, lane, region);
}

// ---- main loop ----------------------------------------------------------------------------------------
// acc += A[m0.., K-tiles kt_begin .. kt_begin+nk) . W[n0.., same)^T over the concatenation of G's K segments.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_mainloop(const GemmGroupDev& G, const int N, const int m0, const int n0, const int kt_begin,
                                              const int nk, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], char* smem, const int w,
                                              const int lane) {
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 32, FN = TN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int IA = BM * 8 / NT, IB = BN * 8 / NT;  // LDS-DMA instructions per thread per stage
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/thread mismatch");
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
  const int wm = w / WN, wn = w % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int M = G.M;

  // ---- per-lane DMA source offsets ------------------------------------------------------
  // DMA instruction i of wave w fills LDS bytes [(i*NW + w)*1024, +1024): 8 rows of 128 B.
  // lane -> (row = +lane/8, physical chunk = lane%8); it fetches logical chunk = pc ^ swz(row).
  // The loads are buffer_load_dwordx4 ... lds: SGPR resource (segment base) + 32-bit per-lane byte offset + SGPR
  // K offset.  Against global_load_lds with 64-bit per-lane addresses this halves the address VGPRs and removes the
  // per-piece 64-bit VALU add; the s_memtime timeline (tools/kb_timeline.py) showed the DMA *issue* -- not its
  // latency: the end-of-tile drain is ~50 cycles -- to be the non-MFMA half of a wave's K-tile.
  uint32_t offA[IA], offB[IB];
  rsrc_t rsA, rsB;
  auto setup_ptrs = [&](const bf16_t* Ab, int64_t lda, const bf16_t* Wb, int64_t ldw) {
    rsA = RF_MAKE_RSRC(Ab);
    rsB = RF_MAKE_RSRC(Wb);
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int row = (i * NW + w) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int gm = m0 + row;
      gm = gm < M ? gm : M - 1;
      offA[i] = (uint32_t)(((int64_t)gm * lda + chunk * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int row = (i * NW + w) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int gn = n0 + row;
      gn = gn < N ? gn : N - 1;
      offB[i] = (uint32_t)(((int64_t)gn * ldw + chunk * 8) * 2);
    }
  };

  // piece q of a stage: q < IA -> A piece q, else W piece q - IA
  auto stage_piece = [&](int q, int kt_in_seg, int buf) {
    char* base = smem + buf * STAGE;
    if (q < IA) {
      RF_BUF_LOAD_LDS(rsA, (lds_void*)(base + (q * NW + w) * 1024), offA[q], kt_in_seg * 128);
    } else {
      RF_BUF_LOAD_LDS(rsB, (lds_void*)(base + A_BYTES + ((q - IA) * NW + w) * 1024), offB[q - IA], kt_in_seg * 128);
    }
  };
  auto stage = [&](int kt_in_seg, int buf) {
#pragma unroll
    for (int q = 0; q < IA + IB; ++q) stage_piece(q, kt_in_seg, buf);
  };

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (bytes): row*128 + ((ks*2 + h) ^ swz)*16, swz = (l31>>1)&7
  const int swz = (l31 >> 1) & 7;
  const int a_row_off = (wm * TM + l31) * 128;
  const int b_row_off = A_BYTES + (wn * TN + l31) * 128;

  // (seg, kk) is the NEXT tile to stage; position at the range's first tile (host compacts segments)
  int seg = 0, kk = kt_begin, cur_nk = G.seg[0].nk;
  while (kk >= cur_nk && seg < 2) {
    kk -= cur_nk;
    ++seg;
    cur_nk = G.seg[seg].nk;
  }
  auto advance = [&]() {  // move (seg,kk) to the next tile; re-point the DMA sources at a boundary.
    ++kk;                 // (a used segment is never followed by an empty one)
    if (kk >= cur_nk) {
      kk = 0;
      ++seg;
      if (seg == 1 && G.seg[1].nk > 0) {
        cur_nk = G.seg[1].nk;
        setup_ptrs(G.seg[1].A, G.seg[1].lda, G.seg[1].W, G.seg[1].ldw);
      } else if (seg == 2 && G.seg[2].nk > 0) {
        cur_nk = G.seg[2].nk;
        setup_ptrs(G.seg[2].A, G.seg[2].lda, G.seg[2].W, G.seg[2].ldw);
      }
    }
  };
  if (seg == 0) setup_ptrs(G.seg[0].A, G.seg[0].lda, G.seg[0].W, G.seg[0].ldw);  // segment 0 is never empty
  else if (seg == 1) setup_ptrs(G.seg[1].A, G.seg[1].lda, G.seg[1].W, G.seg[1].ldw);
  else setup_ptrs(G.seg[2].A, G.seg[2].lda, G.seg[2].W, G.seg[2].ldw);
  stage(kk, 0);
  advance();

  for (int kt = 0; kt < nk; ++kt) {
    // explicit drain of this wave's LDS-DMA before the barrier: never rely on the compiler's own
    // vmcnt placement for LDS-DMA in a loop (see attention.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed for every wave; every wave is done with tile kt-1
    // The next tile's 8 LDS-DMA pieces are issued BEHIND the fragment reads of the first three k-steps rather than all
    // at once after the barrier: the matrix pipe restarts ~300 cycles earlier per K-tile (+5-10 %, measured; profiles/r01_gemm_variants.md)
    const bool more = kt + 1 < nk;
    const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((ks * 2 + h) ^ swz) << 4;
      bf16x8 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *(const bf16x8*)(base + a_row_off + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *(const bf16x8*)(base + b_row_off + j * 32 * 128 + coff);
      if (more) {
        // next tile's DMA pieces: 3, 3, 2, 0 over the four k-steps -- the last piece gets a whole k-step to land before
        // the drain at the end of the tile (vs 2,2,2,2: +3..7 % on the FLUX shapes; 4,4,0,0 equal; 0,3,3,2 -3 %)
        constexpr int NP = IA + IB;
        const int q0 = ks * 3 < NP ? ks * 3 : NP, q1 = (ks + 1) * 3 < NP ? (ks + 1) * 3 : NP;
#pragma unroll
        for (int q = 0; q < NP; ++q)
          if (q >= q0 && q < q1) stage_piece(q, kk, (kt + 1) & 1);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) advance();
  }
}

// local tile id of a group -> (tm, tn).  Grouped raster: column bands of GW tiles; consecutive ids (= one XCD's
// chunk) walk tm inside a band, so the CUs of an XCD share a few A row-panels AND a few W column-panels in their
// private L2 (+5 % at 8192^3, neutral on the FLUX shapes; profiles/r01_gemm_variants.md)
__device__ __forceinline__ void tile_coords(const int lt, const int tiles_m, const int tiles_n, int& tm, int& tn) {
  constexpr int GW = 8;
  const int band = lt / (GW * tiles_m);
  const int first = band * GW;
  const int gw = (tiles_n - first) < GW ? (tiles_n - first) : GW;
  const int rr = lt - band * GW * tiles_m;
  tm = rr / gw;
  tn = first + rr % gw;
}

// VEC: LDS-staged 16-byte epilogue (all pointers 16-byte aligned, N % 8 == 0) vs the per-element fallback
// NO PACKED-FP32 VALU OPS IN THIS KERNEL (round 5).  Two workgroups of the 128 x 128 form share a CU (67.5 KiB of LDS, 224 registers:
// two waves per SIMD from DIFFERENT workgroups, one in its K loop while the other is in its epilogue).  In that state the fused
// RMSNorm + RoPE epilogue returned wrong values -- the low half of a `v_pk_fma_f32 ... neg_lo neg_hi` result, lanes 48-63 only, a
// different handful of q / k rows on every run (tools/kb_qkv_bitstable.py: > 256 tiles + fused RoPE: every run differs; one workgroup
// per CU, or the same code compiled without packed-fp32 ops: bit-stable; waiting out every load before the arithmetic: no change).
// The 256 x 256 kernels run one workgroup per CU with a barrier between K loop and epilogue and have never shown it (bit-stability
// tests at cfg2 / cfg4 / cfg5 sizes).  Found by tests/test_fullsize_gpu.py::test_fast_denoise_is_bit_equal_... at 512 + 256 tokens.
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define RF_NO_PACKED_FP32   // (the host pass does not know the feature)
#endif
template <int BM, int BN, int WM, int WN, bool VEC>
RF_NO_PACKED_FP32 __global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const GemmParams p) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 32, FN = TN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WN, wn = w % WN;

  // ---- tile decode (wave-uniform) -------------------------------------------------------
  const int tile = xcd_remap(blockIdx.x, p.total_tiles);
  int gi = 0;
#pragma unroll
  for (int t = 1; t < 4; ++t)
    if (t < p.ngroups && tile >= p.g[t].tile_start) gi = t;
  const GemmGroupDev& G = p.g[gi];
  int tm, tn;
  tile_coords(tile - G.tile_start, G.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // with split-K this block covers K-tiles [kt_begin, kt_begin + nk) of the segment concatenation
  const int nk_all = G.seg[0].nk + G.seg[1].nk + G.seg[2].nk;
  const int kt_begin = p.ksplit > 1 ? (int)blockIdx.y * p.kchunk : 0;
  const int nk = p.ksplit > 1 ? ((nk_all - kt_begin) < p.kchunk ? (nk_all - kt_begin) : p.kchunk) : nk_all;

  f32x16 acc[FM][FN];
  gemm_mainloop<BM, BN, WM, WN>(G, p.N, m0, n0, kt_begin, nk, acc, smem, w, lane);

  if constexpr (VEC) {
    static_assert(FN == 4, "LDS-staged epilogue expects 128-column wave strips");
    __syncthreads();  // every wave is done reading the staged operands: the LDS is free
    gemm_epilogue_lds<FM>(p, G, acc, m0, n0, wm * TM, wn * TN, lane, smem + w * EPI_REGION);
  } else {
    gemm_epilogue<FM, FN>(p, G, acc, m0, n0, wm * TM, wn * TN, lane & 31, lane >> 5);
  }
}

// ---- ping-pong main loop (256x256x64 tile, 8 waves) -------------------------------------------------------
// s_memtime timeline of the plain loop (tools/kb_timeline.py): the two waves of a SIMD do NOT share its matrix pipe
// fairly -- the older wave wins every arbitration, finishes its K-tile at single-wave speed (~1590 cycles) and then
// waits ~1000 cycles at the barrier while the younger one finishes alone at ~60 % duty: 2750 cycles per K-tile for
// 2048 cycles of MFMA work.  Here the alternation is enforced: the 8 waves form two groups of 4 (one wave of each
// group per SIMD) that run half a phase apart; while one group multiplies, the other reads fragments and issues DMA.
// A K-tile is 4 phases; a phase multiplies one 32x64 quadrant of the wave's 64x128 tile over the whole BK=64:
//     p0: read A0,B0 | Q(A0,B0)     p1: read A1 | Q(A1,B0)     p2: read B1 | Q(A1,B1)     p3: -- | Q(A0,B1)
// The tile is staged as four 16 KiB half-tiles {A0, A1, B0, B1} (sub-block s of EVERY wave), so each half-tile is
// read in exactly one phase and can be re-staged two phases later while its tile is still being multiplied:
//     p0: stage A1(t+1)   p1: stage B1(t+1)   p2: stage A0(t+2)   p3: stage B0(t+2); s_waitcnt vmcnt(4)
// Hazards (cdna_hip_programming.md, "256^2 8-phase template"): a staged half-tile is read >= 1 phase after the
// counted vmcnt + barrier that retires it (the p3 wait leaves only the two newest half-tiles in flight, which are
// first read a tile later); a half-tile is re-staged >= 2 phases after its last read.  vmcnt never drains to 0 in
// steady state.  (A first build of this schedule with global_load_lds + 64-bit per-lane addresses lost 8 %: its
// load phases were longer than the 256-cycle MFMA phases; buffer_load ... lds makes them fit.)
// W8 = true: the operands are fp8 e4m3 (rf_gemm_w8a8).  A K-tile is then 128 ELEMENTS but the same 128 BYTES per row,
// so staging, LDS image, swizzle, barriers and phases are byte-identical; only the multiply changes: per 128-byte row
// two v_mfma_scale_f32_32x32x64_f8f6f4 (64 fp8 per lane pair, unit E8M0 block scales; 64 cycles each at twice the
// bf16 FLOP rate) instead of four v_mfma_f32_32x32x16_bf16.  A lane's 32 operand bytes of a k-step are the two adjacent
// 16-byte chunks (4*j + 2*h, +1); A and W use the same (lane, byte) -> k map, which is all a dot product needs
// (tools/ubench/mx_probe.py: rows = lane % 32, results identical under every consistent k permutation).
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ i32x8 cat_frag(const bf16x8& lo, const bf16x8& hi) {
  const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
  i32x8 r;
  r[0] = (int)a[0]; r[1] = (int)a[1]; r[2] = (int)a[2]; r[3] = (int)a[3];
  r[4] = (int)b[0]; r[5] = (int)b[1]; r[6] = (int)b[2]; r[7] = (int)b[3];
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define RF_MFMA_FP8(a, b, c) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127)
#define RF_MFMA_FP8_16(a, b, c) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127)
#else
#define RF_MFMA_FP8(a, b, c) (c)
#define RF_MFMA_FP8_16(a, b, c) (c)
#endif


// The same balanced schedule on v_mfma_f32_16x16x32_bf16.  Under the 1.4 kW cap the matrix pipes sustain 1.82 PFLOP/s with
// 32x32x16 MFMAs on random bf16 operands and 2.06 PFLOP/s with 16x16x32 (tools/ubench/mfma_power.py, no memory traffic at
// all: the chip settles at 1.89 vs 2.13 GHz) -- the small shape reads and writes half the accumulator bytes per MAC.  Fragment
// counts per phase, LDS image, staging and barriers are unchanged (a 32-row A half = 2 row tiles x 2 k-steps = 4 fragments,
// a 64-column W half = 4 x 2 = 8); only the lane -> (row, chunk) map of a fragment read and the accumulator layout differ.
template <bool W8 = false>
__device__ __forceinline__ void gemm_mainloop_pp2_m16(const GemmGroupDev& G, const int N, const int m0, const int n0, const int kt_begin,
                                                  const int nk, f32x4 (&acc)[4][8], char* smem, const int w, const int lane) {
  constexpr int ESZ = W8 ? 1 : 2;  // bytes per element
  constexpr int HT = 128 * 128;  // half-tile bytes
  constexpr int BUF = 4 * HT;    // {A0, A1, B0, B1} of one K-tile
  const int wm = w >> 1, wn = w & 1, grp = w >> 2;
  const int l31 = lane & 31, h = lane >> 5;
  const int M = G.M;

  // staging geometry: DMA instruction i (0,1) of wave w fills local rows (i*8 + w)*8 + lane/8 of a half-tile;
  // local row lr of A_s is tile row (lr>>5)*64 + s*32 + (lr&31), of B_s tile column (lr>>6)*128 + s*64 + (lr&63)
  const int r8 = lane >> 3;
  const uint32_t chunk_b = (uint32_t)(((lane & 7) ^ (((w & 1) << 2) + (lane >> 4))) * 16);  // swizzled 16-byte chunk
  // K-tile cursors of the tiles being staged (c1 = tile t+1, c2 = tile t+2): segment, tile-in-segment and the
  // segment's buffer resources / row pitches in SGPRs (re-loaded only when a cursor crosses a segment boundary)
  // (the per-lane byte offsets row * pitch + chunk are formed HERE, once per segment: in the loop they were two
  // v_mad_u64_u32 per stage call = 16 quarter-rate VALU per K-tile in the load phases, which are the critical ones)
  struct Cur { int seg, kk, nk; rsrc_t A, W; uint32_t offA[2][2], offB[2][2]; };
  auto load_seg = [&](Cur& c) {
    const KSegDev& S = G.seg[c.seg];
    c.nk = S.nk; c.A = RF_MAKE_RSRC(S.A); c.W = RF_MAKE_RSRC(S.W);
    const uint32_t lda2 = (uint32_t)(S.lda * ESZ), ldw2 = (uint32_t)(S.ldw * ESZ);
    // (rows are re-derived here, once per segment, instead of living in eight registers across the loop)
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int gm = m0 + (2 * i + (w >> 2)) * 64 + sb * 32 + 8 * (w & 3) + r8;
        const int gn = n0 + i * 128 + sb * 64 + 8 * w + r8;
        c.offA[sb][i] = (uint32_t)(gm < M ? gm : M - 1) * lda2 + chunk_b;
        c.offB[sb][i] = (uint32_t)(gn < N ? gn : N - 1) * ldw2 + chunk_b;
      }
  };
  auto next = [&](Cur& c) {
    ++c.kk;
    if (c.kk >= c.nk && c.seg < 2 && G.seg[c.seg + 1].nk > 0) {
      c.kk = 0;
      ++c.seg;
      load_seg(c);
    }
  };
  // kind: 0 = A0, 1 = A1, 2 = B0, 3 = B1
  auto stage = [&](const int kind, const Cur& c, const int buf) {
    char* dst = smem + buf * BUF + kind * HT + w * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (kind < 2) RF_BUF_LOAD_LDS(c.A, (lds_void*)(dst + i * 8192), c.offA[kind & 1][i], c.kk * 128);
      else RF_BUF_LOAD_LDS(c.W, (lds_void*)(dst + i * 8192), c.offB[kind & 1][i], c.kk * 128);
    }
  };

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets inside a buffer (16x16x32 fragments: lane -> row lane&15, 16-byte chunk 4*ks + lane>>4)
  const int l15 = lane & 15;
  const int swz = (l15 >> 1) & 7;
  const int a_off = (wm * 32 + l15) * 128;                 // + sb*HT + rt*16*128
  const int b_off = 2 * HT + (wn * 64 + l15) * 128;        // + sb*HT + ct*16*128
  auto frag_coff = [&](int ks) { return ((ks * 4 + (lane >> 4)) ^ swz) << 4; };
  // A half: fragments [rt*2 + ks] (2 row tiles x 2 k-steps); W half: [ct*2 + ks] (4 column tiles x 2 k-steps)
  auto rdA = [&](bf16x8 (&dst)[4], const char* half) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dst[rt * 2 + ks] = *(const bf16x8*)(half + a_off + rt * 2048 + frag_coff(ks));
  };
  auto rdB = [&](bf16x8 (&dst)[8], const char* half) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)      // k-step 0 of all four column tiles first: the phase's first MFMAs need those
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) dst[ct * 2 + ks] = *(const bf16x8*)(half + b_off + ct * 2048 + frag_coff(ks));
  };
  // one phase: the 32 x 64 quadrant (row tiles rb, rb+1) x (column tiles cb .. cb+3) over the whole K-tile, 16 MFMAs;
  // the two updates of an accumulator are 8 MFMAs apart
  auto mma16 = [&](const int rb, const int cb, const bf16x8 (&A)[4], const bf16x8 (&B)[8]) {
    if constexpr (!W8) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[rb + rt][cb + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rt * 2 + ks], B[ct * 2 + ks], acc[rb + rt][cb + ct], 0, 0, 0);
    } else {
      // fp8: a 128-byte row is the whole k-step of v_mfma_scale_f32_16x16x128_f8f6f4; a lane's 32 operand bytes are
      // k = 16g .. 16g+15 and 64+16g .. (g = lane >> 4; measured, profiles/r02_mx_probe.md) = chunks g and 4 + g, which
      // are exactly the two fragment reads the bf16 map makes for its k-steps 0 and 1.  Unit E8M0 scales.
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const i32x8 av = cat_frag(A[rt * 2], A[rt * 2 + 1]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rb + rt][cb + ct] = RF_MFMA_FP8_16(av, cat_frag(B[ct * 2], B[ct * 2 + 1]), acc[rb + rt][cb + ct]);
      }
      // pin the results to this phase (see mma_quadrant: IR passes sink scaled MFMAs below the phase barriers)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        asm volatile("" : "+v"(acc[rb + rt][cb]), "+v"(acc[rb + rt][cb + 1]), "+v"(acc[rb + rt][cb + 2]), "+v"(acc[rb + rt][cb + 3]));
    }
  };

  Cur c1;
  c1.seg = 0; c1.kk = kt_begin;
  while (c1.seg < 2 && c1.kk >= G.seg[c1.seg].nk && G.seg[c1.seg + 1].nk > 0) {
    c1.kk -= G.seg[c1.seg].nk;
    ++c1.seg;
  }
  load_seg(c1);
  // prologue, in steady-state issue order: A0(0) | B0(0) A1(0) | A0(1) B1(0) | B0(1) A1(1)
  Cur c2 = c1;
  if (nk > 1) {
    next(c2);                                        // c1 = tile 0, c2 = tile 1
    stage(0, c1, 0); stage(2, c1, 0); stage(1, c1, 0);
    stage(0, c2, 1); stage(3, c1, 0);
    stage(2, c2, 1); stage(1, c2, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0 but B1 has landed
    c1 = c2;                                         // c1 -> tile 1
    next(c2);                                        // c2 -> tile 2
  } else {
    stage(0, c1, 0); stage(2, c1, 0); stage(1, c1, 0); stage(3, c1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 X[4], Y[4], bq[8];
  rdA(X, smem);   // A0 of tile 0
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs half a phase behind group 0
  __builtin_amdgcn_sched_barrier(0);

#define RF_PP2_BAR()                    \
  __builtin_amdgcn_sched_barrier(0);    \
  __builtin_amdgcn_s_barrier();         \
  __builtin_amdgcn_sched_barrier(0)
  // one K-tile: P holds its A0 fragments (read during the previous tile's phase 3), Q receives A1, then the next A0
  auto tile = [&](const int t, bf16x8 (&P)[4], bf16x8 (&Q)[4]) {
    const char* base = smem + (t & 1) * BUF;
    const char* nbase = smem + ((t + 1) & 1) * BUF;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // ---- p0: 8 reads ---------------------------------------------------------------------
    rdB(bq, base);
    RF_PP2_BAR();
    mma16(0, 0, P, bq);
    RF_PP2_BAR();
    // ---- p1: 4 reads, 4 pieces -----------------------------------------------------------
    rdA(Q, base + HT);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      stage(0, c2, t & 1); stage(3, c1, (t + 1) & 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (more1) {
      stage(3, c1, (t + 1) & 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RF_PP2_BAR();
    mma16(2, 0, Q, bq);
    RF_PP2_BAR();
    // ---- p2: 8 reads ---------------------------------------------------------------------
    rdB(bq, base + HT);
    RF_PP2_BAR();
    mma16(2, 4, Q, bq);
    // the next tile's A0 goes into Q: keep its reads behind these MFMAs' operand fetch
    asm volatile("" : "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3]));
    RF_PP2_BAR();
    // ---- p3: 4 reads (next tile's A0), 4 pieces ------------------------------------------
    if (more1) rdA(Q, nbase);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      stage(2, c2, t & 1); stage(1, c2, t & 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RF_PP2_BAR();
    mma16(0, 4, P, bq);
    RF_PP2_BAR();
    next(c1);
    next(c2);
  };
  for (int t = 0; t < nk; t += 2) {
    tile(t, X, Y);
    if (t + 1 < nk) tile(t + 1, Y, X);
  }
#undef RF_PP2_BAR
  if (grp == 0) __builtin_amdgcn_s_barrier();  // match group 1's extra barrier
}


// THE SHIPPED bf16 LOOP (gemm_bf16_pp16e_kernel; the stream-K kernel's EVEN form): the evenly loaded 6/6/6/6 + 2/2/2/2 ping-pong
// schedule on 16x16x32 MFMAs: the k-step-0 fragments of column tiles 0, 1 of each W half are read one phase early (e0 / e1), the other six
// share one register set m; a phase starts with the four MFMAs that need only the early pair.
__device__ __forceinline__ void gemm_mainloop_pp3_m16(const GemmGroupDev& G, const int N, const int m0, const int n0, const int kt_begin,
                                                  const int nk, f32x4 (&acc)[4][8], char* smem, const int w, const int lane) {
  constexpr int ESZ = 2;  // bytes per element
  constexpr int HT = 128 * 128;  // half-tile bytes
  constexpr int BUF = 4 * HT;    // {A0, A1, B0, B1} of one K-tile
  const int wm = w >> 1, wn = w & 1, grp = w >> 2;
  const int l31 = lane & 31, h = lane >> 5;
  const int M = G.M;

  // staging geometry: DMA instruction i (0,1) of wave w fills local rows (i*8 + w)*8 + lane/8 of a half-tile;
  // local row lr of A_s is tile row (lr>>5)*64 + s*32 + (lr&31), of B_s tile column (lr>>6)*128 + s*64 + (lr&63)
  const int r8 = lane >> 3;
  const uint32_t chunk_b = (uint32_t)(((lane & 7) ^ (((w & 1) << 2) + (lane >> 4))) * 16);  // swizzled 16-byte chunk
  // K-tile cursors of the tiles being staged (c1 = tile t+1, c2 = tile t+2): segment, tile-in-segment and the
  // segment's buffer resources / row pitches in SGPRs (re-loaded only when a cursor crosses a segment boundary)
  // (the per-lane byte offsets row * pitch + chunk are formed HERE, once per segment: in the loop they were two
  // v_mad_u64_u32 per stage call = 16 quarter-rate VALU per K-tile in the load phases, which are the critical ones)
  struct Cur { int seg, kk, nk; rsrc_t A, W; uint32_t offA[2][2], offB[2][2]; };
  auto load_seg = [&](Cur& c) {
    const KSegDev& S = G.seg[c.seg];
    c.nk = S.nk; c.A = RF_MAKE_RSRC(S.A); c.W = RF_MAKE_RSRC(S.W);
    const uint32_t lda2 = (uint32_t)(S.lda * ESZ), ldw2 = (uint32_t)(S.ldw * ESZ);
    // (rows are re-derived here, once per segment, instead of living in eight registers across the loop)
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int gm = m0 + (2 * i + (w >> 2)) * 64 + sb * 32 + 8 * (w & 3) + r8;
        const int gn = n0 + i * 128 + sb * 64 + 8 * w + r8;
        c.offA[sb][i] = (uint32_t)(gm < M ? gm : M - 1) * lda2 + chunk_b;
        c.offB[sb][i] = (uint32_t)(gn < N ? gn : N - 1) * ldw2 + chunk_b;
      }
  };
  auto next = [&](Cur& c) {
    ++c.kk;
    if (c.kk >= c.nk && c.seg < 2 && G.seg[c.seg + 1].nk > 0) {
      c.kk = 0;
      ++c.seg;
      load_seg(c);
    }
  };
  // kind: 0 = A0, 1 = A1, 2 = B0, 3 = B1
  auto stage = [&](const int kind, const Cur& c, const int buf) {
    char* dst = smem + buf * BUF + kind * HT + w * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (kind < 2) RF_BUF_LOAD_LDS(c.A, (lds_void*)(dst + i * 8192), c.offA[kind & 1][i], c.kk * 128);
      else RF_BUF_LOAD_LDS(c.W, (lds_void*)(dst + i * 8192), c.offB[kind & 1][i], c.kk * 128);
    }
  };

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets inside a buffer (16x16x32 fragments: lane -> row lane&15, 16-byte chunk 4*ks + lane>>4)
  const int l15 = lane & 15;
  const int swz = (l15 >> 1) & 7;
  const int a_off = (wm * 32 + l15) * 128;                 // + sb*HT + rt*16*128
  const int b_off = 2 * HT + (wn * 64 + l15) * 128;        // + sb*HT + ct*16*128
  auto frag_coff = [&](int ks) { return ((ks * 4 + (lane >> 4)) ^ swz) << 4; };
  // A half: fragments [rt*2 + ks] (2 row tiles x 2 k-steps); W half: [ct*2 + ks] (4 column tiles x 2 k-steps)
  auto rdA = [&](bf16x8 (&dst)[4], const char* half) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dst[rt * 2 + ks] = *(const bf16x8*)(half + a_off + rt * 2048 + frag_coff(ks));
  };
  // W half fragments [ct][ks]: early pair e = {(ct 0, ks 0), (ct 1, ks 0)}, main m = {(2,0), (3,0), (0,1), (1,1), (2,1), (3,1)}
  auto rdBe = [&](bf16x8 (&e)[2], const char* half) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) e[ct] = *(const bf16x8*)(half + b_off + ct * 2048 + frag_coff(0));
  };
  auto rdBm = [&](bf16x8 (&m)[6], const char* half) {
#pragma unroll
    for (int ct = 2; ct < 4; ++ct) m[ct - 2] = *(const bf16x8*)(half + b_off + ct * 2048 + frag_coff(0));
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) m[2 + ct] = *(const bf16x8*)(half + b_off + ct * 2048 + frag_coff(1));
  };
  auto mma16s = [&](const int rb, const int cb, const bf16x8 (&A)[4], const bf16x8 (&e)[2], const bf16x8 (&m)[6]) {
    auto B = [&](int ct, int ks) -> const bf16x8& { return ks == 0 ? (ct < 2 ? e[ct] : m[ct - 2]) : m[2 + ct]; };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc[rb + rt][cb + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rt * 2 + ks], B(ct, ks), acc[rb + rt][cb + ct], 0, 0, 0);
  };

  Cur c1;
  c1.seg = 0; c1.kk = kt_begin;
  while (c1.seg < 2 && c1.kk >= G.seg[c1.seg].nk && G.seg[c1.seg + 1].nk > 0) {
    c1.kk -= G.seg[c1.seg].nk;
    ++c1.seg;
  }
  load_seg(c1);
  // prologue, in steady-state issue order: A0(0) B0(0) A1(0) B1(0) | A0(1) B0(1) A1(1)      (B1(1) goes out in p0 of tile 0)
  Cur c2 = c1;
  stage(0, c1, 0); stage(2, c1, 0); stage(1, c1, 0); stage(3, c1, 0);
  if (nk > 1) {
    next(c2);
    stage(0, c2, 1); stage(2, c2, 1); stage(1, c2, 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // A0(0), B0(0) have landed
    c1 = c2;
    next(c2);
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 X[4], Y[4], e0[2], e1[2], m[6];
  rdA(X, smem);
  rdBe(e0, smem);
  // static wave priority for group 0 for the whole loop (round 6, tools/kb_gemm_patch.py: +0.3-0.4 % on every cfg2 shape in 7 interleaved
  // repetitions, bit-identical; per-phase priorities around the MFMAs or around the loads, and the same for group 1, measured +-0.2 %)
  if (grp == 0) __builtin_amdgcn_s_setprio(1);
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs half a phase behind group 0
  __builtin_amdgcn_sched_barrier(0);

#define RF_PP3_BAR()                    \
  __builtin_amdgcn_sched_barrier(0);    \
  __builtin_amdgcn_s_barrier();         \
  __builtin_amdgcn_sched_barrier(0)
  auto tile = [&](const int t, bf16x8 (&P)[4], bf16x8 (&Q)[4]) {
    const char* base = smem + (t & 1) * BUF;
    const char* nbase = smem + ((t + 1) & 1) * BUF;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // ---- p0: B0 main; DMA B1(t+1) ----
    rdBm(m, base);
    __builtin_amdgcn_sched_barrier(0);
    if (more1) {
      stage(3, c1, (t + 1) & 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RF_PP3_BAR();
    mma16s(0, 0, P, e0, m);
    RF_PP3_BAR();
    // ---- p1: A1, B1 early; DMA A0(t+2) ----
    rdA(Q, base + HT);
    rdBe(e1, base + HT);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) stage(0, c2, t & 1);
    RF_PP3_BAR();
    mma16s(2, 0, Q, e0, m);
    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]));
    RF_PP3_BAR();
    // ---- p2: B1 main; DMA B0(t+2) ----
    rdBm(m, base + HT);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      stage(2, c2, t & 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RF_PP3_BAR();
    mma16s(2, 4, Q, e1, m);
    asm volatile("" : "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3]));
    RF_PP3_BAR();
    // ---- p3: next tile's A0 and B0 early; DMA A1(t+2) ----
    if (more1) {
      rdA(Q, nbase);
      rdBe(e0, nbase);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more2) stage(1, c2, t & 1);
    RF_PP3_BAR();
    mma16s(0, 4, P, e1, m);
    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]));
    RF_PP3_BAR();
    next(c1);
    next(c2);
  };
  for (int t = 0; t < nk; t += 2) {
    tile(t, X, Y);
    if (t + 1 < nk) tile(t + 1, Y, X);
  }
#undef RF_PP3_BAR
  __builtin_amdgcn_s_setprio(0);
  if (grp == 0) __builtin_amdgcn_s_barrier();  // match group 1's extra barrier
}

// launches on the 16x16 MFMA shapes (gemm_mainloop_pp2_m16): bf16, or W8 = mixed precision per token group
template <bool W8>
__device__ __forceinline__ void gemm_pp16_body(const GemmParams& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ClkProbe clk;
  if (p.probe) clk.begin();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, p.total_tiles);
  int gi = 0;
#pragma unroll
  for (int t = 1; t < 4; ++t)
    if (t < p.ngroups && tile >= p.g[t].tile_start) gi = t;
  const GemmGroupDev& G = p.g[gi];
  int tm, tn;
  tile_coords(tile - G.tile_start, G.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = G.seg[0].nk + G.seg[1].nk + G.seg[2].nk;
  f32x4 acc[4][8];
  if (W8 && G.w8) {
    gemm_mainloop_pp2_m16<W8>(G, p.N, m0, n0, 0, nk, acc, smem, w, lane);
    if (p.probe) clk.end(g_clk_probe);
    __syncthreads();
    gemm_epilogue_lds16<2, W8>(p, G, acc, m0, n0, (w >> 1) * 64, (w & 1) * 128, lane, smem + w * EPI_REGION);
  } else {
    gemm_mainloop_pp2_m16<false>(G, p.N, m0, n0, 0, nk, acc, smem, w, lane);
    if (p.probe) clk.end(g_clk_probe);
    __syncthreads();  // every wave is done reading the staged operands: the LDS is free
    gemm_epilogue_lds16<2, false>(p, G, acc, m0, n0, (w >> 1) * 64, (w & 1) * 128, lane, smem + w * EPI_REGION);
  }
  if (p.probe && blockIdx.x == 0 && w == 0) {   // probe only: when has this wave's part of the tile left the CU?
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      unsigned long long c, r;
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
      g_clk_probe_epi[0] = c; g_clk_probe_epi[1] = r;
    }
  }
}
// bf16 launches, evenly loaded phases (gemm_mainloop_pp3_m16)
__global__ __launch_bounds__(512) void gemm_bf16_pp16e_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ClkProbe clk;
  if (p.probe) clk.begin();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, p.total_tiles);
  int gi = 0;
#pragma unroll
  for (int t = 1; t < 4; ++t)
    if (t < p.ngroups && tile >= p.g[t].tile_start) gi = t;
  const GemmGroupDev& G = p.g[gi];
  int tm, tn;
  tile_coords(tile - G.tile_start, G.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = G.seg[0].nk + G.seg[1].nk + G.seg[2].nk;
  f32x4 acc[4][8];
  gemm_mainloop_pp3_m16(G, p.N, m0, n0, 0, nk, acc, smem, w, lane);
  if (p.probe) clk.end(g_clk_probe);
  __syncthreads();  // every wave is done reading the staged operands: the LDS is free
  gemm_epilogue_lds16<2, false>(p, G, acc, m0, n0, (w >> 1) * 64, (w & 1) * 128, lane, smem + w * EPI_REGION);
}
__global__ __launch_bounds__(512) void gemm_w8_pp16_kernel(const GemmParams p) { gemm_pp16_body<true>(p); }

#include "gemm_w4.hpp"
#include "gemm_w4b.hpp"


// ---- stream-K variant ---------------------------------------------------------------------------------
// One persistent block per CU.  The launch's MAC work is measured in K-tile iterations (tile-major) and cut into
// gridDim.x equal contiguous ranges, so 216 or 264 tiles on 256 CUs cost 0.84 / 1.03 tile-times instead of 1 / 2
// (tile quantisation was the largest single GEMM loss: profiles/r01_cfg4_kernel_stats.csv).  A range covers the
// tail of one tile, whole tiles, and the head of another.  Pieces are processed LAST-FIRST:
//   * a piece that does not reach its tile's last K-tile is stored as raw fp32 accumulators into the block's
//     partial slot (register layout, coalesced 16-byte stores) and published with a release flag;
//   * the block holding a tile's final piece adds the partials of the (1..2) blocks before it in index order
//     and runs the normal fused epilogue -- fixed summation order, bit-reproducible.
// Waits only ever target a block with a LOWER hardware index inside the same XCD chunk (ranges are cut per XCD
// at tile boundaries), which was dispatched earlier and produced that partial as its FIRST piece: no deadlock
// even without co-residency, and in practice no spinning.  Flags are reset by their single consumer, so the
// kernel leaves the scratch as it found it (hipGraph-replayable).
struct SkParams {
  int iter_start[5];    // first global iteration of group g (iter_start[ngroups] = total)
  int nk[4];            // K-tiles per tile of group g
  // per-XCD chunk x (cut at tile boundaries): whole-tile rounds first -- worker cl takes tiles
  // chunk_tile[x] + r*PL + cl, r < dp_rounds[x], i.e. the XCD's CUs sweep 32 consecutive tiles together exactly
  // like the one-tile-per-block launch (shared A / W panels in the private L2) -- then the stream-K region, the
  // iterations [sk_begin[x], chunk_end[x]) (one to two tiles per worker), cut evenly
  int chunk_tile[8], dp_rounds[8], sk_begin[8], chunk_end[8];
  int chunk_tiles[8];   // tiles in chunk x (bounds the last, partial whole-tile round of the persistent-tile mode)
  float* partials;      // [gridDim.x] slots of BM*BN floats
  int* flags;           // [gridDim.x], zero outside a launch
};

template <int BM, int BN, int WM, int WN, bool W8, bool MI16 = false, bool EVEN = false>   // MI16: the 16x16 MFMA shapes; EVEN: bf16 groups on gemm_mainloop_pp3_m16
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_sk_kernel(const GemmParams p, const SkParams sk) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 32, FN = TN / 32;
  static_assert(FN == 4, "LDS-staged epilogue expects 128-column wave strips");
  static_assert(NT * 16 == 0x2000, "partial-slot stride is hard-coded in the inline assembly");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WN, wn = w % WN;

  // block b runs on XCD b % 8; it is worker b / 8 of that XCD's chunk
  const int xcd = blockIdx.x & 7, cl = blockIdx.x >> 3, PL = gridDim.x >> 3;
  const int cs = sk.sk_begin[xcd], clen = sk.chunk_end[xcd] - cs;
  auto range_begin = [&](int worker) { return cs + (int)((int64_t)worker * clen / PL); };
  const int it_begin = range_begin(cl);
  int cur_end = range_begin(cl + 1);
  const int dp_rounds = sk.dp_rounds[xcd];
  int round = 0;
  float* const my_slot = sk.partials + (int64_t)blockIdx.x * (BM * BN);
  bool first = true;

  while (true) {
    int gi = 0, lt, ts, ub, nk_u;
    bool is_tail;
    if (round < dp_rounds && round * PL + cl >= sk.chunk_tiles[xcd]) round = dp_rounds;  // partial last round: no tile left for this worker
    if (round < dp_rounds) {
      // whole tile of a full round
      const int tile = sk.chunk_tile[xcd] + round * PL + cl;
      ++round;
#pragma unroll
      for (int t = 1; t < 4; ++t)
        if (t < p.ngroups && tile >= p.g[t].tile_start) gi = t;
      lt = tile - p.g[gi].tile_start;
      nk_u = sk.nk[gi];
      ts = 0; ub = 0;
      is_tail = true;
    } else if (cur_end > it_begin) {
      const int last = cur_end - 1;
#pragma unroll
      for (int t = 1; t < 4; ++t)
        if (t < p.ngroups && last >= sk.iter_start[t]) gi = t;
      const int nk_g = sk.nk[gi];
      lt = (last - sk.iter_start[gi]) / nk_g;
      ts = sk.iter_start[gi] + lt * nk_g;  // the tile's iterations are [ts, ts + nk_g)
      ub = it_begin > ts ? it_begin : ts;
      is_tail = cur_end == ts + nk_g;
      nk_u = cur_end - ub;                 // K-tiles of this piece
      cur_end = ub;
    } else {
      break;
    }
    const GemmGroupDev& G = p.g[gi];
    int tm, tn;
    tile_coords(lt, G.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    if (!first) __syncthreads();  // the previous piece's epilogue staging is done with the LDS
    first = false;
    // launder the lane id once per piece: keeps the per-lane address math of the main loop and the epilogue INSIDE the
    // loop (LICM would hoist ~100 VGPRs of it across the whole kernel and spill)
    int lane_i = lane, tid_i = tid;
    asm volatile("" : "+v"(lane_i), "+v"(tid_i));
    // a thread's accumulators are 32 quads either way: acc[i][j][4rq..] (32x32x16) or acc16[it][jt] (16x16x32); partial
    // sums travel quad by quad in that order, which producer and consumer (the same kernel) share
    typename std::conditional<MI16, f32x4[4][8], f32x16[FM][FN]>::type acc;
    auto quad = [&](const int k) -> f32x4 {
      if constexpr (MI16) return acc[k >> 3][k & 7];
      else return f32x4{acc[k >> 4][(k >> 2) & 3][(k & 3) * 4], acc[k >> 4][(k >> 2) & 3][(k & 3) * 4 + 1],
                        acc[k >> 4][(k >> 2) & 3][(k & 3) * 4 + 2], acc[k >> 4][(k >> 2) & 3][(k & 3) * 4 + 3]};
    };
    auto quad_add = [&](const int k, const f32x4 v) {
      if constexpr (MI16) acc[k >> 3][k & 7] += v;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k >> 4][(k >> 2) & 3][(k & 3) * 4 + e] += v[e];
      }
    };
    const bool g8 = W8 && G.w8;   // mixed-precision launch: multiply chosen per token group
    if constexpr (MI16) {
      if (g8) gemm_mainloop_pp2_m16<W8>(G, p.N, m0, n0, ub - ts, nk_u, acc, smem, w, lane_i);
      else if constexpr (EVEN && !W8) gemm_mainloop_pp3_m16(G, p.N, m0, n0, ub - ts, nk_u, acc, smem, w, lane_i);
      else gemm_mainloop_pp2_m16<false>(G, p.N, m0, n0, ub - ts, nk_u, acc, smem, w, lane_i);
    }
    static_assert(MI16, "librf_flux.so ships the 16x16 MFMA shapes only");

    if (!is_tail) {
      // head or middle piece: raw accumulators -> this block's slot, [quad k][thread] x 16 B, as agent-coherent
      // write-through stores (sc1: what a relaxed agent-scope atomic store compiles to), so publishing them needs
      // no cache-wide writeback; then the flag
      {
        uint32_t voff = (uint32_t)tid_i * 16u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const f32x4 v = quad(k);
          asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\tv_add_u32 %0, 0x2000, %0" : "+v"(voff) : "v"(v), "s"(my_slot) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();  // every thread's stores have been issued and acknowledged
      if (tid == 0) __hip_atomic_store(sk.flags + blockIdx.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (ub > ts) {
        // final piece of a split tile: add the earlier pieces (workers cl-1, cl-2, .. down to the one holding ts)
        for (int c2 = cl - 1;; --c2) {
          const int b2 = xcd + 8 * c2;
          const int rb2 = range_begin(c2);
          if (rb2 == range_begin(c2 + 1)) continue;  // empty range (the host never creates one): nothing to add
          if (tid == 0) {
            while (__hip_atomic_load(sk.flags + b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __hip_atomic_store(sk.flags + b2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // single consumer: reset
          }
          __syncthreads();
          // agent-coherent loads (sc1: they never hit a stale line of this XCD's L2) instead of an acquire fence,
          // which would invalidate the whole L2 under the other 31 CUs' feet.  8 x 16 B in flight per lane.
          const float* slot = sk.partials + (int64_t)b2 * (BM * BN);
          uint32_t voff = (uint32_t)tid_i * 16u;
#pragma unroll
          for (int k8 = 0; k8 < 32; k8 += 8) {
              f32x4 t0, t1, t2, t3, t4, t5, t6, t7;
              asm volatile(
                  "global_load_dwordx4 %0, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %1, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %2, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %3, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %4, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %5, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %6, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "global_load_dwordx4 %7, %8, %9 sc1\n\tv_add_u32 %8, 0x2000, %8\n\t"
                  "s_waitcnt vmcnt(0)"
                  : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "+v"(voff)
                  : "s"(slot)
                  : "memory");
              const f32x4 tt[8] = {t0, t1, t2, t3, t4, t5, t6, t7};
#pragma unroll
              for (int q = 0; q < 8; ++q) quad_add(k8 + q, tt[q]);
            }
          if (rb2 <= ts) break;
        }
      }
      __syncthreads();  // every wave is done reading the staged operands: the LDS is free
      if constexpr (MI16) {
        if (g8) gemm_epilogue_lds16<FM, W8>(p, G, acc, m0, n0, wm * TM, wn * TN, lane_i, smem + w * EPI_REGION);
        else gemm_epilogue_lds16<FM, false>(p, G, acc, m0, n0, wm * TM, wn * TN, lane_i, smem + w * EPI_REGION);
      }
    }
  }
}

template <int BM, int BN>
static void layout_tiles(GemmParams& p) {
  p.tiles_n = cdiv(p.N, BN);
  int start = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    p.g[g].tiles_m = cdiv(p.g[g].M, BM);
    p.g[g].tile_start = start;
    start += p.g[g].tiles_m * p.tiles_n;
  }
  p.total_tiles = start;
}


template <int BM, int BN, int WM, int WN, bool VEC>
static int launch_gemm(GemmParams& p, hipStream_t stream) {
  constexpr int LDS_MAIN = 2 * (BM + BN) * 128, LDS_EPI = WM * WN * EPI_REGION;
  constexpr int LDS = (VEC && LDS_EPI > LDS_MAIN) ? LDS_EPI : LDS_MAIN;
  static bool attr_set = false;
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, VEC>;
  if (!attr_set) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  layout_tiles<BM, BN>(p);
  if (p.total_tiles == 0) return RF_OK;
  hipLaunchKernelGGL(kern, dim3(p.total_tiles, p.ksplit > 1 ? p.ksplit : 1), dim3(WM * WN * 64), LDS, stream, p);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

static int launch_gemm_w4m16(GemmParams& p, hipStream_t stream) {
  constexpr int LDS = 2 * 2 * 256 * 128;   // two stages of {A, W} images; the epilogue's 4 x 16.5 KiB regions fit inside
  static_assert(4 * EPI_REGION <= LDS, "epilogue staging must fit the main loop's LDS");
  static bool attr_set = false;
  if (!attr_set) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_w4m16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  layout_tiles<256, 256>(p);
  if (p.total_tiles == 0) return RF_OK;
  hipLaunchKernelGGL(gemm_bf16_w4m16_kernel, dim3(p.total_tiles), dim3(256), LDS, stream, p);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

static int launch_gemm_w4b(GemmParams& p, hipStream_t stream) {
  constexpr int LDS = 2 * 2 * 32 * (1024 + 32);   // two stages x {A, W} images of 32 padded wave pieces (gemm_w4b.hpp)
  static bool attr_set = false;
  if (!attr_set) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_w4b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  layout_tiles<256, 256>(p);
  if (p.total_tiles == 0) return RF_OK;
  hipLaunchKernelGGL(gemm_bf16_w4b_kernel, dim3(p.total_tiles), dim3(256), LDS, stream, p);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

static int launch_gemm_pp(GemmParams& p, hipStream_t stream) {
  constexpr int LDS_MAIN = 2 * 4 * 128 * 128, LDS_EPI = 8 * EPI_REGION;
  constexpr int LDS = LDS_EPI > LDS_MAIN ? LDS_EPI : LDS_MAIN;
  static bool attr_set = false;
  if (!attr_set) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_w8_pp16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp16e_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  layout_tiles<256, 256>(p);
  if (p.total_tiles == 0) return RF_OK;
  if (p.w8) hipLaunchKernelGGL(gemm_w8_pp16_kernel, dim3(p.total_tiles), dim3(512), LDS, stream, p);
  else hipLaunchKernelGGL(gemm_bf16_pp16e_kernel, dim3(p.total_tiles), dim3(512), LDS, stream, p);
  RF_LAUNCH_CHECK();
  return RF_OK;
}

// scratch layout shared by split-K and stream-K: [0, 4096) stream-K flags (zero outside a launch), partials after
constexpr int64_t WS_FLAG_BYTES = 4096;
static int g_last_path = 0;  // 0 = one tile per block, 1 = split-K, 2 = stream-K / persistent, 3 = skinny-N (read-only introspection)
// Persistent whole-tile schedule (RF_SCHED_PERSISTENT): worth +1..4 % with the round-1 main loop (a tile's stores drain under the
// next tile's loop; profiles/r02_kb_persist.log); with the balanced loop it is neutral at cfg2 and -1..3 % at cfg5 sizes
// (profiles/r02_kb_persist_pp2.log), so dispatch() never picks it on its own.

// stream-K mode of a launch: -1 = heuristic, 0 = never, 1 = whenever feasible, 2 = persistent whole tiles only
static int sk_mode(const GemmParams& p) {
  switch (p.sched) {
    case RF_SCHED_STREAMK: return 1;
    case RF_SCHED_PERSISTENT: return 2;
    case RF_SCHED_AUTO: return -1;
    default: return 0;
  }
}

// The stream-K launch relies on two properties HIP does not promise: lower-indexed blocks are dispatched first and
// block b runs on XCD b % 8 (observed on MI355X in SPX mode; a wrong guess about the XCD costs speed only, but a
// partition mode that breaks dispatch order could make a consumer wait on a block that is not resident).  It is
// therefore enabled only on the device it was verified on (gfx950 with 256 CUs = MI355X SPX) and can be switched
// off with RF_DISABLE_STREAMK=1; everything then runs one tile per block.
static bool streamk_allowed(int num_cus) {
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("RF_DISABLE_STREAMK");
    env = (e != nullptr && e[0] != '\0' && e[0] != '0') ? 1 : 0;
  }
  return env == 0 && num_cus == 256;
}

// stream-K work plan (pure host arithmetic; also reachable through rf_debug_sk_plan so the CPU tests can check its
// invariants without a GPU).  Needs layout_tiles() done.  Returns 1 if the launch qualifies, 0 otherwise.
// persistent_only: no stream-K region at all -- every worker walks whole tiles cl, cl + PL, ... of its XCD chunk (the
// one-tile-per-block schedule as ONE persistent launch: a tile's output stores drain while the next tile's main loop
// runs, and there is no block dispatch between rounds)
static int sk_make_plan(const GemmParams& p, const int P, SkParams& sk, const bool persistent_only = false) {
  const int T = p.total_tiles;
  memset(&sk, 0, sizeof(sk));
  int64_t I = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    sk.iter_start[g] = (int)I;
    sk.nk[g] = p.g[g].seg[0].nk + p.g[g].seg[1].nk + p.g[g].seg[2].nk;
    I += (int64_t)p.g[g].tiles_m * p.tiles_n * sk.nk[g];
  }
  if (I * 8 >= (1ll << 31)) return 0;
  for (int g = p.ngroups; g <= 4; ++g) sk.iter_start[g] = (int)I;
  auto iter_of_tile = [&](int t) {
    int g = 0;
    while (g + 1 < p.ngroups && t >= p.g[g + 1].tile_start) ++g;
    return sk.iter_start[g] + (t - p.g[g].tile_start) * sk.nk[g];
  };
  // per-XCD chunks: cut at the tile boundary nearest to x/8 of the work
  const int PL = P / 8;
  int chunk_tile[9];
  for (int x = 0; x <= 8; ++x) {
    const int64_t target = I * x / 8;
    int g = 0;
    while (g + 1 < p.ngroups && target >= sk.iter_start[g + 1]) ++g;
    const int64_t j = (target - sk.iter_start[g] + sk.nk[g] / 2) / sk.nk[g];
    chunk_tile[x] = p.g[g].tile_start + (int)j;
  }
  chunk_tile[0] = 0;
  chunk_tile[8] = T;
  for (int x = 0; x < 8; ++x) {
    const int Tc = chunk_tile[x + 1] - chunk_tile[x];
    sk.chunk_tile[x] = chunk_tile[x];
    sk.chunk_tiles[x] = Tc;
    sk.chunk_end[x] = chunk_tile[x + 1] == T ? (int)I : iter_of_tile(chunk_tile[x + 1]);
    if (persistent_only) {
      sk.dp_rounds[x] = cdiv(Tc, PL);
      sk.sk_begin[x] = sk.chunk_end[x];   // empty stream-K region
      continue;
    }
    // the stream-K region is the remainder after the full rounds (workers are out of K-phase there and share less
    // in the L2, so it is kept short); if that leaves < 8 K-tiles per worker, the last full round joins it
    sk.dp_rounds[x] = Tc / PL;
    if (sk.dp_rounds[x] > 0 && sk.chunk_end[x] - iter_of_tile(chunk_tile[x] + sk.dp_rounds[x] * PL) < PL * 8) --sk.dp_rounds[x];
    sk.sk_begin[x] = iter_of_tile(chunk_tile[x] + sk.dp_rounds[x] * PL);
    if (sk.chunk_end[x] - sk.sk_begin[x] < PL * 4) return 0;  // < 4 K-tiles per worker: not worth slicing
  }
  return 1;
}

// stream-K launch of the 256x256 kernel; returns 1 if it launched, 0 if the shape does not qualify, < 0 on error
template <int BM, int BN, int WM, int WN, bool W8>
static int try_launch_gemm_sk(GemmParams& p, float* ws, int64_t ws_bytes, hipStream_t stream) {
  static int num_cus = 0;
  if (num_cus == 0) {
    int dev = 0;
    RF_CHECK_HIP(hipGetDevice(&dev));
    RF_CHECK_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int force_sk = sk_mode(p);   // -1 = heuristic, 1 = whenever feasible, 2 = persistent whole tiles (0 never gets here)
  if (!streamk_allowed(num_cus) && force_sk != 1) return 0;
  const int P = num_cus / 8 * 8;  // one persistent block per CU, 8 XCD chunks
  if (P < 8 || (int64_t)P * 4 > WS_FLAG_BYTES) return 0;
  if (ws == nullptr || ws_bytes < WS_FLAG_BYTES + (int64_t)P * BM * BN * 4) return 0;
  layout_tiles<BM, BN>(p);
  const int T = p.total_tiles;
  if (T == 0) return 0;
  const int rounds = cdiv(T, P);
  // Measured on MI355X (tools/kb_sk.py, profiles/r01_stream_k.md): the chip is power-limited (~1.39 kW at 1.87 GHz
  // under this kernel), so a last round that leaves CUs idle costs less than its tile count suggests, while the
  // stream-K region loses the lock-step L2 sharing of A/W panels.  Stream-K wins below ~83 % round utilisation
  // (S=5632: 264 tiles +40..58 %, 792 tiles +9 %, 1056 tiles +5 %) and loses above it (S=4608: 0.84 -> -2..-10 %).
  const bool persistent_only = force_sk == 2 || (force_sk < 0 && !W8 && PERSISTENT_ROUNDS > 0 && rounds >= PERSISTENT_ROUNDS &&
                                                 (double)T / ((double)rounds * P) >= 0.83);
  if (force_sk < 0 && !persistent_only && (double)T / ((double)rounds * P) >= 0.83) return 0;
  SkParams sk;
  {
    const int ok = sk_make_plan(p, P, sk, persistent_only);
    if (ok != 1) return ok;
  }
  sk.flags = (int*)ws;
  sk.partials = (float*)((char*)ws + WS_FLAG_BYTES);
  constexpr int LDS_MAIN = 2 * (BM + BN) * 128, LDS_EPI = WM * WN * EPI_REGION;
  constexpr int LDS = LDS_EPI > LDS_MAIN ? LDS_EPI : LDS_MAIN;
  static bool attr_set = false;
  // bf16 launches: evenly loaded phases (gemm_mainloop_pp3_m16); mixed-precision launches: the 8/4/8/4 loop for every group
  auto kern16 = gemm_bf16_sk_kernel<BM, BN, WM, WN, W8, true, !W8>;
  if (!attr_set) {
    RF_CHECK_HIP(hipFuncSetAttribute((const void*)kern16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern16, dim3(P), dim3(WM * WN * 64), LDS, stream, p, sk);
  RF_LAUNCH_CHECK();
  g_last_path = 2;
  return 1;
}

// out[m][n] = bf16( sum_s ws[s][m][n] + bias[n] ), slices summed in index order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int64_t slice, int ws_ld, int ksplit,
                                                            const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                            int64_t ldo, int M, int N) {
  const int n8 = N >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * n8) return;
  const int m = (int)(idx / n8), n = (int)(idx % n8) * 8;
  const float* src = ws + (int64_t)m * ws_ld + n;
  f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
  for (int s2 = 1; s2 < ksplit; ++s2) {
    src += slice;
    lo += *(const f32x4*)src;
    hi += *(const f32x4*)(src + 4);
  }
  float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  if (bias != nullptr) {
    float b8[8];
    unpack8(*(const u32x4*)(bias + n), b8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += b8[e];
  }
  *(u32x4*)(out + (int64_t)m * ldo + n) = pack8(v);
}

static int build_params(const rf_gemm_desc* d, GemmParams& p, const bool w8 = false) {
  RF_REQUIRE(d != nullptr, RF_ERR_NULL, "rf_gemm_bf16: desc is NULL");
  RF_REQUIRE(d->N > 0, RF_ERR_SHAPE, "rf_gemm_bf16: N=%d", d->N);
  RF_REQUIRE(d->num_groups >= 1 && d->num_groups <= 4, RF_ERR_SHAPE, "rf_gemm_bf16: num_groups=%d", d->num_groups);
  RF_REQUIRE(d->epilogue >= RF_EPI_STORE && d->epilogue <= RF_EPI_QKV_GELU, RF_ERR_SHAPE, "rf_gemm_bf16: bad epilogue %d",
             d->epilogue);
  memset(&p, 0, sizeof(p));
  p.w8 = w8 ? 1 : 0;
  p.N = d->N; p.epi = d->epilogue; p.n_split = d->n_split;
  RF_REQUIRE(d->schedule >= RF_SCHED_AUTO && d->schedule <= RF_SCHED_W4B, RF_ERR_SHAPE, "rf_gemm: schedule=%d", d->schedule);
  p.sched = d->schedule;
  p.heads = d->heads; p.s_pad = d->s_pad;
  p.q = (bf16_t*)d->q; p.k = (bf16_t*)d->k; p.vt = (bf16_t*)d->vt;
  p.rope_cos = d->rope_cos; p.rope_sin = d->rope_sin; p.norm_eps = d->norm_eps;
  p.q_scale = d->q_scale == 0.f ? 1.0f : d->q_scale;
  const bool qkv = d->epilogue == RF_EPI_QKV || d->epilogue == RF_EPI_QKV_GELU;
  if (qkv) {
    RF_REQUIRE(d->q && d->k && d->vt, RF_ERR_NULL, "rf_gemm_bf16: QKV epilogue needs q,k,vt");
    RF_REQUIRE(d->heads > 0 && d->s_pad > 0 && d->s_pad % 64 == 0, RF_ERR_SHAPE, "rf_gemm_bf16: bad heads/s_pad");
    const int qkv_cols = d->epilogue == RF_EPI_QKV ? d->N : d->n_split;
    RF_REQUIRE(qkv_cols == 3 * d->heads * 128, RF_ERR_SHAPE, "rf_gemm_bf16: QKV columns %d != 3*heads*128", qkv_cols);
    if (d->epilogue == RF_EPI_QKV_GELU)
      RF_REQUIRE(d->n_split % 256 == 0 && d->n_split < d->N, RF_ERR_SHAPE, "rf_gemm_bf16: n_split must be tile aligned");
  }
  int ng = 0;
  for (int g = 0; g < d->num_groups; ++g) {
    const rf_gemm_group& s = d->g[g];
    if (s.M <= 0) continue;  // empty token group (e.g. no condition): skip
    GemmGroupDev& t = p.g[ng++];
    RF_REQUIRE(s.seg[0].K > 0, RF_ERR_SHAPE, "rf_gemm_bf16: group %d segment 0 is empty", g);
    // rf_gemm_w8a8: a group WITH a_scale has fp8 operands, a group without one is bf16 (mixed-precision launch)
    const bool g8 = w8 && s.a_scale != nullptr;
    const int ktile = g8 ? 128 : 64, esz = g8 ? 1 : 2;   // elements per 128-byte K-tile row, bytes per element
    int ns = 0;
    for (int k = 0; k < 3; ++k) {
      const rf_kseg& ks = s.seg[k];
      if (ks.K <= 0) continue;
      RF_REQUIRE(ks.K % ktile == 0, RF_ERR_SHAPE, "rf_gemm: group %d segment %d K=%d not a multiple of %d", g, k, ks.K, ktile);
      RF_REQUIRE(ks.A && ks.W, RF_ERR_NULL, "rf_gemm_bf16: group %d segment %d A/W NULL", g, k);
      RF_REQUIRE(aligned16(ks.A) && aligned16(ks.W) && ks.lda % (16 / esz) == 0 && ks.ldw % (16 / esz) == 0, RF_ERR_ALIGN,
                 "rf_gemm: group %d segment %d operands must be 16-byte aligned", g, k);
      // the LDS-DMA addresses operands as buffer resource (num_records 2^31-1) + 32-bit byte offset: the last byte any
      // lane can touch must stay inside that window, or the hardware range check silently returns zeros
      RF_REQUIRE(((int64_t)s.M - 1) * ks.lda * esz + (int64_t)ks.K * esz < 0x7fffffffll &&
                     ((int64_t)d->N - 1) * ks.ldw * esz + (int64_t)ks.K * esz < 0x7fffffffll,
                 RF_ERR_SHAPE, "rf_gemm_bf16: group %d segment %d operand spans >= 2 GiB (32-bit buffer offsets)", g, k);
      KSegDev& kd = t.seg[ns++];  // compact: empty segments are dropped
      kd.A = (const bf16_t*)ks.A; kd.lda = ks.lda; kd.W = (const bf16_t*)ks.W; kd.ldw = ks.ldw; kd.nk = ks.K / ktile;
    }
    t.bias = (const bf16_t*)s.bias;
    t.w8 = g8 ? 1 : 0;
    t.a_scale = g8 ? s.a_scale : nullptr; t.w_scale = g8 ? s.w_scale : nullptr;
    if (g8)
      RF_REQUIRE(s.w_scale != nullptr && aligned16(s.w_scale), RF_ERR_NULL,
                 "rf_gemm_w8a8: group %d has a_scale [M] but no 16-byte aligned w_scale [N]", g);
    if (!w8) RF_REQUIRE(s.a_scale == nullptr, RF_ERR_UNSUPPORTED, "rf_gemm_bf16: group %d carries a_scale -- fp8 groups need rf_gemm_w8a8", g);
    t.M = s.M; t.tok_offset = s.tok_offset;
    t.out = (bf16_t*)s.out; t.ldo = s.ldo;
    t.residual = (const bf16_t*)s.residual; t.ldr = s.ldr; t.gate = (const bf16_t*)s.gate;
    t.norm_q = (const bf16_t*)s.norm_q; t.norm_k = (const bf16_t*)s.norm_k;
    if (qkv && d->rope_cos != nullptr)
      RF_REQUIRE(s.norm_q && s.norm_k && aligned16(s.norm_q) && aligned16(s.norm_k), RF_ERR_NULL,
                 "rf_gemm_bf16: fused rope needs 16-byte aligned norm_q/norm_k in group %d", g);
    if (d->epilogue != RF_EPI_QKV) RF_REQUIRE(s.out != nullptr, RF_ERR_NULL, "rf_gemm_bf16: group %d out NULL", g);
    if (d->epilogue == RF_EPI_GATE_RES) RF_REQUIRE(s.gate != nullptr, RF_ERR_NULL, "rf_gemm_bf16: GATE_RES needs gate");
    if (qkv) RF_REQUIRE(s.tok_offset >= 0 && s.tok_offset + s.M <= d->s_pad, RF_ERR_SHAPE, "rf_gemm_bf16: tokens exceed s_pad");
  }
  p.ngroups = ng;
  bool vec = (d->N % 8 == 0) && (!qkv || (aligned16(d->q) && aligned16(d->k) && aligned16(d->vt)));
  if (d->epilogue == RF_EPI_QKV_GELU) vec = vec && ((d->N - d->n_split) % 8 == 0);
  for (int g = 0; g < ng; ++g) {
    const GemmGroupDev& t = p.g[g];
    vec = vec && (t.bias == nullptr || aligned16(t.bias)) && (t.gate == nullptr || aligned16(t.gate)) &&
          (t.out == nullptr || (aligned16(t.out) && t.ldo % 8 == 0)) &&
          (t.residual == nullptr || (aligned16(t.residual) && t.ldr % 8 == 0));
  }
  p.vec_ok = vec ? 1 : 0;
  p.probe = (d->clock_probe != 0 || prof_open()) ? 1 : 0;
  if (w8) {
    RF_REQUIRE(vec, RF_ERR_ALIGN, "rf_gemm_w8a8: needs N %% 8 == 0 and 16-byte aligned outputs / bias / gate / residual");
    bool any8 = false;
    for (int g = 0; g < ng; ++g) any8 = any8 || p.g[g].w8;
    if (!any8) p.w8 = 0;   // all groups bf16: the plain bf16 kernels
  }
  p.ksplit = 1;
  if (d->splitk_ws != nullptr) {
    RF_REQUIRE(aligned16(d->splitk_ws) && d->splitk_ws_bytes >= 0, RF_ERR_ALIGN, "rf_gemm_bf16: splitk_ws must be 16-byte aligned");
    p.scratch = (float*)d->splitk_ws;
    p.scratch_bytes = d->splitk_ws_bytes;
  }
  if (d->rope_cos != nullptr) {
    RF_REQUIRE(qkv && d->rope_sin != nullptr && aligned16(d->rope_cos) && aligned16(d->rope_sin), RF_ERR_ALIGN,
               "rf_gemm_bf16: rope tables need a QKV epilogue and 16-byte alignment");
    RF_REQUIRE(vec, RF_ERR_ALIGN, "rf_gemm_bf16: fused RMSNorm+RoPE needs the 16-byte aligned epilogue path");
  }
  return RF_OK;
}

static int dispatch(GemmParams& p, hipStream_t stream) {
  if (p.ngroups == 0) return RF_OK;
  int64_t rows = 0;
  for (int g = 0; g < p.ngroups; ++g) rows += p.g[g].M;
  // tile: 128 / 256 = the shipped kernels, 257 = plain 256x256 loop (bit-exact reference of the ping-pong loops);
  // 258 / 259 = experiments build only
  int tile = 0;
  bool sk_or_128 = false;   // AUTO chose 256^2 tiles for their stream-K form only: without it (no scratch / not this device) use 128^2
  if (tile == 0) {
    switch (p.sched) {
      case RF_SCHED_TILE128: tile = 128; break;
      case RF_SCHED_TILE256: case RF_SCHED_STREAMK: case RF_SCHED_PERSISTENT: tile = 256; break;
      case RF_SCHED_PLAIN256: tile = 257; break;
      case RF_SCHED_W4: tile = 260; break;
      case RF_SCHED_W4B: tile = 261; break;
      default: {
        // 256^2 tiles pay from ~half a round of the 256 CUs (plain or stream-K: 144-192 tiles measured 15-30 % ahead of 128^2,
        // tools/kb_gemm_midsize.py / profiles/r04_gemm_midsize.md); below that only as stream-K and only with a long K
        // (>= 128 K-tiles: 48-120 tiles at K = 12288 / 15360 are 4-24 % ahead, at K = 3072 they lose 15-40 %).
        const int64_t t256 = (int64_t)cdiv((int)rows, 256) * cdiv(p.N, 256);
        int nk_min = 1 << 30;
        for (int g = 0; g < p.ngroups; ++g) nk_min = std::min(nk_min, p.g[g].seg[0].nk + p.g[g].seg[1].nk + p.g[g].seg[2].nk);
        if (t256 >= 128) tile = 256;
        else if (t256 >= 32 && nk_min >= 128 && !p.w8 && p.vec_ok) tile = 256, sk_or_128 = true;
        else tile = 128;
      }
    }
  }
  double flops = 0.0;  // algorithmic: 2 M N K over groups and K-segments
  for (int g = 0; g < p.ngroups; ++g)
    flops += 2.0 * p.g[g].M * (double)p.N * (p.g[g].w8 ? 128.0 : 64.0) * (p.g[g].seg[0].nk + p.g[g].seg[1].nk + p.g[g].seg[2].nk);
  ProfScope prof(p.w8 ? RF_KC_GEMM_W8 : (tile >= 256 ? RF_KC_GEMM_MAIN : RF_KC_GEMM_SMALL), flops, stream);
  // split-K: a plain-store, single-group GEMM with a handful of tiles and a long K (LoRA down-projection:
  // [S_cond x K] . [r_pad x K]^T = 8 tiles x up to 240 K-tiles) would run on 8 of 256 CUs.  Slice K over
  // blockIdx.y, >= 4 K-tiles per slice, ~256 blocks in flight, fp32 partials in caller-owned scratch.
  float* const ws_base = p.scratch;
  const int64_t ws_total = ws_base != nullptr ? p.scratch_bytes : 0;
  const int64_t ws_bytes = ws_total > WS_FLAG_BYTES ? ws_total - WS_FLAG_BYTES : 0;
  p.ws = ws_base != nullptr ? (float*)((char*)ws_base + WS_FLAG_BYTES) : nullptr;
  p.ksplit = 1; p.ws_slice = 0;
  if (tile == 260 || tile == 261)
    RF_REQUIRE(p.vec_ok && !p.w8, RF_ERR_UNSUPPORTED, "rf_gemm: RF_SCHED_W4 / W4B need bf16 operands and the 16-byte aligned epilogue path");
  if (tile >= 257 && (!p.vec_ok || p.w8)) tile = 256;
  if (p.w8) tile = 256;  // fp8 operands: only the 256x256 ping-pong / stream-K kernels exist (build_params checked vec_ok)
  if (tile == 256 && p.vec_ok && sk_mode(p) != 0) {   // (the plain reference loop / experimental kernels skip the stream-K paths)
    const int rc = p.w8 ? try_launch_gemm_sk<256, 256, 4, 2, true>(p, ws_base, ws_total, stream)
                        : try_launch_gemm_sk<256, 256, 4, 2, false>(p, ws_base, ws_total, stream);
    if (rc != 0) return rc < 0 ? rc : RF_OK;
  }
  if (sk_or_128) {  // AUTO sized it at 256^2 for stream-K only; stream-K declined: it runs on 128^2 tiles and is booked as such
    tile = 128;
    prof.reclass(RF_KC_GEMM_SMALL);
  }
  if (ws_bytes > 0 && tile == 128 && p.ngroups == 1 && p.epi == RF_EPI_STORE && p.vec_ok) {
    const GemmGroupDev& G = p.g[0];
    const int nk = G.seg[0].nk + G.seg[1].nk + G.seg[2].nk;
    const int tm = cdiv(G.M, 128), tn = cdiv(p.N, 128), tiles = tm * tn;
    const int64_t slice = (int64_t)G.M * tn * 128;  // floats
    int ks = nk / 4 < cdiv(256, tiles) ? nk / 4 : cdiv(256, tiles);
    if (slice * 4 * ks > ws_bytes) ks = (int)(ws_bytes / (slice * 4));
    if (tiles <= 64 && ks >= 2) {
      p.kchunk = cdiv(nk, ks);
      p.ksplit = cdiv(nk, p.kchunk);
      p.ws_slice = slice;
      p.ws_ld = tn * 128;
      int rc = launch_gemm<128, 128, 4, 1, true>(p, stream);
      if (rc != RF_OK) return rc;
      const int64_t items = (int64_t)G.M * (p.N >> 3);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, stream, p.ws, slice, p.ws_ld,
                         p.ksplit, G.bias, G.out, G.ldo, G.M, p.N);
      RF_LAUNCH_CHECK();
      g_last_path = 1;
      return RF_OK;
    }
  }
  g_last_path = 0;
  // wave tiles are (BM/WM) x 128: a wave always owns whole 128-wide head rows / 256-byte output runs
  if (tile == 257) return launch_gemm<256, 256, 4, 2, true>(p, stream);  // plain loop (bit-exact reference)
  if (tile == 260) return launch_gemm_w4m16(p, stream);                   // one wave per SIMD, 128 x 128 wave tiles
  if (tile == 261) return launch_gemm_w4b(p, stream);                     // ... with three half-stage barriers per K-tile
  // (AUTO never picks RF_SCHED_W4B: with warm operands it is 0..4 % ahead of the 8-wave loop on plain-store launches and 8 % on the
  //  K = 15360 single-block projection, but inside the 57-block sequence, where every weight panel arrives HBM-cold, the same
  //  launches are 0.5..10 % SLOWER -- one wave per SIMD has nothing to run while a late LDS-DMA piece holds barrier 3;
  //  profiles/r05_gemm_w4b.md)
  if (tile == 256) return p.vec_ok ? launch_gemm_pp(p, stream) : launch_gemm<256, 256, 4, 2, false>(p, stream);
  return p.vec_ok ? launch_gemm<128, 128, 4, 1, true>(p, stream) : launch_gemm<128, 128, 4, 1, false>(p, stream);
}

}  // namespace rf

extern "C" int rf_gemm_bf16(const rf_gemm_desc* d, void* stream) {
  rf::GemmParams p;
  int rc = rf::build_params(d, p);
  if (rc != RF_OK) return rc;
  return rf::dispatch(p, (hipStream_t)stream);
}

extern "C" int rf_gemm_w8a8(const rf_gemm_desc* d, void* stream) {
  rf::GemmParams p;
  int rc = rf::build_params(d, p, /*w8=*/true);
  if (rc != RF_OK) return rc;
  return rf::dispatch(p, (hipStream_t)stream);
}

// ---- read-only introspection (tests, bench): no entry point below changes what the library runs ----------------------------
// CPU tests: the stream-K plan of a launch for a chip with num_cus CUs (no device access).
// out[0..4] iter_start, [5..8] nk, [9..16] chunk_tile, [17..24] dp_rounds, [25..32] sk_begin, [33..40] chunk_end,
// [41] tiles_n, [42..45] tiles_m per group, [46..49] tile_start per group, [50] total tiles.  Returns 1 = plan made,
// 0 = launch does not qualify, < 0 error.
extern "C" int rf_debug_sk_plan(const rf_gemm_desc* d, int32_t num_cus, int32_t* out) {
  rf::GemmParams p;
  int rc = rf::build_params(d, p);
  if (rc != RF_OK) return rc;
  RF_REQUIRE(out != nullptr && num_cus >= 8, RF_ERR_NULL, "rf_debug_sk_plan: bad arguments");
  rf::layout_tiles<256, 256>(p);
  if (p.total_tiles == 0) return 0;
  rf::SkParams sk;
  const int ok = rf::sk_make_plan(p, num_cus / 8 * 8, sk);
  if (ok != 1) return ok;
  for (int i = 0; i < 5; ++i) out[i] = sk.iter_start[i];
  for (int i = 0; i < 4; ++i) out[5 + i] = sk.nk[i];
  for (int i = 0; i < 8; ++i) { out[9 + i] = sk.chunk_tile[i]; out[17 + i] = sk.dp_rounds[i]; out[25 + i] = sk.sk_begin[i]; out[33 + i] = sk.chunk_end[i]; }
  out[41] = p.tiles_n;
  for (int i = 0; i < 4; ++i) { out[42 + i] = i < p.ngroups ? p.g[i].tiles_m : 0; out[46 + i] = i < p.ngroups ? p.g[i].tile_start : 0; }
  out[50] = p.total_tiles;
  return 1;
}

// which schedule the last GEMM launch took: 0 = one tile per block, 1 = split-K, 2 = stream-K / persistent (3 = skinny-N, experiments build)
extern "C" int rf_debug_last_gemm_path(void) { return rf::g_last_path; }

namespace rf {
int read_clk_probe_gemm(unsigned long long* h) {
  return hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_probe), 4 * sizeof(unsigned long long)) == hipSuccess ? RF_OK : RF_ERR_HIP;
}
}  // namespace rf
// which: 0 = the last completed 256x256 ping-pong GEMM launch (one tile per block), 1 = the last bounded-score attention
// launch.  mhz = shader clocks per microsecond over block 0's main loop, us = that loop's duration.  The caller
// synchronises the stream first; single-stream use only (the probe words are written by block 0 of the last launch).
extern "C" int rf_debug_clock_probe(int which, double* mhz, double* us) {
  unsigned long long h[4];
  if (which == 2) {   // block 0 / wave 0 of the last 16x16x32 GEMM launch: end of its main loop -> its epilogue stores acknowledged
    unsigned long long e[2];
    if (rf::read_clk_probe_gemm(h) != RF_OK) return RF_ERR_HIP;
    if (hipMemcpyFromSymbol(e, HIP_SYMBOL(rf::g_clk_probe_epi), sizeof(e)) != hipSuccess) return RF_ERR_HIP;
    *us = (double)(e[1] - h[3]) / 100.0;
    *mhz = *us > 0 ? (double)(e[0] - h[2]) / *us : 0.0;
    return RF_OK;
  }
  const int rc = which == 0 ? rf::read_clk_probe_gemm(h) : rf::read_clk_probe_attn(h);
  if (rc != RF_OK) return rc;
  *us = (double)(h[3] - h[1]) / 100.0;
  *mhz = *us > 0 ? (double)(h[2] - h[0]) / *us : 0.0;
  return RF_OK;
}


static int time_gemm_impl(const rf_gemm_desc* d, int32_t iters, float* us, void* stream, bool w8);
extern "C" int rf_time_gemm(const rf_gemm_desc* d, int32_t iters, float* us, void* stream) { return time_gemm_impl(d, iters, us, stream, false); }
extern "C" int rf_time_gemm_w8a8(const rf_gemm_desc* d, int32_t iters, float* us, void* stream) { return time_gemm_impl(d, iters, us, stream, true); }
static int time_gemm_impl(const rf_gemm_desc* d, int32_t iters, float* us, void* stream, bool w8) {
  rf::GemmParams p;
  int rc = rf::build_params(d, p, w8);
  if (rc != RF_OK) return rc;
  p.probe = 1;   // the timing hook is bench instrumentation: it wants the in-kernel clock of what it times
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  RF_CHECK_HIP(hipEventCreate(&e0));
  RF_CHECK_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) {
    rc = rf::dispatch(p, s);
    if (rc != RF_OK) return rc;
  }
  RF_CHECK_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) rf::dispatch(p, s);
  RF_CHECK_HIP(hipEventRecord(e1, s));
  RF_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  RF_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us = ms * 1000.f / (float)iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return RF_OK;
}
