// One wave per SIMD, second form (round 5): 256 x 256 x 64 block tile, 4 waves (2 x 2), 128 x 128 wave tiles on
// v_mfma_f32_16x16x32_bf16, with
//   (1) the K-tile SCHEDULE of the fastest 256 x 256 x 64 kernel this part has been seen to run (hipBLASLt's hand-written
//       `Custom_Cijk_..._SK3_MT256x256x64_MI16x16x1`, disassembled: profiles/r05_gemm_w4b.md), and
//   (2) an LDS image whose LDS-DMA source pattern is LANE-LINEAR.
// (included by gemm_bf16.hip after gemm_w4.hpp; namespace rf)
//
// (2) is the larger half.  Every other loop of this file keeps its LDS image conflict-free by XOR-swizzling the 16-byte chunks of
// a 128-byte row, and since LDS-DMA writes lane-linearly the swizzle sits on the SOURCE: lane l of a piece fetches chunk
// (l % 8) ^ f(row).  Knock-outs of this loop (profiles/r05_gemm_w4b.md): no LDS-DMA +17 %, no fragment reads +11 %, no barriers /
// no vmcnt wait +0..1.5 % -- and the same loop fetching chunk l % 8 (wrong results, same bytes, same instructions) +9..11 %.  A
// `buffer_load_dwordx4 ... lds` whose lanes do not walk memory in ascending order inside a 64-byte quad costs the issuing wave
// several times the issue slots, and with one wave per SIMD nothing else fills them.
//     Image: the 256 rows of an operand's K-tile are 32 "wave pieces" of 8 rows x 128 B = 1 KiB, each written by ONE LDS-DMA
//     instruction (lane l -> row l / 8, chunk l % 8: ascending addresses), piece p at byte p * 1056 (32 B of padding).
//     MFMA tile j (0..7) of a wave's 128-row strip is NOT 16 consecutive rows but the rows {8 r + j : r = 0..15} -- row j of 16
//     consecutive pieces -- so lane (r, g) of a fragment read sits at (p0 + r) * 1056 + j * 128 + (4 s + g) * 16: one base register,
//     immediates j * 128 + s * 64.  1056 / 16 = 66 = 2 (mod 16): the 16 lanes of every ds_read_b128 lane group land in 16 distinct
//     16-byte bank slots (tests/test_lds_layouts_cpu.py holds the lane groups and this image).
//     Which rows form an MFMA tile is free: the accumulator of lane (l15, g), register e, tiles (it, jt) is element
//         row 8 (4 g + e) + it,  column 8 l15 + jt
//     of the wave tile -- a lane OWNS 8 CONSECUTIVE COLUMNS (jt = 0..7) of 32 rows, which is exactly what the LDS-staged epilogue
//     of the other loops transposes the accumulators into.  This loop's epilogue is register-direct: no LDS round trip, 16-byte
//     stores, bias / gate / norm weights loaded once per lane, per-head RMSNorm over the 16 lanes of a row by the same butterfly
//     (same partial sums in the same order: bit-identical to the staged epilogue), V^T as 8-byte runs along the token axis (it).
//
// (1) What round 4's RF_SCHED_W4 (gemm_w4.hpp) did not have, read off the library's loop:
//   * THREE barriers per K-tile instead of one, each releasing HALF a stage: the 8 fragment reads of the cycling operand's second
//     k-step come first (one per 2 MFMAs), barrier 1 at MFMA ~20 hands the W image of the tile being multiplied to the LDS-DMA;
//     the other operand's 8 reads follow, barrier 2 at MFMA ~52 hands over the A image; barrier 3 (MFMA ~92, `vmcnt(12)`: the DMA
//     queue is never drained) publishes the NEXT tile, whose first-k-step fragments are read in the last third of the tile.
//   * the 16 LDS-DMA pieces of tile t+2 are therefore spread over MFMAs ~22..122 (one per ~6 MFMAs) instead of one per 4 MFMAs
//     inside the second k-step only, and every piece has >= ~96 MFMAs (1.5 k cycles) between issue and the barrier that
//     publishes it (W4: 64).  In the accounting of DESIGN K1: window + flight <= 200 MFMAs of the 256 two stages allow (W4: 128).
// Same MFMAs in the same order as RF_SCHED_W4 / the 8-wave loop (k-step 0 then 1, K-tiles in order): bit-identical results.
#pragma once
// what happens in front of MFMA m of a K-tile (m = 0..127; MFMA m multiplies A[(m / 8) % 8] with W[m % 8] of k-step m / 64)
struct W4bPlan {
  signed char read[128];   // -1, or fragment to read: 0..7 W k-step 1, 8..15 A k-step 1 (this tile); 16..23 W k-step 0, 24..31 A k-step 0 (next tile)
  signed char dma[128];    // -1, or piece of tile t+2: 0..7 W image, 8..15 A image
  signed char bar[128];    // 0, 1 = lgkmcnt(0) + barrier, 2 = vmcnt(pieces issued so far this tile) + lgkmcnt(0) + barrier
};
constexpr W4bPlan w4b_plan() {
  W4bPlan p{};
  for (int m = 0; m < 128; ++m) p.read[m] = -1, p.dma[m] = -1, p.bar[m] = 0;
  for (int i = 0; i < 8; ++i) p.read[1 + 2 * i] = (signed char)i;            // W k-step 1: MFMAs 1..15
  p.bar[20] = 1;                                                             // W image of this tile is free
  for (int i = 0; i < 8; ++i) p.read[26 + 3 * i] = (signed char)(8 + i);     // A k-step 1: MFMAs 26..47 (A1[i] is needed at MFMA 64 + 8 i)
  p.bar[52] = 1;                                                             // A image of this tile is free
  const int wp[8] = {22, 28, 34, 40, 46, 54, 60, 66};                        // W pieces (22..46 before barrier 2 interleave with the A reads)
  const int ap[8] = {72, 78, 84, 90, 98, 106, 114, 122};                     // A pieces
  for (int i = 0; i < 8; ++i) p.dma[wp[i]] = (signed char)i;
  for (int i = 0; i < 8; ++i) p.dma[ap[i]] = (signed char)(8 + i);
  p.bar[93] = 2;                                                             // the next tile has landed for every wave (12 pieces of this tile in flight)
  for (int i = 0; i < 8; ++i) p.read[94 + 2 * i] = (signed char)(16 + i);    // next tile, W k-step 0: 94..108 (W0 registers are free after MFMA 63)
  for (int i = 0; i < 8; ++i) p.read[110 + 2 * i] = (signed char)(24 + i);   // next tile, A k-step 0: 110..124 (A0[i] is needed at MFMA 8 i of the next tile)
  return p;
}

__device__ __forceinline__ void gemm_mainloop_w4b(const GemmGroupDev& G, const int N, const int m0, const int n0, const int nk,
                                                  f32x4 (&acc)[8][8], char* smem, const int w, const int lane) {
  constexpr int PIECE = 1024 + 32;   // one wave piece (8 rows x 128 B) + padding: 66 sixteen-byte slots, = 2 (mod 16)
  constexpr int IMG = 32 * PIECE;    // one operand image of a K-tile (256 rows)
  constexpr int STAGE = 2 * IMG;     // A image | W image
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, g4 = lane >> 4;
  // per-lane fragment bases inside a stage: k-step 0 (tile j adds the immediate j * 128, k-step 1 the immediate 64)
  const int fa0 = (wm * 16 + l15) * PIECE + g4 * 16, fa1 = fa0 + 64;
  const int fb0 = IMG + (wn * 16 + l15) * PIECE + g4 * 16, fb1 = fb0 + 64;
  // LDS-DMA geometry: piece j (0..7) of wave w is wave piece 4 j + w of an image = rows 32 j + 8 w + lane / 8, chunk lane % 8
  const int r8 = lane >> 3;
  const uint32_t chunk_b = (uint32_t)((lane & 7) * 16);
  struct Cur { int seg, kk, nk; rsrc_t A, W; uint32_t voA, voB, stA, stB; };
  auto load_seg = [&](Cur& c) {
    const KSegDev& S = G.seg[c.seg];
    const uint32_t lda2 = (uint32_t)(S.lda * 2), ldw2 = (uint32_t)(S.ldw * 2);
    c.nk = S.nk;
    c.A = RF_MAKE_RSRC_N(S.A, (int64_t)(G.M - 1) * lda2 + (int64_t)S.nk * 128);
    c.W = RF_MAKE_RSRC_N(S.W, (int64_t)(N - 1) * ldw2 + (int64_t)S.nk * 128);
    c.voA = (uint32_t)(m0 + 8 * w + r8) * lda2 + chunk_b;
    c.voB = (uint32_t)(n0 + 8 * w + r8) * ldw2 + chunk_b;
    c.stA = 32 * lda2;
    c.stB = 32 * ldw2;
  };
  auto next = [&](Cur& c) {
    ++c.kk;
    if (__builtin_expect(c.kk >= c.nk && c.seg < 2 && G.seg[c.seg + 1].nk > 0, 0)) {
      c.kk = 0;
      ++c.seg;
      load_seg(c);
    }
  };
  // piece g of the tile under cursor c into stage `st`: g < 8 -> W piece g (the image barrier 1 frees), else A piece g - 8
  auto piece = [&](const Cur& c, const int st, const int g) {
    char* dst = smem + st * STAGE + (g < 8 ? IMG : 0) + ((g & 7) * 4 + w) * PIECE;
    if (g < 8) RF_BUF_LOAD_LDS(c.W, (lds_void*)dst, c.voB, c.kk * 128 + (g & 7) * c.stB);
    else RF_BUF_LOAD_LDS(c.A, (lds_void*)dst, c.voA, c.kk * 128 + (g & 7) * c.stA);
  };

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Cur c;
  c.seg = 0; c.kk = 0;
  load_seg(c);
#pragma unroll
  for (int g = 0; g < 16; ++g) piece(c, 0, g);
  next(c);
  if (nk > 1) {
#pragma unroll
    for (int g = 0; g < 16; ++g) piece(c, 1, g);
    next(c);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 A0[8], B0[8], A1[8], B1[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    B0[g] = *(const bf16x8*)(smem + fb0 + g * 128);
    A0[g] = *(const bf16x8*)(smem + fa0 + g * 128);
  }

#if defined(__HIP_DEVICE_COMPILE__)
#define RF_W4B_MFMA(C, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(A_), "v"(B_))
#else
#define RF_W4B_MFMA(C, A_, B_) ((void)0)
#endif
  constexpr W4bPlan P = w4b_plan();
  // pieces of a tile issued in front of barrier 3: its `vmcnt` leaves exactly those in flight, i.e. waits for all of the previous tile's
  constexpr int BEFORE3 = [] { int n = 0, b = 0; for (int m = 0; m < 128; ++m) { if (w4b_plan().bar[m] == 2) b = 1; if (!b) n += w4b_plan().dma[m] >= 0; } return n; }();
  // one K-tile.  DMA: tile t+2 exists and is staged into this tile's stage; NEXT: tile t+1 exists (its k-step-0 fragments are read);
  // TAIL_WAIT: barrier 3 must drain the queue (nothing is issued in this tile, tile t+1's last pieces are the youngest in flight)
  // slot m of a K-tile: what the plan puts in front of MFMA m, then MFMA m.  Everything about the slot is a constant expression
  // (`if constexpr`), so the body is straight-line code with statically indexed fragment registers.
#define RF_W4B_SLOT(m)                                                                                       \
  {                                                                                                          \
    constexpr int r_ = P.read[(m)], d_ = P.dma[(m)], b_ = P.bar[(m)];                                        \
    if constexpr (b_ == 1) {                                             \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
      __builtin_amdgcn_s_barrier();                                                                          \
    } else if constexpr (b_ == 2 && NEXT) {                                                \
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BEFORE3) : "memory"); \
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                       \
      __builtin_amdgcn_s_barrier();                                                                          \
    }                                                                                                        \
    if constexpr (r_ >= 0 && r_ < 8) B1[r_ & 7] = *(const bf16x8*)(cur + fb1 + (r_ & 7) * 128);             \
    else if constexpr (r_ >= 8 && r_ < 16) A1[r_ & 7] = *(const bf16x8*)(cur + fa1 + (r_ & 7) * 128);       \
    else if constexpr (r_ >= 16 && r_ < 24 && NEXT) B0[r_ & 7] = *(const bf16x8*)(oth + fb0 + (r_ & 7) * 128); \
    else if constexpr (r_ >= 24 && NEXT) A0[r_ & 7] = *(const bf16x8*)(oth + fa0 + (r_ & 7) * 128);         \
    if constexpr (DMA && d_ >= 0) piece(c, dstage, d_ & 15);                                                 \
    if constexpr ((m) < 64) RF_W4B_MFMA(acc[((m) >> 3) & 7][(m) & 7], A0[((m) >> 3) & 7], B0[(m) & 7]);      \
    else RF_W4B_MFMA(acc[((m) >> 3) & 7][(m) & 7], A1[((m) >> 3) & 7], B1[(m) & 7]);                         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
#define RF_W4B_S2(m) RF_W4B_SLOT(m) RF_W4B_SLOT((m) + 1)
#define RF_W4B_S4(m) RF_W4B_S2(m) RF_W4B_S2((m) + 2)
#define RF_W4B_S8(m) RF_W4B_S4(m) RF_W4B_S4((m) + 4)
#define RF_W4B_S16(m) RF_W4B_S8(m) RF_W4B_S8((m) + 8)
#define RF_W4B_S32(m) RF_W4B_S16(m) RF_W4B_S16((m) + 16)
  // one K-tile.  DMA: tile t+2 exists and is staged into this tile's stage; NEXT: tile t+1 exists (its k-step-0 fragments are read)
  auto tile = [&](auto dma_c, auto next_c, const char* cur, const char* oth, const int dstage) {
    constexpr bool DMA = decltype(dma_c)::value, NEXT = decltype(next_c)::value;
    RF_W4B_S32(0) RF_W4B_S32(32) RF_W4B_S32(64) RF_W4B_S32(96)
  };
#undef RF_W4B_S32
#undef RF_W4B_S16
#undef RF_W4B_S8
#undef RF_W4B_S4
#undef RF_W4B_S2
#undef RF_W4B_SLOT
  using T_ = std::true_type;
  using F_ = std::false_type;
  int t = 0;
  for (; t + 2 < nk; ++t) {      // steady state: tiles t+1 and t+2 exist
    tile(T_{}, T_{}, smem + (t & 1) * STAGE, smem + ((t + 1) & 1) * STAGE, t & 1);
    next(c);
  }
  if (t + 1 < nk) {               // second to last tile: nothing left to stage
    tile(F_{}, T_{}, smem + (t & 1) * STAGE, smem + ((t + 1) & 1) * STAGE, 0);
    ++t;
  }
  tile(F_{}, F_{}, smem + (t & 1) * STAGE, smem, 0);   // last tile
#undef RF_W4B_MFMA
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have left the matrix pipe before the epilogue reads the AGPRs
}

// Register-direct epilogue of the strided tile mapping (file header): lane (l15, g) holds, in acc[it][jt][e], element
//   row 8 (4 g + e) + it, column 8 l15 + jt   of its wave's 128 x 128 tile (wave tile origin: rows wrow0, columns wcol0 of the block tile).
// Arithmetic, operation order and the RMSNorm reduction tree are those of gemm_epilogue_lds_v (bit-identical outputs).
__device__ __forceinline__ void gemm_epilogue_direct(const GemmParams& p, const GemmGroupDev& G, const f32x4 (&acc)[8][8], const int m0, const int n0,
                                                     const int wrow0, const int wcol0, const int lane) {
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
  const int ncol0 = n0 + wcol0;  // first column of this wave's 128-column strip
  if (ncol0 >= N) return;
  int which = 0, head = 0;
  if (epi == RF_EPI_QKV) {
    const int DH = p.heads * 128;
    which = ncol0 / DH;
    head = (ncol0 - which * DH) >> 7;
  }
  const int l15 = lane & 15, g = lane >> 4;
  const int q8 = l15 * 8;
  const int n = ncol0 + q8;
  const bool nok = n < N;  // N % 8 == 0 on this path
  const int mrow0 = m0 + wrow0 + 32 * g;   // + 8 e + it

  if (epi == RF_EPI_QKV && which == 2) {
    // V^T tiles: [head][tok/64][d][64], key position has bits 2,3 swapped; this lane holds, per column and e, the 8 consecutive
    // tokens it = 0..7: two 4-key runs
    if (!nok) return;
    float bias8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = 0.f;
    if (G.bias != nullptr) unpack8(*(const u32x4*)(G.bias + n), bias8);
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      bf16_t* dst = p.vt + (int64_t)head * (p.s_pad >> 6) * (128 * 64) + (q8 + jt) * 64;
      const float bias_v = bias8[jt];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
          const int m = mrow0 + 8 * e + 4 * h4;
          const float v[4] = {acc[4 * h4 + 0][jt][e], acc[4 * h4 + 1][jt][e], acc[4 * h4 + 2][jt][e], acc[4 * h4 + 3][jt][e]};
          const int tok = G.tok_offset + m;
          if (((tok & 3) == 0) && (m + 3 < M)) {
            const int pos = (tok & 51) | ((tok & 4) << 1) | ((tok & 8) >> 1);
            u32x2 o;
            o[0] = pack2(v[0] + bias_v, v[1] + bias_v);
            o[1] = pack2(v[2] + bias_v, v[3] + bias_v);
            *(u32x2*)(dst + (int64_t)(tok >> 6) * (128 * 64) + pos) = o;
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int t2 = tok + x;
              if (m + x < M) {
                const int pos = (t2 & 51) | ((t2 & 4) << 1) | ((t2 & 8) >> 1);
                dst[(int64_t)(t2 >> 6) * (128 * 64) + pos] = f2bf(v[x] + bias_v);
              }
            }
          }
        }
    }
    return;
  }

  float bias8[8], gate8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias8[j] = 0.f, gate8[j] = 0.f;
  if (nok && G.bias != nullptr) unpack8(*(const u32x4*)(G.bias + n), bias8);
  if (nok && epi == RF_EPI_GATE_RES) unpack8(*(const u32x4*)(G.gate + n), gate8);
  const bool fuse_rope = (epi == RF_EPI_QKV) && (p.rope_cos != nullptr);
  float nw8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) nw8[j] = 1.f;
  if (fuse_rope) unpack8(*(const u32x4*)((which == 0 ? G.norm_q : G.norm_k) + q8), nw8);

#pragma unroll
  for (int e = 0; e < 4; ++e) {
    u32x4 resv[8];
    if (epi == RF_EPI_GATE_RES && G.residual != nullptr) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = mrow0 + 8 * e + it;
        if (m < M && nok) resv[it] = *(const u32x4*)(G.residual + (int64_t)m * G.ldr + n);
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = mrow0 + 8 * e + it;
      if (m < M && nok) {
        float v[8] = {acc[it][0][e], acc[it][1][e], acc[it][2][e], acc[it][3][e], acc[it][4][e], acc[it][5][e], acc[it][6][e], acc[it][7][e]};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bias8[j];
        if (epi == RF_EPI_GELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_tanh(v[j]);
        } else if (epi == RF_EPI_GATE_RES) {
          float rr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) rr[j] = 0.f;
          if (G.residual != nullptr) unpack8(resv[it], rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(gate8[j], v[j], rr[j]);
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
            for (int j = 0; j < 8; ++j) ss = fmaf(v[j], v[j], ss);
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
            const float rs = rsqrtf(fmaf(ss, 1.0f / 128.0f, p.norm_eps));
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const float a = v[j] * rs * nw8[j], b = v[j + 1] * rs * nw8[j + 1];
              v[j] = fmaf(a, cs[j], -(b * sn[j]));        // explicit contraction: every epilogue of the library rounds the same way
              v[j + 1] = fmaf(b, cs[j + 1], a * sn[j + 1]);
            }
          }
          if (which == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= p.q_scale;
          }
          dst = (which == 0 ? p.q : p.k) + ((int64_t)head * p.s_pad + G.tok_offset + m) * 128 + q8;
        } else {
          dst = G.out + (int64_t)m * G.ldo + (n - ncol_base);
        }
        *(u32x4*)dst = pack8(v);
      }
    }
  }
}

// (no packed-fp32 VALU ops in this kernel either: its register-direct epilogue runs the RMSNorm + RoPE arithmetic that misbehaved
//  in the 128 x 128 kernel, see gemm_bf16_kernel, and the guide prices packed f32 beside MFMAs as an anti-lever anyway)
RF_NO_PACKED_FP32 __global__ __launch_bounds__(256) void gemm_bf16_w4b_kernel(const GemmParams p) {
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
  f32x4 acc[8][8];
  gemm_mainloop_w4b(G, p.N, m0, n0, nk, acc, smem, w, lane);
  if (p.probe) clk.end(g_clk_probe);
  gemm_epilogue_direct(p, G, acc, m0, n0, (w >> 1) * 128, (w & 1) * 128, lane);
}
