// One wave per SIMD: 256 x 256 x 64 block tile, 4 waves (2 x 2), 128 x 128 wave tiles on v_mfma_f32_16x16x32_bf16.
// (included by gemm_bf16.hip after the epilogues; namespace rf)
//
// Why (profiles/r04_telemetry.md): with the firmware's own counters the 8-wave ping-pong kernel is PPT-limited 31-92 % of the
// time at 0.99-1.11 J per TFLOP while hipBLASLt's assembly kernel of the same macro tile needs 0.90-1.04 and is 9-12 % faster on
// every cfg2 shape -- under a power cap the currency is joules per MAC.  A 64 x 128 wave tile reads 12 operand fragments for 32
// MFMAs (0.75 KiB of LDS per MFMA), a 128 x 128 wave tile 16 for 64 (0.5 KiB), and four waves instead of eight halve the
// fragment copies of a staged K-tile (each A row is read by 2 waves instead of 2, each W row by 2 instead of 4).  Round 2 tried
// one wave per SIMD on 32x32x16 MFMAs (experiments/gemm_kernels_exp.inc); this is the 16x16x32 form with a simpler pipeline.
//
// Pipeline (one barrier per K-tile, no taken branch in the steady state but the back edge):
//   LDS = 2 stages x {A image, W image} of a K-tile (256 rows x 128 B each, 16-byte chunks XOR-swizzled as in the other loops).
//   A K-tile is two k-steps of 64 MFMAs.  While k-step s multiplies out of fragment set F[s], the 16 fragments of the NEXT
//   k-step are read into F[1 - s] (one ds_read_b128 per 4 MFMAs), so no MFMA ever waits for LDS:
//     step A (k-step 0 of tile t): read k-step 1 of tile t;          then vmcnt(0): tile t+1 has landed; barrier
//     step B (k-step 1 of tile t): read k-step 0 of tile t+1 (other stage); LDS-DMA tile t+2 into tile t's stage, one 1 KiB piece
//                                  per 4 MFMAs (16 pieces per wave), which gives every piece the whole step A of tile t+1 to land.
//   The stage of tile t is last READ in step A of tile t (its k-step-1 fragments), every wave waits for those reads before the
//   barrier, and it is first overwritten in step B: no wave can see a half-written image.
//   (A ring of four half-tiles with 64-byte rows -- one more step of flight time per piece, no swizzle -- was measured too: 8 %
//   slower and 6 % more joules per FLOP, the half-line fetches cost more than the slack buys; profiles/r04_gemm_w4.md.)
// Operand rows beyond M / N are not clamped: the buffer resources carry the operands' true extents and the hardware range
// check returns zeros for them.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__)
#define RF_MAKE_RSRC_N(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), 0, (int)(bytes), 0x00020000)
#else
#define RF_MAKE_RSRC_N(p, bytes) 0
#endif

__device__ __forceinline__ void gemm_mainloop_w4m16(const GemmGroupDev& G, const int N, const int m0, const int n0, const int nk,
                                                    f32x4 (&acc)[8][8], char* smem, const int w, const int lane) {
  constexpr int IMG = 256 * 128;     // one operand image of a K-tile
  constexpr int STAGE = 2 * IMG;     // A image | W image
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int swz = l15 >> 1;          // (row >> 1) & 7 of a fragment row (tile rows are 16-row aligned)
  // per-lane fragment offsets inside a stage for the two k-steps (row tile rt / column tile ct add an immediate rt * 2048)
  const int fa0 = (wm * 128 + l15) * 128 + (((0 + g4) ^ swz) << 4), fa1 = (wm * 128 + l15) * 128 + (((4 + g4) ^ swz) << 4);
  const int fb0 = IMG + (wn * 128 + l15) * 128 + (((0 + g4) ^ swz) << 4), fb1 = IMG + (wn * 128 + l15) * 128 + (((4 + g4) ^ swz) << 4);

  // LDS-DMA geometry: piece j (0..7) of wave w fills rows 32 j + 8 w + lane / 8 of an image (1 KiB, lane-linear); the swizzle
  // (slot = chunk ^ ((row >> 1) & 7)) is applied to the SOURCE chunk of a lane
  const int r8 = lane >> 3;
  const uint32_t chunk_b = (uint32_t)(((lane & 7) ^ (((w & 1) << 2) + (lane >> 4))) * 16);
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
  // piece g of the tile under cursor c into stage `st`: g < 8 -> A piece g, else W piece g - 8
  auto piece = [&](const Cur& c, const int st, const int g) {
    char* dst = smem + st * STAGE + (g < 8 ? 0 : IMG) + ((g & 7) * 32 + 8 * w) * 128;
    if (g < 8) RF_BUF_LOAD_LDS(c.A, (lds_void*)dst, c.voA, c.kk * 128 + (g & 7) * c.stA);
    else RF_BUF_LOAD_LDS(c.W, (lds_void*)dst, c.voB, c.kk * 128 + (g & 7) * c.stB);
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
  bf16x8 A0[8], B0[8], A1[8], B1[8];       // F[0] = (A0, B0): k-step 0, F[1] = (A1, B1): k-step 1
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    A0[g] = *(const bf16x8*)(smem + fa0 + g * 2048);
    B0[g] = *(const bf16x8*)(smem + fb0 + g * 2048);
  }

  // The accumulators fill the whole AGPR file (64 tiles x 4 registers = 256): with the builtin the register allocator has no
  // slack, renames accumulators across the MFMAs and pays ~280 v_accvgpr moves + s_nops per K-tile.  The asm form pins every
  // accumulator to its AGPR quad and accumulates in place.  (MFMA -> MFMA on the same accumulator is interlocked by the
  // hardware and here 64 MFMAs apart; the epilogue's first read of an accumulator is fenced by explicit s_nops below.)
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_W4_MFMA(C, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(A_), "v"(B_))
#else
#define RF_W4_MFMA(C, A_, B_) ((void)0)
#endif
  // one k-step: 64 MFMAs out of (FA, FB) in 16 groups of 4; group g also issues fragment read g of the next k-step (READ)
  // and LDS-DMA piece g of tile t+2 (DMA)
#define RF_W4_STEP(FA, FB, NA, NB, NBASE, NOFFA, NOFFB, READ, DMA, DSTAGE)                                              \
  _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                                     \
    if (READ) {                                                                                                        \
      if (g < 8) NA[g & 7] = *(const bf16x8*)((NBASE) + (NOFFA) + (g & 7) * 2048);                                      \
      else NB[g & 7] = *(const bf16x8*)((NBASE) + (NOFFB) + (g & 7) * 2048);                                            \
    }                                                                                                                  \
    if (DMA) piece(c, (DSTAGE), g);                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                    \
      const int rt = g >> 1, ct = 4 * (g & 1) + q;                                                                     \
      RF_W4_MFMA(acc[rt][ct], FA[rt], FB[ct]);                                                                         \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }
#define RF_W4_SYNC()                                          \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_s_barrier();                               \
  __builtin_amdgcn_sched_barrier(0)

  int t = 0;
  for (; t + 2 < nk; ++t) {      // steady state: tiles t+1 and t+2 exist
    const char* cur = smem + (t & 1) * STAGE;
    const char* oth = smem + ((t + 1) & 1) * STAGE;
    RF_W4_STEP(A0, B0, A1, B1, cur, fa1, fb1, true, false, 0)
    RF_W4_SYNC();
    RF_W4_STEP(A1, B1, A0, B0, oth, fa0, fb0, true, true, t & 1)
    next(c);
  }
  if (t + 1 < nk) {               // second to last tile: nothing left to stage
    const char* cur = smem + (t & 1) * STAGE;
    const char* oth = smem + ((t + 1) & 1) * STAGE;
    RF_W4_STEP(A0, B0, A1, B1, cur, fa1, fb1, true, false, 0)
    RF_W4_SYNC();
    RF_W4_STEP(A1, B1, A0, B0, oth, fa0, fb0, true, false, 0)
    ++t;
  }
  {                               // last tile
    const char* cur = smem + (t & 1) * STAGE;
    RF_W4_STEP(A0, B0, A1, B1, cur, fa1, fb1, true, false, 0)
    RF_W4_STEP(A1, B1, A0, B0, cur, fa0, fb0, false, false, 0)
  }
#undef RF_W4_STEP
#undef RF_W4_SYNC
#undef RF_W4_MFMA
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have left the matrix pipe before the epilogue reads the AGPRs
}

__global__ __launch_bounds__(256) void gemm_bf16_w4m16_kernel(const GemmParams p) {
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
  gemm_mainloop_w4m16(G, p.N, m0, n0, nk, acc, smem, w, lane);
  if (p.probe) clk.end(g_clk_probe);
  __syncthreads();  // every wave is done reading the staged operands: the LDS is free
  gemm_epilogue_lds16<4, false>(p, G, acc, m0, n0, (w >> 1) * 128, (w & 1) * 128, lane, smem + w * EPI_REGION);
}
