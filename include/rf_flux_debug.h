/* rf_flux_debug.h -- measurement and introspection entry points of librf_flux.so.
 *
 * NOT part of the drop-in boundary (include/rf_flux.h): nothing here is needed to run the path, nothing here selects a kernel or
 * changes a result.  bench.py's roofline block, tools/ and the tests use them:
 *   rf_time_gemm*            hipEvent timing of one GEMM descriptor, isolated re-launches
 *   rf_profile_begin / _end  in-sequence per-launch timing of everything the library launches (the `roofline` numbers)
 *   rf_debug_last_*_path     which kernel form the LAST launch of a family took (read-only)
 *   rf_debug_sk_plan, rf_debug_attn_mix_plan   the host-side work plans (pure arithmetic, callable without a GPU)
 *   rf_debug_clock_probe     the shader clock a probed launch saw (rf_gemm_desc.clock_probe / an open profile)
 */
#ifndef RF_FLUX_DEBUG_H
#define RF_FLUX_DEBUG_H
#include "rf_flux.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-level timing hook used by bench.py: time `iters` launches of the dominant GEMM
 * shape with hipEvents on `stream`; returns average microseconds in *us. */
int rf_time_gemm(const rf_gemm_desc* d, int32_t iters, float* us, void* stream);
int rf_time_gemm_w8a8(const rf_gemm_desc* d, int32_t iters, float* us, void* stream);

/* In-sequence timing hook (bench.py's `roofline`): between rf_profile_begin and rf_profile_end every launch site of
 * the library records ONE hipEvent on ITS launch stream in front of its kernel(s); rf_profile_end records a closing
 * event.  A launch's duration is the distance to the next event (kernel + the gap behind it), so the durations are
 * those of the kernels inside the real 57-block sequence (cold operands, real neighbours), not of isolated
 * re-launches, and the per-class sums add up exactly to the wall time between the first and the closing event.
 * rf_profile_end synchronises, then fills, per rf_kernel_class: summed duration (us), launch count and summed
 * algorithmic work (FLOPs: 2MNK per GEMM over all groups and K-segments, 4*S^2*128*heads per attention launch;
 * BYTES read+written for the row kernels).  `dropped` = launches beyond max_launches (not timed).
 * Not thread-safe, single stream, not for use during hipGraph capture; costs one event record per launch while open. */
typedef enum rf_kernel_class {
  RF_KC_GEMM_MAIN = 0,   /* 256x256-tile MFMA GEMM launches (tile-per-block ping-pong loop and stream-K) */
  RF_KC_GEMM_SMALL = 1,  /* 128x128-tile launches (embedders, LoRA down-projections incl. split-K + reduce) */
  RF_KC_ATTN = 2,        /* rf_attention / rf_attention_fwd */
  RF_KC_ROWOP = 3,       /* LayerNorm+modulate, RMSNorm+RoPE, Euler, SiLU, add */
  RF_KC_GEMM_W8 = 4,     /* fp8-weight GEMM launches (rf_gemm_w8a8) */
  RF_KC_QUANT = 5,       /* activation quantisation row kernels of the fp8 path */
  RF_KC_ATTN_BWD = 6,    /* rf_attention_bwd (work = 5 products x 2 S^2 128 per head = 2.5 x the forward's) */
  RF_KC_COUNT = 7
} rf_kernel_class;
int rf_profile_begin(int32_t max_launches);
int rf_profile_end(double* us_sum /*[RF_KC_COUNT]*/, int64_t* launches /*[RF_KC_COUNT]*/,
                   double* work_sum /*[RF_KC_COUNT]*/, int32_t* dropped /* may be NULL */);

/* Read-only introspection (no setter exists: a launch's schedule / kernel travels in ITS descriptor).
 * rf_debug_last_gemm_path: 0 = one tile per workgroup, 1 = split-K, 2 = stream-K / persistent whole tiles.
 * rf_debug_last_attn_path: 1 / 2 / 4 / 5 = kernel version, 6 = v5 split launch, 8 = v5 lagged-max, 9 = its split launch, 10 / 11 = the
 * mixed-size launch of the bounded / lagged-max kernel.  rf_debug_last_attn_bwd_path: the RF_ATTN_BWD_* pair the last backward ran. */
int rf_debug_last_gemm_path(void);
int rf_debug_last_attn_path(void);
int rf_debug_last_attn_bwd_path(void);
/* The stream-K plan `d` would get on a device with num_cus compute units (pure host arithmetic, no launch).  Returns 1 = plan made,
 * 0 = the launch does not qualify, < 0 error.  out (>= 51 ints): [0..4] first K-tile iteration per group (+ total), [5..8] K-tiles per
 * tile per group, then per XCD chunk x < 8: [9+x] first tile, [17+x] whole-tile rounds, [25+x] / [33+x] first / one-past-last iteration
 * of the stream-K region; [41] tiles_n, [42..45] tiles_m per group, [46..49] tile_start per group, [50] total tiles. */
int rf_debug_sk_plan(const rf_gemm_desc* d, int32_t num_cus, int32_t* out);
/* The mixed-size attention launch for S keys x heads on num_cus CUs: out = {256-query workgroups per head, 192-query workgroups per
 * head, 1000 x simulated makespan in units of a 256-query workgroup, plain-grid rounds}. */
int rf_debug_attn_mix_plan(int32_t S, int32_t heads, int32_t num_cus, int32_t* out);
/* which: 0 = the last probed 256x256 GEMM launch, 1 = the last bounded-score attention launch, 2 = the GEMM's epilogue (end of main
 * loop -> stores acknowledged).  mhz = shader clocks per microsecond over workgroup 0's region, us = its duration.  Synchronise first. */
int rf_debug_clock_probe(int which, double* mhz, double* us);

#ifdef __cplusplus
}
#endif
#endif /* RF_FLUX_DEBUG_H */
