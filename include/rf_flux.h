/* rf_flux.h -- C ABI of librf_flux.so: the MI355X (gfx950) hot path of ReflectionFlow's
 * FLUX MM-DiT denoise loop.
 *
 * The reference (/root/reference, 100 % Python) has no FFI: its "plugin interface" for
 * this path is four duck-typed Python functions.  Each entry point below names the
 * reference call it replaces; the Python host side (the reflectionflow_amd/flux package) keeps
 * the reference's function names/signatures and binds these symbols with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library never
 *     allocates, frees or synchronises in a launch path (hipGraph-capturable);
 *   - all matrices are bf16 row-major unless stated; accumulation is fp32;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued asynchronously;
 *   - return value: 0 = ok, <0 = rf_status; rf_last_error() gives a thread-local message;
 *   - one process per GPU; calls on one stream are not re-entrant.
 */
#ifndef RF_FLUX_H
#define RF_FLUX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rf_status {
  RF_OK = 0,
  RF_ERR_SHAPE = -1,       /* unsupported / inconsistent shape                      */
  RF_ERR_ALIGN = -2,       /* pointer or leading dimension not 16-byte aligned      */
  RF_ERR_NULL = -3,        /* required pointer is NULL                              */
  RF_ERR_HIP = -4,         /* HIP runtime error (message in rf_last_error)          */
  RF_ERR_UNSUPPORTED = -5, /* feature not built                                     */
  RF_ERR_WORKSPACE = -6    /* workspace too small                                   */
} rf_status;

const char* rf_last_error(void);
/* ABI version: bump on any struct/signature change (v13: rf_attn_bwd_desc.kernel, rf_lora_adamw / rf_lora_prodigy; v14: rf_lora_fuse /
 * rf_lora_unfuse_grads, rf_qkv_train_fwd's forward-only form; v15: rf_lora_clip_grad_norm). */
int rf_abi_version(void);
/* Returns 950 when the library was compiled for gfx950. */
int rf_target_arch(void);

/* ------------------------------------------------------------------------------------
 * GEMM with fused epilogues  (replaces nn.Linear + the element-wise ops around it)
 *   out[m,n] = epi( sum_s sum_k A_s[m,k] W_s[n,k]  + bias[n] )        s = K-segment 0..2
 * W is nn.Linear's [out,in] layout.  Extra K-segments (per token group) carry
 *   - the LoRA low-rank term  (A_s = x.lora_A^T, W_s = scaling*lora_B; PEFT lora.Linear,
 *     active only on the token groups lora_controller.py:5-42 leaves it on for), and
 *   - the single-block proj_out whose input is cat([attn, mlp]) (block.py:320-322)
 *     without materialising the concat.
 * Requirements: every segment K % 64 == 0, 16-byte aligned rows.  M, N arbitrary.
 * ---------------------------------------------------------------------------------- */
typedef enum rf_epilogue {
  RF_EPI_STORE = 0,     /* out = acc + bias                               (nn.Linear)             */
  RF_EPI_GELU = 1,      /* out = gelu_tanh(acc + bias)       (block.py:252,296 ff.net[0]/proj_mlp) */
  RF_EPI_GATE_RES = 2,  /* out = residual + gate[n]*(acc+bias)  (block.py:218-226,263-266,320-323) */
  RF_EPI_QKV = 3,       /* cols split into q|k|v, written head-major  (block.py:27-36,46-58,81-91) */
  RF_EPI_QKV_GELU = 4   /* cols < n_split: QKV; cols >= n_split: GELU -> out  (single block fused) */
} rf_epilogue;

/* How a launch is cut into workgroups.  AUTO is what every product caller passes; the explicit values exist so that tests
 * (and a caller who knows better) can pin a schedule PER LAUNCH -- the library has no process-global kernel switch. */
typedef enum rf_gemm_schedule {
  RF_SCHED_AUTO = 0,        /* by shape: 128x128 tiles for small problems, 256x256 otherwise, stream-K below 83 % round fill */
  RF_SCHED_TILE128 = 1,     /* 128x128x64 tiles (split-K over the grid when it qualifies and scratch is attached)           */
  RF_SCHED_TILE256 = 2,     /* 256x256x64 tiles, one per workgroup, never stream-K                                          */
  RF_SCHED_STREAMK = 3,     /* 256x256 stream-K whenever feasible (needs splitk_ws; else as TILE256)                        */
  RF_SCHED_PERSISTENT = 4,  /* one persistent workgroup per CU walking whole 256x256 tiles (needs splitk_ws)                */
  RF_SCHED_PLAIN256 = 5,    /* 256x256 tiles on the plain double-buffered loop: bit-exact reference of the ping-pong loops  */
  RF_SCHED_W4 = 6,          /* 256x256 tiles, ONE wave per SIMD (4 waves x 128x128 wave tiles): fewer LDS bytes per MFMA     */
  RF_SCHED_W4B = 7          /* ... with three half-stage barriers per K-tile: LDS-DMA spread over the whole tile (round 5)  */
} rf_gemm_schedule;

typedef struct rf_kseg {
  const void* A; int64_t lda;         /* [M x K] activations */
  const void* W; int64_t ldw;         /* [N x K] weights     */
  int32_t K; int32_t _pad;            /* 0 = segment unused  */
} rf_kseg;

typedef struct rf_gemm_group {
  rf_kseg seg[3];
  const void* bias;                   /* [N] or NULL */
  int32_t M;
  int32_t tok_offset;                 /* QKV: first token row of this group in the joint sequence */
  void* out; int64_t ldo;             /* STORE/GELU/GATE_RES: [M x N];  QKV_GELU: [M x (N-n_split)] */
  const void* residual; int64_t ldr;  /* GATE_RES: [M x N] (may alias out); NULL = 0 */
  const void* gate;                   /* GATE_RES: [N] */
  const void* norm_q;                 /* QKV + rope_cos: per-head RMSNorm weights [128] for this group's q rows */
  const void* norm_k;                 /*                 ... and k rows (text group: norm_added_q/k)             */
  const float* a_scale;               /* rf_gemm_w8a8: fp32 [M] per-token dequantisation scale of this group's A rows  */
  const float* w_scale;               /* rf_gemm_w8a8: fp32 [N] per-output-channel dequantisation scale of this group's W */
} rf_gemm_group;

typedef struct rf_gemm_desc {
  int32_t N;
  int32_t epilogue;                   /* rf_epilogue */
  int32_t num_groups;                 /* 1..4 token groups sharing N and epilogue (txt/img/cond) */
  int32_t n_split;                    /* QKV_GELU: first GELU column (= 3*heads*128) */
  /* QKV epilogues: joint attention operands, S_pad = round_up(S_total, 64) rows per head */
  void* q;  void* k;                  /* [heads][S_pad][128] */
  void* vt;                           /* [heads][S_pad/64][128][64] key-permuted, see rf_attention_fwd */
  int32_t heads; int32_t s_pad;
  /* optional fusion of rf_qk_rmsnorm_rope into the QKV epilogue (block.py:38-41,60-67,74-78,92-99):
   * when rope_cos != NULL, q and k rows are RMS-normalised with the group's norm_q/norm_k and rotated
   * with the fp32 tables [S][128] (indexed by joint token row) before they are stored. */
  const float* rope_cos; const float* rope_sin;
  float norm_eps;
  float q_scale;                      /* QKV: multiply the q rows by this (fp32, before the one bf16 rounding);
                                         0 = 1.0.  The engine folds softmax_scale*log2(e) in here and tells
                                         rf_attention_fwd via q_prescaled, saving a multiply per score. */
  int32_t schedule;                   /* rf_gemm_schedule; 0 = RF_SCHED_AUTO */
  int32_t clock_probe;                /* != 0: block 0 of the 256x256 kernels stores {s_memtime, s_memrealtime} around its main loop
                                         (read back by rf_debug_clock_probe; bench.py).  Also on while rf_profile_begin is open;
                                         product launches pass 0. */
  /* optional GEMM scratch (caller-owned, 16-byte aligned; one per stream).  Layout: [0, 4096) flags, then fp32
   * partial tiles.  The first 4 KiB must be ZERO before the first launch that uses the buffer; every launch
   * leaves them zero again.  With it the library may
   *   - stream-K the 256x256-tile kernel: the launch's K-tile iterations are cut into one equal contiguous range
   *     per CU, so e.g. 216 or 264 tiles on 256 CUs cost 0.84 / 1.03 tile-times instead of 1 / 2.  Needs
   *     4096 + num_CUs * 256 KiB (64 MiB + 4 KiB on MI355X); smaller buffers simply disable it;
   *   - split-K a STORE GEMM with one group, <= 64 output tiles and a long K (the LoRA down-projections
   *     x.lora_A^T: 8 tiles x up to 240 K-tiles) over the grid, reduced by a second kernel.
   * Both sum partials in a fixed order: results are bit-reproducible run to run.  NULL = neither. */
  void* splitk_ws; int64_t splitk_ws_bytes;
  rf_gemm_group g[4];
} rf_gemm_desc;

int rf_gemm_bf16(const rf_gemm_desc* d, void* stream);

/* The same GEMM with fp8 operands (BASELINE cfg5, "fp8 MFMA weights"; the reference has no fp8 path -- semantics are
 * defined here): A and W are OCP e4m3fn bytes, symmetric absmax-quantised
 *      A8[m,k] = e4m3(A[m,k] / a_scale[m]),  a_scale[m] = max_k |A[m,k]| / 448      (per token; rf_quant_rows_fp8)
 *      W8[n,k] = e4m3(W[n,k] / w_scale[n]),  w_scale[n] = max_k |W[n,k]| / 448      (per output channel; packed once)
 * and   out[m,n] = epi( a_scale[m] * w_scale[n] * sum_k A8[m,k] W8[n,k]  + bias[n] )   with fp32 accumulation, every
 * epilogue of rf_gemm_bf16 (bias / gate / residual / outputs stay bf16).  The products run on the block-scaled fp8 MFMA
 * (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales) at twice the bf16 MFMA rate.  Every segment of a group shares
 * the group's a_scale (a two-segment input must be quantised with a common row scale); K % 128 == 0 per segment; lda /
 * ldw / K count ELEMENTS (= bytes); needs the 16-byte aligned epilogue path.
 * MIXED precision per token group: a group whose a_scale is NULL is an ordinary bf16 group (bf16 A / W, K % 64 == 0,
 * LoRA K-segments allowed) and rides in the same launch -- cfg5's LoRA'd condition rows next to the fp8 text / image rows. */
int rf_gemm_w8a8(const rf_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused per-head RMSNorm(q,k) + interleaved-pair RoPE, in place on head-major q,k
 * (replaces attn.norm_q/k, norm_added_q/k and apply_rotary_emb: block.py:38-41,60-67,74-78,92-99)
 *   rows [0, n_added) use w_added_* (text stream), rows [n_added, S) use w_* .
 *   (The engine fuses this into the QKV GEMM epilogue -- rf_gemm_desc.rope_cos; the standalone kernel
 *   remains for callers that produce q,k themselves.)
 *   cos,sin: fp32 [S][128] (FluxPosEmbed tables for [txt|img|cond] rows, transformer.py:130-134)
 * ---------------------------------------------------------------------------------- */
int rf_qk_rmsnorm_rope(void* q, void* k, int32_t heads, int32_t S, int32_t s_pad, int32_t n_added,
                       const void* w_q, const void* w_k, const void* w_added_q, const void* w_added_k,
                       const float* cos_tab, const float* sin_tab, float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * Non-causal flash attention forward, head_dim 128
 * (replaces F.scaled_dot_product_attention + the transposes, block.py:106-129)
 *   q,k : [heads][S_pad][128] bf16;  vt : [heads][S_pad/64][128][64] bf16 where the 64 keys of a
 *         tile are stored at position  p(kv) = (kv & ~12) | ((kv & 4) << 1) | ((kv & 8) >> 1)
 *         (bits 2 and 3 swapped) -- the order the PV MFMA consumes them in.
 *   out : [S][heads*128] bf16 (token-major, ready for the out-projection GEMM)
 *   n_main: rows [0,n_main) are text+image tokens, [n_main,S) condition tokens.
 *   mode: 0 = plain; 1 = additive bias `cross_bias` on (main<->cond) blocks (attn.c_factor,
 *         block.py:115-122); 2 = mask (main<->cond) blocks (union_cond_attn=False, block.py:106-114)
 *   q_prescaled: 0 = scores are scaled by `scale` here; 1 = q already carries scale*log2(e)
 *         (rf_gemm_desc.q_scale), `scale` is ignored.
 *   score_bound: 0 = unknown (online softmax with a running row maximum).  > 0: the caller GUARANTEES
 *         |q.k * scale * log2(e)| <= score_bound for every query/key pair.  FLUX RMS-normalises q and k per head
 *         (block.py:38-41,60-67), so sqrt(128) * max|norm_q.weight| * max|norm_k.weight| * log2(e) is such a bound
 *         whatever the activations are (rf_*_block_weights.qk_bound).  With a bound <= 100, mode 0 (no bias/mask), S % 256 == 0
 *         and a prescaled q the library runs the bounded-score kernel: softmax is shift invariant, so P = exp2(s)
 *         needs no running maximum, no exchange and no rescaling of O (bf16 P / fp32 O,l have the exponent range).
 *         A violated guarantee gives inf/NaN -- pass 0 when in doubt.  Without a usable bound (0, or > 100) the same
 *         launches take the LAGGED-MAX form of that kernel (rf_attn_desc below), which is exact for any q, k and within a
 *         few per cent of the bounded form's speed, so a checkpoint's norm weights no longer decide how fast attention runs.
 * ---------------------------------------------------------------------------------- */
int rf_attention_fwd(const void* q, const void* k, const void* vt, void* out, int32_t heads,
                     int32_t S, int32_t s_pad, int64_t ldo, int32_t n_main, int32_t mode,
                     float cross_bias, float scale, int32_t q_prescaled, float score_bound, void* stream);
/* The same call with scratch: when a shift-free kernel applies and its grid of heads * S/256 workgroups would
 * fill < 80 % of its rounds of CUs (S = 5632: 528 workgroups = 2.06 rounds run as 3), the library runs one persistent
 * workgroup per CU over equal shares of the (query block, key range) space; a block whose keys were split between
 * two workgroups leaves its partial (O, l, m) in `ws` and a second launch adds them.
 * ws: 16-byte aligned device memory, >= rf_attention_ws_bytes() bytes, may be shared with any other
 * scratch that is idle during this call (the engine passes its GEMM scratch); NULL / too small = rf_attention_fwd. */
int rf_attention_fwd_ws(const void* q, const void* k, const void* vt, void* out, int32_t heads,
                        int32_t S, int32_t s_pad, int64_t ldo, int32_t n_main, int32_t mode,
                        float cross_bias, float scale, int32_t q_prescaled, float score_bound,
                        void* ws, int64_t ws_bytes, void* stream);
int64_t rf_attention_ws_bytes(void);   /* scratch size that enables the split launch on the current device */

/* Descriptor form of the same call (rf_attention_fwd / _ws fill one with kernel = RF_ATTN_AUTO).  `kernel` pins the kernel
 * PER LAUNCH (tests, A/B); a request whose preconditions do not hold FAILS (RF_ERR_UNSUPPORTED) instead of running something
 * else.  AUTO: no mask / bias, S % 256 == 0 and a prescaled q take a shift-free kernel -- the bounded form when the caller
 * proves 0 < score_bound <= 100, otherwise the LAGGED-MAX form, which needs no bound at all: P = exp2(s - m) with a per-row
 * maximum m that starts as the exact maximum of the first key tile and is re-centred only when a tile's row sums exceed
 * lag_thresh (2^30; the sums are formed anyway), so the result is the exact softmax to rounding for ANY q, k -- the speed of
 * the path no longer depends on the norm weights of the checkpoint.  Everything else runs the online-softmax kernels. */
typedef enum rf_attn_kernel {
  RF_ATTN_AUTO = 0,
  RF_ATTN_ONLINE128 = 1,        /* online softmax, 4 waves x 32 queries (masks, bias, ragged S)            */
  RF_ATTN_ONLINE256 = 2,        /* online softmax, 8 waves x 32 queries, K / VT rings                      */
  RF_ATTN_BOUNDED32 = 4,        /* bounded score, 32x32x16 MFMAs (explicit requests only since round 3)    */
  RF_ATTN_BOUNDED16 = 5,        /* bounded score, 16x16x32 MFMAs                                           */
  RF_ATTN_BOUNDED16_SPLIT = 6,  /* ... as one persistent workgroup per CU + combine launch (needs ws)      */
  RF_ATTN_LAGGED16 = 8,         /* lagged-max, 16x16x32 MFMAs: no bound needed                             */
  RF_ATTN_LAGGED16_SPLIT = 9,   /* ... split launch (needs ws)                                             */
  RF_ATTN_BOUNDED16_MIX = 10,   /* bounded score, workgroups of 256 AND 192 queries sized to fill the CUs  */
  RF_ATTN_LAGGED16_MIX = 11     /* lagged-max, the same launch                                             */
} rf_attn_kernel;
typedef struct rf_attn_desc {
  const void *q, *k, *vt; void* out;  /* as rf_attention_fwd */
  int32_t heads, S, s_pad, n_main;
  int64_t ldo;
  int32_t mode, q_prescaled;
  float cross_bias, scale, score_bound;
  float lag_thresh;                   /* lagged-max kernels: re-centre a row when a lane's 16-key sum of P exceeds this;
                                         0 = 2^30.  (Tests sweep it: any value gives the same softmax to rounding.) */
  int32_t kernel;                     /* rf_attn_kernel */
  int32_t mix_small;                  /* RF_ATTN_*_MIX only: 192-query workgroups per head, b % 4 == 0 and 12 b <= S / 16 (the
                                         other S / 256 - 3 b / 4 workgroups of a head take 256 queries); 0 = sized by the
                                         library for the device's CU count (S = 4608 x 24 heads on 256 CUs: 12) */
  void* ws; int64_t ws_bytes;         /* optional scratch, see rf_attention_fwd_ws */
  float* lse;                         /* optional OUT [heads][s_pad] fp32 (rows < S written): log2 sum_k exp2(s2[q][k]) of every query
                                         row, s2 = the scaled (and biased) scores in the exp2 domain -- the row statistic of the
                                         backward (rf_attn_bwd_desc.lse with lse_given = 1).  NULL: not produced. */
} rf_attn_desc;
int rf_attention(const rf_attn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm(no affine, eps) + (1+scale)*x + shift, row-wise over D
 * (replaces AdaLayerNormZero/ZeroSingle/Continuous' norm+modulate and norm2+modulate:
 *  block.py:186-201,232-247,295-299; transformer.py:243)
 * ---------------------------------------------------------------------------------- */
int rf_layernorm_modulate(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t rows, int32_t D,
                          const void* scale, const void* shift, float eps, void* stream);

/* fp8 producers for rf_gemm_w8a8 (cfg5).  Both write OCP e4m3fn rows (1 byte per element, ldo in elements) and the
 * per-row dequantisation scale  row_scale[m] = max_k |y[m,k]| / 448  (1 for an all-zero row):
 *   rf_layernorm_modulate_fp8: y = LayerNorm(x)*(1+scale)+shift as rf_layernorm_modulate, quantised from fp32;
 *   rf_quant_rows_fp8: y = [x0 | x1] (x1 optional: the single block's cat([attn, mlp]) input, block.py:320) with ONE
 *                      scale per row across both segments; K0 + K1 <= 16384. */
int rf_layernorm_modulate_fp8(const void* x, int64_t ldx, void* out8, int64_t ldo, float* row_scale, int32_t rows,
                              int32_t D, const void* scale, const void* shift, float eps, void* stream);
int rf_quant_rows_fp8(const void* x0, int64_t ld0, int32_t K0, const void* x1, int64_t ld1, int32_t K1, void* out8,
                      int64_t ldo, float* row_scale, int32_t rows, void* stream);

/* Euler step of FlowMatchEulerDiscreteScheduler.step (generate.py:276):
 *   x <- bf16( float(x) + dt * float(v) ),  dt = sigma_next - sigma */
int rf_euler_step(void* x, const void* v, int64_t n, float dt, void* stream);

/* out = bf16(silu(float(x)))  (the SiLU in front of every AdaLN linear) */
int rf_silu(const void* x, void* out, int64_t n, void* stream);

/* out[i] += x[i] (bf16): `hidden_states += cond_attn_output` when add_cond_attn (block.py:227-228) */
int rf_add_inplace(void* out, const void* x, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Whole blocks / whole forward / whole denoise loop
 * ---------------------------------------------------------------------------------- */
typedef struct rf_lora_seg {          /* LoRA of one fused linear (B == NULL: none) */
  const void* A;  /* [r_pad x K]   stacked lora_A rows, zero padded to r_pad % 64 == 0            */
  const void* B;  /* [N x r_pad]   block-diagonal scaling*lora_B                                   */
  int32_t r_pad;
  int32_t merged; /* != 0: B is instead the MERGED weight bf16(W + scaling * lora_B lora_A), [N x K] in the layout of the base
                     weight; the token groups LoRA acts on multiply by it and no low-rank launches are made (A, r_pad unused).
                     Static LoRA only (inference); costs one more copy of the LoRA'd weights and one bf16 rounding of the sum. */
} rf_lora_seg;

typedef struct rf_w8 {                 /* fp8 copy of one (fused) nn.Linear weight for rf_gemm_w8a8; w == NULL: none */
  const void* w;                       /* [N x K] OCP e4m3fn bytes, same row order as the bf16 weight */
  const float* scale;                  /* [N] per-output-channel dequantisation scale */
} rf_w8;

typedef struct rf_double_block_weights {   /* FluxTransformerBlock, SURVEY Appendix A.2 */
  const void *w_qkv, *b_qkv;           /* cat(to_q,to_k,to_v)             [3D x D], [3D] */
  const void *w_add_qkv, *b_add_qkv;   /* cat(add_q,add_k,add_v)_proj     [3D x D], [3D] */
  const void *norm_q, *norm_k, *norm_added_q, *norm_added_k;   /* [128] */
  const void *w_out, *b_out;           /* attn.to_out.0   [D x D] */
  const void *w_add_out, *b_add_out;   /* attn.to_add_out [D x D] */
  const void *w_ff1, *b_ff1;           /* ff.net.0.proj   [4D x D] */
  const void *w_ff2, *b_ff2;           /* ff.net.2        [D x 4D] */
  const void *w_ffc1, *b_ffc1;         /* ff_context.net.0.proj */
  const void *w_ffc2, *b_ffc2;         /* ff_context.net.2 */
  rf_lora_seg lora_qkv, lora_out, lora_ff2;   /* FLUX-Corrector LoRA (config.yaml:53) */
  float qk_bound;                      /* proven bound on |score * log2 e| from the four norm weights (see
                                          rf_attention_fwd score_bound); 0 = none */
  int32_t _pad;
  /* cfg5: fp8 copies of the eight big weights (base weights only; used when rf_flux_dims.fp8 != 0) */
  rf_w8 q_qkv, q_add_qkv, q_out, q_add_out, q_ff1, q_ff2, q_ffc1, q_ffc2;
} rf_double_block_weights;

typedef struct rf_single_block_weights {   /* FluxSingleTransformerBlock, Appendix A.3 */
  const void *w_qkv_mlp, *b_qkv_mlp;   /* cat(to_q,to_k,to_v,proj_mlp)   [(3D+4D) x D] */
  const void *norm_q, *norm_k;
  const void *w_out, *b_out;           /* proj_out [D x 5D]: columns [0,D) attn, [D,5D) mlp */
  rf_lora_seg lora_qkv_mlp, lora_out;
  float qk_bound;                      /* as in rf_double_block_weights */
  int32_t _pad;
  rf_w8 q_qkv_mlp, q_out;              /* cfg5: fp8 copies (q_out columns [attn | mlp] like w_out) */
} rf_single_block_weights;

typedef struct rf_flux_dims {
  int32_t D, heads, mlp;               /* 3072, 24, 12288 */
  int32_t S_txt, S_img, S_cond;        /* token counts; S_cond = 0 without a condition */
  int32_t attn_mode;                   /* rf_attention_fwd mode */
  float cross_bias;                    /* log(c_factor) */
  int32_t lora_on_main;                /* model_config["latent_lora"] */
  int32_t add_cond_attn;               /* model_config["add_cond_attn"] (needs S_cond == S_img) */
  int32_t fp8;                         /* cfg5: run the block GEMMs of every token stream that carries no LoRA on the
                                          fp8 weight copies (rf_gemm_w8a8): activations are quantised per token by
                                          the LayerNorm+modulate kernel (its fp8 form) or by rf_quant_rows_fp8.  Streams
                                          with LoRA (condition rows; image rows under latent_lora) stay bf16. */
  int32_t _pad;
} rf_flux_dims;

typedef struct rf_workspace {
  void* base; int64_t bytes;           /* scratch owned by the caller, 256-byte aligned */
} rf_workspace;

int64_t rf_workspace_bytes(const rf_flux_dims* dims);

/* One DoubleStream block, in place on x_txt/x_img/x_cond  (block.py:173-272).
 *   mod_*: bf16 [6][D] = linear(silu(temb)) chunks in AdaLayerNormZero order
 *          (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp). */
int rf_double_block_fwd(const rf_flux_dims* dims, const rf_double_block_weights* w,
                        void* x_txt, void* x_img, void* x_cond, int64_t ldx,
                        const void* mod_txt, const void* mod_img, const void* mod_cond,
                        const float* cos_tab, const float* sin_tab,
                        const rf_workspace* ws, void* stream);

/* One SingleStream block, in place on x_main = [txt;img] rows and x_cond  (block.py:275-333).
 *   mod_*: bf16 [3][D] = (shift, scale, gate) of AdaLayerNormZeroSingle. */
int rf_single_block_fwd(const rf_flux_dims* dims, const rf_single_block_weights* w,
                        void* x_main, void* x_cond, int64_t ldx,
                        const void* mod_main, const void* mod_cond,
                        const float* cos_tab, const float* sin_tab,
                        const rf_workspace* ws, void* stream);

typedef struct rf_flux_model {
  int32_t num_double, num_single;
  const rf_double_block_weights* dbl;   /* HOST array [num_double] */
  const rf_single_block_weights* sgl;   /* HOST array [num_single] */
  const void *w_x_embed, *b_x_embed;    /* x_embedder   [D x in_ch]   (transformer.py:92) */
  rf_lora_seg lora_x_embed;
  const void *w_ctx_embed, *b_ctx_embed;/* context_embedder [D x joint] (transformer.py:115) */
  const void *w_proj_out, *b_proj_out;  /* proj_out [in_ch x D]        (transformer.py:244) */
  int32_t in_ch, joint_dim;
} rf_flux_model;

/* Per-forward modulation table produced by ONE skinny GEMM over all AdaLN linears
 * (norm1.linear / norm1_context.linear / norm.linear / norm_out.linear; they all see the
 * same temb: transformer.py:166,213).  Layout (bf16, row per call):
 *   [dbl0.img 6D | dbl0.txt 6D | ... | sgl0 3D | ... | norm_out 2D (scale, shift)] */
int64_t rf_mod_table_cols(const rf_flux_model* m, int32_t D);

/* Whole transformer forward (transformer.py:47-252) for one sample:
 *   latents [S_img x in_ch], cond_latents [S_cond x in_ch] or NULL, ctx [S_txt x joint]
 *   mod_main: modulation table row for temb; mod_cond: table row for cond_temb
 *             (cond rows use norm1/norm, i.e. the img/single slots of the table)
 *   out: velocity [S_img x in_ch] */
int rf_flux_forward(const rf_flux_dims* dims, const rf_flux_model* m,
                    const void* latents, const void* cond_latents, const void* ctx,
                    const void* mod_main, const void* mod_cond,
                    const float* cos_tab, const float* sin_tab,
                    void* out, const rf_workspace* ws, void* stream);

/* T-step denoise loop (generate.py:216-296): forward + Euler per step, latents updated in place.
 *   mod_main_steps: [T] table rows (one per timestep); dts: HOST array [T] of sigma_{i+1}-sigma_i */
int rf_flux_denoise(const rf_flux_dims* dims, const rf_flux_model* m,
                    void* latents, const void* cond_latents, const void* ctx,
                    const void* mod_main_steps, int64_t mod_stride, const void* mod_cond,
                    const float* cos_tab, const float* sin_tab,
                    const float* dts, int32_t T, void* vel_scratch,
                    const rf_workspace* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * FLUX VAE (AutoencoderKL, 16 latent channels, 8x) -- SURVEY 8(f) row 1.  Replaces, either side of the denoise loop,
 *   vae.decode(latents / scaling_factor + shift_factor)          generate.py:302-307, tts_reflectionflow.py:273-279
 *   vae.encode(images).latent_dist (the moments mean | logvar)    pipeline_tools.py:7-14, condition.py:96-132
 * (diffusers is not vendored by the reference: the arithmetic is diffusers' published Encoder / Decoder / ResnetBlock2D /
 * Attention / Downsample2D / Upsample2D, GroupNorm(32, eps 1e-6), SiLU -- PARITY UNPINNED at this boundary, see DESIGN.md.)
 * Every image crossing this interface is a ZERO-HALO NHWC bf16 image: (H+2) x (W+2) pixels of C channels, row-major, the
 * one-pixel border zero; C is the layer's PADDED channel count (multiples of 64 on the input side, of 8 on the output side;
 * the host wrapper pads latents 16 -> 64, RGB 3 -> 64 in / 3 -> 8 out and slices the interior of the result).
 * 3x3 convolutions run on the library's grouped MFMA GEMM (three K-segments of 3*Cin contiguous taps), weights repacked once:
 *   conv 3x3 : w bf16 [cout][3 (dy)][3 (dx)][cin], b bf16 [cout]          conv 1x1 / linear : w bf16 [cout][cin]
 * ---------------------------------------------------------------------------------- */
typedef struct rf_vae_conv {                    /* w == NULL: layer absent */
  const void* w; const void* b; int32_t cin, cout;
  /* optional FOLDED copy for narrow outputs (cout < 256): `fold` = g adjacent output pixels share one GEMM row, so the launch
   * is N = g*cout wide (a full 256-column MFMA tile instead of a half / 1/32 empty one) over K = 3 x (g+2)*cin:
   *   wf bf16 [g*cout][3][(g+2)*cin],  wf[j*cout + o][dy][(j+dx)*cin + i] = w[o][dy][dx][i] (zero elsewhere),  bf = b tiled g times.
   * The output memory is unchanged ([pixels][cout] row-major IS [pixels/g][g*cout]).  Used when the image is wide enough
   * (g <= W + 3); wf == NULL: never. */
  const void* wf; const void* bf; int32_t fold; int32_t _pad;
} rf_vae_conv;
typedef struct rf_vae_norm { const void* gamma; const void* beta; } rf_vae_norm;                 /* GroupNorm affine, bf16 [C] */
typedef struct rf_vae_resnet {                  /* ResnetBlock2D: x' = shortcut(x) + conv2(silu(norm2(conv1(silu(norm1(x)))))) */
  rf_vae_norm norm1; rf_vae_conv conv1; rf_vae_norm norm2; rf_vae_conv conv2;
  rf_vae_conv shortcut;                         /* 1x1 conv_shortcut when cin != cout, else w == NULL */
} rf_vae_resnet;
typedef struct rf_vae_attn {                    /* mid-block Attention: one head of C channels, residual */
  rf_vae_norm norm;
  const void* w_qk; const void* b_qk;           /* cat(to_q, to_k) [2C x C], [2C] */
  const void* w_v;                              /* to_v.weight [C x C]; its bias is folded into b_out (softmax rows sum to 1) */
  const void* w_out; const void* b_out;         /* to_out.0 [C x C]; b_out = to_out.0.bias + to_out.0.weight . to_v.bias */
  int32_t C; int32_t _pad;
} rf_vae_attn;
typedef struct rf_vae_weights {                 /* ONE direction: the Decoder or the Encoder */
  int32_t levels, res_per_level;                /* 4 x 3 (decoder: layers_per_block + 1) or 4 x 2 (encoder) */
  int32_t groups, has_attn;                     /* GroupNorm groups (32); mid_block_add_attention */
  rf_vae_conv conv_in;
  rf_vae_resnet mid0, mid1; rf_vae_attn attn;   /* mid block: resnet, attention, resnet */
  rf_vae_resnet res[4][3];                      /* [level][resnet] in execution order */
  rf_vae_conv resample[4];                      /* after level i: decoder = upsampler conv (3x3 on the nearest-2x image),
                                                   encoder = downsampler conv (3x3, stride 2, pad (0,1,0,1)); w NULL = none */
  rf_vae_norm norm_out; rf_vae_conv conv_out;
} rf_vae_weights;
/* workspace (caller-owned, 256-byte aligned) for an input of h x w pixels (decode: latent size, encode: image size) */
int64_t rf_vae_workspace_bytes(const rf_vae_weights* w, int32_t encode, int32_t h, int32_t w_in);
/* INPUT SLACK (both entry points): conv_in runs as a GEMM whose 64-element K-tiles reach past the last pixel's channels against zero
 * weight columns, so the input buffer must be followed by >= 64 * max(conv_in.cin, 64) * 2 readable bytes holding FINITE values
 * (zeros): a NaN / Inf there does not multiply away.  (reflectionflow_amd/flux/vae_hip.py::_padded_input allocates it that way.)
 * z: zero-halo [(h+2)(w+2)][conv_in.cin]  ->  out: [(8h+2)(8w+2)][conv_out.cout]: only the INTERIOR of `out` is defined (its top /
 * bottom halo rows are never written, its halo columns receive garbage); the caller slices channels 0..2 of the interior */
int rf_vae_decode(const rf_vae_weights* w, const void* z, int32_t h, int32_t w_in, void* out, const rf_workspace* ws, void* stream);
/* img: zero-halo [(H+2)(W+2)][conv_in.cin] -> out: [(H/8+2)(W/8+2)][conv_out.cout] = (mean | logvar) moments, interior valid */
int rf_vae_encode(const rf_vae_weights* w, const void* img, int32_t H, int32_t W, void* out, const rf_workspace* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * Text encoders: what FluxPipeline.encode_prompt runs for every candidate's prompt
 * (train_flux/flux/generate.py:148-161; per candidate and round in tts/tts_reflectionflow.py:286-294):
 *   prompt_embeds        = T5EncoderModel(t5_ids [512])[0]              -> rf_t5_encode
 *   pooled_prompt_embeds = CLIPTextModel(clip_ids [77]).pooler_output   -> rf_clip_text_encode
 * B sequences of S token ids (int32, device) per call -- a rank's candidates of a round share every GEMM launch; no attention mask, as in the reference's calls (T5 attends over all padded
 * positions, CLIP is causal).  All weights bf16, head dim 64, S <= 512.  Every projection is an rf_gemm_bf16 launch (residual adds in
 * its epilogue); attention is one kernel with the head's K and V^T resident in LDS and an additive fp32 bias.  The algorithm is
 * transformers' (requirements.txt:2); parity is pinned against transformers 5.15.0 (oracle/text_oracle.py, tests/golden/text_encoders.npz).
 * ---------------------------------------------------------------------------------- */
typedef struct rf_t5_layer {
  const void* ln0;    /* layer.0.layer_norm.weight [d_model] (RMS norm: no mean subtraction, no bias) */
  const void* w_qk;   /* cat(SelfAttention.q, .k) [2 * heads * 64][d_model]; T5 does not scale q.k */
  const void* w_v;    /* SelfAttention.v [heads * 64][d_model]: used as the A operand (V^T = W_v x^T) */
  const void* w_o;    /* SelfAttention.o [d_model][heads * 64] */
  const void* ln1;    /* layer.1.layer_norm.weight */
  const void* w_wi;   /* cat(DenseReluDense.wi_0, .wi_1) [2 * d_ff][d_model]: gelu_new(wi_0 x) * (wi_1 x) */
  const void* w_wo;   /* DenseReluDense.wo [d_model][d_ff] */
} rf_t5_layer;
typedef struct rf_t5_weights {
  int32_t layers, d_model, heads, d_kv;   /* d_kv must be 64 */
  int32_t d_ff, vocab;
  float eps; int32_t bias_S;              /* layer_norm_epsilon (1e-6); pos_bias is built for S_pad = bias_S */
  const void* embed;                      /* shared.weight [vocab][d_model] */
  const float* pos_bias;                  /* [heads][bias_S][bias_S] fp32: relative_attention_bias of layer 0 gathered through the
                                             bidirectional buckets (32 buckets, max distance 128), -inf in columns >= S */
  const void* final_ln;                   /* encoder.final_layer_norm.weight */
  const rf_t5_layer* layer;               /* HOST array [layers] */
} rf_t5_weights;
int64_t rf_t5_workspace_bytes(const rf_t5_weights* w, int32_t B, int32_t S);
/* ids [B][S] -> out [B][S][ld_out] bf16 = last hidden state (final norm applied); 1 <= B <= 64 sequences share every GEMM launch */
int rf_t5_encode(const rf_t5_weights* w, const int32_t* ids, int32_t B, int32_t S, void* out, int64_t ld_out, const rf_workspace* ws, void* stream);

typedef struct rf_clip_layer {
  const void *ln1_scale, *ln1_shift;      /* layer_norm1 as (weight - 1, bias): LN(x) * (1 + scale) + shift */
  const void *w_qk, *b_qk;                /* cat(q_proj / 8, k_proj) [2 * hidden][hidden] (+ biases; 1/sqrt(64) folded into q) */
  const void* w_v;                        /* v_proj.weight [hidden][hidden] (A operand); its bias is folded into b_o */
  const void *w_o, *b_o;                  /* out_proj; b_o = out_proj.bias + out_proj.weight . v_proj.bias */
  const void *ln2_scale, *ln2_shift;
  const void *w_fc1, *b_fc1, *w_fc2, *b_fc2;   /* mlp: fc2(quick_gelu(fc1 x)) */
} rf_clip_layer;
typedef struct rf_clip_weights {
  int32_t layers, hidden, heads, inter;
  int32_t vocab, max_pos;
  float eps; int32_t mask_S;              /* layer_norm_eps (1e-5); mask is built for S_pad = mask_S */
  const void *tok_embed, *pos_embed;      /* [vocab][hidden], [max_pos][hidden] */
  const float* mask;                      /* [mask_S][mask_S] fp32: 0 on / below the diagonal, -inf above and in columns >= S */
  const void *final_ln_scale, *final_ln_shift;
  const rf_clip_layer* layer;             /* HOST array [layers] */
} rf_clip_weights;
int64_t rf_clip_text_workspace_bytes(const rf_clip_weights* w, int32_t B, int32_t S);
/* ids [B][S] -> last_hidden [B][S][hidden] (may be NULL) and pooled [B][hidden] (may be NULL) = final-normed row eos_pos[b] of
 * sequence b (eos_pos: HOST array [B]) */
int rf_clip_text_encode(const rf_clip_weights* w, const int32_t* ids, int32_t B, int32_t S, const int32_t* eos_pos, void* last_hidden, void* pooled,
                        const rf_workspace* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * TRAINING path: backward of the same blocks, for the LoRA reflection tuning of
 * train_flux/train/model.py:164-238 (flow-matching MSE through tranformer_forward; per-block recompute = the
 * gradient-checkpoint branch train_flux/flux/transformer.py:139-157).  Base weights are frozen, so a block's backward is
 *   dX through every GEMM      = rf_gemm_bf16 on TRANSPOSED weight copies  (dX = dY W  ==  dY (W^T)^T)
 *   LoRA factor gradients      = rf_gemm_tn_skinny (contraction over the TOKEN axis, operands as they lie)
 *   everything else            = the kernels below.
 * Gradients travel between kernels as bf16 (as torch's bf16 autograd does), sums are fp32, column reductions are
 * fixed-order two-stage sums: the whole backward is bit-reproducible (no atomics).
 * ---------------------------------------------------------------------------------- */

/* The attention operands of a training step from the RAW q|k|v rows (the QKV GEMM with RF_EPI_STORE, [S][ld_raw],
 * columns [q | k | v] of heads*128 each): per-head RMSNorm + RoPE as rf_qk_rmsnorm_rope (rows < n_added use w_added_*),
 * q scaled by q_scale (softmax_scale * log2 e), written as
 *   q, k, v : [heads][s_pad][128] head-major rows (rows >= S zero),
 *   vt      : rf_attention's V^T tiles,
 *   qt, kt  : [heads][s_pad/32][128][32] TRANSPOSED TILES for rf_attention_bwd: element (d, slot(n)) of block n/32 holds
 *             x[n][d], slot(n) = 8 ((n%16)/4) + 4 ((n%32)/16) + n%4 -- the order in which a 16x16x32 MFMA consumes the
 *             rows of two 16x16 score tiles.
 * v, qt, kt are the backward's operands: all three NULL = a forward nobody differentiates (the no-grad pass of a checkpointed
 * block, train_flux/flux/transformer.py:139-157) writes q, k, vt only.  Giving some but not all is RF_ERR_NULL. */
int rf_qkv_train_fwd(const void* raw, int64_t ld_raw, int32_t heads, int32_t S, int32_t s_pad, int32_t n_added,
                     const void* w_q, const void* w_k, const void* w_added_q, const void* w_added_k,
                     const float* cos_tab, const float* sin_tab, float eps, float q_scale,
                     void* q, void* k, void* v, void* vt, void* qt, void* kt, void* stream);
/* Its backward: (dq, dk, dv) head-major [heads][s_pad][128] (dq = gradient w.r.t. the SCALED q) -> d_raw [S][ld_draw]. */
int rf_qkv_train_bwd(const void* raw, int64_t ld_raw, int32_t heads, int32_t S, int32_t s_pad, int32_t n_added,
                     const void* w_q, const void* w_k, const void* w_added_q, const void* w_added_k,
                     const float* cos_tab, const float* sin_tab, float eps, float q_scale,
                     const void* dq, const void* dk, const void* dv, void* d_raw, int64_t ld_draw, void* stream);

/* Flash-attention backward (F.scaled_dot_product_attention of block.py:123-125), plain joint attention (mode 0):
 *   q (scaled), k, v, qt, kt from rf_qkv_train_fwd;  o = the forward's output, dout = its gradient, both [S][ld] token-major;
 *   dq, dk, dv: [heads][s_pad][128] bf16;  dot ([heads][s_pad/32][128][32] bf16), lse, dsum (fp32 [heads][s_pad]): scratch.
 * Three launches: D = rowsum(dO o O) + dO^T tiles; per 64 queries the row statistics then dq; per 128 keys dk, dv. */
typedef enum rf_attn_bwd_kernel {
  RF_ATTN_BWD_AUTO = 0,          /* per kernel: the form with the fewest CU rounds x rows for this (heads, s_pad)        */
  RF_ATTN_BWD_DQ_256 = 1,        /* dq: 8 waves x 32 queries per workgroup                                             */
  RF_ATTN_BWD_DQ_128 = 2,        /* dq: 4 waves x 32 queries                                                           */
  RF_ATTN_BWD_DQ_192 = 3,        /* dq: 4 waves x 48 queries                                                           */
  RF_ATTN_BWD_DKV_128 = 1 << 8,  /* dK / dV: 128 keys per workgroup, one workgroup per CU (4-slot ring)                 */
  RF_ATTN_BWD_DKV_192 = 2 << 8,  /* dK / dV: 192 keys per workgroup                                                    */
  RF_ATTN_BWD_DKV_128X2 = 3 << 8 /* dK / dV: 128 keys per workgroup, two workgroups per CU (2-slot ring)               */
} rf_attn_bwd_kernel;
typedef struct rf_attn_bwd_desc {
  const void *q, *k, *v, *qt, *kt;
  const void *o, *dout; int64_t ldo, lddo;
  void *dq, *dk, *dv;
  void* dot; float* lse; float* dsum;
  int32_t heads, S, s_pad, mode;
  int32_t lse_given;   /* 1: lse[heads][s_pad] holds the forward's row statistics for rows < S (rf_attn_desc.lse of the SAME q, k): the
                          dq kernel skips its own statistics pass over the keys (a quarter of its work).  0: lse is scratch. */
  int32_t kernel;      /* rf_attn_bwd_kernel: RF_ATTN_BWD_AUTO, or one dq form | one dK / dV form.  Every form gives the same gradients to
                          rounding; the forms differ in queries / keys per workgroup, i.e. in how a (heads, S) fills the CUs. */
} rf_attn_bwd_desc;
int rf_attention_bwd(const rf_attn_bwd_desc* d, void* stream);

/* y = LayerNorm(x) (1 + scale) + shift  (rf_layernorm_modulate):  dx[m] = (dres ? dres[m] : 0) + LN-backward(dy[m]);
 * d_scale[c] = sum_m dy[m,c] xhat[m,c], d_shift[c] = sum_m dy[m,c]  (fp32 [D]; both NULL = not wanted).  partials: scratch of
 * rf_train_partials_bytes(D) bytes.  D % 8 == 0, D <= 3072. */
int64_t rf_train_partials_bytes(int32_t D);
int rf_layernorm_modulate_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* dres, int64_t lddres,
                              void* dx, int64_t lddx, int32_t rows, int32_t D, const void* scale, float eps,
                              float* d_scale, float* d_shift, float* partials, int64_t partials_bytes, void* stream);
/* y = residual + gate o f  (RF_EPI_GATE_RES):  df = gate o dy (bf16),  d_gate[c] = sum_m dy[m,c] f[m,c] (fp32 [D]). */
int rf_gate_bwd(const void* dy, int64_t lddy, const void* f, int64_t ldf, const void* gate, void* df, int64_t lddf,
                int32_t rows, int32_t D, float* d_gate, float* partials, int64_t partials_bytes, void* stream);
/* out = res + gate o f, the gated residual as its own pass (the training forward keeps f for rf_gate_bwd) */
int rf_gate_residual(const void* f, int64_t ldf, const void* gate, const void* res, int64_t ldr, void* out, int64_t ldo,
                     int32_t rows, int32_t D, void* stream);
/* h = gelu_tanh(z);  dz = dh * gelu_tanh'(z)   (block.py:252,296) */
int rf_gelu(const void* z, int64_t ldz, void* h, int64_t ldh, int32_t rows, int32_t cols, void* stream);
int rf_gelu_bwd(const void* z, int64_t ldz, const void* dh, int64_t lddh, void* dz, int64_t lddz, int32_t rows, int32_t cols,
                void* stream);
/* Token-axis ("TN") skinny GEMM -- the LoRA factor gradients dB = dY^T T and dA = (x^T dT)^T (train/model.py's autograd through
 * peft's lora_A / lora_B, lora_controller.py:5-42):
 *   out[n][j] = sum_{s < S} big[s][n] * skinny[s][j]      n < N (N % 8 == 0), j < R (R % 16 == 0)
 * big [S][ld_big], skinny [S][ld_sk] row-major bf16; out bf16 [N][ld_out], or [R][ld_out] when `transposed`.  fp32 partial sums per
 * 128-token chunk in `ws` (rf_gemm_tn_skinny_ws_bytes), added in chunk order: bit-reproducible.  No operand is transposed in HBM. */
int64_t rf_gemm_tn_skinny_ws_bytes(int32_t S, int32_t N, int32_t R);
int rf_gemm_tn_skinny(const void* big, int64_t ld_big, const void* skinny, int64_t ld_sk, void* out, int64_t ld_out, int32_t S,
                      int32_t N, int32_t R, int32_t transposed, float* ws, int64_t ws_bytes, void* stream);
/* dst[c][r] = src[r][c], r < rows; zero for rows <= r < rows_pad (the K % 64 padding of a token-axis contraction) */
int rf_transpose_bf16(const void* src, int64_t ld_src, int32_t rows, int32_t cols, void* dst, int64_t ld_dst,
                      int32_t rows_pad, void* stream);

/* The fused LoRA operands of sibling linears and the way back (ABI v14; reference: peft's per-linear lora_A / lora_B applied one by one
 * inside F.linear calls, train_flux/flux/lora_controller.py + block.py:23-30,146-155).  The training path applies the LoRA of linears
 * that share an input as ONE K-segment (A [r_pad][K]) and one block-diagonal up-projection (Bs [N][r_pad]):
 *   rf_lora_fuse          A rows r0 .. r0+r <- lora_A of entry i ([r][K]), Bs block (n0 .. n0+n, r0 .. r0+r) <- scaling * lora_B
 *                         ([n][r], rounded to bf16), everything else zero -- one launch instead of two fills + a copy and a scaled
 *                         copy per linear;
 *   rf_lora_unfuse_grads  its backward: dA_i (+)= dA rows, dB_i (+)= bf16(scaling * dBs block) -- the product is rounded to bf16 first
 *                         and the sum once more, exactly what `grad += dBs_block * scaling` on bf16 tensors does -- one launch
 *                         instead of a scaled copy and two accumulations per linear.  accumulate = 0 overwrites.
 * Up to RF_LORA_FUSE_MAX entries, K % 8 == 0, r_pad % 8 == 0, 16-byte aligned A / dA pointers. */
#define RF_LORA_FUSE_MAX 8
typedef struct rf_lora_fuse_entry {
  const void* A;      /* lora_A.weight [r][K] bf16 (fuse: read) */
  const void* B;      /* lora_B.weight [n][r] bf16 (fuse: read) */
  void* dA;           /* gradient of A [r][K] bf16 (unfuse: written / accumulated; may be NULL = not wanted) */
  void* dB;           /* gradient of B [n][r] bf16 (same) */
  int32_t n0, n;      /* rows of Bs this linear owns */
  int32_t r0, r;      /* rows of A / columns of Bs this (linear, adapter) owns */
  float scaling;      /* lora_alpha / r */
  int32_t _pad;
} rf_lora_fuse_entry;
int rf_lora_fuse(const rf_lora_fuse_entry* entries, int32_t n_entries, int32_t K, int32_t N, int32_t r_pad, void* A_out, void* Bs_out,
                 void* stream);
int rf_lora_unfuse_grads(const rf_lora_fuse_entry* entries, int32_t n_entries, int32_t K, int32_t N, int32_t r_pad, const void* dA,
                         const void* dBs, int32_t accumulate, void* stream);

/* Optimizer update of the LoRA factors (train_flux/train/model.py:105-117: torch.optim.AdamW or prodigyopt.Prodigy over the LoRA
 * parameters; config.yaml:55-61 ships Prodigy lr 1, use_bias_correction, safeguard_warmup, weight_decay 0.01) over ONE flat bucket:
 *   param, grad : bf16 [n] (every LoRA factor / its gradient is a view into them; n % 8 == 0);
 *   exp_avg, exp_avg_sq (, s) : bf16 [n] (what torch / prodigyopt keep for bf16 parameters) or fp32 [n] when state_fp32;
 *   grad_scale  : multiplies every gradient on the way in (1 / world_size behind a SUM all-reduce: the averaging pass is fused away).
 * rf_lora_adamw = one step of torch.optim.AdamW (decoupled decay, bias correction, amsgrad off); `step` is 1-based.
 * rf_lora_prodigy = one step of Prodigy's Adam form (Mishchenko & Defazio 2023, Algorithm 4; option names and defaults of prodigyopt,
 * which is not available offline: PARITY UNPINNED, restated in oracle/optim_oracle.py):  p0 = the parameters at step 0 (bf16 [n]);
 * dstate = 8 doubles ON THE DEVICE {d, d_max, d_numerator, d_denom, d_hat, k, dlr of the last step, d0}, initialised by the caller to
 * {d0, d0, 0, 0, d0, 0, 0, d0}: the distance estimate never crosses to the host inside a step (the package reads one .item() per
 * parameter tensor), the two global sums are fixed-order two-stage sums over `partials` (rf_lora_prodigy_partials_bytes).
 * beta3 <= 0 means sqrt(beta2).  Three launches: moments + partial sums, the new d (one workgroup), the parameter update. */
int rf_lora_adamw(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, int32_t state_fp32, int32_t step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
int64_t rf_lora_prodigy_partials_bytes(int64_t n);
int rf_lora_prodigy(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, void* s, const void* p0, int64_t n,
                    int32_t state_fp32, double* dstate, float lr, float beta1, float beta2, float beta3, float eps,
                    float weight_decay, int32_t decouple, int32_t use_bias_correction, int32_t safeguard_warmup, float d_coef,
                    float growth_rate, float grad_scale, float* partials, int64_t partials_bytes, void* stream);

/* Gradient clipping by global L2 norm over the flat gradient bucket (ABI v15): the reference's Lightning Trainer runs with
 * gradient_clip_val = 0.5 (train_flux/train/train.py:165), i.e. torch.nn.utils.clip_grad_norm_(params, 0.5) between the DDP all-reduce and
 * optimizer.step:  total_norm = || grad_scale * grad ||_2,  coef = min(1, max_norm / (total_norm + 1e-6)),  grad *= coef  (in place, bf16).
 * `out` = 2 floats ON THE DEVICE {total_norm, coef}: nothing crosses to the host.  `partials`: >= 4 bytes per 8192 elements
 * (rf_lora_prodigy_partials_bytes(n) is enough).  Fixed-order sums: bit-reproducible.  Three launches. */
int rf_lora_clip_grad_norm(void* grad, int64_t n, float max_norm, float grad_scale, float* partials, int64_t partials_bytes, float* out,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RF_FLUX_H */
