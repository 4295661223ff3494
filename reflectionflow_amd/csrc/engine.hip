// Block / forward / denoise sequencing for the FLUX MM-DiT hot path on one MI355X.
//
// A transformer forward is a FIXED sequence of ~350 kernel launches on one HIP stream, with no
// allocation and no host synchronisation inside (hipGraph-capturable): 7 launches per
// DoubleStream block, 4 per SingleStream block (DESIGN.md "step anatomy").  All activations
// live in a caller-owned workspace laid out once per (S_txt, S_img, S_cond):
//
//   XN  [S][D]        LayerNorm+modulate output (GEMM A operand)
//   Q,K [H][S_pad][128]   VT [H][S_pad/64][128][64]   joint attention operands
//   ATT [S][D]        attention output, token-major
//   HID [S][mlp]      FF hidden / single-block MLP branch
//   LT  [S][256]      LoRA low-rank intermediates (x . lora_A^T)
//   SK  fp32          GEMM scratch: stream-K flags + per-CU partial tiles, LoRA split-K partials (64 MiB)
//   X   [S][D]        residual stream (rf_flux_forward only; rows = txt | img | cond)
//   XN8 [S][D] fp8, A8 [S][D+mlp] fp8, sXN / sA [S] fp32   cfg5 (dims.fp8): quantised GEMM inputs + their row scales
//
// Token order everywhere is [text | image | condition] -- the order the reference concatenates
// them in (block.py:70-72,101-104).
#include "common.hpp"

namespace rf {

struct WsLayout {
  int S, s_pad;
  int64_t xn, q, k, vt, att, hid, lt, x, sk, sk_bytes, total;  // byte offsets
  int64_t xn8, a8, s_xn, s_a;  // cfg5 (dims.fp8): fp8 LN-mod output [S][D], fp8 GEMM input [S][D+mlp], row scales [S] fp32 x2
};

static WsLayout ws_layout(const rf_flux_dims& d) {
  WsLayout L;
  L.S = d.S_txt + d.S_img + d.S_cond;
  L.s_pad = (int)round_up(L.S, 64);
  int64_t off = 0;
  auto take = [&](int64_t elems) {
    const int64_t o = off;
    off += round_up(elems * 2, 256);
    return o;
  };
  const int64_t SD = (int64_t)L.S * d.D;
  const int64_t HS = (int64_t)d.heads * L.s_pad * 128;
  L.xn = take(SD);
  L.q = take(HS);
  L.k = take(HS);
  L.vt = take(HS);
  L.att = take(SD);
  L.hid = take((int64_t)L.S * d.mlp);
  L.lt = take((int64_t)L.S * 256);
  L.x = take(SD);
  // GEMM scratch: 4 KiB of stream-K flags (must be zero before the first launch; every launch restores them),
  // then fp32 partial tiles: one 256x256 slot per CU for stream-K (64 MiB at 256 CUs), reused by the LoRA split-K
  // (attention's split launch borrows it too: 2 x 256 slots of 132 KiB = 66 MiB; the kernels run one after the other)
  L.sk_bytes = 4096 + (68ll << 20);
  L.sk = take(L.sk_bytes / 2);
  L.xn8 = L.a8 = L.s_xn = L.s_a = 0;
  if (d.fp8) {
    L.xn8 = take(cdiv64(SD, 2));                                   // 1 byte per element
    L.a8 = take(cdiv64((int64_t)L.S * (d.D + d.mlp), 2));
    L.s_xn = take((int64_t)L.S * 2);                               // fp32 per row
    L.s_a = take((int64_t)L.S * 2);
  }
  L.total = off;
  return L;
}

static int check_dims(const rf_flux_dims* d, const rf_workspace* ws, WsLayout& L) {
  RF_REQUIRE(d && ws, RF_ERR_NULL, "rf engine: dims/workspace NULL");
  RF_REQUIRE(d->D == d->heads * 128 && d->D % 64 == 0 && d->mlp % 256 == 0 && d->mlp > 0, RF_ERR_SHAPE,
             "rf engine: need D == heads*128 and mlp %% 256 == 0 (D=%d heads=%d mlp=%d)", d->D, d->heads, d->mlp);
  RF_REQUIRE(d->S_txt >= 0 && d->S_img > 0 && d->S_cond >= 0, RF_ERR_SHAPE, "rf engine: bad token counts");
  RF_REQUIRE(!d->add_cond_attn || d->S_cond == 0 || d->S_cond == d->S_img, RF_ERR_SHAPE,
             "rf engine: add_cond_attn needs S_cond == S_img (block.py:227-228)");
  L = ws_layout(*d);
  RF_REQUIRE(ws->base != nullptr && aligned16(ws->base), RF_ERR_NULL, "rf engine: workspace base NULL/unaligned");
  RF_REQUIRE(ws->bytes >= L.total, RF_ERR_WORKSPACE, "rf engine: workspace %lld < required %lld bytes",
             (long long)ws->bytes, (long long)L.total);
  return RF_OK;
}

static inline bf16_t* at(const rf_workspace* ws, int64_t byte_off) { return (bf16_t*)((char*)ws->base + byte_off); }

static void set_seg(rf_kseg& s, const void* A, int64_t lda, const void* W, int64_t ldw, int K) {
  s.A = A; s.lda = lda; s.W = W; s.ldw = ldw; s.K = K; s._pad = 0;
}

// softmax scale 1/sqrt(128) times log2(e): the QKV epilogue folds it into q, attention then works in the exp2 domain
static constexpr float QK_PRESCALE = 0.08838834764831845f * 1.4426950408889634f;

#define RF_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != RF_OK) return _rc; \
  } while (0)

// split-K / stream-K scratch of the GEMMs (flags + fp32 partial tiles)
static inline void attach_scratch(rf_gemm_desc& d, const rf_workspace* ws, const WsLayout& L) {
  d.splitk_ws = (char*)ws->base + L.sk;
  d.splitk_ws_bytes = L.sk_bytes;
}

// LoRA intermediate T = A_act . lora_A^T  ([M x r_pad]); two activation segments for proj_out.
static int lora_down(const rf_lora_seg& l, const bf16_t* a0, int64_t lda0, int K0, const bf16_t* a1, int64_t lda1,
                     int K1, int M, bf16_t* T, const rf_workspace* ws, const WsLayout& L, hipStream_t st) {
  rf_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.N = l.r_pad; d.epilogue = RF_EPI_STORE; d.num_groups = 1;
  attach_scratch(d, ws, L);
  rf_gemm_group& g = d.g[0];
  const int64_t ldA = (int64_t)K0 + K1;  // lora_A rows span the concatenated input
  set_seg(g.seg[0], a0, lda0, l.A, ldA, K0);
  if (K1 > 0) set_seg(g.seg[1], a1, lda1, (const bf16_t*)l.A + K0, ldA, K1);
  g.M = M; g.out = T; g.ldo = 256;
  return rf_gemm_bf16(&d, st);
}

}  // namespace rf

using namespace rf;

extern "C" int64_t rf_workspace_bytes(const rf_flux_dims* dims) {
  if (!dims) return RF_ERR_NULL;
  return ws_layout(*dims).total;
}

// =================================================================================================
// DoubleStream block (block.py:173-272)
// =================================================================================================
extern "C" int rf_double_block_fwd(const rf_flux_dims* dims, const rf_double_block_weights* w, void* x_txt,
                                   void* x_img, void* x_cond, int64_t ldx, const void* mod_txt, const void* mod_img,
                                   const void* mod_cond, const float* cos_tab, const float* sin_tab,
                                   const rf_workspace* ws, void* stream) {
  WsLayout L;
  RF_TRY(check_dims(dims, ws, L));
  RF_REQUIRE(w && x_img && mod_img && cos_tab && sin_tab, RF_ERR_NULL, "rf_double_block_fwd: NULL pointer");
  RF_REQUIRE(dims->S_txt == 0 || (x_txt && mod_txt), RF_ERR_NULL, "rf_double_block_fwd: text stream NULL");
  RF_REQUIRE(dims->S_cond == 0 || (x_cond && mod_cond), RF_ERR_NULL, "rf_double_block_fwd: condition stream NULL");
  hipStream_t st = (hipStream_t)stream;
  const int D = dims->D, H = dims->heads, MLP = dims->mlp;
  const int St = dims->S_txt, Si = dims->S_img, Sc = dims->S_cond, S = L.S;
  const int o_txt = 0, o_img = St, o_cond = St + Si;
  bf16_t* XN = at(ws, L.xn); bf16_t* Q = at(ws, L.q); bf16_t* K = at(ws, L.k); bf16_t* VT = at(ws, L.vt);
  bf16_t* ATT = at(ws, L.att); bf16_t* HID = at(ws, L.hid); bf16_t* LT = at(ws, L.lt);
  const bf16_t* mt = (const bf16_t*)mod_txt; const bf16_t* mi = (const bf16_t*)mod_img; const bf16_t* mc = (const bf16_t*)mod_cond;
  struct Stream { bf16_t* x; const bf16_t* mod; int rows, off; bool lora; };
  // LoRA: condition rows always, image rows only when latent_lora (lora_controller.py:5-42); text never
  Stream sx[3] = {{(bf16_t*)x_txt, mt, St, o_txt, false},
                  {(bf16_t*)x_img, mi, Si, o_img, dims->lora_on_main != 0},
                  {(bf16_t*)x_cond, mc, Sc, o_cond, true}};

  // cfg5 (dims.fp8): a stream WITHOUT LoRA runs its GEMMs on the fp8 weight copies; its LayerNorm+modulate output and its
  // attention / FF-hidden rows are quantised per token on the way in.  LoRA'd streams (condition; image under
  // latent_lora) keep the bf16 kernels: their low-rank K-segment is bf16 (rf_gemm_w8a8 header note).
  const bool have8 = dims->fp8 && w->q_qkv.w && w->q_add_qkv.w && w->q_out.w && w->q_add_out.w && w->q_ff1.w && w->q_ff2.w &&
                     w->q_ffc1.w && w->q_ffc2.w;
  RF_REQUIRE(!dims->fp8 || have8, RF_ERR_NULL, "rf_double_block_fwd: dims.fp8 set but the block has no fp8 weight copies");
  const bool any_lora = w->lora_qkv.B || w->lora_out.B || w->lora_ff2.B;
  const bool use8[3] = {have8, have8 && !(sx[1].lora && any_lora), false};
  uint8_t* XN8 = (uint8_t*)ws->base + L.xn8;
  uint8_t* A8 = (uint8_t*)ws->base + L.a8;
  float* sXN = (float*)((char*)ws->base + L.s_xn);
  float* sA = (float*)((char*)ws->base + L.s_a);

  // one GEMM stage = ONE grouped launch: fp8 streams as fp8 groups (a_scale set), the others as bf16 groups of the same
  // mixed-precision launch (rf_gemm_w8a8); with no fp8 stream it is the plain rf_gemm_bf16 launch of rounds 1
  auto run_stage = [&](const rf_gemm_desc& base, auto&& fill) -> int {
    rf_gemm_desc d = base;
    int n = 0;
    bool any8 = false;
    for (int i = 0; i < 3; ++i) {
      if (sx[i].rows <= 0) continue;
      RF_TRY(fill(i, d.g[n], use8[i]));
      any8 = any8 || use8[i];
      ++n;
    }
    if (n == 0) return RF_OK;
    d.num_groups = n;
    attach_scratch(d, ws, L);
    return any8 ? rf_gemm_w8a8(&d, st) : rf_gemm_bf16(&d, st);
  };
  // LayerNorm + modulate of every stream with (scale, shift) = mod rows (r_scale, r_shift)
  auto ln_mod = [&](int r_scale, int r_shift) -> int {
    LnModStream bf[3];
    int nbf = 0;
    for (int i = 0; i < 3; ++i) {
      const Stream& s = sx[i];
      if (s.rows <= 0) continue;
      if (use8[i])
        RF_TRY(rf_layernorm_modulate_fp8(s.x, ldx, XN8 + (int64_t)s.off * D, D, sXN + s.off, s.rows, D, s.mod + r_scale * D,
                                         s.mod + r_shift * D, 1e-6f, st));
      else
        bf[nbf++] = LnModStream{s.x, XN + (int64_t)s.off * D, s.mod + r_scale * D, s.mod + r_shift * D, s.rows};
    }
    return ln_mod_grouped(bf, nbf, ldx, D, D, 1e-6f, st);   // every bf16 stream in ONE launch (row for row rf_layernorm_modulate)
  };
  // per-token fp8 copy of the rows [off, off+rows) of a bf16 [S][K] buffer into A8 (leading dimension K)
  auto quant_rows = [&](const bf16_t* src, int K) -> int {
    for (int i = 0; i < 3; ++i)
      if (use8[i] && sx[i].rows > 0)
        RF_TRY(rf_quant_rows_fp8(src + (int64_t)sx[i].off * K, K, K, nullptr, 0, 0, A8 + (int64_t)sx[i].off * K, K, sA + sx[i].off,
                                 sx[i].rows, st));
    return RF_OK;
  };

  // 1. AdaLN-Zero: XN = LN(x)*(1+scale_msa)+shift_msa  (mod rows: 0 shift_msa, 1 scale_msa, 2 gate_msa,
  //    3 shift_mlp, 4 scale_mlp, 5 gate_mlp)
  RF_TRY(ln_mod(1, 0));

  // 2. QKV projections, written head-major into the joint [txt|img|cond] attention operands
  {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = 3 * D; d.epilogue = RF_EPI_QKV; d.q = Q; d.k = K; d.vt = VT; d.heads = H; d.s_pad = L.s_pad;
    d.rope_cos = cos_tab; d.rope_sin = sin_tab; d.norm_eps = 1e-6f;   // per-head RMSNorm + RoPE fused into the epilogue
    d.q_scale = QK_PRESCALE;                                          // ... and the softmax scale folded into q
    RF_TRY(run_stage(d, [&](int i, rf_gemm_group& g, bool f8) -> int {
      const Stream& s = sx[i];
      const bool txt = (i == 0);
      g.M = s.rows; g.tok_offset = s.off;
      g.norm_q = txt ? w->norm_added_q : w->norm_q;
      g.norm_k = txt ? w->norm_added_k : w->norm_k;
      g.bias = txt ? w->b_add_qkv : w->b_qkv;
      if (f8) {
        const rf_w8& q8 = txt ? w->q_add_qkv : w->q_qkv;
        set_seg(g.seg[0], XN8 + (int64_t)s.off * D, D, q8.w, D, D);
        g.a_scale = sXN + s.off; g.w_scale = q8.scale;
        return RF_OK;
      }
      const bool mrg_lora_qkv = !txt && s.lora && w->lora_qkv.B && w->lora_qkv.merged;   // LoRA folded into a per-group weight copy (W + s B A)
      set_seg(g.seg[0], XN + (int64_t)s.off * D, D, mrg_lora_qkv ? (const bf16_t*)w->lora_qkv.B : (const bf16_t*)(txt ? w->w_add_qkv : w->w_qkv), D, D);
      if (!txt && s.lora && w->lora_qkv.B && !w->lora_qkv.merged) {
        bf16_t* T = LT + (int64_t)s.off * 256;
        RF_TRY(lora_down(w->lora_qkv, XN + (int64_t)s.off * D, D, D, nullptr, 0, 0, s.rows, T, ws, L, st));
        set_seg(g.seg[1], T, 256, w->lora_qkv.B, w->lora_qkv.r_pad, w->lora_qkv.r_pad);
      }
      return RF_OK;
    }));
  }
  // 3. per-head RMSNorm(q,k) (text rows: norm_added_*) + RoPE: fused into the QKV epilogue above
  // 4. joint attention
  RF_TRY(rf_attention_fwd_ws(Q, K, VT, ATT, H, S, L.s_pad, D, St + Si, Sc > 0 ? dims->attn_mode : 0, dims->cross_bias,
                             0.08838834764831845f /* 1/sqrt(128) */, /*q_prescaled=*/1, w->qk_bound,
                             (char*)ws->base + L.sk + 4096, L.sk_bytes - 4096, st));
  // 5. output projections + gated residual: x += gate_msa * proj(attn)
  {
    RF_TRY(quant_rows(ATT, D));
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = D; d.epilogue = RF_EPI_GATE_RES;
    const bool add_cond = dims->add_cond_attn && Sc > 0;
    RF_TRY(run_stage(d, [&](int i, rf_gemm_group& g, bool f8) -> int {
      const Stream& s = sx[i];
      const bool txt = (i == 0);
      g.M = s.rows;
      g.bias = txt ? w->b_add_out : w->b_out;
      g.gate = s.mod + 2 * D;
      g.out = s.x; g.ldo = ldx; g.residual = s.x; g.ldr = ldx;
      if (i == 2 && add_cond) {  // keep cond_attn_output = gate*proj on its own (block.py:224-228)
        g.out = HID; g.ldo = D; g.residual = nullptr; g.ldr = 0;
      }
      if (f8) {
        const rf_w8& q8 = txt ? w->q_add_out : w->q_out;
        set_seg(g.seg[0], A8 + (int64_t)s.off * D, D, q8.w, D, D);
        g.a_scale = sA + s.off; g.w_scale = q8.scale;
        return RF_OK;
      }
      const bool mrg_lora_out = !txt && s.lora && w->lora_out.B && w->lora_out.merged;   // LoRA folded into a per-group weight copy (W + s B A)
      set_seg(g.seg[0], ATT + (int64_t)s.off * D, D, mrg_lora_out ? (const bf16_t*)w->lora_out.B : (const bf16_t*)(txt ? w->w_add_out : w->w_out), D, D);
      if (!txt && s.lora && w->lora_out.B && !w->lora_out.merged) {
        bf16_t* T = LT + (int64_t)s.off * 256;
        RF_TRY(lora_down(w->lora_out, ATT + (int64_t)s.off * D, D, D, nullptr, 0, 0, s.rows, T, ws, L, st));
        set_seg(g.seg[1], T, 256, w->lora_out.B, w->lora_out.r_pad, w->lora_out.r_pad);
      }
      return RF_OK;
    }));
    if (add_cond) {
      for (int r = 0; r < Sc; ++r) {  // rows may be strided (ldx != D): add row by row only then
        if (ldx == D) {
          RF_TRY(rf_add_inplace(x_cond, HID, (int64_t)Sc * D, st));
          RF_TRY(rf_add_inplace(x_img, HID, (int64_t)Sc * D, st));
          break;
        }
        RF_TRY(rf_add_inplace((bf16_t*)x_cond + (int64_t)r * ldx, HID + (int64_t)r * D, D, st));
        RF_TRY(rf_add_inplace((bf16_t*)x_img + (int64_t)r * ldx, HID + (int64_t)r * D, D, st));
      }
    }
  }
  // 6. norm2 + modulate with (scale_mlp, shift_mlp)
  RF_TRY(ln_mod(4, 3));
  // 7. FF up + GELU(tanh)
  {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = MLP; d.epilogue = RF_EPI_GELU;
    RF_TRY(run_stage(d, [&](int i, rf_gemm_group& g, bool f8) -> int {
      const Stream& s = sx[i];
      const bool txt = (i == 0);
      g.M = s.rows;
      g.bias = txt ? w->b_ffc1 : w->b_ff1;
      g.out = HID + (int64_t)s.off * MLP; g.ldo = MLP;
      if (f8) {
        const rf_w8& q8 = txt ? w->q_ffc1 : w->q_ff1;
        set_seg(g.seg[0], XN8 + (int64_t)s.off * D, D, q8.w, D, D);
        g.a_scale = sXN + s.off; g.w_scale = q8.scale;
      } else {
        set_seg(g.seg[0], XN + (int64_t)s.off * D, D, txt ? w->w_ffc1 : w->w_ff1, D, D);
      }
      return RF_OK;
    }));
  }
  // 8. FF down + gated residual: x += gate_mlp * ff(x)
  {
    RF_TRY(quant_rows(HID, MLP));
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = D; d.epilogue = RF_EPI_GATE_RES;
    RF_TRY(run_stage(d, [&](int i, rf_gemm_group& g, bool f8) -> int {
      const Stream& s = sx[i];
      const bool txt = (i == 0);
      g.M = s.rows;
      g.bias = txt ? w->b_ffc2 : w->b_ff2;
      g.gate = s.mod + 5 * D;
      g.out = s.x; g.ldo = ldx; g.residual = s.x; g.ldr = ldx;
      if (f8) {
        const rf_w8& q8 = txt ? w->q_ffc2 : w->q_ff2;
        set_seg(g.seg[0], A8 + (int64_t)s.off * MLP, MLP, q8.w, MLP, MLP);
        g.a_scale = sA + s.off; g.w_scale = q8.scale;
        return RF_OK;
      }
      const bool mrg_lora_ff2 = !txt && s.lora && w->lora_ff2.B && w->lora_ff2.merged;   // LoRA folded into a per-group weight copy (W + s B A)
      set_seg(g.seg[0], HID + (int64_t)s.off * MLP, MLP, mrg_lora_ff2 ? (const bf16_t*)w->lora_ff2.B : (const bf16_t*)(txt ? w->w_ffc2 : w->w_ff2), MLP, MLP);
      if (!txt && s.lora && w->lora_ff2.B && !w->lora_ff2.merged) {
        bf16_t* T = LT + (int64_t)s.off * 256;
        RF_TRY(lora_down(w->lora_ff2, HID + (int64_t)s.off * MLP, MLP, MLP, nullptr, 0, 0, s.rows, T, ws, L, st));
        set_seg(g.seg[1], T, 256, w->lora_ff2.B, w->lora_ff2.r_pad, w->lora_ff2.r_pad);
      }
      return RF_OK;
    }));
  }
  return RF_OK;
}

// =================================================================================================
// SingleStream block (block.py:275-333)
// =================================================================================================
extern "C" int rf_single_block_fwd(const rf_flux_dims* dims, const rf_single_block_weights* w, void* x_main,
                                   void* x_cond, int64_t ldx, const void* mod_main, const void* mod_cond,
                                   const float* cos_tab, const float* sin_tab, const rf_workspace* ws, void* stream) {
  WsLayout L;
  RF_TRY(check_dims(dims, ws, L));
  RF_REQUIRE(w && x_main && mod_main && cos_tab && sin_tab, RF_ERR_NULL, "rf_single_block_fwd: NULL pointer");
  RF_REQUIRE(dims->S_cond == 0 || (x_cond && mod_cond), RF_ERR_NULL, "rf_single_block_fwd: condition stream NULL");
  hipStream_t st = (hipStream_t)stream;
  const int D = dims->D, H = dims->heads, MLP = dims->mlp;
  const int Sm = dims->S_txt + dims->S_img, Sc = dims->S_cond, S = L.S;
  bf16_t* XN = at(ws, L.xn); bf16_t* Q = at(ws, L.q); bf16_t* K = at(ws, L.k); bf16_t* VT = at(ws, L.vt);
  bf16_t* ATT = at(ws, L.att); bf16_t* HID = at(ws, L.hid); bf16_t* LT = at(ws, L.lt);
  struct Stream { bf16_t* x; const bf16_t* mod; int rows, off; bool lora; };
  Stream sx[2] = {{(bf16_t*)x_main, (const bf16_t*)mod_main, Sm, 0, dims->lora_on_main != 0},
                  {(bf16_t*)x_cond, (const bf16_t*)mod_cond, Sc, Sm, true}};

  const bool have8 = dims->fp8 && w->q_qkv_mlp.w && w->q_out.w;
  RF_REQUIRE(!dims->fp8 || have8, RF_ERR_NULL, "rf_single_block_fwd: dims.fp8 set but the block has no fp8 weight copies");
  const bool any_lora = w->lora_qkv_mlp.B || w->lora_out.B;
  const bool use8[2] = {have8 && !(sx[0].lora && any_lora), false};   // see rf_double_block_fwd
  uint8_t* XN8 = (uint8_t*)ws->base + L.xn8;
  uint8_t* A8 = (uint8_t*)ws->base + L.a8;
  float* sXN = (float*)((char*)ws->base + L.s_xn);
  float* sA = (float*)((char*)ws->base + L.s_a);
  auto run_stage = [&](const rf_gemm_desc& base, auto&& fill) -> int {
    rf_gemm_desc d = base;
    int n = 0;
    bool any8 = false;
    for (int i = 0; i < 2; ++i) {
      if (sx[i].rows <= 0) continue;
      RF_TRY(fill(i, d.g[n], use8[i]));
      any8 = any8 || use8[i];
      ++n;
    }
    if (n == 0) return RF_OK;
    d.num_groups = n;
    attach_scratch(d, ws, L);
    return any8 ? rf_gemm_w8a8(&d, st) : rf_gemm_bf16(&d, st);
  };

  // 1. AdaLN-Zero-Single: mod rows 0 shift, 1 scale, 2 gate
  {
    LnModStream bf[2];
    int nbf = 0;
    for (int i = 0; i < 2; ++i) {
      const Stream& s = sx[i];
      if (s.rows <= 0) continue;
      if (use8[i])
        RF_TRY(rf_layernorm_modulate_fp8(s.x, ldx, XN8 + (int64_t)s.off * D, D, sXN + s.off, s.rows, D, s.mod + 1 * D, s.mod + 0 * D,
                                         1e-6f, st));
      else
        bf[nbf++] = LnModStream{s.x, XN + (int64_t)s.off * D, s.mod + 1 * D, s.mod + 0 * D, s.rows};
    }
    RF_TRY(ln_mod_grouped(bf, nbf, ldx, D, D, 1e-6f, st));
  }
  // 2. fused [to_q|to_k|to_v|proj_mlp]: QKV head-major, MLP branch through GELU(tanh) into HID
  {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = 3 * D + MLP; d.epilogue = RF_EPI_QKV_GELU; d.n_split = 3 * D;
    d.q = Q; d.k = K; d.vt = VT; d.heads = H; d.s_pad = L.s_pad;
    d.rope_cos = cos_tab; d.rope_sin = sin_tab; d.norm_eps = 1e-6f;
    d.q_scale = QK_PRESCALE;
    RF_TRY(run_stage(d, [&](int i, rf_gemm_group& g, bool f8) -> int {
      const Stream& s = sx[i];
      g.M = s.rows; g.tok_offset = s.off;
      g.norm_q = w->norm_q; g.norm_k = w->norm_k;
      g.bias = w->b_qkv_mlp;
      g.out = HID + (int64_t)s.off * MLP; g.ldo = MLP;
      if (f8) {
        set_seg(g.seg[0], XN8 + (int64_t)s.off * D, D, w->q_qkv_mlp.w, D, D);
        g.a_scale = sXN + s.off; g.w_scale = w->q_qkv_mlp.scale;
        return RF_OK;
      }
      const bool mrg_lora_qkv_mlp = s.lora && w->lora_qkv_mlp.B && w->lora_qkv_mlp.merged;   // LoRA folded into a per-group weight copy (W + s B A)
      set_seg(g.seg[0], XN + (int64_t)s.off * D, D, mrg_lora_qkv_mlp ? (const bf16_t*)w->lora_qkv_mlp.B : (const bf16_t*)(w->w_qkv_mlp), D, D);
      if (s.lora && w->lora_qkv_mlp.B && !w->lora_qkv_mlp.merged) {
        bf16_t* T = LT + (int64_t)s.off * 256;
        RF_TRY(lora_down(w->lora_qkv_mlp, XN + (int64_t)s.off * D, D, D, nullptr, 0, 0, s.rows, T, ws, L, st));
        set_seg(g.seg[1], T, 256, w->lora_qkv_mlp.B, w->lora_qkv_mlp.r_pad, w->lora_qkv_mlp.r_pad);
      }
      return RF_OK;
    }));
  }
  // 3. RMSNorm(q,k) + RoPE: fused into the epilogue above (no added-norm rows in single blocks)
  // 4. attention
  RF_TRY(rf_attention_fwd_ws(Q, K, VT, ATT, H, S, L.s_pad, D, Sm, Sc > 0 ? dims->attn_mode : 0, dims->cross_bias,
                             0.08838834764831845f, /*q_prescaled=*/1, w->qk_bound,
                             (char*)ws->base + L.sk + 4096, L.sk_bytes - 4096, st));
  // 5. proj_out over cat([attn, mlp]) + gated residual.  bf16: two K segments, no concat.  fp8: the per-token
  //    quantisation writes [attn | mlp] side by side with ONE row scale, so it is a single K = D + mlp segment.
  {
    const int64_t ldw = (int64_t)D + MLP;
    for (int i = 0; i < 2; ++i)
      if (use8[i] && sx[i].rows > 0)
        RF_TRY(rf_quant_rows_fp8(ATT + (int64_t)sx[i].off * D, D, D, HID + (int64_t)sx[i].off * MLP, MLP, MLP,
                                 A8 + (int64_t)sx[i].off * ldw, ldw, sA + sx[i].off, sx[i].rows, st));
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = D; d.epilogue = RF_EPI_GATE_RES;
    RF_TRY(run_stage(d, [&](int i, rf_gemm_group& g, bool f8) -> int {
      const Stream& s = sx[i];
      g.M = s.rows;
      g.bias = w->b_out;
      g.gate = s.mod + 2 * D;
      g.out = s.x; g.ldo = ldx; g.residual = s.x; g.ldr = ldx;
      if (f8) {
        set_seg(g.seg[0], A8 + (int64_t)s.off * ldw, ldw, w->q_out.w, ldw, (int)ldw);
        g.a_scale = sA + s.off; g.w_scale = w->q_out.scale;
        return RF_OK;
      }
      const bool mrg_out = s.lora && w->lora_out.B && w->lora_out.merged;   // LoRA folded into a per-group copy of proj_out
      const bf16_t* w_out = mrg_out ? (const bf16_t*)w->lora_out.B : (const bf16_t*)w->w_out;
      set_seg(g.seg[0], ATT + (int64_t)s.off * D, D, w_out, ldw, D);
      set_seg(g.seg[1], HID + (int64_t)s.off * MLP, MLP, w_out + D, ldw, MLP);
      if (s.lora && w->lora_out.B && !w->lora_out.merged) {
        bf16_t* T = LT + (int64_t)s.off * 256;
        RF_TRY(lora_down(w->lora_out, ATT + (int64_t)s.off * D, D, D, HID + (int64_t)s.off * MLP, MLP, MLP, s.rows, T, ws, L, st));
        set_seg(g.seg[2], T, 256, w->lora_out.B, w->lora_out.r_pad, w->lora_out.r_pad);
      }
      return RF_OK;
    }));
  }
  return RF_OK;
}

// =================================================================================================
// Whole forward (transformer.py:47-252) and the T-step loop (generate.py:216-296)
// =================================================================================================
extern "C" int64_t rf_mod_table_cols(const rf_flux_model* m, int32_t D) {
  if (!m) return RF_ERR_NULL;
  return (int64_t)m->num_double * 12 * D + (int64_t)m->num_single * 3 * D + 2 * (int64_t)D;
}

extern "C" int rf_flux_forward(const rf_flux_dims* dims, const rf_flux_model* m, const void* latents,
                               const void* cond_latents, const void* ctx, const void* mod_main, const void* mod_cond,
                               const float* cos_tab, const float* sin_tab, void* out, const rf_workspace* ws,
                               void* stream) {
  WsLayout L;
  RF_TRY(check_dims(dims, ws, L));
  RF_REQUIRE(m && latents && mod_main && out && cos_tab && sin_tab, RF_ERR_NULL, "rf_flux_forward: NULL pointer");
  RF_REQUIRE(dims->S_txt == 0 || ctx, RF_ERR_NULL, "rf_flux_forward: ctx NULL");
  RF_REQUIRE(dims->S_cond == 0 || (cond_latents && mod_cond), RF_ERR_NULL, "rf_flux_forward: condition inputs NULL");
  RF_REQUIRE(m->in_ch % 64 == 0 && m->joint_dim % 64 == 0, RF_ERR_SHAPE, "rf_flux_forward: in_ch/joint_dim %% 64");
  hipStream_t st = (hipStream_t)stream;
  const int D = dims->D;
  const int St = dims->S_txt, Si = dims->S_img, Sc = dims->S_cond;
  bf16_t* X = at(ws, L.x);
  bf16_t* x_txt = X;
  bf16_t* x_img = X + (int64_t)St * D;
  bf16_t* x_cond = X + (int64_t)(St + Si) * D;
  bf16_t* XN = at(ws, L.xn);
  bf16_t* LT = at(ws, L.lt);
  const bf16_t* mm = (const bf16_t*)mod_main;
  const bf16_t* mc = (const bf16_t*)mod_cond;

  // embedders (transformer.py:91-93,115): x_embedder for image (LoRA only if latent_lora) and
  // condition (LoRA on) tokens, context_embedder for text tokens
  {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = D; d.epilogue = RF_EPI_STORE; d.num_groups = 2;
    const void* src[2] = {latents, cond_latents};
    bf16_t* dst[2] = {x_img, x_cond};
    const int rows[2] = {Si, Sc};
    const bool lora[2] = {dims->lora_on_main != 0, true};
    for (int i = 0; i < 2; ++i) {
      rf_gemm_group& g = d.g[i];
      g.M = rows[i];
      if (rows[i] <= 0) continue;
      const bool mrg = lora[i] && m->lora_x_embed.B && m->lora_x_embed.merged;
      set_seg(g.seg[0], src[i], m->in_ch, mrg ? m->lora_x_embed.B : m->w_x_embed, m->in_ch, m->in_ch);
      g.bias = m->b_x_embed; g.out = dst[i]; g.ldo = D;
      if (lora[i] && m->lora_x_embed.B && !m->lora_x_embed.merged) {
        bf16_t* T = LT + (int64_t)(i == 0 ? St : St + Si) * 256;
        RF_TRY(lora_down(m->lora_x_embed, (const bf16_t*)src[i], m->in_ch, m->in_ch, nullptr, 0, 0, rows[i], T, ws, L, st));
        set_seg(g.seg[1], T, 256, m->lora_x_embed.B, m->lora_x_embed.r_pad, m->lora_x_embed.r_pad);
      }
    }
    attach_scratch(d, ws, L);
    RF_TRY(rf_gemm_bf16(&d, st));
  }
  if (St > 0) {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = D; d.epilogue = RF_EPI_STORE; d.num_groups = 1;
    set_seg(d.g[0].seg[0], ctx, m->joint_dim, m->w_ctx_embed, m->joint_dim, m->joint_dim);
    d.g[0].bias = m->b_ctx_embed; d.g[0].M = St; d.g[0].out = x_txt; d.g[0].ldo = D;
    attach_scratch(d, ws, L);
    RF_TRY(rf_gemm_bf16(&d, st));
  }

  // 19 x DoubleStream: table slots [img 6D | txt 6D] per block; condition rows use the img slot
  // of the cond_temb table (they go through norm1, block.py:194-201)
  for (int b = 0; b < m->num_double; ++b) {
    const int64_t o = (int64_t)b * 12 * D;
    RF_TRY(rf_double_block_fwd(dims, &m->dbl[b], x_txt, x_img, Sc > 0 ? x_cond : nullptr, D, mm + o + 6 * D, mm + o,
                               Sc > 0 ? mc + o : nullptr, cos_tab, sin_tab, ws, stream));
  }
  // 38 x SingleStream on [txt;img] (contiguous rows of X) + condition
  const int64_t so = (int64_t)m->num_double * 12 * D;
  for (int b = 0; b < m->num_single; ++b) {
    const int64_t o = so + (int64_t)b * 3 * D;
    RF_TRY(rf_single_block_fwd(dims, &m->sgl[b], X, Sc > 0 ? x_cond : nullptr, D, mm + o, Sc > 0 ? mc + o : nullptr,
                               cos_tab, sin_tab, ws, stream));
  }
  // norm_out (AdaLN-Continuous: scale first, then shift) on the image rows + proj_out (transformer.py:241-244)
  const int64_t no = so + (int64_t)m->num_single * 3 * D;
  RF_TRY(rf_layernorm_modulate(x_img, D, XN, D, Si, D, mm + no, mm + no + D, 1e-6f, st));
  {
    rf_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = m->in_ch; d.epilogue = RF_EPI_STORE; d.num_groups = 1;
    set_seg(d.g[0].seg[0], XN, D, m->w_proj_out, D, D);
    d.g[0].bias = m->b_proj_out; d.g[0].M = Si; d.g[0].out = out; d.g[0].ldo = m->in_ch;
    attach_scratch(d, ws, L);
    RF_TRY(rf_gemm_bf16(&d, st));
  }
  return RF_OK;
}

extern "C" int rf_flux_denoise(const rf_flux_dims* dims, const rf_flux_model* m, void* latents,
                               const void* cond_latents, const void* ctx, const void* mod_main_steps,
                               int64_t mod_stride, const void* mod_cond, const float* cos_tab, const float* sin_tab,
                               const float* dts, int32_t T, void* vel_scratch, const rf_workspace* ws, void* stream) {
  RF_REQUIRE(dts && vel_scratch && latents && mod_main_steps && m && dims, RF_ERR_NULL, "rf_flux_denoise: NULL pointer");
  RF_REQUIRE(T > 0, RF_ERR_SHAPE, "rf_flux_denoise: T=%d", T);
  const int64_t n = (int64_t)dims->S_img * m->in_ch;
  for (int i = 0; i < T; ++i) {
    const bf16_t* mod = (const bf16_t*)mod_main_steps + (int64_t)i * mod_stride;
    RF_TRY(rf_flux_forward(dims, m, latents, cond_latents, ctx, mod, mod_cond, cos_tab, sin_tab, vel_scratch, ws, stream));
    RF_TRY(rf_euler_step(latents, vel_scratch, n, dts[i], stream));
  }
  return RF_OK;
}
