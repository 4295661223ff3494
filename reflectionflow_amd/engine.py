"""FluxEngine: packs a FluxTransformer2DModel's weights into the C-ABI structs once, owns the
device workspace, and runs whole forwards / whole T-step denoises through librf_flux.

MI355X-first choices made here:
  * weights of sibling projections are concatenated ONCE (to_q|to_k|to_v, and for single blocks
    |proj_mlp) so each block runs 4 (double) / 2 (single) large grouped GEMMs;
  * all 57 blocks' AdaLN modulation vectors depend only on (t, guidance, pooled text), not on the
    latents, and the T timesteps are known before the loop starts: the modulation table for ALL
    steps is produced up front by ~58 skinny GEMM launches (each streams one AdaLN weight once for
    all T rows) instead of 2.3 k GEMV launches x 6.5 GB of weight traffic per step;
  * LoRA (FLUX-Corrector) is never merged: it rides as an extra K-segment on the token groups it
    is active for (condition rows; image rows too iff model_config["latent_lora"]).
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import ops
from .flux import modules as M


# (the packed copies below are inference operands: made under no_grad -- a copy that carried a grad_fn would keep the factors' gradient
#  accumulators alive on the stream it was made on, and a later backward under hipGraph capture would then synchronise with that
#  stream from inside the capture)
@torch.no_grad()
def _pad_lora(A: torch.Tensor, B: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, int]:
    r = A.shape[0]
    r_pad = (r + 63) // 64 * 64
    if r_pad > 256:
        raise ops.RFError(f"stacked LoRA rank {r} exceeds the 256-column low-rank buffer")
    Ap = torch.zeros(r_pad, A.shape[1], dtype=A.dtype, device=A.device)
    Ap[:r] = A
    Bp = torch.zeros(B.shape[0], r_pad, dtype=B.dtype, device=B.device)
    Bp[:, :r] = B
    return Ap.contiguous(), Bp.contiguous(), r_pad


def _base(lin):
    return lin.base_layer if isinstance(lin, M.LoraLinear) else lin


@torch.no_grad()
def _fused_lora(linears) -> Optional[Tuple[torch.Tensor, torch.Tensor, int]]:
    """Stack the LoRA factors of sibling linears that were concatenated along N:
    A = rows stacked, B = block diagonal (scaling folded in)."""
    if not any(isinstance(l, M.LoraLinear) for l in linears):
        return None
    As, blocks = [], []
    for l in linears:
        if isinstance(l, M.LoraLinear):
            A, B = l.lora_factors()
        else:
            A = torch.zeros(0, l.in_features, dtype=l.weight.dtype, device=l.weight.device)
            B = torch.zeros(l.out_features, 0, dtype=l.weight.dtype, device=l.weight.device)
        As.append(A)
        blocks.append(B)
    A = torch.cat(As, 0)
    B = torch.block_diag(*[b.float() for b in blocks]).to(A.dtype)
    return _pad_lora(A, B)


# Modules the engine has a LoRA K-segment (or a modulation-table term) for.  enable_lora() gating in the reference
# (lora_controller.py:5-42; call sites block.py:23,185,148-155,252-259,288-296,319-322, transformer.py:91) covers
# exactly the `gated` kinds: there LoRA acts on condition rows always and on image/text rows iff latent_lora.
# LoRA on norm1_context.linear / norm_out.linear is never gated by the reference (no enable_lora around them):
# always on.  Any other LoRA'd linear (add_*_proj, to_add_out, ff.net.0.proj, ff_context.*, context_embedder,
# proj_out, time_text_embed.* -- reachable with PEFT `target_modules: all-linear`, train/model.py:73) has no
# segment here and would be silently ignored, so packing refuses it.
_LORA_GATED = (r"x_embedder", r"transformer_blocks\.\d+\.norm1\.linear", r"transformer_blocks\.\d+\.attn\.to_[qkv]",
               r"transformer_blocks\.\d+\.attn\.to_out\.0", r"transformer_blocks\.\d+\.ff\.net\.2",
               r"single_transformer_blocks\.\d+\.norm\.linear", r"single_transformer_blocks\.\d+\.proj_mlp",
               r"single_transformer_blocks\.\d+\.proj_out", r"single_transformer_blocks\.\d+\.attn\.to_[qkv]")
_LORA_ALWAYS = (r"transformer_blocks\.\d+\.norm1_context\.linear", r"norm_out\.linear")


def check_lora_placement(transformer=None, names=None) -> None:
    """Raise RFError if a LoraLinear sits (or, given `names`, would be put) on a module the HIP engine cannot
    apply it for."""
    import re
    ok = re.compile("^(?:" + "|".join(_LORA_GATED + _LORA_ALWAYS) + ")$")
    if names is None:
        names = [n for n, m in transformer.named_modules() if isinstance(m, M.LoraLinear)]
    bad = [n for n in names if not ok.match(n)]
    if bad:
        raise ops.RFError("LoRA on modules the HIP engine has no K-segment for (it would be silently dropped): "
                          f"{bad[:6]}{' ...' if len(bad) > 6 else ''}; supported: the FLUX-Corrector target list "
                          "(train_flux/config.yaml:53) plus norm1_context.linear / norm_out.linear")


class _Packed:
    """A C weights struct plus the tensors its pointers refer to (kept alive together)."""

    def __init__(self, struct):
        self.struct = struct
        self.keep: List[torch.Tensor] = []
        self.by_ptr: Dict[int, torch.Tensor] = {}
        self.fp8 = False

    def k(self, t: torch.Tensor) -> int:
        t = t.detach().contiguous()
        self.keep.append(t)
        self.by_ptr[t.data_ptr()] = t
        return t.data_ptr()

    def w8(self, dst: L.rf_w8, weight: torch.Tensor):
        """fp8 copy (e4m3fn bytes + per-output-channel scale) of a fused bf16 weight for rf_gemm_w8a8."""
        q, sc = ops.quantize_weight_fp8(weight)
        dst.w, dst.scale = self.k(q), self.k(sc)

    def lora(self, seg: L.rf_lora_seg, linears, merge_into: Optional[torch.Tensor] = None):
        """LoRA of the fused `linears`: the low-rank K-segment (A, s B), or -- merge_into = the fused base weight -- the MERGED
        weight bf16(W + s B A) for the token groups LoRA acts on (rf_lora_seg.merged; no low-rank launches at run time)."""
        f = _fused_lora(linears)
        if f is not None:
            A, B, r_pad = f
            if merge_into is not None:
                Wm = torch.addmm(merge_into.float(), B.float(), A.float()).to(merge_into.dtype)     # one rounding of the sum
                seg.A, seg.B, seg.r_pad, seg.merged = None, self.k(Wm), 0, 1
            else:
                seg.A, seg.B, seg.r_pad, seg.merged = self.k(A), self.k(B), r_pad, 0


def _require_device_bf16(p: torch.Tensor, what: str):
    if not p.is_cuda or p.dtype != torch.bfloat16:
        raise ops.RFError(f"{what}: weights must be bf16 on a HIP device (got {p.dtype} on {p.device}); "
                          "the HIP path has no CPU fallback")


def _merged_lora(module) -> bool:
    return bool(getattr(module, "_rf_merged_lora", False))


def pack_double_block(b, refresh: bool = False, fp8: bool = False) -> _Packed:
    """rf_double_block_weights for a FluxTransformerBlock (cached on the module).  fp8: also pack e4m3fn copies of
    the eight big weights (BASELINE cfg5)."""
    pk = getattr(b, "_rf_packed", None)
    if pk is not None and not refresh and (pk.fp8 or not fp8):
        return pk
    a = b.attn
    _require_device_bf16(_base(a.to_q).weight, "pack_double_block")
    pk = _Packed(L.rf_double_block_weights())
    w, cat = pk.struct, torch.cat
    qkv = [a.to_q, a.to_k, a.to_v]
    w.w_qkv = pk.k(cat([_base(l).weight for l in qkv], 0))
    w.b_qkv = pk.k(cat([_base(l).bias for l in qkv], 0))
    add = [a.add_q_proj, a.add_k_proj, a.add_v_proj]
    w.w_add_qkv = pk.k(cat([l.weight for l in add], 0))
    w.b_add_qkv = pk.k(cat([l.bias for l in add], 0))
    w.norm_q, w.norm_k = pk.k(a.norm_q.weight), pk.k(a.norm_k.weight)
    w.norm_added_q, w.norm_added_k = pk.k(a.norm_added_q.weight), pk.k(a.norm_added_k.weight)
    w.w_out, w.b_out = pk.k(_base(a.to_out[0]).weight), pk.k(_base(a.to_out[0]).bias)
    w.w_add_out, w.b_add_out = pk.k(a.to_add_out.weight), pk.k(a.to_add_out.bias)
    w.w_ff1, w.b_ff1 = pk.k(b.ff.net[0].proj.weight), pk.k(b.ff.net[0].proj.bias)
    w.w_ff2, w.b_ff2 = pk.k(_base(b.ff.net[2]).weight), pk.k(_base(b.ff.net[2]).bias)
    w.w_ffc1, w.b_ffc1 = pk.k(b.ff_context.net[0].proj.weight), pk.k(b.ff_context.net[0].proj.bias)
    w.w_ffc2, w.b_ffc2 = pk.k(b.ff_context.net[2].weight), pk.k(b.ff_context.net[2].bias)
    mrg = _merged_lora(b)
    pk.lora(w.lora_qkv, qkv, pk.by_ptr[w.w_qkv] if mrg else None)
    pk.lora(w.lora_out, [a.to_out[0]], pk.by_ptr[w.w_out] if mrg else None)
    pk.lora(w.lora_ff2, [b.ff.net[2]], pk.by_ptr[w.w_ff2] if mrg else None)
    # |score| bound from the norm weights alone: lets attention skip the online softmax (rf_attention_fwd score_bound)
    w.qk_bound = ops.qk_score_bound((a.norm_q.weight, a.norm_added_q.weight), (a.norm_k.weight, a.norm_added_k.weight))
    pk.fp8 = fp8
    if fp8:
        for dst, src in ((w.q_qkv, w.w_qkv), (w.q_add_qkv, w.w_add_qkv), (w.q_out, w.w_out), (w.q_add_out, w.w_add_out),
                         (w.q_ff1, w.w_ff1), (w.q_ff2, w.w_ff2), (w.q_ffc1, w.w_ffc1), (w.q_ffc2, w.w_ffc2)):
            pk.w8(dst, pk.by_ptr[src])
    pk.D, pk.heads, pk.mlp = a.to_q.in_features, a.heads, b.ff.net[0].proj.out_features
    object.__setattr__(b, "_rf_packed", pk)
    return pk


def pack_single_block(b, refresh: bool = False, fp8: bool = False) -> _Packed:
    """rf_single_block_weights for a FluxSingleTransformerBlock (cached on the module)."""
    pk = getattr(b, "_rf_packed", None)
    if pk is not None and not refresh and (pk.fp8 or not fp8):
        return pk
    a = b.attn
    _require_device_bf16(_base(a.to_q).weight, "pack_single_block")
    pk = _Packed(L.rf_single_block_weights())
    w, cat = pk.struct, torch.cat
    fused = [a.to_q, a.to_k, a.to_v, b.proj_mlp]
    w.w_qkv_mlp = pk.k(cat([_base(l).weight for l in fused], 0))
    w.b_qkv_mlp = pk.k(cat([_base(l).bias for l in fused], 0))
    w.norm_q, w.norm_k = pk.k(a.norm_q.weight), pk.k(a.norm_k.weight)
    w.w_out, w.b_out = pk.k(_base(b.proj_out).weight), pk.k(_base(b.proj_out).bias)
    mrg = _merged_lora(b)
    pk.lora(w.lora_qkv_mlp, fused, pk.by_ptr[w.w_qkv_mlp] if mrg else None)
    pk.lora(w.lora_out, [b.proj_out], pk.by_ptr[w.w_out] if mrg else None)
    w.qk_bound = ops.qk_score_bound((a.norm_q.weight,), (a.norm_k.weight,))
    pk.fp8 = fp8
    if fp8:
        pk.w8(w.q_qkv_mlp, pk.by_ptr[w.w_qkv_mlp])
        pk.w8(w.q_out, pk.by_ptr[w.w_out])
    pk.D, pk.heads, pk.mlp = a.to_q.in_features, a.heads, b.proj_mlp.out_features
    object.__setattr__(b, "_rf_packed", pk)
    return pk


def make_dims(D: int, heads: int, mlp: int, S_txt: int, S_img: int, S_cond: int = 0,
              model_config: Optional[dict] = None, c_factor: Optional[float] = None, fp8: bool = False) -> L.rf_flux_dims:
    mc = model_config or {}
    d = L.rf_flux_dims()
    d.D, d.heads, d.mlp = D, heads, mlp
    d.S_txt, d.S_img, d.S_cond = S_txt, S_img, S_cond
    d.attn_mode, d.cross_bias = 0, 0.0
    if S_cond > 0:
        if c_factor is not None:                    # block.py:115-122 (overrides the boolean mask)
            d.attn_mode, d.cross_bias = 1, math.log(c_factor)
        elif not mc.get("union_cond_attn", True):   # block.py:106-114
            d.attn_mode = 2
    d.lora_on_main = 1 if mc.get("latent_lora", False) else 0
    d.add_cond_attn = 1 if mc.get("add_cond_attn", False) else 0
    d.fp8 = 1 if fp8 else 0
    return d


_WS_CACHE: "Dict[tuple, torch.Tensor]" = {}
_WS_CACHE_MAX = 4     # geometries x streams kept alive; least-recently-used entries are dropped (a long search
                      # walks few geometries at a time: plain round, conditioned round, maybe a second resolution)


def get_workspace(device, d: L.rf_flux_dims) -> L.rf_workspace:
    """Caller-owned scratch for one (D, heads, mlp, S_txt, S_img, S_cond) geometry, allocated once per
    stream (calls on different streams may overlap on the device, so they must not share scratch)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, d.D, d.heads, d.mlp, d.S_txt, d.S_img, d.S_cond,
           d.lora_on_main, d.fp8)
    buf = _WS_CACHE.pop(key, None)
    if buf is None:
        n = int(L.load().rf_workspace_bytes(C.byref(d)))
        while len(_WS_CACHE) >= _WS_CACHE_MAX:
            # evict the least recently used workspace; work already enqueued on its stream keeps the memory alive
            # through the caching allocator's stream-ordered reuse (the buffer was allocated on that stream)
            _WS_CACHE.pop(next(iter(_WS_CACHE)))
        # zero-filled: padded key rows of V^T must be finite (they are multiplied by p = 0)
        buf = torch.zeros(n, dtype=torch.uint8, device=device)
    _WS_CACHE[key] = buf          # (re-)insert at the most-recently-used end
    ws = L.rf_workspace()
    ws.base, ws.bytes = buf.data_ptr(), buf.numel()
    return ws


class FluxEngine:
    def __init__(self, transformer: M.FluxTransformer2DModel):
        self._graphs = {}      # hipGraphs of whole denoise loops, see denoise(use_graph=True); LRU order
        self._graph_captures = {}
        self._graph_cache_max = 3
        self._eager_done = False
        self.lib = L.load()
        self.tr = transformer
        self.fp8 = bool(getattr(transformer, "_rf_fp8", False))     # BASELINE cfg5: fp8 weights + activations (W8A8)
        p = _base(transformer.x_embedder).weight
        _require_device_bf16(p, "FluxEngine")
        self.device = p.device
        cfg = transformer.config
        self.D = cfg.num_attention_heads * cfg.attention_head_dim
        self.heads = cfg.num_attention_heads
        self.mlp = transformer.transformer_blocks[0].ff.net[0].proj.out_features if len(transformer.transformer_blocks) \
            else transformer.single_transformer_blocks[0].mlp_hidden_dim
        self.in_ch = cfg.in_channels
        self._rope_cache: Dict[tuple, tuple] = {}
        self._temb_pad: Dict[tuple, tuple] = {}
        self._pack()

    # ------------------------------------------------------------------ packing
    def _pack(self):
        tr = self.tr
        check_lora_placement(tr)
        nd, ns = len(tr.transformer_blocks), len(tr.single_transformer_blocks)
        self._dbl = (L.rf_double_block_weights * max(nd, 1))()
        self._sgl = (L.rf_single_block_weights * max(ns, 1))()
        self._packs = []
        for i, b in enumerate(tr.transformer_blocks):
            pk = pack_double_block(b, refresh=True, fp8=self.fp8)
            self._packs.append(pk)
            self._dbl[i] = pk.struct
        for i, b in enumerate(tr.single_transformer_blocks):
            pk = pack_single_block(b, refresh=True, fp8=self.fp8)
            self._packs.append(pk)
            self._sgl[i] = pk.struct
        top = _Packed(L.rf_flux_model())
        self._packs.append(top)
        m = top.struct
        m.num_double, m.num_single = nd, ns
        m.dbl = C.cast(self._dbl, C.POINTER(L.rf_double_block_weights))
        m.sgl = C.cast(self._sgl, C.POINTER(L.rf_single_block_weights))
        m.w_x_embed, m.b_x_embed = top.k(_base(tr.x_embedder).weight), top.k(_base(tr.x_embedder).bias)
        top.lora(m.lora_x_embed, [tr.x_embedder], top.by_ptr[m.w_x_embed] if _merged_lora(tr) else None)
        m.w_ctx_embed, m.b_ctx_embed = top.k(tr.context_embedder.weight), top.k(tr.context_embedder.bias)
        m.w_proj_out, m.b_proj_out = top.k(tr.proj_out.weight), top.k(tr.proj_out.bias)
        m.in_ch, m.joint_dim = self.in_ch, tr.context_embedder.in_features
        self.model = m
        self.mod_cols = int(self.lib.rf_mod_table_cols(C.byref(m), self.D))
        # AdaLN linears in table order: per double block [norm1 (img) | norm1_context (txt)], singles, norm_out
        self._mod_linears, self._mod_always = [], []      # _mod_always[i]: LoRA not gated by enable_lora in the reference
        for b in tr.transformer_blocks:
            self._mod_linears += [b.norm1.linear, b.norm1_context.linear]
            self._mod_always += [False, True]
        for b in tr.single_transformer_blocks:
            self._mod_linears.append(b.norm.linear)
            self._mod_always.append(False)
        self._mod_linears.append(tr.norm_out.linear)
        self._mod_always.append(True)
        self._mod_lora = [(_pad_lora(*l.lora_factors()) if isinstance(l, M.LoraLinear) else None)
                          for l in self._mod_linears]

    # ------------------------------------------------------------------ small helpers
    def dims(self, S_txt: int, S_img: int, S_cond: int = 0, model_config: Optional[dict] = None,
             c_factor: Optional[float] = None) -> L.rf_flux_dims:
        return make_dims(self.D, self.heads, self.mlp, S_txt, S_img, S_cond, model_config, c_factor, fp8=self.fp8)

    def workspace(self, d: L.rf_flux_dims) -> L.rf_workspace:
        return get_workspace(self.device, d)

    def rope_tables(self, txt_ids, img_ids, cond_ids=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """FluxPosEmbed over [txt | img | cond] ids (transformer.py:129-134) -> fp32 [S,128] x2."""
        parts = [txt_ids, img_ids] + ([cond_ids] if cond_ids is not None else [])
        # first level: the SAME id tensors as last time (a denoise loop passes the same objects every step) -- no device read at all.
        # (tensor identity + version counter: an in-place edit of an id tensor bumps _version and misses)
        # (inference tensors -- created under torch.inference_mode() -- have no version counter and raise on ._version: no
        #  identity fast path for them, the value-level compare below still hits)
        fast = None
        if not any(t.is_inference() for t in parts):
            fast = tuple((id(t), t._version, tuple(t.shape)) for t in parts)
            hit = self._rope_cache.get(("obj", fast))
            if hit is not None and all(a is b for a, b in zip(hit[2], parts)):
                return hit[0], hit[1]
        ids = torch.cat([i.to(self.device).float() for i in parts], 0)
        # second level: equal VALUES in different tensors; the comparison stays on the device except for its one-bit result
        for k, v in self._rope_cache.items():
            if k[0] == "val" and k[1] == ids.shape[0] and bool(torch.equal(v[2], ids)):
                if fast is not None:
                    self._rope_cache[("obj", fast)] = (v[0], v[1], list(parts))
                return v[0], v[1]
        cos, sin = self.tr.pos_embed(ids)
        cos, sin = cos.contiguous(), sin.contiguous()
        if len(self._rope_cache) > 16:
            self._rope_cache.clear()
        self._rope_cache[("val", ids.shape[0], len(self._rope_cache))] = (cos, sin, ids)
        if fast is not None:
            self._rope_cache[("obj", fast)] = (cos, sin, list(parts))
        return cos, sin

    def _pad_k(self, lin) -> torch.Tensor:
        """weight of an embedder linear with its K axis zero-padded to a multiple of 64 (rf_gemm_bf16's K granule); cached per
        (tensor, version).  FLUX.1-dev's 256 / 768 / 3072 need no padding."""
        w = lin.weight
        K = w.shape[1]
        if K % 64 == 0 and w.dtype == torch.bfloat16 and w.is_contiguous():
            return w
        key = (id(w), None if w.is_inference() else w._version)
        hit = self._temb_pad.get(key)
        if hit is None or hit[0] is not w:
            wp = torch.zeros(w.shape[0], (K + 63) // 64 * 64, dtype=torch.bfloat16, device=self.device)
            wp[:, :K] = w
            if len(self._temb_pad) > 16:
                self._temb_pad.clear()
            hit = self._temb_pad[key] = (w, wp)
        return hit[1]

    def temb(self, timestep, guidance, pooled) -> torch.Tensor:
        """time_text_embed (transformer.py:95-114; diffusers' CombinedTimestep(Guidance)TextProjEmbeddings): sinusoid -> two-layer
        SiLU embedders of timestep, guidance and the pooled prompt, summed.  timestep/guidance arrive already x1000 in model dtype.

        ROW-INVARIANT by construction: the six linears run on rf_gemm_bf16 (whose rows do not depend on M), SiLU on rf_silu, the
        sinusoid and the sums are element-wise -- so the all-steps-at-once evaluation of the fast denoise path ([T] timesteps in one
        call) gives bit for bit the rows the per-step path (transformer.py:102-106, M = B per step) forms one at a time.  Up to
        round 4 these linears ran on PyTorch-ROCm, where hipBLASLt picks a different kernel for M = 50 than for M = 1 and the last
        bit of temb differed."""
        te = self.tr.time_text_embed

        def embed(m, x):
            x = x.to(torch.bfloat16)
            w1 = self._pad_k(m.linear_1)
            if w1.shape[1] != x.shape[1]:
                x = torch.nn.functional.pad(x, (0, w1.shape[1] - x.shape[1]))
            h = ops.silu(ops.linear(x.contiguous(), w1, m.linear_1.bias))
            return ops.linear(h, self._pad_k(m.linear_2), m.linear_2.bias)

        for m in (te.timestep_embedder, te.text_embedder) + ((te.guidance_embedder,) if guidance is not None else ()):
            if not (isinstance(m.linear_1, torch.nn.Linear) and isinstance(m.linear_2, torch.nn.Linear)):
                raise ops.RFError("time_text_embed linears must be plain nn.Linear (LoRA there is refused at packing, engine.py _LORA_GATED)")
        dt = pooled.dtype
        emb = embed(te.timestep_embedder, M.get_timestep_embedding(timestep).to(dt))
        if guidance is not None:
            emb = emb + embed(te.guidance_embedder, M.get_timestep_embedding(guidance).to(dt))
        return (emb + embed(te.text_embedder, pooled)).to(dt)

    def mod_table(self, temb: torch.Tensor, lora: bool = False) -> torch.Tensor:
        """[M, D] conditioning rows -> [M, mod_cols] modulation table (all AdaLN linears).
        `lora`: apply the enable_lora-gated LoRA terms (norm1.linear / norm.linear: condition rows, or
        latent_lora); LoRA on norm1_context.linear / norm_out.linear is un-gated in the reference and always applied."""
        if temb.dim() != 2 or temb.shape[1] != self.D:
            raise ops.RFError("mod_table: temb must be [M, D]")
        s = ops.silu(temb.to(torch.bfloat16).contiguous())
        table = torch.empty(s.shape[0], self.mod_cols, dtype=torch.bfloat16, device=self.device)
        col = 0
        for lin, lo, always in zip(self._mod_linears, self._mod_lora, self._mod_always):
            base = _base(lin)
            n = base.out_features
            extra = []
            if (lora or always) and lo is not None:
                A, B, _ = lo
                extra = [ops.Seg(ops.linear(s, A), B)]
            ops.linear(s, base.weight, base.bias, extra=extra, out=table[:, col:col + n])
            col += n
        assert col == self.mod_cols
        return table

    # ------------------------------------------------------------------ forward / denoise
    def forward(self, latents, ctx, mod_main, cos, sin, cond_latents=None, mod_cond=None, model_config=None,
                c_factor=None, out=None) -> torch.Tensor:
        """One sample: latents [S_img, in_ch], ctx [S_txt, joint], mod rows [mod_cols] -> velocity [S_img, in_ch]."""
        Sc = 0 if cond_latents is None else cond_latents.shape[0]
        d = self.dims(ctx.shape[0], latents.shape[0], Sc, model_config, c_factor)
        ws = self.workspace(d)
        if out is None:
            out = torch.empty_like(latents)
        for t in (latents, ctx, mod_main, out):
            if not (t.is_contiguous() and t.dtype == torch.bfloat16 and t.is_cuda):
                raise ops.RFError("FluxEngine.forward: inputs must be contiguous bf16 device tensors")
        if cos.shape[0] != d.S_txt + d.S_img + d.S_cond:
            raise ops.RFError("FluxEngine.forward: rope tables do not cover [txt|img|cond]")
        L.check(self.lib.rf_flux_forward(
            C.byref(d), C.byref(self.model), latents.data_ptr(), ops.ptr(cond_latents), ctx.data_ptr(),
            mod_main.data_ptr(), ops.ptr(mod_cond), cos.data_ptr(), sin.data_ptr(), out.data_ptr(),
            C.byref(ws), ops.stream_ptr()), "rf_flux_forward")
        return out

    def denoise(self, latents, ctx, mod_steps, dts, cos, sin, cond_latents=None, mod_cond=None, model_config=None,
                c_factor=None, use_graph: Optional[bool] = None) -> torch.Tensor:
        """T-step loop in ONE C call, latents [S_img, in_ch] updated in place.
        mod_steps [T, mod_cols] (row i = modulation table of timestep i), dts = sigma_{i+1}-sigma_i.
        use_graph (default: on for T >= 16 unless RF_DENOISE_GRAPH=0; RF_DENOISE_GRAPH=1 forces it for any T): replay the whole
        T-step loop (T x ~300 launches) as ONE hipGraph captured per (geometry, T, schedule) over static buffers -- bit-identical to
        the eager launches (tests/test_round3_gpu.py) and worth +0.4..1.0 % at cfg2 (profiles/r03_graph.md: the device-side cost of
        a dependent kernel boundary is the same 1.1-1.9 us eager or replayed, MI355X_MICROARCH.md 'boundary'; what goes away is the
        host-side enqueue jitter between the 15 000 launches of a candidate).  Never used while the timing hook is open."""
        Sc = 0 if cond_latents is None else cond_latents.shape[0]
        d = self.dims(ctx.shape[0], latents.shape[0], Sc, model_config, c_factor)
        ws = self.workspace(d)
        T = len(dts)
        if mod_steps.shape != (T, self.mod_cols) or not mod_steps.is_contiguous():
            raise ops.RFError("FluxEngine.denoise: mod_steps must be contiguous [T, mod_cols]")
        if use_graph is None:
            env = os.environ.get("RF_DENOISE_GRAPH", "")
            use_graph = (env not in ("", "0")) if env != "" else T >= 16
        if ops.profile.open_count > 0:
            use_graph = False                                # the hook records one event per launch: needs the eager launches
        arr = (C.c_float * T)(*[float(x) for x in dts])

        def launch(lat, cx, mod, cs, sn, cond, mc, vel):
            L.check(self.lib.rf_flux_denoise(
                C.byref(d), C.byref(self.model), lat.data_ptr(), ops.ptr(cond), cx.data_ptr(),
                mod.data_ptr(), mod.stride(0), ops.ptr(mc), cs.data_ptr(), sn.data_ptr(),
                arr, T, vel.data_ptr(), C.byref(ws), ops.stream_ptr()), "rf_flux_denoise")

        if not use_graph or not self._eager_done:
            # (the first launch of a process runs eagerly even under use_graph: one-time hipFuncSetAttribute / device queries of
            #  the library must not happen inside a stream capture)
            launch(latents, ctx, mod_steps, cos, sin, cond_latents, mod_cond, torch.empty_like(latents))
            self._eager_done = True
            return latents
        key = (bytes(d), T, tuple(float(x) for x in dts), ws.base, tuple(cos.shape))
        ent = self._graphs.pop(key, None)                    # LRU: a hit moves to the most-recently-used end (re-inserted below)
        if ent is None:
            # a workload that cycles through more keys than the cache holds would re-capture T x ~300 nodes per call (a capture also
            # syncs the device): count captures per key and fall back to eager launches for a key that keeps being evicted
            # (the demotion is not for life: after 8 eager calls the key may capture again -- the working set may have shrunk -- and
            #  a replayed hit clears its count below; the book-keeping dict itself is bounded)
            n_cap, n_eager = self._graph_captures.pop(key, (0, 0))
            if n_cap >= 2 and n_eager < 8:
                self._graph_captures[key] = (n_cap, n_eager + 1)
                launch(latents, ctx, mod_steps, cos, sin, cond_latents, mod_cond, torch.empty_like(latents))
                return latents
            self._graph_captures[key] = (1 if n_cap >= 2 else n_cap + 1, 0)
            while len(self._graph_captures) > 64:
                self._graph_captures.pop(next(iter(self._graph_captures)))
            if len(self._graphs) >= self._graph_cache_max:   # a captured 50-step loop holds ~110 MB of static tables
                self._graphs.pop(next(iter(self._graphs)))
            st = dict(lat=torch.empty_like(latents), ctx=torch.empty_like(ctx), mod=torch.empty_like(mod_steps), cos=torch.empty_like(cos),
                      sin=torch.empty_like(sin), vel=torch.empty_like(latents),
                      cond=None if cond_latents is None else torch.empty_like(cond_latents),
                      mc=None if mod_cond is None else torch.empty_like(mod_cond))
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(graph, stream=side):
                launch(st["lat"], st["ctx"], st["mod"], st["cos"], st["sin"], st["cond"], st["mc"], st["vel"])
            torch.cuda.current_stream().wait_stream(side)
            ent = (graph, st)
        else:
            self._graph_captures.pop(key, None)              # replayed from the cache: it fits, forget its capture history
        self._graphs[key] = ent
        graph, st = ent
        for k_, src in (("lat", latents), ("ctx", ctx), ("mod", mod_steps), ("cos", cos), ("sin", sin), ("cond", cond_latents), ("mc", mod_cond)):
            if src is not None:
                st[k_].copy_(src)
        graph.replay()
        latents.copy_(st["lat"])
        return latents


def engine_for(transformer) -> FluxEngine:
    """The engine is cached on the transformer; call `invalidate(transformer)` after changing weights
    or loading LoRA."""
    eng = getattr(transformer, "_rf_engine", None)
    if eng is None:
        eng = FluxEngine(transformer)
        object.__setattr__(transformer, "_rf_engine", eng)
    return eng


def set_merged_lora(transformer, on: bool = True):
    """Inference-time option (cfg4 / cfg5): fold every gated LoRA into a second copy of its weight, bf16(W + s B A), used by the token
    groups LoRA acts on (condition rows; image rows too under latent_lora) -- the per-group K-segment form without its 2 launches per
    LoRA'd linear (5.5 ms of an 83 ms cfg4 forward).  +11 GB for FLUX.1-dev; the sum is rounded to bf16 once (diffusers' fuse_lora
    rounds the same way, but for ALL tokens, which would break the reference's enable_lora gating).  Call after load_lora_weights."""
    object.__setattr__(transformer, "_rf_merged_lora", bool(on))
    for b in list(transformer.transformer_blocks) + list(transformer.single_transformer_blocks):
        object.__setattr__(b, "_rf_merged_lora", bool(on))
    invalidate(transformer)


def invalidate(transformer):
    """Drop every cached packed-weight copy: the engine AND the per-block caches the public
    block_forward / single_block_forward use (they hold concatenated qkv / fused weight copies)."""
    if hasattr(transformer, "_rf_engine"):
        object.__delattr__(transformer, "_rf_engine")
    for b in list(getattr(transformer, "transformer_blocks", [])) + list(getattr(transformer, "single_transformer_blocks", [])):
        if hasattr(b, "_rf_packed"):
            object.__delattr__(b, "_rf_packed")
