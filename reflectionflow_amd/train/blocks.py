"""DoubleStream / SingleStream blocks with a backward, on the HIP kernels.

Reference: train_flux/flux/block.py:173-272 (block_forward), :275-333 (single_block_forward), differentiated by
torch autograd in train_flux/train/model.py:164-238 with per-block activation checkpointing
(train_flux/flux/transformer.py:139-157, config.yaml `gradient_checkpointing: true`).

Here a block is ONE torch.autograd.Function.  Its forward runs the TRAINING FORM of the block and keeps nothing but the
block's inputs; its backward runs the same training form again with `keep=True` (the recompute of the checkpoint branch --
the same launches on the same inputs, so bit-identical to the forward) and then walks the block backwards:

    dX through a frozen linear      rf_gemm_bf16 on a transposed copy of the weight (made once per weight)
    LoRA factors (A, Bs = s B)      T = x A^T is kept by the recompute; dBs = dy^T T, dT = dy Bs, dA = dT^T x as GEMMs over the
                                    token axis on rf_transpose_bf16'd operands; dT A joins the dX GEMM as a K-segment
    joint attention                 rf_attention_bwd (+ rf_qkv_train_bwd for per-head RMSNorm + RoPE)
    AdaLN-Zero modulate / gates     rf_layernorm_modulate_bwd, rf_gate_bwd (their column sums are the modulation gradients)
    GELU                            rf_gelu_bwd

The training form differs from the inference block (rf_double_block_fwd) only in WHERE bf16 roundings happen: every
projection is stored (RF_EPI_STORE) before the gate / GELU / RMSNorm + RoPE that follows it, because the backward needs
those pre-activation values -- which is also exactly where torch's bf16 autograd rounds.  The three token streams (text /
image / condition) share every GEMM launch as token groups, as in the inference path; LoRA acts on the condition rows, and
on the image rows iff model_config["latent_lora"] (lora_controller.py:5-42).
"""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import torch

from .. import _lib as L
from .. import engine as E
from .. import ops
from ..ops import RF_EPI_STORE, Group, Seg
from . import kernels as K

BF = torch.bfloat16


# ------------------------------------------------------------------------------------------------------------------ weights
class TrainWeights:
    """Fused frozen weights of one block (shared with the inference pack) + lazily made transposed copies."""

    def __init__(self, pk: "E._Packed"):
        self.pk = pk
        self.D, self.heads, self.mlp = pk.D, pk.heads, pk.mlp
        self._t: Dict[Tuple[str, Optional[Tuple[int, int]]], torch.Tensor] = {}

    def w(self, name: str) -> Optional[torch.Tensor]:
        p = getattr(self.pk.struct, name)
        return None if not p else self.pk.by_ptr[p]

    def wT(self, name: str, cols: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """[K, N] copy of weight `name` ([N, K]); `cols` = a column range of the weight (a K-slice of the forward)."""
        key = (name, cols)
        if key not in self._t:
            w = self.w(name)
            if cols is not None:
                w = w[:, cols[0]:cols[1]]
            self._t[key] = K.transpose(w, rows_pad=w.shape[0])
        return self._t[key]


def double_weights(block) -> TrainWeights:
    tw = getattr(block, "_rf_train", None)
    if tw is None or tw.pk is not getattr(block, "_rf_packed", None):
        tw = TrainWeights(E.pack_double_block(block))
        object.__setattr__(block, "_rf_train", tw)
    return tw


def single_weights(block) -> TrainWeights:
    tw = getattr(block, "_rf_train", None)
    if tw is None or tw.pk is not getattr(block, "_rf_packed", None):
        tw = TrainWeights(E.pack_single_block(block))
        object.__setattr__(block, "_rf_train", tw)
    return tw


class _FuseLora(torch.autograd.Function):
    """(A_pad [r_pad, K], Bs_pad [N, r_pad]) from the LoRA parameters of sibling linears: A = the lora_A weights stacked along r,
    Bs = block-diagonal of scaling * lora_B (one block per linear along N, one column range per (linear, adapter) along r).  ONE launch
    each way (rf_lora_fuse / rf_lora_unfuse_grads, ABI v14).  When a flat bucket of train/optim.py owns the factors, the backward ADDS
    their gradients into the bucket's `.grad` views itself and hands autograd None for them: a scaled copy plus two AccumulateGrad
    launches per linear and site were ~720 launches of a step; the arithmetic is that of `grad += dBs_block * scaling` on bf16 tensors,
    rounding for rounding.  Otherwise the gradients are returned to autograd like any other node's."""

    @staticmethod
    def _table(entries, params, grads: bool):
        tab = (L.rf_lora_fuse_entry * len(entries))()
        for i, (n0, n, r0, r, sc) in enumerate(entries):
            wa, wb = params[2 * i], params[2 * i + 1]
            e = tab[i]
            e.A, e.B, e.n0, e.n, e.r0, e.r, e.scaling = wa.data_ptr(), wb.data_ptr(), n0, n, r0, r, sc
            if grads:
                for p_ in (wa, wb):
                    if p_.requires_grad and p_.grad is None:
                        p_.grad = torch.zeros_like(p_)
                e.dA = wa.grad.data_ptr() if wa.requires_grad else None
                e.dB = wb.grad.data_ptr() if wb.requires_grad else None
        return tab

    @staticmethod
    def forward(ctx, layout, *params):
        # layout: (K, N, r_pad, [(n0, n, r0, r, scaling) per (linear, adapter), in parameter order]); params = A_0, B_0, A_1, B_1, ...
        K_, N, r_pad, entries = layout
        ref = params[0]
        if len(entries) > L.RF_LORA_FUSE_MAX or any(p_.dtype != BF or not p_.is_contiguous() for p_ in params):
            raise ops.RFError(f"fused LoRA: up to {L.RF_LORA_FUSE_MAX} contiguous bf16 (linear, adapter) pairs per site, got {len(entries)}")
        A = torch.empty(r_pad, K_, dtype=BF, device=ref.device)
        B = torch.empty(N, r_pad, dtype=BF, device=ref.device)
        tab = _FuseLora._table(entries, params, False)
        L.check(L.load().rf_lora_fuse(tab, len(entries), K_, N, r_pad, A.data_ptr(), B.data_ptr(), ops.stream_ptr()), "rf_lora_fuse")
        ctx.layout, ctx.params = layout, params
        return A, B

    @staticmethod
    def backward(ctx, dA, dB):
        K_, N, r_pad, entries = ctx.layout
        dA, dB = dA.contiguous(), dB.contiguous()
        params = ctx.params
        # The in-place form (add into .grad, hand autograd None) is a contract between this node and the flat bucket that owns the
        # factors (train/optim.py::FlatLoraBucket marks them): there `.grad` IS the view the optimizer reads.  Parameters nobody
        # re-homed get the ordinary autograd contract -- the gradients are RETURNED (one launch, accumulate = 0, into fresh tensors),
        # so torch.autograd.grad(), backward(inputs=...), tensor hooks and post-accumulate-grad hooks (DDP, hook-based clipping) see
        # them.
        owned = all(getattr(p_, "_rf_flat_bucket", None) is not None for p_ in params if p_.requires_grad)
        with torch.no_grad():
            if owned:
                tab = _FuseLora._table(entries, params, True)
                L.check(L.load().rf_lora_unfuse_grads(tab, len(entries), K_, N, r_pad, dA.data_ptr(), dB.data_ptr(), 1, ops.stream_ptr()),
                        "rf_lora_unfuse_grads")
                return (None,) * (1 + len(params))
            tab = _FuseLora._table(entries, params, False)
            grads = [torch.empty_like(p_) if (p_.requires_grad and ctx.needs_input_grad[1 + i]) else None for i, p_ in enumerate(params)]
            for i in range(len(entries)):
                tab[i].dA = grads[2 * i].data_ptr() if grads[2 * i] is not None else None
                tab[i].dB = grads[2 * i + 1].data_ptr() if grads[2 * i + 1] is not None else None
            L.check(L.load().rf_lora_unfuse_grads(tab, len(entries), K_, N, r_pad, dA.data_ptr(), dB.data_ptr(), 0, ops.stream_ptr()),
                    "rf_lora_unfuse_grads")
        return (None, *grads)


def fused_lora(linears, device=None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(A [r_pad, K], Bs [N, r_pad]) of sibling linears fused along N -- the layout of engine._fused_lora -- as ONE autograd node over
    the modules' PARAMETERS, so the gradients of the fused tensors reach every lora_A / lora_B."""
    from ..flux.modules import LoraLinear
    if not any(isinstance(l, LoraLinear) for l in linears):
        return None, None
    entries, params = [], []
    n0 = r0 = 0
    for l in linears:
        if isinstance(l, LoraLinear):
            for a in l.active_adapters:
                wa, wb = l.lora_A[a].weight, l.lora_B[a].weight
                entries.append((n0, l.out_features, r0, wa.shape[0], float(l.scaling[a])))
                params += [wa, wb]
                r0 += wa.shape[0]
        n0 += l.out_features
    r_pad = (r0 + 63) // 64 * 64
    return _FuseLora.apply((linears[0].in_features, n0, r_pad, tuple(entries)), *params)


# ------------------------------------------------------------------------------------------------------------------ helpers
class _Stream:
    __slots__ = ("x", "mod", "rows", "sl", "lora", "qkv", "out", "ff1", "ff2")

    def __init__(self, x, mod, off, lora, qkv, out, ff1, ff2):
        self.x, self.mod, self.rows, self.sl, self.lora = x, mod, x.shape[0], slice(off, off + x.shape[0]), lora
        self.qkv, self.out, self.ff1, self.ff2 = qkv, out, ff1, ff2      # weight names (w, b) of this stream


def _grouped(streams, A_buf, names, N, out_buf, loras=None, keep=None, tag="", tc=None):
    """One grouped STORE launch: per stream  out = A W^T + b (+ (A lora_A^T) lora_Bs^T on LoRA streams).  tc: the block's cache of
    LoRA down-projections T = A lora_A^T -- the forward fills it, the recompute inside the backward takes them from it (they are a
    few hundred KB per site, and the only part of a block the checkpoint keeps besides its inputs)."""
    groups = []
    for i, (s, (tw, wn, bn)) in enumerate(zip(streams, names)):
        a = A_buf[s.sl]
        segs = [Seg(a, tw.w(wn))]
        if s.lora and loras is not None and loras[0] is not None:
            key = f"T_{tag}_{i}"
            t = tc.get(key) if tc is not None else None
            if t is None:
                t = ops.linear(a, loras[0])
                if tc is not None:
                    tc[key] = t
            segs.append(Seg(t, loras[1]))
            if keep is not None:
                keep[f"T_{tag}_{i}"] = t
        groups.append(Group(segs, bias=tw.w(bn), out=out_buf[s.sl]))
    ops.gemm(groups, N, RF_EPI_STORE)


def _lora_bwd(x_segs: Sequence[torch.Tensor], T: torch.Tensor, dy: torch.Tensor, A: torch.Tensor, Bs: torch.Tensor):
    """y += (sum_s x_s A_s^T) Bs^T with A = [A_0 | A_1 ...] along K.  -> (dA [r_pad, K], dBs [N, r_pad], dT [M, r_pad])."""
    dBs = K.gemm_tn(dy, T)                                      # [N, r_pad] = dy^T T   (contraction over the tokens, operands as they lie)
    dT = ops.linear(dy, K.transpose(Bs, rows_pad=Bs.shape[0]))  # [M, r_pad] = dy Bs
    dA = torch.empty_like(A)
    k0 = 0
    for x in x_segs:
        K.gemm_tn(x, dT, transposed=True, out=dA[:, k0:k0 + x.shape[1]])   # [r_pad, K_s] = dT^T x_s
        k0 += x.shape[1]
    return dA, dBs, dT


def _dx_grouped(streams, dY_buf, names_T, N, out_buf, lora_dT=None, lora_A=None):
    """dX = dY W (+ dT A on LoRA streams) for all streams in one launch; names_T[i] = (tw, weight name, column range)."""
    groups = []
    AT = K.transpose(lora_A, rows_pad=lora_A.shape[0]) if lora_A is not None else None      # [K, r_pad]
    for i, (s, (tw, wn, cols)) in enumerate(zip(streams, names_T)):
        segs = [Seg(dY_buf[s.sl], tw.wT(wn, cols))]
        if lora_dT is not None and lora_dT.get(i) is not None:
            segs.append(Seg(lora_dT[i], AT))
        groups.append(Group(segs, out=out_buf[s.sl]))
    ops.gemm(groups, N, RF_EPI_STORE)


def _acc(dst: Optional[torch.Tensor], g: torch.Tensor) -> torch.Tensor:
    return g if dst is None else dst + g


# ------------------------------------------------------------------------------------------------------------------ DoubleStream
def _double_streams(tw: TrainWeights, x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, latent_lora: bool) -> List[_Stream]:
    D = tw.D
    chunks = lambda m: [m[i * D:(i + 1) * D] for i in range(6)]   # noqa: E731  (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp)
    main = dict(qkv=("w_qkv", "b_qkv"), out=("w_out", "b_out"), ff1=("w_ff1", "b_ff1"), ff2=("w_ff2", "b_ff2"))
    ctx = dict(qkv=("w_add_qkv", "b_add_qkv"), out=("w_add_out", "b_add_out"), ff1=("w_ffc1", "b_ffc1"), ff2=("w_ffc2", "b_ffc2"))
    st = [_Stream(x_txt, chunks(mod_txt), 0, False, **ctx),
          _Stream(x_img, chunks(mod_img), x_txt.shape[0], bool(latent_lora), **main)]
    if x_cond is not None:
        st.append(_Stream(x_cond, chunks(mod_cond), x_txt.shape[0] + x_img.shape[0], True, **main))
    return st


def double_train_forward(tw: TrainWeights, x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin, lora: dict,
                         latent_lora: bool = False, keep: Optional[dict] = None, tc: Optional[dict] = None):
    """Training form of block.py:173-272 for ONE sample: x_* [rows, D] bf16, mod_* [6 D] bf16, lora = {"qkv": (A, Bs), "out": ...,
    "ff2": ...} (entries may be (None, None)).  Returns [y_txt, y_img, y_cond?]; `keep` (a dict) receives the intermediates."""
    D, H, mlp = tw.D, tw.heads, tw.mlp
    st = _double_streams(tw, x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, latent_lora)
    S, dev = sum(s.rows for s in st), x_img.device
    new = lambda *shape: torch.empty(*shape, dtype=BF, device=dev)   # noqa: E731
    XN, RAW = new(S, D), new(S, 3 * D)
    for s in st:
        ops.layernorm_modulate(s.x, s.mod[1], s.mod[0], out=XN[s.sl])
    _grouped(st, XN, [(tw,) + s.qkv for s in st], 3 * D, RAW, lora.get("qkv"), keep, "qkv", tc)
    norms = (tw.w("norm_q"), tw.w("norm_k"), tw.w("norm_added_q"), tw.w("norm_added_k"))
    a = K.qkv_train_fwd(RAW, H, st[0].rows, norms, cos, sin, backward_operands=keep is not None)
    # (recompute inside the backward: the forward kernel also emits the row statistics its backward needs)
    LSE = torch.empty(H, a.s_pad, dtype=torch.float32, device=dev) if keep is not None else None
    ATT = ops.attention(a.q, a.k, a.vt, S, q_prescaled=True, score_bound=0.0, lse=LSE)
    AOUT = new(S, D)
    _grouped(st, ATT, [(tw,) + s.out for s in st], D, AOUT, lora.get("out"), keep, "out", tc)
    X1, XN2 = new(S, D), new(S, D)
    for s in st:
        K.gate_residual(AOUT[s.sl], s.mod[2], s.x, out=X1[s.sl])
        ops.layernorm_modulate(X1[s.sl], s.mod[4], s.mod[3], out=XN2[s.sl])
    Z1 = new(S, mlp)
    _grouped(st, XN2, [(tw,) + s.ff1 for s in st], mlp, Z1)
    HH = K.gelu(Z1)
    FF = new(S, D)
    _grouped(st, HH, [(tw,) + s.ff2 for s in st], D, FF, lora.get("ff2"), keep, "ff2", tc)
    Y = new(S, D)
    for s in st:
        K.gate_residual(FF[s.sl], s.mod[5], X1[s.sl], out=Y[s.sl])
    if keep is not None:
        keep.update(st=st, XN=XN, RAW=RAW, a=a, ATT=ATT, LSE=LSE, AOUT=AOUT, X1=X1, XN2=XN2, Z1=Z1, HH=HH, FF=FF, norms=norms)
    return [Y[s.sl] for s in st]


def double_train_backward(tw: TrainWeights, kp: dict, dys: Sequence[torch.Tensor], cos, sin, lora: dict,
                          need_dmod: Optional[Sequence[bool]] = None):
    """-> (dxs [per stream], dmods [per stream, bf16 [6 D] or None], dlora {"qkv": (dA, dBs) | None, ...}).  need_dmod[i] = False:
    stream i's modulation rows need no gradient (no LoRA on its AdaLN linear) -- its six column reductions are not launched."""
    D, H, mlp = tw.D, tw.heads, tw.mlp
    st: List[_Stream] = kp["st"]
    S, dev = kp["XN"].shape[0], kp["XN"].device
    new = lambda *shape: torch.empty(*shape, dtype=BF, device=dev)   # noqa: E731
    dlora: Dict[str, Optional[Tuple[torch.Tensor, torch.Tensor]]] = {"qkv": None, "out": None, "ff2": None}
    dmod = [[None] * 6 for _ in st]
    nd = [True] * len(st) if need_dmod is None else list(need_dmod)

    def lora_of(tag, x_buf, dy_buf):
        """LoRA gradients of linear `tag` over its LoRA streams; returns {stream index: dT} for the dX GEMM."""
        A, Bs = lora.get(tag, (None, None))
        dTs = {}
        if A is None:
            return dTs, None
        dA_sum = dB_sum = None
        for i, s in enumerate(st):
            if not s.lora:
                continue
            dA, dBs, dT = _lora_bwd([x_buf[s.sl]], kp[f"T_{tag}_{i}"], dy_buf[s.sl], A, Bs)
            dA_sum, dB_sum, dTs[i] = _acc(dA_sum, dA), _acc(dB_sum, dBs), dT
        dlora[tag] = (dA_sum, dB_sum)
        return dTs, A

    # y = x1 + gate_mlp o f
    DF = new(S, D)
    for i, (s, dy) in enumerate(zip(st, dys)):
        _, dmod[i][5] = K.gate_bwd(dy.contiguous(), kp["FF"][s.sl], s.mod[5], out=DF[s.sl], need_dmod=nd[i])
    dTs, A = lora_of("ff2", kp["HH"], DF)
    DH = new(S, mlp)
    _dx_grouped(st, DF, [(tw, s.ff2[0], None) for s in st], mlp, DH, dTs, A)
    DZ1 = K.gelu_bwd(kp["Z1"], DH)
    DXN2 = new(S, D)
    _dx_grouped(st, DZ1, [(tw, s.ff1[0], None) for s in st], D, DXN2)
    DX1 = new(S, D)
    for i, (s, dy) in enumerate(zip(st, dys)):
        _, dmod[i][4], dmod[i][3] = K.layernorm_modulate_bwd(kp["X1"][s.sl], DXN2[s.sl], s.mod[4], dres=dy.contiguous(), out=DX1[s.sl],
                                                             need_dmod=nd[i])
    # x1 = x + gate_msa o a_out
    DAO = new(S, D)
    for i, s in enumerate(st):
        _, dmod[i][2] = K.gate_bwd(DX1[s.sl], kp["AOUT"][s.sl], s.mod[2], out=DAO[s.sl], need_dmod=nd[i])
    dTs, A = lora_of("out", kp["ATT"], DAO)
    DATT = new(S, D)
    _dx_grouped(st, DAO, [(tw, s.out[0], None) for s in st], D, DATT, dTs, A)
    dq, dk, dv = K.attention_bwd(kp["a"], kp["ATT"], DATT, lse=kp.get("LSE"))
    DRAW = new(S, 3 * D)
    K.qkv_train_bwd(kp["RAW"], H, st[0].rows, kp["norms"], cos, sin, dq, dk, dv, DRAW)
    dTs, A = lora_of("qkv", kp["XN"], DRAW)
    DXN = new(S, D)
    _dx_grouped(st, DRAW, [(tw, s.qkv[0], None) for s in st], D, DXN, dTs, A)
    dxs = []
    for i, s in enumerate(st):
        dx, dmod[i][1], dmod[i][0] = K.layernorm_modulate_bwd(s.x, DXN[s.sl], s.mod[1], dres=DX1[s.sl], need_dmod=nd[i])
        dxs.append(dx)
    return dxs, [torch.cat(m).to(BF) if nd[i] else None for i, m in enumerate(dmod)], dlora


class BlockOpts(NamedTuple):
    """Second argument of the block Functions (a plain bool is read as latent_lora, recompute=True).
    recompute=True is the reference's `gradient_checkpointing: true` (train_flux/flux/transformer.py:139-157): a block keeps its inputs
    and runs its forward again inside the backward.  recompute=False keeps the forward's intermediates instead (~0.8 GB per block at
    5632 tokens, ~45 GB per sample of the 57-block model -- what 288 GB of HBM are for): the same kernels on the same operands in the
    backward, so the gradients are bit-identical, with one forward less per block and step."""
    latent_lora: bool = False
    recompute: bool = True


def _opts(o) -> BlockOpts:
    return o if isinstance(o, BlockOpts) else BlockOpts(bool(o), True)


class DoubleBlockFn(torch.autograd.Function):
    """y_txt, y_img, y_cond = DoubleBlockFn.apply(tw, BlockOpts(latent_lora, recompute), x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin,
    A_qkv, B_qkv, A_out, B_out, A_ff2, B_ff2)    (x_cond / mod_cond / any LoRA pair may be None)"""

    @staticmethod
    def forward(ctx, tw, opts, x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin, *lo):
        latent_lora, recompute = _opts(opts)
        lora = {"qkv": (lo[0], lo[1]), "out": (lo[2], lo[3]), "ff2": (lo[4], lo[5])}
        ctx.tc = {}
        ctx.kp = None if recompute else {}
        ys = double_train_forward(tw, x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin, lora, latent_lora, keep=ctx.kp, tc=ctx.tc)
        ctx.tw, ctx.latent_lora, ctx.has_cond = tw, latent_lora, x_cond is not None
        ctx.save_for_backward(x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin, *lo)
        return (ys[0], ys[1], ys[2] if x_cond is not None else None)

    @staticmethod
    def backward(ctx, dy_txt, dy_img, dy_cond):
        x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin, *lo = ctx.saved_tensors
        lora = {"qkv": (lo[0], lo[1]), "out": (lo[2], lo[3]), "ff2": (lo[4], lo[5])}
        kp = ctx.kp
        ctx.kp = None                                            # (the intermediates die with this call)
        if kp is None:
            kp = {}
            double_train_forward(ctx.tw, x_txt, x_img, x_cond, mod_txt, mod_img, mod_cond, cos, sin, lora, ctx.latent_lora, keep=kp, tc=ctx.tc)   # recompute
        zeros = lambda x: torch.zeros_like(x)   # noqa: E731
        dys = [dy_txt if dy_txt is not None else zeros(x_txt), dy_img if dy_img is not None else zeros(x_img)]
        if ctx.has_cond:
            dys.append(dy_cond if dy_cond is not None else zeros(x_cond))
        nd = [ctx.needs_input_grad[5], ctx.needs_input_grad[6]] + ([ctx.needs_input_grad[7]] if ctx.has_cond else [])
        dxs, dmods, dl = double_train_backward(ctx.tw, kp, dys, cos, sin, lora, need_dmod=nd)
        g = lambda tag, j: None if dl[tag] is None else dl[tag][j]   # noqa: E731
        return (None, None, dxs[0], dxs[1], dxs[2] if ctx.has_cond else None, dmods[0], dmods[1], dmods[2] if ctx.has_cond else None,
                None, None, g("qkv", 0), g("qkv", 1), g("out", 0), g("out", 1), g("ff2", 0), g("ff2", 1))


# ------------------------------------------------------------------------------------------------------------------ SingleStream
def single_train_forward(tw: TrainWeights, x_main, x_cond, mod_main, mod_cond, cos, sin, lora: dict, latent_lora: bool = False,
                         keep: Optional[dict] = None, tc: Optional[dict] = None):
    """Training form of block.py:275-333 for ONE sample: x_main = [text; image] rows, mod_* [3 D] = (shift, scale, gate);
    lora = {"qkv_mlp": (A [r, D], Bs [3 D + mlp, r]), "out": (A [r, D + mlp], Bs [D, r])}."""
    D, H, mlp = tw.D, tw.heads, tw.mlp
    ch = lambda m: [m[i * D:(i + 1) * D] for i in range(3)]   # noqa: E731
    names = dict(qkv=("w_qkv_mlp", "b_qkv_mlp"), out=("w_out", "b_out"), ff1=None, ff2=None)
    st = [_Stream(x_main, ch(mod_main), 0, bool(latent_lora), **names)]
    if x_cond is not None:
        st.append(_Stream(x_cond, ch(mod_cond), x_main.shape[0], True, **names))
    S, dev = sum(s.rows for s in st), x_main.device
    new = lambda *shape: torch.empty(*shape, dtype=BF, device=dev)   # noqa: E731
    XN, Z = new(S, D), new(S, 3 * D + mlp)
    for s in st:
        ops.layernorm_modulate(s.x, s.mod[1], s.mod[0], out=XN[s.sl])
    _grouped(st, XN, [(tw,) + s.qkv for s in st], 3 * D + mlp, Z, lora.get("qkv_mlp"), keep, "qkv_mlp", tc)
    norms = (tw.w("norm_q"), tw.w("norm_k"), None, None)
    a = K.qkv_train_fwd(Z, H, 0, norms, cos, sin, backward_operands=keep is not None)
    # (recompute inside the backward: the forward kernel also emits the row statistics its backward needs)
    LSE = torch.empty(H, a.s_pad, dtype=torch.float32, device=dev) if keep is not None else None
    ATT = ops.attention(a.q, a.k, a.vt, S, q_prescaled=True, score_bound=0.0, lse=LSE)
    HM = K.gelu(Z[:, 3 * D:])
    FF = new(S, D)
    A_o, B_o = lora.get("out", (None, None))
    groups = []
    w_out = tw.w("w_out")
    for i, s in enumerate(st):
        segs = [Seg(ATT[s.sl], w_out[:, :D]), Seg(HM[s.sl], w_out[:, D:])]
        if s.lora and A_o is not None:
            t = tc.get(f"T_out_{i}") if tc is not None else None
            if t is None:
                t = torch.empty(s.rows, A_o.shape[0], dtype=BF, device=dev)
                ops.gemm([Group([Seg(ATT[s.sl], A_o[:, :D]), Seg(HM[s.sl], A_o[:, D:])], out=t)], A_o.shape[0], RF_EPI_STORE)
                if tc is not None:
                    tc[f"T_out_{i}"] = t
            segs.append(Seg(t, B_o))
            if keep is not None:
                keep[f"T_out_{i}"] = t
        groups.append(Group(segs, bias=tw.w("b_out"), out=FF[s.sl]))
    ops.gemm(groups, D, RF_EPI_STORE)
    Y = new(S, D)
    for s in st:
        K.gate_residual(FF[s.sl], s.mod[2], s.x, out=Y[s.sl])
    if keep is not None:
        keep.update(st=st, XN=XN, Z=Z, a=a, ATT=ATT, LSE=LSE, HM=HM, FF=FF, norms=norms)
    return [Y[s.sl] for s in st]


def single_train_backward(tw: TrainWeights, kp: dict, dys: Sequence[torch.Tensor], cos, sin, lora: dict,
                          need_dmod: Optional[Sequence[bool]] = None):
    D, H, mlp = tw.D, tw.heads, tw.mlp
    st: List[_Stream] = kp["st"]
    S, dev = kp["XN"].shape[0], kp["XN"].device
    new = lambda *shape: torch.empty(*shape, dtype=BF, device=dev)   # noqa: E731
    dlora: Dict[str, Optional[Tuple[torch.Tensor, torch.Tensor]]] = {"qkv_mlp": None, "out": None}
    dmod = [[None] * 3 for _ in st]
    nd = [True] * len(st) if need_dmod is None else list(need_dmod)
    DF = new(S, D)
    for i, (s, dy) in enumerate(zip(st, dys)):
        _, dmod[i][2] = K.gate_bwd(dy.contiguous(), kp["FF"][s.sl], s.mod[2], out=DF[s.sl], need_dmod=nd[i])
    # f = [att | hm] W_out^T (+ LoRA):  d[att | hm] = df W_out (+ dT A)
    A_o, B_o = lora.get("out", (None, None))
    dTs = {}
    if A_o is not None:
        dA_sum = dB_sum = None
        for i, s in enumerate(st):
            if s.lora:
                dA, dBs, dT = _lora_bwd([kp["ATT"][s.sl], kp["HM"][s.sl]], kp[f"T_out_{i}"], DF[s.sl], A_o, B_o)
                dA_sum, dB_sum, dTs[i] = _acc(dA_sum, dA), _acc(dB_sum, dBs), dT
        dlora["out"] = (dA_sum, dB_sum)
    DCAT = new(S, D + mlp)
    _dx_grouped(st, DF, [(tw, "w_out", None) for _ in st], D + mlp, DCAT, dTs, A_o)
    DZ = new(S, 3 * D + mlp)
    K.gelu_bwd(kp["Z"][:, 3 * D:], DCAT[:, D:], out=DZ[:, 3 * D:])
    dq, dk, dv = K.attention_bwd(kp["a"], kp["ATT"], DCAT[:, :D], lse=kp.get("LSE"))
    K.qkv_train_bwd(kp["Z"], H, 0, kp["norms"], cos, sin, dq, dk, dv, DZ)
    A_q, B_q = lora.get("qkv_mlp", (None, None))
    dTs = {}
    if A_q is not None:
        dA_sum = dB_sum = None
        for i, s in enumerate(st):
            if s.lora:
                dA, dBs, dT = _lora_bwd([kp["XN"][s.sl]], kp[f"T_qkv_mlp_{i}"], DZ[s.sl], A_q, B_q)
                dA_sum, dB_sum, dTs[i] = _acc(dA_sum, dA), _acc(dB_sum, dBs), dT
        dlora["qkv_mlp"] = (dA_sum, dB_sum)
    DXN = new(S, D)
    _dx_grouped(st, DZ, [(tw, "w_qkv_mlp", None) for _ in st], D, DXN, dTs, A_q)
    dxs = []
    for i, (s, dy) in enumerate(zip(st, dys)):
        dx, dmod[i][1], dmod[i][0] = K.layernorm_modulate_bwd(s.x, DXN[s.sl], s.mod[1], dres=dy.contiguous(), need_dmod=nd[i])
        dxs.append(dx)
    return dxs, [torch.cat(m).to(BF) if nd[i] else None for i, m in enumerate(dmod)], dlora


class SingleBlockFn(torch.autograd.Function):
    """y_main, y_cond = SingleBlockFn.apply(tw, BlockOpts(latent_lora, recompute), x_main, x_cond, mod_main, mod_cond, cos, sin, A_qkv_mlp, B_qkv_mlp, A_out, B_out)"""

    @staticmethod
    def forward(ctx, tw, opts, x_main, x_cond, mod_main, mod_cond, cos, sin, *lo):
        latent_lora, recompute = _opts(opts)
        lora = {"qkv_mlp": (lo[0], lo[1]), "out": (lo[2], lo[3])}
        ctx.tc = {}
        ctx.kp = None if recompute else {}
        ys = single_train_forward(tw, x_main, x_cond, mod_main, mod_cond, cos, sin, lora, latent_lora, keep=ctx.kp, tc=ctx.tc)
        ctx.tw, ctx.latent_lora, ctx.has_cond = tw, latent_lora, x_cond is not None
        ctx.save_for_backward(x_main, x_cond, mod_main, mod_cond, cos, sin, *lo)
        return (ys[0], ys[1] if x_cond is not None else None)

    @staticmethod
    def backward(ctx, dy_main, dy_cond):
        x_main, x_cond, mod_main, mod_cond, cos, sin, *lo = ctx.saved_tensors
        lora = {"qkv_mlp": (lo[0], lo[1]), "out": (lo[2], lo[3])}
        kp = ctx.kp
        ctx.kp = None
        if kp is None:
            kp = {}
            single_train_forward(ctx.tw, x_main, x_cond, mod_main, mod_cond, cos, sin, lora, ctx.latent_lora, keep=kp, tc=ctx.tc)    # recompute
        dys = [dy_main if dy_main is not None else torch.zeros_like(x_main)]
        if ctx.has_cond:
            dys.append(dy_cond if dy_cond is not None else torch.zeros_like(x_cond))
        nd = [ctx.needs_input_grad[4]] + ([ctx.needs_input_grad[5]] if ctx.has_cond else [])
        dxs, dmods, dl = single_train_backward(ctx.tw, kp, dys, cos, sin, lora, need_dmod=nd)
        g = lambda tag, j: None if dl[tag] is None else dl[tag][j]   # noqa: E731
        return (None, None, dxs[0], dxs[1] if ctx.has_cond else None, dmods[0], dmods[1] if ctx.has_cond else None, None, None,
                g("qkv_mlp", 0), g("qkv_mlp", 1), g("out", 0), g("out", 1))
