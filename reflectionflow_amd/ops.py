"""Tensor-level wrappers over the librf_flux C ABI.

Every function takes bf16 tensors resident on a HIP device, enqueues the kernel on torch's
CURRENT stream and returns immediately.  There is no CPU path: a CPU tensor or a missing
library raises.
"""
from __future__ import annotations

import contextlib
import contextvars
import ctypes as C
import math
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import (RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_QKV, RF_EPI_QKV_GELU, RF_EPI_STORE, RFError)

__all__ = ["linear", "gemm", "build_gemm_desc", "time_gemm", "Group", "Seg", "qk_rmsnorm_rope", "attention", "layernorm_modulate",
           "euler_step_", "silu", "add_", "alloc_attn_operands", "stream_ptr", "ptr", "RFError", "QK_PRESCALE", "profile",
           "quantize_weight_fp8", "dequantize_fp8", "quant_rows_fp8", "layernorm_modulate_fp8", "gemm_w8a8", "qk_score_bound",
           "gemm_schedule", "attn_kernel"]


# Per-context defaults for the `schedule` / `kernel` field of the launch descriptors (thread- and task-local; every launch still
# carries its own value in its descriptor -- the library itself has no global switch).  Tests and tools use the context managers.
_SCHED = contextvars.ContextVar("rf_gemm_schedule", default=L.RF_SCHED_AUTO)
_ATTN_KERNEL = contextvars.ContextVar("rf_attn_kernel", default=L.RF_ATTN_AUTO)


@contextlib.contextmanager
def gemm_schedule(schedule: int):
    """`with ops.gemm_schedule(L.RF_SCHED_STREAMK): ...` -- GEMM launches built inside default to this rf_gemm_schedule."""
    tok = _SCHED.set(schedule)
    try:
        yield
    finally:
        _SCHED.reset(tok)


@contextlib.contextmanager
def attn_kernel(kernel: int):
    """`with ops.attn_kernel(L.RF_ATTN_ONLINE256): ...` -- attention launches inside default to this rf_attn_kernel."""
    tok = _ATTN_KERNEL.set(kernel)
    try:
        yield
    finally:
        _ATTN_KERNEL.reset(tok)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class profile:
    """`with ops.profile(max_launches) as pr:` -- every library kernel launched inside is timed by a hipEvent
    pair on its own launch stream (rf_profile_begin / rf_profile_end).  After the block, `pr.classes` maps
    kernel class -> {"launches", "us", "work"} (work = algorithmic FLOPs, bytes for row kernels)."""

    open_count = 0          # > 0 while a profile is open (FluxEngine.denoise then launches eagerly, never through a hipGraph)

    def __init__(self, max_launches: int = 4096):
        self.max_launches, self.classes, self.dropped = max_launches, {}, 0

    def __enter__(self):
        L.check(L.load().rf_profile_begin(self.max_launches), "rf_profile_begin")
        profile.open_count += 1
        return self

    def __exit__(self, et, ev, tb):
        n = len(L.RF_KC_NAMES)
        us, cnt, work, dropped = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), C.c_int32(0)
        profile.open_count = max(0, profile.open_count - 1)
        rc = L.load().rf_profile_end(us, cnt, work, C.byref(dropped))
        if et is None:
            L.check(rc, "rf_profile_end")
        self.dropped = dropped.value
        self.classes = {name: {"launches": int(cnt[i]), "us": float(us[i]), "work": float(work[i])}
                        for i, name in enumerate(L.RF_KC_NAMES) if cnt[i] > 0}
        return False


def _chk(t: torch.Tensor, name: str, dtype=torch.bfloat16):
    if not isinstance(t, torch.Tensor):
        raise RFError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RFError(f"{name}: tensor is on {t.device}; the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise RFError(f"{name}: dtype {t.dtype}, expected {dtype}")
    return t


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor, name: str, dtype=torch.bfloat16):
    _chk(t, name, dtype)
    if t.dim() != 2 or t.stride(1) != 1:
        raise RFError(f"{name}: need a 2-D row-major tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


class Seg:
    """One K-segment: activations A [M,K] and weights W [N,K] (nn.Linear layout); bf16, or uint8 holding
    fp8 e4m3fn bytes for gemm_w8a8."""

    def __init__(self, A: torch.Tensor, W: torch.Tensor):
        dt = torch.uint8 if A.dtype == torch.uint8 else torch.bfloat16
        self.A, self.W = _rows2d(A, "A", dt), _rows2d(W, "W", dt)
        if A.shape[1] != W.shape[1]:
            raise RFError(f"segment K mismatch: A {tuple(A.shape)} vs W {tuple(W.shape)}")


class Group:
    """One token group of a grouped GEMM."""

    def __init__(self, segs: Sequence[Seg], bias=None, out=None, residual=None, gate=None, tok_offset=0,
                 norm_q=None, norm_k=None, a_scale=None, w_scale=None):
        self.segs, self.bias, self.out, self.residual, self.gate, self.tok_offset = (
            list(segs), bias, out, residual, gate, tok_offset)
        self.norm_q, self.norm_k = norm_q, norm_k
        self.a_scale, self.w_scale = a_scale, w_scale        # gemm_w8a8: fp32 [M] / fp32 [N]


def build_gemm_desc(groups: Sequence[Group], N: int, epilogue: int = RF_EPI_STORE, n_split: int = 0,
                    q=None, k=None, vt=None, heads: int = 0, s_pad: int = 0, rope=None, norm_eps: float = 1e-6,
                    q_scale: float = 0.0, splitk_ws: Optional[torch.Tensor] = None,
                    schedule: Optional[int] = None) -> "L.rf_gemm_desc":
    """rope=(cos, sin) fp32 [S,128]: fuse per-head RMSNorm (each group's norm_q/norm_k) + RoPE into the
    QKV epilogue.  schedule: rf_gemm_schedule for THIS launch (tests pin kernels with it; the product passes AUTO)."""
    d = L.rf_gemm_desc()
    d.schedule = _SCHED.get() if schedule is None else schedule
    d.N, d.epilogue, d.num_groups, d.n_split = N, epilogue, len(groups), n_split
    d.q, d.k, d.vt, d.heads, d.s_pad = ptr(q), ptr(k), ptr(vt), heads, s_pad
    d.q_scale = q_scale
    if splitk_ws is not None:
        _chk(splitk_ws, "splitk_ws", torch.float32)
        d.splitk_ws, d.splitk_ws_bytes = ptr(splitk_ws), splitk_ws.numel() * 4
    if rope is not None:
        cos, sin = rope
        _chk(cos, "cos", torch.float32), _chk(sin, "sin", torch.float32)
        if not (cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == 128 and cos.shape == sin.shape):
            raise RFError("rope tables must be contiguous fp32 [S,128]")
        d.rope_cos, d.rope_sin, d.norm_eps = cos.data_ptr(), sin.data_ptr(), norm_eps
    for gi, g in enumerate(groups):
        G = d.g[gi]
        G.M = g.segs[0].A.shape[0]
        G.tok_offset = g.tok_offset
        for si, s in enumerate(g.segs):
            if s.A.shape[0] != G.M:
                raise RFError("all segments of a group must have the same M")
            if s.W.shape[0] != N:
                raise RFError(f"W has {s.W.shape[0]} rows, expected N={N}")
            S = G.seg[si]
            S.A, S.lda, S.W, S.ldw, S.K = s.A.data_ptr(), s.A.stride(0), s.W.data_ptr(), s.W.stride(0), s.A.shape[1]
        if g.bias is not None:
            G.bias = _chk(g.bias, "bias").data_ptr()
        if g.out is not None:
            o = _rows2d(g.out, "out")
            G.out, G.ldo = o.data_ptr(), o.stride(0)
        if g.residual is not None:
            r = _rows2d(g.residual, "residual")
            G.residual, G.ldr = r.data_ptr(), r.stride(0)
        if g.gate is not None:
            G.gate = _chk(g.gate, "gate").data_ptr()
        if g.norm_q is not None:
            G.norm_q, G.norm_k = _chk(g.norm_q, "norm_q").data_ptr(), _chk(g.norm_k, "norm_k").data_ptr()
        if g.a_scale is not None:
            if g.a_scale.numel() != G.M or g.w_scale.numel() != N or not (g.a_scale.is_contiguous() and g.w_scale.is_contiguous()):
                raise RFError("a_scale must be contiguous fp32 [M], w_scale contiguous fp32 [N]")
            G.a_scale = _chk(g.a_scale, "a_scale", torch.float32).data_ptr()
            G.w_scale = _chk(g.w_scale, "w_scale", torch.float32).data_ptr()
    return d


def gemm(groups: Sequence[Group], N: int, epilogue: int = RF_EPI_STORE, **kw):
    """Grouped GEMM launch.  splitk_ws=None (default) attaches the per-stream scratch so the library may use
    stream-K / split-K; splitk_ws=False forbids both."""
    if kw.get("splitk_ws", None) is None:
        kw["splitk_ws"] = splitk_scratch(groups[0].segs[0].A.device)
    elif kw["splitk_ws"] is False:
        kw["splitk_ws"] = None
    d = build_gemm_desc(groups, N, epilogue, **kw)
    L.check(L.load().rf_gemm_bf16(C.byref(d), stream_ptr()), "rf_gemm_bf16")


def time_gemm(groups: Sequence[Group], N: int, epilogue: int = RF_EPI_STORE, iters: int = 10, **kw) -> float:
    """Average duration (seconds) of one launch, measured with hipEvents on the launch stream."""
    if kw.get("splitk_ws", None) is None:
        kw["splitk_ws"] = splitk_scratch(groups[0].segs[0].A.device)
    elif kw["splitk_ws"] is False:
        kw["splitk_ws"] = None
    d = build_gemm_desc(groups, N, epilogue, **kw)
    us = C.c_float(0.0)
    w8 = groups[0].segs[0].A.dtype == torch.uint8
    fn = L.load().rf_time_gemm_w8a8 if w8 else L.load().rf_time_gemm
    L.check(fn(C.byref(d), iters, C.byref(us), stream_ptr()), "rf_time_gemm")
    return us.value * 1e-6


# ---- fp8 (W8A8) path: BASELINE cfg5 ------------------------------------------------------------------------------
FP8_MAX = 448.0


def quantize_weight_fp8(W: torch.Tensor):
    """nn.Linear weight [N, K] -> (uint8 [N, K] holding e4m3fn bytes, fp32 [N] per-output-channel scale):
    scale[n] = max_k |W[n,k]| / 448,  W8 = e4m3fn(W / scale) (round to nearest even, saturating)."""
    Wf = W.detach().float()
    amax = Wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    q = (Wf / scale[:, None]).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.contiguous()


def dequantize_fp8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """fp32 values of (uint8 e4m3fn bytes [R, K], fp32 [R] row scales)."""
    return q.view(torch.float8_e4m3fn).float() * scale[:, None]


def quant_rows_fp8(x0: torch.Tensor, x1: Optional[torch.Tensor] = None):
    """bf16 rows [M, K0] (| [M, K1]) -> (uint8 [M, K0+K1] e4m3fn bytes, fp32 [M] scale), one scale per row."""
    x0 = _rows2d(x0, "x0")
    K1 = 0
    if x1 is not None:
        x1 = _rows2d(x1, "x1")
        K1 = x1.shape[1]
    M, K0 = x0.shape
    out = torch.empty(M, K0 + K1, dtype=torch.uint8, device=x0.device)
    sc = torch.empty(M, dtype=torch.float32, device=x0.device)
    L.check(L.load().rf_quant_rows_fp8(x0.data_ptr(), x0.stride(0), K0, ptr(x1), x1.stride(0) if x1 is not None else 0, K1,
                                       out.data_ptr(), out.stride(0), sc.data_ptr(), M, stream_ptr()), "rf_quant_rows_fp8")
    return out, sc


def layernorm_modulate_fp8(x, scale, shift, eps: float = 1e-6):
    """LayerNorm(x)*(1+scale)+shift -> (uint8 [M, D] e4m3fn bytes, fp32 [M] row scale)."""
    x = _rows2d(x, "x")
    _chk(scale, "scale"), _chk(shift, "shift")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    sc = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    L.check(L.load().rf_layernorm_modulate_fp8(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), sc.data_ptr(),
                                               x.shape[0], x.shape[1], scale.data_ptr(), shift.data_ptr(), eps, stream_ptr()),
            "rf_layernorm_modulate_fp8")
    return out, sc


def gemm_w8a8(groups: Sequence[Group], N: int, epilogue: int = RF_EPI_STORE, **kw):
    """rf_gemm_w8a8: like gemm(), operands uint8 (e4m3fn bytes), every group carries a_scale / w_scale."""
    if kw.get("splitk_ws", None) is None:
        kw["splitk_ws"] = splitk_scratch(groups[0].segs[0].A.device)
    elif kw["splitk_ws"] is False:
        kw["splitk_ws"] = None
    d = build_gemm_desc(groups, N, epilogue, **kw)
    L.check(L.load().rf_gemm_w8a8(C.byref(d), stream_ptr()), "rf_gemm_w8a8")


def linear(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, *, epilogue: int = RF_EPI_STORE,
           residual: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
           extra: Sequence[Seg] = (), out: Optional[torch.Tensor] = None,
           splitk_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epi(x @ W^T (+ extra segments) + bias), x [M,K] bf16, W [N,K] bf16.
    splitk_ws: optional fp32 scratch; lets few-tile / long-K STORE GEMMs (LoRA down) split K over the grid."""
    x = _rows2d(x, "x")
    N = W.shape[0]
    if out is None:
        out = torch.empty(x.shape[0], N, dtype=torch.bfloat16, device=x.device)
    gemm([Group([Seg(x, W), *extra], bias=bias, out=out, residual=residual, gate=gate)], N, epilogue,
         splitk_ws=splitk_ws)
    return out


_SPLITK = {}


def splitk_scratch(device) -> torch.Tensor:
    """Per-(device, stream) GEMM scratch of the per-op API: 4 KiB of zeroed stream-K flags + 64 MiB of fp32
    partial tiles (the whole-forward engine carves its own out of the workspace)."""
    key = (torch.device(device).index, stream_ptr())
    if key not in _SPLITK:
        _SPLITK[key] = torch.zeros(1024 + (16 << 20), dtype=torch.float32, device=device)
    return _SPLITK[key]


_ATTN_WS = {}


def attn_scratch(device) -> torch.Tensor:
    """Per-(device, stream) scratch of rf_attention_fwd_ws (partial (O, l) of split query blocks)."""
    key = (torch.device(device).index, stream_ptr())
    if key not in _ATTN_WS:
        _ATTN_WS[key] = torch.empty(L.load().rf_attention_ws_bytes() // 4, dtype=torch.float32, device=device)
    return _ATTN_WS[key]


def lora_down(x: torch.Tensor, A: torch.Tensor) -> torch.Tensor:
    """T = x . lora_A^T  ([M, r_pad]); few output tiles and a long K, so it runs split-K."""
    return linear(x, A)


def alloc_attn_operands(heads: int, S: int, device) -> tuple:
    """q, k [H, S_pad, 128] and vt [H, S_pad/64, 128, 64]; zero-filled so padded keys are finite."""
    s_pad = (S + 63) // 64 * 64
    q = torch.zeros(heads, s_pad, 128, dtype=torch.bfloat16, device=device)
    k = torch.zeros_like(q)
    vt = torch.zeros(heads, s_pad // 64, 128, 64, dtype=torch.bfloat16, device=device)
    return q, k, vt, s_pad


def qk_rmsnorm_rope(q, k, S: int, n_added: int, w_q, w_k, w_added_q, w_added_k, cos, sin, eps: float = 1e-6):
    lib = L.load()
    _chk(q, "q"), _chk(k, "k"), _chk(cos, "cos", torch.float32), _chk(sin, "sin", torch.float32)
    if cos.shape != (S, 128) or sin.shape != (S, 128) or not cos.is_contiguous() or not sin.is_contiguous():
        raise RFError(f"cos/sin must be contiguous fp32 [{S},128], got {tuple(cos.shape)}")
    heads, s_pad = q.shape[0], q.shape[1]
    L.check(lib.rf_qk_rmsnorm_rope(q.data_ptr(), k.data_ptr(), heads, S, s_pad, n_added, ptr(w_q), ptr(w_k),
                                   ptr(w_added_q), ptr(w_added_k), cos.data_ptr(), sin.data_ptr(), eps,
                                   stream_ptr()), "rf_qk_rmsnorm_rope")


QK_PRESCALE = (1.0 / math.sqrt(128.0)) * 1.4426950408889634   # softmax scale * log2(e), folded into q by the QKV GEMM


def qk_score_bound(*norm_weights_qk) -> float:
    """Proven bound on |q.k / sqrt(128)| * log2(e) for per-head RMS-normalised q and k (block.py:38-41,60-67):
    |RMSNorm(x) * w| <= sqrt(128) max|w|, and RoPE is a rotation, so |q.k| <= 128 max|w_q| max|w_k|.
    Arguments: (q weights..., ) and (k weights..., ) as two tuples/lists of tensors; 2 % margin for the bf16
    rounding of q and k."""
    wq, wk = norm_weights_qk
    mq = max(float(w.detach().abs().max()) for w in wq)
    mk = max(float(w.detach().abs().max()) for w in wk)
    return 1.02 * math.sqrt(128.0) * mq * mk * 1.4426950408889634


def attention(q, k, vt, S: int, out: Optional[torch.Tensor] = None, n_main: Optional[int] = None, mode: int = 0,
              cross_bias: float = 0.0, scale: Optional[float] = None, q_prescaled: bool = False,
              score_bound: float = 0.0, scratch: bool = True, kernel: Optional[int] = None, mix_small: int = 0,
              lag_thresh: float = 0.0, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """lse: optional fp32 [heads, s_pad] output -- log2-sum-exp2 of every query's score row (rows < S), the row statistic of
    train.kernels.attention_bwd.  scratch=True attaches the per-(device, stream) scratch that lets the library split a poorly filling grid
    (rf_attention_fwd_ws); scratch=False is rf_attention_fwd.  kernel: rf_attn_kernel for THIS launch (an unrunnable
    request raises); lag_thresh: re-centring threshold of the lagged-max kernels (0 = 2^30)."""
    lib = L.load()
    _chk(q, "q"), _chk(k, "k"), _chk(vt, "vt")
    heads, s_pad = q.shape[0], q.shape[1]
    if out is None:
        out = torch.empty(S, heads * 128, dtype=torch.bfloat16, device=q.device)
    _rows2d(out, "out")
    if scale is None:
        scale = 1.0 / math.sqrt(128.0)
    # the split launch exists for the shift-free kernels only (mode 0, whole rounds of the ring, prescaled q, heads % 8 == 0):
    # do not pin 68 MiB per (device, stream) for launches that can never use it
    ws = attn_scratch(q.device) if (scratch and mode == 0 and S % 256 == 0 and q_prescaled and heads % 8 == 0) else None
    d = L.rf_attn_desc()
    d.q, d.k, d.vt, d.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.heads, d.S, d.s_pad, d.n_main, d.ldo = heads, S, s_pad, S if n_main is None else n_main, out.stride(0)
    d.mode, d.q_prescaled, d.cross_bias, d.scale = mode, 1 if q_prescaled else 0, cross_bias, scale
    d.score_bound, d.lag_thresh, d.kernel = float(score_bound), float(lag_thresh), _ATTN_KERNEL.get() if kernel is None else kernel
    d.mix_small = int(mix_small)
    d.ws, d.ws_bytes = ptr(ws), ws.numel() * 4 if ws is not None else 0
    if lse is not None:
        _chk(lse, "lse", torch.float32)
        if lse.shape != (heads, s_pad) or not lse.is_contiguous():
            raise RFError(f"attention: lse must be contiguous fp32 [{heads}, {s_pad}]")
    d.lse = ptr(lse)
    L.check(lib.rf_attention(C.byref(d), stream_ptr()), "rf_attention")
    return out


def layernorm_modulate(x, scale, shift, out: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    lib = L.load()
    x = _rows2d(x, "x")
    _chk(scale, "scale"), _chk(shift, "shift")
    if out is None:
        out = torch.empty_like(x)
    L.check(lib.rf_layernorm_modulate(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
                                      scale.data_ptr(), shift.data_ptr(), eps, stream_ptr()), "rf_layernorm_modulate")
    return out


def euler_step_(x: torch.Tensor, v: torch.Tensor, dt: float):
    lib = L.load()
    _chk(x, "x"), _chk(v, "v")
    if not (x.is_contiguous() and v.is_contiguous()) or x.numel() != v.numel():
        raise RFError("euler_step_: x and v must be contiguous and equally sized")
    L.check(lib.rf_euler_step(x.data_ptr(), v.data_ptr(), x.numel(), float(dt), stream_ptr()), "rf_euler_step")
    return x


def silu(x: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    _chk(x, "x")
    x = x.contiguous()
    out = torch.empty_like(x)
    L.check(lib.rf_silu(x.data_ptr(), out.data_ptr(), x.numel(), stream_ptr()), "rf_silu")
    return out


def add_(out: torch.Tensor, x: torch.Tensor):
    lib = L.load()
    _chk(out, "out"), _chk(x, "x")
    if not (out.is_contiguous() and x.is_contiguous()) or x.numel() != out.numel():
        raise RFError("add_: tensors must be contiguous and equally sized")
    L.check(lib.rf_add_inplace(out.data_ptr(), x.data_ptr(), x.numel(), stream_ptr()), "rf_add_inplace")
    return out
