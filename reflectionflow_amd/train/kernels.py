"""Tensor-level wrappers over the training entry points of librf_flux.so (include/rf_flux.h, "TRAINING path").
Every call enqueues on torch's current stream; there is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Tuple

import torch

from .. import _lib as L
from .. import ops
from ..ops import RFError, _chk, _rows2d, ptr, stream_ptr

BF = torch.bfloat16


class AttnOperands(NamedTuple):
    """What rf_qkv_train_fwd writes: the operands of the forward attention kernel (q, k, vt) and of its backward."""
    q: torch.Tensor
    k: torch.Tensor
    v: torch.Tensor
    vt: torch.Tensor
    qt: torch.Tensor
    kt: torch.Tensor
    S: int
    s_pad: int


_PARTIALS = {}


def _partials(device, D: int) -> torch.Tensor:
    key = (torch.device(device).index, stream_ptr(), D)
    if key not in _PARTIALS:
        _PARTIALS[key] = torch.empty(int(L.load().rf_train_partials_bytes(D)) // 4, dtype=torch.float32, device=device)
    return _PARTIALS[key]


def _norm_ptrs(norms):
    wq, wk, waq, wak = norms
    return ptr(_chk(wq, "norm_q")), ptr(_chk(wk, "norm_k")), ptr(waq), ptr(wak)


def qkv_train_fwd(raw: torch.Tensor, heads: int, n_added: int, norms, cos: torch.Tensor, sin: torch.Tensor, eps: float = 1e-6,
                  q_scale: float = ops.QK_PRESCALE, backward_operands: bool = True) -> AttnOperands:
    """raw [S, >= 3 * heads * 128] (columns q | k | v; a wider row pitch is fine: the single block's [q | k | v | mlp] buffer).
    norms = (norm_q.weight, norm_k.weight, norm_added_q.weight or None, norm_added_k.weight or None).
    backward_operands=False: the no-grad forward of a checkpointed block -- v / qt / kt are not written (None in the result)."""
    raw = _rows2d(raw, "raw")
    S, dev = raw.shape[0], raw.device
    _chk(cos, "cos", torch.float32), _chk(sin, "sin", torch.float32)
    if cos.shape != (S, 128) or not (cos.is_contiguous() and sin.is_contiguous()):
        raise RFError(f"rope tables must be contiguous fp32 [{S}, 128]")
    s_pad = (S + 63) // 64 * 64
    q = torch.empty(heads, s_pad, 128, dtype=BF, device=dev)
    k = torch.empty_like(q)
    vt = torch.empty(heads, s_pad // 64, 128, 64, dtype=BF, device=dev)
    v = qt = kt = None
    if backward_operands:
        v = torch.empty_like(q)
        qt = torch.empty(heads, s_pad // 32, 128, 32, dtype=BF, device=dev)
        kt = torch.empty_like(qt)
    wq, wk, waq, wak = _norm_ptrs(norms)
    L.check(L.load().rf_qkv_train_fwd(raw.data_ptr(), raw.stride(0), heads, S, s_pad, n_added, wq, wk, waq, wak, cos.data_ptr(),
                                      sin.data_ptr(), eps, q_scale, q.data_ptr(), k.data_ptr(), ptr(v), vt.data_ptr(),
                                      ptr(qt), ptr(kt), stream_ptr()), "rf_qkv_train_fwd")
    return AttnOperands(q, k, v, vt, qt, kt, S, s_pad)


def qkv_train_bwd(raw: torch.Tensor, heads: int, n_added: int, norms, cos, sin, dq, dk, dv, d_raw: torch.Tensor, eps: float = 1e-6,
                  q_scale: float = ops.QK_PRESCALE) -> torch.Tensor:
    """(dq, dk, dv) head-major -> d_raw (columns q | k | v of a caller-provided [S, >= 3 D] buffer)."""
    raw, d_raw = _rows2d(raw, "raw"), _rows2d(d_raw, "d_raw")
    S, s_pad = raw.shape[0], dq.shape[1]
    wq, wk, waq, wak = _norm_ptrs(norms)
    L.check(L.load().rf_qkv_train_bwd(raw.data_ptr(), raw.stride(0), heads, S, s_pad, n_added, wq, wk, waq, wak, cos.data_ptr(),
                                      sin.data_ptr(), eps, q_scale, _chk(dq, "dq").data_ptr(), _chk(dk, "dk").data_ptr(),
                                      _chk(dv, "dv").data_ptr(), d_raw.data_ptr(), d_raw.stride(0), stream_ptr()), "rf_qkv_train_bwd")
    return d_raw


def attention_bwd(a: AttnOperands, o: torch.Tensor, dout: torch.Tensor, lse: Optional[torch.Tensor] = None,
                  kernel: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dq (w.r.t. the scaled q), dk, dv), each [heads, s_pad, 128] bf16.  lse: the row statistics ops.attention(..., lse=) wrote for
    the SAME operands ([heads, s_pad] fp32; rows >= S are filled in here) -- without it the dq kernel makes its own pass for them.
    kernel: rf_attn_bwd_kernel bits (_lib.RF_ATTN_BWD_*), 0 = the library sizes both launches for the device."""
    o, dout = _rows2d(o, "o"), _rows2d(dout, "dout")
    H, dev = a.q.shape[0], a.q.device
    dq, dk, dv = torch.empty_like(a.q), torch.empty_like(a.q), torch.empty_like(a.q)
    dot = torch.empty_like(a.qt)
    given = lse is not None
    if given:
        _chk(lse, "lse", torch.float32)
        if lse.shape != (H, a.s_pad) or not lse.is_contiguous():
            raise RFError(f"attention_bwd: lse must be contiguous fp32 [{H}, {a.s_pad}]")
    else:
        lse = torch.empty(H, a.s_pad, dtype=torch.float32, device=dev)
    dsum = torch.empty_like(lse)
    d = L.rf_attn_bwd_desc()
    d.q, d.k, d.v, d.qt, d.kt = a.q.data_ptr(), a.k.data_ptr(), a.v.data_ptr(), a.qt.data_ptr(), a.kt.data_ptr()
    d.o, d.dout, d.ldo, d.lddo = o.data_ptr(), dout.data_ptr(), o.stride(0), dout.stride(0)
    d.dq, d.dk, d.dv, d.dot, d.lse, d.dsum = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dot.data_ptr(), lse.data_ptr(), dsum.data_ptr()
    d.heads, d.S, d.s_pad, d.mode, d.lse_given, d.kernel = H, a.S, a.s_pad, 0, 1 if given else 0, int(kernel)
    L.check(L.load().rf_attention_bwd(C.byref(d), stream_ptr()), "rf_attention_bwd")
    return dq, dk, dv


def layernorm_modulate_bwd(x, dy, scale, dres: Optional[torch.Tensor] = None, eps: float = 1e-6, out: Optional[torch.Tensor] = None,
                           need_dmod: bool = True):
    """-> (dx bf16 [rows, D] (+ dres), d_scale fp32 [D], d_shift fp32 [D]); need_dmod=False: (dx, None, None), no column reduction"""
    x, dy = _rows2d(x, "x"), _rows2d(dy, "dy")
    rows, D = x.shape
    dx = torch.empty(rows, D, dtype=BF, device=x.device) if out is None else _rows2d(out, "dx")
    dsc = torch.empty(D, dtype=torch.float32, device=x.device) if need_dmod else None
    dsh = torch.empty_like(dsc) if need_dmod else None
    part = _partials(x.device, D)
    if dres is not None:
        dres = _rows2d(dres, "dres")
    L.check(L.load().rf_layernorm_modulate_bwd(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), ptr(dres),
                                               dres.stride(0) if dres is not None else 0, dx.data_ptr(), dx.stride(0), rows, D,
                                               _chk(scale, "scale").data_ptr(), eps, ptr(dsc), ptr(dsh), part.data_ptr(),
                                               part.numel() * 4, stream_ptr()), "rf_layernorm_modulate_bwd")
    return dx, dsc, dsh


def gate_bwd(dy, f, gate, out: Optional[torch.Tensor] = None, need_dmod: bool = True):
    """y = res + gate o f  ->  (df = gate o dy  bf16, d_gate fp32 [D]); need_dmod=False: (df, None), no column reduction"""
    dy, f = _rows2d(dy, "dy"), _rows2d(f, "f")
    rows, D = dy.shape
    df = torch.empty(rows, D, dtype=BF, device=dy.device) if out is None else _rows2d(out, "df")
    dg = torch.empty(D, dtype=torch.float32, device=dy.device) if need_dmod else None
    part = _partials(dy.device, D)
    L.check(L.load().rf_gate_bwd(dy.data_ptr(), dy.stride(0), f.data_ptr(), f.stride(0), _chk(gate, "gate").data_ptr(), df.data_ptr(),
                                 df.stride(0), rows, D, ptr(dg), part.data_ptr(), part.numel() * 4, stream_ptr()), "rf_gate_bwd")
    return df, dg


def gate_residual(f, gate, res, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    f, res = _rows2d(f, "f"), _rows2d(res, "res")
    rows, D = f.shape
    out = torch.empty(rows, D, dtype=BF, device=f.device) if out is None else _rows2d(out, "out")
    L.check(L.load().rf_gate_residual(f.data_ptr(), f.stride(0), _chk(gate, "gate").data_ptr(), res.data_ptr(), res.stride(0),
                                      out.data_ptr(), out.stride(0), rows, D, stream_ptr()), "rf_gate_residual")
    return out


def gelu(z, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    z = _rows2d(z, "z")
    out = torch.empty(z.shape, dtype=BF, device=z.device) if out is None else _rows2d(out, "out")
    L.check(L.load().rf_gelu(z.data_ptr(), z.stride(0), out.data_ptr(), out.stride(0), z.shape[0], z.shape[1], stream_ptr()), "rf_gelu")
    return out


def gelu_bwd(z, dh, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    z, dh = _rows2d(z, "z"), _rows2d(dh, "dh")
    out = torch.empty(z.shape, dtype=BF, device=z.device) if out is None else _rows2d(out, "out")
    L.check(L.load().rf_gelu_bwd(z.data_ptr(), z.stride(0), dh.data_ptr(), dh.stride(0), out.data_ptr(), out.stride(0), z.shape[0],
                                 z.shape[1], stream_ptr()), "rf_gelu_bwd")
    return out


_TN_WS = {}


def gemm_tn(big: torch.Tensor, skinny: torch.Tensor, transposed: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n, j] = sum_s big[s, n] skinny[s, j] (the contraction runs over the ROWS of both operands: dB = dY^T T, dA = (x^T dT)^T
    with transposed=True -> out [R, N]).  R = skinny.shape[1], a multiple of 16."""
    big, skinny = _rows2d(big, "big"), _rows2d(skinny, "skinny")
    S, N = big.shape
    R = skinny.shape[1]
    if skinny.shape[0] != S:
        raise RFError(f"gemm_tn: {tuple(big.shape)} vs {tuple(skinny.shape)}")
    if out is None:
        out = torch.empty((R, N) if transposed else (N, R), dtype=BF, device=big.device)
    out = _rows2d(out, "out")
    lib = L.load()
    need = int(lib.rf_gemm_tn_skinny_ws_bytes(S, N, R))
    key = (big.device.index, stream_ptr())
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _TN_WS[key] = torch.empty(max(need // 4, 1 << 20), dtype=torch.float32, device=big.device)
    L.check(lib.rf_gemm_tn_skinny(big.data_ptr(), big.stride(0), skinny.data_ptr(), skinny.stride(0), out.data_ptr(), out.stride(0),
                                  S, N, R, 1 if transposed else 0, ws.data_ptr(), ws.numel() * 4, stream_ptr()), "rf_gemm_tn_skinny")
    return out


def transpose(x, rows_pad: Optional[int] = None) -> torch.Tensor:
    """[R, C] -> [C, R_pad] with zero columns R .. R_pad (R_pad defaults to R rounded up to 64: a GEMM K-segment)."""
    x = _rows2d(x, "x")
    R, Cc = x.shape
    rows_pad = (R + 63) // 64 * 64 if rows_pad is None else rows_pad
    out = torch.empty(Cc, rows_pad, dtype=BF, device=x.device)
    L.check(L.load().rf_transpose_bf16(x.data_ptr(), x.stride(0), R, Cc, out.data_ptr(), out.stride(0), rows_pad, stream_ptr()),
            "rf_transpose_bf16")
    return out
