"""Optimizers of the LoRA factors over ONE flat bucket (train_flux/train/model.py:105-117: `torch.optim.AdamW(self.trainable_params,
**params)` or `prodigyopt.Prodigy(...)`; config.yaml:55-61 ships Prodigy lr 1, use_bias_correction, safeguard_warmup, weight_decay 0.01).

`FlatLoraBucket` re-homes every trainable parameter as a view into one contiguous bf16 buffer and every gradient as a view into a
second one (autograd accumulates into an existing .grad in place, so the backward fills the buffer directly).  Consequences:

  * the data-parallel reduction is `all_reduce(bucket.grad)` on the buffer as it lies -- no pack, no unpack, no per-tensor averaging:
    the 1 / world_size rides into the update kernel as `grad_scale`  (round 4: zeros + 354 slice copies in, 354 copies out);
  * zero_grad is one memset, the update is one launch (AdamW) or three (Prodigy) of csrc/optim.hip instead of a multi-tensor foreach;
  * Prodigy's distance estimate d stays on the device: nothing crosses to the host inside a step (prodigyopt reads one .item() per
    parameter tensor and step, i.e. 354 host syncs).

The classes keep torch.optim's calling convention (`opt.step()`, `opt.zero_grad()`, `opt.state_dict()`), which is all the reference's
Lightning loop uses (model.py:105-139)."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch

from .. import _lib as L
from ..ops import RFError, stream_ptr

BF = torch.bfloat16


class FlatLoraBucket:
    """params -> views of `self.param` (bf16 [n]); gradients -> views of `self.grad`.  Call `engine.invalidate(transformer)` afterwards
    if an inference engine packed pointers to the old storage (FluxTrainer does)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise RFError("FlatLoraBucket: no trainable parameter")
        p0 = self.params[0]
        if any(p.dtype != BF or p.device != p0.device for p in self.params):
            raise RFError("FlatLoraBucket: parameters must be bf16 on one device")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8                       # every view starts 16-byte aligned (rf_gemm_bf16 operands)
        self.numel = off
        self.param = torch.zeros(off, dtype=BF, device=p0.device)
        self.grad = torch.zeros(off, dtype=BF, device=p0.device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                old_grad = p.grad
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                if old_grad is not None:
                    p.grad.copy_(old_grad)
                p._rf_flat_bucket = True          # train/blocks.py::_FuseLora may add into .grad in place: it IS the optimizer's buffer

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):               # (someone may have set p.grad = None: re-attach the views)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 2 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def all_reduce(self, world_size: int, group=None) -> int:
        """SUM over the data-parallel ranks, in place on the gradient buffer; the average is taken by the optimizer (`grad_scale`).
        Returns the bytes reduced."""
        if world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return self.grad.numel() * self.grad.element_size()


class _FlatOptimizer:
    def __init__(self, params, state_dtype: torch.dtype):
        self.bucket = params if isinstance(params, FlatLoraBucket) else FlatLoraBucket(params)
        if not self.bucket.param.is_cuda:
            raise RFError("the optimizer kernels run on the GPU only (the product path has no CPU fallback); the bucket is on "
                          f"{self.bucket.param.device}")
        if state_dtype not in (BF, torch.float32):
            raise RFError("optimizer state must be bf16 (what torch / prodigyopt keep for bf16 parameters) or fp32")
        self.state_fp32 = state_dtype == torch.float32
        n, dev = self.bucket.numel, self.bucket.param.device
        self.exp_avg = torch.zeros(n, dtype=state_dtype, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=state_dtype, device=dev)
        self.grad_scale = 1.0                                     # set to 1 / world_size behind bucket.all_reduce()
        self.lib = L.load()
        self._clip_ws = self._clip_out = None

    @property
    def param_groups(self):
        return [{"params": self.bucket.params, **self.defaults}]

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.bucket.zero_grad()

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_(params, max_norm) over the flat gradient bucket -- what Lightning's `gradient_clip_val` runs
        between the DDP all-reduce and optimizer.step (the reference trains with 0.5, train_flux/train/train.py:165): the L2 norm of
        `grad_scale * grad` (the AVERAGED gradient when the bucket holds a SUM), coef = min(1, max_norm / (norm + 1e-6)), gradients
        scaled in place.  Norm and coefficient stay on the device: returns the 2-float tensor {total_norm, coef} without a sync."""
        b = self.bucket
        if self._clip_ws is None:
            self._clip_ws = torch.empty(int(self.lib.rf_lora_prodigy_partials_bytes(b.numel)) // 4, dtype=torch.float32, device=b.param.device)
            self._clip_out = torch.zeros(2, dtype=torch.float32, device=b.param.device)
        L.check(self.lib.rf_lora_clip_grad_norm(b.grad.data_ptr(), b.numel, float(max_norm), float(self.grad_scale), self._clip_ws.data_ptr(),
                                                self._clip_ws.numel() * 4, self._clip_out.data_ptr(), stream_ptr()), "rf_lora_clip_grad_norm")
        return self._clip_out

    _STATE_TENSORS: tuple = ()

    def load_state_dict(self, sd: Dict) -> None:
        """Resume from `state_dict()` (Lightning's checkpoint of the optimizer): state tensors are copied INTO the flat buffers (the
        kernels hold their addresses), hyper-parameters replace the defaults."""
        for k in self._STATE_TENSORS:
            t, dst = sd[k], getattr(self, k)
            if tuple(t.shape) != tuple(dst.shape):
                raise RFError(f"load_state_dict: {k} has shape {tuple(t.shape)}, the bucket needs {tuple(dst.shape)}")
            with torch.no_grad():
                dst.copy_(t.to(dst.device))
        for k in self.defaults:
            if k in sd:
                self.defaults[k] = tuple(sd[k]) if k == "betas" else sd[k]


class LoraAdamW(_FlatOptimizer):
    """torch.optim.AdamW's update (adam.py::_single_tensor_adam, decoupled decay, amsgrad off) as ONE launch of rf_lora_adamw."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 state_dtype: torch.dtype = BF):
        super().__init__(params, state_dtype)
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.step_count = 0

    @torch.no_grad()
    def step(self) -> None:
        d, b = self.defaults, self.bucket
        self.step_count += 1
        L.check(self.lib.rf_lora_adamw(b.param.data_ptr(), b.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), b.numel,
                                       int(self.state_fp32), self.step_count, d["lr"], d["betas"][0], d["betas"][1], d["eps"], d["weight_decay"],
                                       self.grad_scale, stream_ptr()), "rf_lora_adamw")

    _STATE_TENSORS = ("exp_avg", "exp_avg_sq")

    def state_dict(self) -> Dict:
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, **self.defaults)

    def load_state_dict(self, sd: Dict) -> None:
        super().load_state_dict(sd)
        self.step_count = int(sd["step"])


class LoraProdigy(_FlatOptimizer):
    """prodigyopt.Prodigy's step (Mishchenko & Defazio 2023, Adam form; PARITY UNPINNED -- the package is not available offline, see
    oracle/optim_oracle.py) as three launches of rf_lora_prodigy; defaults are the package's, the reference overrides lr / use_bias_correction /
    safeguard_warmup / weight_decay (config.yaml:55-61)."""

    def __init__(self, params, lr: float = 1.0, betas=(0.9, 0.999), beta3: Optional[float] = None, eps: float = 1e-8, weight_decay: float = 0.0,
                 decouple: bool = True, use_bias_correction: bool = False, safeguard_warmup: bool = False, d0: float = 1e-6, d_coef: float = 1.0,
                 growth_rate: float = float("inf"), state_dtype: torch.dtype = BF):
        super().__init__(params, state_dtype)
        self.defaults = dict(lr=lr, betas=tuple(betas), beta3=beta3, eps=eps, weight_decay=weight_decay, decouple=decouple,
                             use_bias_correction=use_bias_correction, safeguard_warmup=safeguard_warmup, d0=d0, d_coef=d_coef, growth_rate=growth_rate)
        b = self.bucket
        self.s = torch.zeros_like(self.exp_avg)
        self.p0 = b.param.clone()
        # {d, d_max, d_numerator, d_denom, d_hat, k, dlr of the last step, d0} -- lives on the device
        self.dstate = torch.tensor([d0, d0, 0.0, 0.0, d0, 0.0, 0.0, d0], dtype=torch.float64, device=b.param.device)
        self.partials = torch.empty(int(self.lib.rf_lora_prodigy_partials_bytes(b.numel)) // 4, dtype=torch.float32, device=b.param.device)

    @torch.no_grad()
    def step(self) -> None:
        d, b = self.defaults, self.bucket
        beta3 = d["beta3"] if d["beta3"] is not None else math.sqrt(d["betas"][1])
        gr = d["growth_rate"] if math.isfinite(d["growth_rate"]) else 3.0e38
        L.check(self.lib.rf_lora_prodigy(b.param.data_ptr(), b.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.s.data_ptr(),
                                         self.p0.data_ptr(), b.numel, int(self.state_fp32), self.dstate.data_ptr(), d["lr"], d["betas"][0],
                                         d["betas"][1], beta3, d["eps"], d["weight_decay"], int(d["decouple"]), int(d["use_bias_correction"]),
                                         int(d["safeguard_warmup"]), d["d_coef"], gr, self.grad_scale, self.partials.data_ptr(),
                                         self.partials.numel() * 4, stream_ptr()), "rf_lora_prodigy")

    def d_state(self) -> Dict[str, float]:
        """the distance estimate, read back (a host sync: for logging / tests, never inside the step)"""
        v = self.dstate.tolist()
        return dict(d=v[0], d_max=v[1], d_numerator=v[2], d_denom=v[3], d_hat=v[4], k=int(v[5]), dlr=v[6], d0=v[7])

    _STATE_TENSORS = ("exp_avg", "exp_avg_sq", "s", "p0", "dstate")       # d, d_max, the numerator and k travel in dstate

    def state_dict(self) -> Dict:
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, s=self.s, p0=self.p0, dstate=self.dstate, **self.defaults)


def build_optimizer(params, optimizer_config: Dict, state_dtype: torch.dtype = BF):
    """train/model.py:105-117: `optimizer_config = {"type": "AdamW" | "Prodigy" | "SGD", "params": {...}}`."""
    kind, kw = optimizer_config["type"], dict(optimizer_config.get("params", {}))
    if kind == "AdamW":
        return LoraAdamW(params, state_dtype=state_dtype, **kw)
    if kind == "Prodigy":
        return LoraProdigy(params, state_dtype=state_dtype, **kw)
    if kind == "SGD":
        raise RFError("optimizer type SGD: use torch.optim.SGD on lora_parameters() (no HIP kernel; the shipped config trains with Prodigy)")
    raise NotImplementedError(kind)                                # as the reference (model.py:116-117)
