"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg may import this; the product path never
does).  CPU restatement of the two optimizers the reference can build over its LoRA parameters
(/root/reference/train_flux/train/model.py:105-117; config.yaml:55-61 ships Prodigy with lr 1, use_bias_correction,
safeguard_warmup, weight_decay 0.01):

  adamw_step    torch.optim.AdamW, single-tensor form (torch/optim/adam.py::_single_tensor_adam with decoupled weight decay) --
                PINNED: tests/test_train_cpu.py checks it bit for bit against torch.optim.AdamW itself (fp32, CPU).
  prodigy_step  prodigyopt.Prodigy.  The package is a third-party dependency that is NOT vendored in /root/reference
                (requirements.txt: `prodigyopt`, un-pinned; 1.0 at the time of the reference's commits) and not installed in this
                image: **PARITY UNPINNED**.  Restated from the published algorithm -- K. Mishchenko, A. Defazio, "Prodigy: An
                Expeditiously Adaptive Parameter-Free Learner" (2023), Algorithm 4 (Adam form) -- with the package's option names
                and defaults (lr 1.0, betas (0.9, 0.999), beta3 = sqrt(beta2), eps 1e-8, weight_decay 0, decouple True,
                use_bias_correction False, safeguard_warmup False, d0 1e-6, d_coef 1.0, growth_rate inf) and its order of operations:
                    dlr   = d lr [ sqrt(1 - b2^(k+1)) / (1 - b1^(k+1)) ]
                    num   = b3 num + (d / d0) dlr <g, x0 - x>
                    m     = b1 m + d (1 - b1) g ;   v = b2 v + d^2 (1 - b2) g^2
                    s     = b3 s + (d / d0) (d if safeguard_warmup else dlr) g ;   den = ||s||_1
                    d_hat = d_coef num / den ;  d = max(d, d_hat) while d == d0 ;  d_max = max(d_max, d_hat) ;  d = min(d_max, d growth)
                    x     = x (1 - wd dlr) - dlr m / (sqrt(v) + d eps)              (dlr of the OLD d, eps scaled by the NEW d)
                tests check what can be checked without the package: the invariants of the algorithm (d non-decreasing, d_hat from the
                two sums, the skipped all-zero step, weight decay decoupled) and convergence on a quadratic from d0 = 1e-6.

All arithmetic here is fp32 / python floats on flat tensors; the HIP kernels (csrc/optim.hip) are compared with it on the GPU."""
from __future__ import annotations

import math
from typing import Dict

import torch


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float = 1e-3, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
    """in place on (p, m, v); `step` is 1-based (torch increments before use)"""
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def prodigy_init(p: torch.Tensor, d0: float = 1e-6) -> Dict:
    return dict(m=torch.zeros_like(p), v=torch.zeros_like(p), s=torch.zeros_like(p), p0=p.detach().clone(), d=d0, d0=d0, d_max=d0,
                d_numerator=0.0, d_denom=0.0, d_hat=d0, k=0)


def prodigy_step(p: torch.Tensor, g: torch.Tensor, st: Dict, lr: float = 1.0, beta1: float = 0.9, beta2: float = 0.999, beta3=None,
                 eps: float = 1e-8, weight_decay: float = 0.0, decouple: bool = True, use_bias_correction: bool = False,
                 safeguard_warmup: bool = False, d_coef: float = 1.0, growth_rate: float = float("inf")) -> None:
    """one Prodigy step over ONE flat parameter tensor, in place on p and st"""
    if beta3 is None:
        beta3 = math.sqrt(beta2)
    d, d0, k = st["d"], st["d0"], st["k"]
    bc = math.sqrt(1 - beta2 ** (k + 1)) / (1 - beta1 ** (k + 1)) if use_bias_correction else 1.0
    dlr = d * lr * bc
    if weight_decay != 0 and not decouple:
        g = g + weight_decay * p
    d_numerator = st["d_numerator"] * beta3
    if lr > 0.0:
        d_numerator += (d / d0) * dlr * float(torch.dot(g.flatten().double(), (st["p0"] - p).flatten().double()))
        st["m"].mul_(beta1).add_(g, alpha=d * (1 - beta1))
        st["v"].mul_(beta2).addcmul_(g, g, value=d * d * (1 - beta2))
        st["s"].mul_(beta3).add_(g, alpha=(d / d0) * (d if safeguard_warmup else dlr))
        d_denom = float(st["s"].abs().double().sum())                # the package accumulates d_denom INSIDE the lr > 0 gate ...
    else:
        d_denom = 0.0
    if d_denom == 0:
        return                                                       # ... so lr == 0, or no gradient seen yet: nothing moves, k stays
    d_hat = d
    d_max = st["d_max"]
    if lr > 0.0:
        d_hat = d_coef * d_numerator / d_denom
        if d == d0:
            d = max(d, d_hat)
        d_max = max(d_max, d_hat)
        d = min(d_max, d * growth_rate)
    st.update(d=d, d_max=d_max, d_numerator=d_numerator, d_denom=d_denom, d_hat=d_hat, k=k + 1)
    denom = st["v"].sqrt().add_(d * eps)
    if weight_decay != 0 and decouple:
        p.add_(p, alpha=-weight_decay * dlr)
    p.addcdiv_(st["m"], denom, value=-dlr)
