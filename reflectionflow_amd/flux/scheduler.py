"""FlowMatchEulerDiscreteScheduler with FLUX.1-dev's config (diffusers is
not installable here).  `step` runs the HIP Euler kernel (rf_euler_step)."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import ops


class _Config(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5,
                    max_shift: float = 1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self):
        self.config = _Config(num_train_timesteps=1000, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
                              base_image_seq_len=256, max_image_seq_len=4096)
        self.timesteps = None
        self.sigmas = None          # fp32 host copy: the loop needs (sigma_{i+1} - sigma_i) as a kernel scalar
        self._step_index = None

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        if sigmas is None:
            sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
        sigmas = np.array(sigmas).astype(np.float32)
        if self.config.use_dynamic_shifting:
            sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1) ** 1.0)
        sigmas = torch.from_numpy(np.asarray(sigmas, dtype=np.float32))
        self.timesteps = (sigmas * self.config.num_train_timesteps).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self._step_index = None

    def dts(self):
        """sigma_{i+1} - sigma_i for every step, as python floats (fp32 arithmetic like the reference)."""
        return [float(x) for x in (self.sigmas[1:] - self.sigmas[:-1])]

    def step(self, model_output, timestep, sample, return_dict=False):
        if self._step_index is None:
            self._step_index = 0
        dt = float(self.sigmas[self._step_index + 1] - self.sigmas[self._step_index])
        out = sample.to(model_output.dtype).contiguous().clone()
        ops.euler_step_(out, model_output.contiguous(), dt)
        self._step_index += 1
        return (out,)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kw):
    if timesteps is not None:
        raise ValueError("custom `timesteps` are not supported by FlowMatchEulerDiscreteScheduler; pass sigmas")
    scheduler.set_timesteps(num_inference_steps, device=device, sigmas=sigmas, **kw)
    return scheduler.timesteps, len(scheduler.timesteps)
