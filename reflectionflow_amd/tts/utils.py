"""Noise protocol and small helpers with the reference's names (tts/utils.py:15,71-87,131-155).

Differences that are deliberate and documented: `device` is an argument
instead of the hard-coded "cuda", and `get_noises` can take explicit `seeds` so that results do not
depend on how candidates are sharded across ranks (the reference draws them with torch.randint from
the global RNG)."""
from __future__ import annotations

import hashlib
import re
from typing import Dict, Optional, Sequence

import torch

from ..flux.pipeline import FluxPipeline

TORCH_DTYPE_MAP = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
MODEL_NAME_MAP = {"black-forest-labs/FLUX.1-dev": "flux.1-dev"}


def prepare_latents_for_flux(batch_size: int, height: int, width: int, generator: torch.Generator, device: str,
                             dtype: torch.dtype) -> torch.Tensor:
    num_latent_channels, vae_scale_factor = 16, 8
    height = 2 * (int(height) // (vae_scale_factor * 2))
    width = 2 * (int(width) // (vae_scale_factor * 2))
    shape = (batch_size, num_latent_channels, height, width)
    # diffusers.randn_tensor with a CPU generator draws on the CPU in `dtype`, then moves
    latents = torch.randn(shape, generator=generator, dtype=dtype).to(device)
    return FluxPipeline._pack_latents(latents, batch_size, num_latent_channels, height, width)


def get_latent_prep_fn(pretrained_model_name_or_path: str):
    return {"black-forest-labs/FLUX.1-dev": prepare_latents_for_flux}[pretrained_model_name_or_path]


def get_noises(max_seed: int, num_samples: int, height: int, width: int, device="cuda",
               dtype: torch.dtype = torch.bfloat16, fn=prepare_latents_for_flux,
               seeds: Optional[Sequence[int]] = None) -> Dict[int, torch.Tensor]:
    if seeds is None:
        seeds = torch.randint(0, high=max_seed, size=(num_samples,)).tolist()
    noises = {}
    for noise_seed in seeds:
        noises[int(noise_seed)] = fn(batch_size=1, height=height, width=width,
                                     generator=torch.manual_seed(int(noise_seed)), device=device, dtype=dtype)
    assert len(noises) == len(seeds)
    return noises


def prompt_to_filename(prompt, max_length=100):
    filename = re.sub(r"_+", "_", re.sub(r"[^a-zA-Z0-9]", "_", prompt.strip()))
    hash_digest = hashlib.sha256(prompt.encode()).hexdigest()[:8]
    base_filename = f"prompt@{filename}_hash@{hash_digest}"
    if len(base_filename) > max_length:
        base_length = max_length - len(hash_digest) - 7
        base_filename = f"prompt@{filename[:base_length]}_hash@{hash_digest}"
    return base_filename
