"""FLUX VAE (`AutoencoderKL`, 16 latent channels, 8x) + `VaeImageProcessor` as PyTorch(-ROCm) modules.

The north star leaves VAE decode in PyTorch-ROCm (MIOpen convolutions): this file is therefore plain `torch.nn` --
it is NOT part of the HIP hot path and not behind the C ABI.  It exists so that the calls either side of the denoise
loop work offline exactly as the reference makes them:
  * `generate(..., output_type="pil")`  -> `vae.decode(latents / scaling_factor + shift_factor)` + `postprocess`
    (reference train_flux/flux/generate.py:302-307);
  * `Condition(condition=PIL).encode(pipe)` -> `image_processor.preprocess` + `vae.encode(x).latent_dist.sample()`
    (train_flux/flux/pipeline_tools.py:7-14, condition.py:96-132);
  * the reflection rounds' hand-off decode -> resize(condition_size) -> encode (tts/tts_reflectionflow.py:273-279).

diffusers is not vendored by the reference and cannot be installed here, so the module tree restates diffusers'
published `AutoencoderKL` for the FLUX.1-dev VAE config (state-dict key names = the attribute paths below, so a
diffusers `vae/diffusion_pytorch_model.safetensors` loads unchanged).  **Parity unpinned** at this boundary: there is
no source, test or vector for it under /root/reference; tests check it against an independent functional
restatement in oracle/vae_oracle.py driven by the same state dict.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

FLUX_VAE_CONFIG = dict(
    in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
    norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159, mid_block_add_attention=True,
)


class _Config(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class VaeAttention(nn.Module):
    """diffusers `Attention` as the VAE mid block uses it: GroupNorm, 1 head of `channels`, residual."""

    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)            # [B, HW, C]
        q, k, v = self.to_q(h)[:, None], self.to_k(h)[:, None], self.to_v(h)[:, None]   # 1 head
        o = F.scaled_dot_product_attention(q, k, v)[:, 0]
        o = self.to_out[0](o).transpose(1, 2).reshape(B, C, H, W)
        return x + o


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))            # diffusers Downsample2D(padding=0): pad right/bottom


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Down(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Up(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _Mid(nn.Module):
    def __init__(self, c, groups, attn):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups)]) if attn else None

    def forward(self, x):
        x = self.resnets[0](x)
        if self.attentions is not None:
            x = self.attentions[0](x)
        return self.resnets[1](x)


class Encoder(nn.Module):
    def __init__(self, cin, latent, chans, layers, groups, attn):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = chans[0]
        for i, co in enumerate(chans):
            self.down_blocks.append(_EncBlock(c, co, layers, groups, down=i < len(chans) - 1))
            c = co
        self.mid_block = _Mid(c, groups, attn)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, latent, cout, chans, layers, groups, attn):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups, attn)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_DecBlock(c, co, layers + 1, groups, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, cout, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters: torch.Tensor):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL(nn.Module):
    """`AutoencoderKL()` is the FLUX.1-dev VAE shape (83.8 M parameters)."""

    def __init__(self, **cfg):
        super().__init__()
        c = dict(FLUX_VAE_CONFIG)
        c.update(cfg)
        c["block_out_channels"] = tuple(c["block_out_channels"])
        self.config = _Config(c)
        self.encoder = Encoder(c["in_channels"], c["latent_channels"], c["block_out_channels"], c["layers_per_block"],
                               c["norm_num_groups"], c["mid_block_add_attention"])
        self.decoder = Decoder(c["latent_channels"], c["out_channels"], c["block_out_channels"], c["layers_per_block"],
                               c["norm_num_groups"], c["mid_block_add_attention"])

    @property
    def dtype(self):
        return self.encoder.conv_in.weight.dtype

    @property
    def device(self):
        return self.encoder.conv_in.weight.device

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        dist = DiagonalGaussianDistribution(self.encoder(x))
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        y = self.decoder(z)
        return DecoderOutput(y) if return_dict else (y,)


def init_synthetic_vae_(vae: nn.Module, seed: int = 0):
    """Random-init weights for offline runs under a fixed seed: matrices / conv kernels ~ N(0, 1 / fan_in), norm scales
    1 + N(0, 0.02^2), biases N(0, 0.02^2).  (Synthetic 'images' are whatever such a decoder emits; postprocess clamps them
    to [0, 1].)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in vae.named_parameters():
            if p.ndim == 1:
                if name.endswith("weight"):
                    p.copy_(1.0 + 0.02 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
    return vae


class VaeImageProcessor:
    """diffusers `VaeImageProcessor(vae_scale_factor=16)` as the FLUX pipeline uses it: PIL / numpy / tensor ->
    [-1, 1] NCHW float tensor with sides rounded down to a multiple of `vae_scale_factor`; and back."""

    def __init__(self, vae_scale_factor: int = 16):
        self.vae_scale_factor = vae_scale_factor

    def _size(self, w: int, h: int) -> Tuple[int, int]:
        f = self.vae_scale_factor
        return w - w % f, h - h % f

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        if isinstance(image, torch.Tensor):
            x = image if image.ndim == 4 else image[None]
            if x.min() >= 0:                                  # [0,1] tensors are normalised; [-1,1] pass through
                x = 2.0 * x - 1.0
            return x
        from PIL import Image
        imgs = image if isinstance(image, (list, tuple)) else [image]
        out = []
        for im in imgs:
            if isinstance(im, np.ndarray):
                arr = im.astype(np.float32)
                if arr.max() > 1.0:
                    arr = arr / 255.0
            else:
                w, h = self._size(width or im.width, height or im.height)
                if (w, h) != im.size:
                    im = im.resize((w, h), resample=Image.LANCZOS)
                arr = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
            out.append(torch.from_numpy(arr).permute(2, 0, 1))
        return 2.0 * torch.stack(out) - 1.0

    def postprocess(self, image: torch.Tensor, output_type: str = "pil"):
        if output_type == "latent":
            return image
        x = (image.float() / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return x
        arr = x.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            return arr
        if output_type != "pil":
            raise ValueError(f"unknown output_type {output_type!r}")
        from PIL import Image
        return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
