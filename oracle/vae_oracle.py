"""TEST INFRASTRUCTURE -- CPU restatement of the FLUX VAE (diffusers `AutoencoderKL`, 16 latent channels) as plain
functions over a diffusers-layout state dict.  Only tests/ may import this.

**Parity unpinned**: diffusers is not vendored by the reference (requirements.txt:1, no version pin) and cannot be
installed here; there is no source, test or vector for the VAE under /root/reference.  The call sites that fix WHAT is
computed are train_flux/flux/generate.py:302-307 (decode), train_flux/flux/pipeline_tools.py:7-14 (encode +
shift/scale + pack) and tts/tts_reflectionflow.py:273-279 (resize of the decoded candidate).  The arithmetic below
restates diffusers' published modules (AutoencoderKL / Encoder / Decoder / ResnetBlock2D / Downsample2D /
Upsample2D / Attention with one head, GroupNorm eps 1e-6, SiLU) independently of reflectionflow_amd/flux/vae.py's
module tree -- it walks the state dict by key -- so a wiring mistake in either shows up as a mismatch.
"""
import math

import torch
import torch.nn.functional as F


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _resnet(sd, p, x, groups):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups)))
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _attn(sd, p, x, groups):
    B, C, H, W = x.shape
    h = _gn(sd, p + ".group_norm", x, groups).reshape(B, C, H * W).transpose(1, 2)
    lin = lambda n, t: t @ sd[f"{p}.{n}.weight"].t() + sd[f"{p}.{n}.bias"]  # noqa: E731
    q, k, v = lin("to_q", h), lin("to_k", h), lin("to_v", h)
    w = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
    o = lin("to_out.0", w @ v)
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def _mid(sd, p, x, groups):
    x = _resnet(sd, p + ".resnets.0", x, groups)
    if p + ".attentions.0.to_q.weight" in sd:
        x = _attn(sd, p + ".attentions.0", x, groups)
    return _resnet(sd, p + ".resnets.1", x, groups)


def _count(sd, prefix):
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def vae_encode_moments(sd, x, groups=32):
    """[B,3,H,W] in [-1,1] -> [B, 2*latent, H/8, W/8] (mean | logvar)."""
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(_count(sd, "encoder.down_blocks")):
        p = f"encoder.down_blocks.{i}"
        for j in range(_count(sd, p + ".resnets")):
            h = _resnet(sd, f"{p}.resnets.{j}", h, groups)
        if f"{p}.downsamplers.0.conv.weight" in sd:
            h = _conv(sd, f"{p}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _mid(sd, "encoder.mid_block", h, groups)
    return _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h, groups)))


def vae_decode(sd, z, groups=32):
    """[B, latent, h, w] -> [B,3,8h,8w]."""
    h = _mid(sd, "decoder.mid_block", _conv(sd, "decoder.conv_in", z), groups)
    for i in range(_count(sd, "decoder.up_blocks")):
        p = f"decoder.up_blocks.{i}"
        for j in range(_count(sd, p + ".resnets")):
            h = _resnet(sd, f"{p}.resnets.{j}", h, groups)
        if f"{p}.upsamplers.0.conv.weight" in sd:
            h = _conv(sd, f"{p}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.conv_norm_out", h, groups)))


def sample_latent(moments, noise):
    """DiagonalGaussianDistribution.sample with an explicit noise tensor: mean + exp(0.5*clamp(logvar,-30,20))*noise."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise
