"""FLUX VAE decode / encode on the HIP path (librf_flux.so: rf_vae_decode / rf_vae_encode, csrc/vae.hip).

`HipVAE(vae)` wraps an `AutoencoderKL` module tree (flux/vae.py: diffusers key names, so a diffusers
`vae/diffusion_pytorch_model.safetensors` loads unchanged) and exposes the same `.encode()` / `.decode()` / `.config`
surface the reference's call sites use:
    generate.py:302-307            vae.decode(latents / scaling_factor + shift_factor, return_dict=False)[0]
    pipeline_tools.py:7-14         vae.encode(images).latent_dist.sample()
    tts_reflectionflow.py:273-279  decode -> resize -> encode hand-off between reflection rounds
so `pipe.vae = HipVAE(pipe.vae)` (FluxPipeline.enable_hip_vae()) is the whole switch.  No CPU fallback: CPU tensors or a
missing library raise.  What stays in PyTorch is plumbing only: NCHW <-> zero-halo NHWC packing of the (tiny) latent and of
the RGB image, the reparameterisation `mean + exp(0.5 logvar) * noise`, and the weight repacking done once here.

Weight repacking (once, at construction):
    conv 3x3  [cout, cin, 3, 3] -> [cout_pad][3][3][cin_pad] bf16   (taps of one dy are 3*cin contiguous values: the conv is a
                                                                    3-K-segment GEMM over a zero-halo NHWC image, vae.hip header)
    conv 1x1  [cout, cin, 1, 1] -> [cout][cin]
    attention to_q | to_k -> [2C][C];  to_v stays [C][C] and is used as the A operand (V^T = W_v . x^T);  to_v.bias is folded
              into the out-projection bias: softmax rows sum to 1, so P (V + 1 b_v^T) W_o^T = P V W_o^T + (W_o b_v)^T.
Channel padding: latent 16 -> 64 and RGB 3 -> 64 on the input side (zero weights for the padding), RGB 3 -> 8 on the output side.
Narrow layers (cout = 128 at the 1024^2 level, the 8-channel conv_out) additionally get a FOLDED weight copy: g = 256 / cout
adjacent output pixels become one GEMM row (N = 256: a full MFMA tile instead of a half- or 31/32-empty one).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib as L
from ..ops import RFError, stream_ptr
from .vae import AutoencoderKLOutput, DecoderOutput, DiagonalGaussianDistribution

BF = torch.bfloat16


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _Packer:
    """Collects the repacked device tensors (kept alive on `self.keep`) and fills the ctypes weight structs."""

    def __init__(self, device):
        self.device = device
        self.keep: List[torch.Tensor] = []

    def t(self, x: torch.Tensor) -> int:
        x = x.detach().to(self.device, BF).contiguous()
        self.keep.append(x)
        return x.data_ptr()

    def conv3(self, out: "L.rf_vae_conv", m, cin_pad: Optional[int] = None, cout_pad: Optional[int] = None):
        w = m.weight.detach().float()                                   # [cout, cin, 3, 3]
        co, ci = w.shape[0], w.shape[1]
        cip, cop = cin_pad or _pad_to(ci, 64), cout_pad or _pad_to(co, 8)
        wp = torch.zeros(cop, 3, 3, cip)
        wp[:co, :, :, :ci] = w.permute(0, 2, 3, 1).cpu()
        bp = torch.zeros(cop)
        bp[:co] = m.bias.detach().float().cpu()
        out.w, out.b, out.cin, out.cout = self.t(wp.reshape(cop, 9 * cip)), self.t(bp), cip, cop
        # narrow outputs: fold g = 256 / cout adjacent output pixels into one GEMM row (rf_vae_conv.wf): a full 256-column tile
        if cop < 256 and 256 % cop == 0:
            g = 256 // cop
            wf = torch.zeros(g, cop, 3, g + 2, cip)
            for j in range(g):
                wf[j, :, :, j:j + 3, :] = wp
            out.wf, out.bf, out.fold = self.t(wf.reshape(g * cop, 3 * (g + 2) * cip)), self.t(bp.repeat(g)), g

    def conv1(self, out: "L.rf_vae_conv", m):
        w = m.weight.detach()
        co, ci = w.shape[0], w.shape[1]
        if ci % 64 or co % 8:
            raise RFError(f"HipVAE: 1x1 convolution {ci} -> {co} needs cin % 64 == 0 and cout % 8 == 0")
        out.w, out.b, out.cin, out.cout = self.t(w.reshape(co, ci)), self.t(m.bias), ci, co

    def norm(self, out: "L.rf_vae_norm", m):
        out.gamma, out.beta = self.t(m.weight), self.t(m.bias)

    def resnet(self, out: "L.rf_vae_resnet", m):
        self.norm(out.norm1, m.norm1)
        self.conv3(out.conv1, m.conv1)
        self.norm(out.norm2, m.norm2)
        self.conv3(out.conv2, m.conv2)
        if m.conv_shortcut is not None:
            self.conv1(out.shortcut, m.conv_shortcut)

    def attn(self, out: "L.rf_vae_attn", m):
        Cc = m.to_q.weight.shape[0]
        self.norm(out.norm, m.group_norm)
        out.w_qk = self.t(torch.cat([m.to_q.weight, m.to_k.weight], 0))
        out.b_qk = self.t(torch.cat([m.to_q.bias, m.to_k.bias], 0))
        out.w_v = self.t(m.to_v.weight)
        wo, bo = m.to_out[0].weight.detach().float(), m.to_out[0].bias.detach().float()
        out.w_out = self.t(wo)
        out.b_out = self.t(bo + wo @ m.to_v.bias.detach().float())
        out.C = Cc


def _pack_direction(vae, encode: bool, device) -> Tuple["L.rf_vae_weights", _Packer]:
    p = _Packer(device)
    w = L.rf_vae_weights()
    cfg = vae.config
    net = vae.encoder if encode else vae.decoder
    blocks = net.down_blocks if encode else net.up_blocks
    if len(blocks) > 4 or any(len(b.resnets) > 3 for b in blocks):
        raise RFError("HipVAE: at most 4 levels of at most 3 resnets")
    w.levels, w.res_per_level = len(blocks), len(blocks[0].resnets)
    w.groups = cfg["norm_num_groups"]
    w.has_attn = 1 if net.mid_block.attentions is not None else 0
    p.conv3(w.conv_in, net.conv_in, cin_pad=64)
    p.resnet(w.mid0, net.mid_block.resnets[0])
    p.resnet(w.mid1, net.mid_block.resnets[1])
    if w.has_attn:
        p.attn(w.attn, net.mid_block.attentions[0])
    for i, b in enumerate(blocks):
        if len(b.resnets) != w.res_per_level:
            raise RFError("HipVAE: every level must have the same number of resnets")
        for j, r in enumerate(b.resnets):
            p.resnet(w.res[i][j], r)
        samp = b.downsamplers if encode else b.upsamplers
        if samp is not None:
            p.conv3(w.resample[i], samp[0].conv)
    p.norm(w.norm_out, net.conv_norm_out)
    p.conv3(w.conv_out, net.conv_out)
    return w, p


def _padded_input(h: int, w: int, c: int, device) -> torch.Tensor:
    """The zero-halo NHWC image [(h + 2), (w + 2), c] a convolution reads, as a view of a ZEROED allocation with 64 pixels of slack
    behind it: conv_in runs as a GEMM whose K-tiles (64 elements) reach past the last pixel's 3 * c channels, against zero weight
    columns -- finite garbage there multiplies away, a NaN / Inf left in a recycled block does not (it did: one image of a batch
    came out all-NaN through the next GroupNorm when the allocator placed the input in front of such a block)."""
    n = (h + 2) * (w + 2) * c
    flat = torch.zeros(n + 64 * max(c, 64), dtype=BF, device=device)
    return flat[:n].view(h + 2, w + 2, c)


class HipVAE:
    """`AutoencoderKL`-shaped object whose encode / decode run on librf_flux.so.  `module` keeps the original torch modules
    (state_dict, dtype bookkeeping); nothing of it executes in encode() / decode()."""

    def __init__(self, module):
        dev = module.device
        if dev.type != "cuda":
            raise RFError(f"HipVAE: the VAE is on {dev}; the HIP path has no CPU fallback")
        if module.dtype != BF:
            raise RFError(f"HipVAE: needs a bf16 module, got {module.dtype}")
        L.load()
        self.module = module
        self.config = module.config
        self._dec, self._dec_keep = _pack_direction(module, False, dev)
        self._enc, self._enc_keep = _pack_direction(module, True, dev)
        self._ws: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self.latent_pad = self._dec.conv_in.cin
        self.image_pad = self._enc.conv_in.cin
        self.scale = 2 ** (self._dec.levels - 1)

    dtype = property(lambda self: BF)
    device = property(lambda self: self.module.device)

    def check_geometry(self, height: int, width: int, what: str = "image") -> None:
        """Raise NOW (before a search has spent minutes denoising) if an image of height x width pixels cannot go through this VAE:
        sides must be multiples of the down-scaling factor, and the mid-block attention kernel needs (h / scale)(w / scale) % 64 == 0
        latent pixels.  The torch AutoencoderKL it replaces takes any size; callers with odd sizes keep the torch module
        (do not call pipe.enable_hip_vae())."""
        s = self.scale
        if height % s or width % s:
            raise RFError(f"HipVAE: {what} {height}x{width}: sides must be multiples of {s}")
        if (self._dec.has_attn or self._enc.has_attn) and ((height // s) * (width // s)) % 64:
            raise RFError(f"HipVAE: {what} {height}x{width} has {(height // s) * (width // s)} latent pixels; the mid-block attention "
                          "kernel needs a multiple of 64 (e.g. sides that are multiples of 64 pixels)")

    def to(self, *a, **k):                                  # pipeline.to(device) walks its parts
        return self

    def _workspace(self, w, encode: int, h: int, wd: int) -> "L.rf_workspace":
        key = (encode, h, wd)
        if key not in self._ws:
            n = L.load().rf_vae_workspace_bytes(C.byref(w), encode, h, wd)
            if n <= 0:
                raise RFError(f"rf_vae_workspace_bytes failed ({n})")
            if len(self._ws) >= 4:                           # bounded cache (a 1024^2 decode workspace is ~1.2 GB)
                self._ws.pop(next(iter(self._ws)))
            self._ws[key] = torch.zeros(n, dtype=torch.uint8, device=self.device)
        t = self._ws[key]
        ws = L.rf_workspace()
        ws.base, ws.bytes = t.data_ptr(), t.numel()
        return ws

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        if not z.is_cuda:
            raise RFError("HipVAE.decode: tensor is on the CPU; the HIP path has no CPU fallback")
        B, Cz, h, wd = z.shape
        lib, w = L.load(), self._dec
        ws = self._workspace(w, 0, h, wd)
        H, W = self.scale * h, self.scale * wd
        outs = []
        for b in range(B):
            zp = _padded_input(h, wd, w.conv_in.cin, z.device)
            zp[1:-1, 1:-1, :Cz] = z[b].to(BF).permute(1, 2, 0)
            out = torch.empty(H + 2, W + 2, w.conv_out.cout, dtype=BF, device=z.device)
            L.check(lib.rf_vae_decode(C.byref(w), zp.data_ptr(), h, wd, out.data_ptr(), C.byref(ws), stream_ptr()), "rf_vae_decode")
            outs.append(out[1:-1, 1:-1, : self.config["out_channels"]].permute(2, 0, 1))
        y = torch.stack(outs).contiguous()
        return DecoderOutput(y) if return_dict else (y,)

    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """[B, 3, H, W] in [-1, 1] -> [B, 2 * latent, H/8, W/8] (mean | logvar), bf16."""
        if not x.is_cuda:
            raise RFError("HipVAE.encode: tensor is on the CPU; the HIP path has no CPU fallback")
        B, Cx, H, W = x.shape
        if H % self.scale or W % self.scale:
            raise RFError(f"HipVAE.encode: image sides must be multiples of {self.scale}, got {H}x{W}")
        lib, w = L.load(), self._enc
        ws = self._workspace(w, 1, H, W)
        h, wd = H // self.scale, W // self.scale
        outs = []
        for b in range(B):
            xp = _padded_input(H, W, w.conv_in.cin, x.device)
            xp[1:-1, 1:-1, :Cx] = x[b].to(BF).permute(1, 2, 0)
            out = torch.empty(h + 2, wd + 2, w.conv_out.cout, dtype=BF, device=x.device)
            L.check(lib.rf_vae_encode(C.byref(w), xp.data_ptr(), H, W, out.data_ptr(), C.byref(ws), stream_ptr()), "rf_vae_encode")
            outs.append(out[1:-1, 1:-1, : 2 * self.config["latent_channels"]].permute(2, 0, 1))
        return torch.stack(outs).contiguous()

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        dist = DiagonalGaussianDistribution(self.encode_moments(x))
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)
