"""FLUX transformer module tree with diffusers-compatible attribute names and state-dict keys
(SURVEY.md Appendix A), so FLUX.1-dev safetensors and the FLUX-Corrector LoRA load unchanged.

`diffusers` / `peft` are neither vendored by the reference nor installable here, so this tree is
the build's own.  Modules are PARAMETER CONTAINERS: the hot path never calls their `forward`
one by one -- `block_forward` / `single_block_forward` / `tranformer_forward` hand the raw
weight pointers to fused HIP kernels.  Leaf `forward`s that exist (HipLinear, AdaLN*, FeedForward)
also run on the HIP ops, never on a CPU/rocBLAS fallback; only the timestep/guidance/text
embedding and the RoPE tables stay in PyTorch-ROCm, as the north star prescribes.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class HipLinear(nn.Linear):
    """nn.Linear whose forward is the bf16 MFMA GEMM (rf_gemm_bf16)."""

    def forward(self, x):
        lead = x.shape[:-1]
        y = ops.linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias)
        return y.reshape(*lead, self.out_features)


class BaseTunerLayer:
    """The slice of peft.tuners.tuners_utils.BaseTunerLayer that lora_controller.py relies on."""

    scaling: Dict[str, float]
    active_adapters: List[str]

    def scale_layer(self, scale: float) -> None:
        if scale == 1:
            return
        for a in self.active_adapters:
            self.scaling[a] *= scale


class LoraLinear(nn.Module, BaseTunerLayer):
    """PEFT lora.Linear (Appendix A.9): y = base(x) + scaling * lora_B(lora_A(x)).
    State-dict keys: base_layer.{weight,bias}, lora_A.<adapter>.weight, lora_B.<adapter>.weight."""

    def __init__(self, base: nn.Linear, r: int, alpha: float, adapter: str = "default"):
        super().__init__()
        self.base_layer = base
        kw = dict(device=base.weight.device, dtype=base.weight.dtype)
        self.lora_A = nn.ModuleDict({adapter: HipLinear(base.in_features, r, bias=False, **kw)})
        self.lora_B = nn.ModuleDict({adapter: HipLinear(r, base.out_features, bias=False, **kw)})
        self.scaling = {adapter: alpha / r}
        self.active_adapters = [adapter]
        self.r = r

    in_features = property(lambda self: self.base_layer.in_features)
    out_features = property(lambda self: self.base_layer.out_features)
    weight = property(lambda self: self.base_layer.weight)
    bias = property(lambda self: self.base_layer.bias)

    def lora_factors(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(A [r_total, in], scaling*B [out, r_total]) over the active adapters."""
        As = [self.lora_A[a].weight for a in self.active_adapters]
        Bs = [self.lora_B[a].weight * self.scaling[a] for a in self.active_adapters]
        return torch.cat(As, 0), torch.cat(Bs, 1)

    def forward(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        A, B = self.lora_factors()
        r = A.shape[0]
        r_pad = (r + 63) // 64 * 64
        Ap = torch.zeros(r_pad, A.shape[1], dtype=A.dtype, device=A.device)
        Ap[:r] = A
        Bp = torch.zeros(B.shape[0], r_pad, dtype=B.dtype, device=B.device)
        Bp[:, :r] = B
        t = ops.linear(x2, Ap)
        y = ops.linear(x2, self.base_layer.weight, self.base_layer.bias, extra=[ops.Seg(t, Bp)])
        return y.reshape(*lead, self.out_features)


class RMSNorm(nn.Module):
    """Per-head RMSNorm weight holder; the arithmetic is fused with RoPE in rf_qk_rmsnorm_rope."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        raise RuntimeError("RMSNorm is fused into rf_qk_rmsnorm_rope on the HIP path; it has no standalone forward")


def _mod_chunks(linear, emb):
    """linear(silu(emb)) through the HIP GEMM; emb [B, D] -> [B, n*D]."""
    return linear(ops.silu(emb))


def _ln_mod(x, scale, shift):
    out = torch.empty_like(x)
    for b in range(x.shape[0]):
        ops.layernorm_modulate(x[b], scale[b].contiguous(), shift[b].contiguous(), out=out[b])
    return out


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.linear = HipLinear(dim, 6 * dim)

    def forward(self, x, emb=None):
        e = _mod_chunks(self.linear, emb)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = e.chunk(6, dim=1)
        return _ln_mod(x, scale_msa, shift_msa), gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.linear = HipLinear(dim, 3 * dim)

    def forward(self, x, emb=None):
        shift, scale, gate = _mod_chunks(self.linear, emb).chunk(3, dim=1)
        return _ln_mod(x, scale, shift), gate


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim: int, cond_dim: int):
        super().__init__()
        self.linear = HipLinear(cond_dim, 2 * dim)

    def forward(self, x, conditioning_embedding):
        scale, shift = _mod_chunks(self.linear, conditioning_embedding).chunk(2, dim=1)   # scale FIRST
        return _ln_mod(x, scale, shift)


class GELU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = HipLinear(dim_in, dim_out)

    def forward(self, x):
        lead = x.shape[:-1]
        y = ops.linear(x.reshape(-1, x.shape[-1]), self.proj.weight, self.proj.bias, epilogue=ops.RF_EPI_GELU)
        return y.reshape(*lead, -1)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GELU(dim, dim * mult), nn.Identity(), HipLinear(dim * mult, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class Attention(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, pre_only: bool = False, eps: float = 1e-6):
        super().__init__()
        if head_dim != 128:
            raise ValueError("the HIP attention path is built for head_dim 128 (FLUX)")
        self.heads = heads
        inner = heads * head_dim
        self.to_q, self.to_k, self.to_v = HipLinear(dim, inner), HipLinear(dim, inner), HipLinear(dim, inner)
        self.norm_q, self.norm_k = RMSNorm(head_dim, eps), RMSNorm(head_dim, eps)
        self.pre_only = pre_only
        if not pre_only:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = (HipLinear(dim, inner) for _ in range(3))
            self.norm_added_q, self.norm_added_k = RMSNorm(head_dim, eps), RMSNorm(head_dim, eps)
            self.to_out = nn.ModuleList([HipLinear(inner, dim), nn.Identity()])
            self.to_add_out = HipLinear(inner, dim)


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim)
        self.norm1_context = AdaLayerNormZero(dim)
        self.attn = Attention(dim, heads, head_dim, pre_only=False)
        self.norm2 = nn.Identity()          # LayerNorm(no affine): parameter-free, fused
        self.ff = FeedForward(dim)
        self.norm2_context = nn.Identity()
        self.ff_context = FeedForward(dim)


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZeroSingle(dim)
        self.proj_mlp = HipLinear(dim, self.mlp_hidden_dim)
        self.act_mlp = nn.GELU(approximate="tanh")
        self.proj_out = HipLinear(dim + self.mlp_hidden_dim, dim)
        self.attn = Attention(dim, heads, head_dim, pre_only=True)


# ---- the parts the north star leaves in PyTorch-ROCm --------------------------------------------
def get_1d_rotary_pos_embed(dim: int, pos: torch.Tensor, theta: float = 10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64, device=pos.device)[: (dim // 2)] / dim))
    freqs = torch.outer(pos.to(torch.float64), freqs)
    return freqs.cos().repeat_interleave(2, dim=1).float(), freqs.sin().repeat_interleave(2, dim=1).float()


class FluxPosEmbed(nn.Module):
    def __init__(self, theta: int, axes_dim):
        super().__init__()
        self.theta, self.axes_dim = theta, tuple(axes_dim)

    def forward(self, ids: torch.Tensor):
        pos = ids.float()
        cs = [get_1d_rotary_pos_embed(self.axes_dim[i], pos[:, i], self.theta) for i in range(ids.shape[-1])]
        return torch.cat([c for c, _ in cs], dim=-1), torch.cat([s for _, s in cs], dim=-1)


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int = 256, max_period: int = 10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)      # flip_sin_to_cos=True


class _TwoLayerSiLU(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(in_dim, dim), nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    def __init__(self, dim: int, pooled_dim: int, guidance_embeds: bool = True):
        super().__init__()
        self.timestep_embedder = _TwoLayerSiLU(256, dim)
        if guidance_embeds:
            self.guidance_embedder = _TwoLayerSiLU(256, dim)
        self.text_embedder = _TwoLayerSiLU(pooled_dim, dim)
        self.guidance_embeds = guidance_embeds

    def forward(self, timestep, *rest):
        guidance, pooled = (rest if self.guidance_embeds else (None, rest[0]))
        emb = self.timestep_embedder(get_timestep_embedding(timestep).to(pooled.dtype))
        if guidance is not None:
            emb = emb + self.guidance_embedder(get_timestep_embedding(guidance).to(pooled.dtype))
        return emb + self.text_embedder(pooled)


class _Config(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


FLUX_DEV_CONFIG = dict(
    in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
    num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
    guidance_embeds=True, axes_dims_rope=(16, 56, 56),
)


class FluxTransformer2DModel(nn.Module):
    """FLUX.1-dev transformer (Appendix A.1).  `FluxTransformer2DModel()` is FLUX.1-dev sized."""

    def __init__(self, **cfg):
        super().__init__()
        c = dict(FLUX_DEV_CONFIG)
        c.update(cfg)
        self.config = _Config(c)
        dim = c["num_attention_heads"] * c["attention_head_dim"]
        self.inner_dim = dim
        self.pos_embed = FluxPosEmbed(10000, c["axes_dims_rope"])
        self.time_text_embed = CombinedTimestepGuidanceTextProjEmbeddings(
            dim, c["pooled_projection_dim"], c["guidance_embeds"])
        self.context_embedder = HipLinear(c["joint_attention_dim"], dim)
        self.x_embedder = HipLinear(c["in_channels"], dim)
        self.transformer_blocks = nn.ModuleList(
            [FluxTransformerBlock(dim, c["num_attention_heads"], c["attention_head_dim"]) for _ in range(c["num_layers"])])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(dim, c["num_attention_heads"], c["attention_head_dim"])
             for _ in range(c["num_single_layers"])])
        self.norm_out = AdaLayerNormContinuous(dim, dim)
        self.proj_out = HipLinear(dim, c["in_channels"])
        self.gradient_checkpointing = False

    @property
    def dtype(self):
        return self.x_embedder.weight.dtype

    @property
    def device(self):
        return self.x_embedder.weight.device


def init_synthetic_(model: nn.Module, seed: int = 0, std: float = 0.02, lora_std=None):
    """Random-init weights for benchmarking (no checkpoints offline): N(0, std^2) linears,
    RMSNorm scales 1 + N(0, std^2); seeded per parameter name so the values do not depend on
    module order or on LoRA wrappers being present (same recipe as the oracle's)."""
    import zlib

    with torch.no_grad():
        for name, p in model.named_parameters():
            canon = name.replace(".base_layer", "")
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(canon.encode())) % (2 ** 31))
            noise = torch.randn(p.shape, generator=g, dtype=torch.float32)
            if ".lora_A." in name:
                val = noise * (lora_std[0] if lora_std else 1.0 / p.shape[0])
            elif ".lora_B." in name:
                val = noise * (lora_std[1] if lora_std else std)
            elif name.endswith("weight") and p.ndim == 1:
                val = 1.0 + noise * std
            else:
                val = noise * std
            p.copy_(val.to(device=p.device, dtype=p.dtype))
    return model
