"""Test-only emulation of the fp8 (W8A8) semantics of rf_gemm_w8a8 on top of the fp32 oracle, WITHOUT touching the
oracle's restated functions: the nn.Linear modules the product runs in fp8 are wrapped so that, for the token streams
the product quantises, their input rows are fake-quantised per token (e4m3fn, scale = amax/448) and their weight is
the dequantised per-output-channel e4m3fn copy of the bf16 weight.  Everything else (accumulation, bias, the rest of
the block) stays the oracle's fp32 arithmetic, so   product(fp8)  vs  emulated oracle   isolates implementation
errors from the (large, by construction) quantisation error.

Which rows are quantised is decided by the row count of the call (`lengths`): the oracle pushes text, image and
condition rows through the same modules in separate calls, and the product keeps LoRA'd streams (condition rows) in
bf16 -- the tests use distinct stream lengths."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import flux_oracle as O

FP8_MAX = 448.0


def fq_rows(x: torch.Tensor) -> torch.Tensor:
    """Per-row (last dim) symmetric e4m3fn fake quantisation, the kernels' arithmetic (tests/test_w8_gpu.py)."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax * (1.0 / FP8_MAX), torch.ones_like(amax))
    q = (xf * (1.0 / scale)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()
    return (q * scale).to(x.dtype)


def fq_weight(W: torch.Tensor) -> torch.Tensor:
    """Dequantised per-output-channel e4m3fn copy of the bf16 weight (what engine.py packs)."""
    Wb = W.detach().to(torch.bfloat16).float()
    amax = Wb.abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    q = (Wb / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()
    return (q * scale).to(W.dtype)


class FakeQuantLinear(nn.Module):
    """`stored_bf16`: the product keeps this linear's INPUT in HBM as bf16 before it quantises it (the attention output ATT
    and the FF hidden HID go through rf_quant_rows_fp8 from bf16 rows; the LayerNorm+modulate outputs are quantised straight
    from fp32 registers).  With `emulate_storage` the emulation rounds such inputs to bf16 first, so that product and
    emulation quantise (almost) the same numbers and far fewer e4m3 codes flip between them."""
    emulate_storage = False

    def __init__(self, base: nn.Linear, lengths, stored_bf16: bool = False):
        super().__init__()
        self.base, self.lengths, self.stored_bf16 = base, set(lengths), stored_bf16
        self.in_features, self.out_features = base.in_features, base.out_features

    @property
    def weight(self):
        return self.base.weight

    @property
    def bias(self):
        return self.base.bias

    def forward(self, x):
        if x.shape[-2] not in self.lengths:
            return self.base(x)
        if self.stored_bf16 and FakeQuantLinear.emulate_storage:
            x = x.to(torch.bfloat16).to(x.dtype)
        return F.linear(fq_rows(x), fq_weight(self.base.weight), self.base.bias)


_DOUBLE = ("attn.to_q", "attn.to_k", "attn.to_v", "attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "attn.to_out.0",
           "attn.to_add_out", "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2")
_SINGLE = ("attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp", "proj_out")
_STORED_BF16 = ("attn.to_out.0", "attn.to_add_out", "ff.net.2", "ff_context.net.2", "proj_out")   # inputs: ATT / HID rows in bf16


def _wrap(block: nn.Module, names, lengths):
    for name in names:
        parent, leaf = O._get_parent(block, name)
        mod = parent[int(leaf)] if leaf.isdigit() else getattr(parent, leaf)
        if isinstance(mod, O.LoraLinear):                      # quantise underneath the LoRA wrapper
            if not isinstance(mod.base_layer, FakeQuantLinear):
                mod.base_layer = FakeQuantLinear(mod.base_layer, lengths, name in _STORED_BF16)
            else:
                mod.base_layer.lengths = set(lengths)
            continue
        if isinstance(mod, FakeQuantLinear):
            mod.lengths = set(lengths)
            continue
        w = FakeQuantLinear(mod, lengths, name in _STORED_BF16)
        if leaf.isdigit():
            parent[int(leaf)] = w
        else:
            setattr(parent, leaf, w)


def emulate_fp8_block(block: nn.Module, lengths):
    """Wrap one oracle FluxTransformerBlock / FluxSingleTransformerBlock in place."""
    _wrap(block, _DOUBLE if hasattr(block, "ff_context") else _SINGLE, lengths)
    return block


def emulate_fp8(model: nn.Module, S_txt: int, S_img: int):
    """Wrap every block of an oracle FluxTransformer2DModel in place: text rows (S_txt), image rows (S_img) and the
    single blocks' [text; image] rows (S_txt + S_img) are quantised; any other row count (the condition stream) is not."""
    for b in model.transformer_blocks:
        emulate_fp8_block(b, (S_txt, S_img))
    for b in model.single_transformer_blocks:
        emulate_fp8_block(b, (S_txt + S_img,))
    return model


def remove_emulation(model: nn.Module):
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if isinstance(child, FakeQuantLinear):
                setattr(parent, name, child.base)
    for parent in model.modules():
        if isinstance(parent, (nn.ModuleList, nn.Sequential)):
            for i, child in enumerate(parent):
                if isinstance(child, FakeQuantLinear):
                    parent[i] = child.base
    return model
