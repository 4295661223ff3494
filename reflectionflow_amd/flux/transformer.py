"""HIP-backed `tranformer_forward` (sic -- the reference's spelling is the API:
train_flux/flux/transformer.py:47-55).  Same keyword surface (`prepare_params`, :18-44) and the
same return convention (`(sample,)` when return_dict=False, :250-251).

What stays in PyTorch-ROCm, as the north star prescribes: the timestep/guidance/pooled-text
embedding (a few [B,3072] linears) and the RoPE cos/sin tables.  Everything else -- embedders,
19 double + 38 single blocks, final AdaLN + projection -- is ONE C call, rf_flux_forward.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from .. import engine as E
from .. import ops


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample


def prepare_params(hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None,
                   txt_ids=None, guidance=None, joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                   controlnet_block_samples=None, controlnet_single_block_samples=None, return_dict: bool = True,
                   **kwargs):
    return (hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
            joint_attention_kwargs, controlnet_block_samples, controlnet_single_block_samples, return_dict)


def _attn_c_factor(transformer) -> Optional[float]:
    """generate() sets `attn.c_factor` on every attention module when condition_scale != 1
    (reference generate.py:86-90); the blocks read it off the module (block.py:115)."""
    blocks = list(transformer.transformer_blocks) + list(transformer.single_transformer_blocks)
    if not blocks:
        return None
    cf = getattr(blocks[0].attn, "c_factor", None)
    return None if cf is None else float(cf.reshape(-1)[0])


def tranformer_forward(transformer, condition_latents, condition_ids, condition_type_ids,
                       model_config: Optional[Dict[str, Any]] = {}, c_t=0, **params):
    model_config = model_config or {}
    (hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
     joint_attention_kwargs, controlnet_block_samples, controlnet_single_block_samples, return_dict
     ) = prepare_params(**params)
    if controlnet_block_samples is not None or controlnet_single_block_samples is not None:
        raise ops.RFError("ControlNet residuals are not on the ReflectionFlow path (transformer.py:172-181,230-239 unused)")
    if joint_attention_kwargs is not None and joint_attention_kwargs.get("scale", 1.0) != 1.0:
        raise ops.RFError("a global LoRA `scale` != 1 is not supported on the HIP path")
    eng = E.engine_for(transformer)
    use_condition = condition_latents is not None
    dtype = transformer.dtype
    for t, n in ((hidden_states, "hidden_states"), (encoder_hidden_states, "encoder_hidden_states")):
        if not t.is_cuda:
            raise ops.RFError(f"tranformer_forward: {n} is on {t.device}; the HIP path has no CPU fallback")

    # transformer.py:95-114 -- bf16-quantised (t*1000, g*1000) are inherited, not "fixed"
    timestep = timestep.to(dtype) * 1000
    guidance = guidance.to(dtype) * 1000 if guidance is not None else None
    temb = eng.temb(timestep, guidance, pooled_projections.to(dtype))
    latent_lora = model_config.get("latent_lora", False)
    mod_main = eng.mod_table(temb, lora=latent_lora)                               # [B, cols]
    mod_cond = None
    if use_condition:
        ct = torch.ones_like(timestep) * c_t * 1000
        cg = torch.ones_like(guidance) * 1000 if guidance is not None else None
        cond_temb = eng.temb(ct, cg, pooled_projections.to(dtype))
        mod_cond = eng.mod_table(cond_temb, lora=True)

    if txt_ids.ndim == 3:
        txt_ids = txt_ids[0]
    if img_ids.ndim == 3:
        img_ids = img_ids[0]
    cos, sin = eng.rope_tables(txt_ids, img_ids, condition_ids if use_condition else None)

    B = hidden_states.shape[0]
    x = hidden_states.to(dtype).contiguous()
    ctx = encoder_hidden_states.to(dtype).contiguous()
    cond = condition_latents.to(dtype).contiguous() if use_condition else None
    out = torch.empty_like(x)
    cf = _attn_c_factor(transformer)
    for b in range(B):
        eng.forward(x[b], ctx[b], mod_main[b], cos, sin, cond_latents=cond[b] if use_condition else None,
                    mod_cond=mod_cond[b] if use_condition else None, model_config=model_config, c_factor=cf,
                    out=out[b])
    if not return_dict:
        return (out,)
    return Transformer2DModelOutput(sample=out)
