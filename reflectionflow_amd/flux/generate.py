"""HIP-backed `generate` -- the sampler the search drivers call per candidate.
Same name, keyword surface (`prepare_params`) and return object as the reference's
train_flux/flux/generate.py:74-321.

The T-step loop (reference :216-296) has two forms here:
  * the FAST path -- no per-step callback, image_guidance_scale == 1 (what the tts scripts use):
    the modulation tables of all T timesteps are computed up front, then ONE C call
    (rf_flux_denoise) runs T x (forward + Euler) back to back on the stream;
  * the GENERAL path -- per step `tranformer_forward` + `scheduler.step`, supporting
    callback_on_step_end and the image-CFG second pass (:250-272), quirks included.
Text encoding, condition VAE-encode and VAE decode stay PyTorch modules on the pipeline object.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from .. import engine as E
from .condition import Condition
from .pipeline import FluxPipelineOutput
from .scheduler import calculate_shift, retrieve_timesteps
from .transformer import _attn_c_factor, tranformer_forward


def get_config(config_path: str = None):
    config_path = config_path or os.environ.get("XFL_CONFIG")
    if not config_path:
        return {}
    import yaml
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def prepare_params(prompt: Union[str, List[str]] = None, prompt_2=None, height: Optional[int] = 512,
                   width: Optional[int] = 512, num_inference_steps: int = 28, timesteps: List[int] = None,
                   guidance_scale: float = 3.5, num_images_per_prompt: Optional[int] = 1, generator=None,
                   latents=None, prompt_embeds=None, pooled_prompt_embeds=None, output_type: Optional[str] = "pil",
                   return_dict: bool = True, joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                   callback_on_step_end: Optional[Callable] = None,
                   callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 512,
                   **kwargs):
    return (prompt, prompt_2, height, width, num_inference_steps, timesteps, guidance_scale, num_images_per_prompt,
            generator, latents, prompt_embeds, pooled_prompt_embeds, output_type, return_dict, joint_attention_kwargs,
            callback_on_step_end, callback_on_step_end_tensor_inputs, max_sequence_length)


def seed_everything(seed: int = 42):
    torch.manual_seed(seed)
    np.random.seed(seed)


@torch.no_grad()
def generate(pipeline, conditions: List[Condition] = None, config_path: str = None,
             model_config: Optional[Dict[str, Any]] = {}, condition_scale: float = 1.0, default_lora: bool = False,
             image_guidance_scale: float = 1.0, **params):
    model_config = model_config or get_config(config_path).get("model", {})
    if condition_scale != 1:                       # reference :86-90
        for name, module in pipeline.transformer.named_modules():
            if name.endswith(".attn"):
                module.c_factor = torch.ones(1, 1) * condition_scale
    self = pipeline
    (prompt, prompt_2, height, width, num_inference_steps, timesteps, guidance_scale, num_images_per_prompt,
     generator, latents, prompt_embeds, pooled_prompt_embeds, output_type, return_dict, joint_attention_kwargs,
     callback_on_step_end, callback_on_step_end_tensor_inputs, max_sequence_length) = prepare_params(**params)
    height = height or self.default_sample_size * self.vae_scale_factor
    width = width or self.default_sample_size * self.vae_scale_factor
    self.check_inputs(prompt, prompt_2, height, width, prompt_embeds=prompt_embeds,
                      pooled_prompt_embeds=pooled_prompt_embeds,
                      callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs,
                      max_sequence_length=max_sequence_length)
    self._guidance_scale = guidance_scale
    self._joint_attention_kwargs = joint_attention_kwargs
    self.interrupt = False
    if prompt is not None and isinstance(prompt, str):
        batch_size = 1
    elif prompt is not None and isinstance(prompt, list):
        batch_size = len(prompt)
    else:
        batch_size = prompt_embeds.shape[0]
    device = self._execution_device
    prompt_embeds, pooled_prompt_embeds, text_ids = self.encode_prompt(
        prompt=prompt, prompt_2=prompt_2, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
        device=device, num_images_per_prompt=num_images_per_prompt, max_sequence_length=max_sequence_length,
        lora_scale=None)

    num_channels_latents = self.transformer.config.in_channels // 4
    latents, latent_image_ids = self.prepare_latents(
        batch_size * num_images_per_prompt, num_channels_latents, height, width, prompt_embeds.dtype, device,
        generator, latents)

    # 4.1 conditions (reference :177-190; `conditions is not None or []` quirk kept)
    condition_latents, condition_ids, condition_type_ids = ([] for _ in range(3))
    use_condition = conditions is not None or []
    # geometry the HIP VAE cannot take is refused BEFORE the denoise loop, not after it
    if hasattr(self.vae, "check_geometry"):
        if output_type != "latent":
            self.vae.check_geometry(height, width, "output image")
        for c in (conditions or []) if use_condition else []:
            if getattr(c, "condition", None) is not None and hasattr(c.condition, "size"):
                self.vae.check_geometry(c.condition.size[1], c.condition.size[0], "condition image")
    condition = None
    if use_condition:
        assert len(conditions) <= 1, "Only one condition is supported for now."
        if not default_lora:
            pipeline.set_adapters(conditions[0].condition_type)
        for condition in conditions:
            tokens, ids, type_id = condition.encode(self)
            condition_latents.append(tokens)
            condition_ids.append(ids)
            condition_type_ids.append(type_id)
        condition_latents = torch.cat(condition_latents, dim=1).to(device=device, dtype=prompt_embeds.dtype)
        condition_ids = torch.cat(condition_ids, dim=0)
        condition_type_ids = torch.cat(condition_type_ids, dim=0)

    # 5. timesteps (reference :193-209)
    sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
    image_seq_len = latents.shape[1]
    mu = calculate_shift(image_seq_len, self.scheduler.config.base_image_seq_len,
                         self.scheduler.config.max_image_seq_len, self.scheduler.config.base_shift,
                         self.scheduler.config.max_shift)
    timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps,
                                                        sigmas, mu=mu)

    fast = callback_on_step_end is None and image_guidance_scale == 1.0 and joint_attention_kwargs is None
    if fast:
        latents = _denoise_fast(self, latents, prompt_embeds, pooled_prompt_embeds, text_ids, latent_image_ids,
                                timesteps, guidance_scale, condition_latents if use_condition else None,
                                condition_ids if use_condition else None, model_config)
    else:
        with self.progress_bar(total=num_inference_steps) as progress_bar:
            for i, t in enumerate(timesteps):
                if self.interrupt:
                    continue
                timestep = t.expand(latents.shape[0]).to(latents.dtype)           # :222
                if self.transformer.config.guidance_embeds:
                    guidance = torch.tensor([guidance_scale], device=device).expand(latents.shape[0])
                else:
                    guidance = None
                common = dict(model_config=model_config, hidden_states=latents, timestep=timestep / 1000,
                              pooled_projections=pooled_prompt_embeds, encoder_hidden_states=prompt_embeds,
                              txt_ids=text_ids, img_ids=latent_image_ids,
                              joint_attention_kwargs=self.joint_attention_kwargs, return_dict=False,
                              condition_ids=condition_ids if use_condition else None,
                              condition_type_ids=condition_type_ids if use_condition else None)
                noise_pred = tranformer_forward(self.transformer, guidance=guidance,
                                                condition_latents=condition_latents if use_condition else None,
                                                **common)[0]
                if image_guidance_scale != 1.0:                                   # :250-272
                    uncondition_latents = condition.encode(self, empty=True)[0].to(latents)
                    unc_pred = tranformer_forward(self.transformer, guidance=torch.ones_like(guidance),
                                                  condition_latents=uncondition_latents if use_condition else None,
                                                  **common)[0]
                    noise_pred = unc_pred + image_guidance_scale * (noise_pred - unc_pred)
                latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
                if callback_on_step_end is not None:
                    avail = {"latents": latents, "prompt_embeds": prompt_embeds, "noise_pred": noise_pred}
                    callback_kwargs = {k: avail[k] for k in callback_on_step_end_tensor_inputs}
                    callback_outputs = callback_on_step_end(self, i, t, callback_kwargs) or {}
                    latents = callback_outputs.pop("latents", latents)
                    prompt_embeds = callback_outputs.pop("prompt_embeds", prompt_embeds)
                progress_bar.update()

    if output_type == "latent":
        image = latents
    else:
        if self.vae is None or self.image_processor is None:
            raise RuntimeError("output_type != 'latent' needs a VAE and an image processor on the pipeline "
                               "(they stay PyTorch-ROCm modules; none are available offline)")
        latents = self._unpack_latents(latents, height, width, self.vae_scale_factor)
        latents = (latents / self.vae.config.scaling_factor) + self.vae.config.shift_factor
        image = self.vae.decode(latents, return_dict=False)[0]
        image = self.image_processor.postprocess(image, output_type=output_type)
    self.maybe_free_model_hooks()
    if condition_scale != 1:
        for name, module in pipeline.transformer.named_modules():
            if name.endswith(".attn"):
                del module.c_factor
    if not return_dict:
        return (image,)
    return FluxPipelineOutput(images=image)


def _denoise_fast(pipe, latents, prompt_embeds, pooled, text_ids, img_ids, timesteps, guidance_scale,
                  condition_latents, condition_ids, model_config):
    """All-steps modulation tables up front, then one rf_flux_denoise call per sample."""
    tr = pipe.transformer
    eng = E.engine_for(tr)
    dtype = tr.dtype
    B, T = latents.shape[0], len(timesteps)
    dev = latents.device
    use_condition = condition_latents is not None
    latent_lora = model_config.get("latent_lora", False)
    # per-step conditioning exactly as the per-step path computes it: t -> latents.dtype (:222),
    # /1000 (:235), then .to(dtype)*1000 inside the transformer (transformer.py:95)
    ts = (timesteps.to(dev).to(latents.dtype) / 1000).to(dtype) * 1000            # [T]
    cos, sin = eng.rope_tables(text_ids, img_ids, condition_ids if use_condition else None)
    dts = pipe.scheduler.dts()
    cf = _attn_c_factor(tr)
    out = latents.to(dtype).contiguous().clone()
    for b in range(B):
        p = pooled[b:b + 1].to(dtype).expand(T, -1)
        if tr.config.guidance_embeds:
            g = (torch.full((T,), float(guidance_scale), device=dev)).to(dtype) * 1000
            temb = eng.temb(ts, g, p)
        else:
            temb = eng.temb(ts, None, p)
        mod_steps = eng.mod_table(temb, lora=latent_lora)                          # [T, cols]
        mod_cond = None
        if use_condition:
            c_ts = torch.zeros(1, device=dev, dtype=dtype)                          # c_t = 0
            c_g = torch.ones(1, device=dev, dtype=dtype) * 1000 if tr.config.guidance_embeds else None
            mod_cond = eng.mod_table(eng.temb(c_ts, c_g, pooled[b:b + 1].to(dtype)), lora=True)[0]
        eng.denoise(out[b], prompt_embeds[b].to(dtype).contiguous(), mod_steps, dts, cos, sin,
                    cond_latents=condition_latents[b].to(dtype).contiguous() if use_condition else None,
                    mod_cond=mod_cond, model_config=model_config, c_factor=cf)
    pipe.scheduler._step_index = T
    return out
