"""Minimal FluxPipeline with the surface the reference's `generate()` and tts scripts use
(`.transformer`, `.scheduler`, `.encode_prompt`, `.prepare_latents`,
`._pack_latents`, `._unpack_latents`, `._prepare_latent_image_ids`, `.load_lora_weights`,
`pipe(prompt=, latents=, guidance_scale=, num_inference_steps=, height=, width=).images`, ...).

diffusers is not vendored by the reference and cannot be installed here, so this shim carries the
pack/unpack/ids/scheduler logic itself (Appendix A.8, A.10).  Text encoders and the VAE are
OPTIONAL plug-ins (PyTorch-ROCm modules the caller supplies): without them prompts are mapped to
deterministic synthetic embeddings (no T5/CLIP weights offline) and only `output_type="latent"`
is available.
"""
from __future__ import annotations

import hashlib
import os
import re
from contextlib import contextmanager
from typing import Dict, List, Optional, Union

import torch

from .. import engine as E
from . import modules as M
from .scheduler import FlowMatchEulerDiscreteScheduler


class FluxPipelineOutput:
    def __init__(self, images):
        self.images = images


class SyntheticTextEncoder:
    """Stand-in for T5-XXL + CLIP-L: a prompt string -> reproducible N(0,1) embeddings
    (seeded by the prompt's SHA-256).  Declared synthetic in every bench/test that uses it."""

    def __init__(self, joint_dim: int = 4096, pooled_dim: int = 768):
        self.joint_dim, self.pooled_dim = joint_dim, pooled_dim

    def __call__(self, prompt: str, max_sequence_length: int, dtype, device):
        seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:4], "little") & 0x7FFFFFFF
        g = torch.Generator().manual_seed(seed)
        pe = torch.randn(max_sequence_length, self.joint_dim, generator=g)
        pooled = torch.randn(self.pooled_dim, generator=g)
        return pe.to(device=device, dtype=dtype), pooled.to(device=device, dtype=dtype)


class FluxPipeline:
    def __init__(self, transformer: M.FluxTransformer2DModel, scheduler=None, vae=None, text_encoder=None,
                 image_processor=None):
        self.transformer = transformer
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler()
        self.vae = vae
        self.text_encoder = text_encoder or SyntheticTextEncoder(
            transformer.config.joint_attention_dim, transformer.config.pooled_projection_dim)
        self.vae_scale_factor = 8
        if image_processor is None and vae is not None:
            from .vae import VaeImageProcessor
            image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor * 2)   # as diffusers' FluxPipeline
        self.image_processor = image_processor
        self.default_sample_size = 128
        self.interrupt = False
        self._joint_attention_kwargs = None
        self._guidance_scale = 3.5
        self._progress = {}

    # ---- construction --------------------------------------------------------------------------
    @classmethod
    def synthetic(cls, config: Optional[dict] = None, seed: int = 0, torch_dtype=torch.bfloat16, device=None,
                  with_vae: bool = False, vae_config: Optional[dict] = None):
        """Random-init FLUX-shaped transformer (there are no checkpoints offline); with_vae: also a random-init
        AutoencoderKL (FLUX.1-dev VAE shape unless vae_config says otherwise) so PIL in / PIL out works."""
        tr = M.FluxTransformer2DModel(**(config or {})).to(torch_dtype)
        M.init_synthetic_(tr, seed=seed)
        vae = None
        if with_vae:
            from .vae import AutoencoderKL, init_synthetic_vae_
            vae = init_synthetic_vae_(AutoencoderKL(**(vae_config or {})), seed=seed + 1).to(torch_dtype)
        pipe = cls(tr, vae=vae)
        return pipe.to(device) if device is not None else pipe

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, torch_dtype=torch.bfloat16, cache_dir=None, **kw):
        """Loads `<path>/transformer/*.safetensors` (diffusers layout) when the path exists locally."""
        root = pretrained_model_name_or_path
        if cache_dir and not os.path.isdir(root):
            root = os.path.join(cache_dir, pretrained_model_name_or_path)
        tdir = os.path.join(root, "transformer")
        if not os.path.isdir(tdir):
            raise FileNotFoundError(
                f"no local FLUX checkpoint at {tdir!r} (no network here). Use FluxPipeline.synthetic() for "
                "random-init weights of the same architecture.")
        from safetensors.torch import load_file
        # diffusers layout: transformer/config.json carries the architecture (FluxTransformer2DModel.__init__ kwargs);
        # absent -> FLUX.1-dev
        cfg, cfg_path = {}, os.path.join(tdir, "config.json")
        if os.path.exists(cfg_path):
            import json
            raw = json.load(open(cfg_path))
            cfg = {k: (tuple(raw[k]) if k == "axes_dims_rope" else raw[k]) for k in M.FLUX_DEV_CONFIG if k in raw}
        tr = M.FluxTransformer2DModel(**cfg).to(torch_dtype)
        sd = {}
        for f in sorted(os.listdir(tdir)):
            if f.endswith(".safetensors"):
                sd.update(load_file(os.path.join(tdir, f)))
        missing, unexpected = tr.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"FLUX checkpoint mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        # optional diffusers `vae/` sub-directory (AutoencoderKL; stays a PyTorch-ROCm module)
        vae, vdir = None, os.path.join(root, "vae")
        if os.path.isdir(vdir):
            from .vae import FLUX_VAE_CONFIG, AutoencoderKL
            vcfg, vcfg_path = {}, os.path.join(vdir, "config.json")
            if os.path.exists(vcfg_path):
                import json
                raw = json.load(open(vcfg_path))
                vcfg = {k: raw[k] for k in FLUX_VAE_CONFIG if k in raw}
            vae = AutoencoderKL(**vcfg).to(torch_dtype)
            vsd = {}
            for f in sorted(os.listdir(vdir)):
                if f.endswith(".safetensors"):
                    vsd.update(load_file(os.path.join(vdir, f)))
            missing, unexpected = vae.load_state_dict(vsd, strict=False)
            if missing or unexpected:
                raise RuntimeError(f"VAE checkpoint mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        return cls(tr, vae=vae)

    def to(self, device=None, dtype=None):
        self.transformer.to(device=device, dtype=dtype)
        for m in (self.vae,):
            if m is not None and hasattr(m, "to"):
                m.to(device=device, dtype=dtype)
        E.invalidate(self.transformer)
        return self

    def enable_hip_vae(self):
        """Run `vae.decode` / `vae.encode` on the HIP path (librf_flux.so rf_vae_decode / rf_vae_encode)
        instead of the PyTorch-ROCm / MIOpen modules: wraps the AutoencoderKL already on this pipeline (bf16, on the GPU).
        `generate(output_type="pil")`, `Condition.encode` and the runner's decode -> resize -> encode hand-off then launch no
        MIOpen kernel.  Idempotent."""
        from .vae_hip import HipVAE
        if self.vae is None:
            raise ValueError("enable_hip_vae(): this pipeline has no VAE")
        if not isinstance(self.vae, HipVAE):
            self.vae = HipVAE(self.vae)
        return self

    def enable_hip_text_encoders(self, t5_state_dict=None, clip_state_dict=None, tokenize=None, root: str = None, t5_heads: int = 64,
                                 clip_heads: int = 12, clip_eos_token_id: int = 2):
        """Run FluxPipeline.encode_prompt's two encoder calls (generate.py:148-161) on the HIP path (rf_t5_encode / rf_clip_text_encode,
        SURVEY 8f row 2).  Weights: transformers-layout state dicts, or `root` = a diffusers FLUX directory whose `text_encoder_2/` and
        `text_encoder/` sub-directories hold the safetensors (+ config.json for the head counts) and whose `tokenizer/`, `tokenizer_2/` hold the
        vocabularies (flux/tokenizers.py).  `tokenize(prompts, max_sequence_length) -> (t5_ids [B, L], clip_ids [B, 77])` overrides them."""
        from .text_hip import HipClipTextEncoder, HipT5Encoder, HipTextEncoders
        if tokenize is None:
            if root is None:
                raise ValueError("enable_hip_text_encoders(): a tokenize(prompts, max_sequence_length) callable or a checkpoint root is required")
            from .tokenizers import load_flux_tokenizers
            tokenize = load_flux_tokenizers(root)          # <root>/tokenizer + <root>/tokenizer_2, transformers' tokenizer classes
        if root is not None:
            import json
            from safetensors.torch import load_file

            def load_dir(d):
                sd = {}
                for f in sorted(os.listdir(d)):
                    if f.endswith(".safetensors"):
                        sd.update(load_file(os.path.join(d, f)))
                cfg = json.load(open(os.path.join(d, "config.json"))) if os.path.exists(os.path.join(d, "config.json")) else {}
                return sd, cfg
            t5_state_dict, c2 = load_dir(os.path.join(root, "text_encoder_2"))
            clip_state_dict, c1 = load_dir(os.path.join(root, "text_encoder"))
            t5_heads, clip_heads = c2.get("num_heads", t5_heads), c1.get("num_attention_heads", clip_heads)
            clip_eos_token_id = c1.get("eos_token_id", clip_eos_token_id)
            # everything else the kernels assume is READ from the configs and refused when it differs (no silent defaults)
            act = c2.get("feed_forward_proj", "gated-gelu")
            if act != "gated-gelu" or c2.get("d_kv", 64) != 64 or not c2.get("is_gated_act", True):
                raise ValueError(f"enable_hip_text_encoders(): the HIP T5 path is T5 v1.1 (gated-gelu, d_kv 64); config has feed_forward_proj={act!r}, d_kv={c2.get('d_kv')}")
            if c1.get("hidden_act", "quick_gelu") != "quick_gelu":
                raise ValueError(f"enable_hip_text_encoders(): the HIP CLIP path uses quick_gelu; config has hidden_act={c1.get('hidden_act')!r}")
            t5_kw = dict(eps=float(c2.get("layer_norm_epsilon", 1e-6)), num_buckets=int(c2.get("relative_attention_num_buckets", 32)),
                         max_distance=int(c2.get("relative_attention_max_distance", 128)))
            clip_kw = dict(eps=float(c1.get("layer_norm_eps", 1e-5)))
        else:
            t5_kw, clip_kw = {}, {}
        if t5_state_dict is None or clip_state_dict is None:
            raise ValueError("enable_hip_text_encoders(): state dicts or a checkpoint root are required")
        dev = self.device
        self.text_encoder = HipTextEncoders(HipT5Encoder(t5_state_dict, t5_heads, dev, **t5_kw),
                                            HipClipTextEncoder(clip_state_dict, clip_heads, dev, eos_token_id=clip_eos_token_id, **clip_kw), tokenize)
        return self

    def enable_merged_lora(self, on: bool = True):
        """See engine.set_merged_lora: static LoRA folded into per-token-group weight copies (no low-rank launches per step)."""
        E.set_merged_lora(self.transformer, on)
        return self

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    @property
    def device(self):
        return self.transformer.device

    _execution_device = device

    @property
    def joint_attention_kwargs(self):
        """diffusers exposes the kwargs of the running call as a read-only property; generate() sets the
        private attribute (reference generate.py:139) and the per-step path reads this one (:243)."""
        return self._joint_attention_kwargs

    @property
    def dtype(self):
        return self.transformer.dtype

    # ---- LoRA (FLUX-Corrector) -----------------------------------------------------------------
    def load_lora_weights(self, path_or_state_dict: Union[str, Dict[str, torch.Tensor]], adapter_name: str = "default",
                          weight_name: str = "pytorch_lora_weights.safetensors", alpha: Optional[float] = None):
        """PEFT/diffusers LoRA file (`transformer.<module>.lora_{A,B}.weight`, train/model.py:87-92) ->
        LoraLinear wrappers on the target modules.  scaling = alpha/r, alpha defaults to r
        (train_flux/config.yaml:50-51)."""
        if isinstance(path_or_state_dict, str):
            from safetensors.torch import load_file
            p = path_or_state_dict
            if os.path.isdir(p):
                p = os.path.join(p, weight_name)
            sd = load_file(p)
        else:
            sd = dict(path_or_state_dict)
        pat = re.compile(r"^(?:transformer\.)?(.+)\.lora_A(?:\.[^.]+)?\.weight$")
        E.check_lora_placement(names=[pat.match(k).group(1) for k in sd if pat.match(k)])   # before touching the model
        n = 0
        for key in sorted(sd):
            m = pat.match(key)
            if not m:
                continue
            name = m.group(1)
            kb = key.replace("lora_A", "lora_B")
            A, B = sd[key], sd[kb]
            parent, leaf = _get_parent(self.transformer, name)
            base = parent[int(leaf)] if leaf.isdigit() else getattr(parent, leaf)
            if isinstance(base, M.LoraLinear):
                base = base.base_layer
            r = A.shape[0]
            wrapped = M.LoraLinear(base, r, alpha if alpha is not None else float(r), adapter_name)
            with torch.no_grad():
                wrapped.lora_A[adapter_name].weight.copy_(A.to(base.weight))
                wrapped.lora_B[adapter_name].weight.copy_(B.to(base.weight))
            if leaf.isdigit():
                parent[int(leaf)] = wrapped
            else:
                setattr(parent, leaf, wrapped)
            n += 1
        if n == 0:
            raise ValueError("no `*.lora_A.weight` keys found in the LoRA state dict")
        E.invalidate(self.transformer)
        return n

    def set_adapters(self, *a, **k):
        pass

    # ---- BASELINE cfg5: fp8 weights ------------------------------------------------------------------------------
    def enable_fp8_weights(self, enabled: bool = True):
        """Run the 57 blocks' big GEMMs in fp8 (OCP e4m3fn): weights quantised once per output channel, activations
        per token on the fly (include/rf_flux.h: rf_gemm_w8a8, rf_flux_dims.fp8).  Token streams that carry a LoRA
        (the condition rows; the image rows under latent_lora) keep the bf16 kernels.  Not a reference feature
        (the reference is bf16/fp32 only): accuracy cost and tolerance are stated in DESIGN.md / tests/test_w8_gpu.py."""
        object.__setattr__(self.transformer, "_rf_fp8", bool(enabled))
        E.invalidate(self.transformer)
        return self

    # ---- diffusers FluxPipeline helpers (Appendix A.10) ------------------------------------------
    def check_inputs(self, prompt, prompt_2, height, width, prompt_embeds=None, pooled_prompt_embeds=None,
                     callback_on_step_end_tensor_inputs=None, max_sequence_length=None):
        if height % 16 != 0 or width % 16 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 16 but are {height} and {width}.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.")
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f"`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}")

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt: int = 1,
                      prompt_embeds=None, pooled_prompt_embeds=None, max_sequence_length: int = 512, lora_scale=None):
        device = device or self.device
        if prompt_embeds is None:
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompts_2 = prompts if prompt_2 is None else ([prompt_2] if isinstance(prompt_2, str) else list(prompt_2))
            if hasattr(self.text_encoder, "encode_t5") and hasattr(self.text_encoder, "encode_clip"):
                # HipTextEncoders: each tower runs once, on the prompts it sees, and ALL prompts of the call share its launches
                prompt_embeds = self.text_encoder.encode_t5(prompts_2, max_sequence_length, self.dtype, device)
                pooled_prompt_embeds = self.text_encoder.encode_clip(prompts, self.dtype, device)
            else:
                pes, pools = [], []
                for p1, p2 in zip(prompts, prompts_2):
                    pe, _ = self.text_encoder(p2, max_sequence_length, self.dtype, device)    # T5 sees prompt_2
                    _, pooled = self.text_encoder(p1, max_sequence_length, self.dtype, device)  # CLIP sees prompt
                    pes.append(pe)
                    pools.append(pooled)
                prompt_embeds, pooled_prompt_embeds = torch.stack(pes), torch.stack(pools)
        if num_images_per_prompt != 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, 0)
            pooled_prompt_embeds = pooled_prompt_embeds.repeat_interleave(num_images_per_prompt, 0)
        prompt_embeds = prompt_embeds.to(device=device, dtype=self.dtype)
        pooled_prompt_embeds = pooled_prompt_embeds.to(device=device, dtype=self.dtype)
        text_ids = torch.zeros(prompt_embeds.shape[1], 3).to(device=device, dtype=self.dtype)
        return prompt_embeds, pooled_prompt_embeds, text_ids

    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        latents = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2)
        latents = latents.permute(0, 2, 4, 1, 3, 5)
        return latents.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        batch_size, _, channels = latents.shape
        height = 2 * (int(height) // (vae_scale_factor * 2))
        width = 2 * (int(width) // (vae_scale_factor * 2))
        latents = latents.view(batch_size, height // 2, width // 2, channels // 4, 2, 2)
        latents = latents.permute(0, 3, 1, 4, 2, 5)
        return latents.reshape(batch_size, channels // 4, height, width)

    @staticmethod
    def _prepare_latent_image_ids(batch_size, height, width, device, dtype):
        ids = torch.zeros(height, width, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(height)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(width)[None, :]
        return ids.reshape(height * width, 3).to(device=device, dtype=dtype)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        height = 2 * (int(height) // (self.vae_scale_factor * 2))
        width = 2 * (int(width) // (self.vae_scale_factor * 2))
        ids = self._prepare_latent_image_ids(batch_size, height // 2, width // 2, device, dtype)
        if latents is not None:
            return latents.to(device=device, dtype=dtype), ids
        shape = (batch_size, num_channels_latents, height, width)
        gdev = generator.device if generator is not None else torch.device("cpu")
        latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        return self._pack_latents(latents, batch_size, num_channels_latents, height, width), ids

    @contextmanager
    def progress_bar(self, total=None):
        class _PB:
            def update(self_inner, n=1):
                pass
        yield _PB()

    def maybe_free_model_hooks(self):
        pass

    # ---- stock text-to-image call (tts_t2i_noise_scaling.py:60) ---------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 28, guidance_scale: float = 3.5, latents=None, prompt_embeds=None,
                 pooled_prompt_embeds=None, output_type: str = "pil", generator=None, max_sequence_length: int = 512,
                 **kw):
        from .generate import generate
        return generate(self, conditions=None, model_config={}, prompt=prompt, prompt_2=prompt_2, height=height,
                        width=width, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                        latents=latents, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                        output_type=output_type, generator=generator, max_sequence_length=max_sequence_length, **kw)


def _get_parent(model, dotted: str):
    parts = dotted.split(".")
    parent = model
    for p in parts[:-1]:
        parent = parent[int(p)] if p.isdigit() else getattr(parent, p)
    return parent, parts[-1]


def lora_target_names(transformer) -> List[str]:
    """The modules the FLUX-Corrector LoRA touches: the regex at train_flux/config.yaml:53 expanded
    against the module tree."""
    names = ["x_embedder"]
    for i in range(len(transformer.transformer_blocks)):
        p = f"transformer_blocks.{i}."
        names += [p + "norm1.linear", p + "attn.to_q", p + "attn.to_k", p + "attn.to_v", p + "attn.to_out.0",
                  p + "ff.net.2"]
    for i in range(len(transformer.single_transformer_blocks)):
        p = f"single_transformer_blocks.{i}."
        names += [p + "norm.linear", p + "proj_mlp", p + "proj_out", p + "attn.to_q", p + "attn.to_k", p + "attn.to_v"]
    return names


def synthetic_lora_state_dict(transformer, r: int = 32, seed: int = 0, std_b: float = 0.02) -> Dict[str, torch.Tensor]:
    """A FLUX-Corrector-shaped LoRA file with random factors (A ~ N(0,1/r^2), B ~ N(0,std_b^2))."""
    import zlib
    sd = {}
    for name in lora_target_names(transformer):
        parent, leaf = _get_parent(transformer, name)
        base = parent[int(leaf)] if leaf.isdigit() else getattr(parent, leaf)
        if isinstance(base, M.LoraLinear):
            base = base.base_layer
        for which, shape, std in (("lora_A", (r, base.in_features), 1.0 / r), ("lora_B", (base.out_features, r), std_b)):
            key = f"transformer.{name}.{which}.weight"
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
            sd[key] = (torch.randn(shape, generator=g) * std).to(torch.bfloat16)
    return sd
