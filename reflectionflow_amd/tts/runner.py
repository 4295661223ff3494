"""Candidate-parallel search runner shared by the two driver scripts.

Mirrors the control flow of the reference drivers for the denoise path only:
  * noise scaling  (tts/tts_t2i_noise_scaling.py:126-159, sample :16-77): per prompt x round draw
    `search_branch` seeded noises and generate one candidate per noise -- no dependency between rounds;
  * reflection loop (tts/tts_reflectionflow.py:591-629, sample :94-463): score the previous round,
    keep the top-k, build a "cot" Condition from each kept candidate, generate the next round with
    the FLUX-Corrector LoRA active on the condition tokens, re-score.
What is NOT here (outside the hot path): the GPT-4o / NVILA verifiers and
the reflection / prompt-refinement LLM calls.  `search.stub_verifier` stands in for their output
contract; prompts are therefore not rewritten between rounds.

Sharding: candidate i of a round runs on rank i % world_size; the only collective is the all-gather
of {score, label} at the round boundary (search.allgather_scores).  Seeds are a pure function of
(prompt index, round, candidate), so results do not depend on the world size.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import torch

from ..flux.condition import Condition
from ..flux.generate import generate
from ..flux.pipeline import FluxPipeline
from . import search
from .utils import TORCH_DTYPE_MAP, get_noises

MAX_SEED = 2 ** 31 - 1


def candidate_seeds(prompt_index: int, search_round: int, n: int, base: int = 0) -> List[int]:
    g = torch.Generator().manual_seed((base * 1000003 + prompt_index * 9176 + search_round * 131) % MAX_SEED)
    return torch.randint(0, MAX_SEED, (n,), generator=g).tolist()


def build_pipeline(config: dict, device, synthetic: bool = False, small: bool = False) -> FluxPipeline:
    pa = config["pipeline_args"]
    dtype = TORCH_DTYPE_MAP[pa.get("torch_dtype", "bf16")]
    if synthetic:
        cfg = dict(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=256,
                   pooled_projection_dim=64) if small else None
        pipe = FluxPipeline.synthetic(cfg, seed=0, torch_dtype=dtype, device=device)
    else:
        pipe = FluxPipeline.from_pretrained(pa["pretrained_model_name_or_path"], torch_dtype=dtype,
                                            cache_dir=pa.get("cache_dir") or None).to(device)
    pipe.set_progress_bar_config(disable=True)
    if pa.get("lora_path"):
        pipe.load_lora_weights(pa["lora_path"], adapter_name="reflection")
    return pipe


def candidate_condition(pipe: FluxPipeline, latents: torch.Tensor, height: int, width: int, condition_size: int) -> Condition:
    """The condition a selected candidate becomes for the next round.  With a VAE on the pipeline this is the
    REFERENCE computation (tts_reflectionflow.py:273-279): decode the candidate (generate.py:302-307), take the 8-bit
    image the reference would have written to PNG and re-opened, `resize((condition_size, condition_size))` with PIL's
    default filter, and hand it over as `Condition(condition=img, "cot", position_delta=[0, -condition_size // 16])` --
    `Condition.encode` then runs `image_processor.preprocess` + `vae.encode(x).latent_dist.sample()` + shift/scale +
    pack (pipeline_tools.py:7-30).  Only the PNG file round trip itself is skipped (lossless).
    Without a VAE (no weights offline) it falls back to `latent_to_condition`, a declared stand-in."""
    if pipe.vae is None or pipe.image_processor is None:
        return latent_to_condition(latents, height, width, condition_size)
    z = pipe._unpack_latents(latents, height, width, pipe.vae_scale_factor)
    z = z / pipe.vae.config.scaling_factor + pipe.vae.config.shift_factor
    img = pipe.image_processor.postprocess(pipe.vae.decode(z.to(pipe.vae.dtype), return_dict=False)[0], output_type="pil")[0]
    img = img.resize((condition_size, condition_size))
    return Condition(condition=img, condition_type="cot", position_delta=[0, -condition_size // 16])   # as the reference: -size // 16


@torch.no_grad()
def decode_candidates(pipe: FluxPipeline, latents: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """Verifier input without a PNG round trip (the reference saves every candidate as PNG and the verifier
    re-opens it one by one, tts_reflectionflow.py:145,160,328-332): packed latents [N, S, 64] of a round's candidates ->
    images [N, 3, H, W] in [0, 1] ON THE DEVICE, decoded by the pipeline's VAE (the HIP path after pipe.enable_hip_vae()).
    A batched on-device verifier (`score_batch` contract of tts/search.py) takes this tensor directly."""
    if pipe.vae is None:
        raise ValueError("decode_candidates(): this pipeline has no VAE")
    lat = latents if latents.dim() == 3 else latents.reshape(-1, latents.shape[-2], latents.shape[-1])
    z = pipe._unpack_latents(lat, height, width, pipe.vae_scale_factor)
    z = z / pipe.vae.config.scaling_factor + pipe.vae.config.shift_factor
    img = pipe.vae.decode(z.to(pipe.vae.dtype), return_dict=False)[0]
    return (img.float() / 2 + 0.5).clamp(0, 1)


def latent_to_condition(latents: torch.Tensor, height: int, width: int, condition_size: int) -> Condition:
    """STAND-IN (used only when the pipeline has no VAE) for `resize(decoded image, condition_size) -> VAE encode`
    (tts_reflectionflow.py:273-279 + condition.py:96-132): area-downsample the candidate's latent grid to the
    condition resolution in latent space.  Position ids follow the reference: delta = [0, -size//16]."""
    from ..flux.pipeline import FluxPipeline as P
    z = P._unpack_latents(latents, height, width, 8).float()                   # [1,16,h,w]
    hc = 2 * (condition_size // 16)
    z = torch.nn.functional.interpolate(z, size=(hc, hc), mode="area")
    tokens = P._pack_latents(z, 1, 16, hc, hc).to(latents.dtype)
    ids = P._prepare_latent_image_ids(1, hc // 2, hc // 2, latents.device, latents.dtype)
    return Condition("cot", tokens=tokens, ids=ids, position_delta=[0, -condition_size // 16])


def _save(path: str, latents: torch.Tensor, pipe: FluxPipeline, height: int, width: int):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if pipe.vae is not None and pipe.image_processor is not None:              # full pipeline available: PNG like the reference
        z = pipe._unpack_latents(latents, height, width, pipe.vae_scale_factor)
        z = z / pipe.vae.config.scaling_factor + pipe.vae.config.shift_factor
        img = pipe.image_processor.postprocess(pipe.vae.decode(z, return_dict=False)[0], output_type="pil")[0]
        img.save(path + ".png")
    else:
        torch.save(latents.cpu(), path + ".pt")


def run_noise_scaling(config: dict, prompts: List[str], output_dir: str, pipe: FluxPipeline, shard: search.Shard,
                      start_index: int = 0) -> List[dict]:
    pa, sa = config["pipeline_args"], config["search_args"]
    dev, dtype = pipe.device, pipe.dtype
    out = []
    for index, prompt in enumerate(prompts):
        sample_dir = os.path.join(output_dir, f"{index + start_index:0>5}", "samples")
        for rnd in range(1, sa["search_rounds"] + 1):
            seeds = candidate_seeds(index + start_index, rnd, sa["search_branch"])
            for i in shard.mine(len(seeds)):
                noise = get_noises(MAX_SEED, 1, pa["height"], pa["width"], device=dev, dtype=dtype, seeds=[seeds[i]])[seeds[i]]
                lat = pipe(prompt=[prompt], latents=noise, guidance_scale=pa["guidance_scale"],
                           num_inference_steps=pa["num_inference_steps"], height=pa["height"], width=pa["width"],
                           output_type="latent").images
                _save(os.path.join(sample_dir, f"{rnd}_round@{seeds[i]}"), lat, pipe, pa["height"], pa["width"])
            out.append({"prompt": prompt, "search_round": rnd, "num_noises": len(seeds)})
    if shard.world_size > 1:
        torch.distributed.barrier()
    return out


def run_reflection_search(config: dict, prompts: List[str], output_dir: str, pipe: FluxPipeline, shard: search.Shard,
                          start_index: int = 0, verifier=None, score_batch=None, refine_prompt=None) -> List[dict]:
    """Reflection rounds.  Deliberate deviation from tts_reflectionflow.py:314-322: the reference's `generate`
    call passes neither `latents`, `num_inference_steps` nor `guidance_scale` (so its defaults -- 28 steps,
    guidance 3.5, noise from the global RNG -- apply and the `get_noises` seeds only name files);
    here the config's steps/guidance and the per-candidate seeded noise ARE passed, so that a candidate is a pure
    function of (prompt, round, index) and results do not depend on the world size.  Each prompt's
    `search_log.jsonl` holds that prompt's rounds only; the return value is the concatenation over prompts.

    `refine_prompt(prompt, round, candidate_index, seed) -> str` is the hook of the reference's reflection / prompt-refinement LLMs
    (tts_reflectionflow.py:286-294: `refined + " [Reflexion]: " + reflection`, a different prompt per candidate and round; the LLMs
    themselves are out of scope).  The prompts of a rank's candidates are encoded ONCE per round, de-duplicated, in one
    batched `encode_prompt` call (T5-XXL + CLIP-L share their GEMM launches across the prompts), and handed to `generate` as
    embeddings -- the reference encodes inside every generate() call."""
    pa, sa, model_cfg = config["pipeline_args"], config["search_args"], config.get("model", {})
    dev, dtype = pipe.device, pipe.dtype
    N, topk = sa["search_branch"], max(1, sa.get("topk", 1))
    all_logs: List[dict] = []
    for index, prompt in enumerate(prompts):
        pdir = os.path.join(output_dir, f"{index + start_index:0>5}")
        log: List[dict] = []                                                    # this prompt's rounds only
        kept: List[torch.Tensor] = []                                           # selected latents, identical on every rank
        for rnd in range(0, sa["search_rounds"] + 1):
            seeds = candidate_seeds(index + start_index, rnd, N)

            cond_of_kept: Dict[int, Condition] = {}   # one decode -> resize per kept latent and round, shared by its candidates
            # this rank's prompts of the round: one batched text-encoder call over the distinct ones
            mine = list(shard.mine(len(seeds)))
            prompt_of = {i: (refine_prompt(prompt, rnd, i, seeds[i]) if refine_prompt is not None else prompt) for i in mine}
            uniq = list(dict.fromkeys(prompt_of.values()))
            if uniq:
                pe_all, pooled_all, _ = pipe.encode_prompt(prompt=uniq, max_sequence_length=pa.get("max_sequence_length", 512))
            slot = {p: k for k, p in enumerate(uniq)}

            def gen(i, seed):
                noise = get_noises(MAX_SEED, 1, pa["height"], pa["width"], device=dev, dtype=dtype, seeds=[seed])[seed]
                conds = None
                if rnd > 0:                                                     # round 0 = plain t2i (noise scaling)
                    j = i % len(kept)
                    if j not in cond_of_kept:
                        cond_of_kept[j] = candidate_condition(pipe, kept[j], pa["height"], pa["width"], pa["condition_size"])
                    # Condition.encode samples the VAE posterior (pipeline_tools.py:10; the reference draws from the global
                    # RNG): a per-candidate generator makes the sample a function of the candidate's seed alone --
                    # world-size independent, and the search loop leaves the global CPU / GPU RNG state untouched
                    conds = [cond_of_kept[j].with_generator(torch.Generator(device="cpu").manual_seed(seed))]
                k = slot[prompt_of[i]]
                lat = generate(pipe, prompt_embeds=pe_all[k:k + 1], pooled_prompt_embeds=pooled_all[k:k + 1], conditions=conds,
                               height=pa["height"], width=pa["width"], max_sequence_length=pe_all.shape[1],
                               num_inference_steps=pa["num_inference_steps"], guidance_scale=pa["guidance_scale"],
                               latents=noise, model_config=model_cfg, default_lora=True, output_type="latent").images
                _save(os.path.join(pdir, "samples", f"{rnd}_round@{seed}"), lat, pipe, pa["height"], pa["width"])
                return lat

            sel, scores, local = search.run_round(shard, seeds, gen, verifier, topk=topk, score_batch=score_batch)
            # hand the selected latents to every rank: ONE all-gather (topk x 512 KiB per rank at 1024^2; the reference hands
            # PNG paths over)
            like = torch.empty(1, (pa["height"] // 16) * (pa["width"] // 16), 64, device=dev, dtype=dtype)
            kept = search.allgather_selected_latents(shard, sel, local, like)
            log.append({"prompt": prompt, "round": rnd, "seeds": seeds, "scores": scores, "selected": sel})
        if shard.rank == 0:
            os.makedirs(pdir, exist_ok=True)
            with open(os.path.join(pdir, "search_log.jsonl"), "w") as f:
                for r in log:
                    f.write(json.dumps(r) + "\n")
        all_logs += log
    return all_logs
