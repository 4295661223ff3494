"""Candidate-parallel search runner shared by the two driver scripts.

Mirrors the control flow of the reference drivers:
  * noise scaling  (tts/tts_t2i_noise_scaling.py:126-159, sample :16-77): per prompt x round draw
    `search_branch` seeded noises and generate one candidate per noise -- no dependency between rounds; the
    output tree is what the reflection driver's `--imgpath` reads;
  * reflection loop (tts/tts_reflectionflow.py:591-629, sample :94-463): order the previous round's images by the
    verifier's key, keep topk = search_branch, generate candidate i from the i-th best parent as a "cot" Condition
    with the FLUX-Corrector LoRA active on the condition tokens, re-score, file the candidates into chains, write
    the best-of-chain / best-overall artefacts.  Pinned by tests/golden/search_tree.json, recorded from the
    reference's own main().
What is NOT here (outside the hot path): the GPT-4o / NVILA verifiers and
the reflection / prompt-refinement LLMs.  They plug in through `score_batch` / `score_images` and the
`reflect` / `refine` hooks; `search.stub_score_batch` stands in for the verifier's output contract.

Sharding: candidate i of a round runs on rank i % world_size; the collectives are the all-gather of {score, label}
records and one owners-only all-gather of the parents' latents per round (tts/search.py).  Seeds are a pure
function of (prompt index, round, candidate), so results do not depend on the world size.
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
        # The search is inference with ONE static LoRA on a 288 GB part: fold it into per-token-group weight copies (+11 GB for
        # FLUX.1-dev at r = 32) instead of paying two low-rank launches per LoRA'd linear and step -- the reference's enable_lora gating
        # is kept (only the token groups LoRA acts on multiply by the merged copy), parity vs the reference fixture 7.40e-3 (K-segment
        # form 7.42e-3), cfg4 0.460 -> 0.469 of peak (DESIGN K1 "LoRA without launches").  `"merged_lora": false` in pipeline_args keeps
        # the K-segment form (what training and adapter hot-swapping need).
        if pa.get("merged_lora", True) and next(pipe.transformer.parameters()).is_cuda:
            pipe.enable_merged_lora()
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


def _save(path: str, latents: torch.Tensor, pipe: FluxPipeline, height: int, width: int) -> str:
    """Write one candidate: PNG like the reference when the pipeline has a VAE, else the packed latents as .pt.  Returns the file."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if pipe.vae is not None and pipe.image_processor is not None:              # full pipeline available: PNG like the reference
        z = pipe._unpack_latents(latents, height, width, pipe.vae_scale_factor)
        z = z / pipe.vae.config.scaling_factor + pipe.vae.config.shift_factor
        img = pipe.image_processor.postprocess(pipe.vae.decode(z, return_dict=False)[0], output_type="pil")[0]
        img.save(path + ".png")
        return path + ".png"
    torch.save(latents.cpu(), path + ".pt")
    return path + ".pt"


def run_noise_scaling(config: dict, prompts: List[str], output_dir: str, pipe: FluxPipeline, shard: search.Shard,
                      start_index: int = 0, metadatas: Optional[List[dict]] = None, seeds_fn=None) -> List[dict]:
    """The plain t2i candidate loop, tts_t2i_noise_scaling.py:126-159 (`main`'s prompt x round loop) + :16-77 (`sample`),
    sharded over ranks; BASELINE cfg1 / cfg3 are quoted on it.  Pinned by `tests/golden/noise_scaling.json`, recorded
    from the reference's own main(): per prompt `<index + start_index:05>/metadata.jsonl` (`json.dump` of the meta line,
    :136-137) and `<index>/samples/` (:132-133); per round `search_branch` seeded noises (:141-148), one stock `pipe(...)`
    call per candidate with prompt / latents / guidance_scale / num_inference_steps / height / width and nothing else
    (:60), the candidate written as `samples/<round>_round@<seed>.png` (:47-49, :70), and the datapoint
    {prompt, search_round, num_noises} (:72-77) -- the tree the reflection driver's `--imgpath` reads.

    Deviations, deliberate: candidate i of a round runs on rank i % world_size (the reference: a serial loop on one GPU);
    its seed is `seeds_fn(prompt index, round, N)[i]` -- by default `candidate_seeds`, a pure function of the indices, where
    the reference draws `torch.randint` from the global RNG (:141, utils.py:141) -- so the tree does not depend on the world
    size; `batch_size_for_img_gen` > 1 (:37-58: b candidates in one pipe call) still means one call per candidate here
    (the engine runs batch 1; same prompts, noises and files); the pipe is asked for latents and `_save` decodes them (PNG
    with a VAE on the pipeline, the packed latents as `.pt` without one); `use_low_gpu_vram` (:51-52, :62-63: the pipeline
    hops between CPU and GPU around every batch) is ignored: 24 GB of weights stay resident in 288 GB."""
    pa, sa = config["pipeline_args"], config["search_args"]
    dev, dtype = pipe.device, pipe.dtype
    seeds_fn = seeds_fn or candidate_seeds
    out = []
    for index, prompt in enumerate(prompts):
        outpath = os.path.join(output_dir, f"{index + start_index:0>5}")
        sample_dir = os.path.join(outpath, "samples")
        if shard.rank == 0:
            os.makedirs(sample_dir, exist_ok=True)
            with open(os.path.join(outpath, "metadata.jsonl"), "w") as fp:       # tts_t2i_noise_scaling.py:136-137
                json.dump(metadatas[index] if metadatas is not None else {"prompt": prompt}, fp)
        if shard.world_size > 1:
            torch.distributed.barrier()                                         # rank 0 made the directories every rank writes into
        for rnd in range(1, sa["search_rounds"] + 1):
            seeds = [int(s) for s in seeds_fn(index + start_index, rnd, sa["search_branch"])]
            if len(set(seeds)) != len(seeds):                                   # utils.py:153 (`assert len(noises) == len(seeds)`)
                raise ValueError(f"prompt {index + start_index} round {rnd}: duplicate candidate seeds {seeds}")
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


# ------------------------------------------------------------------------------------------------ reflection search
def read_imgpath(imgpath: str) -> List[dict]:
    """The `--imgpath` pool reader of tts_reflectionflow.py:535-556: one sub-folder per prompt with `metadata.jsonl`
    (first line = {"prompt", "tag", ...}) and `samples/` (every file, sorted, is a pool image)."""
    out = []
    for folder in sorted(os.listdir(imgpath)):
        fp = os.path.join(imgpath, folder)
        if not os.path.isdir(fp):
            continue
        with open(os.path.join(fp, "metadata.jsonl")) as f:
            meta = [json.loads(line) for line in f if line.strip()]
        sp = os.path.join(fp, "samples")
        images = [os.path.join(sp, x) for x in sorted(os.listdir(sp))] if os.path.exists(sp) else []
        out.append({"metadata": meta, "images": images})
    return out


def stub_score_images(paths: List[str], prompt: str):
    """Stand-in verifier for pool images that come from disk (`--imgpath`): a score from the file's bytes."""
    import hashlib
    sc = [int(hashlib.sha256(open(p, "rb").read()).hexdigest()[:8], 16) / 0xFFFFFFFF for p in paths]
    return torch.tensor(sc, dtype=torch.float32), torch.tensor([int(v >= 0.5) for v in sc], dtype=torch.int32)


def _bcast(shard: search.Shard, obj):
    """rank 0's Python object on every rank (the LLM hooks run once, on rank 0)."""
    if shard.world_size == 1:
        return obj
    box = [obj if shard.rank == 0 else None]
    torch.distributed.broadcast_object_list(box, src=0)
    return box[0]


def _copy(src: str, dst_stem: str):
    import shutil
    os.makedirs(os.path.dirname(dst_stem), exist_ok=True)
    shutil.copyfile(src, dst_stem + os.path.splitext(src)[1])


def run_reflection_search(config: dict, prompts: Optional[List[str]], output_dir: str, pipe: FluxPipeline, shard: search.Shard,
                          start_index: int = 0, verifier=None, score_batch=None, reflect=None, refine=None,
                          imgpath: Optional[str] = None, score_images=None, metadatas: Optional[List[dict]] = None) -> List[dict]:
    """The reflection search of tts_reflectionflow.py:591-629 (`main`'s round loop) + :94-463 (`sample`), sharded over ranks.

    Per prompt: a POOL of images -- the `--imgpath` folder's `samples/` when `imgpath` is given (as the reference), else
    `search_branch` plain t2i candidates generated here as "round 0" (= one round of tts_t2i_noise_scaling) -- then rounds
    1..search_rounds, each exactly the reference's: order the pool by the verifier's key, keep `topk = search_branch`
    (:609; `search_args.topk` overrides), pad by repetition (:179-182), write `best_img_detailedscore.jsonl`; run the
    reflection / prompt-refinement hooks (:196-259) and write `best_img_meta.jsonl`; generate candidate i from PARENT
    selected[i] -- its image resized to `condition_size` as a "cot" Condition with position_delta [0, -size // 16]
    (:273-279) -- under prompt i (`refined[i] + " [Reflexion]: " + reflection[i]`, :286-294); score the candidates; file
    them into chains (:358-395); write `midimg/<round>_round@<seed>`, `samples_lastround/`, `samples_path_bestround/` (best
    of each chain) and, after the last round, `samples_best/` (:397-444).  The new candidates are the next pool.
    `tests/golden/search_tree.json` (recorded from the reference's own main()) pins all of that.

    Hooks (the LLMs are out of scope; both run on rank 0 and their strings are broadcast):
      reflect(ctx) -> List[str]         ctx = dict(selected=[{"image_name", "score", "label", "path" | "latents"}...] in
                                        selection order, original_prompt, current_prompt (list), reflections (list, "" at
                                        round 1), evaluations (JSON strings), search_round)
      refine(ctx, reflections) -> List[str]
    `score_batch(latents [n, S, 64], seeds) -> (f32 [n], i32 [n])` scores generated candidates on the device (one call per
    round and rank); `score_images(paths, prompt) -> (f32, i32)` scores an `--imgpath` pool.

    Deliberate deviations: the reference's `generate` call passes neither `latents`,
    `num_inference_steps` nor `guidance_scale` (so 28 steps / 3.5 / global-RNG noise apply and the `get_noises` seeds only
    name files); here the config's steps / guidance and the per-candidate seeded noise ARE passed, so a candidate is a pure
    function of (prompt, round, index) and results do not depend on the world size.  A candidate is scored ONCE (the
    reference scores it after generation and again as next round's pool, :337-356 / :157-170 -- the same numbers).  The
    parent hand-off is one all-gather of latents instead of PNG paths.  `search_log.jsonl` is an addition."""
    pa, sa, model_cfg = config["pipeline_args"], config["search_args"], config.get("model", {})
    dev, dtype = pipe.device, pipe.dtype
    N, R = sa["search_branch"], sa["search_rounds"]
    topk = int(sa.get("topk") or N)                                              # tts_reflectionflow.py:609
    kind = config.get("verifier_args", {}).get("name", "nvila")
    kind = kind if kind in search.SORT_KEYS else "nvila"
    H, W, csize = pa["height"], pa["width"], pa["condition_size"]
    if hasattr(pipe.vae, "check_geometry"):          # HIP VAE: refuse unsupported sizes before any candidate is generated
        pipe.vae.check_geometry(H, W, "candidate image")
        pipe.vae.check_geometry(csize, csize, "condition image")
    use_reflection = reflect is not None
    use_refine = refine is not None
    if use_reflection and not use_refine:
        refine = lambda ctx, refl: list(ctx["current_prompt"])                  # noqa: E731  (the reference needs both, :288-290)
    pools = read_imgpath(imgpath) if imgpath else None
    if pools is not None:
        end = config.get("end_index", -1)
        pools = pools[start_index:] if end in (-1, None) else pools[start_index:end]
        prompts = [p["metadata"][0]["prompt"] for p in pools]
        metadatas = [p["metadata"][0] for p in pools]
    all_logs: List[dict] = []
    like = torch.empty(1, (H // 16) * (W // 16), 64, device=dev, dtype=dtype)
    for index, prompt in enumerate(prompts):
        outpath = os.path.join(output_dir, f"{index + start_index:0>5}")
        dirs = {k: os.path.join(outpath, k) for k in ("samples_lastround", "samples_best", "samples_path_bestround", "midimg")}
        if shard.rank == 0:
            for d in dirs.values():
                os.makedirs(d, exist_ok=True)
            with open(os.path.join(outpath, "metadata.jsonl"), "w") as fp:       # :576-577
                json.dump(metadatas[index] if metadatas is not None else {"prompt": prompt}, fp)
            for f in ("best_img_detailedscore.jsonl", "best_img_meta.jsonl"):    # appended to per round
                if os.path.exists(os.path.join(outpath, f)):
                    os.remove(os.path.join(outpath, f))
        if shard.world_size > 1:
            torch.distributed.barrier()
            # every rank writes its candidates' files under `outpath` and rank 0 copies the round's artefacts from there
            # (samples_lastround / samples_path_bestround / samples_best): all ranks must see ONE output_dir -- one node, or a shared
            # filesystem.  Checked before any work: a node-local output_dir would otherwise fail on rank 0 mid-round while the other
            # ranks wait in the next collective.
            seen = torch.tensor([1 if os.path.exists(os.path.join(outpath, "metadata.jsonl")) else 0], dtype=torch.int32,
                                device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(seen, op=torch.distributed.ReduceOp.MIN)
            if int(seen.item()) == 0:
                raise RuntimeError(f"output_dir {output_dir!r} is not shared by all {shard.world_size} ranks (rank 0's metadata.jsonl is not visible "
                                   "everywhere): run on one node or point --output_dir at a shared filesystem")
        log: List[dict] = []
        tree = search.ReflectionTree(kind)
        updated_prompt = [prompt] * N                                           # :579
        reflections = [""] * N if use_reflection else None                      # :582-585

        def generate_round(rnd, parents, parent_payload, round_prompts, subdir):
            """This rank's candidates of one round -> (seeds, names, files, scores [(f, i)] of ALL candidates, local latents)."""
            seeds = candidate_seeds(index + start_index, rnd, N)
            mine = list(shard.mine(N))
            uniq = list(dict.fromkeys(round_prompts[i] for i in mine))
            if uniq:                                                            # one batched text-encoder call per round and rank
                pe_all, pooled_all, _ = pipe.encode_prompt(prompt=uniq, max_sequence_length=pa.get("max_sequence_length", 512))
            slot = {p: k for k, p in enumerate(uniq)}
            cond_cache: Dict[int, Condition] = {}                               # one decode -> resize per parent and rank
            local: Dict[int, torch.Tensor] = {}
            for i in mine:
                seed = seeds[i]
                noise = get_noises(MAX_SEED, 1, H, W, device=dev, dtype=dtype, seeds=[seed])[seed]
                conds = None
                if parents is not None:
                    conds = []                                                  # beyond the padded selection: no condition (:307)
                    if i < len(parents):
                        j = parents[i]
                        if j not in cond_cache:
                            cond_cache[j] = _payload_condition(pipe, parent_payload[j], H, W, csize)
                        # Condition.encode samples the VAE posterior (pipeline_tools.py:10; the reference draws from the global
                        # RNG): a per-candidate generator makes the sample a function of the candidate's seed alone
                        conds = [cond_cache[j].with_generator(torch.Generator(device="cpu").manual_seed(seed))]
                k = slot[round_prompts[i]]
                lat = generate(pipe, prompt_embeds=pe_all[k:k + 1], pooled_prompt_embeds=pooled_all[k:k + 1], conditions=conds,
                               height=H, width=W, max_sequence_length=pe_all.shape[1],
                               num_inference_steps=pa["num_inference_steps"], guidance_scale=pa["guidance_scale"],
                               latents=noise, model_config=model_cfg, default_lora=True, output_type="latent").images
                _save(os.path.join(outpath, subdir, f"{rnd}_round@{seed}"), lat, pipe, H, W)
                local[i] = lat
            scores = _score_local(shard, N, mine, local, seeds, verifier, score_batch)
            ext = ".png" if (pipe.vae is not None and pipe.image_processor is not None) else ".pt"
            names = [f"{subdir}/{rnd}_round@{s}{ext}" for s in seeds]
            return seeds, names, scores, local

        # ---- the pool
        if pools is not None:
            pool_names = list(pools[index]["images"])
            mine = shard.mine(len(pool_names))
            fn = score_images or stub_score_images
            sc, lab = fn([pool_names[i] for i in mine], prompt) if mine else (torch.empty(0), torch.empty(0, dtype=torch.int32))
            s_all, l_all = search.allgather_score_tensors(shard, len(pool_names), sc, lab)
            pool_scores = [(float(a), int(b)) for a, b in zip(s_all.tolist(), l_all.tolist())]
            pool_local: Dict[int, object] = {i: p for i, p in enumerate(pool_names)}      # files: every rank can read them
            pool_on_disk = True
            log.append({"prompt": prompt, "round": 0, "pool": pool_names, "scores": pool_scores})
        else:
            seeds0, pool_names, pool_scores, pool_local = generate_round(0, None, None, [prompt] * N, "samples")
            pool_on_disk = False
            log.append({"prompt": prompt, "round": 0, "seeds": seeds0, "generated": pool_names, "scores": pool_scores})

        for rnd in range(1, R + 1):
            # ---- selection (:157-182) + its artefact (:185-190)
            sel = tree.select(pool_scores, topk)
            selected_names = [pool_names[j] for j in sel]
            evaluation = [_evaluation(pool_names[j], pool_scores[j], tree.scalar) for j in sel]
            if shard.rank == 0:
                with open(os.path.join(outpath, "best_img_detailedscore.jsonl"), "a") as f:
                    f.write(json.dumps({"evaluation": evaluation, "filenames_batch": selected_names}) + "\n")
            # ---- the parents' payloads on every rank: ONE owners-only all-gather (pool images on disk need none)
            if pool_on_disk:
                payload = {j: pool_local[j] for j in set(sel)}
            else:
                kept = search.allgather_selected_latents(shard, sel, pool_local, like)
                payload = {j: kept[k] for k, j in enumerate(sel)}
            # ---- reflection / refinement hooks (:196-259) on rank 0, strings broadcast; best_img_meta.jsonl (:262-271)
            round_prompts = [prompt] * N                                        # :295
            new_reflections = refined = None
            if use_reflection or use_refine:
                if shard.rank == 0:
                    # (round-1 parents read from --imgpath are named relative to the CWD or absolutely, generated ones relative to outpath)
                    ctx = dict(selected=[dict(e, **({"path": n if (pool_on_disk or os.path.isabs(n)) else os.path.join(outpath, n)}),
                                              latents=None if pool_on_disk else payload[j])
                                         for e, n, j in zip(evaluation, selected_names, sel)],
                               original_prompt=prompt, current_prompt=list(updated_prompt), reflections=reflections,
                               evaluations=[json.dumps(e) for e in evaluation], search_round=rnd)
                    new_reflections = list(reflect(ctx)) if use_reflection else None
                    refined = list(refine(ctx, new_reflections))
                    with open(os.path.join(outpath, "best_img_meta.jsonl"), "a") as f:
                        if use_reflection:
                            f.write(f"reflections{rnd}: " + json.dumps(new_reflections) + "\n")
                        if use_refine:
                            f.write(f"refined_prompt{rnd}: " + json.dumps(refined) + "\n")
                        f.write(f"filenames_batch{rnd}: " + json.dumps(selected_names) + "\n")
                new_reflections, refined = _bcast(shard, (new_reflections, refined))
                if use_reflection:                                              # :286-294
                    round_prompts = ([refined[i] + " [Reflexion]: " + new_reflections[i] for i in range(len(new_reflections))]
                                     if new_reflections else list(refined))
                    round_prompts += [prompt] * (N - len(round_prompts))
            # ---- generation + scoring (:297-356)
            seeds, names, scores, local = generate_round(rnd, sel, payload, round_prompts, "midimg")
            parents = [selected_names[i] if i < len(sel) else None for i in range(N)]
            tree.record(rnd, names, scores, parents)                            # :358-395
            best_chain = tree.best_of_chains()
            # ---- artefacts (:397-444), rank 0, after every rank's files exist (the score all-gather ordered the writes on one node; the
            # barrier makes that explicit for a shared filesystem, whose visibility the collective does not order)
            if shard.world_size > 1:
                torch.distributed.barrier()
            if shard.rank == 0:
                if rnd == R:
                    for i, n in enumerate(names):
                        _copy(os.path.join(outpath, n), os.path.join(dirs["samples_lastround"], f"{i:05}"))
                for i, n in enumerate(names if rnd == 1 else best_chain):
                    _copy(os.path.join(outpath, n), os.path.join(dirs["samples_path_bestround"], f"{i:05}"))
                if rnd == R:                                                    # the reference names this file by a leftover loop index
                    last_i = (len(names) if rnd == 1 else len(best_chain)) - 1
                    _copy(os.path.join(outpath, tree.best_overall()), os.path.join(dirs["samples_best"], f"{last_i:05}"))
            rec = {"prompt": prompt, "round": rnd, "seeds": seeds, "scores": scores, "selected": sel,
                   "selected_names": selected_names, "parents": parents, "generated": names, "prompts": round_prompts,
                   "chains": tree.snapshot(), "best_of_chains": best_chain, "flag_terminated": rnd == R}
            if use_reflection:
                rec["reflections"] = new_reflections
                reflections = new_reflections                                   # :619-620
            if use_refine:
                rec["refined_prompt"] = refined
                updated_prompt = refined                                        # :621-622
            log.append(rec)
            pool_names, pool_scores, pool_local, pool_on_disk = names, scores, local, False     # :623
        if shard.rank == 0:
            with open(os.path.join(outpath, "search_log.jsonl"), "w") as f:
                for r in log:
                    f.write(json.dumps(r) + "\n")
        all_logs += log
    if shard.world_size > 1:
        torch.distributed.barrier()
    return all_logs


def _evaluation(name: str, score, scalar: bool) -> dict:
    """One verifier record in the reference's shape (:162-164 nvila; the scalar form keeps the metric name of refine_args)."""
    if scalar:
        return {"image_name": name, "overall_score": {"score": score[0]}}
    return {"image_name": name, "label": "yes" if score[1] == 1 else "no", "score": score[0]}


def _payload_condition(pipe: FluxPipeline, payload, height: int, width: int, condition_size: int) -> Condition:
    """A parent -> its "cot" Condition (tts_reflectionflow.py:273-279): an image file is opened and resized as the reference
    does; a candidate's packed latents go through `candidate_condition` (decode -> 8-bit image -> resize, or the latent stand-in)."""
    if isinstance(payload, str):
        if payload.endswith(".pt"):
            return candidate_condition(pipe, torch.load(payload).to(pipe.device, pipe.dtype), height, width, condition_size)
        from PIL import Image
        img = Image.open(payload).resize((condition_size, condition_size))
        return Condition(condition=img, condition_type="cot", position_delta=[0, -condition_size // 16])
    return candidate_condition(pipe, payload, height, width, condition_size)


def _score_local(shard, n, mine, local, seeds, verifier, score_batch):
    """This rank's candidates -> ONE batched verifier call -> ONE all-gather of {f32, i32} records -> all n (score, label)."""
    if verifier is not None and score_batch is None:
        res = [verifier(local[i], int(seeds[i])) for i in mine]
        sc = torch.tensor([r[0] for r in res], dtype=torch.float32)
        lab = torch.tensor([r[1] for r in res], dtype=torch.int32)
    elif mine:
        fn = score_batch or search.stub_score_batch
        sc, lab = fn(torch.stack([local[i].reshape(-1, local[i].shape[-1]) for i in mine]), [int(seeds[i]) for i in mine])[:2]
    else:
        sc, lab = torch.empty(0, dtype=torch.float32), torch.empty(0, dtype=torch.int32)
    s_all, l_all = search.allgather_score_tensors(shard, n, sc, lab)
    return [(float(a), int(b)) for a, b in zip(s_all.tolist(), l_all.tolist())]
