#!/usr/bin/env python3
"""Golden fixtures for the plain t2i candidate loop (BASELINE cfg1 / cfg3)  -- build container only.

Runs the REFERENCE's own `tts/tts_t2i_noise_scaling.py::main()` end to end (its config / CLI merge, meta slicing,
output tree, round loop, `sample()` batching and file naming) under stub namespaces for everything that is not the
loop: `diffusers.DiffusionPipeline` (a recording pipeline whose images encode a running id) and `utils` (`get_noises`
hands out fixed seeds and records its arguments; `parse_cli_args` returns the scenario's namespace).

Recorded per scenario (tests/golden/noise_scaling.json): the loader call, every `get_noises` call, every `pipe(...)` call
(keyword names, prompts, the seeds of the stacked latents, their shape, guidance / steps / height / width), the
datapoint each `sample()` returned, and the output tree (relative file names with the candidate each PNG holds,
`metadata.jsonl` text).  Nothing of the reference's source is stored: the fixture is inputs + observed outputs.

    python tests/golden/make_noise_scaling_golden.py
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import shutil
import sys
import tempfile
import types

import torch
from PIL import Image

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF_TTS = "/root/reference/tts"
STATE: dict = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _mk_img(i: int) -> Image.Image:
    return Image.new("RGB", (8, 8), (i % 256, i // 256, 11))


def _img_id(img: Image.Image) -> int:
    r, g, _b = img.convert("RGB").getpixel((0, 0))
    return r + 256 * g


class _Pipe:
    def to(self, where):
        STATE["to"].append(str(where))
        return self

    def set_progress_bar_config(self, **kw):
        STATE["progress_bar"] = kw

    def __call__(self, **kw):
        lat = kw["latents"]
        seeds = [int(v) for v in lat[:, 0, 0].tolist()]              # the stub noise carries its seed
        ids = []
        for s in seeds:
            iid = STATE["next_id"]
            STATE["next_id"] += 1
            STATE["id2cand"][iid] = STATE["seed2cand"][s]
            ids.append(iid)
        STATE["calls"].append(dict(
            kwargs=sorted(kw), prompt=list(kw["prompt"]), seeds=seeds, candidates=[STATE["seed2cand"][s] for s in seeds],
            latents_shape=list(lat.shape), guidance_scale=kw["guidance_scale"], num_inference_steps=kw["num_inference_steps"],
            height=kw["height"], width=kw["width"]))
        return types.SimpleNamespace(images=[_mk_img(i) for i in ids])


class _DiffusionPipeline:
    @staticmethod
    def from_pretrained(name, torch_dtype=None, cache_dir=None):
        STATE["from_pretrained"] = [name, str(torch_dtype), cache_dir]
        return _Pipe()


def _prep_fn(**_k):
    raise AssertionError("the stub get_noises never calls the latent prep function")


def _get_noises(max_seed, num_samples, height, width, device="cuda", dtype=None, fn=None):
    k = len(STATE["noise_calls"])
    per_prompt = STATE["rounds"]
    p, r = k // per_prompt, k % per_prompt + 1
    STATE["noise_calls"].append(dict(max_seed=int(max_seed), num_samples=num_samples, height=height, width=width, device=device,
                                     dtype=str(dtype), fn="prepare_latents_for_flux" if fn is _prep_fn else repr(fn)))
    out = {}
    for i in range(num_samples):
        seed = 100000 * (p + 1) + 1000 * r + 7 * i + 3
        STATE["seed2cand"][seed] = f"p{p}r{r}c{i}"
        out[seed] = torch.full((1, (height // 16) * (width // 16), 64), float(seed), dtype=torch.float64)
    return out


def install_stubs():
    _mod("diffusers", DiffusionPipeline=_DiffusionPipeline)
    _mod("utils", get_noises=_get_noises, TORCH_DTYPE_MAP={"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16},
         get_latent_prep_fn=lambda name: {"black-forest-labs/FLUX.1-dev": _prep_fn}[name], parse_cli_args=lambda: STATE["args"])


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_tts_t2i_noise_scaling", os.path.join(REF_TTS, "tts_t2i_noise_scaling.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


SCENARIOS = [
    dict(name="n4_r2_two_prompts", N=4, R=2, metas=2, height=1024, width=1024),
    dict(name="cfg3_n32_r1", N=32, R=1, metas=1, height=1024, width=1024),                     # BASELINE cfg3's round
    dict(name="cfg1_n1_256_fp32", N=1, R=1, metas=1, height=256, width=256, steps=4, dtype="fp32"),   # BASELINE cfg1
    dict(name="slice_1_to_3_of_5", N=2, R=1, metas=5, start=1, end=3, height=512, width=768),
    dict(name="slice_from_2", N=1, R=2, metas=4, start=2, height=512, width=512),
    dict(name="batched_generation_2", N=4, R=1, metas=1, batch=2, height=1024, width=1024),
    dict(name="batched_generation_3_ragged", N=5, R=1, metas=1, batch=3, height=1024, width=1024),
    dict(name="low_vram", N=2, R=1, metas=1, low_vram=True, height=1024, width=1024),
]


def run_scenario(ref, sc):
    tmp = tempfile.mkdtemp(prefix="rf_noise_golden_")
    try:
        metas = [{"prompt": "a photo of %s #%d" % (sc["name"], k), "tag": ["counting", "color"][k % 2], "include": [{"class": "cat", "count": k}]}
                 for k in range(sc["metas"])]
        meta_path = os.path.join(tmp, "meta.jsonl")
        with open(meta_path, "w") as f:
            for m in metas:
                f.write(json.dumps(m) + "\n")
        cfg = {
            "pipeline_args": {"pretrained_model_name_or_path": "black-forest-labs/FLUX.1-dev", "cache_dir": "C",
                              "torch_dtype": sc.get("dtype", "bf16"), "height": sc["height"], "width": sc["width"],
                              "guidance_scale": 3.5, "num_inference_steps": sc.get("steps", 50), "max_sequence_length": 512},
            "search_args": {"search_method": "random", "search_branch": sc["N"], "search_rounds": sc["R"]},
            "use_low_gpu_vram": bool(sc.get("low_vram")), "batch_size_for_img_gen": sc.get("batch", 1),
        }
        cfg_path = os.path.join(tmp, "cfg.json")
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        out = os.path.join(tmp, "out")
        STATE.clear()
        STATE.update(calls=[], noise_calls=[], seed2cand={}, id2cand={}, next_id=0, to=[], rounds=sc["R"])
        STATE["args"] = argparse.Namespace(pipeline_config_path=cfg_path, start_index=sc.get("start", 0), end_index=sc.get("end", -1),
                                           imgpath="", output_dir=out, meta_path=meta_path)
        os.environ["OPENAI_API_KEY"] = "none"
        datapoints = []
        orig_sample = ref.sample

        def recording_sample(**kw):
            dp = orig_sample(**kw)
            datapoints.append(dict(dp, _call=dict(prompts=list(kw["prompts"]), search_round=kw["search_round"], original_prompt=kw["original_prompt"],
                                                  midimg_path=os.path.relpath(kw["midimg_path"], out), seeds=[int(s) for s in kw["noises"]])))
            return dp
        ref.sample = recording_sample
        try:
            ref.main()
        finally:
            ref.sample = orig_sample
        tree = {}
        for root, _dirs, files in os.walk(out):
            for fn in sorted(files):
                rel = os.path.relpath(os.path.join(root, fn), out)
                if fn.endswith(".png"):
                    tree[rel] = STATE["id2cand"][_img_id(Image.open(os.path.join(root, fn)))]
                else:
                    tree[rel] = open(os.path.join(root, fn)).read()
        return dict(scenario=sc, metas=metas, config=cfg, from_pretrained=STATE["from_pretrained"], to=STATE["to"],
                    progress_bar=STATE["progress_bar"], noise_calls=STATE["noise_calls"], calls=STATE["calls"],
                    datapoints=datapoints, tree=dict(sorted(tree.items())),
                    seed_of={v: k for k, v in STATE["seed2cand"].items()})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    install_stubs()
    sys.path.insert(0, REF_TTS)
    ref = load_reference()
    assert ref.MAX_SEED == 2 ** 31 - 1
    out = {sc["name"]: run_scenario(ref, sc) for sc in SCENARIOS}
    path = os.path.join(HERE, "noise_scaling.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: (len(v["calls"]), len(v["tree"])) for k, v in out.items()})


if __name__ == "__main__":
    main()
