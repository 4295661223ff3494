#!/usr/bin/env python3
"""Golden fixtures for the candidate / round loop  -- build container only.

Runs the REFERENCE's own `tts/tts_reflectionflow.py::main()` end to end (its CLI parsing, its `--imgpath` pool reader,
its `sample()` selection / parent mapping / chain bookkeeping / artefact writers) under stub namespaces for everything
that is not the search loop: `diffusers` (pipeline loader), `openai`, the NVILA / OpenAI verifier classes, the two LLM
calls, `train_flux.flux.generate` / `Condition`, and `utils.get_noises`.  The stubs are deterministic:

  * a generated image is an 8x8 PNG whose pixel value encodes a running id; candidate i of round r gets the logical
    name "r{r}c{i}", image k of the --imgpath pool "init{k}";
  * the verifier's (label, score) for a name comes from a table written into the fixture;
  * the LLMs return strings that spell out their inputs, so the prompt plumbing across rounds is pinned too.

What is recorded per scenario (tests/golden/search_tree.json): the score table, and per round the selection (with the
repeat padding), the parent of each candidate, the prompt and condition handed to every generate() call, the chains,
best-of-chain, the final best, and the text of best_img_detailedscore.jsonl / best_img_meta.jsonl with paths replaced
by logical names.  Nothing of the reference's source is stored: the fixture is inputs + observed outputs.

    python tests/golden/make_search_golden.py
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import re
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


def _img_id(img: Image.Image) -> int:
    r, g, _b = img.convert("RGB").getpixel((0, 0))
    return r + 256 * g


def _mk_img(i: int) -> Image.Image:
    return Image.new("RGB", (8, 8), (i % 256, i // 256, 7))


def _name_of_id(i: int) -> str:
    return STATE["id2name"][i]


def _name_of_path(p: str) -> str:
    return _name_of_id(_img_id(Image.open(p)))


# ----------------------------------------------------------------------------------------------- stubs
class _Pipe:
    def to(self, *_a, **_k):
        return self

    def set_progress_bar_config(self, **_k):
        pass

    def load_lora_weights(self, path, adapter_name=None):
        STATE["lora"] = [path, adapter_name]


class _DiffusionPipeline:
    @staticmethod
    def from_pretrained(name, torch_dtype=None, cache_dir=None):
        STATE["from_pretrained"] = [name, str(torch_dtype), cache_dir]
        return _Pipe()


class _Nvila:
    """generate_content([image, prompt]) -> (answer, scores) with scores[0][0, token_id] = the verifier logit
    (call shape of tts_reflectionflow.py:159-163)."""

    def generate_content(self, content):
        img, prompt = content
        assert prompt == STATE["prompt"]
        label, score = STATE["table"][_name_of_id(_img_id(img))]
        t = torch.zeros(1, 4)
        t[0, 1 if label == "yes" else 2] = score
        STATE["verifier_calls"] += 1
        return label, [t]


class _OpenAIVerifier:
    def __init__(self, **paths):
        self.paths = paths

    # verifier role (name == "openai")
    def prepare_inputs(self, images, prompts):
        assert all(p == STATE["prompt"] for p in prompts)
        return [_name_of_id(_img_id(i)) for i in images]

    def score(self, inputs, tag=None, max_new_tokens=None):
        STATE["verifier_calls"] += len(inputs)
        return [{"image": n, "overall_score": {"score": STATE["table"][n][1], "explanation": "e(" + n + ")"}} for n in inputs]

    # refiner role
    def prepare_reflexion_prompt_inputs(self, images, original_prompt, current_prompt, reflections, evaluations):
        return [dict(image=_name_of_path(im), original=o, current=c, prev=r, ev=json.loads(e))
                for im, o, c, r, e in zip(images, original_prompt, current_prompt, reflections, evaluations)]

    def generate_reflections(self, inputs, max_new_tokens=None):
        return ["fix(%s|was:%s)" % (x["image"], x["prev"]) for x in inputs]

    def prepare_refine_prompt_inputs(self, images, original_prompt, current_prompt, reflections, evaluations=None):
        return [dict(image=_name_of_path(im), original=o, current=c, refl=r)
                for im, o, c, r in zip(images, original_prompt, current_prompt, reflections)]

    def refine_prompt(self, inputs):
        return ["P<%s;%s>" % (x["image"], x["current"][:24]) for x in inputs]


class _Chat:
    class completions:
        @staticmethod
        def create(messages, model):
            url = messages[1]["content"][0]["image_url"]["url"]
            txt = messages[1]["content"][1]["text"]
            assert STATE["prompt"] in txt
            msg = types.SimpleNamespace(content="qwen(%s)" % _name_of_path(url))
            return types.SimpleNamespace(choices=[types.SimpleNamespace(message=msg)])


class _OpenAI:
    def __init__(self, **_k):
        self.chat = _Chat


class _Condition:
    def __init__(self, condition_type=None, raw_img=None, condition=None, mask=None, position_delta=None):
        self.rec = dict(type=condition_type, image=_name_of_id(_img_id(condition)), size=list(condition.size),
                        position_delta=[int(v) for v in position_delta])


def _generate(pipe, prompt=None, conditions=None, height=None, width=None, model_config=None, default_lora=None, **kw):
    assert not kw, f"unexpected generate() kwargs {sorted(kw)}"      # no latents / steps / guidance are passed (:314-322)
    rnd, i = STATE["round"], STATE["cand"]
    name = f"r{rnd}c{i}"
    iid = STATE["next_id"]
    STATE["next_id"] += 1
    STATE["id2name"][iid] = name
    STATE["calls"].append(dict(round=rnd, candidate=i, prompt=list(prompt), conditions=[c.rec for c in conditions],
                               height=height, width=width, model_config=model_config, default_lora=default_lora))
    STATE["cand"] += 1
    return types.SimpleNamespace(images=[_mk_img(iid)])


def _get_noises(max_seed, num_samples, height, width, device="cuda", dtype=None, fn=None):
    STATE["round"] += 1
    STATE["cand"] = 0
    seeds = [1000 * STATE["round"] + 7 * k + 3 for k in range(num_samples)]
    STATE["seeds"].append(seeds)
    return {s: torch.zeros(1) for s in seeds}


def install_stubs():
    _mod("diffusers", DiffusionPipeline=_DiffusionPipeline)
    _mod("openai", OpenAI=_OpenAI)
    _mod("verifiers")
    _mod("verifiers.openai_verifier", OpenAIVerifier=_OpenAIVerifier)
    _mod("verifiers.nvila_verifier", load_model=lambda model_name, cache_dir: (_Nvila(), 1, 2))
    _mod("train_flux")
    _mod("train_flux.flux")
    _mod("train_flux.flux.generate", generate=_generate)
    _mod("train_flux.flux.condition", Condition=_Condition)
    _mod("utils", get_noises=_get_noises, TORCH_DTYPE_MAP={"bf16": torch.bfloat16, "fp32": torch.float32},
         get_latent_prep_fn=lambda name: None, parse_cli_args=lambda: STATE["args"])


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_tts_reflectionflow", os.path.join(REF_TTS, "tts_reflectionflow.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ----------------------------------------------------------------------------------------------- scenarios
def _table(names, seed, ties=False, all_no=False):
    g = torch.Generator().manual_seed(seed)
    t = {}
    for n in names:
        s = float(torch.rand((), generator=g))
        s = round(s, 1) if ties else round(s, 4)                  # one decimal -> many equal scores
        lab = "no" if all_no else ("yes" if float(torch.rand((), generator=g)) < 0.5 else "no")
        t[n] = [lab, s]
    return t


SCENARIOS = [
    dict(name="nvila_n4_r3", N=4, R=3, pool=4, verifier="nvila", seed=1),
    dict(name="nvila_pool2_pad", N=4, R=2, pool=2, verifier="nvila", seed=2),
    dict(name="nvila_pool6_ties", N=3, R=3, pool=6, verifier="nvila", seed=3, ties=True),
    dict(name="nvila_all_no", N=3, R=2, pool=3, verifier="nvila", seed=4, all_no=True),
    dict(name="nvila_n8_r3", N=8, R=3, pool=8, verifier="nvila", seed=5),
    dict(name="nvila_reflect_openai_refine", N=3, R=3, pool=3, verifier="nvila", seed=6, reflection="openai", refine=True),
    dict(name="nvila_reflect_qwen_refine", N=2, R=2, pool=2, verifier="nvila", seed=7, reflection="qwen", refine=True),
    dict(name="openai_n4_r3", N=4, R=3, pool=4, verifier="openai", seed=8),
    dict(name="openai_ties_pool3", N=4, R=2, pool=3, verifier="openai", seed=9, ties=True),
    dict(name="nvila_single_round", N=3, R=1, pool=3, verifier="nvila", seed=10),
]


def run_scenario(ref, sc):
    tmp = tempfile.mkdtemp(prefix="rf_search_golden_")
    try:
        N, R, pool = sc["N"], sc["R"], sc["pool"]
        prompt = "a photo of %s" % sc["name"]
        names = [f"init{k}" for k in range(pool)] + [f"r{r}c{i}" for r in range(1, R + 1) for i in range(N)]
        table = _table(names, sc["seed"], sc.get("ties", False), sc.get("all_no", False))
        STATE.clear()
        STATE.update(table=table, prompt=prompt, id2name={}, next_id=0, calls=[], seeds=[], round=0, cand=0, verifier_calls=0)
        imgpath = os.path.join(tmp, "img")
        samples = os.path.join(imgpath, "00000", "samples")
        os.makedirs(samples)
        for k in range(pool):
            STATE["id2name"][5000 + k] = f"init{k}"
            _mk_img(5000 + k).save(os.path.join(samples, f"{k:05}.png"))
        with open(os.path.join(imgpath, "00000", "metadata.jsonl"), "w") as f:
            f.write(json.dumps({"prompt": prompt, "tag": "counting"}) + "\n")
        cfg = {
            "pipeline_args": {"pretrained_model_name_or_path": "black-forest-labs/FLUX.1-dev", "cache_dir": "C", "torch_dtype": "bf16",
                              "height": 1024, "width": 1024, "condition_size": 512, "lora_path": "L"},
            "verifier_args": {"name": sc["verifier"], "model_name": "M", "cache_dir": "V", "refine_prompt_relpath": "a",
                              "reflexion_prompt_relpath": "b", "verifier_prompt_relpath": "c"},
            "refine_args": {"name": "openai", "choice_of_metric": "overall_score", "max_new_tokens": 1280,
                            "refine_prompt_relpath": "a", "reflexion_prompt_relpath": "b", "verifier_prompt_relpath": "c"},
            "search_args": {"search_method": "random", "search_branch": N, "search_rounds": R},
            "model": {"add_cond_attn": False, "latent_lora": False, "union_cond_attn": True},
            "reflection_args": {"run_reflection": bool(sc.get("reflection")), "name": sc.get("reflection") or "openai"},
            "prompt_refiner_args": {"run_refinement": bool(sc.get("refine"))},
            "use_low_gpu_vram": False, "batch_size_for_img_gen": 1,
        }
        cfg_path = os.path.join(tmp, "cfg.json")
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        out = os.path.join(tmp, "out")
        STATE["args"] = argparse.Namespace(pipeline_config_path=cfg_path, start_index=0, end_index=-1, imgpath=imgpath,
                                           output_dir=out, meta_path="unused")
        os.environ["OPENAI_API_KEY"] = "none"

        datapoints = []
        orig_sample = ref.sample

        def recording_sample(**kw):
            assert kw["topk"] == N                                   # tts_reflectionflow.py:609
            dp = orig_sample(**kw)
            datapoints.append(json.loads(json.dumps(dp)))            # chains is mutated in place: snapshot
            return dp
        ref.sample = recording_sample
        try:
            ref.main()
        finally:
            ref.sample = orig_sample

        pdir = os.path.join(out, "00000")

        def logical(text):                                           # file paths -> logical image names
            def sub(m):
                return _name_of_path(m.group(0))
            return re.sub(re.escape(tmp) + r"[^\"\s,\]]*?\.png", sub, text)

        def names_in(d):
            return {f: _name_of_path(os.path.join(pdir, d, f)) for f in sorted(os.listdir(os.path.join(pdir, d)))}

        rounds = []
        for r, dp in enumerate(datapoints, start=1):
            calls = [c for c in STATE["calls"] if c["round"] == r]
            rounds.append(dict(
                round=r, seeds=STATE["seeds"][r - 1],
                generated=[_name_of_path(p) for p in dp["generated_img"]],
                generated_files=[os.path.relpath(p, pdir) for p in dp["generated_img"]],
                parents=[c["conditions"][0]["image"] for c in calls],
                prompts=[c["prompt"] for c in calls],
                conditions=[c["conditions"] for c in calls],
                chains={_name_of_path(k): dict(images=[_name_of_path(p) for p in v["images"]], scores=v["scores"],
                                               **({"labels": v["labels"]} if "labels" in v else {}))
                        for k, v in dp["chains"].items()},
                flag_terminated=dp["flag_terminated"],
                refined_prompt=dp.get("refined_prompt"), reflections=dp.get("reflections"),
            ))
        c0 = STATE["calls"][0]
        res = dict(
            scenario={k: v for k, v in sc.items()}, prompt=prompt, table=table,
            generate_kwargs=dict(height=c0["height"], width=c0["width"], model_config=c0["model_config"], default_lora=c0["default_lora"]),
            lora=STATE.get("lora"), rounds=rounds,
            detailedscore=[json.loads(logical(line)) for line in open(os.path.join(pdir, "best_img_detailedscore.jsonl"))],
            meta=(logical(open(os.path.join(pdir, "best_img_meta.jsonl")).read()).splitlines()
                  if os.path.exists(os.path.join(pdir, "best_img_meta.jsonl")) else None),
            metadata_jsonl=open(os.path.join(pdir, "metadata.jsonl")).read(),
            midimg=names_in("midimg"), samples_lastround=names_in("samples_lastround"),
            samples_path_bestround=names_in("samples_path_bestround"), samples_best=names_in("samples_best"),
            verifier_calls=STATE["verifier_calls"],
        )
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    install_stubs()
    sys.path.insert(0, REF_TTS)
    ref = load_reference()
    out = {sc["name"]: run_scenario(ref, sc) for sc in SCENARIOS}
    path = os.path.join(HERE, "search_tree.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v["rounds"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
