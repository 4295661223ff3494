"""the candidate / round loop against fixtures RECORDED FROM THE REFERENCE's own `main()`
(tests/golden/make_search_golden.py ran tts/tts_reflectionflow.py end to end under stub verifier / LLM / generate
namespaces).  Each scenario replays the same pool, score table and hook behaviour through
`runner.run_reflection_search` and must reproduce, round by round: the selection incl. repeat padding, the parent of
every candidate, the prompt every candidate is generated under, the chains, best-of-chain, the final best, and the
best_img_detailedscore.jsonl / best_img_meta.jsonl artefacts -- at world size 1 and, over gloo, at world size 2."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reflectionflow_amd.tts import runner, search

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "search_tree.json")) as _f:
    GOLD = json.load(_f)


# --------------------------------------------------------------------------------------------- CPU stand-ins for the GPU work
def _fake_generate(pipe, prompt_embeds=None, conditions=None, latents=None, **kw):
    from reflectionflow_amd.flux.pipeline import FluxPipelineOutput
    x = latents.float() * 0.5 + prompt_embeds.float().mean()
    return FluxPipelineOutput(images=x.to(latents.dtype))


class _FakePipe:
    device, dtype, vae, image_processor = torch.device("cpu"), torch.float32, None, None

    def encode_prompt(self, prompt, max_sequence_length=512):
        v = torch.tensor([[float(sum(p.encode()) % 97)] for p in prompt])
        return v[:, None, :].expand(len(prompt), 8, 1).contiguous(), v, torch.zeros(8, 3)


class _Replay:
    """The scenario's verifier table and LLM stubs, spoken through this build's hooks."""

    def __init__(self, gold, tmp):
        from PIL import Image
        self.g = gold
        sc = gold["scenario"]
        self.N, self.R = sc["N"], sc["R"]
        self.table = gold["table"]
        self.logical = {}                                        # this build's image names -> the fixture's logical names
        self.imgpath = os.path.join(tmp, "img")
        samples = os.path.join(self.imgpath, "00000", "samples")
        os.makedirs(samples, exist_ok=True)
        for k in range(sc["pool"]):
            p = os.path.join(samples, f"{k:05}.png")
            if not os.path.exists(p):
                Image.new("RGB", (8, 8), (k, 0, 7)).save(p)
            self.logical[p] = f"init{k}"
        meta = os.path.join(self.imgpath, "00000", "metadata.jsonl")
        if not os.path.exists(meta):
            with open(meta, "w") as f:
                f.write(json.dumps({"prompt": gold["prompt"], "tag": "counting"}) + "\n")
        self.seed2name = {}
        for r in range(1, self.R + 1):
            for i, s in enumerate(runner.candidate_seeds(0, r, self.N)):
                self.seed2name[s] = f"r{r}c{i}"
                self.logical[f"midimg/{r}_round@{s}.pt"] = f"r{r}c{i}"
        self.config = {
            "pipeline_args": dict(height=64, width=64, condition_size=32, num_inference_steps=2, guidance_scale=3.5),
            "search_args": dict(search_branch=self.N, search_rounds=self.R),         # no "topk": the default must be N
            "verifier_args": {"name": sc["verifier"]}, "model": {"union_cond_attn": True},
        }

    def _lookup(self, names):
        lab = torch.tensor([1 if self.table[n][0] == "yes" else 0 for n in names], dtype=torch.int32)
        return torch.tensor([self.table[n][1] for n in names], dtype=torch.float32), lab

    def score_images(self, paths, prompt):
        assert prompt == self.g["prompt"]
        return self._lookup([self.logical[p] for p in paths])

    def score_batch(self, latents, seeds):
        assert latents.shape[0] == len(seeds)
        return self._lookup([self.seed2name[s] for s in seeds])

    def hooks(self):
        sc = self.g["scenario"]
        if not sc.get("reflection"):
            return None, None
        L = self.logical

        def reflect(ctx):
            for s in ctx["selected"]:                              # the hooks are handed files that exist (--imgpath pools)
                assert os.path.exists(s["path"]), s["path"]
            if sc["reflection"] == "openai":
                return ["fix(%s|was:%s)" % (L[s["image_name"]], prev) for s, prev in zip(ctx["selected"], ctx["reflections"])]
            return ["qwen(%s)" % L[s["image_name"]] for s in ctx["selected"]]

        def refine(ctx, reflections):
            return ["P<%s;%s>" % (L[s["image_name"]], cur[:24]) for s, cur in zip(ctx["selected"], ctx["current_prompt"])]
        return reflect, refine

    def run(self, shard, out):
        reflect, refine = self.hooks()
        orig = runner.generate
        runner.generate = _fake_generate
        try:
            return runner.run_reflection_search(self.config, None, out, _FakePipe(), shard, imgpath=self.imgpath,
                                                score_images=self.score_images, score_batch=self.score_batch,
                                                reflect=reflect, refine=refine)
        finally:
            runner.generate = orig

    # ---- comparison against the fixture
    def check(self, log, out):
        g, L = self.g, self.logical
        exact = g["scenario"]["verifier"] == "nvila"            # nvila scores are f32 on both sides; the scalar table is Python floats
        same = (lambda a, b: a == b) if exact else (lambda a, b: a == pytest.approx(b, abs=1e-6))
        assert [r["round"] for r in log] == list(range(0, self.R + 1))
        for rec, gr in zip(log[1:], g["rounds"]):
            assert rec["round"] == gr["round"]
            assert [L[n] for n in rec["generated"]] == gr["generated"]
            assert [L[p] if p is not None else None for p in rec["parents"]] == gr["parents"], f"round {gr['round']}: parent mapping"
            assert [[p] for p in rec["prompts"]] == gr["prompts"], f"round {gr['round']}: prompts"
            assert rec.get("reflections") == gr["reflections"] and rec.get("refined_prompt") == gr["refined_prompt"]
            assert rec["flag_terminated"] == gr["flag_terminated"]
            chains = {L[k]: v for k, v in rec["chains"].items()}
            assert list(chains) == list(gr["chains"]), "chain order"
            for k, ch in chains.items():
                assert [L[n] for n in ch["images"]] == gr["chains"][k]["images"]
                assert same(ch["scores"], gr["chains"][k]["scores"])
                assert ch.get("labels") == gr["chains"][k].get("labels")
        pdir = os.path.join(out, "00000")
        # best_img_detailedscore.jsonl: the selection with its padding, per round
        rows = [json.loads(l) for l in open(os.path.join(pdir, "best_img_detailedscore.jsonl"))]
        assert len(rows) == len(g["detailedscore"])
        for row, gd in zip(rows, g["detailedscore"]):
            assert [L[n] for n in row["filenames_batch"]] == gd["filenames_batch"]
            if exact:
                assert [(L[e["image_name"]], e["label"], e["score"]) for e in row["evaluation"]] == \
                       [(e["image_name"], e["label"], e["score"]) for e in gd["evaluation"]]
            else:
                assert [L[e["image_name"]] for e in row["evaluation"]] == [e["image"] for e in gd["evaluation"]]
        # best_img_meta.jsonl
        meta_path = os.path.join(pdir, "best_img_meta.jsonl")
        if g["meta"] is None:
            assert not os.path.exists(meta_path)
        else:
            lines = open(meta_path).read().splitlines()
            assert len(lines) == len(g["meta"])
            for mine, ref in zip(lines, g["meta"]):
                key, val = mine.split(": ", 1)
                rkey, rval = ref.split(": ", 1)
                val = json.loads(val)
                if key.startswith("filenames_batch"):
                    val = [L[n] for n in val]
                assert key == rkey and val == json.loads(rval)
        assert json.load(open(os.path.join(pdir, "metadata.jsonl"))) == json.loads(g["metadata_jsonl"])

        # the image-copy artefacts: same file names, same images
        def holds(d):
            res = {}
            for f in sorted(os.listdir(os.path.join(pdir, d))):
                data = open(os.path.join(pdir, d, f), "rb").read()
                src = [n for n in L if not os.path.isabs(n) and open(os.path.join(pdir, n), "rb").read() == data]
                assert len(src) == 1
                res[os.path.splitext(f)[0]] = L[src[0]]
            return res
        for d in ("samples_lastround", "samples_path_bestround", "samples_best"):
            assert holds(d) == {os.path.splitext(k)[0]: v for k, v in g[d].items()}, d
        assert sorted(os.listdir(os.path.join(pdir, "midimg"))) == sorted(os.path.splitext(os.path.basename(f))[0] + ".pt"
                                                                           for r in g["rounds"] for f in
                                                                           [f"{r['round']}_round@{s}" for s in runner.candidate_seeds(0, r["round"], self.N)])


@pytest.mark.parametrize("name", sorted(GOLD))
def test_reflection_search_equals_reference_fixture(name, tmp_path):
    rp = _Replay(GOLD[name], str(tmp_path))
    out = str(tmp_path / "out")
    log = rp.run(search.Shard(0, 1), out)
    rp.check(log, out)


def test_relative_imgpath_reaches_the_hooks_as_existing_files(tmp_path, monkeypatch):
    """with `--imgpath img` (relative to the CWD) the round-1 parents' names are CWD-relative; the reflect / refine hooks
    must receive them as they are, not re-rooted under the output directory (generated names ARE output-relative)."""
    rp = _Replay(GOLD["nvila_reflect_openai_refine"], str(tmp_path))
    monkeypatch.chdir(tmp_path)
    rel = _Replay(GOLD["nvila_reflect_openai_refine"], str(tmp_path))
    rel.imgpath = "img"
    rel.logical.update({os.path.join("img", "00000", "samples", os.path.basename(k)): v for k, v in rp.logical.items() if k.endswith(".png")})
    log = rel.run(search.Shard(0, 1), str(tmp_path / "out_rel"))          # (the reflect hook asserts every path it is handed exists)
    ref = rp.run(search.Shard(0, 1), str(tmp_path / "out_abs"))
    for a, b in zip(log[1:], ref[1:]):                                      # the same tree as with the absolute --imgpath
        assert [rel.logical[n] for n in a["selected_names"]] == [rp.logical[n] for n in b["selected_names"]]
        assert a["prompts"] == b["prompts"] and a["scores"] == b["scores"]
    assert not os.path.isabs(log[1]["selected_names"][0])


def test_reference_generate_kwargs_and_condition_geometry():
    """What the reference hands to generate() per candidate (recorded): a single "cot" condition resized to condition_size with
    position_delta [0, -condition_size // 16], `default_lora=True`, the config's `model` dict."""
    g = GOLD["nvila_n4_r3"]
    assert g["generate_kwargs"] == {"default_lora": True, "height": 1024, "width": 1024,
                                    "model_config": {"add_cond_attn": False, "latent_lora": False, "union_cond_attn": True}}
    for r in g["rounds"]:
        for c in r["conditions"]:
            assert c == [{"image": c[0]["image"], "position_delta": [0, -32], "size": [512, 512], "type": "cot"}]
    cond = runner._payload_condition(_FakePipe(), torch.zeros(1, 16, 64), 64, 64, 32)
    assert cond.condition_type == "cot" and list(cond.position_delta) == [0, -2]


def test_tree_unit_rules():
    t = search.ReflectionTree("nvila")
    assert t.select([(0.2, 0), (0.9, 1), (0.4, 1), (0.1, 0), (0.9, 1)], 5) == [1, 4, 2, 3, 0]
    assert t.select([(0.3, 1)], 4) == [0, 0]                     # ONE repetition of the head (:179-182), not a fill
    assert t.select([(0.3, 1), (0.5, 0), (0.1, 1)], 4) == [0, 2, 1, 0]
    assert search.ReflectionTree("openai").select([(0.5, 0), (0.9, 0), (0.5, 1)], 3) == [1, 0, 2]
    with pytest.raises(ValueError):
        search.ReflectionTree("gemini")


def test_owner_only_latent_handoff_volume():
    """With the reference's topk = N every candidate is selected once: the hand-off moves N latents in total, not world x N
    (8 x 32 x 512 KiB = 128 MiB per round -> 16 MiB)."""
    like = torch.empty(1, 4096, 64, dtype=torch.bfloat16)
    sel = list(range(31, -1, -1))
    assert search.selected_latents_bytes(search.Shard(0, 8), sel, like) == 32 * 512 * 1024
    assert search.selected_latents_bytes(search.Shard(3, 8), [5, 5, 5, 5], like) == 8 * 512 * 1024   # one owner, one slot per rank


# --------------------------------------------------------------------------------------------- world size 2 (gloo)
def _worker(rank, world, port, q, name, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    shard = search.init_distributed("gloo")
    rp = _Replay(GOLD[name], tmp)
    out = os.path.join(tmp, "out")
    log = rp.run(shard, out)
    if rank == 0:
        rp.check(log, out)
    q.put((rank, log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("name", ["nvila_n8_r3", "nvila_pool2_pad", "nvila_reflect_openai_refine"])
def test_reflection_search_two_ranks_equal_the_fixture(name, tmp_path):
    _Replay(GOLD[name], str(tmp_path))                           # the pool on disk, before the ranks start
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, name, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=200) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[1], "the two ranks disagree on the search log"
