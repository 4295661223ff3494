"""SURVEY 8a row a10, the plain t2i candidate loop (BASELINE cfg1 / cfg3 are quoted on it): `runner.run_noise_scaling` and
`cli.main("noise_scaling")` against fixtures RECORDED FROM THE REFERENCE's own `tts/tts_t2i_noise_scaling.py::main()`
(tests/golden/make_noise_scaling_golden.py ran it end to end under a stub pipeline / stub `get_noises`).

Each scenario replays the same meta file, config and seeds through this build and must reproduce: the output tree
(`<index>/metadata.jsonl` text, `<index>/samples/<round>_round@<seed>` per candidate), the stock pipeline call of every
candidate (keyword names, prompt, noise of that seed, guidance / steps / height / width), the `get_noises` geometry, the
datapoints, and the meta slicing of the CLI -- at world size 1 and, over gloo, at world size 2 (cfg3's N = 32 round).
The GPU tests run the same entry point on the HIP path with a small synthetic model."""
import json
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reflectionflow_amd.tts import cli, runner, search
from reflectionflow_amd.tts.utils import get_noises

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "noise_scaling.json")) as _f:
    GOLD = json.load(_f)

PIPE_KWARGS = ["guidance_scale", "height", "latents", "num_inference_steps", "prompt", "width"]     # tts_t2i_noise_scaling.py:60


class _RecordingPipe:
    """CPU stand-in for the pipeline: records the call, returns `0.5 * latents` as the candidate (no VAE -> `.pt` files)."""
    device, vae, image_processor = torch.device("cpu"), None, None

    def __init__(self, dtype):
        self.dtype = dtype
        self.calls = []

    def set_progress_bar_config(self, **kw):
        pass

    def __call__(self, **kw):
        self.calls.append(kw)
        return types.SimpleNamespace(images=kw["latents"] * 0.5)


def _seeds_fn(gold):
    """(absolute prompt index, round, N) -> the seeds the reference run used (its stub get_noises drew them per call, in order)."""
    start = gold["scenario"].get("start", 0)
    return lambda index, rnd, n: [gold["seed_of"][f"p{index - start}r{rnd}c{i}"] for i in range(n)]


def _metas_in_file_order(gold):
    return [{"prompt": m["prompt"], "tag": m["tag"], "include": m["include"]} for m in gold["metas"]]      # the fixture is dumped with sort_keys


def _sliced(gold):
    sc, metas = gold["scenario"], _metas_in_file_order(gold)
    start, end = sc.get("start", 0), sc.get("end", -1)
    return start, (metas[start:] if end == -1 else metas[start:end])


def _tree(out):
    t = {}
    for root, _d, files in os.walk(out):
        for fn in files:
            t[os.path.relpath(os.path.join(root, fn), out)] = os.path.join(root, fn)
    return t


def _check_tree(gold, out, ext=".pt"):
    """Same files as the reference wrote (PNG there; here the candidate's file type), same metadata.jsonl text."""
    tree = _tree(out)
    want = {(k[:-4] + ext if k.endswith(".png") else k): v for k, v in gold["tree"].items()}
    assert sorted(tree) == sorted(want)
    for rel, v in want.items():
        if rel.endswith("metadata.jsonl"):
            assert open(tree[rel]).read() == v, rel
    return tree, want


def _check_calls(gold, calls, dtype):
    """One stock call per candidate: the reference's keyword set (+ output_type), prompt, the noise of ITS seed, sizes."""
    flat = [(c["prompt"][k], c["seeds"][k], c["guidance_scale"], c["num_inference_steps"], c["height"], c["width"], c["latents_shape"][1:])
            for c in gold["calls"] for k in range(len(c["seeds"]))]
    assert all(c["kwargs"] == PIPE_KWARGS for c in gold["calls"])
    assert len(calls) == len(flat)
    for kw, (prompt, seed, g, steps, h, w, shape) in zip(calls, flat):
        assert sorted(kw) == sorted(PIPE_KWARGS + ["output_type"]) and kw["output_type"] == "latent"
        assert kw["prompt"] == [prompt]
        assert (kw["guidance_scale"], kw["num_inference_steps"], kw["height"], kw["width"]) == (g, steps, h, w)
        assert list(kw["latents"].shape) == [1] + shape and kw["latents"].dtype == dtype
        ref = get_noises(runner.MAX_SEED, 1, h, w, device="cpu", dtype=dtype, seeds=[seed])[seed]
        assert torch.equal(kw["latents"], ref), f"candidate of seed {seed} did not start from get_noises(seed)"


def _dtype(gold):
    return {"bf16": torch.bfloat16, "fp32": torch.float32}[gold["config"]["pipeline_args"]["torch_dtype"]]


@pytest.mark.parametrize("name", sorted(GOLD))
def test_run_noise_scaling_reproduces_the_reference_main(name, tmp_path):
    gold = GOLD[name]
    assert gold["from_pretrained"] == ["black-forest-labs/FLUX.1-dev", str(_dtype(gold)), "C"]
    for nc in gold["noise_calls"]:                                  # the reference's noise geometry = this build's get_noises defaults
        pa = gold["config"]["pipeline_args"]
        assert (nc["max_seed"], nc["num_samples"], nc["height"], nc["width"], nc["fn"]) == \
               (runner.MAX_SEED, gold["scenario"]["N"], pa["height"], pa["width"], "prepare_latents_for_flux")
    start, metas = _sliced(gold)
    pipe = _RecordingPipe(_dtype(gold))
    out = str(tmp_path / "out")
    dps = runner.run_noise_scaling(gold["config"], [m["prompt"] for m in metas], out, pipe, search.Shard(0, 1), start_index=start,
                                   metadatas=metas, seeds_fn=_seeds_fn(gold))
    assert dps == [{k: v for k, v in d.items() if k != "_call"} for d in gold["datapoints"]]
    assert [d["_call"]["midimg_path"] for d in gold["datapoints"]] == \
           [f"{i + start:0>5}/samples" for i in range(len(metas)) for _ in range(gold["scenario"]["R"])]
    tree, want = _check_tree(gold, out)
    _check_calls(gold, pipe.calls, _dtype(gold))
    for rel, cand in want.items():                                  # the file named after a seed holds THAT candidate
        if rel.endswith(".pt"):
            seed = gold["seed_of"][cand]
            assert rel.endswith(f"@{seed}.pt")
            pa = gold["config"]["pipeline_args"]
            ref = get_noises(runner.MAX_SEED, 1, pa["height"], pa["width"], device="cpu", dtype=_dtype(gold), seeds=[seed])[seed] * 0.5
            assert torch.equal(torch.load(tree[rel]), ref)


@pytest.mark.parametrize("name", ["slice_1_to_3_of_5", "slice_from_2", "n4_r2_two_prompts"])
def test_cli_noise_scaling_slices_the_meta_file_like_the_reference(name, tmp_path, monkeypatch):
    """`cli.main("noise_scaling")`: config + argv merge, --start_index / --end_index (tts_t2i_noise_scaling.py:120-124),
    folder numbering by absolute index (:130)."""
    gold = GOLD[name]
    sc = gold["scenario"]
    meta = tmp_path / "meta.jsonl"
    meta.write_text("".join(json.dumps(m) + "\n" for m in _metas_in_file_order(gold)))
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps(gold["config"]))
    pipe = _RecordingPipe(_dtype(gold))
    seen = {}

    def fake_build(config, device, synthetic=False, small=False):
        seen["config"] = config
        return pipe
    monkeypatch.setattr(runner, "build_pipeline", fake_build)
    monkeypatch.setattr(runner, "candidate_seeds", _seeds_fn(gold))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    out = str(tmp_path / "out")
    argv = ["--pipeline_config_path", str(cfgp), "--meta_path", str(meta), "--output_dir", out, "--start_index", str(sc.get("start", 0))]
    if "end" in sc:
        argv += ["--end_index", str(sc["end"])]
    dps = cli.main("noise_scaling", argv)
    assert seen["config"]["output_dir"] == out and seen["config"]["start_index"] == sc.get("start", 0)     # config.update(vars(args)), :98
    assert dps == [{k: v for k, v in d.items() if k != "_call"} for d in gold["datapoints"]]
    _check_tree(gold, out)
    _check_calls(gold, pipe.calls, _dtype(gold))


def test_entry_point_module_has_the_reference_script_name():
    from reflectionflow_amd.tts import tts_t2i_noise_scaling as entry
    assert callable(entry.main) and entry.parse_cli_args is cli.parse_cli_args


def test_default_seeds_are_a_pure_function_of_prompt_round_and_index():
    a = runner.candidate_seeds(3, 2, 32)
    assert a == runner.candidate_seeds(3, 2, 32) and len(set(a)) == 32 and all(0 <= s < runner.MAX_SEED for s in a)
    assert a != runner.candidate_seeds(3, 1, 32) and a != runner.candidate_seeds(4, 2, 32)
    assert runner.candidate_seeds(3, 2, 8) == a[:8]                  # a smaller branch is a prefix: N=8 and N=32 runs share candidates


def test_duplicate_seeds_are_refused(tmp_path):
    gold = GOLD["n4_r2_two_prompts"]
    with pytest.raises(ValueError, match="duplicate candidate seeds"):      # utils.py:153 asserts the same in the reference
        runner.run_noise_scaling(gold["config"], ["p"], str(tmp_path), _RecordingPipe(torch.bfloat16), search.Shard(0, 1),
                                 seeds_fn=lambda i, r, n: [5] * n)


# ------------------------------------------------------------------------------------------------ world size 2 (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gold = GOLD[name]
        start, metas = _sliced(gold)
        pipe = _RecordingPipe(_dtype(gold))
        dps = runner.run_noise_scaling(gold["config"], [m["prompt"] for m in metas], out, pipe, search.Shard(rank, world), start_index=start,
                                       metadatas=metas, seeds_fn=_seeds_fn(gold))
        q.put((rank, dps, [(c["prompt"][0], float(c["latents"].float().abs().sum())) for c in pipe.calls]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["cfg3_n32_r1", "n4_r2_two_prompts", "batched_generation_3_ragged"])
def test_noise_scaling_world2_writes_the_same_tree(name, tmp_path):
    """Candidate i on rank i % 2: the union of both ranks' files is the reference's tree; every rank returns the same datapoints."""
    gold = GOLD[name]
    out = str(tmp_path / "out")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_dps = [{k: v for k, v in d.items() if k != "_call"} for d in gold["datapoints"]]
    assert res[0][1] == want_dps and res[1][1] == want_dps
    tree, want = _check_tree(gold, out)
    n_total = sum(len(c["seeds"]) for c in gold["calls"])
    assert len(res[0][2]) + len(res[1][2]) == n_total
    N = gold["scenario"]["N"]
    assert len(res[0][2]) == len(gold["datapoints"]) * ((N + 1) // 2) and len(res[1][2]) == len(gold["datapoints"]) * (N // 2)
    pa = gold["config"]["pipeline_args"]
    for rel, cand in want.items():
        if rel.endswith(".pt"):
            seed = gold["seed_of"][cand]
            ref = get_noises(runner.MAX_SEED, 1, pa["height"], pa["width"], device="cpu", dtype=_dtype(gold), seeds=[seed])[seed] * 0.5
            assert torch.equal(torch.load(tree[rel]), ref)


# ------------------------------------------------------------------------------------------------ on the HIP path
def _small_cfg(N, R, size=256, steps=4):
    return {"pipeline_args": {"pretrained_model_name_or_path": "black-forest-labs/FLUX.1-dev", "torch_dtype": "bf16", "height": size, "width": size,
                              "guidance_scale": 3.5, "num_inference_steps": steps, "max_sequence_length": 512, "condition_size": size // 2},
            "search_args": {"search_branch": N, "search_rounds": R}}


@pytest.mark.gpu
def test_cli_noise_scaling_on_the_hip_path(tmp_path):
    """`cli.main("noise_scaling", --synthetic --small)` end to end on the GPU: the tree has the reference's names and every
    file holds what a direct stock `pipe(...)` call makes of that seed's noise (bit-equal: a candidate is a pure function
    of its seed)."""
    cfg = _small_cfg(N=4, R=2)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    metas = [{"prompt": "a photo of two cats", "tag": "counting"}, {"prompt": "a red cube", "tag": "color"}, {"prompt": "unused"}]
    (tmp_path / "meta.jsonl").write_text("".join(json.dumps(m) + "\n" for m in metas))
    out = str(tmp_path / "out")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    dps = cli.main("noise_scaling", ["--pipeline_config_path", str(tmp_path / "cfg.json"), "--meta_path", str(tmp_path / "meta.jsonl"),
                                     "--output_dir", out, "--end_index", "2", "--synthetic", "--small"])
    assert dps == [{"prompt": m["prompt"], "search_round": r, "num_noises": 4} for m in metas[:2] for r in (1, 2)]
    tree = _tree(out)
    want = {f"{i:0>5}/metadata.jsonl" for i in range(2)} | \
           {f"{i:0>5}/samples/{r}_round@{s}.pt" for i in range(2) for r in (1, 2) for s in runner.candidate_seeds(i, r, 4)}
    assert set(tree) == want
    assert open(tree["00001/metadata.jsonl"]).read() == json.dumps(metas[1])
    dev = torch.device("cuda", 0)
    pipe = runner.build_pipeline(cfg, dev, synthetic=True, small=True)
    pa = cfg["pipeline_args"]
    for i, r in ((0, 1), (1, 2)):
        for s in runner.candidate_seeds(i, r, 4)[::3]:
            noise = get_noises(runner.MAX_SEED, 1, pa["height"], pa["width"], device=dev, dtype=torch.bfloat16, seeds=[s])[s]
            lat = pipe(prompt=[metas[i]["prompt"]], latents=noise, guidance_scale=3.5, num_inference_steps=pa["num_inference_steps"],
                       height=pa["height"], width=pa["width"], output_type="latent").images
            got = torch.load(tree[f"{i:0>5}/samples/{r}_round@{s}.pt"])
            assert torch.isfinite(got.float()).all() and got.float().std() > 0.1
            assert torch.equal(got, lat.cpu())


@pytest.mark.gpu
def test_noise_scaling_cfg3_round_of_32_on_one_rank_feeds_the_reflection_driver(tmp_path):
    """BASELINE cfg3's shape on the one GPU there is: N = 32 candidates of one round through `run_noise_scaling` (small
    synthetic model), 32 distinct finite candidates; with a VAE on the pipeline the tree is PNG and is what the
    reflection driver's `--imgpath` reader takes (tts_reflectionflow.py:535-556)."""
    dev = torch.device("cuda", 0)
    cfg = _small_cfg(N=32, R=1, size=256, steps=2)
    pipe = runner.build_pipeline(cfg, dev, synthetic=True, small=True)
    out = str(tmp_path / "lat")
    dps = runner.run_noise_scaling(cfg, ["thirty-two cats"], out, pipe, search.Shard(0, 1), metadatas=[{"prompt": "thirty-two cats", "tag": "counting"}])
    assert dps == [{"prompt": "thirty-two cats", "search_round": 1, "num_noises": 32}]
    files = sorted(os.listdir(os.path.join(out, "00000", "samples")))
    assert files == sorted(f"1_round@{s}.pt" for s in runner.candidate_seeds(0, 1, 32))
    lats = torch.stack([torch.load(os.path.join(out, "00000", "samples", f)).float().flatten() for f in files])
    assert torch.isfinite(lats).all()
    d = torch.cdist(lats, lats) + torch.eye(32) * 1e9
    assert d.min() > 1.0, "two candidates of the round coincide"
    # with a VAE: PNGs, and the reflection driver reads the tree as its round-1 pool
    from reflectionflow_amd.flux import vae as V
    pipe.vae = V.init_synthetic_vae_(V.AutoencoderKL(block_out_channels=(32, 64, 64, 64), norm_num_groups=8), seed=1).eval().to(dev, torch.bfloat16)
    pipe.image_processor = V.VaeImageProcessor(vae_scale_factor=16)
    out2 = str(tmp_path / "png")
    cfg4 = _small_cfg(N=4, R=1, size=256, steps=2)
    runner.run_noise_scaling(cfg4, ["four cats"], out2, pipe, search.Shard(0, 1), metadatas=[{"prompt": "four cats", "tag": "counting"}])
    pools = runner.read_imgpath(out2)
    assert len(pools) == 1 and pools[0]["metadata"][0]["prompt"] == "four cats"
    assert [os.path.basename(p) for p in pools[0]["images"]] == sorted(f"1_round@{s}.png" for s in runner.candidate_seeds(0, 1, 4))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_cfg3_shape_eight_ranks_share_the_gpu_through_the_cli(tmp_path):
    """BASELINE cfg3's round, N = 32 candidates -> 4 per rank on 8 ranks, launched the way the 8-GPU node launches it
    (`torch.distributed.run --nproc-per-node 8 -m reflectionflow_amd.tts.tts_t2i_noise_scaling`), here with all ranks on the one GPU
    there is (`--ranks_share_gpu --dist_backend gloo --synthetic --small`): the tree is complete, has the reference's names, and
    every candidate is BIT-EQUAL to the one a single rank generates for that seed (results do not depend on the world size)."""
    import subprocess
    import sys
    cfg = _small_cfg(N=32, R=1, size=256, steps=2)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    (tmp_path / "meta.jsonl").write_text(json.dumps({"prompt": "thirty-two cats", "tag": "counting"}) + "\n")
    out = str(tmp_path / "out8")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "reflectionflow_amd.tts.tts_t2i_noise_scaling", "--pipeline_config_path", str(tmp_path / "cfg.json"),
           "--meta_path", str(tmp_path / "meta.jsonl"), "--output_dir", out, "--synthetic", "--small", "--dist_backend", "gloo", "--ranks_share_gpu"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    seeds = runner.candidate_seeds(0, 1, 32)
    files = sorted(os.listdir(os.path.join(out, "00000", "samples")))
    assert files == sorted(f"1_round@{s}.pt" for s in seeds)
    assert open(os.path.join(out, "00000", "metadata.jsonl")).read() == json.dumps({"prompt": "thirty-two cats", "tag": "counting"})
    # the same round on ONE rank in this process
    dev = torch.device("cuda", 0)
    pipe = runner.build_pipeline(cfg, dev, synthetic=True, small=True)
    out1 = str(tmp_path / "out1")
    runner.run_noise_scaling(cfg, ["thirty-two cats"], out1, pipe, search.Shard(0, 1))
    for f in files:
        a, b = torch.load(os.path.join(out, "00000", "samples", f)), torch.load(os.path.join(out1, "00000", "samples", f))
        assert torch.isfinite(a.float()).all() and torch.equal(a, b), f
