"""Shared command line of the two search drivers (reference: tts/tts_t2i_noise_scaling.py:79-124 and
tts/tts_reflectionflow.py:466-589 -- same flag names; see runner.py for what is and is not carried over).

Launch one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m reflectionflow_amd.tts.tts_reflectionflow --pipeline_config_path <json> --meta_path <jsonl> --output_dir out

`tts_t2i_noise_scaling` and `tts_reflectionflow` are three-line entry points that call main() with their mode.
"""
import argparse
import json
import os

import torch

from . import runner, search

MODES = {"noise_scaling": runner.run_noise_scaling, "reflection": runner.run_reflection_search}


def parse_cli_args(argv=None):
    p = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    p.add_argument("--pipeline_config_path", type=str, default=os.path.join(here, "configs", "flux1_dev_mi355x.json"))
    p.add_argument("--start_index", type=int, default=0)
    p.add_argument("--end_index", type=int, default=-1)
    p.add_argument("--imgpath", type=str, default="")
    p.add_argument("--output_dir", type=str, default="output")
    p.add_argument("--meta_path", type=str, default="meta.jsonl")
    p.add_argument("--synthetic", action="store_true", help="random-init FLUX.1-dev-shaped weights (no checkpoint offline)")
    p.add_argument("--small", action="store_true", help="with --synthetic: 2+2-block model for plumbing tests")
    p.add_argument("--dist_backend", choices=["nccl", "gloo"], default=None, help="process-group backend for WORLD_SIZE > 1 (default: nccl = RCCL)")
    p.add_argument("--ranks_share_gpu", action="store_true",
                   help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --dist_backend gloo: RCCL refuses two ranks on one device)")
    return p.parse_args(argv)


@torch.no_grad()
def main(mode: str, argv=None):
    if mode not in MODES:
        raise ValueError(f"mode must be one of {sorted(MODES)}")
    args = parse_cli_args(argv)
    with open(args.pipeline_config_path) as f:
        config = json.load(f)
    config.update(vars(args))
    if args.ranks_share_gpu:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.dist_backend != "gloo":
            raise SystemExit("--ranks_share_gpu needs --dist_backend gloo (RCCL refuses two ranks on one device)")
        os.environ["LOCAL_RANK"] = "0"                # init_distributed() binds the rank to cuda:LOCAL_RANK for nccl
    shard = search.init_distributed(args.dist_backend)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    search.bind_host_threads_to_gpu_numa_node(dev.index)
    pipe = runner.build_pipeline(config, dev, synthetic=args.synthetic, small=args.small)
    os.makedirs(args.output_dir, exist_ok=True)
    if mode == "reflection" and args.imgpath:
        # tts_reflectionflow.py:535-565: prompts and the round-1 pool come from the --imgpath tree (one folder per prompt with
        # metadata.jsonl + samples/), i.e. from a tts_t2i_noise_scaling output directory
        return runner.run_reflection_search(config, None, args.output_dir, pipe, shard, start_index=args.start_index,
                                            imgpath=args.imgpath)
    with open(args.meta_path) as fp:
        metadatas = [json.loads(line) for line in fp]
    metadatas = metadatas[args.start_index:] if args.end_index == -1 else metadatas[args.start_index:args.end_index]
    prompts = [m["prompt"] for m in metadatas]
    return MODES[mode](config, prompts, args.output_dir, pipe, shard, start_index=args.start_index, metadatas=metadatas)
