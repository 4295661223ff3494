"""CPU, world_size 2 over gloo: candidate sharding + the round-boundary all-gather of verifier
scores (RCCL on the GPU node) + identical top-k on every rank, independent of the world size."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reflectionflow_amd.tts import search


def test_selection_rule_matches_reference_key():
    # tts_reflectionflow.py:165-170: "yes" first by descending score, then "no" by ascending score
    scores = [(0.2, 0), (0.9, 1), (0.4, 1), (0.1, 0), (0.9, 1)]
    assert search.select_topk(scores, 3) == [1, 4, 2]
    assert search.select_topk(scores, 5) == [1, 4, 2, 3, 0]
    assert search.select_topk([(0.3, 1)], 3) == [0, 0]  # padded by repetition (:179-182), capped by pool
    assert search.Shard(1, 4).mine(10) == [1, 5, 9] and search.Shard(0, 1).mine(3) == [0, 1, 2]


def _gen(i, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(8, 4, generator=g)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    shard = search.init_distributed("gloo")
    assert (shard.rank, shard.world_size) == (rank, world)
    seeds = [11, 22, 33, 44, 55]
    sel, scores, local = search.run_round(shard, seeds, _gen, topk=2)
    assert sorted(local) == shard.mine(len(seeds))
    q.put((rank, sel, scores))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_round_equals_single_rank():
    seeds = [11, 22, 33, 44, 55]
    ref_sel, ref_scores, _ = search.run_round(search.Shard(0, 1), seeds, _gen, topk=2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, sel, scores in got:
        assert sel == ref_sel, f"rank {rank} selected {sel}, single-process {ref_sel}"
        assert scores == pytest.approx(ref_scores)
