"""Candidate-parallel search tree.

The reference generates the N candidates of a round one after the other on one GPU
(tts/tts_t2i_noise_scaling.py:44-70, tts/tts_reflectionflow.py:297-332).  Candidates share nothing
but the weights, so here candidate i of a round runs on rank i % world_size (one process per GPU,
weights replicated: 24 GB of 288 GB), and the ONLY exchange is one all-gather of the per-candidate
verifier outputs {score f32, label i32} at the round boundary (<= 256 B for N = 32 -> pure latency
on xGMI; RCCL when the backend is "nccl", gloo in the CPU tests).  Every rank then runs the same
deterministic top-k, so no further communication is needed.

Selection rule = the reference's NVILA key (tts_reflectionflow.py:165-170): label "yes" first by
descending score, then "no" by ascending score; ties broken by candidate index; the selection is
padded by repetition when topk exceeds the pool (:179-182).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class Shard:
    rank: int
    world_size: int

    def mine(self, n: int) -> List[int]:
        """Indices of the candidates this rank generates (round-robin: i -> rank i % world)."""
        return list(range(self.rank, n, self.world_size))

    def owner(self, i: int) -> int:
        return i % self.world_size


def init_distributed(backend: Optional[str] = None) -> Shard:
    """One process per GPU (torchrun env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return Shard(rank, world)


def allgather_scores(shard: Shard, n: int, local: Dict[int, Tuple[float, int]], device=None) -> List[Tuple[float, int]]:
    """All-gather {score, label} of the n candidates of a round.  `local` maps this rank's candidate
    indices to (score, label in {0: no, 1: yes}).  Returns the full list on every rank."""
    per = (n + shard.world_size - 1) // shard.world_size
    buf = torch.full((per, 2), float("nan"), dtype=torch.float32)
    for slot, i in enumerate(shard.mine(n)):
        s, lab = local[i]
        buf[slot, 0], buf[slot, 1] = float(s), float(lab)
    if shard.world_size > 1:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        buf = buf.to(device)
        out = torch.empty(shard.world_size * per, 2, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(out, buf)
        out = out.cpu().view(shard.world_size, per, 2)
    else:
        out = buf.view(1, per, 2)
    res: List[Tuple[float, int]] = []
    for i in range(n):
        r, slot = i % shard.world_size, i // shard.world_size
        res.append((float(out[r, slot, 0]), int(out[r, slot, 1])))
    return res


def nvila_sort_key(score: float, label: int, index: int):
    """tts_reflectionflow.py:165-170 (+ index tie-break so every rank picks the same candidates)."""
    return (0, -score, index) if label == 1 else (1, score, index)


def select_topk(scores: Sequence[Tuple[float, int]], topk: int) -> List[int]:
    order = sorted(range(len(scores)), key=lambda i: nvila_sort_key(scores[i][0], scores[i][1], i))
    sel = order[:topk]
    if topk > len(sel) and sel:                       # tts_reflectionflow.py:179-182
        sel = sel + sel[: topk - len(sel)]
    return sel


def stub_verifier(latents: torch.Tensor, seed: int) -> Tuple[float, int]:
    """Deterministic stand-in for NVILA-Lite-2B / GPT-4o (remote-code VLM / HTTPS API, unavailable
    offline and outside the hot path): a score derived from the candidate's seed and latent
    statistics.  Only its OUTPUT CONTRACT (float score, yes/no label) matters for the exchange."""
    h = (seed * 2654435761) & 0xFFFFFFFF
    base = (h % 10007) / 10007.0
    stat = float(latents.float().abs().mean().item())
    score = 0.5 * base + 0.5 * (stat % 1.0)
    return score, int(score >= 0.5)


def run_round(shard: Shard, seeds: Sequence[int], generate_fn: Callable[[int, int], torch.Tensor],
              verifier: Callable[[torch.Tensor, int], Tuple[float, int]] = stub_verifier, topk: int = 1):
    """One search round: this rank generates its candidates, scores them, all ranks exchange the
    scores and agree on the top-k.  Returns (selected candidate indices, all scores, local latents)."""
    n = len(seeds)
    local_lat: Dict[int, torch.Tensor] = {}
    local_scores: Dict[int, Tuple[float, int]] = {}
    for i in shard.mine(n):
        lat = generate_fn(i, int(seeds[i]))
        local_lat[i] = lat
        local_scores[i] = verifier(lat, int(seeds[i]))
    scores = allgather_scores(shard, n, local_scores)
    return select_topk(scores, topk), scores, local_lat
