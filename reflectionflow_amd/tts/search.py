"""Candidate-parallel search tree.

The reference generates the N candidates of a round one after the other on one GPU
(tts/tts_t2i_noise_scaling.py:44-70, tts/tts_reflectionflow.py:297-332).  Candidates share nothing
but the weights, so here candidate i of a round runs on rank i % world_size (one process per GPU,
weights replicated: 24 GB of 288 GB), and the ONLY exchange is one all-gather of the per-candidate
verifier outputs {score f32, label i32} (8-byte records) at the round boundary (<= 256 B for N = 32 -> pure latency
on xGMI; RCCL when the backend is "nccl", gloo in the CPU tests).  Every rank then runs the same
deterministic top-k.  The reflection loop adds ONE all-gather of the selected packed latents per round
(`allgather_selected_latents`, topk x 512 KiB per rank at 1024^2) instead of the reference's PNG-path hand-off.

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


def _collective_device(device=None):
    if device is not None:
        return torch.device(device)
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allgather_scores(shard: Shard, n: int, local: Dict[int, Tuple[float, int]], device=None) -> List[Tuple[float, int]]:
    """All-gather the round message of the n candidates: ONE collective of 8-byte records {float32 score, int32 label}
    (<= 256 B for N = 32), label in {0: no, 1: yes}, -1 = empty slot.  `local` maps this rank's candidate
    indices to (score, label).  Returns the full list on every rank."""
    s = torch.tensor([float(local[i][0]) for i in shard.mine(n)], dtype=torch.float32)
    lab = torch.tensor([int(local[i][1]) for i in shard.mine(n)], dtype=torch.int32)
    sc, lb = allgather_score_tensors(shard, n, s, lab, device=device)
    return [(float(a), int(b)) for a, b in zip(sc.tolist(), lb.tolist())]


def allgather_score_tensors(shard: Shard, n: int, scores: torch.Tensor, labels: torch.Tensor, device=None,
                            collective: Optional[bool] = None):
    """Tensor form of the exchange (what an on-device verifier feeds): `scores` f32 / `labels` i32 of this rank's candidates
    (in shard.mine(n) order, on any device) -> (scores [n] f32, labels [n] i32) in candidate order, on the CPU, identical on
    every rank.  The wire format is one int32 [per, 2] tensor per rank: {bit pattern of the f32 score, label}.
    collective: None = only when world_size > 1; True = also at world size 1 (the GPU-box RCCL rehearsal test)."""
    per = (n + shard.world_size - 1) // shard.world_size
    mine = shard.mine(n)
    assert scores.numel() == len(mine) and labels.numel() == len(mine)
    if collective is None:
        collective = shard.world_size > 1
    dev = _collective_device(device) if collective else scores.device
    rec = torch.empty(per, 2, dtype=torch.int32, device=dev)
    rec[:, 0] = torch.tensor(float("nan"), dtype=torch.float32).view(torch.int32)
    rec[:, 1] = -1
    if len(mine):
        rec[: len(mine), 0] = scores.to(dev, torch.float32).contiguous().view(torch.int32)
        rec[: len(mine), 1] = labels.to(dev, torch.int32)
    if collective:
        out = torch.empty(shard.world_size * per, 2, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(out, rec)
    else:
        out = rec
    out = out.cpu().view(shard.world_size, per, 2)
    idx = torch.arange(n)
    sel = out[idx % shard.world_size, idx // shard.world_size]             # candidate i lives at [i % world][i // world]
    return sel[:, 0].contiguous().view(torch.float32), sel[:, 1].contiguous()


def allgather_selected_latents(shard: Shard, sel: Sequence[int], local: Dict[int, torch.Tensor], like: torch.Tensor, device=None):
    """Hand the selected candidates' packed latents to every rank with ONE all-gather (the reference hands PNG paths over,
    tts_reflectionflow.py:145,160,328-332): every rank contributes a [len(sel), ...] buffer holding the selected latents it
    owns (zeros elsewhere); slot j of the result is read from its owner's block.  len(sel) x 512 KiB per rank at 1024^2.
    `like`: a tensor with a latent's shape/dtype (for ranks that own none).  Returns the list of latents in `sel` order."""
    if shard.world_size == 1:
        return [local[i] for i in sel]
    dev = _collective_device(device)
    k = len(sel)
    mine = torch.zeros((k,) + tuple(like.shape), dtype=like.dtype, device=dev)
    for j, i in enumerate(sel):
        if i in local:
            mine[j] = local[i].to(dev)
    out = torch.empty((shard.world_size * k,) + tuple(like.shape), dtype=like.dtype, device=dev)
    dist.all_gather_into_tensor(out, mine)
    out = out.view((shard.world_size, k) + tuple(like.shape))
    return [out[shard.owner(i), j].to(like.device) for j, i in enumerate(sel)]


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
    sc, lab = stub_score_batch(latents[None] if latents.dim() == 2 else latents, [seed])
    return float(sc[0]), int(lab[0])


def stub_score_batch(latents: torch.Tensor, seeds: Sequence[int]):
    """Batched verifier contract `score_batch(latents [n, S, 64], seeds) -> (scores f32 [n], labels i32 [n])` ON THE
    LATENTS' DEVICE: what a real on-device verifier (NVILA yes/no logit, tts/verifiers/nvila_verifier.py:4-10;
    tts_reflectionflow.py:157-170) plugs into -- one call per round and rank, no per-candidate host round trip; the round's
    only host sync is the all-gather's.  The stub's score = f(seed, latent statistic), deterministic."""
    lat = latents.reshape(latents.shape[0], -1) if latents.dim() > 2 else latents
    base = torch.tensor([(((int(sd) * 2654435761) & 0xFFFFFFFF) % 10007) / 10007.0 for sd in seeds], dtype=torch.float32,
                        device=lat.device)
    stat = lat.float().abs().mean(dim=1)
    scores = 0.5 * base + 0.5 * torch.remainder(stat, 1.0)
    return scores, (scores >= 0.5).to(torch.int32)


def run_round(shard: Shard, seeds: Sequence[int], generate_fn: Callable[[int, int], torch.Tensor],
              verifier: Optional[Callable[[torch.Tensor, int], Tuple[float, int]]] = None, topk: int = 1,
              score_batch: Optional[Callable] = None):
    """One search round: this rank generates its candidates, scores them (ONE batched verifier call: `score_batch`, default
    the stub; a per-candidate `verifier(latents, seed) -> (score, label)` is wrapped), all ranks exchange the {f32, i32}
    records with one all-gather and agree on the top-k.  Returns (selected candidate indices, all scores, local latents)."""
    n = len(seeds)
    mine = shard.mine(n)
    local_lat: Dict[int, torch.Tensor] = {}
    for i in mine:
        local_lat[i] = generate_fn(i, int(seeds[i]))
    if verifier is not None and score_batch is None:
        res = [verifier(local_lat[i], int(seeds[i])) for i in mine]
        sc = torch.tensor([r[0] for r in res], dtype=torch.float32)
        lab = torch.tensor([r[1] for r in res], dtype=torch.int32)
    elif mine:
        fn = score_batch or stub_score_batch
        sc, lab = fn(torch.stack([local_lat[i].reshape(-1, local_lat[i].shape[-1]) for i in mine]), [int(seeds[i]) for i in mine])
    else:
        sc, lab = torch.empty(0, dtype=torch.float32), torch.empty(0, dtype=torch.int32)
    s_all, l_all = allgather_score_tensors(shard, n, sc, lab)
    scores = [(float(a), int(b)) for a, b in zip(s_all.tolist(), l_all.tolist())]
    return select_topk(scores, topk), scores, local_lat
