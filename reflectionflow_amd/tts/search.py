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
padded by repetition when topk exceeds the pool (:179-182).  A scalar ("openai") verifier sorts by
descending score (:152-156).

`ReflectionTree` is the reference's per-prompt search state (tts_reflectionflow.py:157-182 selection,
:273-279 / :297-313 candidate i <- i-th best parent, :337-395 chains, :402-444 best-of-chain / best
overall), pinned by `tests/golden/search_tree.json`, which was recorded from the reference's own `main()`.
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


def gpu_numa_node(index: int) -> Optional[int]:
    """NUMA node of GPU `index` from sysfs (its PCI function's `numa_node`), or None when the platform does not say (-1, containers
    without /sys/bus/pci, CPU-only hosts)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except Exception:
        return None


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_host_threads_to_gpu_numa_node(index: int) -> Dict[str, object]:
    """One process per GPU: pin this rank's host threads to the CPUs of its GPU's NUMA node (8 ranks of a node otherwise enqueue
    15 000 launches per candidate from wherever the scheduler put them, across the socket interconnect).  Only narrows the current
    affinity mask (a launcher's / container's own restriction is kept); a platform that does not expose the node is left alone.
    Returns what was done, for the bench line's `dist.per_rank`."""
    info: Dict[str, object] = {"numa_node": None, "cpus_bound": None}
    node = gpu_numa_node(index)
    if node is None or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            want = set(_parse_cpulist(f.read()))
        now = set(os.sched_getaffinity(0))
        cpus = sorted(want & now)
        if cpus:
            os.sched_setaffinity(0, cpus)                 # this thread (threads created from here on inherit it) ...
            try:                                           # ... and the ones that already run (the interpreter's / torch's pools, RCCL's proxies)
                for tid in os.listdir("/proc/self/task"):
                    try:
                        os.sched_setaffinity(int(tid), cpus)
                    except OSError:
                        pass
            except OSError:
                pass
            info.update(numa_node=node, cpus_bound=len(cpus))
    except Exception:
        pass
    return info


def describe_ranks(shard: "Shard", device_index: int, numa: Optional[Dict[str, object]] = None) -> Optional[dict]:
    """What an N > 1 line must carry so that a reader can PROVE N ranks on N devices ran it: the world size the backend itself
    reports, the collective library's version, and per rank {device index, name, PCI bus id, uuid, host, NUMA binding}."""
    if shard.world_size == 1:
        return None
    import socket
    mine = {"rank": shard.rank, "device_index": device_index, "device_name": None, "gcn_arch": None, "pci": None, "uuid": None,
            "host": socket.gethostname(), "pid": os.getpid(), **(numa or {})}
    if torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(device_index)
        mine.update(device_name=pr.name, gcn_arch=getattr(pr, "gcnArchName", None), uuid=str(getattr(pr, "uuid", "")),
                    pci=f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}")
    every: List[Optional[dict]] = [None] * shard.world_size
    dist.all_gather_object(every, mine)
    backend = dist.get_backend()
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
    return {"backend": backend, "world_size_seen_by_backend": dist.get_world_size(), "rccl_version": ver,
            "distinct_devices": len({(r["host"], r["pci"]) for r in every}), "per_rank": every}


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


def _owner_slots(shard: Shard, sel: Sequence[int]):
    """Where each distinct selected candidate travels: candidate i -> (owner rank, slot in the owner's buffer); every rank
    derives the same table from `sel`.  Returns (slots, per) with per = the largest number of selected candidates one rank owns."""
    slots: Dict[int, Tuple[int, int]] = {}
    count = [0] * shard.world_size
    for i in sorted(set(int(x) for x in sel)):
        r = shard.owner(i)
        slots[i] = (r, count[r])
        count[r] += 1
    return slots, max(count) if count else 0


def allgather_selected_latents(shard: Shard, sel: Sequence[int], local: Dict[int, torch.Tensor], like: torch.Tensor, device=None):
    """Hand the selected candidates' packed latents to every rank with ONE all-gather (the reference hands PNG paths over,
    tts_reflectionflow.py:145,160,328-332).  OWNERS ONLY: a rank contributes the selected latents it generated, each exactly once
    (repeats in `sel` and candidates of other ranks cost nothing), in a buffer of `per` slots = the largest per-rank count.  With
    the reference's topk = N every candidate is selected once: N x 512 KiB in total at 1024^2 (16 MiB for N = 32), not world x that.
    `like`: a tensor with a latent's shape/dtype (for ranks that own none).  Returns the list of latents in `sel` order."""
    if shard.world_size == 1:
        return [local[i] for i in sel]
    dev = _collective_device(device)
    slots, per = _owner_slots(shard, sel)
    mine = torch.zeros((per,) + tuple(like.shape), dtype=like.dtype, device=dev)
    for i, (r, j) in slots.items():
        if r == shard.rank:
            mine[j] = local[i].to(dev)
    out = torch.empty((shard.world_size * per,) + tuple(like.shape), dtype=like.dtype, device=dev)
    dist.all_gather_into_tensor(out, mine)
    out = out.view((shard.world_size, per) + tuple(like.shape))
    return [out[slots[int(i)][0], slots[int(i)][1]].to(like.device) for i in sel]


def selected_latents_bytes(shard: Shard, sel: Sequence[int], like: torch.Tensor) -> int:
    """Bytes one rank RECEIVES in allgather_selected_latents (world x per slots)."""
    _, per = _owner_slots(shard, sel)
    return shard.world_size * per * like.numel() * like.element_size()


def nvila_sort_key(score: float, label: int, index: int):
    """tts_reflectionflow.py:165-170 (+ index tie-break = Python's stable sort, so every rank picks the same candidates)."""
    return (0, -score, index) if label == 1 else (1, score, index)


def scalar_sort_key(score: float, label: int, index: int):
    """tts_reflectionflow.py:152-156: a scalar-metric ("openai") verifier -- descending score, stable."""
    return (-score, index)


SORT_KEYS = {"nvila": nvila_sort_key, "stub": nvila_sort_key, "openai": scalar_sort_key}


def select_topk(scores: Sequence[Tuple[float, int]], topk: int, kind: str = "nvila") -> List[int]:
    key = SORT_KEYS[kind]
    order = sorted(range(len(scores)), key=lambda i: key(scores[i][0], scores[i][1], i))
    sel = order[:topk]
    if topk > len(sel) and sel:                       # tts_reflectionflow.py:179-182 (one repetition of the head, no more)
        sel = sel + sel[: topk - len(sel)]
    return sel


class ReflectionTree:
    """Per-prompt state of the reflection search, the way the reference keeps it.

    Vocabulary: a POOL is the previous round's images (round 1: the `--imgpath` images, tts_reflectionflow.py:556-565);
    `select` orders it (:157-182); candidate i of the round is generated from parent `selected[i]` (:273-279, 297-313;
    candidates beyond the padded selection get no condition); `record` files the round's candidates into CHAINS
    (:358-395): round 1 opens one chain per candidate, later a candidate joins the first chain that holds its parent.
    `best_of_chains` / `best_overall` are :402-444.  Images are identified by caller-chosen names (the reference uses file
    paths).  Labels are 1 = "yes", 0 = "no" (None for a scalar verifier)."""

    def __init__(self, kind: str = "nvila"):
        if kind not in SORT_KEYS:
            raise ValueError(f"verifier kind must be one of {sorted(SORT_KEYS)}")
        self.kind = kind
        self.scalar = SORT_KEYS[kind] is scalar_sort_key
        self.chains: Dict[str, Dict[str, list]] = {}          # insertion-ordered, keyed by the round-1 image

    def select(self, scores: Sequence[Tuple[float, int]], topk: int) -> List[int]:
        return select_topk(scores, topk, self.kind)

    def record(self, search_round: int, names: Sequence[str], scores: Sequence[Tuple[float, int]],
               parents: Sequence[Optional[str]]) -> None:
        for i, name in enumerate(names):
            sc, lab = float(scores[i][0]), scores[i][1]
            if search_round == 1:
                ch = self.chains.setdefault(name, {"images": [], "scores": [], "labels": []})
                self._append(ch, name, sc, lab)
                continue
            for ch in self.chains.values():               # the first chain holding the parent (:388-392); an image is in one chain
                if parents[i] in ch["images"]:
                    self._append(ch, name, sc, lab)
                    break

    @staticmethod
    def _append(ch, name, sc, lab):
        ch["images"].append(name)
        ch["scores"].append(sc)
        ch["labels"].append(lab)

    def _best_index(self, ch) -> int:
        key = SORT_KEYS[self.kind]
        return min(range(len(ch["scores"])), key=lambda j: key(ch["scores"][j], ch["labels"][j], j))

    def best_of_chains(self) -> List[str]:
        """One image per chain, in chain order (:410-425)."""
        return [ch["images"][self._best_index(ch)] for ch in self.chains.values()]

    def best_overall(self) -> Optional[str]:
        """The best image over all chains (:428-444)."""
        key = SORT_KEYS[self.kind]
        flat = [(ch["scores"][j], ch["labels"][j], ch["images"][j]) for ch in self.chains.values() for j in range(len(ch["images"]))]
        if not flat:
            return None
        k = min(range(len(flat)), key=lambda j: key(flat[j][0], flat[j][1], j))
        return flat[k][2]

    def snapshot(self) -> Dict[str, Dict[str, list]]:
        """JSON form with the reference's field names ("labels" as yes / no; absent for a scalar verifier)."""
        out = {}
        for k, ch in self.chains.items():
            d = {"images": list(ch["images"]), "scores": list(ch["scores"])}
            if not self.scalar:
                d["labels"] = ["yes" if l == 1 else "no" for l in ch["labels"]]
            out[k] = d
        return out


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
    # one reduction PER ROW: the fp32 summation order of a row must not depend on how many rows (= which world size) share the
    # call, or a near-tie could flip between world sizes (score_batch implementations must be row-wise batch-invariant)
    stat = torch.stack([row.float().abs().mean() for row in lat]) if lat.shape[0] else lat.new_zeros(0, dtype=torch.float32)
    scores = 0.5 * base + 0.5 * torch.remainder(stat, 1.0)
    return scores, (scores >= 0.5).to(torch.int32)


def run_round(shard: Shard, seeds: Sequence[int], generate_fn: Callable[[int, int], torch.Tensor],
              verifier: Optional[Callable[[torch.Tensor, int], Tuple[float, int]]] = None, topk: int = 1,
              score_batch: Optional[Callable] = None, kind: str = "nvila"):
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
    return select_topk(scores, topk, kind), scores, local_lat
