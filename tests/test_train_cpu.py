"""SURVEY 8f row 4 (training), CPU side: the oracle of the training step against the fixture RECORDED FROM THE REFERENCE's own
`OminiModel.step` (tests/golden/make_train_golden.py), and the data-parallel LoRA-gradient all-reduce over gloo at world size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import train_oracle as TO
from tests.golden_util import T, build, load, wsum


def _fixture_step():
    z = load("train_step_hd128")
    m = TO.set_trainable(build("hd128", lora=True).train())
    assert np.array_equal(np.frombuffer(bytes.fromhex(wsum(m)), dtype=np.uint8), z["weights_sha256"])
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    loss, pred = TO.training_step(m, T(z["x_0"]), T(z["img_ids"]), T(z["pe"]), T(z["pooled"]), T(z["txt_ids"]), T(z["cond"]),
                                  T(z["cond_ids"]), T(z["t"]), T(z["x_1"]), cfg)
    loss.backward()
    return z, m, loss, pred


def test_oracle_training_step_equals_the_reference_fixture():
    """loss, prediction and EVERY LoRA gradient of the restated step == what the reference's own step produced (fp32, bit-exact)."""
    z, m, loss, pred = _fixture_step()
    assert torch.equal(loss.detach(), T(z["loss"]))
    assert torch.equal(pred.detach(), T(z["pred"]))
    grads = TO.lora_parameters(m)
    assert len(grads) == 50 and sorted("grad/" + n for n in grads) == sorted(k for k in z if k.startswith("grad/"))
    for n, p in grads.items():
        assert torch.equal(p.grad, T(z["grad/" + n])), n
    # the draws follow the reference's order: t first, then x_1 (model.py:185-186)
    torch.manual_seed(1234)
    t, x_1 = TO.draw_t_x1(T(z["x_0"]))
    assert torch.equal(t, T(z["t"])) and torch.equal(x_1, T(z["x_1"]))


def test_only_lora_factors_are_trainable():
    m = TO.set_trainable(build("hd128", lora=True))
    tr = [n for n, p in m.named_parameters() if p.requires_grad]
    assert tr and all("lora_" in n for n in tr)


# ------------------------------------------------------------------------------------------------- world size 2
def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reflectionflow_amd.train.step import allreduce_lora_grads
    g = torch.Generator().manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(4, 16, generator=g).to(torch.bfloat16)), torch.nn.Parameter(torch.randn(16, 4, generator=g).to(torch.bfloat16)),
          torch.nn.Parameter(torch.randn(3, generator=g).to(torch.bfloat16))]
    gr = torch.Generator().manual_seed(100 + rank)
    ps[0].grad = torch.randn(4, 16, generator=gr).to(torch.bfloat16)
    ps[1].grad = torch.randn(16, 4, generator=gr).to(torch.bfloat16)
    if rank == 0:
        ps[2].grad = torch.ones(3, dtype=torch.bfloat16)            # rank 1 has none: it must still take part with zeros
    nbytes = allreduce_lora_grads(ps, world)
    q.put((rank, nbytes, [p.grad.float().clone() for p in ps]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_lora_gradient_allreduce_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (n, g) for r, n, g in (q.get(timeout=100) for _ in range(2))}
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == (64 + 64 + 3) * 2              # ONE flat bf16 bucket
    for a, b in zip(got[0][1], got[1][1]):
        assert torch.equal(a, b), "ranks hold different averaged gradients"
    want = []
    for rank in range(2):
        gr = torch.Generator().manual_seed(100 + rank)
        want.append([torch.randn(4, 16, generator=gr).to(torch.bfloat16).float(), torch.randn(16, 4, generator=gr).to(torch.bfloat16).float()])
    for i in range(2):
        exp = ((want[0][i].to(torch.bfloat16) + want[1][i].to(torch.bfloat16)) / 2).float()
        assert torch.allclose(got[0][1][i], exp, atol=2e-2)
    assert torch.allclose(got[0][1][2], torch.full((3,), 0.5))


def test_batched_adaln_lora_terms_equal_the_per_block_form():
    """train.step._lora_terms_batched: the LoRA terms of all blocks' AdaLN linears from ONE stacked product must equal the per-linear
    F.linear(F.linear(s, A), B) * scaling, values and parameter gradients; mixed shapes fall back to the per-linear form."""
    import torch.nn as nn
    from reflectionflow_amd.flux import modules as M
    from reflectionflow_amd.train.step import _lora_term, _lora_terms_batched
    torch.manual_seed(1)

    def lin(i, o, r):
        l = M.LoraLinear(nn.Linear(i, o), r, 2 * r)
        for p in (l.lora_A["default"].weight, l.lora_B["default"].weight):
            nn.init.normal_(p)
        return l
    s = torch.randn(1, 64)
    for lins in ([lin(64, 96, 8) for _ in range(5)], [lin(64, 96, 8), lin(64, 48, 4)], [lin(64, 96, 8), nn.Linear(64, 96)]):
        got = _lora_terms_batched(lins, s)
        ref = [None if (t := _lora_term(l, s)) is None else t[0] for l in lins]
        ps = [p for l in lins if isinstance(l, M.LoraLinear) for p in (l.lora_A["default"].weight, l.lora_B["default"].weight)]
        live = [(a, b) for a, b in zip(got, ref) if b is not None]
        assert all((a is None) == (b is None) for a, b in zip(got, ref))
        assert all(torch.allclose(a, b, atol=1e-5) for a, b in live)
        gs = [torch.randn_like(b) for _, b in live]
        for x, y in zip(torch.autograd.grad([a for a, _ in live], ps, gs), torch.autograd.grad([b for _, b in live], ps, gs)):
            assert torch.allclose(x, y, atol=1e-4)


# ------------------------------------------------------------------------------------------------- optimizers (round 5)
def test_adamw_oracle_is_torch_adamw_bit_for_bit():
    """oracle/optim_oracle.adamw_step is PINNED: 6 steps on 3 tensors against torch.optim.AdamW itself (fp32, CPU, single-tensor path),
    parameters and both moments bit-equal after every step."""
    from oracle import optim_oracle as OO
    g = torch.Generator().manual_seed(0)
    shapes = [(33, 7), (128,), (5, 4, 3)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [p.detach().clone() for p in ps]
    ms, vs = [torch.zeros_like(p) for p in mine], [torch.zeros_like(p) for p in mine]
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    opt = torch.optim.AdamW(ps, foreach=False, fused=False, **kw)
    for step in range(1, 7):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        opt.step()
        for p, gr, m, v in zip(mine, grads, ms, vs):
            OO.adamw_step(p, gr, m, v, step, lr=kw["lr"], beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.05)
        for p, q, m, v in zip(ps, mine, ms, vs):
            st = opt.state[p]
            assert torch.equal(p.detach(), q) and torch.equal(st["exp_avg"], m) and torch.equal(st["exp_avg_sq"], v), step


def test_prodigy_oracle_follows_the_published_algorithm():
    """prodigyopt is not available offline (PARITY UNPINNED, stated in oracle/optim_oracle.py): what can be held is the algorithm's own
    structure -- d never decreases, d_hat is the ratio of the two running sums, an all-zero gradient moves nothing, weight decay is
    decoupled, safeguard_warmup only changes the s-recursion -- and that from d0 = 1e-6 it finds the scale of a quadratic problem
    (distance 3 to the optimum: d grows by four orders of magnitude, stays below the distance, and the iterate converges) without a learning rate."""
    from oracle import optim_oracle as OO
    g = torch.Generator().manual_seed(1)
    target = torch.full((256,), 3.0 / 16.0)                    # ||x0 - x*|| = 3
    x = torch.zeros(256)
    st = OO.prodigy_init(x)
    OO.prodigy_step(x, torch.zeros(256), st, use_bias_correction=True, safeguard_warmup=True, weight_decay=0.01)
    assert st["k"] == 0 and st["d"] == 1e-6 and float(x.abs().max()) == 0.0         # the skipped step
    ds = []
    for _ in range(400):
        grad = (x - target) + 0.01 * torch.randn(256, generator=g)
        d_before = st["d"]
        OO.prodigy_step(x, grad, st, use_bias_correction=True, safeguard_warmup=True, weight_decay=0.0)
        assert st["d"] >= d_before
        assert abs(st["d_hat"] - st["d_numerator"] / st["d_denom"]) <= 1e-12 * abs(st["d_hat"])
        ds.append(st["d"])
    assert 1e3 * 1e-6 < ds[-1] < 3.0, ds[-1]                  # from 1e-6 up by four orders of magnitude, below the true distance
    assert float((x - target).norm()) < 0.1 * 3.0
    # decoupled decay: with a zero gradient history (m = 0) the update is x <- x (1 - wd dlr) exactly
    y = torch.ones(8)
    st2 = OO.prodigy_init(y)
    st2.update(d=0.5, d_max=0.5, k=10)
    st2["s"] += 1.0                                            # a non-zero denominator without moments
    y_before = y.clone()
    OO.prodigy_step(y, torch.zeros(8), st2, weight_decay=0.1)
    assert torch.allclose(y, y_before * (1 - 0.1 * 0.5), rtol=1e-6)
    # the flat-bucket binding refuses what it cannot run (CPU tensors): the product has no CPU fallback
    from reflectionflow_amd.ops import RFError
    from reflectionflow_amd.train.optim import FlatLoraBucket, LoraAdamW, build_optimizer
    with pytest.raises(RFError):
        LoraAdamW([torch.nn.Parameter(torch.zeros(8, dtype=torch.bfloat16))])
    with pytest.raises(RFError):
        FlatLoraBucket([torch.nn.Parameter(torch.zeros(8))])              # fp32 parameters: not the LoRA training setup
    with pytest.raises(NotImplementedError):
        build_optimizer([], {"type": "Lion", "params": {}})


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reflectionflow_amd.train.optim import FlatLoraBucket
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 16).to(torch.bfloat16)), torch.nn.Parameter(torch.randn(16, 4).to(torch.bfloat16)),
          torch.nn.Parameter(torch.randn(3).to(torch.bfloat16))]
    before = [p.detach().clone() for p in ps]
    b = FlatLoraBucket(ps)
    assert all(torch.equal(p.detach(), v) for p, v in zip(ps, before))           # values survive the re-homing
    assert b.numel == 64 + 64 + 8 and all(p.data_ptr() == b.param.data_ptr() + 2 * o for p, o in zip(ps, b.offsets))
    gr = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 16, generator=gr).to(torch.bfloat16)
    for rep in range(2):                                                         # autograd accumulates INTO the bucket's views, twice
        loss = ((x @ ps[0].t()) @ ps[1].t()).float().pow(2).sum() * (1.0 if rank == 0 or rep == 0 else 0.0)
        loss.backward()
    assert ps[0].grad.data_ptr() == b.grad.data_ptr() and float(b.grad[:128].abs().max()) > 0
    assert float(b.grad[128:].abs().max()) == 0.0                                # the unused parameter contributes zeros, no None
    local = b.grad.float().clone()
    nbytes = b.all_reduce(world)
    q.put((rank, nbytes, local, b.grad.float().clone()))
    b.zero_grad()
    assert float(b.grad.abs().max()) == 0.0 and ps[2].grad.data_ptr() == b.grad.data_ptr() + 2 * 128
    ps[1].grad = None                                                            # a caller dropping a gradient does not detach the bucket
    b.zero_grad()
    assert ps[1].grad.data_ptr() == b.grad.data_ptr() + 2 * 64
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_allreduce_two_ranks():
    """train/optim.FlatLoraBucket over gloo, world 2: the backward writes straight into the flat gradient buffer, ONE all-reduce (SUM)
    runs on it as it lies, both ranks end with the same sums (the optimizer kernel's grad_scale = 1 / world takes the average)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (n, loc, red) for r, n, loc, red in (q.get(timeout=100) for _ in range(2))}
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == (64 + 64 + 8) * 2
    assert torch.equal(got[0][2], got[1][2]), "ranks hold different reduced gradients"
    want = (got[0][1].to(torch.bfloat16) + got[1][1].to(torch.bfloat16)).float()
    assert torch.allclose(got[0][2], want, rtol=2e-2, atol=1e-3)
