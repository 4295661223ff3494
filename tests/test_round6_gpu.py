"""Round 6, training row closed around the step: gradient clipping as the reference's Trainer runs it, the LoRA
checkpoint writer (train -> save -> `load_lora_weights` -> generate), optimizer resume, Prodigy's lr == 0 gate."""
import os

import pytest
import torch

from tests.golden_util import T, build, load
from tests.test_model_gpu import rel_l2, to_product
from tests.test_round5_gpu import _params, _ulps_bf16

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
CFG = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _hd128(dev):
    from oracle import train_oracle as TO
    z = load("train_step_hd128")
    om = TO.set_trainable(build("hd128", lora=True).train())
    pipe = to_product(om, dev)
    tb = lambda k: T(z[k]).to(dev)   # noqa: E731
    batch = dict(x_0=tb("x_0").to(BF), img_ids=tb("img_ids"), prompt_embeds=tb("pe").to(BF), pooled_prompt_embeds=tb("pooled").to(BF),
                 text_ids=tb("txt_ids"), condition_latents=tb("cond").to(BF), condition_ids=tb("cond_ids"), t=tb("t"), x_1=tb("x_1").to(BF))
    return pipe, batch


@pytest.mark.parametrize("scale,max_norm,grad_scale", [(0.1, 0.5, 1.0), (1e-4, 0.5, 1.0), (0.1, 0.5, 0.25), (3.0, 1.0, 0.5)])
def test_clip_grad_norm_kernel_against_torch(dev, scale, max_norm, grad_scale):
    """rf_lora_clip_grad_norm over the flat gradient bucket vs torch.nn.utils.clip_grad_norm_ on the same (averaged) gradients:
    the norm against fp64 (1e-6), the clipped gradients within one bf16 ulp of torch's on fp32 copies (torch on bf16 tensors rounds
    the per-tensor norms and the coefficient to bf16 first: that form is checked at bf16's resolution, 1 %); a norm under the limit
    leaves the bucket untouched; bit-reproducible."""
    from reflectionflow_amd.train.optim import LoraAdamW
    ps, g = _params(dev, seed=11)
    opt = LoraAdamW(ps, lr=1e-3)
    opt.grad_scale = grad_scale
    grads = [(torch.randn(p.shape, generator=g, device=dev) * scale).to(BF) for p in ps]
    for p, gr in zip(ps, grads):
        p.grad.copy_(gr)
    before = opt.bucket.grad.clone()
    out = opt.clip_grad_norm_(max_norm).clone()
    norm64 = float(torch.sqrt(sum((gr.double() * grad_scale).pow(2).sum() for gr in grads)))
    assert abs(float(out[0]) - norm64) <= 1e-6 * norm64
    coef = min(1.0, max_norm / (norm64 + 1e-6))
    assert abs(float(out[1]) - coef) <= 2e-6 * coef
    if coef == 1.0:
        assert torch.equal(opt.bucket.grad, before), "a gradient under the limit must not be touched"
    else:
        # torch on fp32 copies of the averaged gradients (the arithmetic without bf16 intermediates), scaled back to the bucket's SUM form
        f32 = [torch.nn.Parameter(torch.zeros_like(gr, dtype=torch.float32)) for gr in grads]
        for q, gr in zip(f32, grads):
            q.grad = gr.float() * grad_scale
        tn = torch.nn.utils.clip_grad_norm_(f32, max_norm)
        assert abs(float(tn) - norm64) <= 1e-5 * norm64
        for p, q in zip(ps, f32):
            assert _ulps_bf16(p.grad, (q.grad / grad_scale).to(BF)) <= 1.0
        # torch on the bf16 tensors themselves (what the reference's Trainer calls): same result at bf16's own resolution
        b16 = [torch.nn.Parameter(torch.zeros_like(gr)) for gr in grads]
        for q, gr in zip(b16, grads):
            q.grad = (gr.float() * grad_scale).to(BF)
        torch.nn.utils.clip_grad_norm_(b16, max_norm)
        a = torch.cat([p.grad.float().flatten() * grad_scale for p in ps])
        b = torch.cat([q.grad.float().flatten() for q in b16])
        assert rel_l2(a, b) < 1e-2
        assert abs(float(a.norm()) - max_norm) <= 6e-3 * max_norm          # the clipped (averaged) gradient has the limit's norm
    # bit-reproducible
    opt.bucket.grad.copy_(before)
    again = opt.clip_grad_norm_(max_norm)
    first = opt.bucket.grad.clone()
    opt.bucket.grad.copy_(before)
    opt.clip_grad_norm_(max_norm)
    assert torch.equal(first, opt.bucket.grad) and torch.equal(again, out)


def test_training_step_clips_between_allreduce_and_update(dev):
    """FluxTrainer.training_step = zero_grad -> step -> backward -> (all-reduce) -> clip_grad_norm_(gradient_clip_val) -> optimizer.step
    (the reference: Lightning Trainer(gradient_clip_val=0.5), train/train.py:165): with a limit far under the gradient norm the
    parameters after one AdamW step equal those of the manual sequence (bit-equal: same kernels, same order), they differ from the
    unclipped step, and the default limit is the reference's 0.5."""
    from reflectionflow_amd.train.step import FluxTrainer
    oc = {"type": "AdamW", "params": {"lr": 1e-3, "weight_decay": 0.0}}
    res = {}
    for mode in ("clipped", "manual", "unclipped"):
        pipe, batch = _hd128(dev)
        tr = FluxTrainer(pipe.transformer, CFG, gradient_clip_val=None if mode != "clipped" else 1e-3)
        opt = tr.configure_optimizers(oc)
        if mode == "clipped":
            tr.training_step(batch, sample_by_sample=False)
            norm, coef = opt._clip_out.tolist()
            assert norm > 1e-2 and abs(coef - 1e-3 / (norm + 1e-6)) <= 1e-5 * coef
        else:
            opt.zero_grad()
            tr.step(batch).backward()
            if mode == "manual":
                opt.clip_grad_norm_(1e-3)
            opt.step()
        res[mode] = opt.bucket.param.clone()
    assert torch.equal(res["clipped"], res["manual"])
    assert not torch.equal(res["clipped"], res["unclipped"])
    pipe, _ = _hd128(dev)
    assert FluxTrainer(pipe.transformer, CFG).gradient_clip_val == 0.5
    assert FluxTrainer(pipe.transformer, CFG, gradient_clip_val=0).gradient_clip_val is None
    with pytest.raises(Exception):
        FluxTrainer(pipe.transformer, CFG, gradient_clip_val=-1.0)


def test_save_lora_round_trip_train_save_load_generate(dev, tmp_path):
    """train/model.py:87-92: after training steps on the flat bucket, `FluxTrainer.save_lora(path)` writes
    `<path>/pytorch_lora_weights.safetensors` with the reference's keys (`transformer.<module>.lora_{A,B}.weight`, no adapter name,
    independent storages); a FRESH pipeline that loads it with `load_lora_weights(path)` holds bit-equal factors and generates the
    bit-equal latent of the trained pipeline -- train -> search is closed."""
    from safetensors.torch import load_file
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from reflectionflow_amd.train.step import FluxTrainer, lora_parameters
    pipe, batch = _hd128(dev)
    tr = FluxTrainer(pipe.transformer, CFG)
    tr.configure_optimizers({"type": "AdamW", "params": {"lr": 5e-3, "weight_decay": 0.01}})
    before = [p.detach().clone() for p in lora_parameters(pipe.transformer)]
    for _ in range(3):
        tr.training_step(batch)
    moved = sum(not torch.equal(a, p.detach()) for a, p in zip(before, lora_parameters(pipe.transformer)))
    assert moved >= 40, moved
    fn = tr.save_lora(str(tmp_path / "ckpt" / "3"))
    assert fn.endswith(os.path.join("ckpt", "3", "pytorch_lora_weights.safetensors"))
    sd = load_file(fn)
    named = {n: p for n, p in pipe.transformer.named_parameters() if "lora_" in n}
    assert len(sd) == len(named) == 50
    for n, p in named.items():
        key = "transformer." + n.replace(".default.weight", ".weight")
        assert key in sd and sd[key].dtype == BF and torch.equal(sd[key], p.detach().cpu()), key
    # a fresh pipeline (same base weights, no LoRA yet) + the file
    from oracle import train_oracle as TO   # noqa: F401  (the builder of the fixture model)
    fresh = to_product(build("hd128", lora=False), dev)
    assert fresh.load_lora_weights(str(tmp_path / "ckpt" / "3"), adapter_name="reflection", alpha=4.0) == 25
    z = load("train_step_hd128")
    cond = Condition("cot", tokens=batch["condition_latents"][:1], ids=batch["condition_ids"])
    gen = lambda p: generate(p, prompt_embeds=batch["prompt_embeds"][:1], pooled_prompt_embeds=batch["pooled_prompt_embeds"][:1],   # noqa: E731
                             conditions=[cond], latents=batch["x_1"][:1].clone(), height=128, width=128, num_inference_steps=4, guidance_scale=3.5,
                             model_config=CFG, default_lora=True, output_type="latent").images
    from reflectionflow_amd import engine as E
    E.invalidate(pipe.transformer)
    a, b = gen(pipe), gen(fresh)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    base = to_product(build("hd128", lora=False), dev)
    c = generate(base, prompt_embeds=batch["prompt_embeds"][:1], pooled_prompt_embeds=batch["pooled_prompt_embeds"][:1], conditions=[cond],
                 latents=batch["x_1"][:1].clone(), height=128, width=128, num_inference_steps=4, guidance_scale=3.5, model_config=CFG,
                 default_lora=True, output_type="latent").images
    assert rel_l2(a, c) > 1e-3, "the trained LoRA does not move the latent"
    del z


@pytest.mark.parametrize("kind", ["AdamW", "Prodigy"])
def test_optimizer_state_dict_resumes(dev, kind):
    """`state_dict()` -> `load_state_dict()` on a new optimizer over a copy of the parameters: the next steps are bit-equal to the
    uninterrupted run (Prodigy: d, d_max, the numerator, k, p0 and s travel too)."""
    from reflectionflow_amd.train.optim import build_optimizer
    oc = {"type": kind, "params": {"lr": 2e-3, "weight_decay": 0.01} if kind == "AdamW" else
          {"lr": 1.0, "use_bias_correction": True, "safeguard_warmup": True, "weight_decay": 0.01}}
    ps, g = _params(dev, seed=21)
    opt = build_optimizer(ps, oc)
    grads = [[(torch.randn(p.shape, generator=g, device=dev) * 0.1).to(BF) for p in ps] for _ in range(6)]

    def run(o, params, steps):
        for gs in steps:
            for p, gr in zip(params, gs):
                p.grad.copy_(gr)
            o.clip_grad_norm_(0.5)
            o.step()
    run(opt, ps, grads[:3])
    snap = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in opt.state_dict().items()}
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    run(opt, ps, grads[3:])
    opt2 = build_optimizer(ps2, {"type": kind, "params": {}})
    opt2.load_state_dict(snap)
    assert opt2.defaults["lr"] == oc["params"]["lr"] and opt2.defaults["weight_decay"] == 0.01
    run(opt2, ps2, grads[3:])
    assert torch.equal(opt.bucket.param, opt2.bucket.param)
    assert torch.equal(opt.exp_avg, opt2.exp_avg) and torch.equal(opt.exp_avg_sq, opt2.exp_avg_sq)
    if kind == "Prodigy":
        assert opt.d_state() == opt2.d_state() and opt.d_state()["k"] == 6
    with pytest.raises(Exception):
        opt2.load_state_dict(dict(snap, exp_avg=snap["exp_avg"][:-8]))


def test_prodigy_with_lr_zero_moves_nothing(dev):
    """prodigyopt gates the moment / s / numerator updates on group_lr > 0 and returns on d_denom == 0: a step with
    lr = 0 is a no-op -- parameters, moments, s, d and k unchanged -- also with safeguard_warmup, where a_s = (d / d0) d != 0."""
    from reflectionflow_amd.train.optim import LoraProdigy
    ps, g = _params(dev, seed=31)
    opt = LoraProdigy(ps, lr=0.0, safeguard_warmup=True, use_bias_correction=True, weight_decay=0.01)
    for p in ps:
        p.grad.copy_((torch.randn(p.shape, generator=g, device=dev) * 0.1).to(BF))
    p0 = opt.bucket.param.clone()
    for _ in range(3):
        opt.step()
    ds = opt.d_state()
    assert torch.equal(opt.bucket.param, p0) and ds["k"] == 0 and ds["d"] == 1e-6 and ds["d_denom"] == 0.0
    assert float(opt.exp_avg.float().abs().max()) == 0.0 and float(opt.exp_avg_sq.float().abs().max()) == 0.0 and float(opt.s.float().abs().max()) == 0.0


def test_auto_checkpointing_falls_back_to_recompute_on_oom(dev):
    """gradient_checkpointing="auto" keeps what its cost model says fits; when that is wrong the step must not die: an
    out-of-memory error inside the attempt -> ONE retry with every block re-computing (the reference's flag), which stays on.  The
    update is bit-equal to a trainer built with gradient_checkpointing=True (the gradients are bit-equal in every mode)."""
    from reflectionflow_amd.train.step import FluxTrainer
    oc = {"type": "AdamW", "params": {"lr": 1e-3, "weight_decay": 0.0}}
    pipe, batch = _hd128(dev)
    tr = FluxTrainer(pipe.transformer, CFG)
    opt = tr.configure_optimizers(oc)
    real, calls = tr.step, []

    def flaky(b, generator=None):
        calls.append(tr._auto_fell_back)
        if not tr._auto_fell_back:
            loss = real(b, generator=generator)          # the attempt allocates, then dies
            assert tr.kept_blocks > 0
            raise torch.cuda.OutOfMemoryError("simulated")
        return real(b, generator=generator)
    tr.step = flaky
    loss = tr.training_step(batch, sample_by_sample=False)
    assert calls == [False, True] and tr._auto_fell_back and tr.kept_blocks == 0 and torch.isfinite(loss.float())
    got = opt.bucket.param.clone()
    tr.training_step(batch, sample_by_sample=False)
    assert calls == [False, True, True] and tr.kept_blocks == 0
    pipe2, batch2 = _hd128(dev)
    tr2 = FluxTrainer(pipe2.transformer, CFG, gradient_checkpointing=True)
    opt2 = tr2.configure_optimizers(oc)
    tr2.training_step(batch2, sample_by_sample=False)
    assert torch.equal(got, opt2.bucket.param)


@pytest.mark.parametrize("kind,mode", [("AdamW", "auto"), ("Prodigy", "auto"), ("AdamW", True)])
def test_captured_training_step_is_bit_equal_to_the_eager_step(dev, kind, mode):
    """`FluxTrainer.capture_training_step`: zero_grad + forward + backward as ONE hipGraph, the clip + optimizer tail eager.  Capturing
    must not train (parameters / optimizer state restored after its warm-up steps), and three replays on three different batches must
    leave bit-equal parameters, optimizer state and losses to three eager `training_step` calls from the same start -- with AdamW
    (host-computed bias corrections: the reason the tail stays eager), with Prodigy (device-resident distance estimate), with kept
    activations ("auto": the keep plan of the warm-up is replayed) and with the reference's per-block recompute."""
    from reflectionflow_amd.train.step import FluxTrainer
    oc = {"type": kind, "params": {"lr": 2e-3, "weight_decay": 0.01} if kind == "AdamW" else
          {"lr": 1.0, "use_bias_correction": True, "safeguard_warmup": True, "weight_decay": 0.01}}

    def batches(batch):
        g = torch.Generator(device=dev).manual_seed(5)
        out = []
        for i in range(3):
            b = dict(batch)
            b["x_0"] = (batch["x_0"].float() + 0.1 * i * torch.randn(batch["x_0"].shape, generator=g, device=dev)).to(BF)
            b["t"] = torch.sigmoid(torch.randn(batch["t"].shape, generator=g, device=dev))
            out.append(b)
        return out
    pipe_e, batch_e = _hd128(dev)
    tr_e = FluxTrainer(pipe_e.transformer, CFG, gradient_checkpointing=mode)
    opt_e = tr_e.configure_optimizers(oc)
    start = opt_e.bucket.param.clone()
    losses_e = [tr_e.training_step(b, sample_by_sample=False).clone() for b in batches(batch_e)]

    pipe_g, batch_g = _hd128(dev)
    tr_g = FluxTrainer(pipe_g.transformer, CFG, gradient_checkpointing=mode)
    opt_g = tr_g.configure_optimizers(oc)
    assert torch.equal(opt_g.bucket.param, start)
    run = tr_g.capture_training_step(batch_g, sample_by_sample=False)
    assert torch.equal(opt_g.bucket.param, start), "capturing (its warm-up steps) must not train"
    assert float(opt_g.exp_avg.float().abs().max()) == 0.0
    if kind == "Prodigy":
        assert opt_g.d_state()["k"] == 0
    else:
        assert opt_g.step_count == 0
    losses_g = [run(b).clone() for b in batches(batch_g)]
    torch.cuda.synchronize()
    for a, b in zip(losses_e, losses_g):
        assert torch.equal(a, b), (float(a), float(b))
    assert torch.equal(opt_e.bucket.param, opt_g.bucket.param)
    assert torch.equal(opt_e.exp_avg, opt_g.exp_avg) and torch.equal(opt_e.exp_avg_sq, opt_g.exp_avg_sq)
    assert not torch.equal(opt_g.bucket.param, start)
    if kind == "Prodigy":
        assert opt_e.d_state() == opt_g.d_state() and opt_g.d_state()["k"] == 3
    with pytest.raises(Exception):
        run({k: v for k, v in batch_g.items() if k != "t"})            # the captured step's inputs are fixed


def test_lora_gradients_reach_autograd_when_no_bucket_owns_the_factors(dev):
    """the fused-LoRA node writes `.grad` in place only for factors a FlatLoraBucket owns (the optimizer's buffer); otherwise it
    returns the gradients to autograd: `torch.autograd.grad(loss, factors)` yields them, bit-equal to what `.backward()` accumulates,
    tensor hooks fire, and nothing is written into `.grad` as a side effect."""
    from reflectionflow_amd.train.step import FluxTrainer, lora_parameters
    pipe, batch = _hd128(dev)
    tr = FluxTrainer(pipe.transformer, CFG)
    ps = lora_parameters(pipe.transformer)
    for p in ps:
        p.grad = None
    tr.step(batch).backward()
    want = [None if p.grad is None else p.grad.clone() for p in ps]
    assert sum(g is not None and float(g.float().abs().max()) > 0 for g in want) >= 44
    for p in ps:
        p.grad = None
    fired = []
    hooks = [p.register_hook(lambda g, i=i: fired.append(i)) for i, p in enumerate(ps)]
    got = torch.autograd.grad(tr.step(batch), ps, allow_unused=True)
    for h in hooks:
        h.remove()
    assert all(p.grad is None for p in ps), "torch.autograd.grad must not write .grad"
    for a, b in zip(got, want):
        assert (a is None and b is None) or torch.equal(a, b)
    assert len(set(fired)) >= 44
    # ... and the bucket-owned form is unchanged: after configure_optimizers the same step fills the bucket with the same gradients
    opt = tr.configure_optimizers({"type": "AdamW", "params": {"lr": 0.0, "weight_decay": 0.0}})
    opt.zero_grad()
    tr.step(batch).backward()
    for p, b in zip(ps, want):
        assert b is None or torch.equal(p.grad, b)
