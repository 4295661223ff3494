"""Round-3 GPU tests.

(1) The LAGGED-MAX attention kernel (rf_attn_desc.kernel = RF_ATTN_LAGGED16 / _SPLIT; AUTO without a usable score bound):
    P = exp2(s - m) with a per-row maximum m that is re-centred only when a tile's row sums overflow lag_thresh.
    cdna_hip_programming.md rule 26 asks three things of a rare data-dependent branch, all done here:
      * a FULL-tensor independent reference (fp64 softmax on the device, not bitwise-vs-self);
      * inputs that FORCE the branch (one key row spiked against chosen query rows at a chosen tile, scores far beyond
        the bounded kernel's |s| <= 100 contract; scores of huge common magnitude);
      * a threshold sweep: lag_thresh tiny (every tile re-centres), default (2^30) and huge (only inf trips) agree.
(2) The bounded kernel at the edge of its contract: |s| ~ 95 with adversarially aligned q / k rows vs the
    online-softmax kernel; and a model whose norm weights push the proven bound beyond 100 through the engine.
(3) RCCL: backend "nccl" initialises on the GPU box and all-gathers the {f32 score, i32 label} round message.
"""
import math
import os

import pytest
import torch

from tests.test_kernels_gpu import BF, assert_close, make_qkv
from tests.test_model_gpu import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def softmax_ref64(qp, kf, vf):
    """fp64 softmax(q k^T) v in the exp2 domain (q already carries scale * log2 e): [H,S,128] x3 -> [S, H*128] fp32."""
    s = torch.einsum("hqd,hkd->hqk", qp.double(), kf.double())
    s = s - s.amax(dim=-1, keepdim=True)
    p = torch.exp2(s)
    o = torch.einsum("hqk,hkd->hqd", p / p.sum(-1, keepdim=True), vf.double())
    return o.permute(1, 0, 2).reshape(qp.shape[1], -1).float()


def prescaled(H, S, dev, seed, qscale=1.0):
    from reflectionflow_amd import ops
    q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=seed, qscale=qscale)
    qp = (qf.float() * ops.QK_PRESCALE).to(BF)
    q[:, :S] = qp
    return q, k, vt, qp, kf, vf


@pytest.mark.parametrize("S,H", [(256, 3), (1024, 8), (2048, 8), (4608, 8), (5632, 8)])
def test_lagged_max_matches_fp64_and_the_other_kernels(dev, S, H):
    from reflectionflow_amd import _lib as L, ops
    q, k, vt, qp, kf, vf = prescaled(H, S, dev, seed=S)
    ref = softmax_ref64(qp, kf, vf)
    lib = L.load()
    o = {}
    can_split = H % 8 == 0 and (H // 8) * (S // 256) ** 2 >= 64       # >= 2 key quads per persistent workgroup of an XCD
    kernels = [("lag", L.RF_ATTN_LAGGED16, 8), ("lag again", L.RF_ATTN_LAGGED16, 8), ("bounded", L.RF_ATTN_BOUNDED16, 5), ("online", L.RF_ATTN_ONLINE256, 2)]
    # the mixed-size launch (workgroups of 256 and 192 queries; waves 4-7 of the small ones carry ONE q-tile): same rows, same code
    kernels += [("lag mix", L.RF_ATTN_LAGGED16_MIX, 11), ("bounded mix", L.RF_ATTN_BOUNDED16_MIX, 10)]
    if can_split:
        kernels += [("lag split", L.RF_ATTN_LAGGED16_SPLIT, 9), ("lag split again", L.RF_ATTN_LAGGED16_SPLIT, 9)]
    else:
        with pytest.raises(ops.RFError):       # an unrunnable request fails loudly
            ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_LAGGED16_SPLIT)
    with pytest.raises(ops.RFError):
        ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_LAGGED16_MIX, mix_small=S // 16)      # more q-tiles than the head has
    ms = 4 * min(3, S // 768)                 # 192-query workgroups per head, forced (the library's own plan needs > 256 workgroups)
    for name, kern, path in kernels:
        o[name] = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=60.0, kernel=kern, mix_small=ms if "mix" in name else 0)
        assert lib.rf_debug_last_attn_path() == path, (name, lib.rf_debug_last_attn_path())
    # AUTO without a bound -> the lagged-max kernel (plain, mixed-size or split grid), never the online-softmax ones
    ops.attention(q, k, vt, S, q_prescaled=True, score_bound=0.0)
    assert lib.rf_debug_last_attn_path() in (8, 9, 11)
    ops.attention(q, k, vt, S, q_prescaled=True, score_bound=250.0)
    assert lib.rf_debug_last_attn_path() in (8, 9, 11)
    for name, t in o.items():
        assert_close(t, ref, f"attention {name} S={S}", atol=2e-3)
    assert torch.equal(o["lag"], o["lag again"]), "not bit-stable"
    assert torch.equal(o["lag mix"], o["lag"]) and torch.equal(o["bounded mix"], o["bounded"]), "mixed-size launch differs from the plain one"
    if can_split:
        assert torch.equal(o["lag split"], o["lag split again"]), "split launch not bit-stable"
    e = {n: rel_l2(t, ref) for n, t in o.items()}
    print(f"  S={S}: rel-L2 vs fp64: " + ", ".join(f"{n} {v:.2e}" for n, v in e.items()))
    assert all(v <= 1.3 * e["online"] + 1e-4 for n, v in e.items() if n.startswith("lag"))


@pytest.mark.parametrize("kern_name", ["LAGGED16", "LAGGED16_SPLIT", "LAGGED16_MIX"])
def test_lagged_max_forced_recentring(dev, kern_name):
    """Spike one key row against chosen query rows: raw scores of +180 / +400 / +3000 (exp2 domain) appear at key tiles 0,
    5, 37 and at the last tile -- far outside the bounded kernel's contract, P = exp2(s - m_old) overflows to inf -- while
    the other rows of the same waves stay ordinary.  Every lane group of a wave sees different rows trip at different tiles."""
    from reflectionflow_amd import _lib as L, ops
    H, S = 8, 4608
    q, k, vt, qp, kf, vf = prescaled(H, S, dev, seed=11)
    kf = kf.clone()
    qp = qp.clone()
    g = torch.Generator().manual_seed(5)

    def unit():
        u = torch.randn(128, generator=g)
        return (u / u.norm()).to(dev)
    spikes = [(0, 3, 180.0), (5 * 64 + 17, 40, 400.0), (37 * 64 + 63, 300, 3000.0), (S - 1, 2049, 250.0), (S - 64, 4607, 900.0)]
    for key, qrow, target in spikes:          # key `key` = 8 u, query `qrow` = (target / 8) u  ->  q.k ~ target; one direction per spike
        u = unit()
        kf[:, key] = (8.0 * u).to(BF)
        qp[:, qrow] = ((target / 8.0) * u).to(BF)
    # a block of 16 consecutive queries (one q-tile) that all spike on the same late key: the whole tile re-centres
    u = unit()
    kf[:, 2000] = (6.0 * u).to(BF)
    qp[:, 1024:1040] = ((500.0 / 6.0) * u).to(BF)[None, None]
    q[:, :S], k[:, :S] = qp, kf
    ref = softmax_ref64(qp, kf, vf)
    kern = getattr(L, "RF_ATTN_" + kern_name)
    ms = 12 if kern_name.endswith("MIX") else 0          # 9 x 256 + 12 x 192 queries per head: rows 2304.. are in small workgroups
    o = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=0.0, kernel=kern, mix_small=ms)
    o2 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=0.0, kernel=kern, mix_small=ms)
    on = ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_ONLINE256)
    assert torch.isfinite(o.float()).all()
    assert_close(o, ref, f"lagged-max with spikes ({kern_name})", atol=3e-3)
    assert torch.equal(o, o2)
    assert rel_l2(o, ref) <= 1.3 * rel_l2(on, ref) + 1e-4
    # the spiked rows attend (almost) only to their spike key: the output row is that key's value row
    vrow = vf.float()[:, 37 * 64 + 63]                                  # [H, 128]
    assert (o.float()[300].reshape(H, 128) - vrow).abs().max() < 2e-2


def test_lagged_max_threshold_sweep(dev):
    """lag_thresh = 1e-30 (EVERY tile of every wave takes the re-centring path: m is the running maximum, as in online
    softmax), 2^30 (default), 3e38 (only an inf row sum trips): the same softmax to rounding -- on ordinary data, on data
    whose scores share a huge offset (|s| ~ 600: the default path must re-centre at tile 0 only), and with q scaled x12."""
    from reflectionflow_amd import _lib as L, ops
    H, S = 8, 2048
    for case, qscale, shift in (("plain", 1.0, 0.0), ("wide", 12.0, 0.0), ("offset", 1.0, 600.0)):
        q, k, vt, qp, kf, vf = prescaled(H, S, dev, seed=23, qscale=qscale)
        if shift:                                        # every key gets a large common component along q's mean direction
            d = qp.float().mean(1, keepdim=True)
            d = d / d.norm(dim=-1, keepdim=True)
            kf = (kf.float() + shift * d / (qp.float() * d).sum(-1).abs().mean()).to(BF)
            k[:, :S] = kf
        ref = softmax_ref64(qp, kf, vf)
        outs = {}
        for thr in (1e-30, 0.0, 3e38):
            for kern in (L.RF_ATTN_LAGGED16, L.RF_ATTN_LAGGED16_SPLIT):
                outs[(thr, kern)] = ops.attention(q, k, vt, S, q_prescaled=True, kernel=kern, lag_thresh=thr)
        on = ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_ONLINE256)
        e_on = rel_l2(on, ref)
        for key, o in outs.items():
            assert torch.isfinite(o.float()).all(), (case, key)
            e = rel_l2(o, ref)
            # (near one-hot rows -- the "wide" case -- are a handful of bf16-rounded P per row: which kernel rounds them luckier varies)
            assert e <= 1.6 * e_on + 3e-4, (case, key, e, e_on)
        print(f"  {case}: rel-L2 vs fp64 online {e_on:.2e}, lagged " + " ".join(f"{rel_l2(o, ref):.2e}" for o in outs.values()))


def test_bounded_kernel_at_the_edge_of_its_contract(dev):
    """scores of +-95 (exp2 domain) with adversarially aligned q / k rows -- the largest the bounded kernel
    accepts (score_bound <= 100: P up to 2^95, row sums up to 2^95 * S) -- against the online-softmax kernel and fp64."""
    from reflectionflow_amd import _lib as L, ops
    H, S = 8, 1024
    q, k, vt, qp, kf, vf = prescaled(H, S, dev, seed=31)
    u = torch.randn(128, generator=torch.Generator().manual_seed(9))
    u = (u / u.norm()).to(dev)
    kf, qp = kf.clone(), qp.clone()
    # half of the keys point along +u, a quarter along -u (norm 9.5); a third of the queries along +-u (norm 10): s = +-95
    kf[:, 0::2] = (9.5 * u).to(BF)
    kf[:, 1::4] = (-9.5 * u).to(BF)
    qp[:, 0::3] = (10.0 * u).to(BF)
    qp[:, 1::6] = (-10.0 * u).to(BF)
    q[:, :S], k[:, :S] = qp, kf
    smax = float(torch.einsum("hqd,hkd->hqk", qp.float(), kf.float()).abs().max())
    assert 90.0 < smax <= 99.0, smax
    ref = softmax_ref64(qp, kf, vf)
    ob = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=smax * 1.005, kernel=L.RF_ATTN_BOUNDED16)
    ob32 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=smax * 1.005, kernel=L.RF_ATTN_BOUNDED32)
    ol = ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_LAGGED16)
    on = ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_ONLINE256)
    for name, o in (("bounded 16x16", ob), ("bounded 32x32", ob32), ("lagged", ol), ("online", on)):
        assert torch.isfinite(o.float()).all(), name
        assert_close(o, ref, f"|s| = {smax:.1f}: {name}", atol=3e-3)


@torch.no_grad()
def test_engine_with_norm_weights_beyond_the_bound(dev):
    """A checkpoint whose learned norm_q / norm_k scales make the proven bound > 100 (here x3.2 -> ~200): the packed engine
    must hand attention to the lagged-max kernel (not silently to the 26 % slower online-softmax one) and the model output
    must still match the fp32 oracle within the usual calibrated tolerance.  Geometry: 64 text + 8 x 24 image tokens = 256
    (whole rounds of the attention rings), the 2 + 2 block hd128 model."""
    from oracle import flux_oracle as O
    from reflectionflow_amd import _lib as L, ops
    from reflectionflow_amd.flux.transformer import tranformer_forward
    from tests.golden_util import GEOMS, build
    from tests.test_model_gpu import bf16_oracle, check, g, to_product
    om = build("hd128")
    for blk in list(om.transformer_blocks) + list(om.single_transformer_blocks):
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            m = getattr(blk.attn, n, None)
            if m is not None and getattr(m, "weight", None) is not None:
                m.weight.mul_(3.2)
    a = om.transformer_blocks[0].attn
    bound = ops.qk_score_bound((a.norm_q.weight, a.norm_added_q.weight), (a.norm_k.weight, a.norm_added_k.weight))
    assert bound > 100.0, bound
    ob = bf16_oracle(om)
    pipe = to_product(om, dev)
    cfgm = GEOMS["hd128"]
    gen = torch.Generator().manual_seed(77)
    St, gh, gw = 64, 8, 24
    lat = torch.randn(1, gh * gw, cfgm["in_channels"], generator=gen)
    pe = torch.randn(1, St, cfgm["joint_attention_dim"], generator=gen)
    pooled = torch.randn(1, cfgm["pooled_projection_dim"], generator=gen)
    t, gd = torch.tensor([0.5]), torch.tensor([4.0])          # t*1000 = 500, g*1000 = 4000: exact in bf16 (the inherited quirk of
                                                              # transformer.py:95-98 -- 3.5*1000 -> 3504 in bf16 -- must not enter the yardstick)
    img_ids, txt_ids = O.prepare_latent_image_ids(gh, gw), torch.zeros(St, 3)
    kw = lambda f: dict(hidden_states=f(lat), encoder_hidden_states=f(pe), pooled_projections=f(pooled), timestep=f(t),  # noqa: E731
                        guidance=f(gd), img_ids=f(img_ids), txt_ids=f(txt_ids), return_dict=False)
    ref = O.tranformer_forward(om, None, None, None, model_config={}, **kw(lambda x: x))[0]
    tb = O.tranformer_forward(ob, None, None, None, model_config={}, **kw(lambda x: x.to(BF)))[0]
    out = tranformer_forward(pipe.transformer, None, None, None, model_config={}, **kw(lambda x: g(x, dev)))[0]
    assert L.load().rf_debug_last_attn_path() == 8, L.load().rf_debug_last_attn_path()
    e = check(out, ref, tb, f"transformer with qk bound {bound:.0f}")
    print(f"  qk bound {bound:.0f}: hip {e[0]:.3e}  torch-bf16 {e[1]:.3e}")


@torch.no_grad()
def test_denoise_loop_as_one_hipgraph_equals_eager(dev, monkeypatch):
    """FluxEngine.denoise(use_graph=True): the whole T-step loop captured once per (geometry, T, schedule) over static buffers
    and replayed per candidate -- bit-identical to the eager launches, also for later candidates and with a condition + LoRA."""
    from oracle import flux_oracle as O
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from tests.golden_util import SHAPES, T, build, load
    from tests.test_model_gpu import g, to_product
    z = load("loop_hd128")
    s_ = SHAPES["hd128"]
    om = build("hd128", lora=True)
    pipe = to_product(om, dev)
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}

    def run(seed, use_c):
        lat = O.get_noises([seed], s_["gh"] * 16, s_["gw"] * 16, dtype=torch.float32)[seed]
        conds = [Condition("cot", tokens=g(T(z["cond"]), dev), ids=T(z["cond_ids"]).to(dev))] if use_c else None
        return generate(pipe, conditions=conds, model_config=cfg, default_lora=True, height=s_["gh"] * 16, width=s_["gw"] * 16,
                        num_inference_steps=4, guidance_scale=3.5, latents=g(lat, dev), prompt_embeds=g(T(z["pe"]), dev),
                        pooled_prompt_embeds=g(T(z["pooled"]), dev), output_type="latent").images.clone()
    eager = {(sd, c): run(sd, c) for sd in (1, 2, 3) for c in (False, True)}
    monkeypatch.setenv("RF_DENOISE_GRAPH", "1")
    for rep in range(2):                      # capture on first use, replay afterwards
        for (sd, c), ref in eager.items():
            out = run(sd, c)
            assert torch.equal(out, ref), f"graph replay differs from eager (seed {sd}, condition {c}, pass {rep})"
    from reflectionflow_amd import engine as E
    assert len(E.engine_for(pipe.transformer)._graphs) == 2


def test_rccl_world1_allgather_of_the_round_message(dev):
    """backend="nccl" IS RCCL on ROCm: initialise it once on the GPU box (world size 1, dmabuf IPC mode) and run the
    round-boundary exchange of tts/search.py -- two all_gather_into_tensor calls, f32 scores and i32 labels."""
    import torch.distributed as dist
    from reflectionflow_amd.tts import search
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        assert dist.get_backend() == "nccl"
        torch.cuda.set_device(0)
        sc = torch.tensor([0.25, 0.75, 0.5], device=dev)
        lab = torch.tensor([1, 0, 1], dtype=torch.int32, device=dev)
        s_all, l_all = search.allgather_score_tensors(search.Shard(0, 1), 3, sc, lab, device=dev, collective=True)
        assert s_all.tolist() == [0.25, 0.75, 0.5] and l_all.tolist() == [1, 0, 1] and l_all.dtype == torch.int32
        t = torch.arange(8, device=dev, dtype=torch.float32)
        out = torch.empty(8, device=dev)
        dist.all_gather_into_tensor(out, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), torch.arange(8.0))
    finally:
        if created:
            dist.destroy_process_group()
