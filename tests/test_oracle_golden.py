"""CPU: the oracle (oracle/flux_oracle.py) reproduces the fixtures that
tests/golden/make_golden.py produced from the REFERENCE's own block.py / transformer.py /
generate.py.  Bit-exact where the fixture was made by the same torch build; a tight
tolerance guards against BLAS-threading differences on other hosts."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import flux_oracle as O
from tests.golden_util import BLOCK_MODES, SHAPES, T, build, load, wsum

RT, AT = 2e-5, 2e-5


def close(a, b, what):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    assert a.shape == b.shape, what
    assert torch.allclose(a, b, rtol=RT, atol=AT), f"{what}: max|d|={(a - b).abs().max():.3e}"


@pytest.mark.parametrize("geom", ["hd32", "hd128"])
@torch.no_grad()
def test_blocks_match_reference_fixture(geom):
    z = load(f"blocks_{geom}")
    m = build(geom)
    assert wsum(m) == bytes(z["weights_sha256"]).hex(), "synthetic weights drifted (torch RNG changed?)"
    x, e, c, temb, ctemb = (T(z[k]) for k in ("x", "e", "c", "temb", "ctemb"))
    rope = m.pos_embed(torch.cat([T(z["txt_ids"]), T(z["img_ids"])]))
    crope = m.pos_embed(T(z["cond_ids"]))
    blk, sblk = m.transformer_blocks[0], m.single_transformer_blocks[0]
    for mode, (use_c, cfg, cf) in BLOCK_MODES.items():
        cc, ccrope = c, crope
        if mode == "cond_addattn" and "c_addattn" in z:
            cc, ccrope = T(z["c_addattn"]), m.pos_embed(T(z["img_ids"]))
        for a in (blk.attn, sblk.attn):
            if cf is not None:
                a.c_factor = torch.ones(1, 1) * cf
            elif hasattr(a, "c_factor"):
                del a.c_factor
        kw = dict(condition_latents=cc if use_c else None, cond_rotary_emb=ccrope if use_c else None)
        o = O.attn_forward(blk.attn, hidden_states=x, encoder_hidden_states=e, image_rotary_emb=rope,
                           model_config=cfg, **kw)
        for i, b in enumerate(o):
            close(b, z[f"attn_{mode}_{i}"], f"attn[{mode}][{i}]")
        o = O.block_forward(blk, hidden_states=x, encoder_hidden_states=e, temb=temb,
                            cond_temb=ctemb if use_c else None, image_rotary_emb=rope, model_config=cfg, **kw)
        for i, b in enumerate(o):
            if b is not None:
                close(b, z[f"block_{mode}_{i}"], f"block[{mode}][{i}]")
        if mode == "cond_addattn":
            continue
        skw = dict(condition_latents=cc, cond_temb=ctemb, cond_rotary_emb=ccrope) if use_c else {}
        o = O.single_block_forward(sblk, hidden_states=torch.cat([e, x], 1), temb=temb,
                                   image_rotary_emb=rope, model_config=cfg, **skw)
        o = o if isinstance(o, tuple) else (o,)
        for i, b in enumerate(o):
            close(b, z[f"single_{mode}_{i}"], f"single[{mode}][{i}]")


@pytest.mark.parametrize("geom", ["hd32", "hd128"])
@torch.no_grad()
def test_transformer_and_lora_gating(geom):
    z = load(f"transformer_{geom}")
    kw = dict(hidden_states=T(z["lat"]), encoder_hidden_states=T(z["pe"]), pooled_projections=T(z["pooled"]),
              timestep=T(z["t"]), guidance=T(z["g"]), img_ids=T(z["img_ids"]), txt_ids=T(z["txt_ids"]))
    for lora in (False, True):
        m = build(geom, lora=lora)
        tag = "lora" if lora else "base"
        assert wsum(m) == bytes(z[f"weights_sha256_{tag}"]).hex()
        for use_c in (False, True):
            for latent_lora in ((False, True) if lora else (False,)):
                cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": latent_lora}
                b = O.tranformer_forward(m, T(z["cond"]) if use_c else None, T(z["cond_ids"]) if use_c else None,
                                         None, model_config=cfg, **kw)[0]
                key = f"out_{tag}_{'cond' if use_c else 'nocond'}_{'latlora' if latent_lora else 'nolatlora'}"
                close(b, z[key], key)
    # enable_lora semantics (lora_controller.py:5-42): image/text rows ignore LoRA unless latent_lora
    assert np.array_equal(z["out_lora_nocond_nolatlora"], z["out_base_nocond_nolatlora"])
    assert not np.array_equal(z["out_lora_nocond_latlora"], z["out_base_nocond_nolatlora"])
    assert not np.array_equal(z["out_lora_cond_nolatlora"], z["out_base_cond_nolatlora"])


@pytest.mark.parametrize("geom", ["hd32", "hd128"])
@torch.no_grad()
def test_generate_loop(geom):
    z = load(f"loop_{geom}")
    s = SHAPES[geom]
    for use_c in (False, True):
        m = build(geom, lora=use_c)
        cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
        traj = []
        b = O.denoise(m, T(z["lat"]).clone(), T(z["pe"]), T(z["pooled"]), 4, guidance_scale=3.5,
                      condition_latents=T(z["cond"]) if use_c else None,
                      condition_ids=T(z["cond_ids"]) if use_c else None, model_config=cfg,
                      image_hw=(s["gh"], s["gw"]), callback=lambda i, t, x: traj.append(x.clone()))
        key = "cond" if use_c else "nocond"
        close(b, z[f"final_{key}"], f"final[{key}]")
        close(torch.stack(traj), z[f"traj_{key}"], f"traj[{key}]")
        if use_c:
            # image-CFG second forward (generate.py:250-272) and condition_scale -> c_factor (generate.py:86-90)
            for tag, kw in (("imgcfg", dict(image_guidance_scale=1.5)), ("cscale", dict(condition_scale=1.5))):
                b = O.denoise(m, T(z["lat"]).clone(), T(z["pe"]), T(z["pooled"]), 4, guidance_scale=3.5,
                              condition_latents=T(z["cond"]), condition_ids=T(z["cond_ids"]), model_config=cfg,
                              image_hw=(s["gh"], s["gw"]), **kw)
                close(b, z[f"final_cond_{tag}"], f"final[cond,{tag}]")
                assert not torch.equal(b, T(z["final_cond"])), f"{tag} must change the result"


def test_schedule_fixture_and_formulae():
    z = load("schedule")
    for Si, mu in ((256, 0.5), (4096, 1.15)):
        assert abs(float(z[f"mu_{Si}"]) - mu) < 1e-12
    assert abs(float(z["mu_16384"]) - 3.23) < 5e-3
    for Si in (256, 1024, 4096, 16384):
        for Tn in (4, 28, 50):
            sch = O.FlowMatchEulerDiscreteScheduler()
            ts, n = O.retrieve_timesteps(sch, Tn, None, None, np.linspace(1.0, 1 / Tn, Tn), mu=O.calculate_shift(Si))
            assert n == Tn
            close(ts, z[f"timesteps_{Si}_{Tn}"], "timesteps")
            close(sch.sigmas, z[f"sigmas_{Si}_{Tn}"], "sigmas")
            assert float(sch.sigmas[0]) == 1.0 and float(sch.sigmas[-1]) == 0.0
            assert torch.all(sch.sigmas[1:] < sch.sigmas[:-1])


def test_noise_protocol():
    z = load("noise")
    for dtype, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        for seed in (0, 1, 12345):
            n = O.get_noises([seed], 256, 256, device="cpu", dtype=dtype)[seed]
            assert n.shape == (1, 256, 64) and n.dtype == dtype
            close(n[0, :2, :8], z[f"head_{tag}_{seed}"], "noise head")
            assert hashlib.sha256(n.float().numpy().tobytes()).digest() == bytes(z[f"sha_{tag}_{seed}"])
    # pack/unpack round trip (Appendix A.10)
    x = torch.randn(2, 16, 8, 12)
    p = O.pack_latents(x, 2, 16, 8, 12)
    assert p.shape == (2, 24, 64)
    assert torch.equal(O.unpack_latents(p, 64, 96), x)


@torch.no_grad()
def test_fullwidth_single_block_checksum():
    z = load("fullwidth_single")
    torch.manual_seed(0)
    blk = O.FluxSingleTransformerBlock(3072, 24, 128).float().eval()
    O.init_synthetic_(blk, seed=5, std=0.02)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(1, 768, 3072, generator=gen)
    temb = torch.randn(1, 3072, generator=gen)
    ids = torch.cat([torch.zeros(512, 3), O.prepare_latent_image_ids(16, 16)])
    rope = O.FluxPosEmbed(10000, (16, 56, 56))(ids)
    b = O.single_block_forward(blk, hidden_states=x, temb=temb, image_rotary_emb=rope, model_config={})
    close(x[0, ::97, ::389], z["x_probe"], "inputs")
    assert torch.allclose(b[0, ::97, ::389], T(z["probe"]), rtol=1e-4, atol=1e-4)
    assert abs(float(b.abs().mean()) - float(z["absmean"])) < 1e-4
