"""CPU: host-side mirror of the reference interface -- scheduler, latent packing, noise protocol,
LoRA file loading and gating, Condition ids, module-tree key names -- against the oracle and the
reference-derived fixtures."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import flux_oracle as O
from tests.golden_util import load


def test_scheduler_matches_reference_fixture():
    from reflectionflow_amd.flux.scheduler import FlowMatchEulerDiscreteScheduler, calculate_shift, retrieve_timesteps
    z = load("schedule")
    for Si in (256, 1024, 4096, 16384):
        assert calculate_shift(Si) == pytest.approx(float(z[f"mu_{Si}"]), abs=1e-12)
        for T in (4, 28, 50):
            s = FlowMatchEulerDiscreteScheduler()
            ts, n = retrieve_timesteps(s, T, None, None, np.linspace(1.0, 1 / T, T), mu=calculate_shift(Si))
            assert n == T
            assert np.array_equal(ts.numpy(), z[f"timesteps_{Si}_{T}"])
            assert np.array_equal(s.sigmas.numpy(), z[f"sigmas_{Si}_{T}"])
            dts = s.dts()
            assert len(dts) == T and all(d < 0 for d in dts) and abs(sum(dts) + 1.0) < 1e-6


def test_pack_unpack_ids_match_oracle():
    from reflectionflow_amd.flux.pipeline import FluxPipeline as P
    x = torch.randn(2, 16, 8, 12)
    assert torch.equal(P._pack_latents(x, 2, 16, 8, 12), O.pack_latents(x, 2, 16, 8, 12))
    p = P._pack_latents(x, 2, 16, 8, 12)
    assert torch.equal(P._unpack_latents(p, 64, 96, 8), x)
    assert torch.equal(P._prepare_latent_image_ids(1, 5, 7, None, torch.float32), O.prepare_latent_image_ids(5, 7))


def test_noise_protocol_matches_reference_fixture():
    from reflectionflow_amd.tts.utils import get_noises
    z = load("noise")
    for dtype, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        for seed in (0, 1, 12345):
            n = get_noises(2 ** 31 - 1, 1, 256, 256, device="cpu", dtype=dtype, seeds=[seed])[seed]
            assert hashlib.sha256(n.float().numpy().tobytes()).digest() == bytes(z[f"sha_{tag}_{seed}"])
    a = get_noises(2 ** 31 - 1, 3, 64, 64, device="cpu", dtype=torch.float32, seeds=[5, 6, 7])
    assert list(a) == [5, 6, 7] and a[5].shape == (1, 16, 64)
    torch.manual_seed(3)
    r = get_noises(2 ** 31 - 1, 2, 64, 64, device="cpu")      # reference behaviour: seeds from torch.randint
    assert len(r) == 2


def test_module_tree_uses_diffusers_key_names():
    from reflectionflow_amd.flux import modules as M
    cfg = O.tiny_config()
    pk, ok = set(M.FluxTransformer2DModel(**cfg).state_dict()), set(O.FluxTransformer2DModel(**cfg).state_dict())
    assert pk == ok
    for k in ("transformer_blocks.0.attn.to_q.weight", "transformer_blocks.1.norm1.linear.bias",
              "transformer_blocks.0.ff.net.0.proj.weight", "transformer_blocks.0.ff_context.net.2.weight",
              "single_transformer_blocks.1.proj_mlp.weight", "single_transformer_blocks.0.attn.norm_k.weight",
              "time_text_embed.timestep_embedder.linear_1.weight", "time_text_embed.guidance_embedder.linear_2.bias",
              "x_embedder.weight", "context_embedder.bias", "norm_out.linear.weight", "proj_out.weight"):
        assert k in pk, k
    # FLUX.1-dev default geometry: 19 double + 38 single blocks, D = 3072 (meta device: no memory)
    with torch.device("meta"):
        full = M.FluxTransformer2DModel()
    n = sum(p.numel() for p in full.parameters())
    assert len(full.transformer_blocks) == 19 and len(full.single_transformer_blocks) == 38
    assert abs(n / 1e9 - 11.90) < 0.02, n


def test_lora_file_loading_and_gating():
    from reflectionflow_amd.flux import modules as M
    from reflectionflow_amd.flux.lora_controller import enable_lora, set_lora_scale
    from reflectionflow_amd.flux.pipeline import FluxPipeline, lora_target_names, synthetic_lora_state_dict
    cfg = O.tiny_config()
    pipe = FluxPipeline(M.FluxTransformer2DModel(**cfg).to(torch.bfloat16))
    names = lora_target_names(pipe.transformer)
    assert set(names) == set(O.lora_target_names(O.FluxTransformer2DModel(**cfg)))      # config.yaml:53 regex
    sd = synthetic_lora_state_dict(pipe.transformer, r=8)
    assert all(k.startswith("transformer.") and (".lora_A.weight" in k or ".lora_B.weight" in k) for k in sd)
    assert pipe.load_lora_weights(sd, adapter_name="reflection") == len(names)
    lin = pipe.transformer.transformer_blocks[0].attn.to_q
    assert isinstance(lin, M.LoraLinear) and lin.scaling == {"reflection": 1.0} and lin.r == 8
    assert not isinstance(pipe.transformer.transformer_blocks[0].attn.add_q_proj, M.LoraLinear)      # text stream: none
    assert not isinstance(pipe.transformer.transformer_blocks[0].ff.net[0].proj, M.LoraLinear)
    assert "transformer_blocks.0.attn.to_q.base_layer.weight" in pipe.transformer.state_dict()
    A, B = lin.lora_factors()
    assert A.shape == (8, 256) and B.shape == (256, 8)
    # lora_controller semantics (reference lora_controller.py:5-75)
    with enable_lora((lin, pipe.transformer.context_embedder), False):
        assert lin.scaling["reflection"] == 0
    assert lin.scaling["reflection"] == 1.0
    with enable_lora((lin,), True):
        assert lin.scaling["reflection"] == 1.0
    with set_lora_scale((lin,), 0.5):
        assert lin.scaling["reflection"] == 0.5
    assert lin.scaling["reflection"] == 1.0


def test_condition_contract():
    from reflectionflow_amd.flux.condition import Condition, condition_dict
    assert condition_dict["cot"] == 12
    tokens = torch.randn(1, 16, 64)
    ids = O.prepare_latent_image_ids(4, 4)
    c = Condition("cot", tokens=tokens, ids=ids, position_delta=[0, -4])
    t, i, ty = c.encode(pipe=None)
    assert torch.equal(t, tokens) and torch.equal(i, O.condition_ids_for(64)) and torch.equal(ty, torch.full((16, 1), 12.0))
    assert torch.equal(ids, O.prepare_latent_image_ids(4, 4)), "encode must not mutate the stored ids"
    with pytest.raises(NotImplementedError):
        Condition("nope", tokens=tokens, ids=ids)


def test_config_schema_and_pipeline_surface():
    import json
    import os
    from reflectionflow_amd.flux import modules as M
    from reflectionflow_amd.flux.pipeline import FluxPipeline
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = json.load(open(os.path.join(here, "reflectionflow_amd", "tts", "configs", "flux1_dev_mi355x.json")))
    assert cfg["model"] == {"add_cond_attn": False, "latent_lora": False, "union_cond_attn": True}
    # same schema as the reference's tts/configs/*.json (flux.1_dev_nvilascore.json:3-14,29-37,46)
    for sect, keys in (("pipeline_args", ("pretrained_model_name_or_path", "torch_dtype", "height", "width",
                                          "condition_size", "max_sequence_length", "guidance_scale",
                                          "num_inference_steps", "lora_path")),
                       ("search_args", ("search_method", "search_branch", "search_rounds"))):
        assert all(k in cfg[sect] for k in keys), sect
    assert "batch_size_for_img_gen" in cfg and cfg["pipeline_args"]["height"] == 1024
    pipe = FluxPipeline(M.FluxTransformer2DModel(**O.tiny_config()).to(torch.bfloat16))
    pe, pooled, ids = pipe.encode_prompt(prompt="a cat", prompt_2=None, max_sequence_length=16)
    pe2, _, _ = pipe.encode_prompt(prompt="a cat", prompt_2=None, max_sequence_length=16)
    assert pe.shape == (1, 16, 256) and pooled.shape == (1, 64) and ids.shape == (16, 3) and torch.equal(pe, pe2)
    lat, lids = pipe.prepare_latents(1, 16, 64, 64, torch.bfloat16, "cpu", torch.manual_seed(1))
    assert lat.shape == (1, 16, 64) and lids.shape == (16, 3)
    for attr in ("transformer", "scheduler", "vae_scale_factor", "default_sample_size", "_execution_device", "device",
                 "dtype", "progress_bar", "maybe_free_model_hooks", "set_adapters", "interrupt",
                 "joint_attention_kwargs", "check_inputs", "load_lora_weights", "set_progress_bar_config"):
        assert hasattr(pipe, attr), attr
    with pytest.raises(ValueError):
        pipe.check_inputs("p", None, 100, 64)


def test_checkpoint_and_lora_files_load_unchanged(tmp_path):
    """INTEGRATION.md's claim, end to end on files: a diffusers-layout checkpoint directory
    (`transformer/config.json` + `*.safetensors` with diffusers key names) and a FLUX-Corrector style
    `pytorch_lora_weights.safetensors` (keys `transformer.<module>.lora_{A,B}.weight`, written by
    FluxPipeline.save_lora_weights in the reference's train/model.py:87-92) load through
    `FluxPipeline.from_pretrained` / `.load_lora_weights(path, adapter_name=)` exactly as the tts scripts call
    them (tts_reflectionflow.py:498-505) and reproduce every tensor bit for bit."""
    import json
    import os
    from safetensors.torch import save_file
    from reflectionflow_amd import engine as E
    from reflectionflow_amd.flux import modules as M
    from reflectionflow_amd.flux.pipeline import FluxPipeline, lora_target_names, synthetic_lora_state_dict
    from tests.golden_util import GEOMS
    cfg = dict(GEOMS["hd128"])
    src = M.FluxTransformer2DModel(**cfg).to(torch.bfloat16)
    M.init_synthetic_(src, seed=4)
    root = tmp_path / "ckpt"
    os.makedirs(root / "transformer")
    json.dump({**{k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
               "_class_name": "FluxTransformer2DModel"}, open(root / "transformer" / "config.json", "w"))
    sd = {k: v.contiguous() for k, v in src.state_dict().items()}
    keys = sorted(sd)
    half = len(keys) // 2                                   # two shards, like the real checkpoint
    save_file({k: sd[k] for k in keys[:half]}, str(root / "transformer" / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in keys[half:]}, str(root / "transformer" / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    # diffusers key names are the attribute paths of SURVEY 8b
    for k in ("transformer_blocks.0.attn.to_q.weight", "transformer_blocks.1.norm1.linear.weight",
              "transformer_blocks.0.ff.net.0.proj.weight", "transformer_blocks.0.ff.net.2.weight",
              "single_transformer_blocks.1.proj_mlp.weight", "time_text_embed.timestep_embedder.linear_1.weight",
              "transformer_blocks.0.attn.norm_added_q.weight", "norm_out.linear.bias", "x_embedder.weight"):
        assert k in sd, k
    pipe = FluxPipeline.from_pretrained(str(root), torch_dtype=torch.bfloat16)
    got = pipe.transformer.state_dict()
    assert sorted(got) == keys and all(torch.equal(got[k], sd[k]) for k in keys)
    assert pipe.transformer.config.num_layers == cfg["num_layers"]
    # LoRA file
    lora = synthetic_lora_state_dict(pipe.transformer, r=8, seed=2)
    ldir = tmp_path / "corrector"
    os.makedirs(ldir)
    save_file({k: v.contiguous() for k, v in lora.items()}, str(ldir / "pytorch_lora_weights.safetensors"))
    n = pipe.load_lora_weights(str(ldir), adapter_name="reflection")
    assert n == len(lora_target_names(pipe.transformer)) == len(lora) // 2
    mods = dict(pipe.transformer.named_modules())
    for name in lora_target_names(pipe.transformer):
        m = mods[name]
        assert isinstance(m, M.LoraLinear) and m.active_adapters == ["reflection"] and m.scaling["reflection"] == 1.0
        assert torch.equal(m.lora_A["reflection"].weight, lora[f"transformer.{name}.lora_A.weight"])
        assert torch.equal(m.lora_B["reflection"].weight, lora[f"transformer.{name}.lora_B.weight"])
        assert torch.equal(m.base_layer.weight, sd[name + ".weight"])
    # a LoRA file that targets a module the engine has no K-segment for is refused, not silently dropped
    bad = {"transformer.transformer_blocks.0.attn.add_q_proj.lora_A.weight": torch.zeros(8, 256, dtype=torch.bfloat16),
           "transformer.transformer_blocks.0.attn.add_q_proj.lora_B.weight": torch.zeros(256, 8, dtype=torch.bfloat16)}
    with pytest.raises(E.ops.RFError, match="no K-segment"):
        pipe.load_lora_weights(bad)


def test_joint_attention_kwargs_scale_is_not_silently_ignored():
    """pipeline.joint_attention_kwargs must reflect the running call so the `scale != 1 unsupported`
    guard in tranformer_forward can fire."""
    from reflectionflow_amd.flux import modules as M
    from reflectionflow_amd.flux.pipeline import FluxPipeline
    from tests.golden_util import GEOMS
    pipe = FluxPipeline(M.FluxTransformer2DModel(**GEOMS["hd128"]))
    assert pipe.joint_attention_kwargs is None
    pipe._joint_attention_kwargs = {"scale": 0.5}
    assert pipe.joint_attention_kwargs == {"scale": 0.5}


def test_rope_table_cache_works_under_inference_mode():
    """FluxEngine.rope_tables keyed its identity fast path on Tensor._version, which raises for tensors created
    under torch.inference_mode() -- a caller wrapping generate() in inference_mode crashed.  The cache must fall through to the
    value-level compare there and keep the identity fast path for ordinary tensors (host logic only: a bare engine object on CPU)."""
    from reflectionflow_amd import engine as E
    from reflectionflow_amd.flux import modules as M

    class _Tr:
        pos_embed = M.FluxPosEmbed(10000, (16, 56, 56))
    eng = E.FluxEngine.__new__(E.FluxEngine)
    eng._rope_cache, eng.device, eng.tr = {}, torch.device("cpu"), _Tr()
    txt, img = torch.zeros(8, 3), O.prepare_latent_image_ids(4, 4)
    cos, sin = eng.rope_tables(txt, img)
    ref_cos, ref_sin = O.FluxPosEmbed(10000, (16, 56, 56))(torch.cat([txt, img.float()], 0))
    assert torch.equal(cos, ref_cos) and torch.equal(sin, ref_sin)
    assert eng.rope_tables(txt, img)[0] is cos                                   # identity fast path
    with torch.inference_mode():
        t2, i2 = torch.zeros(8, 3), O.prepare_latent_image_ids(4, 4)
        assert t2.is_inference()
        c2, s2 = eng.rope_tables(t2, i2)                                          # used to raise RuntimeError
        assert c2 is cos and s2 is sin                                            # value-level hit
        c3, _ = eng.rope_tables(torch.zeros(9, 3), i2)                            # a miss computes inside inference mode
        assert c3.shape[0] == 9 + 16
    img.add_(1)                                                                   # an in-place edit must miss
    assert not torch.equal(eng.rope_tables(txt, img)[0], cos)


def test_trace_gaps_tool_on_a_synthetic_kernel_trace(tmp_path):
    """tools/trace_gaps.py (profiles/r05_train_step_gaps*.md): a step = the kernels between two optimizer launches; busy time, idle
    time and the idle time booked on the kernel in front of each gap."""
    import os
    import subprocess
    import sys
    rows = [("rf::prodigy_apply_kernel<false>", 0, 10_000),
            ("rf::gemm_bf16_pp16e_kernel(rf::GemmParams)", 20_000, 320_000),          # 10 us gap behind the optimizer (outside the step)
            ("void rf::ln_mod_kernel<6>(x)", 326_000, 336_000),                       # 6 us gap behind the GEMM
            ("void at::native::vectorized_elementwise_kernel<8, Foo>(int)", 1_336_000, 1_340_000),   # 1 ms gap behind ln_mod
            ("rf::prodigy_apply_kernel<false>", 1_342_000, 1_350_000)]
    f = tmp_path / "t_kernel_trace.csv"
    f.write_text('"Kernel_Name","Start_Timestamp","End_Timestamp"\n' + "".join(f'"{n}",{a},{b}\n' for n, a, b in rows))
    out = tmp_path / "gaps.md"
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "trace_gaps.py")
    r = subprocess.run([sys.executable, tool, str(f), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    assert "4 kernels" in text and "busy 0.32 ms" in text and "idle 1.01 ms" in text
    assert "`ln_mod_kernel<6>` | 1 | 1.00" in text and "1.00 ms at kernel 2 / 4" in text
