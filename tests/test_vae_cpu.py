"""CPU: the VAE plug-in (reflectionflow_amd/flux/vae.py -- PyTorch(-ROCm) modules, not part of the HIP hot path) against
the independent functional restatement in oracle/vae_oracle.py, its diffusers wire format, the image processor, and
the reference call sites either side of the loop (pipeline_tools.py:7-30, condition.py:96-132, generate.py:302-307).
Parity with diffusers itself is UNPINNED (no source / vectors offline) -- see oracle/vae_oracle.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import flux_oracle as O
from oracle import vae_oracle as VO
from reflectionflow_amd.flux import modules as M
from reflectionflow_amd.flux import vae as V
from reflectionflow_amd.flux.condition import Condition
from reflectionflow_amd.flux.pipeline import FluxPipeline
from tests.golden_util import GEOMS

SMALL = dict(block_out_channels=(32, 64, 64, 64), norm_num_groups=8)


@torch.no_grad()
def test_vae_modules_match_functional_restatement():
    m = V.init_synthetic_vae_(V.AutoencoderKL(**SMALL), seed=1).float().eval()
    sd = m.state_dict()
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(0)) * 2 - 1
    mo = m.encoder(x)
    assert mo.shape == (2, 32, 8, 12)
    assert torch.allclose(mo, VO.vae_encode_moments(sd, x, groups=8), atol=1e-5, rtol=1e-5)
    z = torch.randn(2, 16, 8, 12, generator=torch.Generator().manual_seed(1))
    y = m.decode(z).sample
    assert y.shape == (2, 3, 64, 96)
    assert torch.allclose(y, VO.vae_decode(sd, z, groups=8), atol=1e-4, rtol=1e-4)
    # posterior sampling: mean + std * noise with the caller's generator
    g = torch.Generator().manual_seed(5)
    lat = m.encode(x).latent_dist.sample(g)
    noise = torch.randn(2, 16, 8, 12, generator=torch.Generator().manual_seed(5))
    assert torch.allclose(lat, VO.sample_latent(mo, noise), atol=1e-5)


def test_flux_vae_wire_format():
    """FLUX.1-dev `vae/` checkpoint layout: 244 tensors, 83 819 683 parameters, diffusers key names."""
    with torch.device("meta"):
        m = V.AutoencoderKL()
    sd = m.state_dict()
    assert len(sd) == 244 and sum(v.numel() for v in sd.values()) == 83_819_683
    shapes = {
        "encoder.conv_in.weight": (128, 3, 3, 3), "encoder.down_blocks.0.downsamplers.0.conv.weight": (128, 128, 3, 3),
        "encoder.down_blocks.1.resnets.0.conv_shortcut.weight": (256, 128, 1, 1),
        "encoder.mid_block.attentions.0.to_q.weight": (512, 512), "encoder.mid_block.attentions.0.to_out.0.bias": (512,),
        "encoder.mid_block.attentions.0.group_norm.weight": (512,), "encoder.conv_out.weight": (32, 512, 3, 3),
        "decoder.conv_in.weight": (512, 16, 3, 3), "decoder.up_blocks.0.resnets.2.conv1.weight": (512, 512, 3, 3),
        "decoder.up_blocks.2.resnets.0.conv_shortcut.weight": (256, 512, 1, 1),
        "decoder.up_blocks.2.upsamplers.0.conv.weight": (256, 256, 3, 3), "decoder.conv_norm_out.weight": (128,),
        "decoder.conv_out.weight": (3, 128, 3, 3),
    }
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape))
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in sd and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert (m.config.scaling_factor, m.config.shift_factor) == (0.3611, 0.1159)


def test_image_processor_round_trip():
    from PIL import Image
    ip = V.VaeImageProcessor(16)
    arr = (np.random.default_rng(0).random((70, 100, 3)) * 255).astype("uint8")
    im = Image.fromarray(arr)
    t = ip.preprocess(im)
    assert t.shape == (1, 3, 64, 96) and -1.0 <= float(t.min()) and float(t.max()) <= 1.0     # sides rounded down to x16
    im2 = Image.fromarray(arr[:64, :96])
    t2 = ip.preprocess(im2)
    back = ip.postprocess(t2, "pil")[0]
    assert np.array_equal(np.asarray(back), arr[:64, :96])                                        # 8-bit exact
    assert ip.postprocess(t2, "np").shape == (1, 64, 96, 3) and ip.postprocess(t2, "pt").shape == (1, 3, 64, 96)


@torch.no_grad()
def test_condition_encode_and_decode_call_sites(tmp_path):
    """Condition(condition=PIL).encode(pipe) == preprocess -> vae.encode -> sample -> (z - shift) * scale -> pack, with
    the ids / position_delta / type id of the reference; and from_pretrained picks the `vae/` directory up."""
    from PIL import Image
    from safetensors.torch import save_file
    cfg = dict(GEOMS["hd128"])
    tr = M.FluxTransformer2DModel(**cfg)
    vae = V.init_synthetic_vae_(V.AutoencoderKL(**SMALL), seed=2).float()
    pipe = FluxPipeline(tr.float(), vae=vae)
    assert isinstance(pipe.image_processor, V.VaeImageProcessor)
    img = Image.fromarray((np.random.default_rng(1).random((64, 64, 3)) * 255).astype("uint8"))
    cond = Condition("cot", condition=img, position_delta=[0, -4])
    torch.manual_seed(11)
    tokens, ids, type_id = cond.encode(pipe)
    assert tokens.shape == (1, 16, 64) and ids.shape == (16, 3) and type_id.shape == (16, 1) and int(type_id[0]) == 12
    assert ids[:, 1].tolist() == [float(i // 4) for i in range(16)] and ids[:, 2].tolist() == [float(i % 4 - 4) for i in range(16)]
    # the same through the oracle's functions
    x = pipe.image_processor.preprocess(img)
    mo = VO.vae_encode_moments(vae.state_dict(), x, groups=8)
    torch.manual_seed(11)
    noise = torch.randn(1, 16, 8, 8)
    z = (VO.sample_latent(mo, noise) - 0.1159) * 0.3611
    assert torch.allclose(tokens, O.pack_latents(z, 1, 16, 8, 8), atol=1e-5)
    # decode call site (generate.py:302-307): unpack -> / scale + shift -> decode -> postprocess
    lat = torch.randn(1, 16, 64)
    zz = pipe._unpack_latents(lat, 64, 64, 8) / 0.3611 + 0.1159
    im = pipe.image_processor.postprocess(pipe.vae.decode(zz, return_dict=False)[0], output_type="pil")[0]
    ref = (VO.vae_decode(vae.state_dict(), zz, groups=8) / 2 + 0.5).clamp(0, 1)[0].permute(1, 2, 0).numpy()
    assert im.size == (64, 64) and np.abs(np.asarray(im).astype(np.float32) / 255 - ref).max() < 1.0 / 255 + 1e-4
    # diffusers-layout directory with transformer/ and vae/
    root = tmp_path / "ckpt"
    os.makedirs(root / "transformer")
    os.makedirs(root / "vae")
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, open(root / "transformer" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in tr.state_dict().items()}, str(root / "transformer" / "m.safetensors"))
    json.dump({**{k: (list(v) if isinstance(v, tuple) else v) for k, v in vae.config.items()}, "_class_name": "AutoencoderKL"},
              open(root / "vae" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in vae.state_dict().items()}, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    p2 = FluxPipeline.from_pretrained(str(root), torch_dtype=torch.float32)
    assert p2.vae is not None and all(torch.equal(a, b) for a, b in zip(p2.vae.state_dict().values(), vae.state_dict().values()))
