"""Shared helpers for golden-fixture tests: rebuild the synthetic models exactly as
tests/golden/make_golden.py did (same seeds, same geometry tables) WITHOUT the reference."""
import hashlib
import os

import numpy as np
import torch

from oracle import flux_oracle as O

GEOMS = {
    "hd32": dict(num_attention_heads=2, attention_head_dim=32, axes_dims_rope=(4, 14, 14),
                 joint_attention_dim=48, pooled_projection_dim=24, in_channels=64,
                 num_layers=2, num_single_layers=2),
    "hd128": dict(num_attention_heads=2, attention_head_dim=128, axes_dims_rope=(16, 56, 56),
                  joint_attention_dim=256, pooled_projection_dim=64, in_channels=64,
                  num_layers=2, num_single_layers=2),
}
SHAPES = {"hd32": dict(St=8, gh=4, gw=4, gc=2), "hd128": dict(St=32, gh=8, gw=8, gc=4)}
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.asarray(a)).float()


def wsum(model):
    h = hashlib.sha256()
    for k, v in sorted(model.state_dict().items()):
        h.update(k.encode())
        h.update(v.detach().float().numpy().tobytes())
    return h.hexdigest()


def build(geom, lora=False, seed=0):
    torch.manual_seed(1234)
    m = O.FluxTransformer2DModel(**GEOMS[geom]).float().eval()
    if lora:
        O.inject_lora(m, r=4, alpha=4.0)
    O.init_synthetic_(m, seed=seed, std=0.05)
    return m


BLOCK_MODES = {
    "nocond": (False, {}, None),
    "cond_union": (True, {"union_cond_attn": True}, None),
    "cond_nounion": (True, {"union_cond_attn": False}, None),
    "cond_cfactor": (True, {"union_cond_attn": True}, 1.5),
    "cond_addattn": (True, {"union_cond_attn": True, "add_cond_attn": True}, None),
}
