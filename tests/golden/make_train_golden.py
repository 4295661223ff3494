#!/usr/bin/env python3
"""Golden fixture of the TRAINING step -- build container only.

Imports the REFERENCE's own /root/reference/train_flux/train/model.py under stub namespaces (lightning, diffusers, peft,
torchvision, prodigyopt are absent; flux.* comes from make_golden's stubbed import of the reference's flux package), builds an
`OminiModel` without running its __init__ (which downloads FLUX) and calls the reference's `step(batch)` on the oracle's
hd128 2+2 transformer with LoRA -- the reference's own t / x_1 draws, its x_t, its call of its own tranformer_forward, its
mse_loss -- then back-propagates with torch autograd.  It asserts, bit for bit in fp32:

    reference loss          == oracle.train_oracle.training_step loss      (same draws from the same seeded global RNG)
    reference LoRA grads    == the restatement's LoRA grads                 (every factor)
    checkpointed (transformer.py:139-157, gradient_checkpointing = True)   == not checkpointed

and writes tests/golden/train_step_hd128.npz: inputs, draws, loss, prediction, every LoRA gradient.

    python tests/golden/make_train_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG          # noqa: E402  installs the diffusers / peft stubs and imports the reference's flux package
from oracle import flux_oracle as O   # noqa: E402
from oracle import train_oracle as TO  # noqa: E402

BATCH = {}


def load_reference_model_py():
    """train/model.py with its unavailable imports stubbed; encode_images / prepare_text_input hand back the fixture tensors
    (the VAE and the text encoders are outside this row)."""
    class _LM:                                     # lightning.LightningModule surface the class body needs
        pass
    MG._mod("lightning", LightningModule=_LM)
    MG._mod("torchvision", transforms=types.SimpleNamespace())
    MG._mod("torchvision.transforms")
    MG._mod("peft", LoraConfig=object, get_peft_model_state_dict=lambda m: {})
    MG._mod("prodigyopt")
    import flux.pipeline_tools as PT              # the reference's module (imported through make_golden's sys.path)

    def encode_images(pipe, images):
        return BATCH[images]

    def prepare_text_input(pipe, prompts, prompts_2=None, max_sequence_length=512):
        return BATCH["text"]
    PT.encode_images, PT.prepare_text_input = encode_images, prepare_text_input
    spec = importlib.util.spec_from_file_location("ref_train_model", os.path.join(MG.REF, "train", "model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    RM = load_reference_model_py()
    geom = "hd128"
    s = MG.SHAPES[geom]
    cfgm = MG.GEOMS[geom]
    St, Si, Sc = s["St"], s["gh"] * s["gw"], s["gc"] * s["gc"]
    gen = torch.Generator().manual_seed(77)
    x_0 = MG.rnd(gen, 2, Si, 64)                                       # batch of 2: the loss mean and the per-sample t both matter
    cond = MG.rnd(gen, 2, Sc, 64)
    pe = MG.rnd(gen, 2, St, cfgm["joint_attention_dim"])
    pooled = MG.rnd(gen, 2, cfgm["pooled_projection_dim"])
    txt_ids = torch.zeros(St, 3)
    img_ids = O.prepare_latent_image_ids(s["gh"], s["gw"])
    cond_ids0 = O.prepare_latent_image_ids(s["gc"], s["gc"])
    delta = [0, -s["gc"]]
    model_config = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}

    def fresh():
        m = MG.build(geom, lora=True)
        m.train()
        return TO.set_trainable(m)

    class _Imgs(str):
        """batch["image"] stands for the image tensor: the step reads `.shape[0]` off it (:185) and hands it to encode_images,
        whose stub looks the fixture tensors up by this key"""
        shape = (2,)

    def ref_step(ckpt: bool):
        m = fresh()
        m.gradient_checkpointing = ckpt
        self = object.__new__(RM.OminiModel)
        self.__dict__.update(flux_pipe=None, transformer=m, model_config=model_config, dtype=torch.float32)
        BATCH.clear()
        img_key, cond_key = _Imgs("IMG"), _Imgs("COND")
        BATCH.update({img_key: (x_0, img_ids), cond_key: (cond, cond_ids0.clone()), "text": (pe, pooled, txt_ids)})
        batch = {"image": img_key, "condition": cond_key, "condition_type": ["cot", "cot"], "original_prompt": ["p", "p"],
                 "position_delta": [delta], "description": ["d", "d"]}
        torch.manual_seed(1234)
        loss = RM.OminiModel.step(self, batch)
        loss.backward()
        return m, loss.detach()
    RM.OminiModel.device = property(lambda s_: torch.device("cpu"))

    m_ref, loss_ref = ref_step(False)
    m_ck, loss_ck = ref_step(True)

    # the restatement, with the same draws
    m_o = fresh()
    torch.manual_seed(1234)
    t, x_1 = TO.draw_t_x1(x_0)
    cond_ids = cond_ids0.clone()
    cond_ids[:, 1] += delta[0]
    cond_ids[:, 2] += delta[1]
    loss_o, pred_o = TO.training_step(m_o, x_0, img_ids, pe, pooled, txt_ids, cond, cond_ids, t, x_1, model_config)
    loss_o.backward()

    assert torch.equal(loss_ref, loss_o.detach()), (loss_ref, loss_o)
    assert torch.equal(loss_ref, loss_ck)
    g_ref, g_ck, g_o = (TO.lora_parameters(m) for m in (m_ref, m_ck, m_o))
    n_nonzero = 0
    for name in g_o:
        a, b, c = g_ref[name].grad, g_o[name].grad, g_ck[name].grad
        assert a is not None and b is not None, name
        assert torch.equal(a, b), f"{name}: reference grad != restatement (max diff {(a - b).abs().max():.3e})"
        assert torch.equal(a, c), f"{name}: checkpointed != plain"
        n_nonzero += int(a.abs().max() > 0)
    print(f"  loss {float(loss_ref):.8f}; {len(g_o)} LoRA factors, {n_nonzero} with a non-zero gradient: reference == restatement, bit-exact")

    out = dict(x_0=x_0, cond=cond, pe=pe, pooled=pooled, txt_ids=txt_ids, img_ids=img_ids, cond_ids=cond_ids, t=t, x_1=x_1,
               loss=loss_ref, pred=pred_o.detach(),
               weights_sha256=np.frombuffer(bytes.fromhex(MG.wsum(fresh())), dtype=np.uint8))
    for name, p in g_o.items():
        out["grad/" + name] = p.grad
    MG.save("train_step_hd128", **out)


if __name__ == "__main__":
    main()
