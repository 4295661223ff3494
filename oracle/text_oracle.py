"""TEST INFRASTRUCTURE -- CPU restatement of the two text encoders FLUX conditions on, as plain functions over
Hugging Face `transformers`-layout state dicts.  Only tests/ may import this.

What the reference runs (train_flux/flux/generate.py:148-161 -> diffusers `FluxPipeline.encode_prompt`; per candidate and round in
tts/tts_reflectionflow.py:286-294):
    pooled_prompt_embeds = CLIPTextModel(clip_ids [B, 77]).pooler_output                     (CLIP-L: 12 layers, width 768)
    prompt_embeds        = T5EncoderModel(t5_ids [B, 512])[0]                                 (T5-v1.1-XXL: 24 layers, d_model 4096)
Neither call passes an attention mask: T5 attends over all 512 padded positions, CLIP is causal.

The algorithm lives in a third-party dependency: `transformers` (requirements.txt:2, unpinned).  **Parity pinned** against the
installed transformers 5.15.0: tests/golden/make_text_golden.py builds random-weight T5EncoderModel / CLIPTextModel instances from a
seeded state-dict recipe (`synthetic_t5_state` / `synthetic_clip_state` below), runs them in fp32 and stores inputs + outputs in
tests/golden/text_encoders.npz; tests/test_text_cpu.py rebuilds the same state dicts without transformers and checks this file
against those outputs (and, where transformers is importable, against the live modules).

Restated from transformers' modeling_t5.py (T5LayerNorm = RMS norm without mean subtraction or bias; T5Attention without 1/sqrt(d)
scaling and with the bucketed relative-position bias of layer 0 shared by all layers; T5DenseGatedActDense with gelu_new) and
modeling_clip.py (pre-LN blocks, causal mask, quick_gelu, final LayerNorm, pooled = hidden state at the EOS position).
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

T = torch.Tensor


# ---------------------------------------------------------------------------------------------------------------- T5 encoder
def t5_relative_position_bucket(rel: T, num_buckets: int = 32, max_distance: int = 128) -> T:
    """Bidirectional bucketing of rel = key position - query position (modeling_t5.py T5Attention._relative_position_bucket)."""
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def t5_position_bias(rel_emb: T, S: int, num_buckets: int = 32, max_distance: int = 128) -> T:
    """[H, S, S] additive attention bias from layer 0's `relative_attention_bias.weight` [num_buckets, H]."""
    ctx = torch.arange(S)[:, None]
    mem = torch.arange(S)[None, :]
    bucket = t5_relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return rel_emb.float()[bucket].permute(2, 0, 1).contiguous()


def t5_rms_norm(x: T, w: T, eps: float) -> T:
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w.float() * (x.float() * torch.rsqrt(var + eps))


def gelu_new(x: T) -> T:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def t5_encode(sd: Dict[str, T], ids: T, num_heads: int, eps: float = 1e-6, num_buckets: int = 32, max_distance: int = 128) -> T:
    """ids [B, S] -> last hidden state [B, S, d_model], fp32.  `sd` uses T5EncoderModel's keys."""
    sd = {k: v.float() for k, v in sd.items()}
    emb = sd["shared.weight"] if "shared.weight" in sd else sd["encoder.embed_tokens.weight"]
    h = emb[ids]
    B, S, D = h.shape
    bias = t5_position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], S, num_buckets, max_distance)
    L = 0
    while f"encoder.block.{L}.layer.0.layer_norm.weight" in sd:
        L += 1
    for i in range(L):
        p = f"encoder.block.{i}.layer.0."
        n = t5_rms_norm(h, sd[p + "layer_norm.weight"], eps)
        q, k, v = (n @ sd[p + f"SelfAttention.{x}.weight"].t() for x in "qkv")
        dk = q.shape[-1] // num_heads
        sp = lambda t: t.reshape(B, S, num_heads, dk).transpose(1, 2)  # noqa: E731
        w = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) + bias[None], dim=-1)        # no 1/sqrt(d): folded into the init
        o = (w @ sp(v)).transpose(1, 2).reshape(B, S, num_heads * dk)
        h = h + o @ sd[p + "SelfAttention.o.weight"].t()
        p = f"encoder.block.{i}.layer.1."
        n = t5_rms_norm(h, sd[p + "layer_norm.weight"], eps)
        g = gelu_new(n @ sd[p + "DenseReluDense.wi_0.weight"].t()) * (n @ sd[p + "DenseReluDense.wi_1.weight"].t())
        h = h + g @ sd[p + "DenseReluDense.wo.weight"].t()
    return t5_rms_norm(h, sd["encoder.final_layer_norm.weight"], eps)


# ---------------------------------------------------------------------------------------------------------------- CLIP text
def _strip(sd: Dict[str, T]) -> Dict[str, T]:
    """transformers < 5 prefixes the text tower's keys with `text_model.`; FLUX's text_encoder/model.safetensors does too."""
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v.float() for k, v in sd.items()}


def quick_gelu(x: T) -> T:
    return x * torch.sigmoid(1.702 * x)


def clip_text_encode(sd: Dict[str, T], ids: T, num_heads: int, eos_token_id: int = 2, eps: float = 1e-5, act: str = "quick_gelu"):
    """ids [B, S] -> (last_hidden_state [B, S, D] after the final LayerNorm, pooler_output [B, D]), fp32.
    eos_token_id == 2 is the legacy config (openai/clip-vit-large-patch14, what FLUX ships): pooled = the position of the LARGEST
    token id (EOS = 49407); otherwise the first position holding eos_token_id (modeling_clip.py CLIPTextTransformer.forward)."""
    sd = _strip(sd)
    B, S = ids.shape
    h = sd["embeddings.token_embedding.weight"][ids] + sd["embeddings.position_embedding.weight"][:S][None]
    D = h.shape[-1]
    dk = D // num_heads
    mask = torch.full((S, S), float("-inf")).triu(1)
    ln = lambda t, p: F.layer_norm(t, (D,), sd[p + ".weight"], sd[p + ".bias"], eps)  # noqa: E731
    lin = lambda t, p: t @ sd[p + ".weight"].t() + sd[p + ".bias"]  # noqa: E731
    fn = quick_gelu if act == "quick_gelu" else F.gelu
    L = 0
    while f"encoder.layers.{L}.layer_norm1.weight" in sd:
        L += 1
    for i in range(L):
        p = f"encoder.layers.{i}."
        n = ln(h, p + "layer_norm1")
        sp = lambda t: t.reshape(B, S, num_heads, dk).transpose(1, 2)  # noqa: E731
        q, k, v = sp(lin(n, p + "self_attn.q_proj") * dk ** -0.5), sp(lin(n, p + "self_attn.k_proj")), sp(lin(n, p + "self_attn.v_proj"))
        w = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        h = h + lin((w @ v).transpose(1, 2).reshape(B, S, D), p + "self_attn.out_proj")
        n = ln(h, p + "layer_norm2")
        h = h + lin(fn(lin(n, p + "mlp.fc1")), p + "mlp.fc2")
    last = ln(h, "final_layer_norm")
    if eos_token_id == 2:
        pos = ids.argmax(-1)
    else:
        pos = (ids == eos_token_id).int().argmax(-1)
    return last, last[torch.arange(B), pos]


# ------------------------------------------------------------------------------------- seeded state-dict recipes (golden fixtures)
def _fill(shapes: Dict[str, tuple], seed: int, scales: Dict[str, float]) -> Dict[str, T]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(shapes):
        t = torch.randn(shapes[k], generator=g)
        sc = next((v for pat, v in scales.items() if pat in k), 1.0)
        if k.endswith("layer_norm.weight") or ("layer_norm" in k and k.endswith(".weight")):
            t = 1.0 + 0.1 * t
        else:
            t = sc * t
        sd[k] = t
    return sd


def synthetic_t5_state(vocab: int, d_model: int, d_kv: int, heads: int, d_ff: int, layers: int, seed: int, num_buckets: int = 32):
    inner = d_kv * heads
    shapes = {"shared.weight": (vocab, d_model), "encoder.final_layer_norm.weight": (d_model,),
              "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": (num_buckets, heads)}
    for i in range(layers):
        p = f"encoder.block.{i}.layer."
        for x in "qkv":
            shapes[p + f"0.SelfAttention.{x}.weight"] = (inner, d_model)
        shapes[p + "0.SelfAttention.o.weight"] = (d_model, inner)
        shapes[p + "0.layer_norm.weight"] = (d_model,)
        shapes[p + "1.DenseReluDense.wi_0.weight"] = (d_ff, d_model)
        shapes[p + "1.DenseReluDense.wi_1.weight"] = (d_ff, d_model)
        shapes[p + "1.DenseReluDense.wo.weight"] = (d_model, d_ff)
        shapes[p + "1.layer_norm.weight"] = (d_model,)
    s = d_model ** -0.5
    sd = _fill(shapes, seed, {"SelfAttention.q": s * d_kv ** -0.25 * 2.0, "SelfAttention.k": s * d_kv ** -0.25 * 2.0, "SelfAttention.v": s,
                              "SelfAttention.o": (inner) ** -0.5, "wi_0": s, "wi_1": s, "wo": d_ff ** -0.5, "relative_attention_bias": 1.0,
                              "shared": 1.0})
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    return sd


def synthetic_clip_state(vocab: int, hidden: int, heads: int, inter: int, layers: int, max_pos: int, seed: int):
    shapes = {"embeddings.token_embedding.weight": (vocab, hidden), "embeddings.position_embedding.weight": (max_pos, hidden),
              "final_layer_norm.weight": (hidden,), "final_layer_norm.bias": (hidden,)}
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for x in ("q_proj", "k_proj", "v_proj", "out_proj"):
            shapes[p + f"self_attn.{x}.weight"] = (hidden, hidden)
            shapes[p + f"self_attn.{x}.bias"] = (hidden,)
        for n in ("layer_norm1", "layer_norm2"):
            shapes[p + n + ".weight"] = (hidden,)
            shapes[p + n + ".bias"] = (hidden,)
        shapes[p + "mlp.fc1.weight"] = (inter, hidden)
        shapes[p + "mlp.fc1.bias"] = (inter,)
        shapes[p + "mlp.fc2.weight"] = (hidden, inter)
        shapes[p + "mlp.fc2.bias"] = (hidden,)
    s = hidden ** -0.5
    return _fill(shapes, seed, {"q_proj.weight": 2.0 * s, "k_proj.weight": 2.0 * s, "v_proj.weight": s, "out_proj.weight": s, "fc1.weight": s,
                                "fc2.weight": inter ** -0.5, ".bias": 0.1, "token_embedding": 0.5, "position_embedding": 0.2})
