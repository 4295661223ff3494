"""FLUX's two text encoders on the HIP path (librf_flux.so: rf_t5_encode / rf_clip_text_encode, csrc/text.hip).

What the reference runs for every candidate's prompt (train_flux/flux/generate.py:148-161 -> diffusers FluxPipeline.encode_prompt;
per candidate and round in tts/tts_reflectionflow.py:286-294, where the prompt changes with every reflection):
    pooled_prompt_embeds = CLIPTextModel(clip_ids [B, 77]).pooler_output
    prompt_embeds        = T5EncoderModel(t5_ids [B, 512])[0]
`HipT5Encoder` / `HipClipTextEncoder` take a Hugging Face `transformers`-layout state dict (so text_encoder_2/*.safetensors and
text_encoder/model.safetensors load unchanged, with or without the `text_model.` prefix) and token ids; `HipTextEncoders` is the
`pipe.text_encoder` callable of flux/pipeline.py (prompt -> (prompt_embeds, pooled)) once a tokenizer callable is supplied --
vocabulary files are host-side data this repo does not ship.  No CPU fallback: CPU tensors or a missing library raise.

What stays in PyTorch is plumbing only: the weight repacking done once here (q|k and wi_0|wi_1 concatenated, 1/sqrt(64) folded into
CLIP's q projection, LayerNorm weights as (w - 1, b), v bias folded into the out-projection bias) and the integer work of the
attention biases (T5's relative-position buckets gathered once per sequence length; CLIP's causal mask).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Tuple

import torch

from .. import _lib as L
from ..ops import RFError, stream_ptr

BF = torch.bfloat16


def _pad32(n: int) -> int:
    return (n + 31) // 32 * 32


def t5_relative_position_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """transformers' bidirectional bucketing of rel = key position - query position (integer plumbing for the bias table)."""
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(n < max_exact, n, large)


class _Keep:
    def __init__(self, device):
        self.device = device
        self.keep: List[torch.Tensor] = []

    def t(self, x: torch.Tensor, dtype=BF) -> int:
        x = x.detach().to(self.device, dtype).contiguous()
        self.keep.append(x)
        return x.data_ptr()


class _Workspace:
    def __init__(self, device):
        self.device = device
        self.buf: Optional[torch.Tensor] = None

    def get(self, n: int) -> "L.rf_workspace":
        if n <= 0:
            raise RFError(f"workspace size query failed ({n})")
        if self.buf is None or self.buf.numel() < n:
            self.buf = torch.zeros(n, dtype=torch.uint8, device=self.device)
        ws = L.rf_workspace()
        ws.base, ws.bytes = self.buf.data_ptr(), self.buf.numel()
        return ws


class HipT5Encoder:
    """T5EncoderModel (T5 v1.1: gated gelu_new, no biases, RMS norms, bucketed relative-position bias shared by all layers)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_heads: int, device, eps: float = 1e-6, num_buckets: int = 32,
                 max_distance: int = 128):
        device = torch.device(device)
        if device.type != "cuda":
            raise RFError(f"HipT5Encoder: device {device}; the HIP path has no CPU fallback")
        L.load()
        sd = state_dict
        emb = sd["shared.weight"] if "shared.weight" in sd else sd["encoder.embed_tokens.weight"]
        n_layers = 0
        while f"encoder.block.{n_layers}.layer.0.layer_norm.weight" in sd:
            n_layers += 1
        if n_layers == 0:
            raise RFError("HipT5Encoder: no encoder.block.* keys in the state dict")
        inner = sd["encoder.block.0.layer.0.SelfAttention.q.weight"].shape[0]
        if inner != 64 * num_heads:
            raise RFError(f"HipT5Encoder: inner dim {inner} with {num_heads} heads: the attention kernel is built for d_kv = 64")
        self.device, self.heads, self.num_buckets, self.max_distance = device, num_heads, num_buckets, max_distance
        self.d_model = emb.shape[1]
        k = self._k = _Keep(device)
        self._layers = (L.rf_t5_layer * n_layers)()
        for i in range(n_layers):
            p0, p1 = f"encoder.block.{i}.layer.0.", f"encoder.block.{i}.layer.1."
            y = self._layers[i]
            y.ln0, y.ln1 = k.t(sd[p0 + "layer_norm.weight"]), k.t(sd[p1 + "layer_norm.weight"])
            y.w_qk = k.t(torch.cat([sd[p0 + "SelfAttention.q.weight"], sd[p0 + "SelfAttention.k.weight"]], 0))
            y.w_v, y.w_o = k.t(sd[p0 + "SelfAttention.v.weight"]), k.t(sd[p0 + "SelfAttention.o.weight"])
            y.w_wi = k.t(torch.cat([sd[p1 + "DenseReluDense.wi_0.weight"], sd[p1 + "DenseReluDense.wi_1.weight"]], 0))
            y.w_wo = k.t(sd[p1 + "DenseReluDense.wo.weight"])
        w = self._w = L.rf_t5_weights()
        w.layers, w.d_model, w.heads, w.d_kv = n_layers, self.d_model, num_heads, 64
        w.d_ff, w.vocab, w.eps = sd["encoder.block.0.layer.1.DenseReluDense.wo.weight"].shape[1], emb.shape[0], eps
        w.embed, w.final_ln = k.t(emb), k.t(sd["encoder.final_layer_norm.weight"])
        w.layer = C.cast(self._layers, C.POINTER(L.rf_t5_layer))
        self._rel = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().to(device, BF).float()   # [buckets, H]
        self._bias: Dict[int, torch.Tensor] = {}
        self._ws = _Workspace(device)

    def _pos_bias(self, S: int) -> torch.Tensor:
        if S not in self._bias:
            sp = _pad32(S)
            pos = torch.arange(sp, device=self.device)
            bucket = t5_relative_position_bucket(pos[None, :] - pos[:, None], self.num_buckets, self.max_distance)
            b = self._rel[bucket].permute(2, 0, 1).contiguous()            # [H, sp, sp] fp32
            b[:, :, S:] = float("-inf")
            if len(self._bias) >= 4:
                self._bias.pop(next(iter(self._bias)))
            self._bias[S] = b
        return self._bias[S]

    @torch.no_grad()
    def encode(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [B, S] (any integer dtype, on the device) -> last hidden state [B, S, d_model] bf16."""
        if not ids.is_cuda:
            raise RFError("HipT5Encoder.encode: ids are on the CPU; the HIP path has no CPU fallback")
        B, S = ids.shape
        lib, w = L.load(), self._w
        bias = self._pos_bias(S)
        w.pos_bias, w.bias_S = bias.data_ptr(), _pad32(S)
        ids32 = ids.to(torch.int32).contiguous()
        out = torch.empty(B, S, self.d_model, dtype=BF, device=self.device)
        for b0 in range(0, B, 64):                      # (the library takes up to 64 sequences per call: one GEMM launch over all their rows)
            nb = min(64, B - b0)
            ws = self._ws.get(lib.rf_t5_workspace_bytes(C.byref(w), nb, S))
            L.check(lib.rf_t5_encode(C.byref(w), ids32[b0].data_ptr(), nb, S, out[b0].data_ptr(), self.d_model, C.byref(ws), stream_ptr()), "rf_t5_encode")
        return out


class HipClipTextEncoder:
    """CLIPTextModel (pre-LN blocks, causal mask, quick_gelu): last_hidden_state and pooler_output."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_heads: int, device, eos_token_id: int = 2, eps: float = 1e-5):
        device = torch.device(device)
        if device.type != "cuda":
            raise RFError(f"HipClipTextEncoder: device {device}; the HIP path has no CPU fallback")
        L.load()
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()}
        n_layers = 0
        while f"encoder.layers.{n_layers}.layer_norm1.weight" in sd:
            n_layers += 1
        if n_layers == 0:
            raise RFError("HipClipTextEncoder: no encoder.layers.* keys in the state dict")
        D = sd["embeddings.token_embedding.weight"].shape[1]
        if D != 64 * num_heads:
            raise RFError(f"HipClipTextEncoder: hidden {D} with {num_heads} heads: the attention kernel is built for a head dim of 64")
        self.device, self.heads, self.hidden, self.eos_token_id = device, num_heads, D, eos_token_id
        k = self._k = _Keep(device)
        f = lambda t: t.detach().float()  # noqa: E731
        self._layers = (L.rf_clip_layer * n_layers)()
        for i in range(n_layers):
            p = f"encoder.layers.{i}."
            y = self._layers[i]
            y.ln1_scale, y.ln1_shift = k.t(f(sd[p + "layer_norm1.weight"]) - 1.0), k.t(sd[p + "layer_norm1.bias"])
            y.ln2_scale, y.ln2_shift = k.t(f(sd[p + "layer_norm2.weight"]) - 1.0), k.t(sd[p + "layer_norm2.bias"])
            y.w_qk = k.t(torch.cat([f(sd[p + "self_attn.q_proj.weight"]) * 0.125, f(sd[p + "self_attn.k_proj.weight"])], 0))
            y.b_qk = k.t(torch.cat([f(sd[p + "self_attn.q_proj.bias"]) * 0.125, f(sd[p + "self_attn.k_proj.bias"])], 0))
            y.w_v = k.t(sd[p + "self_attn.v_proj.weight"])
            wo = f(sd[p + "self_attn.out_proj.weight"])
            y.w_o, y.b_o = k.t(wo), k.t(f(sd[p + "self_attn.out_proj.bias"]) + wo @ f(sd[p + "self_attn.v_proj.bias"]))
            y.w_fc1, y.b_fc1 = k.t(sd[p + "mlp.fc1.weight"]), k.t(sd[p + "mlp.fc1.bias"])
            y.w_fc2, y.b_fc2 = k.t(sd[p + "mlp.fc2.weight"]), k.t(sd[p + "mlp.fc2.bias"])
        w = self._w = L.rf_clip_weights()
        w.layers, w.hidden, w.heads, w.inter = n_layers, D, num_heads, sd["encoder.layers.0.mlp.fc1.weight"].shape[0]
        w.vocab, w.max_pos, w.eps = sd["embeddings.token_embedding.weight"].shape[0], sd["embeddings.position_embedding.weight"].shape[0], eps
        w.tok_embed, w.pos_embed = k.t(sd["embeddings.token_embedding.weight"]), k.t(sd["embeddings.position_embedding.weight"])
        w.final_ln_scale, w.final_ln_shift = k.t(f(sd["final_layer_norm.weight"]) - 1.0), k.t(sd["final_layer_norm.bias"])
        w.layer = C.cast(self._layers, C.POINTER(L.rf_clip_layer))
        self._mask: Dict[int, torch.Tensor] = {}
        self._ws = _Workspace(device)

    def _causal(self, S: int) -> torch.Tensor:
        if S not in self._mask:
            sp = _pad32(S)
            m = torch.full((sp, sp), float("-inf"), device=self.device).triu(1)
            m[:, S:] = float("-inf")
            self._mask[S] = m.contiguous()
        return self._mask[S]

    def eos_positions(self, ids: torch.Tensor) -> List[int]:
        if self.eos_token_id == 2:                 # legacy config (what FLUX ships): the largest token id is EOS
            return ids.argmax(-1).tolist()
        return (ids == self.eos_token_id).int().argmax(-1).tolist()

    @torch.no_grad()
    def encode(self, ids: torch.Tensor, pool_in_library: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """ids [B, S] -> (last_hidden_state [B, S, hidden], pooler_output [B, hidden]), bf16.  pool_in_library: let rf_clip_text_encode
        copy the pooled rows itself from host-side EOS positions (costs a host sync to find them; tests)."""
        if not ids.is_cuda:
            raise RFError("HipClipTextEncoder.encode: ids are on the CPU; the HIP path has no CPU fallback")
        B, S = ids.shape
        lib, w = L.load(), self._w
        m = self._causal(S)
        w.mask, w.mask_S = m.data_ptr(), _pad32(S)
        ids32 = ids.to(torch.int32).contiguous()
        last = torch.empty(B, S, self.hidden, dtype=BF, device=self.device)
        if pool_in_library:
            eos = self.eos_positions(ids)
            pooled = torch.empty(B, self.hidden, dtype=BF, device=self.device)
        for b0 in range(0, B, 64):
            nb = min(64, B - b0)
            ws = self._ws.get(lib.rf_clip_text_workspace_bytes(C.byref(w), nb, S))
            pos_h = (C.c_int32 * nb)(*[int(e) for e in eos[b0:b0 + nb]]) if pool_in_library else None
            L.check(lib.rf_clip_text_encode(C.byref(w), ids32[b0].data_ptr(), nb, S, pos_h, last[b0].data_ptr(),
                                            pooled[b0].data_ptr() if pool_in_library else None, C.byref(ws), stream_ptr()), "rf_clip_text_encode")
        if pool_in_library:
            return last, pooled
        # pooler_output = the final-normed row at the EOS position: a device-side gather (no host sync; the C entry point takes host
        # positions for bindings that have them)
        pos = ids.argmax(-1) if self.eos_token_id == 2 else (ids == self.eos_token_id).int().argmax(-1)
        pooled = last[torch.arange(B, device=self.device), pos]
        return last, pooled


class HipTextEncoders:
    """`pipe.text_encoder` for flux/pipeline.py, i.e. FluxPipeline.encode_prompt's two encoder calls.
    Contract of the pipeline: `te(prompt: str, max_sequence_length, dtype, device) -> (prompt_embeds [L, D_t5], pooled [D_clip])`;
    `encode_t5` / `encode_clip` run one tower only (encode_prompt gives T5 `prompt_2` and CLIP `prompt`).  A list of prompts returns
    batched tensors.  `tokenize(prompts: list[str], max_sequence_length) -> (t5_ids [B, L], clip_ids [B, 77])` is the caller's (T5's
    SentencePiece model and CLIP's BPE vocabulary are files this repo does not ship)."""

    def __init__(self, t5: HipT5Encoder, clip: HipClipTextEncoder, tokenize: Callable):
        self.t5, self.clip, self._tokenize = t5, clip, tokenize
        self._memo = None          # (prompts, L) -> ids of the last tokenize() call

    def tokenize(self, prompts, max_sequence_length: int):
        """The caller's tokenizer runs BOTH towers' vocabularies per call; encode_prompt asks for the T5 ids (L = max_sequence_length)
        and the CLIP ids (L = 77) of the same prompts one after the other -- the T5 pass is memoised per prompt list so that one
        encode_prompt costs one SentencePiece pass at the T5 length, not two passes of each tokenizer."""
        key = (tuple(prompts), int(max_sequence_length))
        if self._memo is not None and self._memo[0] == key:
            return self._memo[1]
        out = self._tokenize(list(prompts), max_sequence_length)
        self._memo = (key, out)
        return out

    def encode_t5(self, prompt, max_sequence_length: int, dtype, device):
        single = isinstance(prompt, str)
        t5_ids, _ = self.tokenize([prompt] if single else list(prompt), max_sequence_length)
        pe = self.t5.encode(t5_ids.to(device)).to(dtype)
        return pe[0] if single else pe

    def encode_clip(self, prompt, dtype, device):
        single = isinstance(prompt, str)
        prompts = [prompt] if single else list(prompt)
        # CLIP ids do not depend on the T5 length: reuse the ids of the T5 call on the same prompts when there was one
        if self._memo is not None and self._memo[0][0] == tuple(prompts):
            clip_ids = self._memo[1][1]
        else:
            clip_ids = self.tokenize(prompts, 77)[1]
        pooled = self.clip.encode(clip_ids.to(device))[1].to(dtype)
        return pooled[0] if single else pooled

    def __call__(self, prompt, max_sequence_length: int, dtype, device):
        return self.encode_t5(prompt, max_sequence_length, dtype, device), self.encode_clip(prompt, dtype, device)
