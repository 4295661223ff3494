"""Prompt -> token ids for the HIP text encoders, exactly as diffusers' FluxPipeline does it before its two encoder calls
(`_get_clip_prompt_embeds` / `_get_t5_prompt_embeds`, reached from train_flux/flux/generate.py:148-161):

    clip_ids = tokenizer(prompt, padding="max_length", max_length=tokenizer.model_max_length (77), truncation=True,
                         return_tensors="pt").input_ids                  # BOS ... EOS, padded with EOS (= the largest id: pooled at argmax)
    t5_ids   = tokenizer_2(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                           return_tensors="pt").input_ids                # ... </s> (1), padded with <pad> (0); no attention mask is used

Host-side plumbing only: the vocabularies are files of the checkpoint (`<root>/tokenizer/{vocab.json, merges.txt, ...}`,
`<root>/tokenizer_2/{spiece.model | tokenizer.json, ...}`) and the tokenizer classes are `transformers`' (the reference's own dependency,
requirements.txt:2).  `load_flux_tokenizers(root)` returns the `tokenize(prompts, max_sequence_length)` callable that
`HipTextEncoders` / `FluxPipeline.enable_hip_text_encoders` take; with no such directories the caller supplies its own callable.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence, Tuple

import torch


def make_tokenize(clip_tokenizer, t5_tokenizer) -> Callable[[Sequence[str], int], Tuple[torch.Tensor, torch.Tensor]]:
    """Wrap two transformers tokenizer objects (CLIPTokenizer[Fast], T5Tokenizer[Fast]) into the tokenize callable."""
    clip_len = int(min(getattr(clip_tokenizer, "model_max_length", 77), 77))

    def tokenize(prompts, max_sequence_length: int):
        prompts = [prompts] if isinstance(prompts, str) else list(prompts)
        clip_ids = clip_tokenizer(prompts, padding="max_length", max_length=clip_len, truncation=True, return_overflowing_tokens=False,
                                  return_length=False, return_tensors="pt").input_ids
        t5_ids = t5_tokenizer(prompts, padding="max_length", max_length=int(max_sequence_length), truncation=True, return_length=False,
                              return_overflowing_tokens=False, return_tensors="pt").input_ids
        return t5_ids, clip_ids
    return tokenize


def load_flux_tokenizers(root: str):
    """`<root>/tokenizer` (CLIP BPE) + `<root>/tokenizer_2` (T5 SentencePiece) of a diffusers FLUX checkpoint directory."""
    try:
        from transformers import AutoTokenizer
    except ImportError as e:      # pragma: no cover
        raise RuntimeError("load_flux_tokenizers needs the `transformers` package (the reference's own dependency) for its tokenizer classes") from e
    d1, d2 = os.path.join(root, "tokenizer"), os.path.join(root, "tokenizer_2")
    for d in (d1, d2):
        if not os.path.isdir(d):
            raise FileNotFoundError(f"no tokenizer directory at {d!r}: pass a tokenize(prompts, max_sequence_length) callable instead")
    return make_tokenize(AutoTokenizer.from_pretrained(d1), AutoTokenizer.from_pretrained(d2))
