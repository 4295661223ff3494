"""CPU test of flux/tokenizers.py: the prompt -> ids plumbing in front of the HIP text encoders follows diffusers' FluxPipeline
conventions (generate.py:148-161 -> encode_prompt): CLIP ids = BOS ... EOS padded with EOS to 77 (pooled row = argmax of the ids), T5
ids = pieces + </s> padded with <pad> = 0 to max_sequence_length, both truncated.  The vocabularies are built here (a 64-piece
SentencePiece model trained on four sentences; a character-level CLIP BPE vocabulary without merges) and loaded through
`load_flux_tokenizers(root)` from a diffusers-style directory; skipped where transformers / sentencepiece are not importable."""
import json
import os

import pytest
import torch

tr = pytest.importorskip("transformers")
spm = pytest.importorskip("sentencepiece")


def _make_root(tmp_path):
    root = tmp_path / "flux"
    d1, d2 = root / "tokenizer", root / "tokenizer_2"
    d1.mkdir(parents=True)
    d2.mkdir(parents=True)
    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["a photo of a cat", "two dogs playing in the park", "a red cube left of a blue ball",
                                 "the quick brown fox jumps over the lazy dog"] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d2 / "spiece"), vocab_size=64, model_type="unigram", pad_id=0,
                                   eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False, minloglevel=2)
    # the directories as a diffusers FLUX checkpoint lays them out (files + tokenizer_config.json), not objects built in memory
    (d2 / "tokenizer_config.json").write_text(json.dumps({"tokenizer_class": "T5Tokenizer", "eos_token": "</s>", "unk_token": "<unk>",
                                                          "pad_token": "<pad>", "extra_ids": 0, "model_max_length": 512}))
    chars = "abcdefghijklmnopqrstuvwxyz"
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)            # the largest id, as in openai/clip-vit-large-patch14 (49407)
    (d1 / "vocab.json").write_text(json.dumps(vocab))
    (d1 / "merges.txt").write_text("#version: 0.2\n")
    (d1 / "tokenizer_config.json").write_text(json.dumps({"tokenizer_class": "CLIPTokenizer", "bos_token": "<|startoftext|>",
                                                          "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>",
                                                          "pad_token": "<|endoftext|>", "model_max_length": 77, "do_lower_case": True}))
    return str(root), vocab


def test_flux_tokenizer_conventions(tmp_path):
    from reflectionflow_amd.flux.tokenizers import load_flux_tokenizers
    root, vocab = _make_root(tmp_path)
    tokenize = load_flux_tokenizers(root)
    prompts = ["a photo of a cat", "two dogs " * 60]                 # the second one overflows both tokenizers
    t5_ids, clip_ids = tokenize(prompts, 32)
    assert t5_ids.shape == (2, 32) and clip_ids.shape == (2, 77) and t5_ids.dtype == torch.long
    bos, eos = vocab["<|startoftext|>"], vocab["<|endoftext|>"]
    # CLIP: BOS first, one EOS after the text, EOS as padding; the EOS id is the largest -> argmax finds the first of them
    assert (clip_ids[:, 0] == bos).all() and clip_ids[1, -1] == eos and int(clip_ids.max()) == eos
    n = len("aphotoofacat")                                            # character-level vocabulary: one id per letter
    assert clip_ids[0, 1 + n] == eos and (clip_ids[0, 1 + n:] == eos).all() and int(clip_ids[0].argmax()) == 1 + n
    # T5: </s> = 1 closes the text, <pad> = 0 fills the rest; a truncated row still ends with </s>
    row = t5_ids[0].tolist()
    k = row.index(1)
    assert all(v == 0 for v in row[k + 1:]) and all(v != 0 for v in row[:k]) and t5_ids[1, -1] == 1 and (t5_ids[1] != 0).all()
    # a single string and a different max_sequence_length
    a, b = tokenize("a red cube", 8)
    assert a.shape == (1, 8) and b.shape == (1, 77)


def test_missing_directories_fail_loudly(tmp_path):
    from reflectionflow_amd.flux.tokenizers import load_flux_tokenizers
    with pytest.raises(FileNotFoundError):
        load_flux_tokenizers(str(tmp_path))
