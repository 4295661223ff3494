"""Generates tests/golden/text_encoders.npz: outputs of Hugging Face transformers' OWN T5EncoderModel / CLIPTextModel (the
third-party dependency the reference's text path runs through: requirements.txt:2 -> diffusers FluxPipeline.encode_prompt,
train_flux/flux/generate.py:148-161) on seeded random weights and token ids, in fp32.

Run here (transformers 5.15.0 is installed in this image):   python tests/golden/make_text_golden.py
The weights are NOT stored: oracle/text_oracle.py's `synthetic_t5_state` / `synthetic_clip_state` recipes rebuild them from the seeds
recorded in the file, so the fixture holds only token ids and output tensors (data, no source text).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import text_oracle as TO  # noqa: E402

import transformers  # noqa: E402
from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel  # noqa: E402

CASES_T5 = [  # name, vocab, d_model, d_kv, heads, d_ff, layers, S, seed
    ("t5_a", 128, 256, 64, 4, 512, 2, 40, 11),
    ("t5_b", 96, 128, 64, 2, 320, 3, 200, 12),      # S beyond max_distance = 128: the logarithmic buckets saturate
    ("t5_c", 64, 64, 16, 4, 128, 1, 7, 13),         # ragged tiny case, head dim 16 (oracle only)
]
CASES_CLIP = [  # name, vocab, hidden, heads, inter, layers, max_pos, S, eos_token_id, seed
    ("clip_a", 128, 256, 4, 512, 2, 77, 77, 2, 21),     # legacy eos id 2: pooled at argmax(ids)
    ("clip_b", 100, 128, 2, 256, 3, 32, 20, 99, 22),    # explicit eos id: pooled at its first occurrence
]


@torch.no_grad()
def main():
    out = {"transformers_version": np.array(transformers.__version__)}
    for name, vocab, d_model, d_kv, heads, d_ff, layers, S, seed in CASES_T5:
        cfg = T5Config(vocab_size=vocab, d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=layers, num_heads=heads,
                       feed_forward_proj="gated-gelu", dense_act_fn="gelu_new", is_gated_act=True, layer_norm_epsilon=1e-6,
                       relative_attention_num_buckets=32, relative_attention_max_distance=128)
        m = T5EncoderModel(cfg).eval().float()
        sd = TO.synthetic_t5_state(vocab, d_model, d_kv, heads, d_ff, layers, seed)
        missing = m.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all("embed_tokens" in k or "shared" in k for k in missing.missing_keys), missing
        ids = torch.randint(0, vocab, (2, S), generator=torch.Generator().manual_seed(seed + 100))
        ref = m(input_ids=ids)[0]
        mine = TO.t5_encode(sd, ids, heads)
        print(f"{name}: oracle vs transformers max|d| = {float((mine - ref).abs().max()):.3e}  (|ref| max {float(ref.abs().max()):.2f})")
        out[name + "_cfg"] = np.array([vocab, d_model, d_kv, heads, d_ff, layers, S, seed])
        out[name + "_ids"] = ids.numpy()
        out[name + "_out"] = ref.numpy()
    for name, vocab, hidden, heads, inter, layers, max_pos, S, eos, seed in CASES_CLIP:
        cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                             max_position_embeddings=max_pos, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=eos,
                             bos_token_id=0, pad_token_id=1)
        m = CLIPTextModel(cfg).eval().float()
        sd = TO.synthetic_clip_state(vocab, hidden, heads, inter, layers, max_pos, seed)
        keys = set(m.state_dict().keys())
        pref = "text_model." if any(k.startswith("text_model.") for k in keys) else ""
        m.load_state_dict({pref + k: v for k, v in sd.items()}, strict=True)
        g = torch.Generator().manual_seed(seed + 100)
        ids = torch.randint(3, vocab - 1, (2, S), generator=g)
        for b, pos in enumerate((S // 3, S - 1)):                   # one EOS per row (the largest id), padding of the same id after it
            ids[b, pos:] = vocab - 1
        o = m(input_ids=ids)
        last, pooled = TO.clip_text_encode(sd, ids, heads, eos_token_id=eos)
        print(f"{name}: oracle vs transformers max|d| hidden {float((last - o.last_hidden_state).abs().max()):.3e} "
              f"pooled {float((pooled - o.pooler_output).abs().max()):.3e}")
        out[name + "_cfg"] = np.array([vocab, hidden, heads, inter, layers, max_pos, S, eos, seed])
        out[name + "_ids"] = ids.numpy()
        out[name + "_last"] = o.last_hidden_state.numpy()
        out[name + "_pooled"] = o.pooler_output.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_encoders.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
