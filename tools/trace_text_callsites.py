"""Kernel trace of the text-encoder call site (run under rocprofv3 --kernel-trace --stats): HipTextEncoders through
FluxPipeline.encode_prompt on small synthetic encoders + one T5-XXL-shaped layer stack -- every launch must be an rf:: kernel or a
torch element-wise copy; no hipBLASLt / rocBLAS / MIOpen kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import text_oracle as TO
from reflectionflow_amd.flux.pipeline import FluxPipeline
from reflectionflow_amd.flux.text_hip import HipT5Encoder
dev = torch.device("cuda:0"); BF = torch.bfloat16
def tokenize(prompts, L):
    t5 = torch.zeros(len(prompts), L, dtype=torch.long); clip = torch.full((len(prompts), 77), 127, dtype=torch.long)
    for i, p in enumerate(prompts):
        b = [3 + (c % 100) for c in p.encode()][: min(L, 77) - 1]
        t5[i, : len(b)] = torch.tensor(b); clip[i, : len(b)] = torch.tensor(b)
    return t5, clip
cfgt = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=256, pooled_projection_dim=64)
pipe = FluxPipeline.synthetic(cfgt, seed=0, torch_dtype=BF, device=dev)
pipe.enable_hip_text_encoders(TO.synthetic_t5_state(128, 256, 64, 4, 512, 2, seed=3), TO.synthetic_clip_state(128, 64, 1, 128, 2, 77, seed=4), tokenize, t5_heads=4, clip_heads=1)
pe, pooled, _ = pipe.encode_prompt(prompt=["a photo of a cat", "two dogs", "a red cube left of a blue ball"], max_sequence_length=64)
t5 = HipT5Encoder(bench.synthetic_t5_xxl_state(dev, layers=4), 64, dev)
out = t5.encode(torch.randint(0, 32128, (4, 512), device=dev))
torch.cuda.synchronize()
print("prompt_embeds", tuple(pe.shape), "pooled", tuple(pooled.shape), "T5-XXL-width x 4 layers", tuple(out.shape))
