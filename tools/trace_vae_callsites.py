"""The reference's two VAE call sites on the HIP path (VERDICT r2 item 2 "done" criterion): `generate(output_type="pil")`
(generate.py:302-307) and the reflection round's decode -> resize -> encode hand-off (tts_reflectionflow.py:273-279 =
runner.candidate_condition + Condition.encode) with pipe.enable_hip_vae().  Run under `rocprofv3 --kernel-trace --stats`: the kernel
list must contain no MIOpen / convolution kernel of PyTorch.  FLUX.1-dev-shaped VAE (random init), a 2 + 2 block transformer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd.flux.generate import generate
from reflectionflow_amd.flux.pipeline import FluxPipeline
from reflectionflow_amd.tts import runner
dev = torch.device("cuda:0"); bf = torch.bfloat16
cfg = dict(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=256, pooled_projection_dim=64)
pipe = FluxPipeline.synthetic(cfg, seed=0, torch_dtype=bf, device=dev, with_vae=True)
pipe.enable_hip_vae()
g = torch.Generator().manual_seed(0)
pe = torch.randn(1, 64, 256, generator=g).to(dev).to(bf); pooled = torch.randn(1, 64, generator=g).to(dev).to(bf)
for rnd in range(2):
    imgs = generate(pipe, model_config={}, height=1024, width=1024, num_inference_steps=2, guidance_scale=3.5,
                    prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="pil").images
    lat = generate(pipe, model_config={}, height=1024, width=1024, num_inference_steps=2, guidance_scale=3.5,
                   prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent").images
    cond = runner.candidate_condition(pipe, lat, 1024, 1024, 512)
    tokens, ids, type_id = cond.with_generator(torch.Generator().manual_seed(1)).encode(pipe)
torch.cuda.synchronize()
print("pil", imgs[0].size, "condition tokens", tuple(tokens.shape), flush=True)
