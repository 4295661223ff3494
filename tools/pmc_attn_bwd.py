"""Driver for tools/pmc_attn_bwd.sh: one warm and one measured launch of the training attention (forward with row statistics,
then rf_attention_bwd) at two shapes, nothing else in the process."""
import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import ops
from reflectionflow_amd.train import kernels as K
dev = torch.device("cuda:0"); BF = torch.bfloat16
for H, S in ((24, 5632), (24, 2560)):
    g = torch.Generator(device=dev).manual_seed(S)
    raw = (torch.randn(S, 3 * H * 128, generator=g, device=dev) * 1.5).to(BF)
    w = torch.ones(128, device=dev).to(BF)
    cos = torch.cos(torch.rand(S, 128, generator=g, device=dev) * 6.28).contiguous()
    sin = torch.sin(torch.rand(S, 128, generator=g, device=dev) * 6.28).contiguous()
    a = K.qkv_train_fwd(raw, H, 0, (w, w, None, None), cos, sin)
    dout = (torch.randn(S, H * 128, generator=g, device=dev)).to(BF)
    for _ in range(2):
        lse = torch.empty(H, a.s_pad, dtype=torch.float32, device=dev)
        out = ops.attention(a.q, a.k, a.vt, S, q_prescaled=True, lse=lse)
        K.attention_bwd(a, out, dout, lse=lse)
    torch.cuda.synchronize()
