"""Three launches of the attention kernel at the cfg5 length (S = 17920, 24 heads) for a rocprofv3 --pmc pass (MFMA-busy, clock)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import ops
dev = torch.device("cuda:0")
S, H = 17920, 24
q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=30.0)
torch.cuda.synchronize()
