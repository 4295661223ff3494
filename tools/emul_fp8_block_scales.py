"""CPU emulation (no GPU): what per-(row, 32-column) E8M0 block scales on the ACTIVATIONS would buy over the per-token scales the
fp8 path uses, on one GEMM y = x W^T with K = 3072, for Gaussian activations and for activations with outlier
channels (a few columns 20x larger: the regime block scaling is made for).  Weights: per-output-channel e4m3 as shipped.
rel-L2 of y against fp32 for every combination; the weight-only and activation-only rows separate the two error sources."""
import torch
torch.manual_seed(0)
FP8_MAX = 448.0
def q_e4m3(x):
    return x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()
def act_per_token(x):
    s = x.abs().amax(-1, keepdim=True) / FP8_MAX
    return q_e4m3(x / s) * s
def act_block_e8m0(x, blk=32):
    M, K = x.shape
    xb = x.reshape(M, K // blk, blk)
    amax = xb.abs().amax(-1, keepdim=True).clamp(min=1e-30)
    e = torch.ceil(torch.log2(amax / FP8_MAX))           # power-of-two scale >= amax / 448
    s = torch.exp2(e)
    return (q_e4m3(xb / s) * s).reshape(M, K)
def w_per_channel(W):
    s = W.abs().amax(1, keepdim=True) / FP8_MAX
    return q_e4m3(W / s) * s
def rel(a, b):
    return float((a - b).norm() / b.norm())
M, K, N = 512, 3072, 3072
W = torch.randn(N, K) * K ** -0.5
rows = []
for name, x in (("gaussian activations", torch.randn(M, K)),
                ("8 outlier channels x20", torch.randn(M, K) * torch.where(torch.arange(K) % 384 == 7, 20.0, 1.0)),
                ("heavy-tailed (student-t, 3 dof)", torch.distributions.StudentT(3.0).sample((M, K)))):
    y = x @ W.t()
    Wq = w_per_channel(W)
    r = {"weights only (per-channel e4m3)": rel(x @ Wq.t(), y),
         "activations only, per-token": rel(act_per_token(x) @ W.t(), y),
         "activations only, E8M0 per 32": rel(act_block_e8m0(x) @ W.t(), y),
         "both, per-token (shipped)": rel(act_per_token(x) @ Wq.t(), y),
         "both, E8M0 per 32": rel(act_block_e8m0(x) @ Wq.t(), y)}
    rows.append((name, r))
print("| activations | " + " | ".join(rows[0][1].keys()) + " |")
print("|---|" + "---|" * len(rows[0][1]))
for name, r in rows:
    print(f"| {name} | " + " | ".join(f"{v:.2e}" for v in r.values()) + " |")
