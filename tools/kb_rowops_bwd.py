"""Times rf_layernorm_modulate_bwd / rf_gate_bwd at the training step's shapes (D = 3072; 512 / 1024 / 4096 / 4608 rows), operands
rotated over three buffer sets so that a launch does not find its inputs in the Infinity Cache of the launch before.
  python tools/kb_rowops_bwd.py [tag]"""
import sys
import torch
sys.path.insert(0, ".")
from reflectionflow_amd.train import kernels as K

BF, dev, D = torch.bfloat16, "cuda", 3072
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"


def timed(fn, sets, iters=60):
    for i in range(6):
        fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(*sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


g = torch.Generator(device=dev).manual_seed(5)
for rows in (512, 1024, 4096, 4608):
    mk = lambda: torch.randn(rows, D, generator=g, device=dev, dtype=torch.float32).to(BF)   # noqa: E731
    scale = (0.1 * torch.randn(D, generator=g, device=dev)).to(BF)
    sets = [(mk(), mk(), mk(), torch.empty(rows, D, dtype=BF, device=dev)) for _ in range(3)]
    t_ln = timed(lambda x, dy, dr, o: K.layernorm_modulate_bwd(x, dy, scale, dres=dr, out=o), sets)
    t_ln0 = timed(lambda x, dy, dr, o: K.layernorm_modulate_bwd(x, dy, scale, dres=None, out=o, need_dmod=False), sets)
    t_g = timed(lambda x, dy, dr, o: K.gate_bwd(dy, x, scale, out=o), sets)
    by = rows * D * 2
    print(f"{tag} rows {rows:5d}: ln_mod_bwd(+dres, +colsums) {t_ln:6.1f} us = {4 * by / t_ln / 1e6:5.2f} TB/s | no dres / no colsums "
          f"{t_ln0:6.1f} us = {3 * by / t_ln0 / 1e6:5.2f} TB/s | gate_bwd {t_g:6.1f} us = {3 * by / t_g / 1e6:5.2f} TB/s", flush=True)
