#!/usr/bin/env python3
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import flux_oracle as O
from reflectionflow_amd import _lib as L, engine as E, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = L.load()
D, H, MLP = 3072, 24, 12288
Sm = 4608
pipe = bench.build_model(dev, dict(num_layers=0, num_single_layers=4), seed=0)
blocks = list(pipe.transformer.single_transformer_blocks)
pks = [E.pack_single_block(b) for b in blocks]
d = E.make_dims(D, H, MLP, 0, Sm, 0)
ws = E.get_workspace(dev, d)
wsbuf = E._WS_CACHE[(str(dev), torch.cuda.current_stream(dev).cuda_stream, D, H, MLP, 0, Sm, 0)]
def r256(n): return (n + 255) // 256 * 256
SD, HS = Sm * D * 2, H * Sm * 128 * 2
names = ["xn", "q", "k", "vt", "att", "hid", "lt", "x"]
sizes = [r256(SD), r256(HS), r256(HS), r256(HS), r256(SD), r256(Sm * MLP * 2), r256(Sm * 256 * 2), r256(SD)]
offs = [sum(sizes[:i]) for i in range(len(sizes))]
assert sum(sizes) == wsbuf.numel(), (sum(sizes), wsbuf.numel())
g = torch.Generator().manual_seed(1)
x0 = torch.randn(Sm, D, generator=g).to(dev).to(BF)
mods = [(torch.randn(3 * D, generator=g) * 0.5).to(dev).to(BF) for _ in blocks]
ids = torch.cat([torch.zeros(512, 3), O.prepare_latent_image_ids(64, 64)])
cos, sin = (t.to(dev).contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))
snaps = []
for rep in range(10):
    x = x0.clone()
    wsbuf.zero_()
    torch.cuda.synchronize()
    run = []
    for bi, pk in enumerate(pks):
        L.check(lib.rf_single_block_fwd(C.byref(d), C.byref(pk.struct), x.data_ptr(), None, D, mods[bi].data_ptr(), None,
                                        cos.data_ptr(), sin.data_ptr(), C.byref(ws), ops.stream_ptr()), "sgl")
        run.append((wsbuf.clone(), x.clone()))
    torch.cuda.synchronize()
    snaps.append(run)
for rep in range(1, 10):
    for bi in range(len(pks)):
        w0, x0_ = snaps[0][bi]; w1, x1_ = snaps[rep][bi]
        diffs = []
        for n, o, s in zip(names[:-1], offs, sizes):
            nd = int((w0[o:o + s] != w1[o:o + s]).sum())
            if nd: diffs.append((n, nd))
        ndx = int((x0_ != x1_).sum())
        if diffs or ndx:
            # locate in x
            idx = torch.nonzero((x0_ != x1_))
            rows = sorted(set(idx[:, 0].tolist()))[:8] if ndx else []
            cols = sorted(set(idx[:, 1].tolist()))[:8] if ndx else []
            print(f"rep {rep} after block {bi}: ws diffs (bytes) {diffs}  x diffs {ndx} rows {rows} cols {cols}", flush=True)
            break
    else:
        print(f"rep {rep}: identical", flush=True)
