"""Operand / scale layout of the gfx950 block-scaled fp8 MFMAs, measured (see mx_probe.hip).
Runs a set of probe patterns on the GPU and saves every input and output to gpurun_out/mx_probe.pt;
`python tools/ubench/mx_probe.py --analyse gpurun_out/mx_probe.pt` (CPU) derives the maps.  Result (r02):
see profiles/r02_mx_probe.md."""
import ctypes as C, os, subprocess, sys
import torch

def D32(d):
    out = torch.zeros(32, 32)
    for l in range(64):
        for r in range(16):
            out[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = d[l, r]
    return out
def D16(d):
    out = torch.zeros(16, 16)
    for l in range(64):
        for r in range(4):
            out[4 * (l >> 4) + r, l & 15] = d[l, r]
    return out

def analyse(path):
    runs = torch.load(path)
    one8 = 0x38
    for name, R, Dfn, key in (("32x32x64", 32, D32, "d32"), ("16x16x128", 16, D16, "d16")):
        G = 64 // R
        print(f"==== {name}: rows = lane % {R}, {G} lane groups ====")
        # (1) which A bytes feed which k: with unit scales any consistent map works; skip.
        # (2) scale map: runs tagged 'sa_lo'/'sa_hi' use A = B = 1.0 everywhere except a mask on A, and per-lane A scales
        for r in runs:
            if not r["tag"].startswith("scale"):
                continue
            D = Dfn(r[key])                      # [R rows i, R cols j]; every column equal
            a_mask = (r["a8"].view(64, 32) == one8)
            sa = r["sa"] & 0xFF
            # expected under hypothesis H(l, b) -> scale lane: D[i] = sum over (l, b) with l % R == i, mask -> 2^(sa[H(l,b)] - 127)
            hyps = {
                "own lane": lambda l, b: l,
                "lane (l % R) [group 0 holds all scales of the row]": lambda l, b: l % R,
                "k-block of 32 contiguous-map: lane (l%R) + R*((32*(l//R)+b)//32 % G)": lambda l, b: l,
                "halves-interleaved map, block = (16*(l//R) + b%16 + 16*G*(b//16)) // 32": lambda l, b: (l % R) + R * (((16 * (l // R) + b % 16 + 16 * G * (b // 16)) // 32) % G),
                "8B-interleaved map, block = (8*(l//R) + b%8 + 8*G*(b//8)) // 32": lambda l, b: (l % R) + R * (((8 * (l // R) + b % 8 + 8 * G * (b // 8)) // 32) % G),
            }
            line = [f"{r['tag']:28s}"]
            for hn, H in hyps.items():
                exp = torch.zeros(R)
                for l in range(64):
                    for b in range(32):
                        if a_mask[l, b]:
                            exp[l % R] += 2.0 ** (int(sa[H(l, b)]) - 127)
                err = (D[:, 0] - exp).abs().max().item()
                line.append(f"{'OK ' if err < 1e-3 else 'no '}")
            print(" ".join(line), "   D[:4,0] =", [round(float(x), 3) for x in D[:4, 0]])
        print("   hypotheses order:", " | ".join(hyps))

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    analyse(sys.argv[2]); sys.exit(0)

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libmxprobe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mx_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.mx_probe.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def run(a8, b8, sa, sb, opa=0, opb=0):
    d32 = torch.zeros(64 * 16, device=dev); d16 = torch.zeros(64 * 4, device=dev)
    ad, bd, sad, sbd = a8.contiguous().to(dev), b8.contiguous().to(dev), sa.to(dev), sb.to(dev)
    rc = lib.mx_probe(ad.data_ptr(), bd.data_ptr(), d32.data_ptr(), d16.data_ptr(), sad.data_ptr(), sbd.data_ptr(), opa, opb)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return d32.cpu().view(64, 16), d16.cpu().view(64, 4)

runs = []
ones = torch.full((64, 32), 0x38, dtype=torch.uint8)     # fp8 e4m3 1.0
unit = torch.full((64,), 127, dtype=torch.int32)
lane = torch.arange(64)
for stag, sa in (("lo", (127 + (lane % 8) - 4).to(torch.int32)), ("hi", (127 + ((lane // 8) % 8) - 4).to(torch.int32))):
    masks = {"all": torch.ones(64, 32, dtype=torch.bool)}
    for g16 in range(4):
        m = torch.zeros(64, 32, dtype=torch.bool); m[16 * g16:16 * g16 + 16] = True; masks[f"lanes{16*g16}-{16*g16+15}"] = m
    for hb in range(2):
        m = torch.zeros(64, 32, dtype=torch.bool); m[:, 16 * hb:16 * hb + 16] = True; masks[f"bytes{16*hb}-{16*hb+15}"] = m
    for q8 in range(4):
        m = torch.zeros(64, 32, dtype=torch.bool); m[:, 8 * q8:8 * q8 + 8] = True; masks[f"bytes{8*q8}-{8*q8+7}"] = m
    for mn, m in masks.items():
        a8 = torch.where(m, ones, torch.zeros_like(ones))
        d32, d16 = run(a8, ones, sa, unit)
        runs.append(dict(tag=f"scale_{stag}_{mn}", a8=a8, b8=ones, sa=sa, sb=unit, d32=d32, d16=d16))
# random data, unit scales (A/B map sanity) and op_sel
def rand_fp8(n):
    b = torch.randint(0, 256, (n,), generator=g, dtype=torch.int32)
    b = torch.where((b & 0x7F) == 0x7F, b & 0x80, b)
    b = torch.where((b & 0x78) > 0x48, (b & 0x87) | 0x40, b)
    return b.to(torch.uint8)
a8, b8 = rand_fp8(64 * 32).view(64, 32), rand_fp8(64 * 32).view(64, 32)
d32, d16 = run(a8, b8, unit, unit)
runs.append(dict(tag="random_unit", a8=a8, b8=b8, sa=unit, sb=unit, d32=d32, d16=d16))
sa2 = ((127 + (lane % 3)) << 8 | 0x7F).to(torch.int32)
d32, d16 = run(ones, ones, sa2, unit, 1, 2)
runs.append(dict(tag="opsel_a1_b2", a8=ones, b8=ones, sa=sa2, sb=unit, d32=d32, d16=d16))
os.makedirs("gpurun_out", exist_ok=True)
torch.save(runs, "gpurun_out/mx_probe.pt")
print(f"saved {len(runs)} probe runs to gpurun_out/mx_probe.pt")
analyse("gpurun_out/mx_probe.pt")
