"""Stream-K vs one-tile-per-block on the FLUX GEMM shapes (cfg2: S=4608, cfg4: 512+4096+1024), plus power/clock
sampling through rocm-smi while a long GEMM loop runs.  Usage: python tools/kb_sk.py"""
import json, subprocess, threading, time
import torch
from reflectionflow_amd import _lib, ops
_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)

dev = torch.device("cuda:0")
lib = _lib.load()
BF = torch.bfloat16


def mk(rows, N, K, epi):
    groups = []
    for M in rows:
        x = torch.randn(M, K, device=dev, dtype=BF)
        W = torch.randn(N, K, device=dev, dtype=BF) * 0.02
        b = torch.randn(N, device=dev, dtype=BF)
        kw = {}
        if epi == ops.RF_EPI_GATE_RES:
            kw = dict(residual=torch.randn(M, N, device=dev, dtype=BF), gate=torch.randn(N, device=dev, dtype=BF))
        groups.append(ops.Group([ops.Seg(x, W)], bias=b, out=torch.empty(M, N, device=dev, dtype=BF), **kw))
    return groups


def run(name, rows, N, K, epi=ops.RF_EPI_STORE):
    g = mk(rows, N, K, epi)
    fl = 2.0 * sum(rows) * N * K
    res = {}
    for mode in (0, 1):
        lib.rf_debug_force_gemm_sk(mode)
        t = min(ops.time_gemm(g, N, epi, iters=20) for _ in range(3))
        res["sk" if mode else "dp"] = (t * 1e6, fl / t / 1e12, lib.rf_debug_last_gemm_path())
    lib.rf_debug_force_gemm_sk(-1)
    t = min(ops.time_gemm(g, N, epi, iters=20) for _ in range(3))
    res["auto"] = (t * 1e6, fl / t / 1e12, lib.rf_debug_last_gemm_path())
    print(f"{name:28s} " + "  ".join(f"{k}: {v[0]:7.1f}us {v[1]:6.0f}TF p{v[2]}" for k, v in res.items()), flush=True)


def smi_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip())
        except Exception as e:  # noqa
            out.append(f"ERR {e}")
        time.sleep(0.3)


if __name__ == "__main__":
    G = ops.RF_EPI_GATE_RES
    print("== exact fits (no split needed) ==")
    run("4096x4096x4096 (256 tiles)", (4096,), 4096, 4096)
    run("8192^3 (1024 tiles)", (8192,), 8192, 8192)
    print("== cfg2 (S=4608) ==")
    run("qkv 9216 K3072", (512, 4096), 9216, 3072)
    run("out 3072 K3072", (512, 4096), 3072, 3072, G)
    run("ff1 12288 K3072", (512, 4096), 12288, 3072)
    run("ff2 3072 K12288", (512, 4096), 3072, 12288, G)
    run("sgl_in 21504 K3072", (4608,), 21504, 3072)
    run("sgl_out 3072 K15360", (4608,), 3072, 15360, G)
    print("== cfg4 (S=5632) ==")
    run("qkv 9216 K3072", (512, 4096, 1024), 9216, 3072)
    run("out 3072 K3072", (512, 4096, 1024), 3072, 3072, G)
    run("ff1 12288 K3072", (512, 4096, 1024), 12288, 3072)
    run("ff2 3072 K12288", (512, 4096, 1024), 3072, 12288, G)
    run("sgl_in 21504 K3072", (4608, 1024), 21504, 3072)
    run("sgl_out 3072 K15360", (4608, 1024), 3072, 15360, G)
    # power / clock under a sustained GEMM loop
    stop, samples = threading.Event(), []
    th = threading.Thread(target=smi_sampler, args=(stop, samples))
    g = mk((8192,), 8192, 8192, ops.RF_EPI_STORE)
    lib.rf_debug_force_gemm_sk(0)
    th.start()
    t0 = time.time()
    while time.time() - t0 < 6:
        ops.time_gemm(g, 8192, ops.RF_EPI_STORE, iters=200)
    stop.set(); th.join()
    print("== rocm-smi samples during 8192^3 loop ==")
    for s in samples[:3] + samples[-6:]:
        print(s[:600])
