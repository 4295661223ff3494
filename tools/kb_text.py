"""Text encoders on one GPU: the HIP path (rf_t5_encode / rf_clip_text_encode) at FLUX's real shapes (T5-v1.1-XXL, 512 tokens;
CLIP-L, 77 tokens; random-init weights) with its per-kernel-class split, next to transformers' own modules in bf16 on the same GPU
(PyTorch-ROCm eager: hipBLASLt GEMMs + SDPA) where transformers is importable."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflectionflow_amd import ops
from reflectionflow_amd.flux.text_hip import HipT5Encoder
dev = torch.device("cuda:0"); bf = torch.bfloat16
print(json.dumps(bench.text_table(dev)), flush=True)
sd = bench.synthetic_t5_xxl_state(dev)
t5 = HipT5Encoder(sd, 64, dev)
ids = torch.randint(0, 32128, (1, 512), device=dev)
t5.encode(ids); torch.cuda.synchronize()
with ops.profile(4096) as pr:
    t5.encode(ids); torch.cuda.synchronize()
for k, v in pr.classes.items():
    extra = f"{v['work'] / (v['us'] * 1e-6) / 1e12:7.1f} TF" if k.startswith("gemm") or k == "attention" else f"{v['work'] / (v['us'] * 1e-6) / 1e9:7.0f} GB/s"
    print(f"  T5-XXL 512 tokens  {k:12s} {v['launches']:4d} launches {v['us'] / 1e3:7.3f} ms  {extra}", flush=True)
try:
    import transformers as tr
    cfg = tr.T5Config(vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64, feed_forward_proj="gated-gelu",
                      dense_act_fn="gelu_new", is_gated_act=True)
    torch.set_default_dtype(bf)
    with torch.device(dev):
        m = tr.T5EncoderModel(cfg).eval()
    torch.set_default_dtype(torch.float32)
    m.load_state_dict(sd, strict=False)
    with torch.no_grad():
        ref = m(input_ids=ids)[0]; m(input_ids=ids); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m(input_ids=ids)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out = t5.encode(ids)
    rel = float((out.float() - ref.float()).norm() / ref.float().norm())
    print(f"transformers {tr.__version__} T5EncoderModel bf16 on the same GPU (PyTorch-ROCm eager): {ms:.2f} ms per 512-token prompt; rel-L2(hip, that) = {rel:.3e}", flush=True)
    with torch.no_grad():
        ref32 = m.float()(input_ids=ids)[0]
    r = lambda a: float((a.float() - ref32).norm() / ref32.norm())  # noqa: E731
    print(f"  against transformers in fp32 on the GPU (same bf16-valued weights), all 24 layers: hip {r(out):.3e}   transformers-bf16 {r(ref):.3e}", flush=True)
except Exception as e:        # noqa: BLE001
    print("transformers baseline skipped:", repr(e)[:300])
