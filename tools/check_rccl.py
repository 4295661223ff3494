import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(4, 2, device="cuda"); out = torch.empty(4, 2, device="cuda")
dist.all_gather_into_tensor(out, x); dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("RCCL single-rank ok", out.sum().item(), t.item(), dist.get_backend())
dist.destroy_process_group()
