import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
try:
    t = torch.tensor([5, 2**63+5], dtype=torch.uint64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    torch.cuda.synchronize()
    print("uint64 MIN all_reduce ok", t)
except Exception as e:
    print("uint64 MIN all_reduce failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
