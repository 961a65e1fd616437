"""What do torch.distributed collectives cost on a single-rank RCCL group (the only group a one-GPU box can form)?
Times reduce / all_reduce / all_to_all_single / all_gather_into_tensor on a 1080p u64 frame with torch events."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 1920 * 1080
a = torch.zeros(n, dtype=torch.int64, device=dev); b = torch.zeros(n, dtype=torch.int64, device=dev)
c = torch.zeros(n, dtype=torch.int32, device=dev); d = torch.zeros(n, dtype=torch.int32, device=dev)
ops = {
    "reduce_min_16.6MB": lambda: dist.reduce(a, dst=0, op=dist.ReduceOp.MIN),
    "all_reduce_min_16.6MB": lambda: dist.all_reduce(a, op=dist.ReduceOp.MIN),
    "all_to_all_single_16.6MB": lambda: dist.all_to_all_single(b, a),
    "all_gather_into_tensor_8.3MB": lambda: dist.all_gather_into_tensor(d, c),
    "tensor_copy_16.6MB": lambda: b.copy_(a),
}
for name, f in ops.items():
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50): f()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("%-32s gpu %.1f us/op, host submit %.1f us/op" % (name, 1e3 * e0.elapsed_time(e1) / 50, 1e6 * (t1 - t0) / 50))
dist.destroy_process_group()
