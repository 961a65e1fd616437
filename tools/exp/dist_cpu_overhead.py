"""How much host time does one pipelined distributed step take (single rank, forced)? submit-only vs. total."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
image, _ = P.synth_encode(n, 0x5EED, nthreads=16)
hf = P.HuffmanFile(image)
ctx = P.Context(0); ctx.set_image_size(1920, 1080); ctx.stream_begin(hf.header(), 0)
for b0 in range(0, hf.numBatches, 100):
    ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), 1920, 1080); p.lod_percent = 100; p.enable_frustum_culling = 0
pipe = pdist.PipelinedBasicRenderer(ctx, 1920, 1080, dev)
for _ in range(5): pipe.step(p)
pipe.finish(); torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K): pipe.step(p)
t1 = time.perf_counter()
pipe.finish(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("points %d: submit %.1f us/step, total %.1f us/step" % (n, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K))
pipe.release(); ctx.close(); dist.destroy_process_group()
