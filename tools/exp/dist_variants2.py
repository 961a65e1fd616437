"""Where do the ~60 us per step of the two-stream pipelined form go? Variants on a single rank."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29580")
import torch, torch.distributed as dist
import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 100_000_000
image, _ = P.synth_encode(n, 0x5EED, nthreads=16)
hf = P.HuffmanFile(image)
ctx = P.Context(0); ctx.set_image_size(1920, 1080); ctx.stream_begin(hf.header(), 0)
for b0 in range(0, hf.numBatches, 100):
    ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), 1920, 1080); p.lod_percent = 100; p.enable_frustum_culling = 0
K = 200
def run(name, step, fence):
    for _ in range(5): step()
    fence()
    t0 = time.perf_counter()
    for _ in range(K): step()
    fence()
    t2 = time.perf_counter()
    print("%-44s total %.1f us/step" % (name, 1e6 * (t2 - t0) / K), flush=True)
full = lambda: (ctx.synchronize(), torch.cuda.synchronize())
frames = [pdist.DeviceFrame(ctx, 1920, 1080, dev, accum=False) for _ in range(2)]
s0 = torch.cuda.Stream(dev); s1 = torch.cuda.Stream(dev)
k = [0]
def alt_one_stream():
    f = frames[k[0] & 1]; k[0] += 1
    f.bind(s0); ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)
run("one stream, two alternating frames", alt_one_stream, full)
def two_streams(comm_work):
    def step():
        i = k[0] & 1; k[0] += 1
        f = frames[i]
        ctx.fence_wait(2 + i, s0.cuda_stream)
        f.bind(s0); ctx.clear(); ctx.render_basic(p)
        ctx.fence_record(i, s0.cuda_stream)
        ctx.fence_wait(i, s1.cuda_stream)
        f.bind(s1)
        if comm_work == "resolve": ctx.resolve_basic(p)
        elif comm_work == "reduce+resolve":
            with torch.cuda.stream(s1):
                dist.reduce(f.fb, dst=0, op=dist.ReduceOp.MIN)
            ctx.resolve_basic(p)
        ctx.fence_record(2 + i, s1.cuda_stream)
    return step
fence2 = lambda: (s0.synchronize(), s1.synchronize(), torch.cuda.synchronize())
run("two streams, comm: nothing", two_streams("none"), fence2)
run("two streams, comm: resolve", two_streams("resolve"), fence2)
run("two streams, comm: reduce+resolve", two_streams("reduce+resolve"), fence2)
def same_stream_resolve_later():
    # software pipelining on ONE stream: frame k rendered, then frame k-1 resolved
    i = k[0] & 1; k[0] += 1
    frames[i].bind(s0); ctx.clear(); ctx.render_basic(p)
    frames[i ^ 1].bind(s0); ctx.resolve_basic(p)
run("one stream, resolve of the previous frame", same_stream_resolve_later, full)
ctx.close(); dist.destroy_process_group()
