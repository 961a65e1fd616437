"""Single-rank forced-dist step time by variant: plain (no dist), one-stream sharded, pipelined (priority / no priority)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
import torch, torch.distributed as dist
import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for n in (20_000_000, 100_000_000):
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
        t1 = time.perf_counter()
        fence()
        t2 = time.perf_counter()
        print("points %d %-28s submit %.1f us/step, total %.1f us/step" % (n, name, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K), flush=True)
    def plain():
        ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)
    run("plain", plain, ctx.synchronize)
    frame = pdist.DeviceFrame(ctx, 1920, 1080, dev); frame.bind()
    run("one-stream sharded", lambda: pdist.render_basic_sharded(ctx, frame, p, 1), lambda: (ctx.synchronize(), torch.cuda.synchronize()))
    frame.release()
    for prio in (True, False):
        pipe = pdist.PipelinedBasicRenderer(ctx, 1920, 1080, dev)
        if not prio:
            pipe.comm = torch.cuda.Stream(dev)
            pass
        run("pipelined prio=%s" % prio, lambda: pipe.step(p), lambda: (pipe.finish(), torch.cuda.synchronize()))
        pipe.release()
    ctx.close()
dist.destroy_process_group()
