#!/usr/bin/env python3
"""bench.py — headline benchmark of the Huffman decode + rasterize path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1e8 synthetic Morton-sorted points, per-batch Huffman-compressed
(1526 batches), 1920x1080, basic {depth,colour} atomicMin raster, LOD% = 100 and frustum culling off so
every point is decoded and rasterized. For N > 1 the scene grows to N x 1e8 points (weak
scaling), chunks are sharded contiguously over the ranks and every step ends with the merge of the partial
framebuffers over RCCL: by default an all-to-all of 1/N slices, a local min + resolve of the slice each rank owns and an
all-gather of the image (--merge reduce: one min-reduce of the whole frame to rank 0, resolved there; --merge allreduce).
A step = clear + decode/rasterize every loaded batch + (merge) + resolve, inputs resident in HBM.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 6553600
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=100_000_000, help="points per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--method", choices=["basic", "hqs"], default="basic")
    ap.add_argument("--lod", type=int, default=100, help="LOD percent (uPointFormat); 100 = all 64 points per chain")
    ap.add_argument("--cull", type=int, default=0)
    ap.add_argument("--camera", choices=["overview", "closeup"], default="overview")
    ap.add_argument("--merge", choices=["reduce", "allreduce", "a2a"], default="a2a",
                    help="multi-GPU exchange of the basic method: min-reduce the partial framebuffers to rank 0 (the display "
                         "rank), all-reduce them, or all-to-all slices + local min/resolve + all-gather of the image")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="keep the per-launch kernel events out of the timed steps (A/B of their overhead)")
    ap.add_argument("--cpu-sample-batches", type=int, default=0, help="0 = automatic (bounded)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for generation / CPU baseline")
    return ap.parse_args()


def camera(P, name, w, h):
    # the synthetic tile is 1 km x 1 km, heights 0..80 m (csrc/pcr_encoder.cpp Scene)
    if name == "overview":     # camera A: whole tile in the frustum
        return P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), w, h)
    return P.camera_orbit(-1.68, -0.39, 70.0, (300.0, 20.0, 45.0), w, h)   # camera B: close-up, heavy overdraw


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    import pcrhpg24_amd as P
    from pcrhpg24_amd import dist as pdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PCR_FORCE_DIST=1 exercises the multi-GPU code path (torch-owned int64-mergeable framebuffers, shared stream,
    # RCCL all-reduce) with a single rank, which is all a one-GPU box can run
    use_dist = world > 1 or os.environ.get("PCR_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nthreads = args.threads or min(os.cpu_count() or 8, 16)

    # ---- synthetic input: this rank's contiguous range of chunks of the global scene -------------------------
    total_points = args.points * world
    nchunks = -(-total_points // CHUNK)
    c0, cn = pdist.shard_range(nchunks, world, rank)
    first = c0 * CHUNK
    count = min(total_points, (c0 + cn) * CHUNK) - first
    t0 = time.time()
    image, enc = P.synth_encode(total_points, args.seed, first, count, CHUNK, nthreads)
    t_gen = time.time() - t0
    hf = P.HuffmanFile(image)

    # ---- load into HBM ----------------------------------------------------------------------------------------
    ctx = P.Context(local_rank)
    ctx.set_image_size(args.width, args.height)
    t0 = time.time()
    ctx.stream_begin(hf.header(), 0)
    for b0 in range(0, hf.numBatches, 100):          # loader tasks of <= 100 records, as HuffmanLasData::process
        ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
    if world > 1:
        # shard boundary: the words that follow this shard in the global stream
        nxt = pdist.exchange_shard_heads(*hf.head_words(0), dev)
        if nxt is not None:
            ctx.upload_tail(*nxt)
    ctx.synchronize()
    t_load = time.time() - t0

    p = camera(P, args.camera, args.width, args.height)
    p.lod_percent = args.lod
    p.enable_frustum_culling = args.cull

    frame, pipe = None, None
    if use_dist and args.method == "basic" and os.environ.get("PCR_NO_OVERLAP") != "1":
        pipe = pdist.PipelinedBasicRenderer(ctx, args.width, args.height, dev, merge=args.merge)   # merge of frame k overlaps render k+1
        step = lambda: pipe.step(p)
    else:
        if use_dist:
            frame = pdist.SlicedFrame(ctx, args.width, args.height, dev, world) if (args.merge == "a2a" and args.method == "basic") \
                else pdist.DeviceFrame(ctx, args.width, args.height, dev)
            frame.bind()
            ctx.clear()
        step = (lambda: pdist.render_basic_sharded(ctx, frame, p, world, merge=args.merge)) if args.method == "basic" else \
               (lambda: pdist.render_hqs_sharded(ctx, frame, p, world, merge=args.merge))

    def fence():
        if pipe is not None:
            pipe.finish()
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # first frame reported separately (one-time costs: code object upload, first launches); the lane-major transcode
    # itself is part of loading (pcr_upload_batches) and inside load_s
    fence()
    t0 = time.perf_counter()
    step()
    fence()
    first_frame_ms = 1e3 * (time.perf_counter() - t0)
    for _ in range(max(0, args.warmup - 1)):
        step()
    fence()
    launches_per_step = 1 if args.method == "basic" else 2
    ctx.kernel_timing(0 if args.no_kernel_events else max(1, args.steps * launches_per_step // 64) | 1)   # odd: samples both HQS passes
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if args.no_kernel_events:       # A/B of the event overhead: time the kernel in a second, untimed pass instead
        ctx.kernel_timing(True)
        for _ in range(min(args.steps, 64)):
            step()
        fence()

    st = ctx.stats()            # counters of the last render launch (per rank)
    pts = torch.tensor([st["points_iterated"]], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(pts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    points_per_step = float(pts.item())
    elapsed = float(tmax.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = points_per_step / (elapsed / args.steps) / 1e6

    # ---- dominant kernel: per-launch HIP event pairs recorded inside pcr_render_* on the context's stream, around
    # k_render only, during the timed steps above (pcr_kernel_timing_*; a stride keeps it to <=64 pairs spread over the
    # whole timed region, since an event pair costs ~5 us of stream time) ---------------------------------------------
    kernel_ms, kernel_launches = ctx.kernel_timing_read()
    ctx.kernel_timing(False)
    alg_bytes = ctx.algorithmic_bytes                      # decode-pass bytes of this rank's shard (SURVEY 8d B_dec * points)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get("points") == args.points and t.get("method") == args.method and t.get("width") == args.width:
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    hbm_read, hbm_copy = ctx.measure_hbm(2 << 30, 5) if rank == 0 else (0.0, 0.0)   # practical ceiling of this box
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "peak_measured": {"stream_read": round(hbm_read, 1), "stream_copy": round(hbm_copy, 1), "unit": "GB/s",
                                  "frac_of_read": round(achieved / hbm_read, 5) if hbm_read else None},
                "kernel": "k_render<%s>" % ("basic" if args.method == "basic" else "hqs_depth+hqs_color"),
                "kernel_ms": round(kernel_ms, 4), "kernel_launches_timed": kernel_launches, "algorithmic_bytes": alg_bytes,
                "bytes_per_point": round(alg_bytes / max(1, st["points_iterated"]), 4)}

    # ---- CPU baseline + full-size parity check (rank 0, N == 1 only) ----------------------------------------
    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        from tests import oracle
        of = oracle.OracleFile(hf.buf)
        nb = of.num_batches
        sample = args.cpu_sample_batches or nb
        q = p.copy()
        t0 = time.perf_counter()
        ofb, ost = of.render_basic(q, first=0, count=sample, nthreads=nthreads) if args.method == "basic" else \
            of.render_hqs_depth(q, first=0, count=sample)
        cpu_s = time.perf_counter() - t0
        cpu_baseline = {"value": round(ost["points_iterated"] / cpu_s / 1e6, 3), "unit": "Mpoints/s",
                        "cores": nthreads if args.method == "basic" else 1, "host_cores": os.cpu_count(), "kind": "port",
                        "sample": "%d of %d batches (%d points) of the same stream and camera, oracle/pcr_oracle.c, %.1f s wall"
                                  % (sample, nb, ost["points_iterated"], cpu_s)}
        if args.method == "basic" and nthreads > 1:       # single core as well, on a bounded part of the same stream
            one = max(1, min(nb, 64))
            t0 = time.perf_counter()
            _, ost1 = of.render_basic(p.copy(), first=0, count=one, nthreads=1)
            s1 = time.perf_counter() - t0
            cpu_baseline["single_core"] = {"value": round(ost1["points_iterated"] / s1 / 1e6, 3), "unit": "Mpoints/s",
                                           "sample": "first %d batches (%d points), %.1f s wall" % (one, ost1["points_iterated"], s1)}
        if sample == nb:        # same inputs end to end: compare the whole framebuffer, bit for bit
            ctx.clear(); (ctx.render_basic if args.method == "basic" else ctx.render_hqs_depth)(p)
            parity = bool(np.array_equal(ctx.read_framebuffer(full=True), ofb))

    if args.method == "hqs":
        merge_desc = "min all-reduce of the depth + sum %s of the colour sums" % ("all-reduce" if args.merge == "allreduce" else "reduce to rank 0")
    else:
        merge_desc = {"reduce": "min reduce to rank 0", "allreduce": "min all-reduce",
                      "a2a": "all-to-all of frame slices + local min/resolve + all-gather of the image"}[args.merge]
    if rank == 0:
        out = {
            "metric": "Mpoints/s decoded+rasterized @%dx%d" % (args.width, args.height),
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 keys / i32 deltas / f32 projection", "data": "synthetic",
            "config": {"workload": "%d synthetic Morton-sorted points per GPU, per-batch 12-bit-clipped Huffman, %dx%d, %s, camera %s, LOD%%=%d, cull=%d"
                                   % (args.points, args.width, args.height,
                                      "basic atomicMin raster" if args.method == "basic" else "HQS two-pass", args.camera, args.lod, args.cull),
                       "points_per_step": int(points_per_step), "batches_per_gpu": hf.numBatches,
                       "encoded_bits_per_point": round(8.0 * enc["encoded_bytes"] / enc["num_points"], 3),
                       "escape_fraction": round(enc["escaped_symbols"] / enc["total_symbols"], 5),
                       "parallelism": ("batch-sharded x%d + RCCL %s%s" % (world, merge_desc, " overlapped with the next frame" if pipe else "")) if use_dist else "single GPU",
                       "generate_s": round(t_gen, 2), "load_s": round(t_load, 2),
                       "first_frame_ms": round(first_frame_ms, 3)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "parity_full_size": parity,
        }
        print(json.dumps(out), flush=True)

    if pipe is not None:
        pipe.release()
    if frame is not None:
        frame.release()
    ctx.close()
    if use_dist:
        dist.barrier()              # rank 0 was busy measuring and printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
