#!/usr/bin/env python3
"""bench.py — headline benchmark of the Huffman decode + rasterize path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at N = 1 (BASELINE.json configs[1]): 1e8 synthetic Morton-sorted points, per-batch Huffman-compressed
(1526 batches), 1920x1080, basic {depth,colour} atomicMin raster, LOD% = 100 and frustum culling off so every point is
decoded and rasterized, stream resident in HBM in the context's default layout (point windows).
At N > 1 (configs[3]): 2e9 points in total (30 518 batches), contiguous BATCH ranges per rank (6 x 3815 + 2 x 3814 at
N = 8: strong scaling over N = 2, 4, 8; --points P makes it P points per GPU instead), every step ending with the merge of
the partial framebuffers over RCCL: by default a min-reduce of the u64 frame to the display rank (--merge allreduce /
sliced: the other forms). `scaling_base` in the line is the SAME 2e9-point stream on one GPU (a stored one-GPU run of this
bench): the 1e8-point line of N = 1 is another stream (other entropy) and no base for a scaling curve.
A step = clear + decode/rasterize every loaded batch + (merge) + resolve, inputs resident in HBM (one GPU: the resolve of a
frame, the clear and the next frame's cull/LOD prepass share one launch, pcr_frame_turn).
Before the W warm-up steps the frame loop runs for --preroll seconds (not counted as warm-up, not timed): the clocks of a
fresh box settle in that time, so a 20-step run reports what a 200-step run reports.
At N = 1 the line also carries `secondary`: the reference-default row (LOD 10 % + culling), the HQS method and
4096x4096 with culling (configs[2], configs[4]'s one-GPU half), 20 steps each on the same resident stream.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 6553600                    # points per Morton-sorted chunk = 100 batches (src/preprocess.cpp:1194-1200)
BATCH = 65536
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
POINTS_ONE_GPU = 100_000_000       # BASELINE.json configs[1]
POINTS_NODE = 2_000_000_000        # BASELINE.json configs[3]: the whole node's stream at N > 1
SCALING_BASE_FILE = os.path.join("profiles", "scaling_base_2e9_one_gpu.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preroll", type=float, default=1.5, help="seconds of untimed frames before the warm-up steps (clock settling)")
    ap.add_argument("--points", type=int, default=0, help="points per GPU; 0 = 1e8 on one GPU, 2e9 / N on N > 1 GPUs")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--method", choices=["basic", "hqs"], default="basic")
    ap.add_argument("--lod", type=int, default=100, help="LOD percent (uPointFormat); 100 = all 64 points per chain")
    ap.add_argument("--cull", type=int, default=0)
    ap.add_argument("--camera", choices=["overview", "closeup", "half"], default="overview")
    ap.add_argument("--layout", choices=["point_windows", "words"], default="point_windows",
                    help="HBM layout of the resident stream = decode variant of the timed steps (pcr_set_stream_layout)")
    ap.add_argument("--merge", choices=["reduce", "allreduce", "sliced", "sliced_p2p"], default="reduce",
                    help="multi-GPU exchange of the basic method: min-reduce the partial framebuffers to rank 0 (the display "
                         "rank), all-reduce them, or cut the frame into N slices: reduce-scatter (sliced_p2p, C++ layer only: all-to-all "
                         "+ local min), resolve of the own slice, gather of the image")
    ap.add_argument("--transport", choices=["torch", "rccl", "auto"], default="auto",
                    help="N > 1: who calls RCCL. auto (default): the C++ layer (include/pcr_dist.h: ncclUint64 min in place on the "
                         "context's stream -- north_star's 'host code stays C++') once one frame merged by it and one merged through "
                         "torch.distributed have produced the same image on rank 0, the torch transport otherwise; torch = torch.distributed "
                         "collectives on torch-owned int64 frames; rccl = the C++ layer unchecked")
    ap.add_argument("--batches", type=int, default=0, help="experiments: load only the first N batches of this rank's shard")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the second pass with the other stream layout")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary rows (LOD 10 % + cull, HQS, 4096x4096 + cull)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="keep the per-launch kernel events out of the timed steps (A/B of their overhead)")
    ap.add_argument("--cpu-sample-batches", type=int, default=0, help="0 = automatic (bounded)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for generation / CPU baseline")
    return ap.parse_args()


def camera(P, name, w, h):
    # the synthetic tile is 1 km x 1 km, heights 0..80 m (csrc/pcr_encoder.cpp Scene)
    if name == "overview":     # camera A: whole tile in the frustum
        return P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), w, h)
    if name == "half":         # camera C: the overview camera moved sideways and a little closer -- about half of the tile's batches (52 %)
        return P.camera_orbit(-0.15, -0.57, 1100.0, (-300.0, 500.0, 40.0), w, h)     # lie outside the frustum, the others fill the image
    return P.camera_orbit(-1.68, -0.39, 70.0, (300.0, 20.0, 45.0), w, h)   # camera B: close-up, heavy overdraw


def load_traffic(P, args):
    """HBM bytes per k_render launch from the committed PMC profile, per variant -- quoted only if the profile was taken on
    this kernel version and this workload (it is not measured by this run: the counters need rocprofv3 passes of their own)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    try:
        t = json.load(open(path))
    except Exception:
        return {}, "no profile"
    ver = P.kernel_version()
    if t.get("kernel_version") != ver:
        return {}, "profiles/pmc_traffic_latest.json is for kernel %s, this library is %s: not quoted" % (t.get("kernel_version"), ver)
    if (t.get("points"), t.get("method"), t.get("width"), t.get("lod"), t.get("cull"), t.get("camera")) != \
            (args.points, args.method, args.width, args.lod, args.cull, args.camera):
        return {}, "profiles/pmc_traffic_latest.json is for another workload: not quoted"
    return t.get("hbm_bytes_per_launch", {}), "profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, kernel %s; not measured by this run)" % ver


def las_row(P, device, args):
    """secondary.las: the 10-10-10 method (modules/compute_loop_las_cuda) on a tile-ordered cloud of the same 1e8
    synthetic points: clear + k_las_prepass + k_las_render + resolve per step, full-size parity against pcr_oracle_render_las."""
    import numpy as np
    n, seed = POINTS_ONE_GPU, args.seed
    t0 = time.time()
    x, y, z, c = P.synth_points(n, seed, 0, n)
    las = P.synth_las_info(n, seed)
    side = max(1, int(round(1_000_000 / max(1.0, (n / 65536) ** 0.5))))          # spatially coherent file order: ~65 536-point square tiles
    idx = np.argsort((y // side).astype(np.int64) * 4096 + x // side, kind="stable")
    x, y, z, c = x[idx], y[idx], z[idx], c[idx]
    q = P.las_quantize(x, y, z, c, las)
    t_prep = time.time() - t0
    ctx = P.Context(device)
    try:
        ctx.set_image_size(args.width, args.height)
        ctx.las_begin(n)
        nb = len(q[0])
        XB = type(q[0][0])
        for b0 in range(0, nb, 100):
            b1 = min(nb, b0 + 100)
            ctx.las_upload(b0, (XB * (b1 - b0)).from_buffer(q[0], b0 * 64), *(a[b0 * 65536:b1 * 65536] for a in q[1:]))
        p = camera(P, "overview", args.width, args.height)
        p.enable_frustum_culling = 0

        def step():
            ctx.clear(); ctx.render_las(p); ctx.resolve_las(p)
        for _ in range(5):
            step()
        ctx.synchronize()
        ctx.kernel_timing(1)
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        ctx.synchronize()
        e = time.perf_counter() - t0
        k_ms, _ = ctx.kernel_timing_read()
        ctx.kernel_timing(0)
        st = ctx.stats()
        alg = ctx.las_algorithmic_bytes
        parity = None
        if not args.no_cpu_baseline:
            from tests import oracle
            ctx.clear(); ctx.render_las(p)
            ofb, ost = oracle.render_las(*q[:4], p)
            parity = bool(np.array_equal(ctx.read_framebuffer(full=True), ofb)) and ost == ctx.stats()
        return {"what": "10-10-10 method (loop_las_cuda, reference modules/compute_loop_las_cuda/render.cu:204-327): %d points in tile order, %dx%d, "
                        "overview camera, cull 0" % (n, args.width, args.height),
                "method": "loop_las_cuda", "steps": 20, "ms_per_step": round(1e3 * e / 20, 4), "points_per_step": int(st["points_iterated"]),
                "Mpoints_per_s": round(st["points_iterated"] / (e / 20) / 1e6, 1), "kernel": "k_las_render", "kernel_ms": round(k_ms, 4),
                "algorithmic_bytes": alg, "bytes_per_point": round(alg / max(1, st["points_iterated"]), 3),
                "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms > 0 else None,
                "parity_full_size": parity, "prepare_s": round(t_prep, 2)}
    finally:
        ctx.close()


def encoder_row(P, device, args, nthreads):
    """secondary.encoder: the GPU encoder (include/pcr_gpu_encode.h, reference src/preprocess.cpp:233-801) on 2e7 points: host arrays in,
    .huffman image out (PCIe both ways inside the time), SHA-256 against the CPU encoder's image of the same points."""
    import hashlib
    n = 20_000_000
    x, y, z, c = P.synth_points(n, args.seed, 0, n)
    las = P.synth_las_info(n, args.seed)
    ctx = P.Context(device)
    try:
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            gpu, st = ctx.gpu_encode_points(x, y, z, c, las, morton_sort=True)
            best = min(best, time.perf_counter() - t0)
        row = {"what": "GPU encoder: %d synthetic points, Morton sort + per-batch Huffman + (time, lane) interleave + BC1; host arrays -> file image" % n,
               "points": n, "batches": int(st["num_batches"]), "seconds": round(best, 3), "Mpoints_per_s": round(n / best / 1e6, 1),
               "sha256": hashlib.sha256(gpu.view()).hexdigest()}
        if not args.no_cpu_baseline:
            t0 = time.perf_counter()
            cpu, st_cpu = P.encode_points(x, y, z, c, las, morton_sort=True, nthreads=nthreads)
            t_cpu = time.perf_counter() - t0
            row["cpu_encoder"] = {"seconds": round(t_cpu, 3), "Mpoints_per_s": round(n / t_cpu / 1e6, 1), "threads": nthreads}
            row["identical_to_cpu_encoder"] = hashlib.sha256(cpu.view()).hexdigest() == row["sha256"] and st_cpu == st
        return row
    finally:
        ctx.close()


def scaling_base(P, total_points, args):
    """The same stream on ONE GPU, from a stored run of this bench (`python bench.py --points 2000000000 > profiles/...`):
    what a 1 -> N curve of the N > 1 lines has to be set against."""
    path = os.path.join(ROOT, SCALING_BASE_FILE)
    try:
        d = json.load(open(path))
        c = d["config"]
        if d["n_gpus"] != 1 or c["points_per_step"] < total_points or (args.width, args.height) != (1920, 1080) or args.method != "basic":
            return {"value": None, "why": "%s holds another workload" % SCALING_BASE_FILE}
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "n_gpus": 1, "points_per_step": c["points_per_step"],
                "kernel_version": d["roofline"]["kernel_version"], "this_kernel_version": P.kernel_version(), "source": SCALING_BASE_FILE}
    except Exception as e:
        return {"value": None, "why": "%s unreadable: %s" % (SCALING_BASE_FILE, e)}


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
    # RCCL collective) with a single rank, which is all a one-GPU box can run
    use_dist = world > 1 or os.environ.get("PCR_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nthreads = args.threads or min(os.cpu_count() or 8, 16)

    # ---- synthetic input: this rank's contiguous range of BATCHES of the global scene ----------
    fixed_total = args.points == 0 and world > 1           # BASELINE configs[3]: 2e9 points over the node, whatever N is
    if args.points == 0:
        args.points = POINTS_ONE_GPU if world == 1 else POINTS_NODE // world
    total_points = POINTS_NODE if fixed_total else args.points * world
    nchunks = -(-total_points // CHUNK)
    nb_total = (nchunks - 1) * (CHUNK // BATCH) + -(-(total_points - (nchunks - 1) * CHUNK) // BATCH)    # every chunk is padded to whole batches
    first_b, count_b = pdist.shard_range(nb_total, world, rank)
    # the chunks that hold my batches and the batch that follows them (its head words close the shard: SURVEY B.4)
    c0 = first_b // (CHUNK // BATCH)
    c1 = min(nchunks, -(-(first_b + count_b + 1) // (CHUNK // BATCH)))
    gen_first = c0 * CHUNK
    t0 = time.time()
    image, enc = P.synth_encode(total_points, args.seed, gen_first, min(total_points, c1 * CHUNK) - gen_first, CHUNK, nthreads)
    t_gen = time.time() - t0
    hf = P.HuffmanFile(image)
    local_first = first_b - c0 * (CHUNK // BATCH)
    if args.batches:
        count_b = min(count_b, args.batches)
    has_follower = first_b + count_b < nb_total

    ctx = P.Context(local_rank)
    ctx.set_image_size(args.width, args.height)
    p = camera(P, args.camera, args.width, args.height)
    p.lod_percent = args.lod
    p.enable_frustum_culling = args.cull
    LAYOUTS = {"point_windows": P.Context.LAYOUT_POINT_WINDOWS, "words": P.Context.LAYOUT_WORDS}

    def load(layout):
        """Stream -> HBM in the given layout (loader tasks of <= 100 records, as HuffmanLasData::process)."""
        if ctx.batches_loaded:
            ctx.stream_unload()
        ctx.set_stream_layout(LAYOUTS[layout])
        t0 = time.time()
        ctx.stream_begin(hf.header(local_first, count_b), first_b)
        for b0 in range(0, count_b, 100):
            ctx.upload_batches(b0, [hf.blob(local_first + b) for b in range(b0, min(b0 + 100, count_b))])
        if has_follower:
            ctx.upload_tail(*hf.head_words(local_first + count_b))
        ctx.synchronize()
        return time.time() - t0

    t_load = load(args.layout)

    frame, native = None, None
    transport, transport_check = ("single GPU", None)

    def single_gpu_step(method, q):
        # One GPU: the steady frame loop is two launches per frame -- k_render (twice for HQS), then pcr_frame_turn = the
        # reference's RESOLVE + CLEAR (huffman_hqs.h:240-270) fused with the next frame's cull/LOD prepass. Same work as
        # clear + render + resolve, rotated: the loop is primed with one pcr_frame_begin (timed_run does it).
        if method == "basic":
            def step():
                ctx.render_basic(q)
                ctx.frame_turn(q, q)
        else:
            def step():
                ctx.render_hqs_depth(q)
                ctx.render_hqs_color(q)
                ctx.frame_turn(q, q, hqs=True)
        return step

    def torch_step_setup():
        """Frames owned by torch (int64-mergeable), collectives through torch.distributed, everything on torch's current stream."""
        fr = pdist.SlicedFrame(ctx, args.width, args.height, dev, world) if (args.merge == "sliced" and args.method == "basic") \
            else pdist.DeviceFrame(ctx, args.width, args.height, dev)
        fr.bind()
        ctx.clear()
        st = (lambda: pdist.render_basic_sharded(ctx, fr, p, world, merge=args.merge)) if args.method == "basic" else \
             (lambda: pdist.render_hqs_sharded(ctx, fr, p, world, merge=args.merge))
        return fr, st

    def image_hash():
        import hashlib
        return hashlib.sha256(ctx.read_rgba().tobytes()).hexdigest() if rank == 0 else ""

    prime = not use_dist
    if use_dist:
        if args.merge == "sliced_p2p" and args.transport == "torch":
            raise SystemExit("--merge sliced_p2p is a form of the C++ layer: use --transport rccl")
        want_native = args.transport in ("auto", "rccl")
        root = -1 if args.merge == "allreduce" else 0
        native_hash = None
        if want_native:
            try:
                native = pdist.NativeDist(ctx, rank, world, dev)
                native.set_exchange({"reduce": "reduce", "allreduce": "reduce"}.get(args.merge, args.merge))
                native_step = (lambda: native.frame_basic(p, root)) if args.method == "basic" else (lambda: native.frame_hqs(p, root))
                native_step()
                ctx.synchronize()
                native_hash = image_hash()
            except Exception as e:                       # RCCL not loadable / communicator refused: the torch transport still runs
                native, transport_check = None, "C++ RCCL layer unavailable: %s" % e
        ok = torch.tensor([1 if native is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # all ranks or none
        if not int(ok.item()):
            if native is not None:
                native.close()
            native = None
        if native is not None and args.transport == "auto" and args.merge != "sliced_p2p":
            frame, tstep = torch_step_setup()
            tstep()
            ctx.synchronize(); torch.cuda.synchronize()
            same = torch.tensor([1], device=dev)
            if rank == 0:
                import hashlib
                timg = frame.image().cpu().numpy()[:args.width * args.height] if isinstance(frame, pdist.SlicedFrame) else ctx.read_rgba()
                same[0] = 1 if hashlib.sha256(timg.tobytes()).hexdigest() == native_hash else 0
            dist.broadcast(same, 0)
            frame.release(); frame = None
            if int(same.item()):
                transport_check = "one frame merged by the C++ layer and one merged through torch.distributed: identical images on rank 0"
            else:
                transport_check = "C++ layer and torch.distributed disagreed on rank 0's merged image: torch transport used"
                native.close(); native = None
        if native is not None:
            step = (lambda: native.step_basic(p, root)) if args.method == "basic" else native_step
            prime = args.method == "basic"
            transport = "RCCL from C++ (include/pcr_dist.h, exchange %s, in place on the context's stream)" % native.set_exchange(
                {"reduce": "reduce", "allreduce": "reduce"}.get(args.merge, args.merge))
        else:
            frame, step = torch_step_setup()
            transport = "torch.distributed (int64-mergeable frames)"
    else:
        step = single_gpu_step(args.method, p)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(step, method, q, steps, warmup, preroll_s, kernel_events=True, prime=True):
        """(elapsed seconds of `steps` steps between fences, average k_render ms, launches timed, first-frame ms)."""
        launches_per_step = 1 if method == "basic" else 2
        if prime:
            ctx.frame_begin(q, hqs=method == "hqs")      # primes the turn-based loops (clear + prepass); untimed
        fence()
        t0 = time.perf_counter()
        step()
        fence()
        first_ms = 1e3 * (time.perf_counter() - t0)   # one-time costs: code object upload, release of the load-time buffers
        # clock pre-roll: frames until the wall clock says so (every rank runs the same count: rank 0 decides)
        if preroll_s > 0:
            t0 = time.perf_counter()
            while True:
                for _ in range(50):
                    step()
                ctx.synchronize()
                go_on = torch.tensor([1 if time.perf_counter() - t0 < preroll_s else 0], device=dev)
                if use_dist:
                    dist.broadcast(go_on, 0)
                if not int(go_on.item()):
                    break
        for _ in range(warmup):
            step()
        fence()
        # (every third launch at least: an event pair costs its step ~7 us, and a 20-step run that brackets every step reads 4 % slower
        # than a 200-step run that brackets every third)
        ctx.kernel_timing((max(3, steps * launches_per_step // 64) | 1) if kernel_events else 0)   # odd: samples both HQS passes
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if not kernel_events:       # A/B of the event overhead: time the kernel in a second, untimed pass instead
            ctx.kernel_timing(True)
            for _ in range(min(steps, 64)):
                step()
            fence()
        kernel_ms, launches = ctx.kernel_timing_read()
        ctx.kernel_timing(False)
        return elapsed, kernel_ms, launches, first_ms

    elapsed, kernel_ms, kernel_launches, first_frame_ms = timed_run(step, args.method, p, args.steps, args.warmup, args.preroll,
                                                                    not args.no_kernel_events, prime)

    # distribution of single steps (after the timed region, each step between its own event pair and a sync: launch gaps
    # included, pipelining across steps excluded)
    step_ms = []
    for _ in range(48):
        ctx.timing_begin()
        step()
        step_ms.append(ctx.timing_end())
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
    # decode-pass bytes of what this rank's launch drew (SURVEY 8d B_dec x points): the whole shard at LOD 100 % without
    # culling; with a level of detail or culling only the batches drawn, and of their words the share the LOD decodes
    full_frame = args.lod >= 100 and not args.cull
    alg_bytes = ctx.algorithmic_bytes if full_frame else ctx.last_frame_algorithmic_bytes
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic_by_variant, traffic_source = load_traffic(P, args) if world == 1 else ({}, "not quoted for N > 1")
    resident_main = ctx.resident_bytes
    hbm_read, hbm_copy = ctx.measure_hbm(2 << 30, 5) if rank == 0 else (0.0, 0.0)   # practical ceiling of this box
    kernel_name = "k_render<%s, %s>" % ("basic" if args.method == "basic" else "hqs_depth + hqs_color",
                                        "point windows" if args.layout == "point_windows" else "packed words")
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_by_variant.get(args.layout), "traffic_source": traffic_source,
                "peak_measured": {"stream_read": round(hbm_read, 1), "stream_copy": round(hbm_copy, 1), "unit": "GB/s",
                                  "frac_of_read": round(achieved / hbm_read, 5) if hbm_read else None},
                "kernel": kernel_name, "kernel_version": P.kernel_version(),
                "kernel_ms": round(kernel_ms, 4), "kernel_launches_timed": kernel_launches, "algorithmic_bytes": alg_bytes,
                "algorithmic_bytes_how": "every byte of the compressed stream once (pcr_stream_algorithmic_bytes)" if full_frame else
                                         "drawn batches only, words by the decoded share npr/64 of each chain (pcr_last_frame_algorithmic_bytes)",
                "bytes_per_point": round(alg_bytes / max(1, st["points_iterated"]), 4)}

    def variant_record(layout, ms_step, k_ms, resident):
        tr = traffic_by_variant.get(layout)
        return {"ms_per_step": round(ms_step, 4), "kernel_ms": round(k_ms, 4),
                "frac": round(alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "Mpoints_per_s": round(points_per_step / (ms_step * 1e-3) / 1e6, 1),
                "resident_bytes_per_point": round(resident / max(1, count_b * BATCH), 3),
                "hbm_bytes_read_per_point": round(tr / max(1, st["points_iterated"]), 3) if tr else None}

    variants = {args.layout: variant_record(args.layout, ms_per_step, kernel_ms, resident_main)}

    # ---- CPU baseline + full-size parity check (rank 0, N == 1 only) ----------------------------------------
    cpu_baseline, parity, ties = None, None, None
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
        cpu_cores = nthreads if args.method == "basic" else 1
        frames = 1
        while args.method == "basic" and cpu_s * cpu_cores < 10.0 and frames < 8:      # ~10-30 s of CPU work: the same frame again
            t0 = time.perf_counter()
            of.render_basic(p.copy(), first=0, count=sample, nthreads=nthreads)
            cpu_s += time.perf_counter() - t0
            frames += 1
        cpu_baseline = {"value": round(frames * ost["points_iterated"] / cpu_s / 1e6, 3), "unit": "Mpoints/s",
                        "cores": cpu_cores, "host_cores": os.cpu_count(), "kind": "port",
                        "sample": "%d frame(s) of %d of %d batches (%d points each) of the same stream and camera, oracle/pcr_oracle.c, %.1f s wall = %.0f core-seconds"
                                  % (frames, sample, nb, ost["points_iterated"], cpu_s, cpu_s * cpu_cores)}
        if args.method == "basic" and nthreads > 1:       # single core as well, on a bounded part of the same stream
            one = max(1, min(nb, 64))
            t0 = time.perf_counter()
            _, ost1 = of.render_basic(p.copy(), first=0, count=one, nthreads=1)
            s1 = time.perf_counter() - t0
            cpu_baseline["single_core"] = {"value": round(ost1["points_iterated"] / s1 / 1e6, 3), "unit": "Mpoints/s",
                                           "sample": "first %d batches (%d points), %.1f s wall" % (one, ost1["points_iterated"], s1)}
        if args.method == "basic" and (os.cpu_count() or 1) > nthreads:     # ... and on ALL host cores
            allc = os.cpu_count()
            t0 = time.perf_counter()
            _, osta = of.render_basic(p.copy(), first=0, count=sample, nthreads=allc)
            sa, fr = time.perf_counter() - t0, 1
            while fr < 3 or (sa * allc < 20.0 and fr < 8):       # (at least three frames: the first one pays for 256 thread starts)
                t0 = time.perf_counter()
                of.render_basic(p.copy(), first=0, count=sample, nthreads=allc)
                sa += time.perf_counter() - t0
                fr += 1
            # (where between 16 threads and all of them does the host stop scaling? two frames per thread count)
            sweep = {}
            for nt in (32, 64, 128):
                if nthreads < nt < allc:
                    of.render_basic(p.copy(), first=0, count=sample, nthreads=nt)
                    t0 = time.perf_counter()
                    of.render_basic(p.copy(), first=0, count=sample, nthreads=nt)
                    sweep[str(nt)] = round(osta["points_iterated"] / (time.perf_counter() - t0) / 1e6, 1)
            cpu_baseline["thread_sweep"] = sweep
            cpu_baseline["all_cores"] = {"value": round(fr * osta["points_iterated"] / sa / 1e6, 3), "unit": "Mpoints/s", "cores": allc,
                                         "sample": "%d frame(s) of %d batches, one thread per host core, one shared framebuffer (compare-exchange min), %.2f s wall"
                                                   % (fr, sample, sa)}
        if sample == nb:        # same inputs end to end: compare the whole framebuffer, bit for bit
            ctx.clear(); (ctx.render_basic if args.method == "basic" else ctx.render_hqs_depth)(p)
            parity = bool(np.array_equal(ctx.read_framebuffer(full=True), ofb))
        if args.method == "basic" and sample == nb and os.environ.get("PCR_BENCH_TIES") == "1":
            # pixels whose winning depth several points share, and those where the tied points differ in colour (there the
            # reference's own result depends on thread order); single-threaded second walk: ~2 min at 1e8
            a, b = of.count_depth_ties(p, ofb)
            ties = {"depth_tie_pixels": a, "depth_tie_pixels_other_colour": b}

    # ---- secondary rows, same resident stream (N == 1 only): 20 steps each --------------------------------------------------
    secondary = None
    if world == 1 and not use_dist and not args.no_secondary:
        secondary = {}
        rows = (("lod10_cull1", "basic", args.width, args.height, 10, 1, args.camera, "LOD 10 % + frustum culling: the reference's defaults (include/Debug.h:21-23)"),
                ("hqs", "hqs", args.width, args.height, 100, 0, args.camera, "HQS two-pass (BASELINE configs[2])"),
                ("4096_cull1", "basic", 4096, 4096, 100, 1, args.camera, "4096x4096 with culling (BASELINE configs[4], one GPU's view of it); this camera sees every batch"),
                ("4096_cull1_half", "basic", 4096, 4096, 100, 1, "half", "4096x4096 with culling, a camera that leaves about half of the batches outside the frustum: "
                                                                       "the prepass's ballot / prefix-sum compaction is on the timed path"))
        for name, method, w, h, lod, cull, cam, what in rows:
            if (w, h) != (args.width, args.height):
                ctx.set_image_size(w, h)
            q = camera(P, cam, w, h)
            q.lod_percent, q.enable_frustum_culling = lod, cull
            s_step = single_gpu_step(method, q)
            e, k_ms, _, _ = timed_run(s_step, method, q, 20, 3, 0.2)
            sst = ctx.stats()
            whole = lod >= 100 and sst["batches_culled"] == 0
            ab = ctx.algorithmic_bytes if whole else ctx.last_frame_algorithmic_bytes
            secondary[name] = {"what": what, "method": method, "width": w, "height": h, "lod_percent": lod, "cull": cull, "camera": cam, "steps": 20,
                               "ms_per_step": round(1e3 * e / 20, 4), "points_per_step": int(sst["points_iterated"]),
                               "Mpoints_per_s": round(sst["points_iterated"] / (e / 20) / 1e6, 1),
                               "batches_culled": int(sst["batches_culled"]),
                               "kernel_ms": round(k_ms, 4), "kernel_launches_per_step": 1 if method == "basic" else 2,
                               "algorithmic_bytes": ab, "frac": round(ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms > 0 else None}
            if (w, h) != (args.width, args.height):
                ctx.set_image_size(args.width, args.height)

    # ---- the other layout, same stream, same camera: a shorter second pass (N == 1 only) ---------------------------
    if world == 1 and not use_dist and not args.no_variants:
        other = "words" if args.layout == "point_windows" else "point_windows"
        load(other)
        n2 = min(args.steps, 100)
        e2, k2, _, _ = timed_run(single_gpu_step(args.method, p), args.method, p, n2, args.warmup, min(args.preroll, 0.3))
        variants[other] = variant_record(other, 1e3 * e2 / n2, k2, ctx.resident_bytes)

    # ---- the rows SURVEY 8f built next, on contexts of their own (N == 1 only): the 10-10-10 method and the GPU encoder -------------
    # (a failure in one of these rows must not cost the line its headline: it is recorded in the row's place)
    if world == 1 and not use_dist and not args.no_secondary and rank == 0:
        for name, row in (("las", lambda: las_row(P, local_rank, args)), ("encoder", lambda: encoder_row(P, local_rank, args, nthreads))):
            try:
                secondary[name] = row()
            except Exception as e:          # noqa: BLE001
                secondary[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    if args.method == "hqs":
        merge_desc = "min all-reduce of the depth + sum %s of the colour sums" % (
            "all-reduce" if args.merge == "allreduce" else "reduce-scatter, resolve per slice, gather of the image" if args.merge.startswith("sliced") and native is not None else "reduce to rank 0")
    else:
        merge_desc = {"reduce": "min reduce to rank 0", "allreduce": "min all-reduce",
                      "sliced": "frame slices: reduce-scatter (torch transport: all-to-all + local min), resolve of the own slice, gather of the image",
                      "sliced_p2p": "frame slices: all-to-all + local min, resolve of the own slice, gather of the image"}[args.merge]
    if rank == 0:
        layout_desc = {"point_windows": "stream resident as per-point 40-bit windows (5 B/point), decode variant point_windows",
                       "words": "stream resident as lane-major packed words (~3 B/point read), decode variant words"}[args.layout]
        out = {
            "metric": "Mpoints/s decoded+rasterized @%dx%d" % (args.width, args.height),
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if fixed_total else "weak", "vs_baseline": None,
            "dtype": "u64 keys / i32 deltas / f32 projection", "data": "synthetic",
            "config": {"workload": "%d synthetic Morton-sorted points %s, per-batch 12-bit-clipped Huffman, %dx%d, %s, camera %s, LOD%%=%d, cull=%d; %s"
                                   % (total_points if fixed_total else args.points, "in total over the node" if fixed_total else "per GPU",
                                      args.width, args.height,
                                      "basic atomicMin raster" if args.method == "basic" else "HQS two-pass", args.camera, args.lod, args.cull, layout_desc),
                       "points_per_step": int(points_per_step), "batches_per_gpu": count_b, "batches_total": nb_total,
                       "batches_culled_rank0": int(st["batches_culled"]),
                       "encoded_bits_per_point": round(8.0 * enc["encoded_bytes"] / enc["num_points"], 3),
                       "escape_fraction": round(enc["escaped_symbols"] / enc["total_symbols"], 5),
                       "parallelism": ("contiguous batch shards x%d + RCCL %s" % (world, merge_desc)) if use_dist else "single GPU",
                       "transport": transport, "transport_check": transport_check,
                       "rccl_ranks": native.comm_ranks() if native is not None else (dist.get_world_size() if use_dist else None),
                       "rccl_ranks_how": "ncclCommCount of the C++ layer's communicator" if native is not None else
                                         ("world size of torch.distributed's nccl group" if use_dist else None),
                       "generate_s": round(t_gen, 2), "load_s": round(t_load, 2), "preroll_s": args.preroll,
                       "first_frame_ms": round(first_frame_ms, 3)},
            "step_ms": {"min": round(min(step_ms), 4), "median": round(statistics.median(step_ms), 4), "max": round(max(step_ms), 4),
                        "n": len(step_ms), "how": "single steps after the timed region, one HIP event pair and one sync each"} if step_ms else None,
            "roofline": roofline,
            "variants": variants,
            "secondary": secondary,
            "scaling_base": scaling_base(P, total_points, args) if fixed_total else None,
            "cpu_baseline": cpu_baseline,
            "parity_full_size": parity,
            "depth_ties": ties,
        }
        print(json.dumps(out), flush=True)

    if frame is not None:
        frame.release()
    if native is not None:
        native.close()
    ctx.close()
    if use_dist:
        dist.barrier()              # rank 0 was busy measuring and printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
