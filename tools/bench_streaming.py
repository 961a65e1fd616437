#!/usr/bin/env python3
"""Progressive loading while rendering: the loader hands over tasks of <= 100 batch records per frame
(HuffmanLasLoader.cpp:106, 301-313) and a frame is drawn after every task, once with the copies in order with the frames on
the context's stream (the reference's behaviour) and once on the loader stream (pcr_set_async_upload). Prints one JSON
line: time until everything is resident, frames drawn meanwhile, and the frame times seen by the render loop.

    python tools/bench_streaming.py [--points 100000000] [--task 100]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--task", type=int, default=100, help="batch records per loader task")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import pcrhpg24_amd as P
    nthreads = args.threads or min(16, os.cpu_count() or 1)
    nb, st = P.synth_encode(args.points, 0x5EED, nthreads=nthreads)
    hf = P.HuffmanFile(nb)
    total = hf.numBatches
    blobs = [hf.blob(b) for b in range(total)]
    file_bytes = sum(len(b) for b in blobs)
    ctx = P.Context(0)
    ctx.set_image_size(args.width, args.height)
    p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), args.width, args.height)
    p.lod_percent, p.enable_frustum_culling = 100, 0

    def frame():
        ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)

    def run(mode):
        ctx.stream_begin(hf.header())
        ctx.set_async_upload(mode == "async")
        ctx.synchronize()
        frames, times, drawn = 0, [], []
        t0 = time.perf_counter()
        for b0 in range(0, total, args.task):
            t1 = time.perf_counter()
            ctx.upload_batches(b0, blobs[b0:b0 + args.task])
            t2 = time.perf_counter()
            if mode != "load_only":
                frame(); ctx.synchronize()          # the render loop presents every frame
                frames += 1; times.append(time.perf_counter() - t2); drawn.append(ctx.last_frame_batches)
        handed_over = time.perf_counter() - t0
        while ctx.batches_resident < total:
            if mode == "load_only":
                ctx.synchronize(); time.sleep(0.0002)
            else:
                t2 = time.perf_counter()
                frame(); ctx.synchronize()
                frames += 1; times.append(time.perf_counter() - t2); drawn.append(ctx.last_frame_batches)
        ctx.synchronize()
        resident = time.perf_counter() - t0
        frame(); ctx.synchronize()
        fb = ctx.read_framebuffer(full=True)
        ctx.set_async_upload(False)
        ctx.stream_unload()
        r = {"handed_over_s": round(handed_over, 4), "all_resident_s": round(resident, 4),
             "load_GBps": round(file_bytes / resident / 1e9, 2), "frames_while_loading": frames}
        if times:
            r["frame_ms_mean"] = round(1e3 * sum(times) / len(times), 3)
            r["frame_ms_max"] = round(1e3 * max(times), 3)
            r["batches_drawn_first_last"] = [drawn[0], drawn[-1]]
        return r, fb

    run("load_only")                                 # warm: pinned arenas, code objects
    out = {"metric": "progressive loading while rendering", "points": int(st["num_points"]), "batches": total,
           "file_bytes": file_bytes, "task_batches": args.task, "image": "%dx%d" % (args.width, args.height)}
    ref = None
    for mode in ("load_only", "sync", "async"):
        r, fb = run(mode)
        out[mode] = r
        if ref is None:
            ref = fb
        out[mode]["final_frame_identical"] = bool(np.array_equal(fb, ref))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
