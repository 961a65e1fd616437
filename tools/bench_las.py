#!/usr/bin/env python3
"""Measurement of the 10-10-10 path ("loop_las_cuda") on one GPU — not the headline bench.

    python tools/bench_las.py [--points 100000000] [--order tiles|strips] [--camera overview|closeup] [--steps 20]

Prints one JSON line: Mpoints/s of clear + render + resolve, the render kernel's HIP-event time and its HBM roofline
fraction (algorithmic bytes = 4/8/12 B per point by batch level + 64 B per drawn batch), and a full-size parity check
against the oracle (the checker, not the thing measured).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--order", choices=["tiles", "strips"], default="tiles")
    ap.add_argument("--camera", choices=["overview", "closeup"], default="overview")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cull", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import pcrhpg24_amd as P
    seed, n = 0x5EED, args.points
    t0 = time.time()
    x, y, z, c = P.synth_points(n, seed, 0, n)
    las = P.synth_las_info(n, seed)
    if args.order == "tiles":            # spatially coherent file order: ~65 536-point square tiles
        side = max(1, int(round(1_000_000 / max(1.0, (n / 65536) ** 0.5))))
        key = (y // side).astype(np.int64) * 4096 + x // side
        idx = np.argsort(key, kind="stable")
        x, y, z, c = x[idx], y[idx], z[idx], c[idx]
    t_gen = time.time() - t0

    ctx = P.Context(0)
    ctx.set_image_size(args.width, args.height)
    t0 = time.time()
    q = P.las_quantize(x, y, z, c, las)
    t_quant = time.time() - t0
    t0 = time.time()
    ctx.las_begin(n)
    step_b = 100
    nb = len(q[0])
    XB = type(q[0][0])
    for b0 in range(0, nb, step_b):
        b1 = min(nb, b0 + step_b)
        sub = (XB * (b1 - b0)).from_buffer(q[0], b0 * 64)
        ctx.las_upload(b0, sub, *(a[b0 * 65536:b1 * 65536] for a in q[1:]))
    t_load = time.time() - t0

    if args.camera == "overview":
        p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), args.width, args.height)
    else:
        p = P.camera_orbit(-1.68, -0.39, 70.0, (300.0, 20.0, 45.0), args.width, args.height)
    p.enable_frustum_culling = args.cull

    def step():
        ctx.clear(); ctx.render_las(p); ctx.resolve_las(p)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    ctx.kernel_timing(True)         # event pair around k_las_render of each timed step (pcr_kernel_timing_*)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    kernel_ms, _ = ctx.kernel_timing_read()
    ctx.kernel_timing(False)
    alg = ctx.las_algorithmic_bytes
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    parity = None
    if not args.no_parity:
        from tests import oracle
        ctx.clear(); ctx.render_las(p)
        t0 = time.perf_counter()
        ofb, ost = oracle.render_las(*q[:4], p)
        cpu_s = time.perf_counter() - t0
        parity = bool(np.array_equal(ctx.read_framebuffer(full=True), ofb)) and ost == ctx.stats()
    out = {"metric": "Mpoints/s rasterized @%dx%d (loop_las_cuda, 10-10-10)" % (args.width, args.height),
           "value": round(st["points_iterated"] / (elapsed / args.steps) / 1e6, 3), "unit": "Mpoints/s",
           "ms_per_step": round(1e3 * elapsed / args.steps, 4), "steps": args.steps, "warmup": args.warmup,
           "config": {"workload": "%d synthetic points in %s order, %dx%d, camera %s, cull=%d" %
                                  (n, args.order, args.width, args.height, args.camera, args.cull),
                      "batches": nb, "points_per_step": st["points_iterated"], "generate_s": round(t_gen, 2),
                      "quantize_s": round(t_quant, 2), "load_s": round(t_load, 2)},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "k_las_render",
                        "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": alg,
                        "bytes_per_point": round(alg / max(1, st["points_iterated"]), 3)},
           "parity_full_size": parity}
    if parity is not None:
        out["cpu_baseline"] = {"value": round(ost["points_iterated"] / cpu_s / 1e6, 3), "unit": "Mpoints/s", "cores": 1,
                               "kind": "port", "sample": "whole workload, oracle/pcr_oracle.c pcr_oracle_render_las, %.1f s" % cpu_s}
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
