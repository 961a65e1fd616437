#!/usr/bin/env python3
"""Soak run (not part of the test suite): many seeded random cameras, image sizes and flags; every frame of the basic,
HQS and 10-10-10 methods must equal the oracle bit for bit. Prints one line per mismatch and a summary.

    python tools/soak/soak_parity.py [--frames 400] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pcrhpg24_amd as P          # noqa: E402
from tests import oracle, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--variant", choices=["auto", "words", "point_windows"], default="auto", help="decode variant (pcr_set_render_variant)")
    ap.add_argument("--points", type=int, default=2_000_000, help="points of the synthetic stream (2e7: 306 batches, ten prepass workgroups)")
    ap.add_argument("--parts", type=int, default=0, help="workgroups per batch (pcr_set_workgroup_parts): 0 = the library's choice, 1, 2")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    image, _ = scenes.synth_stream(args.points)
    of = oracle.OracleFile(image.view())
    hf = P.HuffmanFile(image)
    n = args.points
    x, y, z, c = P.synth_points(n, scenes.SEED, 0, n)
    las = P.synth_las_info(n, scenes.SEED)
    q = P.las_quantize(x, y, z, c, las)
    bad = 0
    t0 = time.time()
    sizes = [(640, 360), (1920, 1080), (333, 777), (64, 64), (1281, 53)]
    ctxs = {}
    for (w, h) in sizes:
        ctx = P.Context(0)
        if args.variant == "words":
            ctx.set_stream_layout(P.Context.LAYOUT_BOTH)      # both resident, the packed-words kernel forced
        ctx.set_render_variant({"auto": 0, "words": 1, "point_windows": 2}[args.variant])
        ctx.set_workgroup_parts(args.parts)
        ctx.set_image_size(w, h)
        ctx.stream_begin(hf.header(), 0)
        ctx.upload_batches(0, [hf.blob(b) for b in range(hf.numBatches)])
        ctx.las_begin(n)
        ctx.las_upload(0, *q)
        ctxs[(w, h)] = ctx
    for k in range(args.frames):
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        ctx = ctxs[(w, h)]
        yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-1.57, 1.0)
        radius = float(10.0 ** rng.uniform(-1.0, 4.5))
        target = (rng.uniform(-300, 1300), rng.uniform(-300, 1300), rng.uniform(-100, 200))
        p = P.camera_orbit(yaw, pitch, radius, target, w, h, fovy=float(rng.uniform(5, 150)),
                           near=float(10.0 ** rng.uniform(-3, 1)), far=float(10.0 ** rng.uniform(3, 6)))
        p = scenes.with_flags(p, lod_percent=int(rng.choice([0, 1, 10, 37, 100])), cull=int(rng.integers(0, 2)),
                              show_num_points=int(rng.integers(0, 2)), colorize_chunks=int(rng.integers(0, 2)))
        tag = f"frame {k} size {w}x{h} yaw {yaw:.3f} pitch {pitch:.3f} r {radius:.3f} lod {p.lod_percent} cull {p.enable_frustum_culling}"
        ctx.clear(); ctx.render_basic(p)
        ofb, ost = of.render_basic(p)
        if ctx.stats() != ost or not np.array_equal(ctx.read_framebuffer(full=True), ofb):
            bad += 1; print("MISMATCH basic:", tag, flush=True)
        ctx.clear(); ctx.render_hqs_depth(p)
        hfb, hst = of.render_hqs_depth(p)
        ok = ctx.stats() == hst and np.array_equal(ctx.read_framebuffer(full=True), hfb)
        ctx.render_hqs_color(p)
        org, oba, _ = of.render_hqs_color(p, hfb)
        rg, ba = ctx.read_accum(full=True)
        if not (ok and np.array_equal(rg, org) and np.array_equal(ba, oba)):
            bad += 1; print("MISMATCH hqs:", tag, flush=True)
        ctx.clear(); ctx.render_las(p)
        lfb, lst = oracle.render_las(*q[:4], p)
        if ctx.stats() != lst or not np.array_equal(ctx.read_framebuffer(full=True), lfb):
            bad += 1; print("MISMATCH las:", tag, flush=True)
        if k % 50 == 49:
            print(f"{k + 1} frames, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"soak: {args.frames} frames x 3 methods, {bad} mismatches")
    for ctx in ctxs.values():
        ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
