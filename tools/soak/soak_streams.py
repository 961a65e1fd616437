#!/usr/bin/env python3
"""Soak run over STREAMS (not part of the test suite): seeded random point clouds of different statistics (smooth,
clustered, uniform noise with many escapes, wide deltas, int32 wrap-around, constant), encoded with random flags by the
CPU or the GPU encoder, each drawn from a few random cameras with the basic and HQS methods and compared with the oracle.

    python tools/soak/soak_streams.py [--streams 200] [--seed 1]
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


def cloud(rng, kind, n):
    if kind == 0:      # smooth surface
        u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        x, y = (u * 400000).astype(np.int64), (v * 400000).astype(np.int64)
        z = (20000 * np.sin(u * 9) * np.cos(v * 7) + rng.normal(0, 3, n)).astype(np.int64)
    elif kind == 1:    # clusters
        k = rng.integers(2, 30)
        cen = rng.integers(-500000, 500000, (k, 3))
        idx = rng.integers(0, k, n)
        p = cen[idx] + rng.normal(0, rng.uniform(5, 5000), (n, 3))
        x, y, z = p[:, 0].astype(np.int64), p[:, 1].astype(np.int64), p[:, 2].astype(np.int64)
    elif kind == 2:    # uniform noise: huge alphabet, escape heavy
        s = int(10 ** rng.uniform(2, 6.5))
        x, y, z = (rng.integers(-s, s, n) for _ in range(3))
    elif kind == 3:    # wide hops with short codes
        hop = np.where(rng.integers(0, 2, n) == 0, 0, 1 << int(rng.integers(18, 31)))
        x = hop + rng.integers(0, 4, n); y = rng.integers(0, 3000, n); z = rng.integers(0, 40, n)
    elif kind == 4:    # full int32 range (wrap-around deltas)
        x, y, z = (rng.integers(-2 ** 31, 2 ** 31, n) for _ in range(3))
    else:              # nearly constant
        x = np.full(n, 7) + (rng.integers(0, 2, n) if rng.integers(0, 2) else 0); y = np.full(n, -3); z = np.full(n, 11)
    c = rng.integers(0, 1 << 24, n).astype(np.uint32)
    x, y, z = (np.asarray(a, np.int64).astype(np.int32) for a in (x, y, z))
    las = P.LasInfo()
    sc = float(10.0 ** rng.uniform(-4, -1))
    for k_, a in enumerate((x, y, z)):
        las.scale[k_] = sc
        las.offset[k_] = float(rng.uniform(-100, 100))
        las.min[k_] = float(a.min()) * sc + las.offset[k_]
        las.max[k_] = float(a.max()) * sc + las.offset[k_]
    return x, y, z, c, las


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--variant", choices=["auto", "words", "point_windows"], default="auto", help="decode variant (pcr_set_render_variant)")
    ap.add_argument("--parts", type=int, default=0, help="workgroups per batch (pcr_set_workgroup_parts): 0 = the library's choice, 1, 2")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    W, H = 480, 270
    ctx = P.Context(0)
    if args.variant == "words":
        ctx.set_stream_layout(P.Context.LAYOUT_BOTH)          # both resident, the packed-words kernel forced
    ctx.set_render_variant({"auto": 0, "words": 1, "point_windows": 2}[args.variant])
    ctx.set_workgroup_parts(args.parts)
    ctx.set_image_size(W, H)
    bad = enc_bad = 0
    t0 = time.time()
    for s in range(args.streams):
        kind = int(rng.integers(0, 6))
        n = int(rng.choice([5, 1000, 65536, 70000, 131072, 200000]))
        x, y, z, c, las = cloud(rng, kind, n)
        sort, pad = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        image, st = P.encode_points(x, y, z, c, las, morton_sort=sort, pad_tails=pad, nthreads=4)
        if rng.integers(0, 3) == 0:                       # the GPU encoder must agree byte for byte
            gimg, gst = ctx.gpu_encode_points(x, y, z, c, las, morton_sort=sort, pad_tails=pad)
            if bytes(gimg.view()) != bytes(image.view()) or gst != st:
                enc_bad += 1; print(f"ENCODER MISMATCH stream {s} kind {kind} n {n} sort {sort} pad {pad}", flush=True)
        of = oracle.OracleFile(image.view())
        hf = P.HuffmanFile(image)
        if ctx.batches_loaded:
            ctx.stream_unload()
        ctx.stream_begin(hf.header(), 0)
        ctx.upload_batches(0, [hf.blob(b) for b in range(hf.numBatches)])
        ext = max(1.0, float(max(las.max[k] - las.min[k] for k in range(3))))
        for f in range(4):
            tgt = tuple(float(rng.uniform(0, 1) * (las.max[k] - las.min[k])) for k in range(3))
            p = P.camera_orbit(rng.uniform(-3.14, 3.14), rng.uniform(-1.5, 0.5), float(ext * 10.0 ** rng.uniform(-2, 1)), tgt, W, H,
                               fovy=float(rng.uniform(20, 110)))
            p = scenes.with_flags(p, lod_percent=int(rng.choice([0, 10, 100])), cull=int(rng.integers(0, 2)))
            tag = f"stream {s} kind {kind} n {n} sort {sort} pad {pad} frame {f} esc {st['escaped_symbols']}"
            ctx.clear(); ctx.render_basic(p)
            ofb, ost = of.render_basic(p)
            if ctx.stats() != ost or not np.array_equal(ctx.read_framebuffer(full=True), ofb):
                bad += 1; print("MISMATCH basic:", tag, flush=True)
            ctx.clear(); ctx.render_hqs_depth(p)
            hfb, _ = of.render_hqs_depth(p)
            ok = np.array_equal(ctx.read_framebuffer(full=True), hfb)
            ctx.render_hqs_color(p)
            org, oba, _ = of.render_hqs_color(p, hfb)
            rg, ba = ctx.read_accum(full=True)
            if not (ok and np.array_equal(rg, org) and np.array_equal(ba, oba)):
                bad += 1; print("MISMATCH hqs:", tag, flush=True)
        if s % 20 == 19:
            print(f"{s + 1} streams, {bad} frame mismatches, {enc_bad} encoder mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"soak: {args.streams} streams x 4 frames x 2 methods, {bad} frame mismatches, {enc_bad} encoder mismatches")
    ctx.close()
    return 1 if (bad or enc_bad) else 0


if __name__ == "__main__":
    sys.exit(main())
