"""Experiment: an UNSORTED stream (preprocess ... sort = 0): batches are 65 536 consecutive input points, not Morton neighbours.
    PCR_HIP_LIB=... python tools/exp/unsorted.py [--points N] [--shuffle]"""
import argparse, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcrhpg24_amd as P
ap = argparse.ArgumentParser(); ap.add_argument("--points", type=int, default=20_000_000); ap.add_argument("--shuffle", action="store_true")
args = ap.parse_args()
n = args.points
x, y, z, c = P.synth_points(n, 0x5EED, 0, n)
if args.shuffle:
    perm = np.random.default_rng(1).permutation(n)
    x, y, z, c = x[perm], y[perm], z[perm], c[perm]
ctx = P.Context(0); ctx.set_image_size(1920, 1080)
image, st = ctx.gpu_encode_points(x, y, z, c, P.synth_las_info(n, 0x5EED), morton_sort=False)
hf = P.HuffmanFile(image)
ctx.stream_begin(hf.header(), 0)
for b0 in range(0, hf.numBatches, 100):
    ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), 1920, 1080); p.lod_percent = 100; p.enable_frustum_culling = 0
ctx.frame_begin(p)
for _ in range(20):
    ctx.render_basic(p); ctx.frame_turn(p, p)
ctx.synchronize()
t0 = time.time()
for _ in range(50):
    ctx.render_basic(p); ctx.frame_turn(p, p)
ctx.synchronize()
print("unsorted%s: %d points, %d batches, %.4f ms per frame, %.2f bits/point" % (" + shuffled" if args.shuffle else "", n, hf.numBatches, (time.time() - t0) / 50 * 1e3, 8.0 * len(image.view()) / n))
