"""Experiment (library built with -DPCR_EXP_FAR_STATS): the points outside their LDS windows in one frame -- how many wave-iterations have
such lanes, how many lanes, how many of those iterations pre-read the global framebuffer word (batches the prepass voted
"mostly outside"). (The filter's pass rates quoted in profiles/r03_experiments.md section 13 came from an earlier form of this hook that
sampled every eighth such point.)   PCR_HIP_LIB=tools/exp/libpcr_hip_far.so python tools/exp/far_stats.py"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcrhpg24_amd as P
from pcrhpg24_amd import _native as N
lib = N.hip_lib()
lib.pcr_exp_read_far.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
def frame(ctx, p, label):
    out = np.zeros(8, np.uint64)
    ctx.frame_begin(p); ctx.render_basic(p); lib.pcr_exp_read_far(ctx.h, out.ctypes.data, 1)
    ctx.frame_begin(p); ctx.render_basic(p); lib.pcr_exp_read_far(ctx.h, out.ctypes.data, 1)
    it, lanes, pre, _, _, waves = (int(v) for v in out[:6])
    print("%-28s wave-iterations with such lanes %9d of %9d (%.2f %%), lanes per such iteration %.1f, of those iterations pre-read (mostly_outside batches) %d"
          % (label, it, waves * 64, 100.0 * it / max(1, waves * 64), lanes / max(1, it), pre))
image, _ = P.synth_encode(100_000_000, 0x5EED, nthreads=16)
hf = P.HuffmanFile(image)
for (w, h, cam, cull, label) in ((1920, 1080, "overview", 0, "1080p overview"), (4096, 4096, "overview", 1, "4096x4096 overview"), (1920, 1080, "closeup", 0, "1080p close-up")):
    ctx = P.Context(0); ctx.set_image_size(w, h)
    ctx.stream_begin(hf.header(), 0)
    for b0 in range(0, hf.numBatches, 100):
        ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
    from tests import scenes
    p = scenes.with_flags(scenes.cameras(w, h)[cam], lod_percent=100, cull=cull)
    frame(ctx, p, label); ctx.close()
n = 20_000_000
x, y, z, c = P.synth_points(n, 0x5EED, 0, n)
ctx = P.Context(0); ctx.set_image_size(1920, 1080)
image, st = ctx.gpu_encode_points(x, y, z, c, P.synth_las_info(n, 0x5EED), morton_sort=False)
hf = P.HuffmanFile(image)
ctx.stream_begin(hf.header(), 0)
for b0 in range(0, hf.numBatches, 100):
    ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), 1920, 1080); p.lod_percent = 100; p.enable_frustum_culling = 0
frame(ctx, p, "unsorted 2e7 points")
