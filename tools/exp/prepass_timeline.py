"""Experiment (library built with -DPCR_EXP_TIMELINE): wall-clock stamps of the prepass block's phases inside the frame turn.
    PCR_HIP_LIB=tools/exp/libpcr_hip_tl.so python tools/exp/prepass_timeline.py"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcrhpg24_amd as P
from pcrhpg24_amd import _native as N
image, _ = P.synth_encode(100_000_000, 0x5EED, nthreads=16)
hf = P.HuffmanFile(image)
nb = hf.numBatches
ctx = P.Context(0); ctx.set_image_size(1920, 1080)
ctx.stream_begin(hf.header(0, nb), 0)
for b0 in range(0, nb, 100):
    ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, nb))])
p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), 1920, 1080); p.lod_percent = 100; p.enable_frustum_culling = 0
ctx.frame_begin(p)
for _ in range(200):
    ctx.render_basic(p); ctx.frame_turn(p, p)
ctx.synchronize()
lib = N.hip_lib()
lib.pcr_exp_read_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
t = np.zeros(8192 * 8, np.uint64)
assert lib.pcr_exp_read_timeline(ctx.h, t.ctypes.data, t.size) == 0
chunks = (nb + 31) // 32
t = t.reshape(8192, 8)[7000:7000 + 2 * chunks].astype(np.int64)     # rows 2 c: the chunk's lists workgroup, 2 c + 1: its plans workgroup
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print("prepass chunks", chunks)
for role, rows, cols, names in (("lists", us[0::2], (0, 1, 2, 3, 4), ["start", "loads requested", "lod computed", "stats committed", "lists written"]),
                                ("plans", us[1::2], (0, 1, 2, 5, 6, 7), ["start", "loads requested", "lod computed", "barrier", "plans written", "barrier / end"])):
    print(role, "workgroups:")
    for k, name in zip(cols, names):
        print("  %-18s mean %6.2f  min %6.2f  max %6.2f us (since the first workgroup's start)" % (name, rows[:, k].mean(), rows[:, k].min(), rows[:, k].max()))
