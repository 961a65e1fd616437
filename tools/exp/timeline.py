"""Experiment (library built with -DPCR_EXP_TIMELINE): when does every workgroup of one k_render launch pass its phases, and where?
    PCR_HIP_LIB=tools/exp/libpcr_hip_tl.so python tools/exp/timeline.py [--batches N]"""
import argparse, ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcrhpg24_amd as P
from pcrhpg24_amd import _native as N
ap = argparse.ArgumentParser(); ap.add_argument("--batches", type=int, default=0); ap.add_argument("--out", default=""); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080); ap.add_argument("--cull", type=int, default=0); ap.add_argument("--lod", type=int, default=100); ap.add_argument("--layout", type=int, default=-1, help="pcr_set_stream_layout value (4: the point-keys patch)")
args = ap.parse_args()
image, _ = P.synth_encode(100_000_000, 0x5EED, nthreads=16)
hf = P.HuffmanFile(image)
nb = args.batches or hf.numBatches
ctx = P.Context(0); ctx.set_image_size(args.width, args.height)
if args.layout >= 0:
    ctx.set_stream_layout(args.layout)
ctx.stream_begin(hf.header(0, nb), 0)
for b0 in range(0, nb, 100):
    ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, nb))])
if nb < hf.numBatches:
    ctx.upload_tail(*hf.head_words(nb))
p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), args.width, args.height); p.lod_percent = args.lod; p.enable_frustum_culling = args.cull
ctx.frame_begin(p)
for _ in range(300):
    ctx.render_basic(p); ctx.frame_turn(p, p)
ctx.synchronize()
parts = int(os.environ.get("PCR_PARTS", "2"))
batches = nb
if parts == 2:
    nb = ((nb + 7) // 8) * 16                 # workgroups of the grid (half-batches; the padding ones never stamp)
waves = 16 // parts
lib = N.hip_lib()
lib.pcr_exp_read_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
t = np.zeros(nb * 8, np.uint64)
assert lib.pcr_exp_read_timeline(ctx.h, t.ctypes.data, t.size) == 0
t = t.reshape(nb, 8)
live = t[:, 5] > 0
t = t[live]
nb = int(live.sum())
t0 = t[:, 0].min()
us = (t[:, :6].astype(np.int64) - int(t0)) / 100.0         # 100 MHz -> microseconds
hw = t[:, 7]
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
xcc = (hw >> np.uint64(32)).astype(int) & 15
print("workgroups", nb, "kernel span %.1f us" % us[:, 5].max())
names = ["start", "loads issued", "barrier passed", "loop done", "merge barrier", "end"]
for k in range(1, 6):
    d = us[:, k] - us[:, k - 1]
    print("%-16s -> %-16s  mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (names[k - 1], names[k], d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
life = us[:, 5] - us[:, 0]
print("lifetime mean %.2f p10 %.2f p90 %.2f max %.2f us" % (life.mean(), *np.percentile(life, [10, 90]), life.max()))
order = np.argsort(us[:, 0])
print("start times (us) of workgroups in start order: ", np.round(us[order, 0][::max(1, nb // 24)], 1))
print("end times quantiles:", np.round(np.percentile(us[:, 5], [1, 10, 25, 50, 75, 90, 99, 100]), 1))
slot = xcc * 1000 + se * 100 + sh * 16 + cu
print("distinct (xcc,se,sh,cu):", len(set(slot.tolist())), " workgroups per CU: min %d max %d" % (np.bincount(np.unique(slot, return_inverse=True)[1]).min(), np.bincount(np.unique(slot, return_inverse=True)[1]).max()))
# busy CUs over time
ts = np.linspace(0, us[:, 5].max(), 41)
print("time_us : resident workgroups / in loop")
for x in ts:
    res = ((us[:, 0] <= x) & (us[:, 5] > x)).sum(); loop = ((us[:, 2] <= x) & (us[:, 3] > x)).sum()
    print("%7.1f : %4d %4d" % (x, res, loop))
lib.pcr_exp_read_wave_ends.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
we = np.zeros(live.size * 16, np.uint64)
assert lib.pcr_exp_read_wave_ends(ctx.h, we.ctypes.data, we.size) == 0
we = (we.reshape(-1, 16)[live][:, :waves].astype(np.int64) - int(t0)) / 100.0
rel = we - us[:, 2:3]                       # loop duration per wave
print("per-wave loop duration: mean %.1f  p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (rel.mean(), *np.percentile(rel, [10, 50, 90]), rel.max()))
print("mean loop duration by wave index   :", np.round(rel.mean(axis=0), 1))
print("mean rank of finish by wave index  :", np.round(np.argsort(np.argsort(rel, axis=1), axis=1).mean(axis=0), 1))
spread = rel.max(axis=1) - rel.min(axis=1)
print("spread within a workgroup (last - first wave): mean %.1f p50 %.1f p90 %.1f us" % (spread.mean(), *np.percentile(spread, [50, 90])))
srt = np.sort(rel, axis=1)
print("mean sorted per-wave durations within a workgroup:", np.round(srt.mean(axis=0), 1))
# steady state only (workgroups that started after 60 us)
ss = us[:, 0] > 60
print("steady state (start > 60 us): n=%d, sorted durations:" % ss.sum(), np.round(srt[ss].mean(axis=0), 1))
if args.out:
    np.save(args.out, t)
# the slowest workgroups: which batches are they, when did they start? (LOD 100 %, no culling: list entry == batch)
idx = np.nonzero(live)[0]
loopd = us[:, 3] - us[:, 2]
print("slowest workgroups (loop us, lifetime us, start us, end us, batch, part):")
for j in np.argsort(-loopd)[:24]:
    x = int(idx[j])
    if parts == 2:
        part, entry = (x >> 3) & 1, (x >> 4) * 8 + (x & 7)
    else:
        part, entry = 0, x
    print("  %7.1f %7.1f %7.1f %7.1f  %5d %d" % (loopd[j], life[j], us[j, 0], us[j, 5], entry, part))
late = np.argsort(-us[:, 5])[:24]
print("last workgroups to end (end us, start us, loop us, batch, part):")
for j in late:
    x = int(idx[j])
    part, entry = ((x >> 3) & 1, (x >> 4) * 8 + (x & 7)) if parts == 2 else (0, x)
    print("  %7.1f %7.1f %7.1f  %5d %d" % (us[j, 5], us[j, 0], loopd[j], entry, part))
