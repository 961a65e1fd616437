"""GPU parity on the corners of the path: wide table values, escape-pool overflow, tiny and huge images, LOD 0,
window rectangles clipped by the screen, reload / resize cycles, two contexts on one device."""
import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, scenes

pytestmark = pytest.mark.gpu


def check_all(ctx, of, p):
    ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)
    ofb, ost = of.render_basic(p)
    assert ctx.stats() == ost
    fb = ctx.read_framebuffer(full=True)
    bad = np.nonzero(fb != ofb)[0]
    assert bad.size == 0, f"basic: {bad.size} words differ, first {bad[:4]}"
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb))
    ctx.clear(); ctx.render_hqs_depth(p)
    hfb, hst = of.render_hqs_depth(p)
    assert ctx.stats() == hst and np.array_equal(ctx.read_framebuffer(full=True), hfb)
    ctx.render_hqs_color(p); ctx.resolve_hqs(p)
    org, oba, _ = of.render_hqs_color(p, hfb)
    rg, ba = ctx.read_accum(full=True)
    assert np.array_equal(rg, org) and np.array_equal(ba, oba)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_hqs(p, hfb, org, oba))
    return ost


def load(ctx, image):
    f = P.HuffmanFile(image)
    if ctx.batches_loaded:
        ctx.stream_unload()
    ctx.stream_begin(f.header())
    for b in range(f.numBatches):
        ctx.upload_batch(b, f.blob(b))
    return f


@pytest.fixture(params=["point_windows", "words", "point_windows_whole"])
def ctx(request):
    c = P.Context(0)
    if request.param == "point_windows_whole":          # one 1024-thread workgroup per batch (the others: the library's choice, half-batches)
        c.set_workgroup_parts(1)
    if request.param == "words":
        c.set_stream_layout(P.Context.LAYOUT_BOTH)      # both layouts resident, the packed-words kernel forced
    c.set_render_variant(P.Context.VARIANT_WORDS if request.param == "words" else P.Context.VARIANT_POINT_WINDOWS)
    yield c
    c.close()


def las_for(lo, hi, scale=0.001):
    las = P.LasInfo()
    for k in range(3):
        las.scale[k] = scale; las.offset[k] = 0.0; las.min[k] = lo[k] * scale; las.max[k] = hi[k] * scale
    return las


@pytest.mark.parametrize("hop_bits", [30, 20])
def test_wide_table_values(ctx, hop_bits):
    """Frequent symbols that do not fit the packed LDS entry (+-2^30 deltas with short codes), and deltas straddling
    the entry's value range (+-2^20 +- 2)."""
    rng = np.random.default_rng(21)
    n = 65536 * 2
    hop = np.where(np.arange(n) % 2 == 0, 0, 1 << hop_bits).astype(np.int64)
    x = (hop + rng.integers(0, 3, n)).astype(np.int32)
    y = rng.integers(0, 2000, n).astype(np.int32)
    z = rng.integers(0, 50, n).astype(np.int32)
    c = rng.integers(0, 1 << 24, n).astype(np.uint32)
    image, st = P.encode_points(x, y, z, c, las_for((0, 0, 0), (1 << 30, 2000, 50)), morton_sort=False, nthreads=2)
    of = oracle.OracleFile(image.view())
    tv = np.ctypeslib.as_array(__import__("ctypes").cast(of.s.dt_values, __import__("ctypes").POINTER(__import__("ctypes").c_int32)), (4096,))
    tl = np.ctypeslib.as_array(__import__("ctypes").cast(of.s.dt_cwlen, __import__("ctypes").POINTER(__import__("ctypes").c_int32)), (4096,))
    in_table = tv[tl > 0].astype(np.int64)
    if hop_bits == 30:
        assert (np.abs(in_table) >= 1 << 25).any(), "the stream must contain in-table values far outside the packed entry"
    else:
        for v in ((1 << 20) - 1, 1 << 20, -(1 << 20), -(1 << 20) - 1):
            assert (in_table == v).any(), f"the stream must contain the in-table value {v}"
    ctx.set_image_size(320, 200)
    load(ctx, image)
    p = scenes.with_flags(P.camera_orbit(0.2, -0.8, 3.0e6, (5.0e5, 1.0, 0.0), 320, 200), lod_percent=100, cull=0)
    st = check_all(ctx, of, p)
    assert st["points_iterated"] == n


def test_escape_pool_overflow_and_lod_zero(ctx):
    """Escape-heavy batches read their escapes from global memory; lod_percent 0 lets far batches render 0 points."""
    x, y, z, c, las = scenes.random_points(131072, seed=3)
    image, st = P.encode_points(x, y, z, c, las, morton_sort=True, nthreads=2)
    assert st["escaped_symbols"] > 6144 * st["num_batches"]
    of = oracle.OracleFile(image.view())
    ctx.set_image_size(256, 144)
    load(ctx, image)
    for rad, lod in ((2000.0, 100), (400000.0, 0), (9000.0, 0)):
        p = scenes.with_flags(P.camera_orbit(0.5, -0.6, rad, (100.0, 100.0, 100.0), 256, 144), lod_percent=lod)
        check_all(ctx, of, p)


def test_batches_that_fall_apart_into_clusters_get_a_window_per_run(ctx):
    """Batches whose points are clusters far apart (a jump of the Morton curve inside the batch, even inside one chain): their
    bounding rectangle covers most of the screen and fits no LDS window, so the prepass gives every run of chains its own
    (k_bounds + plan_windows); chains that straddle a jump scatter through global memory. Any plan has to give the oracle's frame."""
    rng = np.random.default_rng(77)
    n = 300_000
    centres = np.array([[50_000, 60_000, 2_000], [900_000, 80_000, 9_000], [120_000, 950_000, 4_000], [880_000, 900_000, 1_000], [500_000, 500_000, 30_000]])
    which = rng.integers(0, len(centres), n)
    xyz = centres[which] + rng.normal(0, [6_000, 6_000, 800], (n, 3))
    x, y, z = (np.clip(xyz[:, k], 0, 1_000_000).astype(np.int32) for k in range(3))
    c = rng.integers(0, 1 << 24, n, dtype=np.int64).astype(np.uint32)
    image, st = P.encode_points(x, y, z, c, las_for((0, 0, 0), (1_000_000, 1_000_000, 40_000)), morton_sort=True, nthreads=2)
    of = oracle.OracleFile(image.view())
    for w, h in ((1920, 1080), (640, 360)):
        ctx.set_image_size(w, h)
        load(ctx, image)
        for p in (P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 20.0), w, h), P.camera_orbit(0.8, -1.1, 1100.0, (500.0, 500.0, 20.0), w, h)):
            for lod in (100, 10):
                ost = check_all(ctx, of, scenes.with_flags(p, lod_percent=lod, cull=1))
                assert ost["points_iterated"] > 0


def test_unsorted_stream_whose_batches_are_strips_across_the_scene(ctx):
    """`preprocess ... sort = 0`: a batch is 65 536 consecutive input points -- here rows of a height field, a strip across the whole
    scene that no LDS window holds. Every batch of the prepass workgroup lies mostly outside its windows, so the vote sets
    WinPlan::mostly_outside and k_render takes the one path that pre-reads the global framebuffer word of a point outside its
    window (elsewhere such points go to the atomic unfiltered: the frame has to be the oracle's either way)."""
    n = 700_000
    x, y, z, c = P.synth_points(n, scenes.SEED, 0, n)
    image, st = P.encode_points(x, y, z, c, P.synth_las_info(n, scenes.SEED), morton_sort=False, nthreads=2)
    assert st["num_batches"] >= 10
    of = oracle.OracleFile(image.view())
    for w, h in ((1920, 1080), (800, 450)):
        ctx.set_image_size(w, h)
        load(ctx, image)
        for cam in ("overview", "closeup"):
            for lod, cull in ((100, 0), (40, 1)):
                check_all(ctx, of, scenes.with_flags(scenes.cameras(w, h)[cam], lod_percent=lod, cull=cull))


@pytest.mark.parametrize("size", [(64, 36), (33, 97), (4096, 4096)])
def test_image_sizes_and_clipped_windows(ctx, size):
    w, h = size
    image, _ = scenes.synth_stream(600_000)
    of = oracle.OracleFile(image.view())
    ctx.set_image_size(w, h)
    load(ctx, image)
    cams = [P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), w, h),
            P.camera_orbit(-1.68, -0.39, 70.0, (300.0, 20.0, 45.0), w, h),        # batches straddle the screen border
            P.camera_orbit(0.3, -1.2, 900.0, (0.0, 0.0, 40.0), w, h)]              # tile corner at the screen centre
    for p in cams:
        for lod, cull in ((100, 1), (10, 1)):
            check_all(ctx, of, scenes.with_flags(p, lod_percent=lod, cull=cull))


def test_reload_resize_and_two_contexts(ctx):
    a, _ = scenes.synth_stream(200_000)
    b, _ = scenes.synth_stream(600_000)
    oa, ob = oracle.OracleFile(a.view()), oracle.OracleFile(b.view())
    other = P.Context(0)
    try:
        for image, of, (w, h) in ((a, oa, (200, 120)), (b, ob, (400, 240)), (a, oa, (128, 128))):
            ctx.set_image_size(w, h)
            other.set_image_size(w, h)
            load(ctx, image)
            load(other, image)
            p = scenes.with_flags(P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), w, h), lod_percent=50)
            check_all(ctx, of, p)
            # interleaved use of a second context on the same device
            other.clear(); other.render_basic(p)
            ctx.clear(); ctx.render_basic(p)
            assert np.array_equal(other.read_framebuffer(full=True), ctx.read_framebuffer(full=True))
    finally:
        other.close()


def test_sharded_contexts_merge_to_the_single_context_result(ctx):
    """Two contexts each hold one contiguous batch range (+ the follower's head words); pcr_merge_min / pcr_merge_sum
    of their framebuffers equal the unsharded render — the multi-GPU exchange step, on one device."""
    image, _ = scenes.synth_stream(600_000)
    f = P.HuffmanFile(image)
    of = oracle.OracleFile(image.view())
    w, h = 320, 180
    shard = P.Context(0)
    try:
        ctxs = [ctx, shard]
        half = f.numBatches // 2
        ranges = [(0, half), (half, f.numBatches - half)]
        for c, (first, count) in zip(ctxs, ranges):
            c.set_image_size(w, h)
            c.stream_begin(f.header(first, count), first)
            for i in range(count):
                c.upload_batch(i, f.blob(first + i))
            if first + count < f.numBatches:
                c.upload_tail(*f.head_words(first + count))
        p = scenes.with_flags(P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), w, h), lod_percent=100)
        for c in ctxs:
            c.clear(); c.render_basic(p)
        shard.synchronize()
        ctx.merge_min(shard.device_framebuffer())
        ofb, _ = of.render_basic(p)
        assert np.array_equal(ctx.read_framebuffer(full=True), ofb)
        # HQS: min-merge depth both ways, colour pass against the global depth, sum-merge
        for c in ctxs:
            c.clear(); c.render_hqs_depth(p)
        shard.synchronize(); ctx.synchronize()
        ctx.merge_min(shard.device_framebuffer()); ctx.synchronize()
        shard.merge_min(ctx.device_framebuffer())
        for c in ctxs:
            c.render_hqs_color(p)
        shard.synchronize()
        lib = ctx.lib
        ctx.merge_sum(int(lib.pcr_device_rg(shard.h)), int(lib.pcr_device_ba(shard.h)))
        hfb, _ = of.render_hqs_depth(p)
        org, oba, _ = of.render_hqs_color(p, hfb)
        rg, ba = ctx.read_accum(full=True)
        assert np.array_equal(ctx.read_framebuffer(full=True), hfb)
        assert np.array_equal(rg, org) and np.array_equal(ba, oba)
        ctx.flip_sign(); ctx.flip_sign()
        assert np.array_equal(ctx.read_framebuffer(full=True), hfb)
    finally:
        shard.close()


def test_pad_tails_stream(ctx):
    """A stream written with PCR_ENCODE_PAD_TAILS (zero words queued for the refills past each chain's end) goes through
    the same kernels; all 64 points of every chain are drawn from exact coordinates (checked on the CPU side in
    tests/test_oracle_format.py), and the framebuffer equals the oracle's."""
    x, y, z, c = P.synth_points(150_000, scenes.SEED, 0, 150_000)
    image, st = P.encode_points(x, y, z, c, P.synth_las_info(150_000), morton_sort=True, nthreads=2, pad_tails=True)
    of = oracle.OracleFile(image.view())
    ctx.set_image_size(640, 360)
    load(ctx, image)
    for cam in ("overview", "closeup"):
        p = scenes.with_flags(scenes.cameras(640, 360)[cam], lod_percent=100, cull=0)
        stt = check_all(ctx, of, p)
        assert stt["points_iterated"] == of.num_batches * 65536


def test_async_loader_with_shard_tail(ctx):
    """A shard loaded through the loader stream (pcr_set_async_upload) and closed with pcr_upload_tail draws the same
    frame as the oracle's render of that batch range of the whole file (the last batch is re-walked with the follower's words)."""
    image, _ = scenes.synth_stream(600_000)
    f = P.HuffmanFile(image)
    of = oracle.OracleFile(image.view())
    w, h = 320, 180
    first, count = 0, f.numBatches // 2
    ctx.set_image_size(w, h)
    ctx.stream_begin(f.header(first, count), first)
    ctx.set_async_upload(True)
    try:
        for b0 in range(0, count, 2):
            ctx.upload_batches(b0, [f.blob(first + b) for b in range(b0, min(b0 + 2, count))])
        ctx.upload_tail(*f.head_words(first + count))
        assert ctx.batches_resident == count
        p = scenes.with_flags(P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), w, h), lod_percent=100)
        ctx.clear(); ctx.render_basic(p)
        ofb, ost = of.render_basic(p, first=first, count=count)
        assert ctx.stats() == ost
        assert np.array_equal(ctx.read_framebuffer(full=True), ofb)
    finally:
        ctx.set_async_upload(False)


def test_first_frame_releases_what_only_the_transcode_reads():
    """ADVICE r01 / VERDICT item 4: a context keeps only what its layout's kernel reads. The raw word stream, the int32/int8
    tables and the cluster prefix go with the first frame after the last upload; the default layout holds no lane-major
    words at all; a shard tail has to arrive before that frame."""
    nb, st = P.synth_encode(100_000_000, scenes.SEED, 0, 655_360, 6_553_600, 4)     # ten batches of the benchmark stream
    f = P.HuffmanFile(nb.view())
    of = oracle.OracleFile(nb.view())
    p = scenes.with_flags(scenes.cameras(640, 360)["overview"], lod_percent=100, cull=0)
    sizes = {}
    for name, layout in (("point_windows", P.Context.LAYOUT_POINT_WINDOWS), ("words", P.Context.LAYOUT_WORDS), ("both", P.Context.LAYOUT_BOTH)):
        c = P.Context(0)
        try:
            c.set_image_size(640, 360)
            c.set_stream_layout(layout)
            load(c, nb.view())
            before = c.resident_bytes
            c.clear(); c.render_basic(p)
            fb = c.read_framebuffer(full=True)
            after = c.resident_bytes
            assert np.array_equal(fb, of.render_basic(p)[0])
            assert after < before                       # raw words + tables released
            c.clear(); c.render_basic(p)                # and nothing that was released is needed again
            assert np.array_equal(c.read_framebuffer(full=True), fb)
            with pytest.raises(P.PcrError, match="finalised"):
                c.upload_tail(np.ones(4, np.uint32), np.zeros(0, np.int32))
            c.upload_tail(np.zeros(0, np.uint32), np.zeros(0, np.int32))      # an empty tail is a no-op
            sizes[name] = after / (f.numBatches * 65536)
        finally:
            c.close()
    # bytes per point resident: windows 5 + side data; packed words ~2.9 (per 64 chains the rows their longest chain consumed) + side data
    assert sizes["point_windows"] < 6.6 and sizes["words"] < 4.4 and sizes["point_windows"] + 2.5 < sizes["both"] < sizes["point_windows"] + 3.3


def test_degenerate_matrices_w_zero_and_w_out_of_the_fast_range(ctx):
    """rasterize() rejects `pos.w <= 0` (render.cu:296). The kernels' inside test is |x| <= w and |y| <= w -- true for x == y == w == 0,
    a point exactly in the eye, or every point under an all-zero matrix -- so the slow division path (w outside [2^-64, 2^64)) has to
    take such lanes out again. Also: matrices scaled by 2^-80 and 2^80 (every w outside the fast division's range: plain `/` for all
    lanes) draw the frame of the unscaled matrix's pixels with the scaled depths."""
    image, _ = scenes.synth_stream(200_000)
    of = oracle.OracleFile(image.view())
    ctx.set_image_size(320, 200)
    load(ctx, image)
    base = scenes.with_flags(scenes.cameras(320, 200)["overview"], lod_percent=100, cull=0)
    zero = base.copy()
    for k in range(16):
        zero.transform[k] = 0.0
    check_all(ctx, of, zero)
    ctx.clear(); ctx.render_basic(zero)
    assert (ctx.read_framebuffer(full=True) == np.uint64(0xFFFFFFFFFFFFFFFF)).all()       # w == 0 everywhere: nothing is inside
    for scale in (2.0 ** -80, 2.0 ** 80):
        p = base.copy()
        for k in range(16):
            p.transform[k] = base.transform[k] * scale
        check_all(ctx, of, p)
