"""BASELINE.json configs[1] at full size on the GPU: 1e8 synthetic points, 1526 batches, 1920x1080, LOD 100 %, culling
off. Direct parity against the oracle (its multi-threaded basic pass takes well under a second at this size) plus the
size-independent properties of the path: shard-and-merge equals the whole, rendering twice changes nothing, the depth
halves of the basic and the HQS depth pass agree, the GPU encoder reproduces the stream."""
import hashlib

import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, scenes

pytestmark = pytest.mark.gpu

N = 100_000_000
W, H = 1920, 1080


@pytest.fixture(scope="module")
def big():
    image, st = P.synth_encode(N, scenes.SEED, nthreads=16)
    assert st["num_batches"] == 1526 and st["num_points"] == 100_007_936
    return image, P.HuffmanFile(image), oracle.OracleFile(image.view())


@pytest.fixture(scope="module")
def whole(big):
    image, hf, of = big
    ctx = P.Context(0)
    ctx.set_image_size(W, H)
    ctx.set_stream_layout(P.Context.LAYOUT_BOTH)        # test_both_decode_variants_agree_at_full_size draws with either
    ctx.stream_begin(hf.header(), 0)
    for b0 in range(0, hf.numBatches, 100):
        ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
    yield ctx
    ctx.close()


def params(cam="overview"):
    return scenes.with_flags(scenes.cameras(W, H)[cam], lod_percent=100, cull=0)


def test_basic_pass_equals_oracle_at_full_size(big, whole):
    _, _, of = big
    p = params()
    whole.clear(); whole.render_basic(p); whole.resolve_basic(p)
    fb = whole.read_framebuffer(full=True)
    ofb, ost = of.render_basic(p, nthreads=16)
    assert whole.stats() == ost and ost["points_iterated"] == 100_007_936
    assert np.array_equal(fb, ofb)
    assert np.array_equal(whole.read_rgba(), oracle.resolve_basic(p, ofb))
    # rendering the same stream again into the same framebuffer changes nothing (min is idempotent)
    whole.render_basic(p)
    assert np.array_equal(whole.read_framebuffer(full=True), fb)


def test_hqs_passes_equal_oracle_at_full_size(big, whole):
    _, _, of = big
    p = params()
    whole.clear(); whole.render_hqs_depth(p)
    dfb = whole.read_framebuffer(full=True)
    ofb, ost = of.render_hqs_depth(p)
    assert whole.stats() == ost and np.array_equal(dfb, ofb)
    whole.render_hqs_color(p); whole.resolve_hqs(p)
    rg, ba = whole.read_accum(full=True)
    org, oba, _ = of.render_hqs_color(p, ofb)
    assert np.array_equal(rg, org) and np.array_equal(ba, oba)
    assert np.array_equal(whole.read_rgba(), oracle.resolve_hqs(p, ofb, org, oba))
    # the depth halves of the two methods are the same min over the same points
    whole.clear(); whole.render_basic(p)
    bfb = whole.read_framebuffer(full=True)
    assert np.array_equal(bfb >> np.uint64(32), dfb >> np.uint64(32))


def test_both_decode_variants_agree_at_full_size(whole):
    """The stream is resident in both layouts (PCR_LAYOUT_BOTH); the packed-words variant (what PCR_VARIANT_AUTO picks for large images)
    and the point-window variant draw bit-identical frames, basic and HQS, for both cameras -- and so do the two workgroup shapes
    (one 1024-thread workgroup per batch, the reference's; two 512-thread workgroups per batch, pcr_set_workgroup_parts)."""
    try:
        for cam in ("overview", "closeup"):
            p = params(cam)
            got = {}
            for parts in (1, 2):
                whole.set_workgroup_parts(parts)
                for name, v in (("point_windows", P.Context.VARIANT_POINT_WINDOWS), ("words", P.Context.VARIANT_WORDS)):
                    whole.set_render_variant(v)
                    whole.clear(); whole.render_basic(p)
                    basic = whole.read_framebuffer(full=True)
                    st_basic = whole.stats()
                    whole.clear(); whole.render_hqs_depth(p); whole.render_hqs_color(p)
                    got[name, parts] = (basic, st_basic, whole.read_framebuffer(full=True), *whole.read_accum(full=True))
            a = got["point_windows", 1]
            for key, b in got.items():
                assert a[1] == b[1], (cam, key)
                for x, y in zip((a[0], a[2], a[3], a[4]), (b[0], b[2], b[3], b[4])):
                    assert np.array_equal(x, y), (cam, key)
    finally:
        whole.set_render_variant(P.Context.VARIANT_AUTO)
        whole.set_workgroup_parts(0)


def test_two_shards_merge_to_the_whole(big, whole):
    """contiguous batch ranges on separate contexts, one min-merge; the second shard's head words ride along."""
    _, hf, _ = big
    p = params("closeup")
    whole.clear(); whole.render_basic(p)
    ref = whole.read_framebuffer(full=True)
    cut = 763
    parts = []
    for first, count in ((0, cut), (cut, hf.numBatches - cut)):
        c = P.Context(0)
        c.set_image_size(W, H)
        c.stream_begin(hf.header(first, count), first)
        for b0 in range(0, count, 100):
            c.upload_batches(b0, [hf.blob(first + b) for b in range(b0, min(b0 + 100, count))])
        if first + count < hf.numBatches:
            c.upload_tail(*hf.head_words(first + count))
        c.clear(); c.render_basic(p)
        parts.append(c)
    parts[1].synchronize()
    parts[0].merge_min(parts[1].device_framebuffer())
    merged = parts[0].read_framebuffer(full=True)
    for c in parts:
        c.close()
    assert np.array_equal(merged, ref)


def test_gpu_encoder_reproduces_the_stream(big, whole):
    image, _, _ = big
    x, y, z, c = P.synth_points(N, scenes.SEED, 0, N)
    gpu, st = whole.gpu_encode_points(x, y, z, c, P.synth_las_info(N, scenes.SEED), morton_sort=True)
    assert hashlib.sha256(gpu.view()).digest() == hashlib.sha256(image.view()).digest()


def test_4096x4096_with_culling_matches_the_oracle(big):
    """BASELINE configs[4], the one-GPU half: 1e8 points at 4096x4096 with the per-batch frustum cull on. The batches'
    screen rectangles outgrow the small LDS windows here, so the library launches k_render with its 140 KiB windows
    (one workgroup per CU); an orbit camera that culls about half of the batches makes the compacted batch list short."""
    _, hf, of = big
    c = P.Context(0)
    try:
        c.set_image_size(4096, 4096)
        c.stream_begin(hf.header(), 0)
        for b0 in range(0, hf.numBatches, 100):
            c.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
        for name, cam, lod in (("overview", P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), 4096, 4096), 100),
                               ("half culled", P.camera_orbit(0.6, -0.5, 420.0, (250.0, 250.0, 40.0), 4096, 4096), 10)):
            p = scenes.with_flags(cam, lod_percent=lod, cull=1)
            c.frame_begin(p); c.render_basic(p); c.resolve_basic(p)
            fb = c.read_framebuffer(full=True)
            ofb, ost = of.render_basic(p, nthreads=16)
            assert c.stats() == ost, name
            assert np.array_equal(fb, ofb), name
            assert np.array_equal(c.read_rgba(), oracle.resolve_basic(p, ofb)), name
            if name == "half culled":
                assert 0.25 * hf.numBatches < ost["batches_culled"] < 0.75 * hf.numBatches, ost
    finally:
        c.close()


def test_packed_words_layout_keeps_the_compression(big):
    """PCR_LAYOUT_WORDS holds per 64 chains only the rows their longest chain consumed -- at most 4.3 B per
    point resident for the benchmark stream (3.73 in the file), against 6.3 for the point windows -- and PCR_LAYOUT_AUTO picks it
    when the windows would not fit a budget. Frames are the oracle's either way."""
    _, hf, of = big
    p = params()
    ofb, ost = of.render_basic(p, nthreads=16)
    c = P.Context(0)
    try:
        c.set_image_size(W, H)
        per_point = {}
        for name, layout, budget in (("words", P.Context.LAYOUT_WORDS, 0), ("auto, 500 MB budget", P.Context.LAYOUT_AUTO, 500 << 20),
                                     ("auto, no budget", P.Context.LAYOUT_AUTO, 0)):
            c.set_hbm_budget(budget)
            c.set_stream_layout(layout)
            c.stream_begin(hf.header(), 0)
            for b0 in range(0, hf.numBatches, 100):
                c.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 100, hf.numBatches))])
            c.frame_begin(p); c.render_basic(p)
            assert c.stats() == ost and np.array_equal(c.read_framebuffer(full=True), ofb), name
            per_point[name] = c.resident_bytes / hf.numPoints
            want = P.Context.LAYOUT_POINT_WINDOWS if name == "auto, no budget" else P.Context.LAYOUT_WORDS
            assert c.stream_layout == want, name
            c.stream_unload()
        assert per_point["words"] <= 4.3 and per_point["auto, 500 MB budget"] == per_point["words"], per_point
        assert 6.0 < per_point["auto, no budget"] < 6.6, per_point
    finally:
        c.close()
