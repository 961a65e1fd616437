"""Dirty tiles (pcr_frame_turn resolves and clears only the 64 x 16-pixel tiles something was written in): whatever the
sequence of frames and calls, the image after a turn is the oracle's resolve of the finished frame, the framebuffer (and the
HQS sums) are empty afterwards, and the next frame is drawn as if from a full clear. Sequences: cameras that cover different
parts of the image one after the other (tiles the image still holds something in get their background back), frames with the
garbage tails of chains landing anywhere (marked point by point), HQS, changes of method, and every call that
writes the framebuffer behind the flags' back (separate resolves and clears, merges, external buffers, the 10-10-10 method)."""
import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, scenes

pytestmark = pytest.mark.gpu

W, H = 1000, 600          # (not a multiple of the tile size in either direction)
EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module", params=["half_batches", "whole_batches"])
def loaded(request):
    nb, _ = scenes.synth_stream(3_000_000)
    of = oracle.OracleFile(nb.view())
    ctx = P.Context(0)
    ctx.set_workgroup_parts(1 if request.param == "whole_batches" else 0)      # (0: the library's choice, two 512-thread workgroups per batch)
    ctx.set_image_size(W, H)
    hf = P.HuffmanFile(nb)
    ctx.stream_begin(hf.header())
    ctx.upload_batches(0, [hf.blob(b) for b in range(hf.numBatches)])
    yield ctx, of
    ctx.close()


def cams():
    c = scenes.cameras(W, H)
    left = P.camera_orbit(-0.15, -0.57, 600.0, (150.0, 500.0, 40.0), W, H)        # the scene's left part only
    right = P.camera_orbit(-0.15, -0.57, 600.0, (850.0, 500.0, 40.0), W, H)
    return [scenes.with_flags(c["overview"], lod_percent=100, cull=0), scenes.with_flags(left, lod_percent=100, cull=1),
            scenes.with_flags(right, lod_percent=100, cull=1), scenes.with_flags(c["closeup"], lod_percent=10, cull=1),
            scenes.with_flags(c["far"], lod_percent=100, cull=0), scenes.with_flags(c["inside"], lod_percent=100, cull=1)]


def test_steady_loop_of_turns_basic(loaded):
    ctx, of = loaded
    seq = cams() + cams()[::-1]
    ctx.frame_begin(seq[0])
    for k, p in enumerate(seq):
        nxt = seq[(k + 1) % len(seq)]
        ctx.render_basic(p)
        ofb, ost = of.render_basic(p)
        assert ctx.stats() == ost, k
        assert np.array_equal(ctx.read_framebuffer(full=True), ofb), k
        ctx.frame_turn(p, nxt)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb)), k
        assert np.all(ctx.read_framebuffer(full=True) == EMPTY), k      # (reading does not disturb the flags)


def test_steady_loop_of_turns_hqs_and_method_changes(loaded):
    ctx, of = loaded
    seq = cams()
    ctx.frame_begin(seq[0], hqs=True)
    for k, p in enumerate(seq):
        nxt = seq[(k + 1) % len(seq)]
        ctx.render_hqs_depth(p); ctx.render_hqs_color(p)
        hfb, _ = of.render_hqs_depth(p)
        org, oba, _ = of.render_hqs_color(p, hfb)
        ctx.frame_turn(p, nxt, hqs=True)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_hqs(p, hfb, org, oba)), k
        rg, ba = ctx.read_accum(full=True)
        assert np.all(ctx.read_framebuffer(full=True) == EMPTY) and not rg.any() and not ba.any(), k
    # a basic frame behind the HQS ones, and an HQS frame behind that
    p, q = seq[1], seq[2]
    ctx.frame_begin(p)
    ctx.render_basic(p); ctx.frame_turn(p, q, hqs=False)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, of.render_basic(p)[0]))
    ctx.frame_begin(q, hqs=True)
    ctx.render_hqs_depth(q); ctx.render_hqs_color(q); ctx.frame_turn(q, p, hqs=True)
    hfb, _ = of.render_hqs_depth(q)
    org, oba, _ = of.render_hqs_color(q, hfb)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_hqs(q, hfb, org, oba))


def test_calls_that_write_behind_the_flags(loaded):
    """Separate resolves and clears, a merge from another buffer, external buffers and back: after each, turns are exact again."""
    import torch
    ctx, of = loaded
    a, b, c = cams()[1], cams()[2], cams()[0]
    want = lambda p: oracle.resolve_basic(p, of.render_basic(p)[0])
    # separate resolve + clear between turns
    ctx.frame_begin(a); ctx.render_basic(a); ctx.frame_turn(a, b)
    ctx.render_basic(b); ctx.resolve_basic(b)
    assert np.array_equal(ctx.read_rgba(), want(b))
    ctx.clear(); ctx.render_basic(a); ctx.frame_turn(a, c)             # b's pixels must not survive in the image
    assert np.array_equal(ctx.read_rgba(), want(a))
    ctx.render_basic(c); ctx.frame_turn(c, a)
    assert np.array_equal(ctx.read_rgba(), want(c))
    # a merge writes anywhere: the frame that follows is resolved and cleared whole
    other = torch.full((W * (H + 1) + 1,), -1, dtype=torch.int64, device="cuda")      # all ones = empty
    fa, _ = of.render_basic(a)
    fb_b, _ = of.render_basic(b)
    other.copy_(torch.from_numpy(fb_b.view(np.int64)))
    torch.cuda.synchronize()
    ctx.render_basic(a); ctx.merge_min(other.data_ptr()); ctx.frame_turn(a, b)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(a, np.minimum(fa, fb_b)))
    assert np.all(ctx.read_framebuffer(full=True) == EMPTY)
    ctx.render_basic(b); ctx.frame_turn(b, a)
    assert np.array_equal(ctx.read_rgba(), want(b))
    # external buffers and back
    ext = torch.full((W * (H + 1) + 1,), -1, dtype=torch.int64, device="cuda")
    ctx.use_external_buffers(ext.data_ptr(), 0, 0)
    ctx.frame_begin(a); ctx.render_basic(a); ctx.frame_turn(a, c)
    assert np.array_equal(ctx.read_rgba(), want(a))
    ctx.use_external_buffers(0, 0, 0)
    ctx.frame_begin(c); ctx.render_basic(c); ctx.frame_turn(c, b)
    assert np.array_equal(ctx.read_rgba(), want(c))
    ctx.render_basic(b); ctx.frame_turn(b, a)
    assert np.array_equal(ctx.read_rgba(), want(b))
    assert np.all(ctx.read_framebuffer(full=True) == EMPTY)


def test_garbage_tails_land_anywhere_and_are_cleared():
    """The low-entropy fixture packed by the reference's library: hundreds of points from chain tails far outside their batch's
    box. A wide camera shows where they land; two turns later nothing of them is left."""
    import json
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    data = open(os.path.join(gold, "ref_packed_lowentropy.huffman"), "rb").read()
    exp = json.load(open(os.path.join(gold, "ref_packed_lowentropy_expected.json")))
    of = oracle.OracleFile(data)
    w, h = exp["width"], exp["height"]
    wide = scenes.with_flags(P.camera_orbit(0.7, -0.5, 90.0, (5.0, 5.0, 3.0), w, h), lod_percent=100, cull=0)
    near = scenes.with_flags(P.camera_orbit(-0.4, -0.6, 18.0, (5.0, 5.0, 3.0), w, h), lod_percent=100, cull=0)
    r = P.Renderer(w, h, device=0)
    try:
        P.HuffmanLasData.create(data).load_all(r)
        ctx = r.ctx
        ctx.frame_begin(wide)
        for p, nxt in ((wide, near), (near, wide), (wide, wide), (wide, near)):
            ctx.render_basic(p)
            ofb, _ = of.render_basic(p)
            assert np.array_equal(ctx.read_framebuffer(full=True), ofb)
            ctx.frame_turn(p, nxt)
            assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb))
            assert np.all(ctx.read_framebuffer(full=True) == EMPTY)
    finally:
        r.ctx.close()


@pytest.mark.parametrize("parts", [0, 1])
def test_garbage_tails_under_per_run_windows(parts):
    """when a batch's own rectangle outgrows the LDS, its windows are the rectangles of its runs of chains, whose boxes
    include the chains' garbage tails and may stick out of the rectangle the prepass marks the dirty tiles under. A
    point inside such a window is never tested against that rectangle, so the windows are cut to it (plan_windows). Large images
    and near cameras over the two reference-packed fixtures with tails: every turn's image and the emptied framebuffer are exact."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    for name in ("ref_packed_lowentropy.huffman", "ref_packed_batch.huffman"):
        data = open(os.path.join(gold, name), "rb").read()
        of = oracle.OracleFile(data)
        w, h = 3000, 2000
        r = P.Renderer(w, h, device=0)
        try:
            r.ctx.set_workgroup_parts(parts)
            P.HuffmanLasData.create(data).load_all(r)
            ctx = r.ctx
            seq = [scenes.with_flags(P.camera_orbit(yaw, pitch, radius, (5.0, 5.0, 3.0), w, h), lod_percent=100, cull=0)
                   for yaw, pitch, radius in ((0.7, -0.5, 30.0), (-0.4, -0.6, 12.0), (2.0, -0.3, 60.0), (0.1, -1.2, 20.0), (0.7, -0.5, 30.0))]
            ctx.frame_begin(seq[0])
            for k, p in enumerate(seq):
                nxt = seq[(k + 1) % len(seq)]
                ctx.render_basic(p)
                ofb, _ = of.render_basic(p)
                assert np.array_equal(ctx.read_framebuffer(full=True), ofb), (name, k)
                ctx.frame_turn(p, nxt)
                assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb)), (name, k)
                assert np.all(ctx.read_framebuffer(full=True) == EMPTY), (name, k)
        finally:
            r.ctx.close()


def test_a_pointer_fetched_once_and_written_every_frame(loaded):
    """an integrator fetches pcr_device_framebuffer ONCE and merges into the context's own framebuffer every frame
    (their own collective, say) -- outside the tiles the library marked. From the getter on every turn walks the whole frame; after
    pcr_framebuffer_private the tiles are used again. Also: pcr_set_int64_mergeable between turns changes the empty word everywhere."""
    import torch
    ctx, of = loaded
    a, b = cams()[1], cams()[2]              # left part / right part of the scene: disjoint tiles
    fa, fb_b = of.render_basic(a)[0], of.render_basic(b)[0]
    n = W * (H + 1) + 1
    ptr = ctx.device_framebuffer()           # fetched once, before the loop
    want = np.minimum(fa, fb_b)
    merged = torch.from_numpy(want.view(np.int64).copy()).cuda()
    ctx.frame_begin(a)
    for k in range(3):
        ctx.render_basic(a)
        ctx.synchronize()
        # the integrator's own merge through that pointer: b's pixels land in tiles a's prepass never marked
        _device_view(ptr, n).copy_(merged)
        torch.cuda.synchronize()
        ctx.frame_turn(a, a)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(a, want)), k
        assert np.all(ctx.read_framebuffer(full=True) == EMPTY), k
    ctx.framebuffer_private()
    ctx.frame_begin(a); ctx.render_basic(a); ctx.frame_turn(a, b)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(a, fa))
    ctx.render_basic(b); ctx.frame_turn(b, a)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(b, fb_b))
    # the empty word changes between two turns: the clear that follows is a full one
    ctx.set_int64_mergeable(True)
    ctx.render_basic(a); ctx.frame_turn(a, b)
    assert np.all(ctx.read_framebuffer(full=True) == EMPTY)     # (reads map the context's empty word to the reference's)
    # ... and back: a tile-limited turn would leave INT64_MAX words in the tiles nothing was drawn in, visible as pixels now
    ctx.set_int64_mergeable(False)
    ctx.render_basic(b); ctx.frame_turn(b, a)
    assert np.all(ctx.read_framebuffer(full=True) == EMPTY)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(b, fb_b))


def _device_view(ptr, n):
    """torch int64 view of n words of device memory at ptr (the integrator's side of pcr_device_framebuffer)."""
    import torch

    class _Arr:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Arr(), device="cuda")
