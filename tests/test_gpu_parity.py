"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact, on seeded inputs."""
import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, scenes

pytestmark = pytest.mark.gpu

W, H = 640, 360


# Every test of this module runs once per decode variant: "point_windows" (the default layout's own kernel), "words" on a
# stream loaded with PCR_LAYOUT_BOTH (what PCR_VARIANT_AUTO then picks for images with few batches per pixel), and "words_only"
# on a stream loaded with PCR_LAYOUT_WORDS (no point windows in HBM at all). Those draw with the library's choice of workgroup
# shape (half-batches: two 512-thread workgroups per batch); "*_whole" forces the reference's shape, one 1024-thread workgroup per
# batch (pcr_set_workgroup_parts).
@pytest.fixture(scope="module", params=["point_windows", "words", "words_only", "point_windows_whole", "words_whole"])
def renderer(request):
    r = P.Renderer(W, H, device=0)
    kind = request.param
    if kind.endswith("_whole"):
        r.ctx.set_workgroup_parts(1)
        kind = kind[:-len("_whole")]
    request = type("R", (), {"param": kind})
    if request.param == "words_only":
        r.ctx.set_stream_layout(P.Context.LAYOUT_WORDS)
    else:
        if request.param == "words":
            r.ctx.set_stream_layout(P.Context.LAYOUT_BOTH)
        r.ctx.set_render_variant(P.Context.VARIANT_POINT_WINDOWS if request.param == "point_windows" else P.Context.VARIANT_WORDS)
    yield r
    r.ctx.close()


@pytest.fixture(scope="module")
def stream200k():
    """4 batches; every batch huge on screen (double path, 64 points per chain)."""
    nb, st = scenes.synth_stream(200_000)
    return nb, oracle.OracleFile(nb.view())


@pytest.fixture(scope="module")
def stream2m():
    """31 batches: mixes float/double paths, LOD 6..64 points per chain, culled and straddling batches."""
    nb, st = scenes.synth_stream(2_000_000)
    return nb, oracle.OracleFile(nb.view())


def _load(renderer, nb):
    P.Runtime.reset()
    las = P.HuffmanLasData.create(nb)
    if renderer.ctx.batches_loaded:
        renderer.ctx.stream_unload()
    las.load_all(renderer)
    return las


def _check_basic(ctx, of, p):
    ctx.clear()
    ctx.render_basic(p)
    ctx.resolve_basic(p)
    fb = ctx.read_framebuffer(full=True)
    ofb, ost = of.render_basic(p)
    assert ctx.stats() == ost
    diff = np.nonzero(fb != ofb)[0]
    assert diff.size == 0, f"{diff.size} framebuffer words differ, first at {diff[:5]}: gpu {fb[diff[:3]]} oracle {ofb[diff[:3]]}"
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb))
    return ost


def _check_hqs(ctx, of, p):
    ctx.clear()
    ctx.render_hqs_depth(p)
    st_gpu = ctx.stats()
    fb = ctx.read_framebuffer(full=True)
    ofb, ost = of.render_hqs_depth(p)
    assert st_gpu == ost
    assert np.array_equal(fb, ofb)
    ctx.render_hqs_color(p)
    ctx.resolve_hqs(p)
    rg, ba = ctx.read_accum(full=True)
    org, oba, _ = of.render_hqs_color(p, ofb)
    assert np.array_equal(rg, org) and np.array_equal(ba, oba)
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_hqs(p, ofb, org, oba))
    return ost


@pytest.mark.parametrize("cam", ["overview", "closeup", "inside", "far"])
@pytest.mark.parametrize("lod,cull", [(10, 1), (100, 0), (100, 1), (37, 1)])
def test_basic_matches_oracle(renderer, stream2m, cam, lod, cull):
    nb, of = stream2m
    _load(renderer, nb)
    p = scenes.with_flags(scenes.cameras(W, H)[cam], lod_percent=lod, cull=cull)
    st = _check_basic(renderer.ctx, of, p)
    assert st["batches_total"] == of.num_batches


@pytest.mark.parametrize("cam", ["overview", "closeup", "inside", "far"])
@pytest.mark.parametrize("lod", [10, 100])
def test_hqs_matches_oracle(renderer, stream2m, cam, lod):
    nb, of = stream2m
    _load(renderer, nb)
    p = scenes.with_flags(scenes.cameras(W, H)[cam], lod_percent=lod)
    _check_hqs(renderer.ctx, of, p)


def test_small_stream_all_double(renderer, stream200k):
    nb, of = stream200k
    _load(renderer, nb)
    for cam in ("overview", "closeup"):
        p = scenes.with_flags(scenes.cameras(W, H)[cam], lod_percent=10)
        _check_basic(renderer.ctx, of, p)
        _check_hqs(renderer.ctx, of, p)


@pytest.mark.parametrize("flag", ["show_num_points", "colorize_chunks"])
def test_debug_payload_modes(renderer, stream2m, flag):
    nb, of = stream2m
    _load(renderer, nb)
    p = scenes.with_flags(scenes.cameras(W, H)["closeup"], **{flag: 1})
    _check_hqs(renderer.ctx, of, p)
    _check_basic(renderer.ctx, of, p)


def test_random_escape_heavy_stream(renderer):
    """Unstructured points: ~every symbol escapes, large int32 deltas, negative coordinates."""
    x, y, z, c, las = scenes.random_points(150_000, seed=7)
    nb, st = P.encode_points(x, y, z, c, las, morton_sort=True, nthreads=2)
    assert st["escaped_symbols"] > 0.3 * st["total_symbols"]
    of = oracle.OracleFile(nb.view())
    _load(renderer, nb)
    for tgt, rad in (((100.0, 100.0, 100.0), 3000.0), ((100.0, 100.0, 100.0), 300.0)):
        p = scenes.with_flags(P.camera_orbit(0.3, -0.7, rad, tgt, W, H), lod_percent=100)
        _check_basic(renderer.ctx, of, p)
        _check_hqs(renderer.ctx, of, p)


def test_method_plugins_render_like_reference_session(stream200k):
    """Drive the path the way main.cpp does: Runtime.addMethod, update(), render() per frame."""
    nb, of = stream200k
    P.Runtime.reset()
    r = P.Renderer(W, H)
    try:
        las = P.HuffmanLasData.create(nb)
        P.Runtime.addMethod(P.HuffmanMemIter(r, las))
        P.Runtime.addMethod(P.HuffmanHQS(r, las))
        P.Runtime.addMethod(P.ComputeHuffman(r, las))         # "huffman_cuda", modules/huffman_cuda/huffman_cuda.h:66
        r.set_camera(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0))
        for name in ("huffman_mem_iter_cuda", "huffman_hqs", "huffman_cuda"):
            P.Runtime.setSelectedMethod(name)
            m = P.Runtime.getSelectedMethod()
            m.update(r)
            for _ in range(3):          # progressive loading: <= 100 batches per frame
                m.render(r)
            assert las.numBatchesLoaded == of.num_batches
            p = m.last_params
            fb = r.ctx.read_framebuffer(full=True)
            if name == "huffman_hqs":
                ofb, _ = of.render_hqs_depth(p)
                org, oba, _ = of.render_hqs_color(p, ofb)
                assert np.array_equal(fb, ofb)
                assert np.array_equal(r.ctx.read_rgba(), oracle.resolve_hqs(p, ofb, org, oba))
            else:
                ofb, _ = of.render_basic(p)
                assert np.array_equal(fb, ofb)
                assert np.array_equal(r.ctx.read_rgba(), oracle.resolve_basic(p, ofb))
    finally:
        r.ctx.close()
        P.Runtime.reset()


def test_error_behaviour(renderer, stream200k):
    nb, _ = stream200k
    ctx = renderer.ctx
    f = P.HuffmanFile(nb)
    if ctx.batches_loaded:
        ctx.stream_unload()
    p = scenes.cameras(W, H)["overview"]
    with pytest.raises(P.PcrError, match="no stream"):
        ctx.render_basic(p)
    ctx.stream_begin(f.header())
    with pytest.raises(P.PcrError, match="in order"):
        ctx.upload_batch(1, f.blob(1))
    with pytest.raises(P.PcrError, match="too short|does not match"):
        ctx.upload_batch(0, bytes(f.blob(0))[:-8])
    bad = bytearray(f.blob(0))
    bad[8:12] = (512).to_bytes(4, "little")           # num_threads
    with pytest.raises(P.PcrError, match="geometry"):
        ctx.upload_batch(0, bytes(bad))
    q = p.copy(); q.width = 13
    ctx.upload_batch(0, f.blob(0))
    with pytest.raises(P.PcrError, match="image size"):
        ctx.render_basic(q)
    with pytest.raises(P.PcrError, match="image size"):
        ctx.frame_begin(q)
    with pytest.raises(P.PcrError, match="unknown stream layout"):
        ctx.set_stream_layout(7)
    with pytest.raises(P.PcrError, match="unknown render variant"):
        ctx.set_render_variant(-1)
    with pytest.raises(P.PcrError):
        ctx.kernel_timing(-3)
    ctx.stream_unload()


def test_random_cameras_and_flags(renderer, stream2m):
    """Seeded random orbit cameras (near/inside/far, any direction), LOD and flag combinations: every frame of both
    methods equals the oracle's bit for bit."""
    nb, of = stream2m
    _load(renderer, nb)
    rng = np.random.default_rng(2024)
    for k in range(24):
        yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-1.5, 0.3)
        radius = float(10.0 ** rng.uniform(0.3, 4.2))                      # 2 m .. 16 km
        target = (rng.uniform(-200, 1200), rng.uniform(-200, 1200), rng.uniform(-50, 150))
        p = P.camera_orbit(yaw, pitch, radius, target, W, H, fovy=float(rng.uniform(10, 120)))
        p = scenes.with_flags(p, lod_percent=int(rng.choice([0, 3, 10, 50, 100])), cull=int(rng.integers(0, 2)),
                              show_num_points=int(rng.integers(0, 2)), colorize_chunks=int(rng.integers(0, 2)))
        _check_basic(renderer.ctx, of, p)
        if k % 3 == 0:
            _check_hqs(renderer.ctx, of, p)


def test_progressive_loading_frames_equal_the_truncated_stream(renderer):
    """Frames drawn while the resource is still loading (HuffmanLasLoader.cpp:301-313 hands over <= 100 records per
    frame): after k batches the image equals the oracle's render of the file cut after k batches (the last loaded
    batch's tail over-reads see the zero pad), and once the rest arrives that batch is re-walked and the frame equals
    the whole file's."""
    import struct
    nb, st = P.synth_encode(16_000_000, scenes.SEED, nthreads=8)            # 245 batches -> tasks of 100, 100, 45
    hf = P.HuffmanFile(nb)
    assert hf.numBatches == 245

    def truncated(k):
        h = hf.header(0, k)
        return (struct.pack("<5q", h.num_points, h.num_batches, h.encoded_bytes, h.separate_bytes, h.cluster_bytes)
                + hf.batch_data_sizes[:k].tobytes() + b"".join(bytes(hf.blob(b)) for b in range(k)))

    P.Runtime.reset()
    if renderer.ctx.batches_loaded:
        renderer.ctx.stream_unload()
    las = P.HuffmanLasData.create(nb)
    m = P.HuffmanMemIter(renderer, las)
    renderer.set_camera(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0))
    P.Debug.LOD, P.Debug.frustumCullingEnabled = 1.0, False
    try:
        m.update(renderer)
        seen = []
        for _ in range(4):
            m.render(renderer)
            k = las.numBatchesLoaded
            seen.append(k)
            of = oracle.OracleFile(truncated(k)) if k < hf.numBatches else oracle.OracleFile(nb.view())
            ofb, ost = of.render_basic(m.last_params, nthreads=8)
            assert renderer.ctx.stats() == ost
            assert np.array_equal(renderer.ctx.read_framebuffer(full=True), ofb), f"frame with {k} batches loaded"
        assert seen == [100, 200, 245, 245]
    finally:
        P.Debug.LOD, P.Debug.frustumCullingEnabled = 0.1, True
        las.unload(renderer)


def test_kernel_timing_counts_render_launches_only(renderer, stream200k):
    """pcr_kernel_timing_*: one event pair per decode+rasterize launch, none for clear/resolve, off by default."""
    nb, of = stream200k
    _load(renderer, nb)
    ctx = renderer.ctx
    renderer.set_camera(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0))
    p = renderer.render_params()
    ctx.clear(); ctx.render_basic(p)
    assert ctx.kernel_timing_read() == (0.0, 0)
    ctx.kernel_timing(True)
    for _ in range(5):
        ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)
    ms, n = ctx.kernel_timing_read()
    assert n == 5 and 0.0 < ms < 50.0
    for _ in range(70):
        ctx.clear(); ctx.render_basic(p)
    ms, n = ctx.kernel_timing_read()
    assert n == 64 and 0.0 < ms < 50.0
    ctx.kernel_timing(False)
    ctx.render_basic(p)
    assert ctx.kernel_timing_read() == (0.0, 0)
    _check_basic(ctx, of, p)


def test_async_loader_frames_draw_what_has_arrived(renderer):
    """pcr_set_async_upload: copies + transcode on the loader stream; a frame draws the first k batches
    whose tasks have completed, never the newest batch of an incomplete stream, and equals the oracle's render of
    batches [0, k) of the whole file; once everything is resident the frame is the whole file's."""
    import time
    nb, st = P.synth_encode(16_000_000, scenes.SEED, nthreads=8)            # 245 batches
    hf = P.HuffmanFile(nb)
    of = oracle.OracleFile(nb.view())
    total = hf.numBatches
    ctx = renderer.ctx
    P.Runtime.reset()
    if ctx.batches_loaded:
        ctx.stream_unload()
    renderer.set_camera(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0))
    p = renderer.render_params()
    p.lod_percent, p.enable_frustum_culling = 100, 0
    ctx.stream_begin(hf.header())
    ctx.set_async_upload(True)
    try:
        seen = []
        for b0 in range(0, total, 35):
            ctx.upload_batches(b0, [hf.blob(b) for b in range(b0, min(b0 + 35, total))])
            ctx.clear()
            ctx.render_basic(p)
            k = ctx.last_frame_batches
            loaded = ctx.batches_loaded
            assert k <= (loaded if loaded == total else loaded - 1)
            seen.append(k)
            fb = ctx.read_framebuffer(full=True)
            if k == 0:
                assert np.all(fb == np.uint64(0xFFFFFFFFFFFFFFFF))
                continue
            ofb, ost = of.render_basic(p, first=0, count=k, nthreads=8)
            assert ctx.stats() == ost
            assert np.array_equal(fb, ofb), f"frame with {k} of {loaded} handed-over batches"
        assert seen == sorted(seen)
        t0 = time.time()
        while ctx.batches_resident < total:
            assert time.time() - t0 < 30.0, "loader stream made no progress"
            time.sleep(0.001)
        _check_basic(ctx, of, p)
        assert ctx.last_frame_batches == total
        _check_hqs(ctx, of, p)
    finally:
        ctx.set_async_upload(False)
        ctx.stream_unload()


def test_frame_begin_equals_clear_then_prepass(renderer, stream2m):
    """pcr_frame_begin = pcr_clear + the prepass of the next render call in one launch: same frames and statistics; a
    render call with other parameters, another method or after a plain pcr_clear runs its own prepass."""
    nb, of = stream2m
    _load(renderer, nb)
    ctx = renderer.ctx
    cams = scenes.cameras(W, H)
    p = scenes.with_flags(cams["overview"], lod_percent=100, cull=0)
    q = scenes.with_flags(cams["closeup"], lod_percent=10, cull=1)
    for first, second in ((p, p), (p, q), (q, q)):
        ctx.render_basic(first)                         # leaves a dirty framebuffer behind
        ctx.frame_begin(first)
        ctx.render_basic(second)                        # second != first: the prepared prepass must not be used
        ofb, ost = of.render_basic(second)
        assert ctx.stats() == ost
        assert np.array_equal(ctx.read_framebuffer(full=True), ofb)
    # HQS: the depth pass takes the prepared prepass, the colour pass (other window size) runs its own
    ctx.frame_begin(q, hqs=True)
    ctx.render_hqs_depth(q)
    hfb, hst = of.render_hqs_depth(q)
    assert ctx.stats() == hst
    assert np.array_equal(ctx.read_framebuffer(full=True), hfb)
    ctx.render_hqs_color(q)
    org, oba, _ = of.render_hqs_color(q, hfb)
    rg, ba = ctx.read_accum(full=True)
    assert np.array_equal(rg, org) and np.array_equal(ba, oba)
    # prepared for basic, used by HQS depth (other LOD expression): must be recomputed
    ctx.frame_begin(q, hqs=False)
    ctx.render_hqs_depth(q)
    assert ctx.stats() == hst
    assert np.array_equal(ctx.read_framebuffer(full=True), hfb)
    _check_basic(ctx, of, p)


def test_frame_turn_equals_resolve_then_frame_begin(renderer, stream2m):
    """pcr_frame_turn = pcr_resolve_* of the finished frame + pcr_clear + the next frame's prepass, one launch: the image it
    leaves is the resolve's, the framebuffer is empty, the next render (same or other parameters) draws the oracle's frame;
    debug payload flags of the finished frame are honoured; a basic turn behind an HQS frame still zeroes RG/BA."""
    nb, of = stream2m
    _load(renderer, nb)
    ctx = renderer.ctx
    cams = scenes.cameras(W, H)
    p = scenes.with_flags(cams["overview"], lod_percent=100, cull=0)
    q = scenes.with_flags(cams["closeup"], lod_percent=10, cull=1)
    empty = np.uint64(0xFFFFFFFFFFFFFFFF)
    ctx.frame_begin(p)
    for done, nxt in ((p, p), (p, q), (q, q), (q, p)):
        ctx.render_basic(done)
        ofb, ost = of.render_basic(done)
        assert ctx.stats() == ost
        ctx.frame_turn(done, nxt)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(done, ofb))
        assert (ctx.read_framebuffer(full=True) == empty).all()
    ctx.render_basic(p)                                     # the prepass prepared by the last turn (for p) is the one used here
    assert np.array_equal(ctx.read_framebuffer(full=True), of.render_basic(p)[0])
    # HQS frames through turns
    ctx.clear(); ctx.frame_begin(q, hqs=True)
    for done, nxt in ((q, p), (p, q)):
        ctx.render_hqs_depth(done)
        hfb, _ = of.render_hqs_depth(done)
        assert np.array_equal(ctx.read_framebuffer(full=True), hfb)
        ctx.render_hqs_color(done)
        org, oba, _ = of.render_hqs_color(done, hfb)
        ctx.frame_turn(done, nxt, hqs=True)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_hqs(done, hfb, org, oba))
        rg, ba = ctx.read_accum(full=True)
        assert (ctx.read_framebuffer(full=True) == empty).all() and not rg.any() and not ba.any()
    # a basic turn right behind an HQS colour pass: RG/BA must come out zeroed as well
    ctx.render_hqs_depth(q); ctx.render_hqs_color(q)
    ctx.frame_turn(scenes.with_flags(q, colorize_chunks=1), p)
    rg, ba = ctx.read_accum(full=True)
    assert not rg.any() and not ba.any()
    ctx.render_basic(p)
    assert np.array_equal(ctx.read_framebuffer(full=True), of.render_basic(p)[0])


def test_hqs_colour_pass_reuses_the_depth_passes_prepass_only_for_the_same_frame(renderer, stream2m):
    """One cull/LOD prepass per HQS frame: the depth pass's prepass (run by pcr_frame_begin / pcr_frame_turn or by the pass
    itself) also writes the colour pass's LDS window plan, and the colour pass skips its own when it is handed the same
    parameters. With other parameters -- another level of detail, another camera -- it has to run its own: both against the
    oracle, on a 140 KiB-window frame size too (the window plan depends on the dynamic LDS size)."""
    nb, of = stream2m
    ctx = renderer.ctx
    _load(renderer, nb)
    cams = scenes.cameras(W, H)
    p = scenes.with_flags(cams["closeup"], lod_percent=100, cull=1)
    other = (scenes.with_flags(cams["closeup"], lod_percent=10, cull=1), scenes.with_flags(cams["overview"], lod_percent=100, cull=0))
    hfb, hst = of.render_hqs_depth(p)
    for begin in ("clear", "frame_begin", "frame_turn"):
        for q in (p,) + other:
            if begin == "clear":
                ctx.clear()
            elif begin == "frame_begin":
                ctx.frame_begin(p, hqs=True)
            else:
                ctx.frame_begin(other[1], hqs=True); ctx.render_hqs_depth(other[1]); ctx.render_hqs_color(other[1])
                ctx.frame_turn(other[1], p, hqs=True)
            ctx.render_hqs_depth(p)
            assert ctx.stats() == hst and np.array_equal(ctx.read_framebuffer(full=True), hfb), (begin,)
            ctx.render_hqs_color(q)
            org, oba, ost = of.render_hqs_color(q, hfb)
            rg, ba = ctx.read_accum(full=True)
            assert ctx.stats() == ost, (begin, q.lod_percent)
            assert np.array_equal(rg, org) and np.array_equal(ba, oba), (begin, q.lod_percent)
    # a second colour pass over the same depth buffer adds the same sums once more (nothing stale is reused)
    ctx.clear(); ctx.render_hqs_depth(p); ctx.render_hqs_color(p); ctx.render_hqs_color(p)
    org, oba, _ = of.render_hqs_color(p, hfb)
    rg, ba = ctx.read_accum(full=True)
    assert np.array_equal(rg, 2 * org) and np.array_equal(ba, 2 * oba)
