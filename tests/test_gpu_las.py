"""GPU parity of the 10-10-10 path ("loop_las_cuda"): HIP kernels through the C ABI against the CPU oracle, bit-exact."""
import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, scenes

pytestmark = pytest.mark.gpu

W, H = 640, 360
PPB = 65536


@pytest.fixture(scope="module")
def renderer():
    r = P.Renderer(W, H, device=0)
    yield r
    r.ctx.close()


def _points(total, order):
    x, y, z, c = P.synth_points(total, scenes.SEED, 0, total)
    las = P.synth_las_info(total, scenes.SEED)
    if order == "tiles":          # compact batches (as after a spatial sort): many small boxes, all levels appear
        key = (y // 40000).astype(np.int64) * 1000 + x // 40000
        idx = np.argsort(key, kind="stable")
        x, y, z, c = x[idx], y[idx], z[idx], c[idx]
    return x, y, z, c, las


@pytest.fixture(scope="module", params=["strips", "tiles"])
def cloud(request):
    x, y, z, c, las = _points(2_000_000, request.param)
    return (x, y, z, c, las), P.las_quantize(x, y, z, c, las)


def _load(renderer, pts):
    P.Runtime.reset()
    las = P.ComputeLasData.from_points(*pts)
    m = P.ComputeLoopLasCUDA(renderer, las)
    P.Runtime.addMethod(m)
    P.Runtime.setSelectedMethod("loop_las_cuda")
    m.update(renderer)
    while las.state == P.Resource.LOADING:
        las.process(renderer)
    return las, m


def _check(ctx, q, p):
    batches, x12, x8, x4, rgba = q
    ctx.clear()
    ctx.render_las(p)
    ctx.resolve_las(p)
    fb = ctx.read_framebuffer(full=True)
    ofb, ost = oracle.render_las(batches, x12, x8, x4, p)
    assert ctx.stats() == ost
    diff = np.nonzero(fb != ofb)[0]
    assert diff.size == 0, f"{diff.size} framebuffer words differ, first at {diff[:5]}: gpu {fb[diff[:3]]} oracle {ofb[diff[:3]]}"
    assert np.array_equal(ctx.read_rgba(), oracle.resolve_las(p, ofb, rgba))
    return ost


@pytest.mark.parametrize("cam", ["overview", "closeup", "inside", "far"])
@pytest.mark.parametrize("cull", [0, 1])
def test_las_matches_oracle(renderer, cloud, cam, cull):
    pts, q = cloud
    las, _ = _load(renderer, pts)
    assert las.numBatchesLoaded == len(q[0]) == 31
    p = scenes.with_flags(scenes.cameras(W, H)[cam], cull=cull)
    st = _check(renderer.ctx, q, p)
    assert st["batches_total"] == 31
    levels = [oracle.las_level(q[0][b], p) for b in range(30)]
    expect = sum(64 + PPB * 4 * (1 if l >= 2 else 2 if l == 1 else 3) for l in levels if l >= 0)
    assert renderer.ctx.las_algorithmic_bytes == expect


def test_las_method_frame_and_progressive_loading(renderer):
    """Runtime/Method surface: frames while the resource is still loading draw only complete batches but the last."""
    total = 150 * PPB + 777                     # two loader tasks; ragged final batch
    x, y, z, c = P.synth_points(total, scenes.SEED, 0, total)
    pts = (x, y, z, c, P.synth_las_info(total, scenes.SEED))
    P.Runtime.reset()
    if P.Runtime.resource is not None:
        P.Runtime.resource.unload(renderer)
    las = P.ComputeLasData.from_points(*pts)
    m = P.ComputeLoopLasCUDA(renderer, las)
    assert (m.name, m.group) == ("loop_las_cuda", "10-10-10 bit encoded")
    renderer.set_camera(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0))
    P.Debug.frustumCullingEnabled = True
    m.update(renderer)
    q = P.las_quantize(*pts)
    loaded = []
    for _ in range(3):
        m.render(renderer)                      # process() + CLEAR + RENDER + RESOLVE
        loaded.append(las.numBatchesLoaded)
        fb = renderer.ctx.read_framebuffer(full=True)
        ofb, ost = oracle.render_las(*q[:4], m.last_params, num_batches=las.numBatchesLoaded)
        assert renderer.ctx.stats() == ost
        assert np.array_equal(fb, ofb)
        assert np.array_equal(renderer.ctx.read_rgba(), oracle.resolve_las(m.last_params, ofb, q[4]))
    assert loaded == [100, 151, 151] and las.state == P.Resource.LOADED
    las.unload(renderer)
    assert renderer.ctx.las_batches_loaded == 0


def test_las_window_overflow_and_tiny_images(renderer):
    """Batches whose rectangle exceeds the LDS window (partial window + global atomics) and degenerate image sizes."""
    pts, q = _points(400_000, "tiles"), None
    q = P.las_quantize(*pts)
    for (w, h) in ((1920, 1080), (64, 48), (1, 1)):
        r = P.Renderer(w, h, device=0)
        try:
            las, _ = _load(r, pts)
            for cam in ("closeup", "overview"):
                _check(r.ctx, q, scenes.with_flags(scenes.cameras(w, h)[cam], cull=0))
        finally:
            r.ctx.close()


def test_las_argument_errors(renderer):
    ctx = renderer.ctx
    ctx.las_unload()
    p = scenes.cameras(W, H)["overview"]
    with pytest.raises(P.PcrError, match="no 10-10-10 data"):
        ctx.render_las(p)
    with pytest.raises(P.PcrError, match="num_points"):
        ctx.las_begin(0)
    with pytest.raises(P.PcrError, match="31-bit"):
        ctx.las_begin(1 << 31)
    pts = _points(PPB, "strips")
    q = P.las_quantize(*pts)
    ctx.las_begin(PPB)
    with pytest.raises(P.PcrError, match="out of order"):
        ctx.las_upload(1, *q)
    bad = (P.XyzBatch * 1)()
    bad[0].min_x, bad[0].max_x = 1.0, 0.0
    with pytest.raises(P.PcrError, match="bad bounding box"):
        ctx.las_upload(0, bad, *q[1:])
    ctx.las_upload(0, *q)
    with pytest.raises(P.PcrError, match="out of order"):
        ctx.las_upload(1, *q)
    bad_p = p.copy()
    bad_p.width = 10
    with pytest.raises(P.PcrError, match="image size"):
        ctx.render_las(bad_p)
    ctx.clear()
    ctx.render_las(p)                       # a single batch: it is the last one, nothing is drawn
    assert ctx.stats()["points_iterated"] == 0
    assert (ctx.read_framebuffer() == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    ctx.las_unload()


def test_las_random_cameras(renderer, cloud):
    """Seeded random orbit cameras over both point orders: all three precision levels, culling on and off."""
    pts, q = cloud
    _load(renderer, pts)
    rng = np.random.default_rng(77)
    levels = set()
    for _ in range(16):
        yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-1.5, 0.3)
        radius = float(10.0 ** rng.uniform(0.3, 4.2))
        target = (rng.uniform(-200, 1200), rng.uniform(-200, 1200), rng.uniform(-50, 150))
        p = scenes.with_flags(P.camera_orbit(yaw, pitch, radius, target, W, H, fovy=float(rng.uniform(10, 120))), cull=int(rng.integers(0, 2)))
        _check(renderer.ctx, q, p)
        levels |= {oracle.las_level(q[0][b], p) for b in range(len(q[0]))}
    assert {0, 1, 2} <= levels


def test_las_several_chunks_of_the_draw_list():
    """More than 256 batches (two chunks of the prepass's compacted list), a tile-ordered cloud whose batches straddle the ends of the
    tile rows: a few heavy batches among light ones (drawn heaviest class first), a close-up where every batch is heavy (drawn in the
    file's order), culling on and off -- against the oracle."""
    total = 300 * PPB + 123
    x, y, z, c = P.synth_points(total, scenes.SEED, 0, total)
    las = P.synth_las_info(total, scenes.SEED)
    side = 57_000                                # ~65 536 points per square tile of the 1e6-unit scene at this density
    idx = np.argsort((y // side).astype(np.int64) * 4096 + x // side, kind="stable")
    pts = (x[idx], y[idx], z[idx], c[idx], las)
    q = P.las_quantize(*pts)
    assert len(q[0]) == 301
    r = P.Renderer(1920, 1080, device=0)
    try:
        _load(r, pts)
        for cam in ("overview", "closeup"):
            for cull in (0, 1):
                st = _check(r.ctx, q, scenes.with_flags(scenes.cameras(1920, 1080)[cam], cull=cull))
                assert st["batches_total"] == 301
    finally:
        r.ctx.close()


def test_las_degenerate_matrices(renderer, cloud):
    """rasterize() rejects w <= 0 (render.cu:113): under an all-zero matrix x == y == w == 0 passes the kernels' |x| <= w test and
    has to be taken out again by the slow division path; matrices scaled by 2^-80 / 2^80 take that path for every point."""
    pts, q = cloud
    _load(renderer, pts)
    base = scenes.with_flags(scenes.cameras(W, H)["overview"], cull=0)
    zero = base.copy()
    for k in range(16):
        zero.transform[k] = 0.0
    st = _check(renderer.ctx, q, zero)
    assert (renderer.ctx.read_framebuffer(full=True) == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and st["points_iterated"] > 0
    for scale in (2.0 ** -80, 2.0 ** 80):
        p = base.copy()
        for k in range(16):
            p.transform[k] = base.transform[k] * scale
        _check(renderer.ctx, q, p)
