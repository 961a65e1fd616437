"""One rank's share of BASELINE.json configs[3] / configs[4] on the one GPU a test box has: the 2e9-point scene cut into
eight contiguous batch ranges (6 x 3815 + 2 x 3814 batches), of which rank 0's (followed by rank 1's head
words, pcr_upload_tail -- the reference's pad that this stands in for: modules/compute/HuffmanLasLoader.cpp:39-41, the
over-reading fetch huffman_mem_iter_cuda/render.cu:441-450) and the last rank's (zero pad) are drawn at 1920x1080
(LOD 100 %, no culling), at 4096x4096 with the frustum cull on and a camera that culls about half of the shard, and at
1920x1080 with LOD 25 % + culling (all four work classes of the longest-first order), by
both decode variants, against the oracle on the same shard; two adjacent shards merged with pcr_merge_min equal the
oracle over their union. A shard holds more than 2048 batches, so k_render's batch search runs its second round of 64
chunk counts and the prepass more than 64 workgroups."""
import numpy as np
import pytest

import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
from tests import oracle, scenes

pytestmark = pytest.mark.gpu

TOTAL = 2_000_000_000
CHUNK = 6_553_600                    # points per Morton-sorted chunk = 100 batches (src/preprocess.cpp:1194-1200)
NB_TOTAL = 30_518
WORLD = 8


def shard(rank):
    first, count = pdist.shard_range(NB_TOTAL, WORLD, rank)
    return first, count


def generate(batch_first, batch_end):
    """The chunks that hold batches [batch_first, batch_end) of the global stream -> (HuffmanFile, OracleFile, first batch of the image)."""
    c0, c1 = batch_first // 100, -(-batch_end // 100)
    first = c0 * CHUNK
    count = min(TOTAL, c1 * CHUNK) - first
    image, st = P.synth_encode(TOTAL, scenes.SEED, first, count, CHUNK, 16)
    return P.HuffmanFile(image), oracle.OracleFile(image.view()), c0 * 100


def load(ctx, hf, local_first, count, global_first, with_tail):
    ctx.stream_begin(hf.header(local_first, count), global_first)
    for b0 in range(0, count, 100):
        ctx.upload_batches(b0, [hf.blob(local_first + b) for b in range(b0, min(b0 + 100, count))])
    if with_tail:
        ctx.upload_tail(*hf.head_words(local_first + count))
    assert ctx.batches_loaded == count > 2048


def camera_half(y_centre, w, h):
    return P.camera_orbit(0.6, -0.5, 420.0, (250.0, y_centre, 40.0), w, h)


def check_frames(ctx, of, local_first, count, y_centre):
    """1080p LOD 100 cull 0, then 4096x4096 cull 1 with about half of the shard's batches culled; both decode variants."""
    cases = (("1080p", 1920, 1080, scenes.with_flags(scenes.cameras(1920, 1080)["overview"], lod_percent=100, cull=0)),
             ("4096 cull", 4096, 4096, scenes.with_flags(camera_half(y_centre, 4096, 4096), lod_percent=100, cull=1)),
             # a level of detail: the drawn batches fall into all four work classes (points per chain), which k_render's search walks
             # heaviest first, two rounds of chunk counts per class (RenderArgs::work_classes)
             ("1080p lod 25 cull", 1920, 1080, scenes.with_flags(camera_half(y_centre, 1920, 1080), lod_percent=25, cull=1)))
    for name, w, h, p in cases:
        ctx.set_image_size(w, h)
        ofb, ost = of.render_basic(p, first=local_first, count=count, nthreads=16)
        assert ost["batches_total"] == count
        if name == "4096 cull":
            assert 0.25 * count < ost["batches_culled"] < 0.75 * count, ost
        elif name == "1080p lod 25 cull":
            drawn = count - ost["batches_culled"]
            assert drawn > 0.2 * count and 0.03 * drawn * 65536 < ost["points_iterated"] < 0.9 * drawn * 65536, ost
        else:
            assert ost["points_iterated"] == count * 65536
        for variant, parts in ((P.Context.VARIANT_POINT_WINDOWS, 0), (P.Context.VARIANT_WORDS, 0), (P.Context.VARIANT_POINT_WINDOWS, 1)):
            ctx.set_render_variant(variant)
            ctx.set_workgroup_parts(parts)          # 0: the library's choice (half-batches), 1: one 1024-thread workgroup per batch
            ctx.frame_begin(p); ctx.render_basic(p); ctx.resolve_basic(p)
            assert ctx.stats() == ost, (name, variant)
            assert np.array_equal(ctx.read_framebuffer(full=True), ofb), (name, variant)
            assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb)), (name, variant)
    ctx.set_render_variant(P.Context.VARIANT_AUTO)
    ctx.set_workgroup_parts(0)


@pytest.fixture(scope="module")
def front():
    """Ranks 0 and 1 of the eight: batches [0, 7630) of the global stream (+ the chunk remainder behind them)."""
    f1, n1 = shard(1)
    return generate(0, f1 + n1)


def test_rank0_shard_with_the_followers_head_words(front):
    hf, of, base = front
    first, count = shard(0)
    assert (first, count) == (0, 3815) and base == 0
    ctx = P.Context(0)
    try:
        ctx.set_image_size(1920, 1080)
        ctx.set_stream_layout(P.Context.LAYOUT_BOTH)
        load(ctx, hf, first, count, first, with_tail=True)
        check_frames(ctx, of, first, count, 60.0)
    finally:
        ctx.close()


def test_two_adjacent_shards_merge_to_the_oracle_over_their_union(front):
    hf, of, base = front
    (f0, n0), (f1, n1) = shard(0), shard(1)
    assert f1 == f0 + n0 and n1 == 3815
    p = scenes.with_flags(scenes.cameras(1920, 1080)["overview"], lod_percent=100, cull=0)
    parts = []
    try:
        for first, count in ((f0, n0), (f1, n1)):
            c = P.Context(0)
            parts.append(c)
            c.set_image_size(1920, 1080)
            load(c, hf, first - base, count, first, with_tail=True)     # (rank 1's follower is in the image's chunk remainder)
            c.clear(); c.render_basic(p)
        parts[1].synchronize()
        parts[0].merge_min(parts[1].device_framebuffer())
        merged = parts[0].read_framebuffer(full=True)
    finally:
        for c in parts:
            c.close()
    ofb, ost = of.render_basic(p, first=f0 - base, count=n0 + n1, nthreads=16)
    assert ost["points_iterated"] == (n0 + n1) * 65536
    assert np.array_equal(merged, ofb)


def test_last_rank_shard_with_the_zero_pad():
    first, count = shard(WORLD - 1)
    assert (first, count) == (26_704, 3814) and first + count == NB_TOTAL
    hf, of, base = generate(first, first + count)
    assert base + hf.numBatches == NB_TOTAL
    ctx = P.Context(0)
    try:
        ctx.set_image_size(1920, 1080)
        ctx.set_stream_layout(P.Context.LAYOUT_BOTH)
        load(ctx, hf, first - base, count, first, with_tail=False)     # the stream ends here: the loader's zero pad follows
        check_frames(ctx, of, first - base, count, 940.0)
    finally:
        ctx.close()
