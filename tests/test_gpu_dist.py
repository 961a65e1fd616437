"""The multi-GPU code path on the one GPU a test box has: a single-rank RCCL group, torch-owned int64-mergeable
framebuffers (DeviceFrame), the one-stream sharded forms and the pipelined renderer, against the oracle. The N > 1
arithmetic (shard ranges, head exchange, min/sum merges) is covered with gloo in tests/test_dist_cpu.py."""
import os

import numpy as np
import pytest

import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
from tests import oracle, scenes

pytestmark = pytest.mark.gpu

W, H = 640, 360


@pytest.fixture(scope="module")
def group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def loaded():
    nb, st = scenes.synth_stream(2_000_000)
    of = oracle.OracleFile(nb.view())
    ctx = P.Context(0)
    ctx.set_image_size(W, H)
    hf = P.HuffmanFile(nb)
    ctx.stream_begin(hf.header())
    ctx.upload_batches(0, [hf.blob(b) for b in range(hf.numBatches)])
    yield ctx, of
    ctx.close()


@pytest.mark.parametrize("merge", ["reduce", "allreduce"])
def test_sharded_forms_on_a_single_rank_group(group, loaded, merge):
    import torch
    ctx, of = loaded
    p = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100, cull=0)
    frame = pdist.DeviceFrame(ctx, W, H, group)
    try:
        frame.bind()
        pdist.render_basic_sharded(ctx, frame, p, 1, merge=merge)
        torch.cuda.synchronize()
        ofb, _ = of.render_basic(p)
        assert np.array_equal(ctx.read_framebuffer(full=True), ofb)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb))
        # the tensor RCCL reduces holds the same words with INT64_MAX in the empty pixels: signed order == unsigned order
        raw = frame.fb.cpu().numpy()
        assert raw.min() >= 0
        n = ofb.size
        assert np.array_equal(np.where(raw[:n] == np.iinfo(np.int64).max, -1, raw[:n]).view(np.uint64), ofb)

        pdist.render_hqs_sharded(ctx, frame, p, 1, merge=merge)
        torch.cuda.synchronize()
        hfb, _ = of.render_hqs_depth(p)
        org, oba, _ = of.render_hqs_color(p, hfb)
        assert np.array_equal(ctx.read_framebuffer(full=True), hfb)
        rg, ba = ctx.read_accum(full=True)
        assert np.array_equal(rg, org) and np.array_equal(ba, oba)
        assert np.array_equal(ctx.read_rgba(), oracle.resolve_hqs(p, hfb, org, oba))
    finally:
        frame.release()
    # back on its own buffers the context clears to all-ones again
    ctx.clear(); ctx.render_basic(p)
    assert np.array_equal(ctx.read_framebuffer(full=True), of.render_basic(p)[0])


def test_pipelined_renderer_frames(group, loaded):
    ctx, of = loaded
    cams = scenes.cameras(W, H)
    pipe = pdist.PipelinedBasicRenderer(ctx, W, H, group, merge="reduce")
    try:
        seq = [scenes.with_flags(cams[name], lod_percent=lod, cull=cull)
               for name, lod, cull in (("overview", 100, 0), ("closeup", 10, 1), ("overview", 10, 1), ("closeup", 100, 0))]
        for p in seq:
            pipe.step(p)
            pipe.finish()
            f = pipe.last_frame()
            f.bind()
            ofb, _ = of.render_basic(p)
            assert np.array_equal(ctx.read_framebuffer(full=True), ofb)
            assert np.array_equal(ctx.read_rgba(), oracle.resolve_basic(p, ofb))
        # back-to-back frames without a fence in between: the last one is what the buffers hold
        for p in seq:
            pipe.step(p)
        pipe.finish()
        pipe.last_frame().bind()
        assert np.array_equal(ctx.read_framebuffer(full=True), of.render_basic(seq[-1])[0])
    finally:
        pipe.release()
