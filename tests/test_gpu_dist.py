"""The multi-GPU code path on the one GPU a test box has: a single-rank RCCL group, torch-owned int64-mergeable
framebuffers (DeviceFrame / SlicedFrame) and the one-stream sharded forms, against the oracle. The N > 1
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


def test_slice_kernels_against_numpy(group, loaded):
    """pcr_merge_min_slices / pcr_resolve_basic_range on their own: three slices of random keys (empty = INT64_MAX)."""
    import torch
    ctx, _ = loaded
    rng = np.random.default_rng(5)
    S, ns = 10_002, 3
    # depth half: any positive float's bits; payload half: small enough for the showNumPoints shade (an int in the reference)
    keys = (rng.integers(1, 0x7F800000, size=(ns, S), dtype=np.int64) << 32) | rng.integers(0, 2 ** 20, size=(ns, S), dtype=np.int64)
    keys[rng.random((ns, S)) < 0.4] = np.iinfo(np.int64).max
    t = torch.from_numpy(keys.reshape(-1).copy()).to(group)
    stream = torch.cuda.Stream(group)           # (torch's default stream has the handle 0 = "the context's own stream")
    stream.wait_stream(torch.cuda.current_stream(group))
    ctx.set_stream(stream.cuda_stream)
    try:
        ctx.merge_min_slices(t.data_ptr(), ns, S)
        want = keys.min(axis=0)
        for flags in ({}, {"show_num_points": 1}, {"colorize_chunks": 1}):
            p = scenes.with_flags(scenes.cameras(W, H)["overview"], **flags)
            out = torch.empty(S, dtype=torch.int32, device=group)
            ctx.resolve_basic_range(p, t.data_ptr(), S, out.data_ptr())
            torch.cuda.synchronize()
            fbw = np.where(want == np.iinfo(np.int64).max, -1, want).view(np.uint64)
            padded = np.full(W * (H + 1) + 1, 2 ** 64 - 1, dtype=np.uint64)
            m = min(S, W * H)
            padded[:m] = fbw[:m]
            ref = oracle.resolve_basic(p, padded).ravel()[:m]
            assert np.array_equal(out.cpu().numpy().view(np.uint32)[:m], ref), flags
        assert np.array_equal(t.cpu().numpy()[:S], want)
    finally:
        ctx.set_stream(0)


def test_sliced_form_on_a_single_rank_group(group, loaded):
    import torch
    ctx, of = loaded
    p = scenes.with_flags(scenes.cameras(W, H)["closeup"], lod_percent=100, cull=1)
    frame = pdist.SlicedFrame(ctx, W, H, group, 1)
    try:
        frame.bind()
        pdist.render_basic_sharded(ctx, frame, p, 1, merge="sliced")
        torch.cuda.synchronize()
        ofb, _ = of.render_basic(p)
        merged = frame.gather_merged_framebuffer().cpu().numpy()
        assert np.array_equal(np.where(merged == np.iinfo(np.int64).max, -1, merged).view(np.uint64), ofb)
        want = oracle.resolve_basic(p, ofb).ravel()
        assert np.array_equal(frame.image().cpu().numpy().view(np.uint32)[:want.size], want)
    finally:
        frame.release()
