"""Multi-process (gloo, world_size 2) tests of the sharding + framebuffer-merge layer on CPU.
Each rank renders its contiguous batch range with the oracle (standing in for a GPU), merges through
pcrhpg24_amd.dist, and every rank must end with the single-process result, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
from tests import oracle, scenes

W, H = 320, 180


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image, _ = P.synth_encode(total, scenes.SEED, nthreads=2)
        of = oracle.OracleFile(image.view())
        first, count = pdist.shard_range(of.num_batches, world, rank)
        cam = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100)
        # basic: shard render -> min merge
        fb, _ = of.render_basic(cam, first=first, count=count)
        pdist.allreduce_min_u64_numpy(fb)
        # HQS: depth -> min merge -> colour against the GLOBAL depth -> sum merge
        d, _ = of.render_hqs_depth(cam, first=first, count=count)
        pdist.allreduce_min_u64_numpy(d)
        rg, ba, _ = of.render_hqs_color(cam, d, first=first, count=count)
        rg_root, ba_root = rg.copy(), ba.copy()
        pdist.allreduce_sum_u64_numpy(rg); pdist.allreduce_sum_u64_numpy(ba)
        # reduce-to-root forms (what bench.py uses: the finished frame lives on the display rank)
        fb_root, _ = of.render_basic(cam, first=first, count=count)
        pdist.reduce_min_u64_numpy(fb_root, dst=0)
        pdist.reduce_sum_u64_numpy(rg_root, dst=0); pdist.reduce_sum_u64_numpy(ba_root, dst=0)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), fb=fb, d=d, rg=rg, ba=ba, fb_root=fb_root, rg_root=rg_root, ba_root=ba_root)
    finally:
        dist.destroy_process_group()


def _heads_worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank encodes only its own chunk range, as bench.py does
        chunk = 65536 * 2
        nchunks = -(-total // chunk)
        c0, cn = pdist.shard_range(nchunks, world, rank)
        first = c0 * chunk
        image, _ = P.synth_encode(total, scenes.SEED, first, min(total, (c0 + cn) * chunk) - first, chunk, 2)
        hf = P.HuffmanFile(image)
        nxt = pdist.exchange_shard_heads(*hf.head_words(0), "cpu")
        if nxt is None:
            np.savez(os.path.join(out_dir, f"heads{rank}.npz"), none=np.ones(1))
        else:
            np.savez(os.path.join(out_dir, f"heads{rank}.npz"), enc=nxt[0], sep=nxt[1])
    finally:
        dist.destroy_process_group()


def test_shard_heads_exchange_delivers_the_followers_first_words(tmp_path):
    total, world, chunk = 700_000, 3, 65536 * 2          # 6 chunks -> 2 + 2 + 2
    mp.spawn(_heads_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    whole, _ = P.synth_encode(total, scenes.SEED, chunk_points=chunk, nthreads=2)
    hf = P.HuffmanFile(whole)
    per_chunk = chunk // 65536
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"heads{rank}.npz"))
        if rank == world - 1:
            assert "none" in z
            continue
        c0, cn = pdist.shard_range(6, world, rank)
        enc, sep = hf.head_words((c0 + cn) * per_chunk)                 # first batch of the following rank in the global file
        assert np.array_equal(z["enc"], enc) and np.array_equal(z["sep"], sep)
        assert z["enc"].dtype == np.uint32 and z["sep"].dtype == np.int32


def test_shard_range_partitions_exactly():
    for n in (1, 7, 31, 1526, 30518):
        for w in (1, 2, 3, 8):
            r = [pdist.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and sum(c for _, c in r) == n
            assert all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(c for _, c in r) - min(c for _, c in r) <= 1


def test_two_ranks_reproduce_single_process_render(tmp_path):
    total = 600_000                         # 10 batches -> 5 + 5
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    image, _ = P.synth_encode(total, scenes.SEED, nthreads=2)
    of = oracle.OracleFile(image.view())
    cam = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100)
    fb, _ = of.render_basic(cam)
    d, _ = of.render_hqs_depth(cam)
    rg, ba, _ = of.render_hqs_color(cam, d)
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert np.array_equal(z["fb"], fb), "merged basic framebuffer differs from the single-process render"
        assert np.array_equal(z["d"], d) and np.array_equal(z["rg"], rg) and np.array_equal(z["ba"], ba)
        if rank == 0:
            assert np.array_equal(z["fb_root"], fb) and np.array_equal(z["rg_root"], rg) and np.array_equal(z["ba_root"], ba)


def test_shard_upload_view_equals_the_global_stream_segment():
    """What a rank uploads for batches [first, first+count) + the follower's head words is byte-identical to the
    same window of the single-stream layout, so tail over-reads see the same words on any rank."""
    image, _ = scenes.synth_stream(600_000)
    f = P.HuffmanFile(image)
    of = oracle.OracleFile(image.view())
    e, s = of.encoded(), of.separate()
    first, count = pdist.shard_range(f.numBatches, 2, 0)
    hdr = f.header(first, count)
    g0, g1 = of.batch(first), of.batch(first + count)
    assert hdr.encoded_bytes == 4 * (g1.encoding_batch_offset - g0.encoding_batch_offset)
    assert hdr.separate_bytes == 4 * (g1.separate_batch_offset - g0.separate_batch_offset)
    he, hs = f.head_words(first + count)
    assert np.array_equal(he, e[g1.encoding_batch_offset:g1.encoding_batch_offset + len(he)])
    assert np.array_equal(hs, s[g1.separate_batch_offset:g1.separate_batch_offset + len(hs)])
    assert len(he) == 1024 and len(hs) <= 256


def test_sign_flip_min_is_unsigned_min():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 2 ** 64, 1000, dtype=np.uint64); b = rng.integers(0, 2 ** 64, 1000, dtype=np.uint64)
    a[:10] = 0xFFFFFFFFFFFFFFFF
    fa, fb = (a ^ pdist.SIGN).view(np.int64), (b ^ pdist.SIGN).view(np.int64)
    assert np.array_equal(np.minimum(fa, fb).view(np.uint64) ^ pdist.SIGN, np.minimum(a, b))


def _a2a_worker(rank, world, port, total, out_dir):
    """The all-to-all form of the merge (pdist.a2a_merge) over gloo: each rank renders its shard with the oracle into an
    int64-mergeable frame (empty = INT64_MAX), the element-wise steps run as torch/numpy stand-ins for the HIP kernels."""
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image, _ = P.synth_encode(total, scenes.SEED, nthreads=2)
        of = oracle.OracleFile(image.view())
        first, count = pdist.shard_range(of.num_batches, world, rank)
        cam = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100)
        part, _ = of.render_basic(cam, first=first, count=count)
        n = part.size
        S = pdist.slice_elems(n, world)
        big = np.iinfo(np.int64).max
        fb = torch.full((world * S,), big, dtype=torch.int64)
        fb[:n] = torch.from_numpy(np.where(part == np.uint64(2 ** 64 - 1), np.uint64(big), part).view(np.int64))
        recv = torch.empty(world * S, dtype=torch.int64)
        rgba_slice = torch.empty(S, dtype=torch.int32)
        rgba = torch.empty(world * S, dtype=torch.int32)

        def min_slices(t, ns, s):
            t[:s] = t.view(ns, s).min(dim=0).values

        def resolve_range(t, s, out):                      # resolve.cu:149-191, BC1 mode, no debug flags
            lo = (t[:s].numpy().view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            out[:] = torch.from_numpy(np.where(lo != 0xFFFFFFFF, lo, np.uint32(0x00443322)).view(np.int32))

        pdist.a2a_merge(fb, recv, rgba_slice, rgba, world, min_slices, resolve_range)
        merged = torch.empty(world * S, dtype=torch.int64)
        dist.all_gather_into_tensor(merged, recv[:S].contiguous())
        np.savez(os.path.join(out_dir, f"a2a{rank}.npz"), rgba=rgba[:n].numpy().view(np.uint32), merged=merged[:n].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_to_all_merge_reproduces_single_process_render(tmp_path, world):
    total = 600_000                         # 10 batches -> 5 + 5 or 4 + 3 + 3; 320x180 (+ padding) is not a multiple of 3 slices
    mp.spawn(_a2a_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    image, _ = P.synth_encode(total, scenes.SEED, nthreads=2)
    of = oracle.OracleFile(image.view())
    cam = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100)
    fb, _ = of.render_basic(cam)
    want_rgba = oracle.resolve_basic(cam, fb)
    big = np.iinfo(np.int64).max
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"a2a{rank}.npz"))
        merged = np.where(z["merged"] == big, -1, z["merged"]).view(np.uint64)
        assert np.array_equal(merged, fb), f"rank {rank}: merged framebuffer"
        assert np.array_equal(z["rgba"][:want_rgba.size], want_rgba.ravel()), f"rank {rank}: image"
