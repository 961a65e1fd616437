"""include/pcr_dist.h — the multi-GPU layer in C++ over RCCL (libpcr_dist.so) and its headless driver pcr_render_dist.
CPU leg: the library loads, exports what the header declares, and splits batches like the Python transport layer does.
GPU leg (one-GPU box: world size 1 is all that can run here): a one-rank communicator's reduce / all-reduce forms leave the
oracle's frame in place, basic and HQS, and pcr_render_dist prints pcr_render's framebuffer hash."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import pcrhpg24_amd as P
from pcrhpg24_amd import _native as N
from pcrhpg24_amd import build
from pcrhpg24_amd import dist as pdist
from tests import oracle, scenes

INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+(pcr_dist_[a-z0-9_]+)\s*\(", re.M)


def dist_lib():
    # torch first: one HIP runtime (and one RCCL) per process, see _native.hip_lib
    N.hip_lib()
    lib = C.CDLL(build.build_dist())
    lib.pcr_dist_last_error.restype = C.c_char_p
    lib.pcr_dist_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pcr_dist_shard_range.restype = None
    lib.pcr_dist_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.pcr_dist_destroy.argtypes = [C.c_void_p]
    lib.pcr_dist_destroy.restype = None
    for n in ("pcr_dist_merge_min", "pcr_dist_merge_sum"):
        getattr(lib, n).argtypes = [C.c_void_p, C.c_int]
    for n in ("pcr_dist_frame_basic", "pcr_dist_frame_hqs", "pcr_dist_step_basic"):
        getattr(lib, n).argtypes = [C.c_void_p, C.POINTER(P.RenderParams), C.c_int]
    lib.pcr_dist_slice_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.pcr_dist_slice_range.restype = None
    lib.pcr_dist_set_exchange.argtypes = [C.c_void_p, C.c_int]
    lib.pcr_dist_exchange.argtypes = [C.c_void_p]
    return lib


def test_library_exports_every_declared_symbol():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, "pcr_dist.h")).read(), flags=re.S)
    names = set(DECL.findall(text))
    assert {"pcr_dist_create", "pcr_dist_create_local", "pcr_dist_merge_min", "pcr_dist_merge_sum", "pcr_dist_frame_hqs"} <= names
    lib = dist_lib()
    for name in sorted(names):
        assert hasattr(lib, name), f"{name} is declared in include/pcr_dist.h but not exported"


@pytest.mark.parametrize("units,world", [(1526, 8), (30518, 8), (7, 3), (5, 5), (16, 1), (3, 4)])
def test_shard_range_is_the_transport_layers_split(units, world):
    lib = dist_lib()
    covered = 0
    for rank in range(world):
        a, b = C.c_int64(), C.c_int64()
        lib.pcr_dist_shard_range(units, world, rank, C.byref(a), C.byref(b))
        assert (a.value, b.value) == pdist.shard_range(units, world, rank)
        assert a.value == covered
        covered += b.value
    assert covered == units


@pytest.mark.parametrize("elems,world", [(1920 * 1081 + 1, 8), (4096 * 4097 + 1, 8), (640 * 361 + 1, 3), (1001, 1), (10, 7), (16_781_313, 128)])
def test_slice_range_is_the_transport_layers_slicing(elems, world):
    """Equal, 16-byte aligned slices that cover the frame and stay inside its pad (include/pcr_types.h: PCR_FRAME_PAD_ELEMS)."""
    lib = dist_lib()
    S = pdist.slice_elems(elems, world)
    for rank in range(world):
        a, b = C.c_size_t(), C.c_size_t()
        lib.pcr_dist_slice_range(elems, world, rank, C.byref(a), C.byref(b))
        assert (a.value, b.value) == (rank * S, S)
    assert S % 2 == 0 and elems <= world * S <= elems + 256


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [2, 3])                 # PCR_DIST_EXCHANGE_SLICED, _SLICED_P2P
def test_one_rank_communicator_sliced_frames_match_the_oracle(exchange):
    """The sliced exchange on the one rank a test box has: reduce-scatter (or all-to-all + local min) over one slice that
    reaches into the frame's pad, resolve of the own slice, gather of the image; basic (frame and steady-state step) and HQS."""
    lib = dist_lib()
    image, _ = scenes.synth_stream(2_000_000)
    of = oracle.OracleFile(image.view())
    W, H = 640, 360                                            # 640 * 361 + 1 words: odd, the slice is one word longer
    p = scenes.with_flags(scenes.cameras(W, H)["closeup"], lod_percent=100, cull=1)
    r = P.Renderer(W, H, device=0)
    d = C.c_void_p()
    try:
        P.HuffmanLasData.create(image).load_all(r)
        ident = C.create_string_buffer(128)
        assert lib.pcr_dist_unique_id(ident) == 0, lib.pcr_dist_last_error()
        assert lib.pcr_dist_create(r.ctx.h, ident, 0, 1, C.byref(d)) == 0, lib.pcr_dist_last_error()
        assert lib.pcr_dist_exchange(d) == 1                   # AUTO on one rank: the whole-frame reduce
        assert lib.pcr_dist_set_exchange(d, exchange) == 0 and lib.pcr_dist_exchange(d) == exchange
        ofb, ost = of.render_basic(p)
        want = oracle.resolve_basic(p, ofb)
        for root in (0, -1):
            assert lib.pcr_dist_frame_basic(d, C.byref(p), root) == 0, lib.pcr_dist_last_error()
            assert np.array_equal(r.ctx.read_rgba(), want) and r.ctx.stats() == ost
        r.ctx.frame_begin(p)
        for _ in range(2):
            assert lib.pcr_dist_step_basic(d, C.byref(p), 0) == 0, lib.pcr_dist_last_error()
            assert np.array_equal(r.ctx.read_rgba(), want) and r.ctx.stats() == ost
        hfb, _ = of.render_hqs_depth(p)
        org, oba, _ = of.render_hqs_color(p, hfb)
        for root in (0, -1):
            assert lib.pcr_dist_frame_hqs(d, C.byref(p), root) == 0, lib.pcr_dist_last_error()
            assert np.array_equal(r.ctx.read_framebuffer(full=True), hfb)
            assert np.array_equal(r.ctx.read_rgba(), oracle.resolve_hqs(p, hfb, org, oba))
        assert lib.pcr_dist_set_exchange(d, 9) != 0
    finally:
        if d.value:
            lib.pcr_dist_destroy(d)
        r.ctx.close()


@pytest.mark.gpu
def test_one_rank_communicator_frames_match_the_oracle():
    lib = dist_lib()
    image, _ = scenes.synth_stream(2_000_000)
    of = oracle.OracleFile(image.view())
    W, H = 640, 360
    p = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100, cull=0)
    r = P.Renderer(W, H, device=0)
    d = C.c_void_p()
    try:
        P.HuffmanLasData.create(image).load_all(r)
        ident = C.create_string_buffer(128)
        assert lib.pcr_dist_unique_id(ident) == 0, lib.pcr_dist_last_error()
        assert lib.pcr_dist_create(r.ctx.h, ident, 0, 1, C.byref(d)) == 0, lib.pcr_dist_last_error()
        assert lib.pcr_dist_world(d) == 1 and lib.pcr_dist_rank(d) == 0
        lib.pcr_dist_comm_ranks.argtypes = [C.c_void_p]
        assert lib.pcr_dist_comm_ranks(d) == 1                  # ncclCommCount: what bench.py quotes as rccl_ranks
        ofb, ost = of.render_basic(p)
        for root in (0, -1):                              # reduce to rank 0, all-reduce
            assert lib.pcr_dist_frame_basic(d, C.byref(p), root) == 0, lib.pcr_dist_last_error()
            assert np.array_equal(r.ctx.read_framebuffer(full=True), ofb) and r.ctx.stats() == ost
            assert np.array_equal(r.ctx.read_rgba(), oracle.resolve_basic(p, ofb))
        # steady-state form: render + merge + one launch for resolve / clear / next prepass
        lib.pcr_dist_step_basic.argtypes = [C.c_void_p, C.POINTER(P.RenderParams), C.c_int]
        r.ctx.frame_begin(p)
        for _ in range(2):
            assert lib.pcr_dist_step_basic(d, C.byref(p), 0) == 0, lib.pcr_dist_last_error()
            assert np.array_equal(r.ctx.read_rgba(), oracle.resolve_basic(p, ofb)) and r.ctx.stats() == ost
        r.ctx.clear()
        hfb, _ = of.render_hqs_depth(p)
        org, oba, _ = of.render_hqs_color(p, hfb)
        assert lib.pcr_dist_frame_hqs(d, C.byref(p), 0) == 0, lib.pcr_dist_last_error()
        assert np.array_equal(r.ctx.read_framebuffer(full=True), hfb)
        rg, ba = r.ctx.read_accum(full=True)
        assert np.array_equal(rg, org) and np.array_equal(ba, oba)
        assert np.array_equal(r.ctx.read_rgba(), oracle.resolve_hqs(p, hfb, org, oba))
        assert lib.pcr_dist_merge_min(d, 5) != 0 and b"root" in lib.pcr_dist_last_error()
    finally:
        if d.value:
            lib.pcr_dist_destroy(d)
        r.ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["huffman_mem_iter_cuda", "huffman_hqs"])
def test_render_dist_cli_prints_the_single_gpu_hash(tmp_path, method):
    build.build_all()
    image, _ = scenes.synth_stream(2_000_000)
    path = tmp_path / "scene.huffman"
    path.write_bytes(bytes(image.view()))
    common = [str(path), "--method", method, "--size", "640x360", "--camera", "-0.15", "-0.57", "1500", "500", "500", "40", "--lod", "0.1"]
    one = subprocess.run([build.RENDER_BIN, *common], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert one.returncode == 0, one.stderr
    many = subprocess.run([build.RENDER_DIST_BIN, *common, "--ranks", "1", "--frames", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert many.returncode == 0, many.stderr
    a, b = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(many.stdout.strip().splitlines()[-1])
    assert b["ranks"] == 1 and b["batches"] == a["batches"] == 31
    same = ("fb_fnv1a", "rgba_fnv1a", "points_iterated", "covered_pixels")
    assert all(a[k] == b[k] for k in same), (a, b)
    for merge in ("sliced", "sliced_p2p"):                      # the image assembled from the slices is the one-GPU image
        sl = subprocess.run([build.RENDER_DIST_BIN, *common, "--ranks", "1", "--frames", "3", "--merge", merge], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert sl.returncode == 0, sl.stderr
        c = json.loads(sl.stdout.strip().splitlines()[-1])
        assert c["rgba_fnv1a"] == a["rgba_fnv1a"] and c["points_iterated"] == a["points_iterated"] and c["covered_pixels"] == a["covered_pixels"], (merge, a, c)
    import torch
    ngpu = torch.cuda.device_count()
    for extra in ([], ["--allreduce"], ["--merge", "sliced"], ["--merge", "sliced_p2p"], ["--merge", "sliced", "--allreduce"]):
        two = subprocess.run([build.RENDER_DIST_BIN, *common, "--ranks", "2", "--frames", "3", *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        if ngpu < 2:
            # two ranks cannot share this box's one GPU: the library says so instead of hanging in RCCL
            assert two.returncode != 0 and "pcr_create" in two.stderr
            break
        # a box with peers: every exchange form merges the two shards into the one-GPU frame, bit for bit
        assert two.returncode == 0, (extra, two.stderr)
        c = json.loads(two.stdout.strip().splitlines()[-1])
        assert c["ranks"] == 2 and c["rgba_fnv1a"] == a["rgba_fnv1a"], (extra, a, c)
        assert c["points_iterated"] == a["points_iterated"] and c["covered_pixels"] == a["covered_pixels"], (extra, a, c)
        if "--merge" not in extra:
            assert c["fb_fnv1a"] == a["fb_fnv1a"], (extra, a, c)
