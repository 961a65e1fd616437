"""BASELINE.json configs[0]: the 10 000-point synthetic batch, decoded and rasterized to 256x256.
The committed fixture pins the oracle (CPU, every run) and the HIP path (-m gpu) to the same bytes."""
import hashlib
import json
import os

import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    exp = json.load(open(os.path.join(GOLD, "config1_expected.json")))
    data = open(os.path.join(GOLD, "config1.huffman"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == exp["stream_sha256"]
    return data, exp


def params_of(case, exp):
    p = P.RenderParams()
    for k in ("transform", "world_view", "proj"):
        for i, v in enumerate(case["params"][k]):
            getattr(p, k)[i] = v
    p.width, p.height = exp["width"], exp["height"]
    p.points_per_thread = 64
    p.lod_percent = case["lod_percent"]
    p.enable_frustum_culling = case["params"]["enable_frustum_culling"]
    return p


def test_encoder_reproduces_the_committed_stream(golden):
    data, exp = golden
    image, st = P.synth_encode(10_000, 0x5EED, nthreads=1)
    assert bytes(image.view()) == data            # deterministic encoder: no hash-order / libstdc++ dependence
    assert st["num_points"] == 65536 and st["num_batches"] == 1 and st["num_points_in"] == 10_000


def test_camera_helper_reproduces_the_committed_matrices(golden):
    _, exp = golden
    p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), exp["width"], exp["height"])
    c = exp["cases"][0]
    assert c["camera"] == "overview"
    assert list(p.transform) == c["params"]["transform"] and list(p.proj) == c["params"]["proj"]


def test_oracle_matches_golden(golden):
    data, exp = golden
    of = oracle.OracleFile(data)
    for case in exp["cases"]:
        p = params_of(case, exp)
        fb, st = of.render_basic(p)
        assert st == case["stats_basic"]
        assert sha(fb) == case["fb_basic_sha256"]
        assert sha(oracle.resolve_basic(p, fb)) == case["rgba_basic_sha256"]
        hfb, st2 = of.render_hqs_depth(p)
        assert st2 == case["stats_hqs"] and sha(hfb) == case["fb_hqs_sha256"]
        rg, ba, _ = of.render_hqs_color(p, hfb)
        assert sha(rg) == case["rg_sha256"] and sha(ba) == case["ba_sha256"]
        assert sha(oracle.resolve_hqs(p, hfb, rg, ba)) == case["rgba_hqs_sha256"]
        assert int((fb[:p.width * p.height] != 0xFFFFFFFFFFFFFFFF).sum()) == case["covered_pixels"]


def test_oracle_multithreaded_render_equals_single_thread(golden):
    data, exp = golden
    of = oracle.OracleFile(data)
    p = params_of(exp["cases"][0], exp)
    a, sa = of.render_basic(p)
    b, sb = of.render_basic(p, nthreads=4)
    assert np.array_equal(a, b) and sa == sb


@pytest.mark.gpu
def test_hip_matches_golden(golden):
    data, exp = golden
    r = P.Renderer(exp["width"], exp["height"], device=0)
    try:
        P.HuffmanLasData.create(data).load_all(r)
        ctx = r.ctx
        for case in exp["cases"]:
            p = params_of(case, exp)
            ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)
            assert ctx.stats() == case["stats_basic"]
            assert sha(ctx.read_framebuffer(full=True)) == case["fb_basic_sha256"]
            assert sha(ctx.read_rgba()) == case["rgba_basic_sha256"]
            ctx.clear(); ctx.render_hqs_depth(p)
            assert sha(ctx.read_framebuffer(full=True)) == case["fb_hqs_sha256"]
            ctx.render_hqs_color(p); ctx.resolve_hqs(p)
            rg, ba = ctx.read_accum(full=True)
            assert sha(rg) == case["rg_sha256"] and sha(ba) == case["ba_sha256"]
            assert sha(ctx.read_rgba()) == case["rgba_hqs_sha256"]
    finally:
        r.ctx.close()
