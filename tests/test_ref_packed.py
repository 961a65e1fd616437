"""A stream this repository's encoder did NOT write: tests/golden/ref_packed_batch.huffman holds two batches whose
dictionary, 4096-entry table, every chain's words / escapes / completion indices and colour blocks came out of the
reference's own library (include/huffman.h:94-113, 180-300; src/mymorton.h; src/rgbcx.cpp as src/preprocess.cpp:282-297
calls it), compiled unmodified under oracle/_ref by tools/make_golden.py::ref_packed_batch. Only the (time, lane)
interleave of src/preprocess.cpp:552-573 and the record layout were restated (those sources need GL/CUDA headers).

Two more files packed the same way (round 3): ref_packed_bc7.huffman, one batch with BC7 mode-6 colour blocks from the
reference's bc7enc (HQS method only), and ref_packed_lowentropy.huffman, 10 000 points padded to a batch by repeating the last
one -- SURVEY B.4's worst case: one-bit codes, so the tail artefact starts ten symbols before the end of a chain (position 54).

CPU leg: the oracle reproduces the committed hashes, and its lockstep decode differs from the source points only at
SURVEY B.4 tail positions. GPU leg (-m gpu): the HIP kernels, both layouts, draw the committed frames from them."""
import hashlib
import json
import os

import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, refpin

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    exp = json.load(open(os.path.join(GOLD, name + "_expected.json")))
    data = open(os.path.join(GOLD, name + ".huffman"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == exp["stream_sha256"]
    return data, exp


@pytest.fixture(scope="module")
def packed():
    return load("ref_packed_batch")


@pytest.fixture(scope="module", params=["ref_packed_bc7", "ref_packed_lowentropy"])
def packed2(request):
    return (request.param, *load(request.param))


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


def test_fixture_shape(packed):
    data, exp = packed
    hf = P.HuffmanFile(data)
    assert hf.numBatches == exp["batches"] == 2 and hf.numPoints == exp["padded_points"] == 131072
    # the reference's code assignment is not this repository's (libstdc++ hash order): the stream really is foreign
    assert not exp["own_encoder_on_same_points"]["encoded_words"]
    assert exp["own_encoder_on_same_points"]["colour_blocks_equal"] == 0
    # lockstep decode vs source: only the reference's tail artefact, at the very end of chains
    assert exp["lockstep_vs_source"]["wrong_points"] > 0 and exp["lockstep_vs_source"]["min_in_chain_position"] >= 54


def test_oracle_reproduces_the_committed_frames(packed):
    data, exp = packed
    of = oracle.OracleFile(data)
    for case in exp["cases"]:
        p = params_of(case, exp)
        fb, st = of.render_basic(p)
        assert st == case["stats_basic"] and sha(fb) == case["fb_basic_sha256"]
        assert of.count_depth_ties(p, fb) == (case["depth_tie_pixels"], case["depth_tie_pixels_other_colour"])
        assert sha(oracle.resolve_basic(p, fb)) == case["rgba_basic_sha256"]
        hfb, st2 = of.render_hqs_depth(p)
        assert st2 == case["stats_hqs"] and sha(hfb) == case["fb_hqs_sha256"]
        rg, ba, _ = of.render_hqs_color(p, hfb)
        assert sha(rg) == case["rg_sha256"] and sha(ba) == case["ba_sha256"]
        assert sha(oracle.resolve_hqs(p, hfb, rg, ba)) == case["rgba_hqs_sha256"]


def check_oracle(data, exp):
    of = oracle.OracleFile(data)
    for case in exp["cases"]:
        p = params_of(case, exp)
        if "fb_basic_sha256" in case:
            fb, st = of.render_basic(p)
            assert st == case["stats_basic"] and sha(fb) == case["fb_basic_sha256"]
            assert of.count_depth_ties(p, fb) == (case["depth_tie_pixels"], case["depth_tie_pixels_other_colour"])
            assert sha(oracle.resolve_basic(p, fb)) == case["rgba_basic_sha256"]
        hfb, st2 = of.render_hqs_depth(p)
        assert st2 == case["stats_hqs"] and sha(hfb) == case["fb_hqs_sha256"]
        rg, ba, _ = of.render_hqs_color(p, hfb)
        assert sha(rg) == case["rg_sha256"] and sha(ba) == case["ba_sha256"]
        assert sha(oracle.resolve_hqs(p, hfb, rg, ba)) == case["rgba_hqs_sha256"]


def test_second_fixtures_shape_and_oracle_frames(packed2):
    name, data, exp = packed2
    hf = P.HuffmanFile(data)
    assert hf.numBatches == exp["batches"] == 1 and hf.numPoints == exp["padded_points"] == 65536
    lv = exp["lockstep_vs_source"]
    if name == "ref_packed_bc7":
        assert exp["color_format"] == 7 and len(hf.blob(0)) - 124 - 4 * (3072 + 1024 + 4096 + 4096 + 32) - 4 * sum(hf.stream_lengths(0)) == 65536
        assert all("fb_basic_sha256" not in c for c in exp["cases"])          # the reference's basic method has no defined BC7 result
        assert lv["wrong_points"] > 0 and lv["min_in_chain_position"] == 63
    else:
        # the padding chains are one-bit codes: the artefact reaches back to position 54 and hits hundreds of points ("292 of
        # 65 536 points wrong, all at in-chain positions >= 54" for the survey's own 10 000-point file)
        assert exp["source_points"] == 10000 and exp["encoded_bits_per_point"] < 8.0
        assert lv["wrong_points"] > 200 and lv["min_in_chain_position"] == 54
    check_oracle(data, exp)
    of = oracle.OracleFile(data)
    words, counts = of.lane_words(0)
    full = of.decode_batch(0)
    for chain in (0, 31, 32, 160, 500, 1023):                               # the lane-major form decodes to the lockstep points, tails included
        assert np.array_equal(of.decode_chain_from_lane_words(0, chain, words, int(counts[chain])), full[chain])


def test_lane_major_form_of_the_foreign_stream(packed):
    """What k_transcode / k_render do, on the CPU: per-chain word sequences of the lockstep walk decode to the lockstep points."""
    data, _ = packed
    of = oracle.OracleFile(data)
    for b in range(of.num_batches):
        words, counts = of.lane_words(b)
        full = of.decode_batch(b)
        for chain in (0, 31, 32, 500, 1023):
            got = of.decode_chain_from_lane_words(b, chain, words, int(counts[chain]))
            assert np.array_equal(got, full[chain])


@pytest.mark.skipif(not refpin.available(), reason="oracle/_ref (reference build) not present")
def test_generator_reproduces_the_committed_stream(packed, tmp_path, monkeypatch):
    """Where the reference library is present the generator must write the committed bytes again (seeded inputs; the
    reference's code assignment depends on libstdc++ only)."""
    data, _ = packed
    import tools.make_golden as mg
    monkeypatch.setattr(mg, "G", str(tmp_path))
    mg.ref_packed_batch()
    assert open(tmp_path / "ref_packed_batch.huffman", "rb").read() == data
    for name, fn in (("ref_packed_bc7", mg.ref_packed_bc7), ("ref_packed_lowentropy", mg.ref_packed_lowentropy)):
        fn()
        assert open(tmp_path / (name + ".huffman"), "rb").read() == open(os.path.join(GOLD, name + ".huffman"), "rb").read()
        assert json.load(open(tmp_path / (name + "_expected.json"))) == json.load(open(os.path.join(GOLD, name + "_expected.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["point_windows", "words_only"])
def test_hip_draws_the_reference_packed_stream(packed, variant):
    data, exp = packed
    r = P.Renderer(exp["width"], exp["height"], device=0)
    try:
        if variant == "words_only":
            r.ctx.set_stream_layout(P.Context.LAYOUT_WORDS)
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


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["point_windows", "words_only"])
def test_hip_draws_the_second_reference_packed_streams(packed2, variant):
    name, data, exp = packed2
    r = P.Renderer(exp["width"], exp["height"], device=0)
    try:
        if variant == "words_only":
            r.ctx.set_stream_layout(P.Context.LAYOUT_WORDS)
        P.HuffmanLasData.create(data).load_all(r)
        ctx = r.ctx
        assert ctx.stream_color_format() == (7 if name == "ref_packed_bc7" else 1)
        for case in exp["cases"]:
            p = params_of(case, exp)
            if "fb_basic_sha256" in case:
                ctx.clear(); ctx.render_basic(p); ctx.resolve_basic(p)
                assert ctx.stats() == case["stats_basic"]
                assert sha(ctx.read_framebuffer(full=True)) == case["fb_basic_sha256"]
                assert sha(ctx.read_rgba()) == case["rgba_basic_sha256"]
            ctx.clear(); ctx.render_hqs_depth(p)
            assert ctx.stats() == case["stats_hqs"]
            assert sha(ctx.read_framebuffer(full=True)) == case["fb_hqs_sha256"]
            ctx.render_hqs_color(p); ctx.resolve_hqs(p)
            rg, ba = ctx.read_accum(full=True)
            assert sha(rg) == case["rg_sha256"] and sha(ba) == case["ba_sha256"]
            assert sha(ctx.read_rgba()) == case["rgba_hqs_sha256"]
    finally:
        r.ctx.close()
