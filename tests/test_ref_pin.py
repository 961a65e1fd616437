"""Pins the build's restatements against the REFERENCE's own code (oracle/_ref/libpcr_ref.so = include/huffman.h,
src/mymorton.h, src/rgbcx.cpp compiled unmodified) where that library exists, and against the committed vectors it
produced (tests/golden/*.npz) everywhere."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pcrhpg24_amd as P
from pcrhpg24_amd import _native as N
from tests import oracle, refpin

GOLD = os.path.join(os.path.dirname(__file__), "golden")
needs_ref = pytest.mark.skipif(not refpin.available(), reason="oracle/_ref/libpcr_ref.so not built (no reference checkout)")


# ------------------------------------------------------------------------------------------------------------
# helpers over libpcr_host.so building blocks
# ------------------------------------------------------------------------------------------------------------
def host_pack(chain, ds, dc, dl):
    lib = N.host_lib()
    chain = np.ascontiguousarray(chain, np.int32)
    ds = np.ascontiguousarray(ds, np.int32); dc = np.ascontiguousarray(dc, np.uint32); dl = np.ascontiguousarray(dl, np.int32)
    w, s, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nw, ns = C.c_int32(), C.c_int32()
    rc = lib.pcr_pack_chain(chain.ctypes.data, len(chain), ds.ctypes.data, dc.ctypes.data, dl.ctypes.data, len(ds),
                            C.byref(w), C.byref(nw), C.byref(s), C.byref(ns), C.byref(n))
    assert rc == 0, N.host_error()

    def take(p, k, dt):
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (max(k, 1),))[:k].copy().view(dt)
        lib.pcr_host_free(p)
        return a
    return take(w, nw.value, np.uint32), take(s, ns.value, np.int32), take(n, nw.value, np.int32)


def host_table(ds, dc, dl):
    ds = np.ascontiguousarray(ds, np.int32); dc = np.ascontiguousarray(dc, np.uint32); dl = np.ascontiguousarray(dl, np.int32)
    tv, tl = np.zeros(4096, np.int32), np.zeros(4096, np.int32)
    rc = N.host_lib().pcr_table_from_dict(ds.ctypes.data, dc.ctypes.data, dl.ctypes.data, len(ds), tv.ctypes.data, tl.ctypes.data)
    assert rc == 0, N.host_error()
    return tv, tl


def host_build(symbols, query):
    symbols = np.ascontiguousarray(symbols, np.int32); query = np.ascontiguousarray(query, np.int32)
    tv, tl = np.zeros(4096, np.int32), np.zeros(4096, np.int32)
    cw, ln = np.zeros(len(query), np.uint32), np.zeros(len(query), np.int32)
    rc = N.host_lib().pcr_huffman_build(symbols.ctypes.data, len(symbols), tv.ctypes.data, tl.ctypes.data,
                                        query.ctypes.data, len(query), cw.ctypes.data, ln.ctypes.data)
    assert rc == 0, N.host_error()
    return tv, tl, cw, ln


def escape_table_equal(tv_a, tl_a, tv_b, tl_b):
    """Tables agree on every length and on the value of every non-escape key (the symbol stored under an escape
    key is whichever deep leaf was written last — never read: SURVEY B.1)."""
    assert np.array_equal(tl_a, tl_b)
    m = tl_a > 0
    assert np.array_equal(tv_a[m], tv_b[m])


# ------------------------------------------------------------------------------------------------------------
# committed reference vectors (run everywhere)
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def huff_vectors():
    z = np.load(os.path.join(GOLD, "huffman_ref_vectors.npz"))
    cases = {}
    for k in z.files:
        c, f = k.split("/")
        cases.setdefault(c, {})[f] = z[k]
    return cases


@pytest.mark.parametrize("case", ["laplace", "wide", "skewed"])
def test_table_and_packing_match_reference_vectors(huff_vectors, case):
    v = huff_vectors[case]
    tv, tl = host_table(v["dict_symbols"], v["dict_cw"], v["dict_len"])          # huffman.h:220-240
    escape_table_equal(tv, tl, v["table_values"], v["table_lens"])
    for i in range(8):
        w, s, n = host_pack(v[f"chain{i}"], v["dict_symbols"], v["dict_cw"], v["dict_len"])   # huffman.h:242-300
        assert np.array_equal(w, v[f"words{i}"]) and np.array_equal(s, v[f"separate{i}"]) and np.array_equal(n, v[f"numcw{i}"])
        # the oracle's scalar chain decoder (huffman.h:433-477) inverts the reference's packing
        dec = oracle.decode_chain(v[f"words{i}"], v[f"separate{i}"], v["table_values"], v["table_lens"], 192)
        assert np.array_equal(dec, v[f"chain{i}"])


def test_bc1_decode_matches_reference_unpack():
    z = np.load(os.path.join(GOLD, "bc1_ref_blocks.npz"))
    blocks, unpacked = z["blocks"], z["unpacked"]
    flat = np.ascontiguousarray(blocks.reshape(-1))
    checked = 0
    for b in range(len(blocks)):
        c0 = int(blocks[b, 0]) | int(blocks[b, 1]) << 8
        c1 = int(blocks[b, 2]) | int(blocks[b, 3]) << 8
        if c0 <= c1:
            continue            # BC1 3-colour mode in rgbcx's unpacker; the kernels always decode 4-colour (render.cu:47-62)
        for i in range(16):
            assert oracle.decode_bc1(b * 16 + i, flat) == int(unpacked[b, i]) & 0xFFFFFF
        checked += 1
    assert checked > 100
    # the reference encoder is called with allow_3color = false (preprocess.cpp:294): solid blocks may have c0 == c1,
    # where selector 0/1 decode to the same colour in either mode
    for b in range(len(blocks)):
        c0 = int(blocks[b, 0]) | int(blocks[b, 1]) << 8
        c1 = int(blocks[b, 2]) | int(blocks[b, 3]) << 8
        assert c0 >= c1
        if c0 == c1:
            assert all(oracle.decode_bc1(b * 16 + i, flat) == int(unpacked[b, i]) & 0xFFFFFF for i in range(16))


# ------------------------------------------------------------------------------------------------------------
# live reference library (this container, and wherever oracle/_ref travelled)
# ------------------------------------------------------------------------------------------------------------
def xorshift_symbols(n, seed=1, mod=10000):
    """test_huffman.cpp draws rand() % 10000 (non-reproducible there); fixed-seed xorshift here."""
    out = np.empty(n, np.int32)
    x = seed
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF; x ^= x >> 17; x ^= (x << 5) & 0xFFFFFFFF
        out[i] = x % mod
    return out


@needs_ref
def test_reference_round_trip_executable_passes():
    exe = os.path.join(oracle.ORACLE_DIR, "_ref", "test_huffman")
    if not os.path.exists(exe):
        pytest.skip("test_huffman not built")
    r = subprocess.run([exe, "2000"], stdout=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("Are they equal???? : 1") == 11       # src/test_huffman.cpp:55-62


@needs_ref
@pytest.mark.parametrize("sorted_tree", [False, True])
def test_test_huffman_path_with_fixed_seed(sorted_tree):
    data = xorshift_symbols(10000)
    rc = refpin.RefCode(data, sorted_tree=sorted_tree)            # test_huffman.cpp:35-40 uses the sort-based tree
    ds, dc, dl = rc.dict()
    tv, tl = rc.table()
    w, s, n = rc.pack(data)
    assert np.array_equal(rc.unpack(w, s, len(data)), data)       # the reference's own assertion
    hw, hs, hn = host_pack(data, ds, dc, dl)
    assert np.array_equal(hw, w) and np.array_equal(hs, s) and np.array_equal(hn, n)
    escape_table_equal(*host_table(ds, dc, dl), tv, tl)
    assert np.array_equal(oracle.decode_chain(w, s, tv, tl, len(data)), data)


@needs_ref
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_own_code_is_a_valid_optimal_huffman_code(seed):
    rng = np.random.default_rng(seed)
    # small alphabet: every depth <= 12, so lengths are comparable with the reference tree's cost
    alphabet = np.unique(rng.integers(-300, 300, 24)).astype(np.int32)
    probs = rng.dirichlet(np.ones(len(alphabet)) * 4.0)
    data = rng.choice(alphabet, 30000, p=probs).astype(np.int32)
    syms, counts = np.unique(data, return_counts=True)
    tv, tl, cw, ln = host_build(data, syms)
    assert (ln > 0).all()
    assert abs(sum(2.0 ** -int(l) for l in ln) - 1.0) < 1e-12                    # complete prefix code (Kraft equality)
    rs, rcw, rl = refpin.RefCode(data).dict()
    ref_len = dict(zip(rs.tolist(), rl.tolist()))
    assert sum(int(c) * int(l) for c, l in zip(counts, ln)) == sum(int(c) * ref_len[int(s)] for s, c in zip(syms, counts))
    # and it decodes under the oracle
    w, s, n = host_pack(data[:192], syms, cw, ln)
    assert np.array_equal(oracle.decode_chain(w, s, tv, tl, 192), data[:192])


@needs_ref
def test_clipped_code_round_trips_through_reference_decoder():
    rng = np.random.default_rng(5)
    data = np.round(rng.laplace(0, 3000, 50000)).astype(np.int32)     # thousands of symbols -> deep leaves -> escapes
    syms = np.unique(data)
    tv, tl, cw, ln = host_build(data, syms)
    assert (ln == -12).any() and (ln > 0).any()
    w, s, n = host_pack(data[:192 * 4], syms, cw, ln)
    assert len(s) > 0
    assert np.array_equal(oracle.decode_chain(w, s, tv, tl, 192 * 4), data[:192 * 4])


@needs_ref
def test_morton_key_and_order_match_reference():
    rng = np.random.default_rng(3)
    ref, host = refpin.ref_lib(), N.host_lib()
    vals = np.concatenate([rng.integers(0, 2 ** 32, 3000, dtype=np.uint64),
                           np.array([0, 1, 2 ** 21 - 1, 2 ** 21, 2 ** 22, 2 ** 31, 2 ** 32 - 1], np.uint64)]).astype(np.uint32)
    xs, ys, zs = rng.permutation(vals)[:3000], rng.permutation(vals)[:3000], rng.permutation(vals)[:3000]
    for x, y, z in zip(xs.tolist(), ys.tolist(), zs.tolist()):
        h1, l1, h2, l2 = C.c_uint32(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        ref.ref_morton_key(x, y, z, C.byref(h1), C.byref(l1))
        host.pcr_morton_key(x, y, z, C.byref(h2), C.byref(l2))
        assert (h1.value, l1.value) == (h2.value, l2.value)
    # order: encode unsorted points with our encoder, compare the start of every chain with the reference's permutation
    n = 65536
    x = rng.integers(-50000, 50000, n).astype(np.int32); y = rng.integers(-50000, 50000, n).astype(np.int32)
    z = rng.integers(-500, 500, n).astype(np.int32)
    x[100:200] = x[100]; y[100:200] = y[100]; z[100:200] = z[100]          # ties: stable order matters
    order = np.zeros(n, np.uint32)
    ref.ref_morton_order(x.ctypes.data, y.ctypes.data, z.ctypes.data, n, order.ctypes.data)
    las = P.synth_las_info(1)
    image, _ = P.encode_points(x, y, z, np.arange(n, dtype=np.uint32) & 0xFFFFFF, las, morton_sort=True, nthreads=1)
    of = oracle.OracleFile(image.view())
    start = np.ctypeslib.as_array(C.cast(of.s.start_values, C.POINTER(C.c_int32)), (1024 * 3,)).reshape(1024, 3)
    exp = np.stack([x[order], y[order], z[order]], 1)[::64]
    assert np.array_equal(start, exp)


@needs_ref
def test_live_reference_matches_committed_vectors(huff_vectors):
    """The committed .npz really is what the reference code produces (guards against a stale fixture)."""
    v = huff_vectors["laplace"]
    rc = refpin.RefCode(v["batch"])
    tv, tl = rc.table()
    escape_table_equal(tv, tl, v["table_values"], v["table_lens"])
    w, s, n = rc.pack(v["chain0"])
    assert np.array_equal(w, v["words0"]) and np.array_equal(s, v["separate0"]) and np.array_equal(n, v["numcw0"])


def test_bc1_encoder_is_within_half_a_db_of_rgbcx():
    """the native encoder's colour blocks against the reference's rgbcx::encode_bc1(level 8) blocks on the
    committed colour set, both decoded as the kernel decodes (tools/bc1_quality.py): PSNR gap <= 0.5 dB overall and per kind."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bc1_quality as q
    d = np.load(os.path.join(GOLD, "bc1_ref_blocks.npz"))
    lib = N.host_lib()
    own = np.zeros_like(d["blocks"])
    for k in range(len(d["colors"])):
        c = np.ascontiguousarray(d["colors"][k], np.uint32)
        lib.pcr_bc1_encode_block(c.ctypes.data_as(C.c_void_p), own[k].ctypes.data_as(C.c_void_p))
        c0 = int(own[k][0]) | int(own[k][1]) << 8; c1 = int(own[k][2]) | int(own[k][3]) << 8
        assert c0 >= c1                                   # 4-colour mode (equal endpoints: every palette entry the same)
    a, ak = q.psnr(d["colors"], own)
    b, bk = q.psnr(d["colors"], d["blocks"])
    assert b - a <= 0.5, (a, b)
    assert all(y - x <= 0.5 for x, y in zip(ak, bk)), (ak, bk)
