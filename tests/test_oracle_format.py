"""CPU tests of the container format, the native encoder and the oracle's lane-accurate decoder."""
import ctypes as C
import struct

import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import oracle, scenes


def sorted_source(total, seed=scenes.SEED, chunk=None):
    """Source points of the synthetic scene in the order the encoder emits them (pad, then Morton per chunk)."""
    x, y, z, c = P.synth_points(total, seed, 0, total)
    chunk = chunk or 6553600
    outs = []
    for a in range(0, total, chunk):
        xs, ys, zs = x[a:a + chunk], y[a:a + chunk], z[a:a + chunk]
        pad = (-len(xs)) % 65536
        xs = np.concatenate([xs, np.full(pad, xs[-1])]); ys = np.concatenate([ys, np.full(pad, ys[-1])])
        zs = np.concatenate([zs, np.full(pad, zs[-1])])
        keys = []
        for i in range(len(xs)):
            hi, lo = C.c_uint32(), C.c_uint64()
            P._native.host_lib().pcr_morton_key(int(xs[i]) + 2 ** 31, int(ys[i]) + 2 ** 31, int(zs[i]) + 2 ** 31, C.byref(hi), C.byref(lo))
            keys.append((hi.value, lo.value, i))
        order = np.array([k[2] for k in sorted(keys)])
        outs.append(np.stack([xs[order], ys[order], zs[order]], 1))
    return np.concatenate(outs)


def test_header_and_record_layout():
    image, st = scenes.synth_stream(200_000)
    f = P.HuffmanFile(image)
    assert (f.numPoints, f.numBatches) == (262144, 4) == (st["num_points"], st["num_batches"])
    assert f.clusterBytes == 128 * 4
    enc = sep = 0
    for b in range(4):
        blob = bytes(f.blob(b))
        hdr = struct.unpack_from("<5i", blob, 0)
        assert hdr[1:] == (65536, 1024, 64, 1)                       # BatchDumpData.h:63-77
        assert struct.unpack_from("<2i", blob, 116) == (4096, 32)
        ne, ns = f.stream_lengths(b)
        assert len(blob) == 124 + 4 * (3072 + 1024 + 4096 + 4096 + 32) + 4 * (ne + ns) + 32768   # BatchDumpData.h:148
        enc += 4 * ne; sep += 4 * ns
        lens = np.frombuffer(blob, np.int32, 4096, 124 + 4 * (3072 + 1024 + 4096))
        assert ((lens >= 1) & (lens <= 12) | (lens == -12)).all()
        cl = np.frombuffer(blob, np.int32, 32, 124 + 4 * (3072 + 1024 + 8192))
        assert (np.diff(cl) >= 64).all() and cl[0] >= 64            # every cluster holds at least words 0,1 of 32 chains
    assert (enc, sep) == (f.encodedBytes, f.separateBytes) == (st["encoded_bytes"], st["separate_bytes"])
    # sub-stream headers used for sharding
    h = f.header(1, 2)
    assert h.num_batches == 2 and h.num_points == 131072
    assert h.encoded_bytes == 4 * sum(f.stream_lengths(b)[0] for b in (1, 2))


def test_oracle_parse_mirrors_the_loader_layout():
    image, _ = scenes.synth_stream(200_000)
    f = P.HuffmanFile(image)
    of = oracle.OracleFile(image.view())
    e, s = of.encoded(), of.separate()
    eo = so = 0
    for b in range(f.numBatches):
        g = of.batch(b)
        assert (g.encoding_batch_offset, g.separate_batch_offset, g.decoder_table_offset, g.max_cw_len) == (eo, so, b * 4096, 12)
        he, hs = f.head_words(b)
        assert np.array_equal(e[eo:eo + len(he)], he) and np.array_equal(s[so:so + len(hs)], hs)
        ne, ns = f.stream_lengths(b)
        eo += ne; so += ns
    assert len(e) == eo + 1024 and not e[eo:].any()                  # HuffmanLasLoader.cpp:39-41 zero pad
    assert len(s) == so + 256 and not s[so:].any()


@pytest.mark.parametrize("total", [10_000, 200_000])
def test_decode_reproduces_source_except_reference_tail_artefact(total):
    """the reference's interleave queues two words too few per chain, so the last symbols of
    some chains decode to garbage; everything before in-chain position 32 must be exact, and most of the rest."""
    image, _ = scenes.synth_stream(total)
    of = oracle.OracleFile(image.view())
    src = sorted_source(total)
    bad_total = 0
    for b in range(of.num_batches):
        dec = of.decode_batch(b).reshape(65536, 3)
        bad = (dec != src[b * 65536:(b + 1) * 65536]).any(1)
        pos = np.nonzero(bad)[0] % 64
        assert pos.size == 0 or pos.min() >= 32, f"batch {b}: mismatch at in-chain position {pos.min()}"
        bad_total += int(bad.sum())
    assert bad_total < 0.02 * of.num_batches * 65536


@pytest.mark.parametrize("case", ["mostly_padding", "surface", "escape_heavy"])
def test_pad_tails_variant_decodes_every_point_exactly(case):
    """PCR_ENCODE_PAD_TAILS (pcr_encode.h; SURVEY 8f-1 'fixed variant'): with a zero word queued for every refill past a
    chain's last real word, the lane-accurate decode reproduces ALL source points; the default stream does not."""
    if case == "mostly_padding":           # low-entropy chains: the reference artefact starts 32-64 symbols early
        x, y, z, c = P.synth_points(10_000, scenes.SEED, 0, 10_000)
        las = P.synth_las_info(10_000)
    elif case == "surface":
        x, y, z, c = P.synth_points(200_000, scenes.SEED, 0, 200_000)
        las = P.synth_las_info(200_000)
    else:
        x, y, z, c, las = scenes.random_points(131072, seed=9)
    fixed, st_fixed = P.encode_points(x, y, z, c, las, morton_sort=False, nthreads=2, pad_tails=True)
    plain, st_plain = P.encode_points(x, y, z, c, las, morton_sort=False, nthreads=2)
    of, op = oracle.OracleFile(fixed.view()), oracle.OracleFile(plain.view())
    n = of.num_batches * 65536
    pad = n - len(x)
    src = np.stack([np.concatenate([v, np.full(pad, v[-1])]) for v in (x, y, z)], 1)
    wrong_plain = 0
    for b in range(of.num_batches):
        assert np.array_equal(of.decode_batch(b).reshape(65536, 3), src[b * 65536:(b + 1) * 65536]), f"batch {b}"
        wrong_plain += int((op.decode_batch(b).reshape(65536, 3) != src[b * 65536:(b + 1) * 65536]).any(1).sum())
    if case != "escape_heavy":            # (high-entropy chains end so late that the misfires fall after the last symbol)
        assert wrong_plain > 0                                          # the quirk the flag removes
    # only zero words were added: 1-2 per chain, same escapes, same tables
    extra = (st_fixed["encoded_bytes"] - st_plain["encoded_bytes"]) // 4
    assert of.num_batches * 1024 <= extra <= of.num_batches * 2048
    assert st_fixed["separate_bytes"] == st_plain["separate_bytes"] and st_fixed["escaped_symbols"] == st_plain["escaped_symbols"]


@pytest.mark.parametrize("case", ["mostly_padding", "surface", "escape_heavy"])
def test_lane_major_decode_equals_the_lockstep_decode(case):
    """The invariant behind k_transcode / k_render, on the CPU: write down the words every chain receives
    in the reference's lockstep walk, then decode each chain from its own sequence alone — same points, garbage tails
    included, for the full decode and for level-of-detail prefixes; a chain never receives more than 74 words."""
    if case == "mostly_padding":
        image, _ = scenes.synth_stream(10_000)
    elif case == "surface":
        image, _ = scenes.synth_stream(200_000)
    else:
        x, y, z, c, las = scenes.random_points(65536, seed=9)
        image, _ = P.encode_points(x, y, z, c, las, morton_sort=True, nthreads=2)
    of = oracle.OracleFile(image.view())
    b = of.num_batches - 1                                   # the last batch over-reads into the zero pad
    words, counts = of.lane_words(b)
    assert 2 <= counts.min() and counts.max() <= 74
    full = of.decode_batch(b, 64)
    part = of.decode_batch(b, 20)
    for chain in list(range(0, 1024, 37)) + [31, 32, 1023]:
        got = of.decode_chain_from_lane_words(b, chain, words, int(counts[chain]))
        assert np.array_equal(got, full[chain]), f"chain {chain}"
        got20 = of.decode_chain_from_lane_words(b, chain, words, int(counts[chain]), npr=20)
        assert np.array_equal(got20, part[chain, :20])


def test_lod_truncation_is_a_prefix_of_the_full_decode():
    image, _ = scenes.synth_stream(200_000)
    of = oracle.OracleFile(image.view())
    full = of.decode_batch(1, 64)
    part = of.decode_batch(1, 20)
    assert np.array_equal(part[:, :20], full[:, :20]) and not part[:, 20:].any()


def test_chunked_encoding_pads_and_sorts_per_chunk():
    total, chunk = 150_000, 65536 * 2
    image, st = P.synth_encode(total, scenes.SEED, chunk_points=chunk, nthreads=2)
    assert st["num_batches"] == 3 and st["num_points"] == 3 * 65536   # chunks of 131072 and 18928 -> 2 + 1 batches
    of = oracle.OracleFile(image.view())
    src = sorted_source(total, chunk=chunk)
    dec = of.decode_batch(2).reshape(65536, 3)
    ok = (dec == src[2 * 65536:3 * 65536]).all(1)
    assert ok[np.arange(65536) % 64 < 32].all()
    # the padding repeats the last point of the chunk (preprocess.cpp:945-955)
    x, y, z, _ = P.synth_points(total, scenes.SEED, total - 1, 1)
    assert ((src[2 * 65536:] == [x[0], y[0], z[0]]).all(1)).sum() >= 65536 * 3 - total


def test_degenerate_batch_of_identical_points():
    """The reference asserts on a single-symbol alphabet (huffman.h:265); the build encodes it with a 1-bit code."""
    n = 65536
    las = P.synth_las_info(1)
    image, st = P.encode_points(np.full(n, 7, np.int32), np.full(n, -3, np.int32), np.full(n, 11, np.int32),
                                np.full(n, 0x336699, np.uint32), las, morton_sort=True, nthreads=1)
    assert st["escaped_symbols"] == 0
    of = oracle.OracleFile(image.view())
    assert (of.decode_batch(0).reshape(-1, 3) == [7, -3, 11]).all()


def test_int32_wraparound_deltas_round_trip():
    rng = np.random.default_rng(11)
    n = 65536
    x = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    y = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    z = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    image, st = P.encode_points(x, y, z, np.zeros(n, np.uint32), P.synth_las_info(1), morton_sort=False, nthreads=1)
    of = oracle.OracleFile(image.view())
    dec = of.decode_batch(0).reshape(-1, 3)
    ok = (dec == np.stack([x, y, z], 1)).all(1)
    assert ok[np.arange(n) % 64 < 32].all()


def test_malformed_files_are_rejected():
    image, _ = scenes.synth_stream(10_000)
    good = bytes(image.view())
    for bad, msg in ((good[:20], "header"), (good[:-4], "exceeds|mismatch|short"),
                     (good[:8] + struct.pack("<q", 5) + good[16:], "numBatches|header")):
        with pytest.raises((ValueError, P.PcrError)):
            oracle.OracleFile(bad)
    geo = bytearray(good); geo[40 + 8 + 8:40 + 8 + 12] = struct.pack("<i", 512)
    with pytest.raises(ValueError, match="geometry"):
        oracle.OracleFile(bytes(geo))
    with pytest.raises(P.PcrError):
        P.HuffmanFile(good[:30])
    # a size table entry that is negative or smaller than a record's fixed part: refused when the file is opened
    for size in (-1, 100):
        with pytest.raises(P.PcrError, match="shorter than its fixed part"):
            P.HuffmanFile(good[:40] + struct.pack("<q", size) + good[48:])


def test_lod_and_cull_decisions():
    image, _ = scenes.synth_stream(2_000_000)
    of = oracle.OracleFile(image.view())
    cams = scenes.cameras(640, 360)
    far = scenes.with_flags(cams["far"], lod_percent=10)
    res = [of.batch_lod(b, far) for b in range(of.num_batches)]
    assert all(v and not d and 6 <= n < 64 for v, n, d in res)                 # float path, LOD above the 10 % floor
    full = [of.batch_lod(b, scenes.with_flags(cams["far"], lod_percent=100)) for b in range(of.num_batches)]
    assert all(n == 64 for _, n, _ in full)
    close = [of.batch_lod(b, cams["closeup"]) for b in range(of.num_batches)]
    assert any(not v for v, _, _ in close) and any(v and d for v, _, d in close)
    nocull = [of.batch_lod(b, scenes.with_flags(cams["closeup"], cull=0)) for b in range(of.num_batches)]
    assert all(v for v, _, _ in nocull)
    # the two kernels' LOD expressions (float vs double division by 100) may differ by a point but never in visibility
    for b in range(of.num_batches):
        a, h = of.batch_lod(b, far, oracle.MEM_ITER), of.batch_lod(b, far, oracle.HQS)
        assert a[0] == h[0] and abs(a[1] - h[1]) <= 1
