#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where the reference
checkout and oracle/_ref exist). Fixtures are DATA: inputs + expected outputs; nothing of the reference's
source text is stored.

  config1.huffman            BASELINE.json configs[0]: 10 000 synthetic points -> one 65 536-point batch
                             (padding by repeating the last point, src/preprocess.cpp:945-955), our encoder
  config1_expected.json      cameras (as float lists) + SHA-256 of the oracle's 256x256 u64 framebuffer,
                             HQS RG/BA sums and RGBA8 resolves
  bc1_ref_blocks.npz         seeded colours, the BC1 blocks the REFERENCE encoder (src/rgbcx.cpp, called as
                             src/preprocess.cpp:282-297 does) produced for them, and rgbcx's own unpack of them
  huffman_ref_vectors.npz    seeded symbol chains, the REFERENCE Huffman<int32_t> dictionary/table
                             (include/huffman.h:94-113,180-240) and its packed words / escapes / per-word
                             completion indices (huffman.h:242-300) for them
  ref_packed_batch.huffman   two full batches (131 072 points) whose every code bit came out of the REFERENCE's own
                             library: Morton order from src/mymorton.h, dictionary + 4096-entry table from
                             Huffman<int32_t>::{calculate_frequencies, generate_huffman_tree_priority_queue,
                             create_dictionary_pjn, get_gpu_huffman_table_pjn} fed as Batch::calculate feeds them
                             (src/preprocess.cpp:757-770), every chain's words / escapes / completion indices from
                             compress_udtype_subarray_fast_pjn_idea (include/huffman.h:242-300), colour blocks from
                             rgbcx::encode_bc1(level 8) as src/preprocess.cpp:282-297 calls it. Only the (time, lane)
                             interleave of src/preprocess.cpp:552-573 and the record layout of
                             include/BatchDumpData.h:151-202 are restated here (those sources need GL/CUDA headers).
  ref_packed_bc7.huffman     one batch packed the same way with BC7 mode-6 colour blocks from the reference's bc7enc (HQS method only)
  ref_packed_lowentropy.huffman   10 000 points padded to one batch by repeating the last point: SURVEY B.4's worst case (one-bit
                             codes; the reference's tail artefact starts 32-64 symbols before the chain ends)
  ref_packed_batch_expected.json   cameras, SHA-256 of the oracle's framebuffers / sums / resolves for that file, the
                             in-chain positions at which the lockstep decode differs from the reference's own scalar
                             decoder (huffman.h:433-477; must all be SURVEY B.4 tail positions) and the depth-tie counts
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pcrhpg24_amd as P           # noqa: E402
from tests import oracle, refpin   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def config1():
    image, st = P.synth_encode(10_000, 0x5EED, nthreads=1)
    data = bytes(image.view())
    open(os.path.join(G, "config1.huffman"), "wb").write(data)
    of = oracle.OracleFile(data)
    W = H = 256
    cams = {
        "overview": P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), W, H),
        "closeup": P.camera_orbit(0.8, -0.45, 120.0, (500.0, 500.0, 45.0), W, H),
    }
    out = {"stream_sha256": hashlib.sha256(data).hexdigest(), "encode_stats": st, "width": W, "height": H, "cases": []}
    for name, p in cams.items():
        for lod in (10, 100):
            p.lod_percent = lod
            fb, s1 = of.render_basic(p)
            rgba = oracle.resolve_basic(p, fb)
            hfb, s2 = of.render_hqs_depth(p)
            rg, ba, _ = of.render_hqs_color(p, hfb)
            hrgba = oracle.resolve_hqs(p, hfb, rg, ba)
            out["cases"].append({
                "camera": name, "lod_percent": lod,
                "params": {"transform": list(p.transform), "world_view": list(p.world_view), "proj": list(p.proj),
                           "enable_frustum_culling": p.enable_frustum_culling},
                "stats_basic": s1, "stats_hqs": s2,
                "covered_pixels": int((fb[:W * H] != 0xFFFFFFFFFFFFFFFF).sum()),
                "fb_basic_sha256": sha(fb), "rgba_basic_sha256": sha(rgba),
                "fb_hqs_sha256": sha(hfb), "rg_sha256": sha(rg), "ba_sha256": sha(ba), "rgba_hqs_sha256": sha(hrgba),
            })
    json.dump(out, open(os.path.join(G, "config1_expected.json"), "w"), indent=1)
    print("config1:", len(data), "bytes,", [c["covered_pixels"] for c in out["cases"]])


def bc1_ref():
    ref = refpin.ref_lib()
    rng = np.random.default_rng(1234)
    blocks = []
    # smooth gradients, noisy blocks, solid colours, two-colour blocks
    for k in range(256):
        base = rng.integers(0, 256, 3)
        if k % 4 == 0:
            cols = np.tile(base, (16, 1))
        elif k % 4 == 1:
            cols = np.clip(base + rng.integers(-12, 13, (16, 3)), 0, 255)
        elif k % 4 == 2:
            other = rng.integers(0, 256, 3)
            t = rng.random((16, 1))
            cols = (base * (1 - t) + other * t).astype(np.int64)
        else:
            cols = rng.integers(0, 256, (16, 3))
        blocks.append((cols[:, 0] | (cols[:, 1] << 8) | (cols[:, 2] << 16)).astype(np.uint32))
    colors = np.stack(blocks)
    enc = np.zeros((len(colors), 8), np.uint8)
    unp = np.zeros((len(colors), 16), np.uint32)
    for i in range(len(colors)):
        ref.ref_bc1_encode(colors[i].ctypes.data, enc[i].ctypes.data)
        ref.ref_bc1_unpack(enc[i].ctypes.data, unp[i].ctypes.data)
    np.savez_compressed(os.path.join(G, "bc1_ref_blocks.npz"), colors=colors, blocks=enc, unpacked=unp)
    print("bc1:", enc.shape)


def bc7_ref():
    """Seeded colours, the BC7 mode-6 blocks the REFERENCE encoder makes of them (src/bc7enc.cpp, called as Chain::encode_color_bc7
    does, src/preprocess.cpp:299-316) and what the reference's CPU decoder (src/bc7decomp.cpp) unpacks them to."""
    ref = refpin.ref_lib()
    rng = np.random.default_rng(4321)
    blocks = []
    for k in range(256):
        base = rng.integers(0, 256, 3)
        if k % 4 == 0:
            cols = np.tile(base, (16, 1))
        elif k % 4 == 1:
            cols = np.clip(base + rng.integers(-12, 13, (16, 3)), 0, 255)
        elif k % 4 == 2:
            other = rng.integers(0, 256, 3)
            t = rng.random((16, 1))
            cols = (base * (1 - t) + other * t).astype(np.int64)
        else:
            cols = rng.integers(0, 256, (16, 3))
        blocks.append((cols[:, 0] | (cols[:, 1] << 8) | (cols[:, 2] << 16) | (255 << 24)).astype(np.uint32))
    colors = np.stack(blocks)
    enc = np.zeros((len(colors), 16), np.uint8)
    unp = np.zeros((len(colors), 16), np.uint32)
    for i in range(len(colors)):
        ref.ref_bc7_encode(colors[i].ctypes.data, enc[i].ctypes.data)
        assert ref.ref_bc7_unpack(enc[i].ctypes.data, unp[i].ctypes.data) == 1
    np.savez_compressed(os.path.join(G, "bc7_ref_blocks.npz"), colors=colors, blocks=enc, unpacked=unp)
    print("bc7:", enc.shape, "modes", sorted(set(int(b[0]) & 0x7F for b in enc)))


def huffman_ref():
    rng = np.random.default_rng(99)
    cases = {}
    for name, gen in (("laplace", lambda n: np.round(rng.laplace(0, 40, n)).astype(np.int32)),
                      ("wide", lambda n: (rng.integers(-5000, 5000, n)).astype(np.int32)),
                      ("skewed", lambda n: rng.choice(np.array([0, 1, -1, 100, -100, 7, 123456, -2_000_000_000], np.int32), n,
                                                     p=[.5, .2, .1, .08, .06, .03, .02, .01]).astype(np.int32))):
        batch = gen(20000)
        rc = refpin.RefCode(batch)
        chains = [batch[i * 192:(i + 1) * 192].copy() for i in range(8)]
        packed = [rc.pack(c) for c in chains]
        ds, dc, dl = rc.dict()
        tv, tl = rc.table()
        cases[name] = dict(batch=batch, dict_symbols=ds, dict_cw=dc, dict_len=dl, table_values=tv, table_lens=tl)
        for i, (w, s, n) in enumerate(packed):
            cases[name][f"chain{i}"] = chains[i]
            cases[name][f"words{i}"] = w
            cases[name][f"separate{i}"] = s
            cases[name][f"numcw{i}"] = n
    flat = {f"{k}/{kk}": v for k, d in cases.items() for kk, v in d.items()}
    np.savez_compressed(os.path.join(G, "huffman_ref_vectors.npz"), **flat)
    print("huffman:", {k: len(d["dict_symbols"]) for k, d in cases.items()})


def assemble_huffman_file(records) -> bytes:
    """5 x int64 header, batch size table, batch records (include/BatchDumpData.h:151-202 field
    order; src/preprocess.cpp:1205-1234 for the header)."""
    import struct
    blobs, enc_b, sep_b, clu_b, npts = [], 0, 0, 0, 0
    for r in records:
        fixed = struct.pack("<5i", r["point_offset"], 65536, 1024, 64, 1)
        fixed += struct.pack("<6d", *r["las_scale"], *r["las_offset"])
        fixed += struct.pack("<12f", *r["bbox_min"], *r["bbox_max"], *r["las_min"], *r["las_max"])
        fixed += struct.pack("<2i", 4096, 32)
        assert len(fixed) == 124
        body = b"".join(np.ascontiguousarray(r[k], dt).tobytes() for k, dt in (
            ("start_values", np.int32), ("separate_sizes", np.int32), ("decoder_values", np.int32), ("decoder_cw_len", np.int32),
            ("cluster_sizes", np.int32), ("encoding", np.uint32), ("separate", np.int32), ("color", np.uint8)))
        blobs.append(fixed + body)
        enc_b += 4 * len(r["encoding"]); sep_b += 4 * len(r["separate"]); clu_b += 128; npts += 65536
    head = struct.pack("<5q", npts, len(blobs), enc_b, sep_b, clu_b) + np.array([len(b) for b in blobs], np.int64).tobytes()
    return head + b"".join(blobs)


def pack_with_the_reference_library(px, py, pz, col, scale, offset, bc7=False):
    """points -> (.huffman bytes, records) with every code bit from the reference's own library (oracle/_ref): Morton order
    (src/mymorton.h), dictionary + table + per-chain packing (include/huffman.h), colour blocks (rgbcx BC1, or bc7enc mode 6
    as Chain::encode_color_bc7 calls it, src/preprocess.cpp:299-316). Restated: padding (src/preprocess.cpp:945-955), the
    (time, lane) interleave (:552-573) and the record layout (include/BatchDumpData.h:151-202)."""
    ref = refpin.ref_lib()
    las_min = tuple(float(a.min()) * s_ for a, s_ in zip((px, py, pz), scale))
    las_max = tuple(float(a.max()) * s_ for a, s_ in zip((px, py, pz), scale))
    # src/preprocess.cpp:945-955: pad to a multiple of 65 536 by repeating the last point
    pad = (-len(px)) % 65536
    px, py, pz, col = (np.concatenate([a, np.full(pad, a[-1], a.dtype)]) for a in (px, py, pz, col))
    # src/preprocess.cpp:959-977 with the reference's own src/mymorton.h:39-58
    order = np.zeros(len(px), np.uint32)
    ref.ref_morton_order(px.ctypes.data, py.ctypes.data, pz.ctypes.data, len(px), order.ctypes.data)
    px, py, pz, col = px[order], py[order], pz[order], col[order]
    records = []
    for b in range(len(px) // 65536):
        sl = slice(b * 65536, (b + 1) * 65536)
        X, Y, Z, Cc = (a[sl].reshape(1024, 64) for a in (px, py, pz, col))
        # Chain::calculate_deltas + interleave (src/preprocess.cpp:318-343); int32 wrap-around as the C++ does
        d = np.zeros((1024, 64, 3), np.int32)
        for k, A in enumerate((X, Y, Z)):
            d[:, 1:, k] = (A[:, 1:].astype(np.int64) - A[:, :-1].astype(np.int64)).astype(np.int32)
        chains = d.reshape(1024, 192)
        rc = refpin.RefCode(chains.reshape(-1))                      # Batch::calculate: all_deltas in chain order (:757-770)
        tv, tl = rc.table()
        packed = [rc.pack(chains[c]) for c in range(1024)]           # Chain::encode (:345-365)
        # the reference's own scalar decoder on each chain's own words must give the chain back (its ASSERT_DECOMPRESSION path)
        for c in (0, 1, 511, 1023):
            w, sp, _ = packed[c]
            assert np.array_equal(rc.unpack(w, sp, 192), chains[c])
        # Batch::encode_decode_bernhard (:540-587), restated: (time, lane) keys, stable sort, concatenate
        enc, cluster_sizes = [], []
        for wid in range(32):
            pairs = []
            for lane in range(32):
                pairs.append(((-1, lane), 0)); pairs.append(((0, lane), 1))
            for lane in range(32):
                step = packed[wid * 32 + lane][2]
                for i in range(2, len(step)):
                    pairs.append(((int(step[i - 2]), lane), i))
            pairs.sort()
            for (_, lane), i in pairs:
                enc.append(packed[wid * 32 + lane][0][i])
            cluster_sizes.append(len(enc))
        sep = np.concatenate([packed[c][1] for c in range(1024)]) if any(len(packed[c][1]) for c in range(1024)) else np.zeros(0, np.int32)
        sep_sizes = np.cumsum([len(packed[c][1]) for c in range(1024)]).astype(np.int32)
        flat = np.ascontiguousarray(Cc.reshape(4096, 16))
        if bc7:       # Chain::encode_color_bc7 (:299-316): bc7enc, mode 6 only, opaque
            blocks = np.zeros((4096, 16), np.uint8)
            rgba = np.ascontiguousarray(flat | np.uint32(0xFF000000))
            for k in range(4096):
                ref.ref_bc7_encode(rgba[k].ctypes.data, blocks[k].ctypes.data)
        else:         # Chain::encode_color_bc1 (:282-297): the reference's rgbcx, 16 points per block, chain-major
            blocks = np.zeros((4096, 8), np.uint8)
            for k in range(4096):
                ref.ref_bc1_encode(flat[k].ctypes.data, blocks[k].ctypes.data)
        mins = [int(A.min()) for A in (X, Y, Z)]; maxs = [int(A.max()) for A in (X, Y, Z)]
        f32 = np.float32
        records.append(dict(
            point_offset=b * 65536, las_scale=scale, las_offset=offset,
            # float(int) * double scale + double offset, stored in a float field (src/preprocess.cpp:1082-1087)
            bbox_min=[float(f32(float(f32(m)) * s_ + o_)) for m, s_, o_ in zip(mins, scale, offset)],
            bbox_max=[float(f32(float(f32(m)) * s_ + o_)) for m, s_, o_ in zip(maxs, scale, offset)],
            las_min=[float(f32(v)) for v in las_min], las_max=[float(f32(v)) for v in las_max],
            start_values=np.stack([X[:, 0], Y[:, 0], Z[:, 0]], 1).reshape(-1), separate_sizes=sep_sizes,
            decoder_values=tv, decoder_cw_len=tl, cluster_sizes=np.array(cluster_sizes, np.int32),
            encoding=np.array(enc, np.uint32), separate=sep.astype(np.int32), color=blocks.reshape(-1)))
        records[-1]["_chains"] = chains
        records[-1]["_xyz"] = np.stack([X, Y, Z], 2)
    return assemble_huffman_file(records), records, (las_min, las_max), len(px)


def surface_patch(rng, side, spacing=100):
    """A side x side patch of a heightfield at the benchmark stream's density (0.1 m spacing for 100, LAS scale 0.001)."""
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="xy")
    px = (100_000 + gx * spacing + rng.integers(-40, 41, gx.shape)).ravel().astype(np.int32)
    py = (200_000 + gy * spacing + rng.integers(-40, 41, gy.shape)).ravel().astype(np.int32)
    pz = (40_000 + 3000 * np.sin(px / 3000.0) * np.cos(py / 4100.0) + 400 * np.sin(px / 170.0 + py / 230.0)).astype(np.int64)
    pz = (pz + rng.integers(-15, 16, pz.shape)).astype(np.int32)
    cr = np.clip(128 + 100 * np.sin(px / 5000.0) + rng.integers(-6, 7, px.shape), 0, 255).astype(np.uint32)
    cg = np.clip(128 + 100 * np.cos(py / 7000.0) + rng.integers(-6, 7, px.shape), 0, 255).astype(np.uint32)
    cb = np.clip(90 + (pz - 36_000) // 60 + rng.integers(-6, 7, px.shape), 0, 255).astype(np.uint32)
    return px, py, pz, (cr | (cg << 8) | (cb << 16)).astype(np.uint32)


def expected_frames(of, cams, W, H, basic=True):
    """Oracle frames of a fixture for its cameras x {LOD 100 no cull, LOD 10 cull}: SHA-256 of every buffer."""
    cases = []
    for name, p in cams.items():
        for lod, cull in ((100, 0), (10, 1)):
            p.lod_percent = lod; p.enable_frustum_culling = cull
            case = {"camera": name, "lod_percent": lod,
                    "params": {"transform": list(p.transform), "world_view": list(p.world_view), "proj": list(p.proj),
                               "enable_frustum_culling": p.enable_frustum_culling}}
            if basic:
                fb, s1 = of.render_basic(p)
                ties = of.count_depth_ties(p, fb)
                case.update({"stats_basic": s1, "covered_pixels": int((fb[:W * H] != 0xFFFFFFFFFFFFFFFF).sum()),
                             "depth_tie_pixels": ties[0], "depth_tie_pixels_other_colour": ties[1],
                             "fb_basic_sha256": sha(fb), "rgba_basic_sha256": sha(oracle.resolve_basic(p, fb))})
            hfb, s2 = of.render_hqs_depth(p)
            rg, ba, _ = of.render_hqs_color(p, hfb)
            case.update({"stats_hqs": s2, "fb_hqs_sha256": sha(hfb), "rg_sha256": sha(rg), "ba_sha256": sha(ba),
                         "rgba_hqs_sha256": sha(oracle.resolve_hqs(p, hfb, rg, ba))})
            if not basic:
                case["covered_pixels"] = int(((hfb[:W * H] >> np.uint64(32)) != 0xFFFFFFFF).sum())
            cases.append(case)
    return cases


def lockstep_vs_source(of, records):
    tail_positions, wrong_points = [], 0
    for b, r in enumerate(records):
        dec = of.decode_batch(b)                                    # (1024, 64, 3) absolute coordinates
        bad = np.argwhere((dec != r["_xyz"]).any(axis=2))
        wrong_points += len(bad)
        tail_positions += [int(i) for _, i in bad]
    return {"wrong_points": wrong_points, "min_in_chain_position": min(tail_positions) if tail_positions else None,
            "positions_histogram": {str(k): int(v) for k, v in zip(*np.unique(tail_positions, return_counts=True))}}


def ref_packed_batch():
    """Two batches packed by the reference's own huffman.h / mymorton.h / rgbcx (oracle/_ref), see the module docstring."""
    rng = np.random.default_rng(20241004)
    px, py, pz, col = surface_patch(rng, 362)                       # a 36 m x 36 m patch: 362^2 points
    scale, offset = (0.001, 0.001, 0.001), (0.0, 0.0, 0.0)
    n_in = len(px)
    src_points = (px.copy(), py.copy(), pz.copy(), col.copy())
    data, records, (las_min, las_max), n_padded = pack_with_the_reference_library(px, py, pz, col, scale, offset)
    open(os.path.join(G, "ref_packed_batch.huffman"), "wb").write(data)

    # what the lockstep (kernel-order) decode makes of it, against the source points and the reference's scalar decoder
    of = oracle.OracleFile(data)
    # this repository's own encoder on the same points: everything but the colour blocks (and the symbols parked in escape
    # table slots, which no decoder reads) should be what the reference's library produced
    las = P.LasInfo()
    for k in range(3):
        las.scale[k] = scale[k]; las.offset[k] = offset[k]; las.min[k] = las_min[k]; las.max[k] = las_max[k]
    own, _ = P.encode_points(*src_points, las, morton_sort=True, nthreads=1)
    own_of = oracle.OracleFile(bytes(own.view()))
    same = {"encoded_words": bool(np.array_equal(own_of.encoded()[:int(of.s.encoded_words)], of.encoded())),
            "escape_words": bool(np.array_equal(own_of.separate()[:int(of.s.separate_words)], of.separate())),
            "colour_blocks_equal": int(sum(bytes(own.view())[-32768 * (len(records) - b):][:32768] == r["color"].tobytes() for b, r in enumerate(records)))}
    W = H = 512
    cams = {
        # kernel-space positions are relative to the LAS minimum (render.cu:399-400): the patch spans [0, 36] m
        "patch": P.camera_orbit(-0.4, -0.6, 60.0, (18.0, 18.0, 3.0), W, H),
        "near": P.camera_orbit(0.9, -0.35, 14.0, (12.0, 26.0, 3.0), W, H),
        "far": P.camera_orbit(2.2, -0.8, 420.0, (18.0, 18.0, 3.0), W, H),          # small on screen: the LOD percentage decides
    }
    out = {"stream_sha256": hashlib.sha256(data).hexdigest(), "width": W, "height": H,
           "source_points": n_in, "padded_points": n_padded, "batches": len(records),
           "encoded_bits_per_point": round(32.0 * sum(len(r["encoding"]) for r in records) / n_padded, 3),
           "escape_words": int(sum(len(r["separate"]) for r in records)),
           "own_encoder_on_same_points": same,
           "lockstep_vs_source": lockstep_vs_source(of, records),
           "cases": expected_frames(of, cams, W, H)}
    json.dump(out, open(os.path.join(G, "ref_packed_batch_expected.json"), "w"), indent=1)
    print("ref_packed_batch:", len(data), "bytes;", out["encoded_bits_per_point"], "bit/pt;", out["escape_words"], "escape words;",
          "lockstep-vs-source", out["lockstep_vs_source"]["wrong_points"], "min pos", out["lockstep_vs_source"]["min_in_chain_position"],
          "; own encoder", same, "; covered", [c["covered_pixels"] for c in out["cases"]], "ties", [(c["depth_tie_pixels"], c["depth_tie_pixels_other_colour"]) for c in out["cases"]])


def ref_packed_bc7():
    """One batch with BC7 mode-6 colour blocks from the reference's bc7enc (a file of a reference built with COLOR_COMPRESSION == 7,
    include/BatchDumpData.h:130-136): drawn by the HQS method only (huffman_hqs/render.cu:297-303)."""
    rng = np.random.default_rng(20241005)
    px, py, pz, col = surface_patch(rng, 256)                       # 65 536 points exactly: no padding
    scale, offset = (0.001, 0.001, 0.001), (0.0, 0.0, 0.0)
    data, records, _, n_padded = pack_with_the_reference_library(px, py, pz, col, scale, offset, bc7=True)
    open(os.path.join(G, "ref_packed_bc7.huffman"), "wb").write(data)
    of = oracle.OracleFile(data)
    assert of.s.color_format == 7
    W = H = 384
    cams = {"patch": P.camera_orbit(-0.4, -0.6, 42.0, (12.8, 12.8, 3.0), W, H),
            "near": P.camera_orbit(0.9, -0.35, 10.0, (8.0, 18.0, 3.0), W, H)}
    out = {"stream_sha256": hashlib.sha256(data).hexdigest(), "width": W, "height": H, "source_points": len(px), "padded_points": n_padded,
           "batches": len(records), "color_format": 7,
           "encoded_bits_per_point": round(32.0 * sum(len(r["encoding"]) for r in records) / n_padded, 3),
           "escape_words": int(sum(len(r["separate"]) for r in records)),
           "lockstep_vs_source": lockstep_vs_source(of, records),
           "cases": expected_frames(of, cams, W, H, basic=False)}
    json.dump(out, open(os.path.join(G, "ref_packed_bc7_expected.json"), "w"), indent=1)
    print("ref_packed_bc7:", len(data), "bytes;", out["encoded_bits_per_point"], "bit/pt; lockstep-vs-source", out["lockstep_vs_source"]["wrong_points"],
          "covered", [c["covered_pixels"] for c in out["cases"]])


def ref_packed_lowentropy():
    """SURVEY B.4's worst case: 10 000 real points padded to one 65 536-point batch by repeating the last point
    (src/preprocess.cpp:945-955). Five sixths of the chains are runs of identical points -- one-bit codes -- so the reference's
    tail over-reads and the de-synchronisation they cause start 32-64 symbols before the end of a chain instead of in its last
    one or two: the lockstep decode differs from the source in hundreds of points, and the frames below are what the
    reference's kernels would draw of them (garbage tails included)."""
    rng = np.random.default_rng(20241006)
    px, py, pz, col = surface_patch(rng, 100)                       # 10 000 points
    scale, offset = (0.001, 0.001, 0.001), (0.0, 0.0, 0.0)
    data, records, _, n_padded = pack_with_the_reference_library(px, py, pz, col, scale, offset)
    open(os.path.join(G, "ref_packed_lowentropy.huffman"), "wb").write(data)
    of = oracle.OracleFile(data)
    W = H = 384
    cams = {"patch": P.camera_orbit(-0.4, -0.6, 18.0, (5.0, 5.0, 3.0), W, H),
            "wide": P.camera_orbit(0.7, -0.5, 90.0, (5.0, 5.0, 3.0), W, H)}      # wide enough to see where the garbage tails land
    out = {"stream_sha256": hashlib.sha256(data).hexdigest(), "width": W, "height": H, "source_points": len(px), "padded_points": n_padded,
           "batches": len(records),
           "encoded_bits_per_point": round(32.0 * sum(len(r["encoding"]) for r in records) / n_padded, 3),
           "escape_words": int(sum(len(r["separate"]) for r in records)),
           "lockstep_vs_source": lockstep_vs_source(of, records),
           "cases": expected_frames(of, cams, W, H)}
    json.dump(out, open(os.path.join(G, "ref_packed_lowentropy_expected.json"), "w"), indent=1)
    print("ref_packed_lowentropy:", len(data), "bytes;", out["encoded_bits_per_point"], "bit/pt; lockstep-vs-source", out["lockstep_vs_source"],
          "covered", [c["covered_pixels"] for c in out["cases"]])


FIXTURES = {"config1": config1, "bc1_ref": bc1_ref, "bc7_ref": bc7_ref, "huffman_ref": huffman_ref, "ref_packed_batch": ref_packed_batch,
            "ref_packed_bc7": ref_packed_bc7, "ref_packed_lowentropy": ref_packed_lowentropy}

if __name__ == "__main__":
    if len(sys.argv) > 1:                       # only the named fixtures: python tools/make_golden.py bc7_ref ...
        for name in sys.argv[1:]:
            FIXTURES[name]()
        sys.exit(0)
    # (config1.* is this repository's own encoder's output: it is versioned, not regenerated silently -- name it to rewrite it)
    if os.path.exists(oracle.REF_LIB):
        bc1_ref()
        bc7_ref()
        huffman_ref()
        ref_packed_batch()
        ref_packed_bc7()
        ref_packed_lowentropy()
    else:
        print("oracle/_ref/libpcr_ref.so missing: reference-derived fixtures not regenerated")
