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


if __name__ == "__main__":
    config1()
    if os.path.exists(oracle.REF_LIB):
        bc1_ref()
        huffman_ref()
    else:
        print("oracle/_ref/libpcr_ref.so missing: reference-derived fixtures not regenerated")
