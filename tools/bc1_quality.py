#!/usr/bin/env python3
"""PSNR of this repository's BC1 block encoder (csrc/pcr_codec_common.h, shared by the CPU and the GPU encoder) against the
reference's rgbcx::encode_bc1(level 8) as src/preprocess.cpp:282-297 calls it, on the committed colour set
tests/golden/bc1_ref_blocks.npz (256 blocks: solid, noisy, two-colour gradients, random; colours + rgbcx's blocks).
Both block sets are decoded with the kernel's decoder (always 4-colour mode, render.cu:23-65)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcrhpg24_amd import _native as N  # noqa: E402


def decode(block8):
    c0 = int(block8[0]) | int(block8[1]) << 8
    c1 = int(block8[2]) | int(block8[3]) << 8

    def ex(c):
        r, g, b = (c >> 11) & 31, (c >> 5) & 63, c & 31
        return np.array([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)], np.int64)
    p0, p1 = ex(c0), ex(c1)
    pal = [p0, p1, (2 * p0 + p1) // 3, (p0 + 2 * p1) // 3]
    sel = int.from_bytes(bytes(block8[4:8]), "little")
    return np.stack([pal[(sel >> (2 * i)) & 3] for i in range(16)])


def psnr(colors, blocks):
    se, n = 0, 0
    per_kind = [[0, 0] for _ in range(4)]
    for k in range(len(colors)):
        src = np.stack([(colors[k] >> s) & 255 for s in (0, 8, 16)], 1).astype(np.int64)
        e = int(((decode(blocks[k]) - src) ** 2).sum())
        se += e; n += 48
        per_kind[k % 4][0] += e; per_kind[k % 4][1] += 48
    f = lambda s, m: 10 * np.log10(255.0 ** 2 / max(s / m, 1e-12))
    return f(se, n), [f(a, b) for a, b in per_kind]


def main():
    d = np.load(os.path.join(ROOT, "tests", "golden", "bc1_ref_blocks.npz"))
    colors, ref_blocks = d["colors"], d["blocks"]
    lib = N.host_lib()
    own = np.zeros_like(ref_blocks)
    for k in range(len(colors)):
        c = np.ascontiguousarray(colors[k], np.uint32)
        lib.pcr_bc1_encode_block(c.ctypes.data_as(C.c_void_p), own[k].ctypes.data_as(C.c_void_p))
    a, ak = psnr(colors, own)
    b, bk = psnr(colors, ref_blocks)
    kinds = ("solid", "noisy", "two-colour gradient", "random")
    print("PSNR dB  own %.2f   rgbcx level 8 %.2f   gap %.2f" % (a, b, b - a))
    for name, x, y in zip(kinds, ak, bk):
        print("  %-20s own %6.2f  rgbcx %6.2f  gap %5.2f" % (name, x, y, y - x))
    return a, b


if __name__ == "__main__":
    main()
