"""CPU checks of the 10-10-10 path's host side: the quantiser (pcr_las_quantize) against the loader shader's
arithmetic restated in numpy, the LAS reader, and the oracle's level selection / renderer against a numpy restatement."""
import struct

import numpy as np

import pcrhpg24_amd as P
from tests import oracle, scenes

PPB = 65536


def _np_positions(x, y, z, las):
    """getPoint, computeLasLoader.cs:178-180 (uScale / uBoxMin are float uniforms, the arithmetic is double)."""
    out = []
    for k, a in enumerate((x, y, z)):
        sc, mn = np.float64(np.float32(las.scale[k])), np.float64(np.float32(las.min[k]))
        out.append((a.astype(np.float64) * sc + las.offset[k] - mn).astype(np.float32))
    return out


def _unpack(w):
    return w & 1023, (w >> 10) & 1023, (w >> 20) & 1023


def test_quantiser_matches_numpy_restatement():
    n = 3 * PPB + 1234                      # ragged: the last batch is partly filled
    x, y, z, c, las = scenes.random_points(n, seed=11)
    batches, x12, x8, x4, rgba = P.las_quantize(x, y, z, c, las, nthreads=3)
    assert len(batches) == 4 and all(len(a) == 4 * PPB for a in (x12, x8, x4, rgba))
    pos = _np_positions(x, y, z, las)
    for b in range(4):
        lo, hi = b * PPB, min(n, (b + 1) * PPB)
        g = batches[b]
        assert g.num_points == hi - lo
        mn = [pos[k][lo:hi].min() for k in range(3)]
        mx = [pos[k][lo:hi].max() for k in range(3)]
        assert [g.min_x, g.min_y, g.min_z] == mn and [g.max_x, g.max_y, g.max_z] == mx
        comps4, comps8, comps12 = _unpack(x4[lo:hi]), _unpack(x8[lo:hi]), _unpack(x12[lo:hi])
        for k in range(3):
            size = np.float32(mx[k] - mn[k])
            t = ((pos[k][lo:hi] - np.float32(mn[k])) / size * np.float32(2.0 ** 30)).astype(np.float32)
            q = np.minimum(t.astype(np.uint64), 2 ** 30 - 1).astype(np.uint32)       # processPoints :288-294
            assert np.array_equal(comps4[k], q >> 20)
            assert np.array_equal(comps8[k], (q >> 10) & 1023)
            assert np.array_equal(comps12[k], q & 1023)
        assert np.array_equal(rgba[lo:hi], c[lo:hi])
    # unused slots of the ragged batch are zero
    assert not x4[n:].any() and not x8[n:].any() and not x12[n:].any() and not rgba[n:].any()


def test_quantiser_flat_box_and_single_point():
    las = P.LasInfo()
    for k in range(3):
        las.scale[k], las.offset[k], las.min[k], las.max[k] = 0.01, 0.0, 0.0, 10.0
    x = np.array([5, 5, 5], np.int32)
    batches, x12, x8, x4, rgba = P.las_quantize(x, x, x, np.array([1, 2, 3], np.uint32), las)
    assert batches[0].num_points == 3 and batches[0].min_x == batches[0].max_x
    assert not x4[:3].any() and list(rgba[:3]) == [1, 2, 3]


def test_level_reconstruction_error_bounds():
    """Each extra level refines the position: |decoded - exact| <= box/2^10, box/2^20, ~float eps."""
    n = PPB
    x, y, z, c, las = scenes.random_points(n, seed=5, spread=1 << 18)
    batches, x12, x8, x4, _ = P.las_quantize(x, y, z, c, las)
    pos = _np_positions(x, y, z, las)
    g = batches[0]
    lo = [g.min_x, g.min_y, g.min_z]
    size = [g.max_x - g.min_x, g.max_y - g.min_y, g.max_z - g.min_z]
    c4, c8, c12 = _unpack(x4[:n]), _unpack(x8[:n]), _unpack(x12[:n])
    for k in range(3):
        d10 = c4[k].astype(np.float64) * (size[k] / 1024.0) + lo[k]
        d20 = ((c4[k].astype(np.uint64) << 20) | (c8[k].astype(np.uint64) << 10)).astype(np.float64) * (size[k] / 2 ** 30) + lo[k]
        d30 = ((c4[k].astype(np.uint64) << 20) | (c8[k].astype(np.uint64) << 10) | c12[k]).astype(np.float64) * (size[k] / 2 ** 30) + lo[k]
        e = pos[k].astype(np.float64)
        assert np.all(e - d10 >= -1e-3) and np.all(e - d10 <= size[k] / 1024.0 + 1e-3)
        assert np.all(np.abs(e - d20) <= size[k] / 2 ** 20 + 1e-3)
        assert np.all(np.abs(e - d30) <= size[k] * 2.0 ** -22 + 1e-3)


def test_read_las_roundtrip(tmp_path):
    n, fmt, bpp = 1000, 2, 26
    rng = np.random.default_rng(3)
    xyz = rng.integers(-100000, 100000, (n, 3), dtype=np.int64).astype("<i4")
    rgb = rng.integers(0, 65536, (n, 3), dtype=np.int64).astype("<u2")
    rgb[:10] = rng.integers(0, 256, (10, 3))                         # 8-bit colours stored as-is
    hdr = bytearray(227)
    hdr[:4] = b"LASF"
    hdr[24], hdr[25] = 1, 2
    struct.pack_into("<H", hdr, 94, 227)
    struct.pack_into("<I", hdr, 96, 227)
    hdr[104] = fmt
    struct.pack_into("<H", hdr, 105, bpp)
    struct.pack_into("<I", hdr, 107, n)
    struct.pack_into("<6d", hdr, 131, 0.001, 0.002, 0.003, 10.0, 20.0, 30.0)
    struct.pack_into("<6d", hdr, 179, 110.0, -90.0, 220.0, -180.0, 330.0, -270.0)
    rec = np.zeros((n, bpp), np.uint8)
    rec[:, :12] = xyz.view(np.uint8).reshape(n, 12)
    rec[:, 20:26] = rgb.view(np.uint8).reshape(n, 6)
    path = tmp_path / "t.las"
    path.write_bytes(bytes(hdr) + rec.tobytes())
    x, y, z, color, las = P.read_las(str(path))
    assert np.array_equal(x, xyz[:, 0]) and np.array_equal(y, xyz[:, 1]) and np.array_equal(z, xyz[:, 2])
    r8 = np.where(rgb > 255, rgb // 256, rgb).astype(np.uint32)
    assert np.array_equal(color, r8[:, 0] | (r8[:, 1] << 8) | (r8[:, 2] << 16))
    assert list(las.scale) == [0.001, 0.002, 0.003] and list(las.offset) == [10.0, 20.0, 30.0]
    assert list(las.min) == [-90.0, -180.0, -270.0] and list(las.max) == [110.0, 220.0, 330.0]
    d = P.ComputeLasData.create(str(path))
    assert d.numPoints == n and d.state == P.Resource.UNLOADED


def _synth_las(total):
    x, y, z, c = P.synth_points(total, scenes.SEED, 0, total)
    return x, y, z, c, P.synth_las_info(total, scenes.SEED)


def test_oracle_levels_cover_all_precisions():
    x, y, z, c, las = _synth_las(6 * PPB)
    batches, *_ = P.las_quantize(x, y, z, c, las)
    seen = set()
    for cam, p in scenes.cameras(640, 360).items():
        for cull in (0, 1):
            q = scenes.with_flags(p, cull=cull)
            seen |= {oracle.las_level(batches[b], q) for b in range(len(batches))}
    assert {0, 1, 2, -1} <= seen, seen


def test_oracle_render_las_matches_numpy_restatement_at_level_2_plus():
    """Independent numpy restatement of render.cu:130-442 for batches that read only the 4-byte level."""
    x, y, z, c, las = _synth_las(4 * PPB)
    batches, x12, x8, x4, rgba = P.las_quantize(x, y, z, c, las)
    p = scenes.with_flags(scenes.cameras(320, 180)["far"], cull=0)
    levels = [oracle.las_level(batches[b], p) for b in range(4)]
    assert all(l >= 2 for l in levels), levels
    fb, st = oracle.render_las(batches, x12, x8, x4, p)
    assert st == {"batches_total": 4, "batches_culled": 0, "points_iterated": 3 * PPB, "batches_double": 0}
    M = np.array(p.transform, np.float32).reshape(4, 4)
    exp = np.full(P.fb_elems(320, 180), 0xFFFFFFFFFFFFFFFF, np.uint64)
    f32 = np.float32

    def dot4(row, px, py, pz):                                        # matMul as an explicit fma chain (App. C)
        acc = (row[0] * px).astype(f32)
        for coef, v in ((row[1], py), (row[2], pz), (row[3], np.ones_like(px))):
            acc = (coef.astype(np.float64) * v.astype(np.float64) + acc.astype(np.float64)).astype(f32)   # exact product + one rounding = fmaf
        return acc

    for b in range(3):                                                # the last batch is not drawn
        g = batches[b]
        w = x4[b * PPB:(b + 1) * PPB]
        comps = _unpack(w)
        pt = []
        for k, (lo, hi) in enumerate(((g.min_x, g.max_x), (g.min_y, g.max_y), (g.min_z, g.max_z))):
            sc = f32(f32(f32(hi) - f32(lo)) / f32(1024.0))
            pt.append((comps[k].astype(f32).astype(np.float64) * np.float64(sc) + np.float64(f32(lo))).astype(f32))
        px, py, pw = dot4(M[0], *pt), dot4(M[1], *pt), dot4(M[3], *pt)
        with np.errstate(all="ignore"):
            nx, ny = (px / pw).astype(f32), (py / pw).astype(f32)
            ok = (pw > 0) & (nx >= -1) & (nx <= 1) & (ny >= -1) & (ny <= 1)
            ix = ((nx.astype(np.float64) * 0.5 + 0.5).astype(f32) * f32(320)).astype(f32)
            iy = ((ny.astype(np.float64) * 0.5 + 0.5).astype(f32) * f32(180)).astype(f32)
        pix = ix[ok].astype(np.int64) + iy[ok].astype(np.int64) * 320
        key = (pw[ok].view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.nonzero(ok)[0] + b * PPB).astype(np.uint64)
        np.minimum.at(exp, pix, key)
    assert np.array_equal(fb, exp)
    img = oracle.resolve_las(p, fb, rgba)
    ids = (fb[:320 * 180] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert np.array_equal(img, np.where(ids < 0x7FFFFFFF, rgba[np.minimum(ids, len(rgba) - 1)], 0x00443322).astype(np.uint32))
