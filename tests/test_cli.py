"""The C++ host adapters as executables: pcr_preprocess (reference: `preprocess in.las out.huffman sort`,
src/preprocess.cpp:1167-1279) and pcr_render (reference: src/main.cpp flow through Method/Resource)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import pcrhpg24_amd as P
from pcrhpg24_amd import build
from tests import oracle, scenes


def write_las(path, x, y, z, r, g, b, scale=(0.001, 0.001, 0.001), offset=(10.0, 20.0, 30.0)):
    """Minimal LAS 1.2, point format 2 (26-byte records): what LasLoader::loadSync reads (preprocess.cpp:74-171)."""
    n = len(x)
    hdr = bytearray(227)
    hdr[0:4] = b"LASF"
    hdr[24], hdr[25] = 1, 2
    struct.pack_into("<H", hdr, 94, 227)
    struct.pack_into("<I", hdr, 96, 227)
    hdr[104] = 2
    struct.pack_into("<H", hdr, 105, 26)
    struct.pack_into("<I", hdr, 107, n)
    struct.pack_into("<3d", hdr, 131, *scale)
    struct.pack_into("<3d", hdr, 155, *offset)
    wx, wy, wz = x * scale[0] + offset[0], y * scale[1] + offset[1], z * scale[2] + offset[2]
    struct.pack_into("<6d", hdr, 179, wx.max(), wx.min(), wy.max(), wy.min(), wz.max(), wz.min())
    rec = np.zeros(n, dtype=[("x", "<i4"), ("y", "<i4"), ("z", "<i4"), ("pad", "V8"), ("r", "<u2"), ("g", "<u2"), ("b", "<u2")])
    rec["x"], rec["y"], rec["z"], rec["r"], rec["g"], rec["b"] = x, y, z, r, g, b
    assert rec.itemsize == 26
    with open(path, "wb") as f:
        f.write(hdr); f.write(rec.tobytes())
    las = P.LasInfo()
    for k, (lo, hi) in enumerate(((wx.min(), wx.max()), (wy.min(), wy.max()), (wz.min(), wz.max()))):
        las.scale[k], las.offset[k], las.min[k], las.max[k] = scale[k], offset[k], lo, hi
    return las


def test_preprocess_cli_equals_library_encoder(tmp_path):
    build.build_tools()
    rng = np.random.default_rng(2)
    n = 70_000
    x = rng.integers(0, 400000, n).astype(np.int32); y = rng.integers(0, 300000, n).astype(np.int32)
    z = (1000 * np.sin(x / 40000.0) + rng.integers(-20, 20, n)).astype(np.int32)
    r = rng.integers(0, 65536, n).astype(np.uint16); g = rng.integers(0, 256, n).astype(np.uint16); b = rng.integers(0, 65536, n).astype(np.uint16)
    las = write_las(tmp_path / "in.las", x, y, z, r, g, b)
    out = tmp_path / "out.huffman"
    res = subprocess.run([build.PREPROCESS_BIN, str(tmp_path / "in.las"), str(out), "1", "2"], stdout=subprocess.PIPE, text=True)
    assert res.returncode == 0 and "batches 2" in res.stdout
    conv = lambda v: np.where(v > 255, v // 256, v).astype(np.uint32)          # preprocess.cpp:150-152
    color = conv(r) | (conv(g) << 8) | (conv(b) << 16)
    image, st = P.encode_points(x, y, z, color, las, morton_sort=True, nthreads=2)
    assert out.read_bytes() == bytes(image.view())
    of = oracle.OracleFile(out.read_bytes())
    g0 = of.batch(0)
    assert (g0.scale_x, g0.offset_y) == (0.001, 20.0)
    assert abs(g0.las_min_x - (x.min() * 0.001 + 10.0)) < 1e-3
    assert subprocess.run([build.PREPROCESS_BIN, str(tmp_path / "nope.las"), str(out), "1"], stderr=subprocess.PIPE).returncode != 0
    fixed = tmp_path / "fixed.huffman"
    assert subprocess.run([build.PREPROCESS_BIN, str(tmp_path / "in.las"), str(fixed), "1", "2", "--pad-tails"], stdout=subprocess.PIPE).returncode == 0
    image2, _ = P.encode_points(x, y, z, color, las, morton_sort=True, nthreads=2, pad_tails=True)
    assert fixed.read_bytes() == bytes(image2.view()) and fixed.read_bytes() != out.read_bytes()


@pytest.mark.gpu
# ("huffman_cuda": the reference's first name for the basic method, modules/huffman_cuda/huffman_cuda.h:66 -- the one north_star lists)
@pytest.mark.parametrize("method,extra", [("huffman_mem_iter_cuda", []), ("huffman_hqs", []),
                                          ("huffman_mem_iter_cuda", ["--async-load"]), ("huffman_cuda", [])])
def test_render_cli_matches_oracle(tmp_path, method, extra):
    build.build_tools()
    image, _ = scenes.synth_stream(2_000_000)
    path = tmp_path / "scene.huffman"
    path.write_bytes(bytes(image.view()))
    W, H = 640, 360
    cam = ["-0.15", "-0.57", "1500", "500", "500", "40"]
    res = subprocess.run([build.RENDER_BIN, str(path), "--method", method, "--size", f"{W}x{H}", "--camera", *cam,
                          "--lod", "0.1", "--dump-fb", str(tmp_path / "fb.u64"), "--dump-rgba", str(tmp_path / "o.ppm"), *extra],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["method"] == method and info["batches"] == 31
    fb = np.fromfile(tmp_path / "fb.u64", np.uint64)
    of = oracle.OracleFile(image.view())
    p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), W, H)     # Debug::LOD 0.1, culling on
    ofb, ost = (of.render_hqs_depth(p) if method == "huffman_hqs" else of.render_basic(p))
    assert np.array_equal(fb, ofb[:W * H])
    assert info["points_iterated"] == ost["points_iterated"] and info["covered_pixels"] == int((ofb[:W * H] != 2 ** 64 - 1).sum())
    assert (tmp_path / "o.ppm").stat().st_size > W * H * 3


def read_single_channel_exr(path):
    """Minimal reader for what saveSingleChannelEXR writes: scanline OpenEXR, one FLOAT channel, no compression."""
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and struct.unpack_from("<I", b, 4)[0] == 2
    o, attrs = 8, {}
    while b[o] != 0:
        e = b.index(0, o); name = b[o:e].decode(); o = e + 1
        e = b.index(0, o); typ = b[o:e].decode(); o = e + 1
        size = struct.unpack_from("<i", b, o)[0]; o += 4
        attrs[name] = (typ, b[o:o + size]); o += size
    o += 1
    assert attrs["compression"][1] == b"\x00" and attrs["lineOrder"][1] == b"\x00"
    assert attrs["channels"][1][:2] == b"Z\x00" and struct.unpack_from("<i", attrs["channels"][1], 2)[0] == 2
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offs = struct.unpack_from("<%dQ" % h, b, o)
    img = np.zeros((h, w), np.float32)
    for y in range(h):
        yy, n = struct.unpack_from("<2i", b, offs[y])
        assert yy == y and n == 4 * w
        img[y] = np.frombuffer(b, "<f4", w, offs[y] + 8)
    return img


@pytest.mark.gpu
def test_render_cli_depth_dump(tmp_path):
    """--dump-depth: the reference's Debug::saveDepthMap path (huffman_hqs.h:217-237): depth floats, image flipped, 0 = empty."""
    build.build_tools()
    image, _ = scenes.synth_stream(2_000_000)
    path = tmp_path / "scene.huffman"
    path.write_bytes(bytes(image.view()))
    W, H = 640, 360
    cam = ["-0.15", "-0.57", "1500", "500", "500", "40"]
    res = subprocess.run([build.RENDER_BIN, str(path), "--method", "huffman_hqs", "--size", f"{W}x{H}", "--camera", *cam,
                          "--lod", "0.1", "--dump-depth", str(tmp_path / "depth.exr")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    img = read_single_channel_exr(tmp_path / "depth.exr")
    assert img.shape == (H, W)
    of = oracle.OracleFile(image.view())
    p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), W, H)
    ofb, _ = of.render_hqs_depth(p)
    d = (ofb[:W * H] >> np.uint64(32)).astype(np.uint32)
    exp = np.where(d == 0xFFFFFFFF, np.uint32(0), d).view(np.float32).reshape(H, W)[::-1]
    assert np.array_equal(img.view(np.uint32), exp.view(np.uint32))


@pytest.mark.gpu
def test_render_cli_loop_las_cuda_matches_oracle(tmp_path):
    """pcr_render <file.las> --method loop_las_cuda: C++ ComputeLasData + ComputeLoopLasCUDA against the oracle."""
    build.build_tools()
    n = 5 * 65536 + 99
    x, y, z, c = P.synth_points(n, scenes.SEED, 0, n)
    r, g, b = (c & 255).astype(np.uint16), ((c >> 8) & 255).astype(np.uint16), ((c >> 16) & 255).astype(np.uint16)
    write_las(tmp_path / "scene.las", x, y, z, r, g, b, offset=(0.0, 0.0, 0.0))
    W, H = 640, 360
    cam = ["-0.15", "-0.57", "1500", "500", "500", "40"]
    res = subprocess.run([build.RENDER_BIN, str(tmp_path / "scene.las"), "--method", "loop_las_cuda", "--size", f"{W}x{H}",
                          "--camera", *cam, "--dump-fb", str(tmp_path / "fb.u64")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["method"] == "loop_las_cuda" and info["batches"] == 6
    px, py, pz, pc, las = P.read_las(str(tmp_path / "scene.las"))
    q = P.las_quantize(px, py, pz, pc, las)
    p = P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), W, H)
    ofb, ost = oracle.render_las(*q[:4], p)
    assert np.array_equal(np.fromfile(tmp_path / "fb.u64", np.uint64), ofb[:W * H])
    assert info["points_iterated"] == ost["points_iterated"] == 5 * 65536


@pytest.mark.gpu
def test_preprocess_cli_gpu_equals_cpu(tmp_path):
    """pcr_preprocess --gpu (pcr_gpu_encode_points) writes the same file as the CPU path."""
    build.build_tools()
    n = 150_000
    x, y, z, c = P.synth_points(n, scenes.SEED, 0, n)
    r, g, b = (c & 255).astype(np.uint16), ((c >> 8) & 255).astype(np.uint16), ((c >> 16) & 255).astype(np.uint16)
    write_las(tmp_path / "in.las", x, y, z, r, g, b)
    for extra in ([], ["--pad-tails"]):
        a, bb = tmp_path / "cpu.huffman", tmp_path / "gpu.huffman"
        assert subprocess.run([build.PREPROCESS_BIN, str(tmp_path / "in.las"), str(a), "1", "4", *extra], stdout=subprocess.PIPE).returncode == 0
        res = subprocess.run([build.PREPROCESS_BIN, str(tmp_path / "in.las"), str(bb), "1", "--gpu", *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert res.returncode == 0, res.stderr
        assert a.read_bytes() == bb.read_bytes()


@pytest.mark.parametrize("damage", ["negative size", "truncated"])
def test_render_cli_reports_a_damaged_file_instead_of_terminating(tmp_path, damage):
    """the size table is validated when the header is read, and an error inside the reader thread is handed
    to the frame loop (exit code 1 and a message) instead of ending in std::terminate. Needs no GPU: both checks come first."""
    build.build_tools()
    image, _ = scenes.synth_stream(10_000)
    data = bytearray(image.view())
    if damage == "negative size":
        data[40:48] = struct.pack("<q", -5)
    else:
        data = data[:len(data) // 2]
    path = tmp_path / "bad.huffman"
    path.write_bytes(bytes(data))
    res = subprocess.run([build.RENDER_BIN, str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert res.returncode == 1 and "pcr_render:" in res.stderr and "terminate" not in res.stderr
    assert ("record size" in res.stderr) or ("exceed the file" in res.stderr) or ("no HIP device" in res.stderr and damage == "truncated") \
        or "Renderer" in res.stderr or "pcr_create" in res.stderr


@pytest.mark.gpu
def test_bc7_file_through_the_cpp_tools(tmp_path):
    """`pcr_preprocess --bc7` writes what a reference built with COLOR_COMPRESSION == 7 writes (16 colour bytes per 16 points);
    pcr_render draws it with the HQS method like the oracle and refuses the basic method."""
    build.build_tools()
    rng = np.random.default_rng(5)
    n = 140_000
    x = rng.integers(0, 400000, n).astype(np.int32); y = rng.integers(0, 300000, n).astype(np.int32)
    z = (1000 * np.sin(x / 40000.0) + rng.integers(-20, 20, n)).astype(np.int32)
    r = ((x // 1600) % 256).astype(np.uint16); g = ((y // 1200) % 256).astype(np.uint16); b = rng.integers(0, 256, n).astype(np.uint16)
    las = write_las(tmp_path / "in.las", x, y, z, r, g, b)
    out = tmp_path / "bc7.huffman"
    assert subprocess.run([build.PREPROCESS_BIN, str(tmp_path / "in.las"), str(out), "1", "2", "--bc7"], stdout=subprocess.PIPE).returncode == 0
    color = r.astype(np.uint32) | (g.astype(np.uint32) << 8) | (b.astype(np.uint32) << 16)
    image, _ = P.encode_points(x, y, z, color, las, morton_sort=True, nthreads=2, bc7=True)
    assert out.read_bytes() == bytes(image.view())
    of = oracle.OracleFile(out.read_bytes())
    assert of.s.color_format == 7
    W, H = 480, 270
    cam = ["-0.15", "-0.57", "700", "210", "170", "30"]
    res = subprocess.run([build.RENDER_BIN, str(out), "--method", "huffman_hqs", "--size", f"{W}x{H}", "--camera", *cam,
                          "--dump-fb", str(tmp_path / "fb.u64"), "--dump-rgba", str(tmp_path / "o.ppm")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    p = P.camera_orbit(-0.15, -0.57, 700.0, (210.0, 170.0, 30.0), W, H)
    hfb, _ = of.render_hqs_depth(p)
    assert np.array_equal(np.fromfile(tmp_path / "fb.u64", np.uint64), hfb[:W * H])
    rg, ba, _ = of.render_hqs_color(p, hfb)
    want = oracle.resolve_hqs(p, hfb, rg, ba).reshape(H, W)[::-1]                 # the PPM starts with the top row
    ppm = (tmp_path / "o.ppm").read_bytes()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert ppm.startswith(head)
    got = np.frombuffer(ppm, np.uint8, W * H * 3, len(head)).reshape(H, W, 3).astype(np.uint32)
    assert np.array_equal(got[..., 0] | (got[..., 1] << 8) | (got[..., 2] << 16), want & 0xFFFFFF)
    assert int((hfb[:W * H] != 2 ** 64 - 1).sum()) > 2000
    res = subprocess.run([build.RENDER_BIN, str(out), "--method", "huffman_mem_iter_cuda", "--size", f"{W}x{H}", "--camera", *cam],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "BC7" in res.stderr
