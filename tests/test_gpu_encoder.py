"""GPU encoder (include/pcr_gpu_encode.h) against the CPU encoder: the `.huffman` image must be identical byte for byte,
for every flag combination and input shape the CPU encoder is tested with (it is the one pinned to the reference's
huffman.h / mymorton.h). Runs through the C ABI."""
import numpy as np
import pytest

import pcrhpg24_amd as P
from tests import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = P.Context(0)
    yield c
    c.close()


def first_difference(a: bytes, b: bytes) -> str:
    if len(a) != len(b):
        return f"lengths {len(a)} vs {len(b)}"
    va, vb = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
    d = np.nonzero(va != vb)[0]
    return "identical" if d.size == 0 else f"{d.size} bytes differ, first at {d[0]}"


def check(ctx, x, y, z, c, las, **kw):
    cpu, st_cpu = P.encode_points(x, y, z, c, las, nthreads=4, **kw)
    gpu, st_gpu = ctx.gpu_encode_points(x, y, z, c, las, **kw)
    a, b = bytes(cpu.view()), bytes(gpu.view())
    assert a == b, first_difference(a, b)
    assert st_cpu == st_gpu
    return st_gpu


@pytest.mark.parametrize("sort", [True, False])
@pytest.mark.parametrize("pad_tails", [False, True])
def test_surface_scene(ctx, sort, pad_tails):
    n = 300_000                                 # ragged: 4.58 batches
    x, y, z, c = P.synth_points(n, scenes.SEED, 0, n)
    st = check(ctx, x, y, z, c, P.synth_las_info(n), morton_sort=sort, pad_tails=pad_tails)
    assert st["num_batches"] == 5 and st["num_points_in"] == n


def test_multiple_chunks_and_mostly_padding(ctx):
    n = 150_000
    x, y, z, c = P.synth_points(n, scenes.SEED, 0, n)
    st = check(ctx, x, y, z, c, P.synth_las_info(n), morton_sort=True, chunk_points=131072)   # chunks of 131072 + 18928
    assert st["num_batches"] == 3
    x, y, z, c = P.synth_points(10_000, scenes.SEED, 0, 10_000)
    check(ctx, x, y, z, c, P.synth_las_info(10_000), morton_sort=True)                      # 55 536 padding points


def test_escape_heavy_and_wide_alphabets(ctx):
    x, y, z, c, las = scenes.random_points(131072, seed=3)                                   # ~every symbol distinct
    st = check(ctx, x, y, z, c, las, morton_sort=True)
    assert st["escaped_symbols"] > 6144 * st["num_batches"]
    rng = np.random.default_rng(5)
    n = 65536
    x = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)                 # int32 wrap-around deltas
    y = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    z = rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    check(ctx, x, y, z, np.zeros(n, np.uint32), P.synth_las_info(1), morton_sort=False)
    check(ctx, x, y, z, np.zeros(n, np.uint32), P.synth_las_info(1), morton_sort=True)      # exercises the high key bits


def test_degenerate_and_tiny_alphabets(ctx):
    n = 65536
    las = P.synth_las_info(1)
    check(ctx, np.full(n, 7, np.int32), np.full(n, -3, np.int32), np.full(n, 11, np.int32), np.full(n, 0x336699, np.uint32), las)
    x = (np.arange(n) % 2).astype(np.int32)                                                  # two or three symbols
    check(ctx, x, x * 0, x * 0, np.arange(n, dtype=np.uint32) & 0xFFFFFF, las, morton_sort=False)
    check(ctx, np.arange(5, dtype=np.int32), np.arange(5, dtype=np.int32), np.zeros(5, np.int32), np.arange(5, dtype=np.uint32), las)


def test_skewed_alphabet_with_clipped_codes(ctx):
    """Geometric frequencies make the unclipped tree deeper than 12: long codes become escapes with shared prefixes."""
    rng = np.random.default_rng(8)
    n = 65536 * 2
    mag = np.minimum(rng.geometric(0.35, n), 40)
    x = np.cumsum((2 ** mag.astype(np.int64) % 100003) * rng.choice([-1, 1], n)).astype(np.int64)
    x = (x % (1 << 30)).astype(np.int32)
    y = rng.integers(0, 4, n).astype(np.int32)
    z = (rng.geometric(0.5, n) * 3).astype(np.int32)
    c = rng.integers(0, 1 << 24, n).astype(np.uint32)
    st = check(ctx, x, y, z, c, P.synth_las_info(1), morton_sort=False)
    assert st["escaped_symbols"] > 0


def test_argument_errors(ctx):
    x = np.zeros(4, np.int32)
    with pytest.raises(P.PcrError, match="multiple of 65536"):
        ctx.gpu_encode_points(x, x, x, x.astype(np.uint32), P.synth_las_info(1), chunk_points=1000)
    with pytest.raises(P.PcrError, match="must not exceed"):
        ctx.gpu_encode_points(x, x, x, x.astype(np.uint32), P.synth_las_info(1), chunk_points=65536 * 2000)
    with pytest.raises(P.PcrError, match="no points"):
        ctx.gpu_encode_points(x[:0], x[:0], x[:0], x[:0].astype(np.uint32), P.synth_las_info(1))
