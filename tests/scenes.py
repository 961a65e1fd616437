"""Seeded synthetic streams and cameras shared by the tests (inputs only — no expectations here)."""
from __future__ import annotations

import functools

import numpy as np

import pcrhpg24_amd as P

SEED = 0x5EED


@functools.lru_cache(maxsize=8)
def synth_stream(total_points: int, seed: int = SEED, chunk_points: int = 0):
    """(.huffman image as NativeBytes, encoder stats) of the whole synthetic scene of `total_points`."""
    return P.synth_encode(total_points, seed, chunk_points=chunk_points, nthreads=4)


def cameras(width: int, height: int) -> dict:
    """Cameras over the 1 km synthetic tile (world units = metres, las_min = 0)."""
    c = {
        # whole tile in view, every batch far away -> float path, LOD floor
        "overview": P.camera_orbit(-0.15, -0.57, 1500.0, (500.0, 500.0, 40.0), width, height),
        # near the surface: big projected batches -> double path, npr up to 64, heavy overdraw, partial cull
        "closeup": P.camera_orbit(-1.68, -0.39, 70.0, (300.0, 20.0, 45.0), width, height),
        # looking along the strip from inside: w <= 0 points, frustum-straddling batches
        "inside": P.camera_orbit(0.9, -0.05, 5.0, (400.0, 10.0, 48.0), width, height),
        # far away: every batch small on screen -> float path, LOD between the floor and 64
        "far": P.camera_orbit(0.4, -0.9, 6000.0, (500.0, 500.0, 40.0), width, height),
    }
    return c


def with_flags(p: P.RenderParams, lod_percent=None, cull=None, show_num_points=None, colorize_chunks=None):
    q = p.copy()
    if lod_percent is not None:
        q.lod_percent = lod_percent
    if cull is not None:
        q.enable_frustum_culling = int(cull)
    if show_num_points is not None:
        q.show_num_points = int(show_num_points)
    if colorize_chunks is not None:
        q.colorize_chunks = int(colorize_chunks)
    return q


def random_points(n: int, seed: int, spread: int = 1 << 20):
    """Unstructured points with wide deltas: many escapes, large symbols, negative coordinates."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-spread, spread, n, dtype=np.int64).astype(np.int32)
    y = rng.integers(-spread, spread, n, dtype=np.int64).astype(np.int32)
    z = (rng.normal(0, spread / 64, n)).astype(np.int32)
    c = rng.integers(0, 1 << 24, n, dtype=np.int64).astype(np.uint32)
    las = P.LasInfo()
    for k in range(3):
        las.scale[k] = 0.001
        las.offset[k] = 100.0
        las.min[k] = 100.0 - spread * 0.001
        las.max[k] = 100.0 + spread * 0.001
    return x, y, z, c, las
