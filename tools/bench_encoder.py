#!/usr/bin/env python3
"""GPU encoder vs CPU encoder on the synthetic scene: wall time of pcr_gpu_encode_points (host arrays in,
file image out, PCIe both ways included) against pcr_encode_points on the host cores, and a byte comparison.

    python tools/bench_encoder.py [--points 100000000] [--threads 16] [--no-cpu]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 8, 16))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    import pcrhpg24_amd as P
    n, seed = args.points, 0x5EED
    x, y, z, c = P.synth_points(n, seed, 0, n)
    las = P.synth_las_info(n, seed)
    ctx = P.Context(0)
    best = 1e9
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        gpu, st = ctx.gpu_encode_points(x, y, z, c, las, morton_sort=True)
        best = min(best, time.perf_counter() - t0)
    out = {"metric": "Mpoints/s encoded (sort + Huffman + interleave + BC1, host arrays -> .huffman image)",
           "points": n, "batches": st["num_batches"], "file_bytes": st["file_bytes"],
           "gpu": {"seconds": round(best, 3), "mpoints_per_s": round(n / best / 1e6, 2)}}
    gh = hashlib.sha256(gpu.view()).hexdigest()
    if not args.no_cpu:
        t0 = time.perf_counter()
        cpu, st_cpu = P.encode_points(x, y, z, c, las, morton_sort=True, nthreads=args.threads)
        t_cpu = time.perf_counter() - t0
        out["cpu"] = {"seconds": round(t_cpu, 3), "mpoints_per_s": round(n / t_cpu / 1e6, 2), "threads": args.threads}
        out["identical"] = hashlib.sha256(cpu.view()).hexdigest() == gh and st_cpu == st
    out["sha256"] = gh
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
