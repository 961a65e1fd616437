#!/usr/bin/env python3
"""profiles/pmc_traffic_latest.json from rocprofv3 PMC passes (tools/profile_gpu.sh): HBM bytes per k_render launch, per
decode variant, tagged with the kernel version and the workload they were measured on (bench.py quotes them only on a match).

    python tools/update_traffic.py KERNEL_VERSION point_windows=gpurun_out/prof_A words=gpurun_out/prof_B [--points N ...]

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced reads as
64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section); TCC_MISS_sum x 128 B is kept beside it as a cross-check."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render" not in r.get("Kernel_Name", ""):
                continue
            a = acc.setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"] or 0); a[1] += 1
    return {k: s / n for k, (s, n) in acc.items()}


def main():
    ver = sys.argv[1]
    out = {"kernel_version": ver, "points": 100000000, "method": "basic", "width": 1920, "lod": 100, "cull": 0, "camera": "overview",
           "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM "
                         "section); WRITE_SIZE as is; both in KB; separate --pmc passes",
           "hbm_bytes_per_launch": {}, "raw": {}}
    for arg in sys.argv[2:]:
        name, d = arg.split("=", 1)
        c = counters(d)
        fetch, write = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        out["hbm_bytes_per_launch"][name] = int(round(fetch * 1024 * 2 + write * 1024))
        out["raw"][name] = {"source": os.path.relpath(d, ROOT), "fetch_size_kb": round(fetch, 1), "write_size_kb": round(write, 1),
                            "tcc_miss_x128": int(c.get("TCC_MISS_sum", 0) * 128), "tcc_atomic": int(c.get("TCC_ATOMIC_sum", 0))}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json"), "w"), indent=1)
    print(json.dumps(out["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
