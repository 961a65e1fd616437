#!/bin/bash
# usage (on the GPU box): tools/bench_set.sh OUTDIR   runs the round's standard bench configurations, one JSON line each
OUT=${1:-gpurun_out/r03}; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 500 "$@" 2>$OUT/b_$name.err | tail -1 > $OUT/b_$name.json || echo "$name FAILED"; }
run basic   python bench.py
run basic20 python bench.py --steps 20 --warmup 5
run closeup python bench.py --camera closeup --no-cpu-baseline --no-variants --no-secondary
PCR_FORCE_DIST=1 run dist1_torch python bench.py --no-cpu-baseline --no-variants --no-secondary
PCR_FORCE_DIST=1 run dist1_rccl python bench.py --transport rccl --no-cpu-baseline --no-variants --no-secondary
PCR_FORCE_DIST=1 run dist1_rccl_sliced python bench.py --transport rccl --merge sliced --no-cpu-baseline --no-variants --no-secondary
PCR_FORCE_DIST=1 run dist1_auto python bench.py --transport auto --no-cpu-baseline --no-variants --no-secondary
run 2e9 python bench.py --points 2000000000 --no-variants --no-secondary
python3 - $OUT <<'PY'
import json, sys, glob, os
for n in ("basic", "basic20", "closeup", "dist1_torch", "dist1_rccl", "dist1_rccl_sliced", "dist1_auto", "2e9"):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], "b_%s.json" % n)).read())
        r = d["roofline"]; v = d.get("variants") or {}
        print(n, round(d["value"] / 1e3, 1), "Gpts/s step", d["ms_per_step"], "kernel", r.get("kernel_ms"), "frac", round(r["frac"], 4),
              "parity", d.get("parity_full_size"), "words_step", (v.get("words") or {}).get("ms_per_step"), d["config"].get("transport"), d["config"].get("transport_check"))
    except Exception as e:
        print(n, "unreadable:", e)
PY
