#!/bin/bash
# usage (on the GPU box): tools/bench_set.sh OUTDIR   runs the round's standard bench configurations, one JSON line each
OUT=${1:-gpurun_out/r02}; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 "$@" 2>$OUT/b_$name.err | tail -1 > $OUT/b_$name.json || echo "$name FAILED"; }
run basic   python bench.py
run basic20 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants
run hqs     python bench.py --method hqs --no-cpu-baseline --no-variants
run closeup python bench.py --camera closeup --no-cpu-baseline --no-variants
run lod10   python bench.py --lod 10 --cull 1 --no-cpu-baseline --no-variants
run 4096    python bench.py --width 4096 --height 4096 --cull 1 --no-cpu-baseline --no-variants
PCR_FORCE_DIST=1 run dist1 python bench.py --no-cpu-baseline --no-variants
python3 - $OUT <<'PY'
import json, sys, glob, os
for n in ("basic", "basic20", "hqs", "closeup", "lod10", "4096", "dist1"):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], "b_%s.json" % n)).read())
        r = d["roofline"]; v = d.get("variants") or {}
        print(n, round(d["value"] / 1e3, 1), "Gpts/s step", d["ms_per_step"], "kernel", r.get("kernel_ms"), "frac", round(r["frac"], 4),
              "parity", d.get("parity_full_size"), "words_step", (v.get("words") or {}).get("ms_per_step"))
    except Exception as e:
        print(n, "unreadable:", e)
PY
