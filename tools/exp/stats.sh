#!/bin/bash
# usage: tools/exp/stats.sh VARIANT "bench args"   rocprofv3 kernel stats (average ns per kernel) of a short bench run with tools/exp/libpcr_hip_<VARIANT>.so
V=$1; ARGS=$2; OUT=$PWD/gpurun_out/stats_$V; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
PCR_HIP_LIB=$ROOT/tools/exp/libpcr_hip_$V.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py $ARGS --steps 100 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT.log 2>&1
cd $ROOT
python3 - "$OUT" "$V" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Name"] or "k_frame_turn" in r["Name"]: print(sys.argv[2], r["Name"][:50], r["Calls"], r["AverageNs"])
PY
