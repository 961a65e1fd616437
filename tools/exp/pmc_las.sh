#!/bin/bash
# usage: tools/exp/pmc_las.sh TAG "COUNTER COUNTER ..." [bench_las args]   per-dispatch averages for k_las_render
TAG=$1; PMC=$2; shift; shift; OUT=$PWD/gpurun_out/pmc_las_$TAG; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -- python3 $ROOT/tools/bench_las.py --steps 3 --warmup 1 --no-parity "$@" > $OUT.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(lambda:[0,0])
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_las_render" in r["Kernel_Name"]:
            a=acc[r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,(s,n) in sorted(acc.items()): print(k, s/n)
PY
