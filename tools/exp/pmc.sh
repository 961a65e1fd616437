#!/bin/bash
# usage: tools/exp/pmc.sh VARIANT "COUNTER COUNTER ..."   prints per-dispatch averages for k_render
V=$1; PMC=$2; OUT=$PWD/gpurun_out/pmc_$V; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
PCR_HIP_LIB=$ROOT/tools/exp/libpcr_hip_$V.so rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(lambda:[0,0])
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            a=acc[r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,(s,n) in sorted(acc.items()): print(k, s/n)
PY
