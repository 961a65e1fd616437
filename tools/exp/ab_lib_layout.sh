#!/bin/bash
# usage: tools/exp/ab_lib_layout.sh "bench args" VARIANT:LAYOUT...   A/B of tools/exp/libpcr_hip_<VARIANT>.so drawing a stream of the given layout (one box, two rounds)
ARGS=$1; shift
for round in 1 2; do
for VL in "$@"; do
  V=${VL%%:*}; L=${VL##*:}
  PCR_HIP_LIB=$PWD/tools/exp/libpcr_hip_$V.so timeout -k 10 300 python bench.py $ARGS --layout $L --steps 100 --warmup 5 --no-cpu-baseline --no-variants --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$VL', '|', '$ARGS', '|', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
done
