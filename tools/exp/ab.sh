#!/bin/bash
# usage: tools/exp/ab.sh "bench args" VARIANT...   A/B of tools/exp/libpcr_hip_<VARIANT>.so on one box (two rounds, interleaved)
ARGS=$1; shift
for round in 1 2; do
for V in "$@"; do
  PCR_HIP_LIB=$PWD/tools/exp/libpcr_hip_$V.so timeout -k 10 300 python bench.py $ARGS --steps 100 --warmup 5 --no-cpu-baseline --no-variants --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$V', '$ARGS', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
done
