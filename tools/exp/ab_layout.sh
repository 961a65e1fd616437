#!/bin/bash
# usage: tools/exp/ab_layout.sh "bench args" LAYOUT...   A/B of stream layouts (decode variants) on one box (two rounds, interleaved)
ARGS=$1; shift
for round in 1 2; do
for L in "$@"; do
  timeout -k 10 300 python bench.py $ARGS --layout $L --steps 100 --warmup 5 --no-cpu-baseline --no-variants --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', '|', '$ARGS', '|', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
done
