#!/bin/bash
# usage: tools/exp/ab_env.sh "bench args" "ENV=VAL ..." "ENV=VAL ..." ...   A/B of environment settings on one box (two rounds, interleaved)
ARGS=$1; shift
for round in 1 2; do
for E in "$@"; do
  env $E timeout -k 10 300 python bench.py $ARGS --steps 100 --warmup 5 --no-cpu-baseline --no-variants --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$E', '|', '$ARGS', '|', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
done
