#!/bin/bash
# usage: tools/exp/quick.sh VARIANT...   (runs bench with tools/exp/libpcr_hip_<VARIANT>.so, prints step/kernel ms)
for V in "$@"; do
  PCR_HIP_LIB=$PWD/tools/exp/libpcr_hip_$V.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$V', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
