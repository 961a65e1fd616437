#!/bin/bash
# usage: tools/exp/ab_las.sh VARIANT...   A/B of tools/exp/libpcr_hip_<VARIANT>.so on the 10-10-10 path's three workloads (one box, two rounds, interleaved)
for round in 1 2; do
for W in "--order tiles" "--order strips" "--order tiles --camera closeup"; do
for V in "$@"; do
  PCR_HIP_LIB=$PWD/tools/exp/libpcr_hip_$V.so timeout -k 10 300 python tools/bench_las.py $W --steps 50 --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$V', '|', '$W', '|', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
done
done
