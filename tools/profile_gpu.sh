#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the default bench workload.
# Usage: tools/profile_gpu.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
#        PROFILE_SCRIPT=tools/bench_las.py PROFILE_KERNEL=k_las_render tools/profile_gpu.sh <tag> [args of that script]   (another workload)
set -o pipefail
TAG=${1:-r01}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ -n "$PROFILE_SCRIPT" ]; then
  BENCH="$PWD/$PROFILE_SCRIPT --steps 5 --warmup 2 --no-parity $*"
  BENCH_FULL="$PWD/$PROFILE_SCRIPT --steps 200 --warmup 5 --no-parity $*"
else
  BENCH="$PWD/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline --no-variants --no-secondary $*"
  BENCH_FULL="$PWD/bench.py --no-cpu-baseline --no-variants --no-secondary $*"
fi
cd /tmp
# the stats pass runs the bench at its default length so that its k_render average can be set beside roofline.kernel_ms
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $BENCH_FULL > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum" \
           "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($PMC) failed"; tail -5 "$OUT/pmc$i.log"; }
done
cd - > /dev/null
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
