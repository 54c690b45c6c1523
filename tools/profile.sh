#!/bin/bash
# Profile the bench workload on the GPU box (run through gpurun):
#   tools/profile.sh <tag> [bench args...]
# 1) rocprofv3 --kernel-trace --stats  -> gpurun_out/prof_<tag>/stats/*
# 2) separate --pmc passes (never combined with sys/hip/hsa tracing): FETCH_SIZE, WRITE_SIZE,
#    L2 hit/miss, TCP (vector L1) counters -> gpurun_out/prof_<tag>/pmc_*/
# Summaries to keep are copied into profiles/ by tools/summarize_profile.py.
set -u
TAG=$1; shift
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $*"
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $CMD > "$OUT/stats.log" 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o run -- $CMD > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
python tools/summarize_profile.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
