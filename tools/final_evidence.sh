#!/bin/bash
# one GPU-box pass that refreshes the judged evidence: rocprofv3 stats + PMC of the headline command, the
# HBM-traffic figure bench.py quotes (keyed by the kernel-source hash), a bench line, the GPU test suite.
#   gpurun --timeout 2400 -- tools/final_evidence.sh <tag>      then copy gpurun_out/evidence_<tag>/* into profiles/
set -u
TAG=${1:-final}
cd "$(dirname "$0")/.."
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
tools/profile.sh ${TAG}_headline > $OUT/${TAG}_headline_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/prof_${TAG}_headline 1024^3x2B_1920x1080_nearest_bricked_a0.004 > $OUT/traffic.log 2>&1
python tools/pmc_valu.py gpurun_out/prof_${TAG}_headline 1024^3x2B_1920x1080_nearest_bricked_a0.004 >> $OUT/traffic.log 2>&1
cp profiles/traffic.json $OUT/traffic.json; cp profiles/valu.json $OUT/valu.json
cp gpurun_out/prof_${TAG}_headline/stats/*kernel_stats.csv $OUT/${TAG}_headline_kernel_stats.csv 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gputests.log 2>&1
tail -3 $OUT/gputests.log
