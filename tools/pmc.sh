#!/bin/bash
# tools/pmc.sh <tag> "<counter group 1>" "<counter group 2>" ... -- <bench args>
# one rocprofv3 --pmc pass per group (kernel-trace only); prints per-kernel averages
set -u
TAG=$1; shift
CGRP=()
while [ "$1" != "--" ]; do CGRP+=("$1"); shift; done
shift
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
CMD="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $*"
cd /tmp
i=0
for G in "${CGRP[@]}"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/pmc_g$i" -o run -- $CMD > "$OUT/g$i.log" 2>&1
done
cd "$REPO"
python tools/summarize_profile.py "$OUT" "$TAG" 2>&1 | grep -E "raymarch_" | sed "s/void vr:://" | cut -c1-200
