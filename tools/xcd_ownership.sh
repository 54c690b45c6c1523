#!/bin/bash
# XCD ownership of the tile table on the CURRENT headline kernel (round-5 verdict item 4): single tiles dealt round-robin (ships),
# chunks of 2x1 / 2x2 / 4x2 neighbouring tiles per XCD, and one compact equal-work image region per XCD (8 strips / 4x2 grid);
# kernel ms at sustained clocks (default / off-axis / shallow) + L2 hit / miss and fabric read requests at the default and off-axis pose.
#   gpurun --timeout 2400 -- tools/xcd_ownership.sh      -> gpurun_out/xcd_ownership/table.txt
set -u
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=$REPO/gpurun_out/xcd_ownership; mkdir -p $OUT
: > $OUT/table.txt
for CFG in "single:" "chunk2x1:VR_X_CHUNK=2x1" "chunk2x2:VR_X_CHUNK=2x2" "chunk4x2:VR_X_CHUNK=4x2" "strips8:VR_X_XCD_REGIONS=1" "grid4x2:VR_X_XCD_REGIONS=2"; do
  NAME=${CFG%%:*}; ENVV=${CFG#*:}
  MS=$(env $ENVV python tools/quick_ms.py 0 2>&1 | tail -1)
  echo "$NAME  ms: $MS" >> $OUT/table.txt
  for POSE in default offaxis; do
    for G in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
      N=$(echo $G | tr ' ' '_')
      D=$OUT/${NAME}_${POSE}_$N
      (cd /tmp; env $ENVV timeout -k 5 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $D -o run -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --choices none --pose $POSE > $D.log 2>&1)
    done
    python - "$OUT" "$NAME" "$POSE" >> $OUT/table.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
out, name, pose = sys.argv[1:4]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{out}/{name}_{pose}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raymarch_fast_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
row = {}
for c, per in acc.items():
    k = max(per, key=lambda k: len(per[k]))
    v = per[k]
    row[c] = sum(v) / len(v)
hit, miss = row.get("TCC_HIT_sum", 0), row.get("TCC_MISS_sum", 0)
print(f"{name}  {pose}: TCC_HIT {hit/1e6:.2f} M  TCC_MISS {miss/1e6:.2f} M  hit rate {hit/max(hit+miss,1):.3f}  TCC_EA0_RDREQ {row.get('TCC_EA0_RDREQ_sum',0)/1e6:.2f} M = {row.get('TCC_EA0_RDREQ_sum',0)*128/1e9:.3f} GB  TCC_REQ {row.get('TCC_REQ_sum',0)/1e6:.2f} M")
PY
  done
done
cat $OUT/table.txt
