#!/bin/bash
# one GPU-box pass that refreshes everything the round's documents quote:
#   gpurun --timeout 5400 -- tools/round_evidence.sh r06      then copy gpurun_out/evidence_<tag>/* into profiles/ (traffic.json / valu.json / launch_choices.bin: as they are)
# gpurun clamps a call to 60 minutes, the whole pass takes ~2.5 hours: run it in phases, each its own call --
#   tools/round_evidence.sh r06 A   seed of the launch choices, headline + TRILINEAR rocprofv3 stats and PMC
#   tools/round_evidence.sh r06 B   PMC traffic of the extras, first half        (B1 / B2 / B3: thirds, when a call is short)
#   tools/round_evidence.sh r06 C   bench lines, multi-rank stand-ins, prediction, orbit sweeps, stress campaigns, the GPU suite
# (profiles/launch_choices.bin, traffic.json and valu.json written by a phase are copied into the evidence directory: gpurun only merges gpurun_out/)
set -u
TAG=${1:-final}
PHASE=${2:-ABC}
cd "$(dirname "$0")/.."
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
has() { case "$PHASE" in *$1*) return 0;; esac; return 1; }
# a later call starts from what an earlier one left in the evidence directory
for f in launch_choices.bin traffic.json valu.json; do [ -f $OUT/$f ] && [ ! -f profiles/$f -o $OUT/$f -nt profiles/$f ] && cp $OUT/$f profiles/$f; done
if has A; then
# 0) the settled launch choices of the headline configuration, measured by THIS build on THIS device (profiles/launch_choices.bin):
#    everything below starts from it, like the driver's own bench run will
rm -f profiles/launch_choices.bin
python bench.py --no-extras --no-cpu-baseline --save-choices > $OUT/bench_choices_seed.json 2> $OUT/bench_choices_seed.err
# 1) headline: rocprofv3 stats + PMC, HBM traffic keyed by the kernel-source hash
tools/profile.sh ${TAG}_headline > $OUT/${TAG}_headline_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/prof_${TAG}_headline 1024^3x2B_1920x1080_nearest_bricked_a0.004 > $OUT/traffic.log 2>&1
python tools/pmc_valu.py gpurun_out/prof_${TAG}_headline 1024^3x2B_1920x1080_nearest_bricked_a0.004 >> $OUT/traffic.log 2>&1
cp gpurun_out/prof_${TAG}_headline/stats/*kernel_stats.csv $OUT/${TAG}_headline_kernel_stats.csv 2>/dev/null
# 2) TRILINEAR (the LDS-staged kernel): stats + PMC
tools/profile.sh ${TAG}_trilinear --filter trilinear > $OUT/${TAG}_trilinear_summary.txt 2>&1
cp gpurun_out/prof_${TAG}_trilinear/stats/*kernel_stats.csv $OUT/${TAG}_trilinear_kernel_stats.csv 2>/dev/null
tools/pmc.sh ${TAG}_tslab_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" -- --filter trilinear > $OUT/${TAG}_trilinear_sq.txt 2>&1
tools/pmc.sh ${TAG}_tslab_offaxis_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" -- --filter trilinear --pose offaxis > $OUT/${TAG}_trilinear_offaxis_sq.txt 2>&1
# ... and what every renderer of the full bench (extras, the other BASELINE configs) settles on joins the blob: the PMC passes of phase B
# and the bench lines of phase C then all start from settled choices
python bench.py --extras --save-choices > $OUT/bench_cold_extras.json 2> $OUT/bench_cold_extras.err
cp profiles/launch_choices.bin $OUT/launch_choices.bin; cp profiles/traffic.json $OUT/traffic.json; cp profiles/valu.json $OUT/valu.json
fi
# 3) HBM traffic + VALU counts of every configuration bench.py reports under `extras` (incl. the MIP / rotated-view entries of round 6)
if has B; then
case "$PHASE" in *B[123]*) PARTS_NAMED=1;; *) PARTS_NAMED=0;; esac      # "B" alone (or "ABC"): all three parts
for part in 1 2 3; do
  if [ $PARTS_NAMED = 0 ] || has B$part; then tools/extras_traffic.sh part$part >> $OUT/traffic.log 2>&1; cp profiles/traffic.json $OUT/traffic.json; cp profiles/valu.json $OUT/valu.json; fi
done
fi
if has C; then
# 4) the bench line with every extra (traffic and launch choices from the phases above), then the line the driver will see (default flags)
python bench.py --extras > $OUT/bench_extras.json 2> $OUT/bench_extras.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# 5) one-device stand-ins for the multi-GPU launchers, and the N = 1..8 prediction
python bench.py --native-group --gpus 4 --steps 50 > $OUT/bench_native_group4.json 2>&1
VR_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 50 > $OUT/bench_torch_gloo4.json 2>&1
timeout 900 python tools/scale_prediction.py > $OUT/scale_prediction.log 2>&1
cp gpurun_out/r06_scale_prediction.json $OUT/${TAG}_scale_prediction.json 2>/dev/null
timeout 600 python tools/shard_ms.py default > $OUT/${TAG}_shard_ms.txt 2>&1
timeout 600 python tools/shard_modes.py > $OUT/${TAG}_shard_modes.txt 2>&1
# 6) TRILINEAR over orbit poses per kernel variant
{ for v in 2 6 10 11 0; do echo "1024^3 x 1 B, kernel variant $v: $(timeout 300 python tools/tri_ms.py $v orbit6 1024 1 2>&1 | tail -1)"; done
  for v in 2 6 8 9 10 11 0; do echo "1024^3 x 2 B, kernel variant $v: $(timeout 300 python tools/tri_ms.py $v orbit6 1024 2 2>&1 | tail -1)"; done
  for v in 6 8 9 11 0; do echo "1024^3 x 2 B, default + off-axis (zenith 60, azimuth 45), kernel variant $v: $(timeout 300 python tools/tri_ms.py $v both 1024 2 2>&1 | tail -1)"; done
  echo "cfg1 / cfg2 shapes: $(timeout 300 python tools/tri_cfg_ms.py 2>&1 | tail -2)"; } > $OUT/${TAG}_trilinear_orbit.txt 2>&1
# 7) randomised stress campaigns (every frame against the CPU oracle) and the GPU suite
timeout 900 python tools/stress_campaign.py 60000 606 2>&1 | tail -2 > $OUT/${TAG}_stress_campaign.txt
timeout 900 python tools/stress_skip.py 1500 66 2>&1 | tail -2 > $OUT/${TAG}_stress_skip.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gputests.log 2>&1
tail -c 1500 $OUT/bench.json; echo; tail -3 $OUT/gputests.log; cat $OUT/${TAG}_stress_campaign.txt $OUT/${TAG}_stress_skip.txt
fi
