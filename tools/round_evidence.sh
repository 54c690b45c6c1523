#!/bin/bash
# one GPU-box pass that refreshes everything the round's documents quote:
#   gpurun --timeout 3000 -- tools/round_evidence.sh r04      then copy gpurun_out/evidence_<tag>/* into profiles/ (traffic.json / valu.json: into profiles/ as they are)
set -u
TAG=${1:-final}
cd "$(dirname "$0")/.."
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
# headline: rocprofv3 stats + PMC, HBM traffic keyed by the kernel-source hash
tools/profile.sh ${TAG}_headline > $OUT/${TAG}_headline_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/prof_${TAG}_headline 1024^3x2B_1920x1080_nearest_bricked_a0.004 > $OUT/traffic.log 2>&1
python tools/pmc_valu.py gpurun_out/prof_${TAG}_headline 1024^3x2B_1920x1080_nearest_bricked_a0.004 >> $OUT/traffic.log 2>&1
cp gpurun_out/prof_${TAG}_headline/stats/*kernel_stats.csv $OUT/${TAG}_headline_kernel_stats.csv 2>/dev/null
# TRILINEAR (the LDS-staged kernel): stats + PMC
tools/profile.sh ${TAG}_trilinear --filter trilinear > $OUT/${TAG}_trilinear_summary.txt 2>&1
cp gpurun_out/prof_${TAG}_trilinear/stats/*kernel_stats.csv $OUT/${TAG}_trilinear_kernel_stats.csv 2>/dev/null
tools/pmc.sh ${TAG}_tslab_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" "GRBM_GUI_ACTIVE" -- --filter trilinear > $OUT/${TAG}_trilinear_sq.txt 2>&1
tools/pmc.sh ${TAG}_tslab_offaxis_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" -- --filter trilinear --pose offaxis > $OUT/${TAG}_trilinear_offaxis_sq.txt 2>&1
# HBM traffic of every configuration bench.py reports under `extras`
tools/extras_traffic.sh >> $OUT/traffic.log 2>&1
cp profiles/traffic.json $OUT/traffic.json; cp profiles/valu.json $OUT/valu.json
# the bench line (with extras), the native group on shared-device members, the torch path for comparison
python bench.py --extras > $OUT/bench.json 2> $OUT/bench.err
python bench.py --native-group --gpus 4 --steps 50 > $OUT/bench_native_group4.json 2>&1
VR_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 50 > $OUT/bench_torch_gloo4.json 2>&1
# TRILINEAR over orbit poses: batched kernel (2), LDS-staged kernel whole layers (6), layer thickness per tile on 32x16- (8) and 16x32-pixel tiles with rows (9), three workgroups per CU (10), the measured choice (0); cfg4 (6 vs generic)
{ for v in 2 6 9 10 0; do echo "1024^3 x 1 B, kernel variant $v: $(timeout 300 python tools/tri_ms.py $v orbit6 1024 1 2>&1 | tail -1)"; done
  for v in 2 6 8 9 10 0; do echo "1024^3 x 2 B, kernel variant $v: $(timeout 300 python tools/tri_ms.py $v orbit6 1024 2 2>&1 | tail -1)"; done
  for v in 2 6 8 9 0; do echo "1024^3 x 2 B, default + off-axis (zenith 60, azimuth 45), kernel variant $v: $(timeout 300 python tools/tri_ms.py $v both 1024 2 2>&1 | tail -1)"; done
  for v in 6 10 1; do echo "2048^3 x 1 B @3840x2160, kernel variant $v: $(timeout 600 python tools/tri_ms.py $v orbit4 2048 1 2>&1 | tail -1)"; done; } > $OUT/${TAG}_trilinear_orbit.txt 2>&1
# randomised stress campaign (tests/test_parity_gpu.py: run_random_trials; every frame against the CPU oracle)
timeout 900 python tools/stress_campaign.py 60000 304 2>&1 | tail -2 > $OUT/${TAG}_stress_campaign.txt
tail -c 2500 $OUT/bench.json
timeout 1500 python -m pytest tests -q -m gpu > $OUT/gputests.log 2>&1
tail -3 $OUT/gputests.log
