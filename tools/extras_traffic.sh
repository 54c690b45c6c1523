#!/bin/bash
# HBM traffic (PMC: 2 * FETCH_SIZE + WRITE_SIZE, separate passes) and VALU wave-instructions / shader cycles (SQ_INSTS_VALU,
# GRBM_GUI_ACTIVE) of the configurations bench.py reports under `extras`, written to profiles/traffic.json / profiles/valu.json under
# the extras' names (with the kernel each figure belongs to and the kernel-source hash, like the headline's entries).
#   gpurun --timeout 1800 -- tools/extras_traffic.sh
set -u
cd "$(dirname "$0")/.."; REPO=$PWD; export TMPDIR=/tmp
run() {   # key, bench args...
  KEY=$1; shift
  OUT=$REPO/gpurun_out/traffic_$KEY; mkdir -p "$OUT"
  CMD="python $REPO/bench.py --steps 6 --warmup 2 --clock-ramp-frames 120 --no-cpu-baseline --no-extras $*"
  (cd /tmp; for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
     timeout -k 5 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o run -- $CMD > "$OUT/$C.log" 2>&1
   done)
  python tools/summarize_profile.py "$OUT" "$KEY" > "$OUT/summary.txt" 2>&1
  python tools/pmc_traffic.py "$OUT" "$KEY"
  python tools/pmc_valu.py "$OUT" "$KEY"
}
if [ "${1:-all}" = "one" ]; then shift; run "$@"; exit 0; fi      # tools/extras_traffic.sh one <key> <bench args...>
modes() {   # round 6: the reference's other modes at size (MIP, the rotated views), both filters
run mip_deep --mip
run view_top_deep --view top
run view_bottom_deep --view bottom
run trilinear_mip_deep --mip --filter trilinear
run trilinear_view_top_deep --view top --filter trilinear
run trilinear_view_bottom_deep --view bottom --filter trilinear
run cfg4_mip --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004 --mip
run cfg4_mip_trilinear --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004 --mip --filter trilinear
}
if [ "${1:-all}" = "modes" ]; then modes; exit 0; fi
part1() {
run cfg1_shape --volume 256 --bytes 1 --synth sphere --width 1280 --height 720 --alpha 1.0 --window 0 255
run cfg2_shape_ert_window --dims 512 512 452 --bytes 2 --alpha 0.05 --window 1000 5095
run cfg4_grey --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004
run cfg4_tf_skip --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004 --tf --skip-empty
run trilinear_deep --filter trilinear
run offaxis_deep --pose offaxis
run headline_without_pack12 --no-pack12
run shallow_alpha1_ert --alpha 1.0
run trilinear_offaxis_deep --filter trilinear --pose offaxis
run headline_offset1000 --synth noise_ball_ct
}
part2() {
run cfg1_shape_trilinear --volume 256 --bytes 1 --synth sphere --width 1280 --height 720 --alpha 1.0 --window 0 255 --filter trilinear
run cfg2_shape_ert_window_trilinear --dims 512 512 452 --bytes 2 --alpha 0.05 --window 1000 5095 --filter trilinear
run cfg4_grey_trilinear --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004 --filter trilinear
run cfg4_tf_skip_trilinear --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004 --tf --skip-empty --filter trilinear
run cfg4_grey_skip_trilinear --volume 2048 --bytes 1 --width 3840 --height 2160 --window 8 255 --alpha 0.004 --skip-empty --filter trilinear
run cfg1_shape_trilinear_skip --volume 256 --bytes 1 --synth sphere --width 1280 --height 720 --alpha 1.0 --window 0 255 --filter trilinear --skip-empty
run cfg1_shape_skip --volume 256 --bytes 1 --synth sphere --width 1280 --height 720 --alpha 1.0 --window 0 255 --skip-empty
run cfg2_shape_ert_window_trilinear_skip --dims 512 512 452 --bytes 2 --alpha 0.05 --window 1000 5095 --filter trilinear --skip-empty
run cfg2_shape_ert_window_skip --dims 512 512 452 --bytes 2 --alpha 0.05 --window 1000 5095 --skip-empty
}
case "${1:-all}" in
  part1) part1;; part2) part2;; part3) modes;; trilinear_cfgs) part2; modes;; *) part1; part2; modes;;
esac
