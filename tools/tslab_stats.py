#!/usr/bin/env python3
"""experiment (needs a stats build of vr_tslab.hip: make TSLAB_TAG=_st TSLAB_DEFS="-DVR_EXPERIMENTS -DVR_X_STATS", then
VR_CORE_LIB=.../libvr_core_st.so): per-tile load-plan statistics of the LDS-staged TRILINEAR kernel on the bench workload.
  tools/tslab_stats.py [default|offaxis|zenith,azimuth] [N] [bytes] [alpha_scale] [kernel variant: 6, 8, 9] [window low: > 0 switches empty-space skipping on]"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
pose = sys.argv[1] if len(sys.argv) > 1 else "default"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
b = int(sys.argv[3]) if len(sys.argv) > 3 else 2
import os
W, H = (1920, 1080) if N <= 1024 else (3840, 2160)
PRESET = os.environ.get("VR_STATS_PRESET", "")          # "cfg1": bench.py's cfg1 shape (256^3 u8 sphere, 1280x720, alpha 1)
if PRESET == "cfg1":
    W, H, N, b = 1280, 720, 256, 1
r = vra.RendererCore(0)
r.setup((W, H)); r.loadShader("x"); r.setQuirks(0)
if PRESET == "cfg1":
    r.generateSynthetic(R.SYNTH_SPHERE_U8, (N, N, N), 1, 112)
else:
    r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), b, 0x9E3779B9)
r.setWindow(0, 4095 if b == 2 else 255); r.setAlpha(float(sys.argv[4]) if len(sys.argv) > 4 else 0.004); r.setFilter(R.FILTER_TRILINEAR)
if pose == "offaxis":
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
elif "," in pose:                       # "zenith,azimuth" as passed to cameraOrient
    r.cameraOrient(0.0, *[float(v) for v in pose.split(",")])
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 6
if len(sys.argv) > 6 and int(sys.argv[6]) > 0:
    r.setWindow(int(sys.argv[6]), 4095 if b == 2 else 255); r.setSkipEmpty(True)
TH, TWP = (32, 16) if variant == 9 else ((16, 16) if variant == 11 else (16, 32))              # rows / columns per tile
r.setKernelVariant(variant)
r.render()
print("kernel", r.last_kernel_name)
_, spp = r.countSamples(per_pixel=True)
st = spp[::TH, ::TWP]
m = (st & 0x80000000) != 0
v = st[m]
staged, rz, half, slots, phases = v & 1, (v >> 4) & 7, (v >> 7) & 1, (v >> 8) & 255, (v >> 16) & 4095
print("why not staged (1 corner rays disagree on / graze the major axis, 2 layer above the DMA piece limit, 3 ring not three layers deep, 4 layer beyond 64 KiB):", np.bincount((v >> 1) & 7)[:5])
print(f"tiles with a prefix {m.sum()} of {st.size}; staged {int(staged.sum())} ({staged.mean():.3f}), of them in half layers {int((staged & half).sum())}; RZ histogram {np.bincount(rz)[:6]}; "
      f"RA*RB of staged tiles min/mean/max {slots[staged == 1].min() if staged.any() else 0}/{slots[staged == 1].mean() if staged.any() else 0:.1f}/{slots[staged == 1].max() if staged.any() else 0}; "
      f"RA*RB of unstaged tiles: percentiles 10/50/90/100 {np.percentile(slots[staged == 0], [10, 50, 90, 100]) if (staged == 0).any() else None} (255 = 255 or more, or not computed); phases mean {phases.mean():.1f} max {phases.max()}")
clk, wall, iters, samp = spp[::TH, 1::TWP][m].astype(np.float64), spp[::TH, 2::TWP][m].astype(np.float64), spp[::TH, 3::TWP][m].astype(np.float64), spp[::TH, 4::TWP][m].astype(np.float64)
setup = spp[::TH, 5::TWP][m].astype(np.float64)
skipw = spp[::TH, 6::TWP][m]
skipped, runs = (skipw & 0xffff).astype(np.float64), (skipw >> 16).astype(np.float64)
print(f"empty-space skipping: {skipped.sum():.0f} of {phases.sum():.0f} phases crossed in {runs.sum():.0f} empty runs ({skipped.sum() / max(phases.sum(), 1):.3f}); tiles with no sampled phase {(skipped >= phases).sum()}")
ok = wall > 0
print(f"set-up before the march (ray, checked head, load plan, tables): mean {setup[ok].mean():.0f} ticks = {100 * setup[ok].sum() / (setup[ok].sum() + clk[ok].sum()):.1f} % of the tiles' time")
for flag, name in ((1, "staged"), (0, "not staged")):
    sel = ok & (staged == flag)
    if sel.any():
        print(f"  {name}: {int(sel.sum())} tiles, march wall time per tile us: percentiles 10/50/90/100 {np.round(np.percentile(wall[sel] / 100, [10, 50, 90, 100]), 1)}, sum {wall[sel].sum() / 100 / 1e3:.1f} ms; phases mean {phases[sel].mean():.0f}")
print(f"staged loop per tile: {clk[ok].mean():.0f} shader-clock ticks in {wall[ok].mean() / 100:.1f} us -> {np.median(clk[ok] / wall[ok]) * 100 / 1e3:.3f} GHz (median over tiles); wavefront 0: {iters[ok].sum():.0f} iterations for {samp[ok].sum():.0f} lane-0 samples ({iters[ok].sum() / max(samp[ok].sum(), 1):.3f} iterations per sample of lane 3)")
for _ in range(150):
    r.renderAsync()
r.synchronize(); r.render(); r.kernelMsTake()
for _ in range(20):
    r.render()
print("kernel ms", r.kernelMsTake() / 20)
