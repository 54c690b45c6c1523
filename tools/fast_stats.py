#!/usr/bin/env python3
"""experiment (needs a stats build of the fast kernel's u16-bricked unit: make K3_TAG=_fs K3_DEFS="-DVR_EXPERIMENTS -DVR_X_FSTATS",
then VR_CORE_LIB=.../libvr_core_fs.so): per-tile shader-clock split of the headline launch -- set-up (ray, tables, classification
table) | checked head + prefix batches | checked tail -- for wavefront 0 of every tile.
  tools/fast_stats.py [default|offaxis] [variant] [alpha_scale]"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
pose = sys.argv[1] if len(sys.argv) > 1 else "default"
r = vra.RendererCore(0)
r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
r.setWindow(0, 4095); r.setAlpha(float(sys.argv[3]) if len(sys.argv) > 3 else 0.004)
if pose == "offaxis":
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
r.setKernelVariant(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
r.render()
print("kernel", r.last_kernel_name)
_, spp = r.countSamples(per_pixel=True)
m = (spp[::16, ::32] & 0x80000000) != 0
setup = (spp[::16, ::32][m] & 0x7fffffff).astype(np.float64)
march, tail, fetches = spp[::16, 1::32][m].astype(np.float64), spp[::16, 2::32][m].astype(np.float64), spp[::16, 3::32][m].astype(np.float64)
rays = fetches > 0                                          # tiles whose probe ray (lane 3 of wavefront 0) enters the volume
setup, march, tail, fetches = setup[rays], march[rays], tail[rays], fetches[rays]
tot = setup + march + tail
print(f"{int(rays.sum())} tiles with rays; per tile (wavefront 0) shader-clock ticks: set-up {setup.mean():.0f}, head + prefix batches {march.mean():.0f}, checked tail {tail.mean():.0f}")
print(f"share of the tiles' time: set-up {100 * setup.sum() / tot.sum():.2f} %, batches {100 * march.sum() / tot.sum():.2f} %, tail {100 * tail.sum() / tot.sum():.2f} %; ticks per sample of lane 3 in the batches: {march.sum() / max(fetches.sum(), 1):.1f}")
