#!/usr/bin/env python3
"""kernel ms of TRILINEAR on the LDS-staged kernel with and without exact empty-space skipping, sustained clocks:
tools/tri_skip_ms.py [N=1024] [bytes=2] [variants=6,9] -- window raised above the noise (the corners outside the ball are empty),
and the full window with skipping on (nothing is empty: what the SKIP instances cost by themselves)"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "6,9").split(",")]
r = vra.RendererCore(0)
r.setup((1920, 1080) if N <= 1024 else (3840, 2160)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), b, 0x9E3779B9)
r.setAlpha(0.004); r.setFilter(R.FILTER_TRILINEAR)
vmax = 4095 if b == 2 else 255
lo = 64 if b == 2 else 8


def ms(n=10):
    for _ in range(40):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return round(r.kernelMsTake() / n, 4)


for pose in ("default", "offaxis"):
    r.resetCamera()
    if pose == "offaxis":
        r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    for v in variants:
        r.setKernelVariant(v)
        out = {}
        r.setWindow(lo, vmax)
        r.setSkipEmpty(False); out["plain"] = ms()
        r.setSkipEmpty(True); out["skip"] = ms()
        r.setWindow(0, vmax)
        r.setSkipEmpty(False); out["plain_fullwindow"] = ms()
        r.setSkipEmpty(True); out["skip_fullwindow_nothing_empty"] = ms()
        print(f"{N}^3 x {b} B {pose} variant {v} {r.last_kernel_name}: {out}", flush=True)
