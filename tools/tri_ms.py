#!/usr/bin/env python3
"""kernel ms of the bench workload with TRILINEAR filtering (default and off-axis pose), sustained clocks"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
r = vra.RendererCore(0)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
r.setup((1920, 1080) if N <= 1024 else (3840, 2160)); r.loadShader("x"); r.setQuirks(0)
b = int(sys.argv[4]) if len(sys.argv) > 4 else 2
r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), b, 0x9E3779B9)
r.setWindow(0, 4095 if b == 2 else 255); r.setAlpha(0.004); r.setFilter(R.FILTER_TRILINEAR)
r.setKernelVariant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
poses = sys.argv[2] if len(sys.argv) > 2 else "both"


def ms(n=20):
    for _ in range(80):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


out = {"default": ms()}
if poses == "both":
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    out["offaxis"] = ms(10)
elif poses.startswith("orbit"):
    rng = np.random.default_rng(7)
    for k in range(int(poses[5:] or 6)):
        r.resetCamera()
        ze, az = float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-3, 3))
        r.cameraOrient(0.0, ze, az)
        out[f"ze{ze:+.2f}az{az:+.2f}"] = ms(10)
print(r.last_kernel_name, {k: round(v, 4) for k, v in out.items()})
