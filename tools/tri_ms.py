#!/usr/bin/env python3
"""kernel ms of the bench workload with TRILINEAR filtering (default and off-axis pose), sustained clocks"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
r = vra.RendererCore(0)
r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
r.setWindow(0, 4095); r.setAlpha(0.004); r.setFilter(R.FILTER_TRILINEAR)


def ms(n=20):
    for _ in range(40):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


out = {"default": ms()}
r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
out["offaxis"] = ms(10)
print(r.last_kernel_name, {k: round(v, 4) for k, v in out.items()})
