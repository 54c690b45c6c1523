#!/usr/bin/env python3
"""kernel ms of the bench workload for the library in VR_CORE_LIB (side-by-side experiment builds):
tools/quick_ms.py [variant] -- default pose, off-axis pose, shallow regime, at sustained clocks"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
r = vra.RendererCore(0)
r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
r.setWindow(0, 4095); r.setAlpha(0.004); r.setKernelVariant(variant)


def ms(n=40):
    for _ in range(150):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


out = {"default": ms()}
r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
out["offaxis"] = ms(20)
r.resetCamera(); r.setAlpha(1.0)
out["shallow"] = ms()
print(r.last_kernel_name, {k: round(v, 4) for k, v in out.items()})
