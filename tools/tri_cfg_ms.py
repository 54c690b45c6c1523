#!/usr/bin/env python3
"""TRILINEAR kernel ms of the cfg1 / cfg2 shapes (bench.py: config_extras) under kernel variants 2 (batched), 6 (LDS-staged),
7 (staged kernel, staging off), 0 (measured choice), sustained clocks"""
import importlib, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer


def ms(r, n=40):
    for _ in range(200):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


for name, size, synth, dims, b, seed, win, alpha in (("cfg1", (1280, 720), R.SYNTH_SPHERE_U8, (256, 256, 256), 1, 112, (0, 255), 1.0),
                                                     ("cfg2", (1920, 1080), R.SYNTH_NOISE_BALL, (512, 512, 452), 2, 0x9E3779B9, (1000, 5095), 0.05)):
    r = vra.RendererCore(0)
    r.setup(size); r.loadShader("x"); r.setQuirks(0)
    r.generateSynthetic(synth, dims, b, seed)
    r.setWindow(*win); r.setAlpha(alpha); r.setFilter(R.FILTER_TRILINEAR)
    out = {}
    for v in [int(x) for x in os.environ.get("VR_VARIANTS", "2,6,7,10,11,0").split(",")]:
        r.setKernelVariant(v)
        out[v] = (round(ms(r), 4), r.last_kernel_name.replace("raymarch_", ""))
    r.setFilter(R.FILTER_NEAREST); r.setKernelVariant(0)
    out["nearest"] = (round(ms(r), 4), r.last_kernel_name.replace("raymarch_", ""))
    print(name, out)
    r.release() if hasattr(r, "release") else None
