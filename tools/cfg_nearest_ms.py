#!/usr/bin/env python3
"""NEAREST kernel ms of the cfg1 / cfg2 shapes per forced kernel variant (0 measured choice, 2 fast kernel plain loop, 5 pipelined loop, 3 relay), sustained clocks"""
import importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer


def ms(r, n=60):
    for _ in range(300):
        r.render()
    r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


for name, size, synth, dims, b, seed, win, alpha in (("cfg1", (1280, 720), R.SYNTH_SPHERE_U8, (256, 256, 256), 1, 112, (0, 255), 1.0),
                                                     ("cfg2", (1920, 1080), R.SYNTH_NOISE_BALL, (512, 512, 452), 2, 0x9E3779B9, (1000, 5095), 0.05)):
    r = vra.RendererCore(0)
    r.setup(size); r.loadShader("x"); r.setQuirks(0)
    r.generateSynthetic(synth, dims, b, seed)
    r.setWindow(*win); r.setAlpha(alpha)
    out = {}
    for v in (0, 2, 5, 3, 0):
        r.setKernelVariant(v)
        out[f"v{v}" + ("b" if f"v{v}" in out else "")] = (round(ms(r), 4), r.last_kernel_name.replace("raymarch_", "").replace("_kernel", ""), r.last_launch_choice)
    print(name, out)
    r.close()
