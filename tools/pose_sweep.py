#!/usr/bin/env python3
"""kernel ms over poses (default, off-axis, random orbits) for the kernel variants 2 = fast/plain loop, 5 = fast/pipelined,
3 = relay, 0 = what the host picks; usage: tools/pose_sweep.py [world] (one rank's 16-row stripes when world > 1)"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = vra.RendererCore(0)
r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
r.setWindow(0, 4095); r.setAlpha(0.004)
if world > 1:
    r.setRowStripes(16, 0, world)


def ms(n=12):
    for _ in range(60):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


rng = np.random.default_rng(3)
poses = [("default", None), ("offaxis", (0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7))]
poses += [(f"rand{i}", (float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2)), float(rng.uniform(-4, 4)))) for i in range(6)]
for name, pose in poses:
    r.resetCamera()
    if pose:
        r.cameraOrient(*pose)
    row = []
    for v in (2, 5, 3, 0):
        r.setKernelVariant(v)
        row.append((v, ms(), r.last_kernel_name.replace("raymarch_", "").replace("_kernel", "")))
    print(f"{name:8s} " + "  ".join(f"v{v}: {t:.4f} ({k})" for v, t, k in row), flush=True)
