#!/usr/bin/env python3
"""kernel ms of one rank's shard of the cfg3 frame (cyclic 16-row stripes) at N = 1, 2, 4, 8 for the kernel variants:
2 = fast kernel, 5 = fast kernel with the pipelined batch loop, 3 = relay kernel, 0 = what the host picks.
usage: tools/shard_ms.py [pose: default|offaxis] [alpha]"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
pose = sys.argv[1] if len(sys.argv) > 1 else "default"
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
r = vra.RendererCore(0)
r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
r.setWindow(0, 4095); r.setAlpha(alpha)
if pose == "offaxis":
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)


def ms(n=30):
    for _ in range(200):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


for world in (1, 2, 4, 8):
    r.setRowStripes(16 if world > 1 else 1, 0, world)
    row = {}
    for v in (2, 5, 3, 0):
        r.setKernelVariant(v)
        t = ms()
        row[v] = (round(t, 4), r.last_kernel_name.replace("raymarch_", "").replace("_kernel", ""))
    print(f"N={world}: " + "  ".join(f"v{v}: {t:.4f} ({k})" for v, (t, k) in row.items()), flush=True)
