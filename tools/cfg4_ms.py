#!/usr/bin/env python3
"""kernel ms of BASELINE config 4 (2048^3 u8 @ 3840x2160) in the grey / transfer-function modes with and without exact
empty-space skipping, at sustained clocks:  tools/cfg4_ms.py [N=2048] [W H] [trilinear [kernel variant]]"""
import importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
r = vra.RendererCore(0)
r.setup((W, H)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), 1, 0x9E3779B9)
r.setWindow(8, 255); r.setAlpha(0.004)
if len(sys.argv) > 4 and sys.argv[4] == "trilinear":
    r.setFilter(R.FILTER_TRILINEAR)
    r.setKernelVariant(int(sys.argv[5]) if len(sys.argv) > 5 else 0)


def ms(n=10):
    for _ in range(30):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


out = {}
for tf in (False, True):
    if tf:
        r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]])
    for skip in (False, True):
        r.setSkipEmpty(skip)
        out[("tf" if tf else "grey") + ("_skip" if skip else "")] = round(ms(), 4)
print(r.last_kernel_name, out)
