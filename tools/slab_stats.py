#!/usr/bin/env python3
"""experiment (needs a -DVR_EXPERIMENTS build: make TAG=_exp DEFS=-DVR_EXPERIMENTS, VR_CORE_LIB=.../libvr_core_exp.so):
per-tile load-plan statistics of the LDS-staged kernel on the bench workload."""
import importlib, sys, os
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
pose = sys.argv[1] if len(sys.argv) > 1 else "default"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
b = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H = (1920, 1080) if N <= 1024 else (3840, 2160)
r = vra.RendererCore(0)
r.setup((W, H)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), b, 0x9E3779B9)
r.setWindow(0, 4095 if b == 2 else 255); r.setAlpha(0.004)
if pose == "offaxis":
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
r.setKernelVariant(4)
r.render()
print("kernel", r.last_kernel_name)
_, spp = r.countSamples(per_pixel=True)
st = spp[::16, ::32]
m = (st & 0x80000000) != 0
v = st[m]
ep, cold, fb, slots = v & 255, (v >> 8) & 255, (v >> 16) & 255, (v >> 24) & 127
print(f"tiles with stats {m.sum()} of {st.size}; epochs total {ep.sum()}, cold {cold.sum()} ({cold.sum()/max(ep.sum(),1):.3f}), fallback {fb.sum()} ({fb.sum()/max(ep.sum(),1):.3f})")
print("cold per tile histogram", np.bincount(cold)[:12], " fallback per tile", np.bincount(fb)[:20], " last RA*RB", np.bincount(slots)[:130].nonzero()[0])
for _ in range(3):
    r.render()
r.kernelMsTake()
for _ in range(20):
    r.render()
print("kernel ms", r.kernelMsTake() / 20)
