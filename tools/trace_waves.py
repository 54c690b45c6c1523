#!/usr/bin/env python3
"""experiment: per-wave lifetimes of the fast kernel (needs a -DVR_EXP_TRACE build via VR_CORE_LIB)"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vra = importlib.import_module("volume-renderer_amd"); R = vra.renderer
layout = sys.argv[1] if len(sys.argv) > 1 else "linear"
r = vra.RendererCore(0); r.setup((1920, 1080)); r.loadShader("x.cs"); r.setQuirks(0)
r.setLayout(1 if layout == "bricked" else 0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,)*3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004)
for _ in range(3): r.render()
_, spp = r.countSamples(per_pixel=True)
t = spp.ravel()[: 32768 * 4].reshape(-1, 4)
ok = t[:, 1] != 0
t = t[ok]
t0 = t[:, 0].astype(np.int64); t1 = t[:, 1].astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base
dur = (t1 - t0) / 100.0   # us (100 MHz)
print("waves traced", len(t), "kernel span us", t1.max() / 100.0)
print("duration us percentiles", np.percentile(dur, [0, 25, 50, 75, 90, 99, 100]))
print("start-time us percentiles", np.percentile(t0 / 100.0, [0, 25, 50, 75, 90, 99, 100]))
long = dur > 20
print("long waves", long.sum(), "their start percentiles", np.percentile(t0[long] / 100.0, [0, 25, 50, 75, 90, 99, 100]),
      "end", np.percentile(t1[long] / 100.0, [0, 25, 50, 75, 90, 99, 100]))
# concurrency over time
T = int(t1.max()) + 1
ev = np.zeros(T + 1, dtype=np.int64); np.add.at(ev, t0[long], 1); np.add.at(ev, t1[long], -1)
conc = np.cumsum(ev)
for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
    print(f"long waves resident at {frac:.0%} of kernel:", conc[int(T * frac)])
xcc = t[:, 3] & 0xF
print("waves per XCC", np.bincount(xcc, minlength=8), "long per XCC", np.bincount(xcc[long], minlength=8))
for x in range(8):
    m = long & (xcc == x)
    print(f"xcc {x}: last long end {t1[m].max()/100.0:.1f} us")
