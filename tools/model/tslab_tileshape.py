#!/usr/bin/env python3
"""what tile SHAPE does to the staged trilinear kernel's half-layer footprints (tools/model/tslab_footprint2.py's model): 8 wavefronts of
8x8 pixels arranged 4x2 (32x16, the kernel's), 2x4 (16x32), 8x1 (64x8), 1x8 (8x64); 80-byte half-brick slots, 80 KiB"""
import importlib.util, sys
from pathlib import Path
import numpy as np
spec = importlib.util.spec_from_file_location("fp2", Path(__file__).resolve().parent / "tslab_footprint2.py")
fp2 = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(fp2)
except SystemExit:
    pass
P = fp2.poses()
sel = sys.argv[1:] or ["offaxis", "ze-0.47az-1.33", "ze+0.66az-1.65"]
lim = ((80 * 1024 - 512 - 16 - 512 - 8192) // 80) // 3
for tw, th in ((32, 16), (16, 32), (64, 8), (8, 64)):
    for pn in sel:
        r = fp2.model(P[pn], tw, th, 2)
        wk = r[:, 0] / r[:, 0].sum()
        print(f"{tw}x{th} {pn:16s} tiles {len(r):5d} | bbox p50 {np.percentile(r[:,1],50):5.0f} p90 {np.percentile(r[:,1],90):5.0f} fit(work, {lim} slots) {float(np.sum(wk*(r[:,1]<=lim))):.2f}"
              f" | rows p50 {np.percentile(r[:,2],50):5.0f} fit {float(np.sum(wk*(r[:,2]<=lim))):.2f}", flush=True)
