#!/usr/bin/env python3
"""kernel ms of one rank's shard over BASELINE's config shapes x shard counts x poses, for the kernel variants
2 = fast kernel (plain loop), 5 = fast kernel (pipelined loop), 3 = relay kernel, 0 = what the host's work model picks.
Flags the cells in which the automatic choice is more than 3 % behind the best forced variant.
usage: tools/config_sweep.py [cfg1,cfg2,cfg3,cfg4,cfg4tf] [worlds e.g. 1,2,4,8] [number of poses <= 8]"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
which = (sys.argv[1] if len(sys.argv) > 1 else "cfg1,cfg2,cfg3,cfg4").split(",")
worlds = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
nposes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
VARIANTS = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (2, 5, 3, 0)   # e.g. 2,5,3,4,0 adds the LDS-staged NEAREST kernel
CFG = {  # name: (dims, bytes, (W, H), window, alpha, tf + skip)
    "cfg1": ((256, 256, 256), 1, (1280, 720), (0, 255), 1.0, False),
    "cfg2": ((512, 512, 452), 2, (1920, 1080), (1000, 5095), 0.05, False),
    "cfg3": ((1024, 1024, 1024), 2, (1920, 1080), (0, 4095), 0.004, False),
    "cfg4": ((2048, 2048, 2048), 1, (3840, 2160), (8, 255), 0.004, False),
    "cfg4tf": ((2048, 2048, 2048), 1, (3840, 2160), (8, 255), 0.004, True),
}
rng = np.random.default_rng(3)
poses = [("default", None), ("offaxis", (0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7))]
poses += [(f"rand{i}", (0.0, float(rng.uniform(-2, 2)), float(rng.uniform(-4, 4)))) for i in range(6)]
poses = poses[:nposes]
worst = []
for name in which:
    dims, b, (W, H), win, alpha, tf = CFG[name]
    r = vra.RendererCore(0)
    r.setup((W, H)); r.loadShader("x"); r.setQuirks(0)
    if name == "cfg1":
        r.generateSynthetic(R.SYNTH_SPHERE_U8, dims, 1, 112)
    else:
        r.generateSynthetic(R.SYNTH_NOISE_BALL, dims, b, 0x9E3779B9)
    r.setWindow(*win); r.setAlpha(alpha)
    if tf:
        r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]])
        r.setSkipEmpty(True)
    reps = 6 if dims[0] >= 2048 else 12

    def ms():
        for _ in range(30):
            r.renderAsync()
        r.synchronize(); r.render(); r.kernelMsTake()
        for _ in range(reps):
            r.render()
        return r.kernelMsTake() / reps

    for world in worlds:
        r.setRowStripes(16 if world > 1 else 1, 0, world)
        for pname, pose in poses:
            r.resetCamera()
            if pose:
                r.cameraOrient(*pose)
            row = {}
            for v in VARIANTS:
                r.setKernelVariant(v)
                row[v] = (ms(), r.last_kernel_name.replace("raymarch_", "").replace("_kernel", ""))
            best = min(row[v][0] for v in VARIANTS if v != 0)
            gap = row[0][0] / best - 1.0
            flag = "  <-- auto %.1f %% behind" % (100 * gap) if gap > 0.03 else ""
            if gap > 0.03:
                worst.append((name, world, pname, gap))
            print(f"{name:7s} N={world} {pname:8s} " + "  ".join(f"v{v}: {t:.4f} ({k})" for v, (t, k) in row.items()) + flag, flush=True)
    r.close()
print(f"cells with the automatic choice > 3 % behind the best forced variant: {len(worst)}", worst)
