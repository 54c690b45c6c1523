#!/usr/bin/env python3
"""kernel time of ONE rank's shard at N = 1, 2, 4, 8 (stripes vs contiguous), single GPU:
what each rank of the multi-GPU run executes, without the collective."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vra = importlib.import_module("volume-renderer_amd"); R = vra.renderer
sharding = importlib.import_module("volume-renderer_amd.sharding")
r = vra.RendererCore(0); r.setup((1920, 1080)); r.loadShader("x.cs"); r.setQuirks(0); r.setLayout(1)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,) * 3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004)
for mode in ("stripes", "contiguous"):
    for world in (1, 2, 4, 8):
        times = []
        for rank in range(world):
            plan = sharding.plan_rows(1080, world, rank, mode, 16)
            sharding.apply_plan(r, plan)
            for _ in range(3): r.render()
            r.kernelMsTake()
            for _ in range(10): r.render()
            times.append(r.kernelMsTake() / 10)
        print(f"{mode:10s} N={world}: per-rank kernel ms min {min(times):.3f} max {max(times):.3f}  -> ideal speedup {times and (0):d}" if False else
              f"{mode:10s} N={world}: per-rank kernel ms min {min(times):.3f} max {max(times):.3f}")
