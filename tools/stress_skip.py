#!/usr/bin/env python3
"""randomised campaign of exact empty-space skipping (tests/test_tslab_skip_gpu.py: run_random_skip_trials): every frame with
skipping against the frame without it and against the CPU oracle, per-pixel counts included.  tools/stress_skip.py [trials] [seed]"""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
vra = importlib.import_module("volume-renderer_amd")
import oracle
import test_tslab_skip_gpu as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
frames = T.run_random_skip_trials(vra, oracle, seed, n, log=lambda m: print(f"[{time.time() - t0:6.1f}s] {m}", flush=True))
print(f"skip campaign seed {seed}: {n} trials, {frames} frames, all bit-exact with and without skipping and against the oracle, identical per-pixel fetch counts ({time.time() - t0:.1f} s)")
