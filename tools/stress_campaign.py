#!/usr/bin/env python3
"""Large randomised parity campaign: HIP path (C ABI) vs the CPU oracle, bit for bit, over
thousands of random volumes / cameras / modes (the trial generator of
tests/test_parity_gpu.py::run_random_trials).  usage: tools/stress_campaign.py [trials] [seed]"""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
vra = importlib.import_module("volume-renderer_amd")
import oracle  # noqa: E402  (test infrastructure: the checker)
import test_parity_gpu as T  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
n = T.run_random_trials(vra, oracle, seed, trials, extended=True, log=lambda m: print(f"[{time.time() - t0:6.1f}s] {m}", flush=True))
print(f"campaign seed {seed}: {trials} trials, {n} frames, all bit-exact with identical per-pixel fetch counts ({time.time() - t0:.1f} s)")
