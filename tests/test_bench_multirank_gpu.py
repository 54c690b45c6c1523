"""bench.py's torch.distributed path on ONE GPU (two ranks sharing the device), with each of round 5's fall-backs forced:
the JSON line must still come, say which transport carried the shards, and the gathered frame must be bit-exact.
(The real RCCL-over-xGMI run needs an 8-GPU node and is the driver's.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def run_bench(env_extra, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--clock-ramp-frames", "0",
           "--volume", "128", "--width", "320", "--height", "240", "--no-extras", "--no-cpu-baseline", *args]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert proc.returncode == 0 and len(lines) == 1, (proc.returncode, proc.stdout[-1500:], proc.stderr[-2500:])
    return json.loads(lines[0])


def test_gloo_validation_path_reports_transport_and_per_rank_times():
    rec = run_bench({"VR_BENCH_BACKEND": "gloo"})
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["multi_gpu_frame_bit_exact"] is True
    assert rec["transport"].startswith("gloo") and len(rec["per_rank_kernel_ms"]) == 2
    assert rec["config"]["launcher"].startswith("torch.distributed.run")


def test_failed_preflight_switches_every_rank_to_the_host_group():
    rec = run_bench({"VR_BENCH_BACKEND": "gloo", "VR_BENCH_FAIL_PREFLIGHT": "1"})
    assert "FALLBACK" in rec["transport"] and "preflight failed" in rec["transport"]
    assert rec["multi_gpu_frame_bit_exact"] is True and rec["n_ranks_seen"] == 2


def test_failed_nccl_init_continues_on_gloo():
    rec = run_bench({"VR_BENCH_FAIL_NCCL_INIT": "1"})
    assert "FALLBACK" in rec["transport"] and "failed to initialise" in rec["transport"]
    assert rec["multi_gpu_frame_bit_exact"] is True and rec["n_ranks_seen"] == 2
