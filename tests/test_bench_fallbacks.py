"""bench.py's two fall-backs for the first real multi-GPU run, on CPU.

1. `python bench.py --gpus N` (self-launch): when the torch.distributed.run child fails, ONE retry through
   --native-group, and the forwarded line says which launcher produced it.
2. Inside the ranks: the shards can travel over a SECOND process group through host memory
   (sharding.gather_frame(group=, host_staged=True)) -- what bench.py switches to when its RCCL preflight fails.
   World size 2 over gloo; the gathered frame must equal the single-rank frame bit for bit.
"""
import importlib
import json
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def _bench():
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    return importlib.import_module("bench")


class _Proc:
    def __init__(self, rc, out):
        self.returncode, self.stdout = rc, out


def test_self_launch_retries_through_the_native_group_when_the_torchrun_child_fails(capsys, monkeypatch):
    bench = _bench()
    calls = []
    native_line = json.dumps({"metric": "Msamples/sec (+ Mpixels/sec), 1024^3 uint16 @ 1920x1080", "value": 1.0, "n_gpus": 2,
                              "config": {"launcher": "native vr_group"}})

    def fake_run(cmd, **kw):
        calls.append(cmd)
        if "torch.distributed.run" in cmd:
            return _Proc(1, "Traceback ...\nRuntimeError: NCCL error in: ... unhandled system error\n")
        assert cmd[-1] == "--native-group"
        return _Proc(0, "some log line\n" + native_line + "\n")

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    rc = bench.self_launch(SimpleNamespace(gpus=2), run=fake_run)
    out = capsys.readouterr()
    assert rc == 0 and len(calls) == 2
    assert calls[1][-4:] == ["2", "--steps", "3", "--native-group"]
    lines = [ln for ln in out.out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    rec = json.loads(lines[0])
    assert "torch.distributed.run exited with code 1" in rec["config"]["launcher_fallback"]
    assert rec["config"]["launcher"] == "native vr_group"
    assert "retrying once through --native-group" in out.err


def test_self_launch_forwards_the_torchrun_line_untouched_when_it_works(capsys, monkeypatch):
    bench = _bench()
    line = json.dumps({"metric": "m", "value": 2.0, "n_gpus": 2, "config": {}})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    rc = bench.self_launch(SimpleNamespace(gpus=2), run=lambda cmd, **kw: _Proc(0, line + "\n"))
    assert rc == 0 and capsys.readouterr().out.strip() == line


def test_self_launch_reports_failure_when_both_launchers_fail(capsys, monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    rc = bench.self_launch(SimpleNamespace(gpus=2), run=lambda cmd, **kw: _Proc(3, "boom\n"))
    assert rc == 3 and not [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, root):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fallback = dist.new_group(backend="gloo")           # bench.py's host group, next to the primary one
    sys.path.insert(0, str(ROOT))
    sharding = importlib.import_module("volume-renderer_amd.sharding")
    W, H = 24, 37
    plan = sharding.plan_rows(H, world, rank, "stripes", 8)
    rows = plan.global_rows()
    local = np.zeros((plan.local_rows, W, 2), dtype=np.float32)      # (grey, alpha) shards
    for lr, g in enumerate(rows):
        if g >= 0:
            local[lr, :, 0] = g * 100 + np.arange(W)
            local[lr, :, 1] = -g
    frame = sharding.gather_frame(torch.from_numpy(local), plan, root=root, group=fallback, host_staged=True)
    assert (frame is None) == (root is not None and rank != root)
    if frame is not None:
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame.numpy())
    dist.barrier(group=fallback)
    dist.destroy_process_group()


def test_shards_travel_over_the_fallback_group(tmp_path):
    import torch.multiprocessing as mp

    for root in (0, None):
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), root), nprocs=2, join=True)
        H, W = 37, 24
        want = np.zeros((H, W, 4), dtype=np.float32)
        for g in range(H):
            want[g, :, 0] = want[g, :, 1] = want[g, :, 2] = g * 100 + np.arange(W)
            want[g, :, 3] = -g
        for rank in ((0,) if root == 0 else (0, 1)):
            assert np.array_equal(np.load(tmp_path / f"frame_{rank}.npy"), want), (root, rank)
