"""The staged trilinear kernel's load plan under a runtime guard (round-3 verdict: "correct by statistics only").

lib/libvr_core_chk.so is the library with vr_tslab.hip compiled under -DVR_TSLAB_CHECK: for every staged sample the kernel tests
that the bricks of its four tap pairs lie inside the rectangle their layer was planned and loaded with, in a layer that is resident
in the current phase, and reports the violations per pixel in place of the fetch count.  A worker process loads that build
(VR_CORE_LIB) and sweeps volume shapes x spacings (steps of 0.3 ... 2.5 voxels along an axis) x voxel types x cameras (orbit,
pole, inside, close, eye 6 and 15 units away, random) x every staged shape (kernel variants 6, 8, 9, 10) x the rotated views, and
the 1024^3 u16 workload at 1080p: zero violations, and the staged path must actually have run."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CHK = ROOT / "volume-renderer_amd" / "lib" / "libvr_core_chk.so"


def test_every_staged_tap_lies_in_its_layers_planned_rectangle():
    assert CHK.exists(), f"{CHK} is missing: __graft_entry__.build() makes it (make -C volume-renderer_amd TSLAB_TAG=_chk TSLAB_DEFS=-DVR_TSLAB_CHECK)"
    env = dict(os.environ, VR_CORE_LIB=str(CHK))
    proc = subprocess.run([sys.executable, str(ROOT / "tests" / "tslab_plan_guard_worker.py"), "full"], env=env, capture_output=True, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    rep = json.loads(proc.stdout.strip().splitlines()[-1])
    print(rep)
    assert rep["violations"] == 0, rep
    assert rep["frames"] > 300 and rep["staged_pixels"] > 10_000_000, rep          # the guard looked at staged samples, not at fall-backs
    assert rep["unchecked_frames"] == 0, rep


def test_the_guard_sees_a_plan_that_is_too_small():
    """negative control: with the rectangles planned 1.5 voxels short on every side (VR_TSLAB_SABOTAGE, checked build only) the guard must fire"""
    assert CHK.exists()
    env = dict(os.environ, VR_CORE_LIB=str(CHK), VR_TSLAB_SABOTAGE="1")
    proc = subprocess.run([sys.executable, str(ROOT / "tests" / "tslab_plan_guard_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    rep = json.loads(proc.stdout.strip().splitlines()[-1])
    assert rep["violations"] > 1000, rep
