"""integration/RendererCoreShim.h -- the adapter INTEGRATION.md hands to a maintainer of the reference --
goes through a compiler and runs: integration/gui_touchpoints.cpp performs every friend access of
src/RendererGUI.cpp on it (class RendererGUI, same member names and call shapes)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def build(tmp_path, vra):
    exe = tmp_path / "gui_touchpoints"
    lib_dir = vra.LIB_PATH.parent
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'integration'}", f"-I{ROOT / 'integration' / 'stub'}",
           str(ROOT / "integration" / "gui_touchpoints.cpp"), f"-L{lib_dir}", "-lvr_core", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    return exe


def write_raw(tmp_path, with_inf):
    rng = np.random.default_rng(1)
    vol = rng.integers(0, 256, size=(16, 24, 32), dtype=np.uint8)
    raw = tmp_path / ("with_inf.raw" if with_inf else "no_inf.raw")
    vol.tofile(raw)
    if with_inf:
        Path(str(raw) + ".inf").write_text("#dimensions\n32 24 16\n#voxel-spacing\n1 1 1.5\n")
    return raw, vol


def test_shim_compiles_and_reports_errors_without_a_device(vra, tmp_path):
    exe = build(tmp_path, vra)
    raw, _ = write_raw(tmp_path, True)
    proc = subprocess.run([str(exe), str(raw), str(tmp_path / "o.png")], env={"VR_DEVICES": "-1", "PATH": "/usr/bin:/bin"}, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "workgroups 20 13" in proc.stdout and "host-only:" in proc.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0", "0,0,0"], ids=["one_handle", "vr_group_of_three"])
@pytest.mark.parametrize("with_inf", [True, False], ids=["sidecar", "raw_inf_panel"])
def test_shim_session_on_the_gpu(vra, oracle, tmp_path, with_inf, devices):
    exe = build(tmp_path, vra)
    raw, vol = write_raw(tmp_path, with_inf)
    png = tmp_path / "shot.png"
    proc = subprocess.run([str(exe), str(raw), str(png)], env={"VR_DEVICES": devices, "PATH": "/usr/bin:/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "dims 32 24 16 window [0,255] range [0,255]" in proc.stdout
    assert png.exists() and png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    if not with_inf:
        assert Path(str(raw) + ".inf").exists()               # the sidecar is written (src/RendererCore.cpp:304-317)
    # the last frame of the session: orbit (0.06, 0.06) then one zoom step, window [10,235], alpha 0.25
    c = oracle.Camera(); c.orient(0.0, 0.06, 0.06); c.orient(1.0, 0.0, 0.0)
    p = oracle.OracleParams(320, 200, cam=c.block(), alpha_scale=0.25, voxel_size=(1.0, 1.0, 1.5), min_val=10, max_val=235, threads=4)
    want, _ = oracle.render(vol, p)
    assert f"frame sum {float(np.sum(want.astype(np.float64))):.6f}" in proc.stdout
    assert f"centre alpha {want[100, 160, 3]:.8f}" in proc.stdout
