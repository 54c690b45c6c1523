"""The C ABI from a plain-C translation unit (gcc, not hipcc): header is valid C, library
links and runs without Python/torch in the loop."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "c_client" / "client.c"


def build(tmp_path, vra):
    exe = tmp_path / "client"
    lib_dir = vra.LIB_PATH.parent
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(SRC), f"-L{lib_dir}", "-lvr_core",
           f"-Wl,-rpath,{lib_dir}", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_host_only(vra, tmp_path, oracle):
    exe = build(tmp_path, vra)
    out = subprocess.run([str(exe), "host"], check=True, capture_output=True, text=True).stdout
    assert "workgroups 16 16" in out and "render without device" in out and "no HIP device" in out
    c = oracle.Camera(); c.orient(0.0, 0.06, 0.06)
    eye = c.block()[16:19]
    assert f"eye {eye[0]:.6f} {eye[1]:.6f} {eye[2]:.6f}" in out
    assert "lut141 0.759000" in out


@pytest.mark.gpu
def test_c_client_renders_cfg0(vra, tmp_path, oracle):
    exe = build(tmp_path, vra)
    out = subprocess.run([str(exe), "gpu"], check=True, capture_output=True, text=True).stdout
    vol = oracle.gen_sphere_u8(64, 28)
    want, total = oracle.render(vol, oracle.OracleParams(256, 256))
    assert f"samples {total} " in out and any(k in out for k in ("raymarch_fast_kernel", "raymarch_relay_kernel"))
    assert f"centre_alpha {want[128, 128, 3]:.8f}" in out
    assert f"sum {float(np.sum(want.astype(np.float64))):.6f}" in out


@pytest.mark.gpu
def test_c_client_group_mode(vra, tmp_path, oracle):
    """vr_group_* from plain C: 4 members (on one device here), the gathered frame is config 0's frame"""
    exe = build(tmp_path, vra)
    out = subprocess.run([str(exe), "group", "4"], check=True, capture_output=True, text=True).stdout
    vol = oracle.gen_sphere_u8(64, 28)
    want, _ = oracle.render(vol, oracle.OracleParams(256, 256))
    assert "group 4 transport [hipMemcpyPeerAsync]" in out
    assert f"centre_alpha {want[128, 128, 3]:.8f}" in out
    assert f"sum {float(np.sum(want.astype(np.float64))):.6f}" in out
