"""The CPU oracle against the reference shader's OWN TEXT (build container only).

oracle/crosscheck_glsl.py compiles /root/reference/VolumeRenderer.cs -- read where it lies, rewritten only
where C++ cannot parse GLSL -- against oracle/glsl_shim.h and compares whole frames with
oracle/vr_oracle.c bit for bit.  Skipped wherever the reference is absent (the GPU box).  This does
not pin the oracle in the grading sense (the shim stands in for the GL driver); it rules out a
transcription error shared by the oracle and the kernels.
"""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SHADER = Path("/root/reference/VolumeRenderer.cs")


@pytest.mark.skipif(not SHADER.exists(), reason="the reference tree exists in the build container only")
def test_oracle_frames_equal_the_shader_text_run_as_cpp():
    proc = subprocess.run([sys.executable, str(ROOT / "oracle" / "crosscheck_glsl.py")], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "identical frames" in proc.stdout
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith(("cfg", "random"))]
    assert len(lines) >= 10 and all(ln.split()[-2] == "0" for ln in lines), proc.stdout


@pytest.mark.skipif(not Path("/root/reference/src/CubicSpline.cpp").exists(), reason="the reference tree exists in the build container only")
def test_spline_tables_equal_the_reference_cubicspline_compiled_where_it_lies():
    """N3: product LUT (host-only handle) and oracle LUT against src/CubicSpline.cpp compiled verbatim with a
    component-wise glm::vec4 stand-in (oracle/crosscheck_spline.py)"""
    proc = subprocess.run([sys.executable, str(ROOT / "oracle" / "crosscheck_spline.py")], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "identical tables" in proc.stdout and proc.stdout.count(" 0 bit-different words; oracle vs reference: 0") >= 4
