#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, build-container only: the transfer-function spline of the product (vr_get_transfer_lut,
csrc/volume_io.cpp buildSplineLUT) and of the oracle (vro_spline_tf) against the reference's own
src/CubicSpline.cpp, compiled VERBATIM where it lies (with its own include/CubicSpline.h) against a 20-line
glm::vec4 stand-in (oracle/glm_stub: component-wise fp32 arithmetic only).  Output goes to oracle/_ref/
(git-ignored).  The 256 entries are the widget's clamp(getPointOnSpline(iso), 0, 1)
(src/UI/elements/AlphaControlSplineWidget.cpp:247)."""
import importlib
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
OUT = ROOT / "oracle" / "_ref"

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include "CubicSpline.h"
int main(int argc, char **argv)
{   // argv: n, then n x (iso r g b a); prints 256 x 4 floats as hex words
    const int n = std::atoi(argv[1]);
    std::vector<CubicSpline::TransferFuncControlPoint> cps;
    for (int i = 0; i < n; i++) {
        CubicSpline::TransferFuncControlPoint p;
        p.iso_value = std::atoi(argv[2 + 5 * i]);
        p.color = glm::vec4(std::strtof(argv[3 + 5 * i], 0), std::strtof(argv[4 + 5 * i], 0), std::strtof(argv[5 + 5 * i], 0), std::strtof(argv[6 + 5 * i], 0));
        cps.push_back(p);
    }
    CubicSpline s;
    s.calcCubicSpline(cps);
    for (int iso = 0; iso < 256; iso++) {
        const glm::vec4 v = s.getPointOnSpline(iso);
        const float c[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; k++) { unsigned u; __builtin_memcpy(&u, &c[k], 4); std::printf("%08x ", u); }
        std::printf("\n");
    }
    return 0;
}
'''

CASES = [
    ("default alpha knots, black->white ramp", [0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]]),
    ("two knots", [0, 255], [[0, 0, 0, 0], [1, 1, 1, 1]]),
    ("five coloured knots", [0, 40, 90, 160, 255], [[0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.1, 0.4, 0.8, 0.05], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]]),
    ("overshooting knots", [0, 10, 20, 255], [[0, 0, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0], [1, 0.5, 0.25, 1]]),
]


def main():
    if not (REF / "src" / "CubicSpline.cpp").exists():
        print("reference not present (this check runs in the build container only)")
        return 0
    OUT.mkdir(parents=True, exist_ok=True)
    drv = OUT / "spline_driver.cpp"
    exe = OUT / "spline_ref"
    drv.write_text(DRIVER)
    cmd = ["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-w", f"-I{ROOT / 'oracle' / 'glm_stub'}", f"-I{REF / 'include'}",
           str(REF / "src" / "CubicSpline.cpp"), str(drv), "-o", str(exe)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(proc.stderr[-3000:])
    sys.path.insert(0, str(ROOT))
    import oracle
    vra = importlib.import_module("volume-renderer_amd")
    worst = 0
    for name, iso, rgba in CASES:
        args = [str(len(iso))]
        for i, c in zip(iso, rgba):
            args += [str(i)] + [repr(float(np.float32(v))) for v in c]
        out = subprocess.run([str(exe)] + args, check=True, capture_output=True, text=True).stdout.split()
        ref = np.array([int(w, 16) for w in out], dtype=np.uint32).view(np.float32).reshape(256, 4)
        ref = np.minimum(np.maximum(ref, np.float32(0)), np.float32(1))          # the widget's clamp
        with vra.RendererCore(-1) as r:                                             # host-only handle: no GPU involved
            r.setTransferFunction(iso, rgba)
            product = r.getTransferLut().reshape(256, 4)
        orc = oracle.spline_tf(iso, rgba).reshape(256, 4)
        d_prod = int((product.view(np.uint32) != ref.view(np.uint32)).sum())
        d_orc = int((orc.view(np.uint32) != ref.view(np.uint32)).sum())
        worst = max(worst, d_prod, d_orc)
        print(f"{name:42s} product vs reference: {d_prod} bit-different words; oracle vs reference: {d_orc}")
    print("RESULT:", "identical tables" if worst == 0 else "DIFFERENCES FOUND")
    return 0 if worst == 0 else 1


if __name__ == "__main__":
    raise SystemExit(main())
