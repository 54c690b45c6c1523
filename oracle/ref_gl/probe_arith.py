#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY.

Measures how Mesa 23.2.1 llvmpipe rounds the GLSL operations the reference shader's
position arithmetic is made of (VolumeRenderer.cs:86-236), by running one-line compute
shaders on random operands through oracle/_ref/libref_gl.so and comparing the bit patterns
with candidate restatements.  The winners are what oracle/vr_oracle.c implements as
VRO_ARITH_MESA; the table this prints is quoted in DESIGN.md.

Run:  python oracle/ref_gl/probe_arith.py
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import ref_gl  # noqa: E402

f32 = np.float32


def load():
    return ref_gl.load()


HEAD = """#version 430 core
layout(local_size_x = 64) in;
layout(std430, binding = 0) readonly buffer In { vec4 a[]; };
layout(std430, binding = 1) writeonly buffer Out { vec4 r[]; };
void main() { uint i = gl_GlobalInvocationID.x; vec4 x = a[6u*i], y = a[6u*i+1u], z = a[6u*i+2u], u = a[6u*i+3u], v = a[6u*i+4u];
"""


def run(lib, body, inp):
    n = inp.shape[0]
    assert n % 64 == 0 and inp.shape[1:] == (6, 4)
    out = np.zeros((n, 4), dtype=f32)
    src = (HEAD + body + "\n}\n").encode()
    rc = lib.refgl_compute(src, inp.ctypes.data, inp.nbytes, out.ctypes.data, out.nbytes, n // 64)
    if rc != 0:
        raise RuntimeError(lib.refgl_last_error().decode())
    return out


def same(a, b):
    return int((a.view(np.uint32) == b.view(np.uint32)).all(axis=-1).sum()) if a.ndim > 1 else int(
        (a.view(np.uint32) == b.view(np.uint32)).sum())


def main():
    lib = load()
    rng = np.random.default_rng(12345)
    n = 1 << 16
    inp = rng.uniform(-2.0, 2.0, size=(n, 6, 4)).astype(f32)
    inp[:, :3, 3] = rng.uniform(0.1, 3.0, size=(n, 3)).astype(f32)
    x, y, z, u, v = inp[:, 0], inp[:, 1], inp[:, 2], inp[:, 3], inp[:, 4]
    one = f32(1.0)
    rows = []

    def report(name, got, cands):
        line = f"{name:34s}"
        for cname, c in cands.items():
            line += f" | {cname}: {same(got, c)}/{n}"
        rows.append(line)
        print(line)

    # 1. division
    got = run(lib, "r[i] = x / y;", inp)
    report("x / y", got, {"round(x/y)": x / y, "x * round(1/y)": x * (one / y)})
    # 2. a*b+c contraction
    got = run(lib, "r[i] = x * y + z;", inp)
    fma = (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(f32)
    report("x * y + z", got, {"mul then add": x * y + z, "fused": fma})
    got = run(lib, "vec4 d = z; d += x * (1.0 - d.a); r[i] = d;", inp)
    om = one - z[:, 3:4]
    fma2 = (x.astype(np.float64) * om.astype(np.float64) + z.astype(np.float64)).astype(f32)
    report("d += x * (1 - d.a)", got, {"mul then add": z + x * om, "fused": fma2})
    # 3. dot / length / normalize of vec3 and vec4
    x3 = x[:, :3]
    fwd3 = (x3[:, 0] * x3[:, 0] + x3[:, 1] * x3[:, 1]) + x3[:, 2] * x3[:, 2]
    rev3 = (x3[:, 2] * x3[:, 2] + x3[:, 1] * x3[:, 1]) + x3[:, 0] * x3[:, 0]
    got = run(lib, "r[i] = vec4(dot(x.xyz, x.xyz));", inp)[:, 0]
    report("dot(v3,v3)", got, {"(xx+yy)+zz": fwd3, "(zz+yy)+xx": rev3})
    got = run(lib, "r[i] = vec4(length(x.xyz));", inp)[:, 0]
    report("length(v3)", got, {"sqrt fwd": np.sqrt(fwd3), "sqrt rev": np.sqrt(rev3)})
    fwd4 = ((x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]) + x[:, 2] * x[:, 2]) + x[:, 3] * x[:, 3]
    rev4 = ((x[:, 3] * x[:, 3] + x[:, 2] * x[:, 2]) + x[:, 1] * x[:, 1]) + x[:, 0] * x[:, 0]
    got = run(lib, "r[i] = vec4(dot(x, x));", inp)[:, 0]
    report("dot(v4,v4)", got, {"fwd": fwd4, "rev": rev4})
    got = run(lib, "r[i] = normalize(x);", inp)
    cands = {}
    for dn, d in (("fwd", fwd4), ("rev", rev4)):
        s = np.sqrt(d)[:, None]
        cands[f"x/sqrt({dn})"] = x / s
        cands[f"x*(1/sqrt({dn}))"] = x * (one / s)
    report("normalize(v4)", got, cands)
    # the shader's case: w == 0
    xin = inp.copy()
    xin[:, 0, 3] = 0.0
    xw = xin[:, 0]
    got = run(lib, "r[i] = normalize(x);", xin)
    d3f = ((xw[:, 0] * xw[:, 0] + xw[:, 1] * xw[:, 1]) + xw[:, 2] * xw[:, 2]) + xw[:, 3] * xw[:, 3]
    d3r = ((xw[:, 3] * xw[:, 3] + xw[:, 2] * xw[:, 2]) + xw[:, 1] * xw[:, 1]) + xw[:, 0] * xw[:, 0]
    cands = {}
    for dn, d in (("fwd", d3f), ("rev", d3r)):
        s = np.sqrt(d)[:, None]
        cands[f"x/sqrt({dn})"] = xw / s
        cands[f"x*(1/sqrt({dn}))"] = xw * (one / s)
    report("normalize(vec4(v3,0))", got, cands)
    # 4. mat4 * vec4, five independent operands (shared operands let the compiler's algebraic pass factor terms)
    got = run(lib, "mat4 m = mat4(x, y, z, u); r[i] = m * v;", inp)
    fwd = ((x * v[:, 0:1] + y * v[:, 1:2]) + z * v[:, 2:3]) + u * v[:, 3:4]
    rev = ((u * v[:, 3:4] + z * v[:, 2:3]) + y * v[:, 1:2]) + x * v[:, 0:1]
    report("mat4 * vec4", got, {"((c0x+c1y)+c2z)+c3w": fwd, "((c3w+c2z)+c1y)+c0x": rev})
    # 5. 1/x, sqrt
    got = run(lib, "r[i] = 1.0 / y;", inp)
    report("1.0 / y", got, {"round(1/y)": one / y})
    got = run(lib, "r[i] = sqrt(abs(x));", inp)
    report("sqrt", got, {"IEEE sqrt": np.sqrt(np.abs(x))})
    # 6. vector /= vector (cartesianToTextureCoord) and clamp/min/max NaN-free semantics
    got = run(lib, "vec4 p = x; p += y; p /= (z + y); r[i] = p;", inp)
    report("p += y; p /= (z+y)", got, {"round": (x + y) / (z + y), "mul rcp": (x + y) * (one / (z + y))})
    # 7. cartesianToTextureCoord's view_top path: p.z = 1 - p.z; ... 1 - p.z  (VolumeRenderer.cs:185-187)
    got = run(lib, "vec4 p = x; p.z = 1.0 - p.z; r[i] = vec4(p.x, 1.0 - p.z, p.y, 0.0);", inp)[:, 1]
    report("1 - (1 - z)", got, {"two subtractions": one - (one - x[:, 2]), "folded to z": x[:, 2]})
    return rows


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
