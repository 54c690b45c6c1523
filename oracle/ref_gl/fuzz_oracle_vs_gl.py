#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY -- differential test of the oracle against the EXECUTED reference.

Random configurations (volume dims 1..96 per axis, u8 / u16, anisotropic spacing, orbit / zoomed / hand-placed cameras
incl. eyes inside the box, windows incl. negative lower ends and the +1000 upload of 16-bit data, opacity 0..1, composite /
MIP, default / top / bottom view, image sizes with and without a partial workgroup row / column) are rendered by the
reference's unmodified shader under Mesa llvmpipe (oracle/_ref/libref_gl.so) and by oracle/vr_oracle.c; every frame
must agree bit for bit (never-written pixels of the truncated grid included).  A quarter of the trials are TRILINEAR:
the GL side is then the shader with `usampler3D` -> `sampler3D` over an R32F texture of the same values with GL_LINEAR
(a real GL's linear filter), the oracle side its TRILINEAR.  Nothing is stored but the summary line;
the committed frames of tests/golden/ref_gl/ are the travelling subset of the same comparison.

usage: python oracle/ref_gl/fuzz_oracle_vs_gl.py [trials=2000] [seed=1]
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from oracle import ref_gl  # noqa: E402
from oracle.ref_gl.mint_ref_gl_goldens import tri_shader_path  # noqa: E402

CLEAR = -7.0


def random_case(rng):
    b = int(rng.integers(1, 3))
    dims = tuple(int(v) for v in rng.integers(1, 97, size=3)) if rng.random() < 0.8 else tuple(int(v) for v in rng.choice([1, 2, 4, 8, 16, 32, 64], size=3))
    vmax = 255 if b == 1 else 4095
    kind = rng.random()
    if kind < 0.5:
        vol = oracle.gen_noise_ball(dims, b, int(rng.integers(1, 1 << 31)))
    elif kind < 0.8:
        vol = rng.integers(0, vmax + 1, size=(dims[2], dims[1], dims[0])).astype(np.uint8 if b == 1 else np.uint16)
    else:
        vol = rng.integers(0, 65536 if b == 2 else 256, size=(dims[2], dims[1], dims[0])).astype(np.uint8 if b == 1 else np.uint16)
    lo = int(rng.integers(-200 if b == 2 else 0, vmax // 2))
    hi = int(rng.integers(lo + 1, (70000 if b == 2 else 256) if rng.random() < 0.2 else vmax + 2))
    if b == 2:
        lo, hi = lo + 1000, hi + 1000                      # what setMinVal / setMaxVal upload (src/RendererCore.cpp:66-67,77-78)
    W = int(rng.integers(1, 14)) * 16 + (int(rng.integers(0, 16)) if rng.random() < 0.3 else 0)
    H = int(rng.integers(1, 11)) * 16 + (int(rng.integers(0, 16)) if rng.random() < 0.3 else 0)
    cam = oracle.Camera()
    for _ in range(int(rng.integers(0, 4))):
        if rng.random() < 0.3:
            cam.orient(float(rng.choice([-1.0, 1.0])), 0.0, 0.0)
        else:
            cam.orient(0.0, float(np.float32(rng.uniform(-1.6, 1.6))), float(np.float32(rng.uniform(-6.5, 6.5))))
    block = cam.block()
    if rng.random() < 0.15:                                # eye anywhere near / inside the box
        block = block.copy()
        block[16:19] = rng.uniform(-0.7, 0.7, size=3).astype(np.float32)
    mode = int(rng.integers(0, 6))
    return dict(vol=vol, img=(W, H), cam=block, alpha=float(np.float32(rng.choice([0.0, 1.0, 0.5, 0.11, 0.02, 0.004, float(rng.uniform(0, 1))]))),
                spacing=tuple(float(np.float32(v)) for v in (rng.uniform(0.3, 2.5, size=3) if rng.random() < 0.6 else (1.0, 1.0, 1.0))),
                lo=lo, hi=hi, mip=int(mode in (1, 4)), top=int(mode in (2, 4)), bottom=int(mode == 3), tri=int(rng.random() < 0.25))


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    pixels = bad_frames = lit = nan_frames = tri_frames = 0
    worst = None
    for k in range(trials):
        c = random_case(rng)
        W, H = c["img"]
        extra = dict(tex_filter=ref_gl.GL_LINEAR, shader_path=tri_shader_path()) if c["tri"] else {}
        g = ref_gl.render(c["vol"].astype(np.float32) if c["tri"] else c["vol"], (W, H), c["cam"], alpha_scale=c["alpha"], voxel_size=c["spacing"],
                          min_val=c["lo"], max_val=c["hi"], is_mip=c["mip"], view_top=c["top"], view_bottom=c["bottom"], clear_value=CLEAR, **extra)
        p = oracle.OracleParams(W, H, cam=c["cam"], alpha_scale=c["alpha"], voxel_size=c["spacing"], min_val=c["lo"], max_val=c["hi"],
                                is_mip=c["mip"], view_top=c["top"], view_bottom=c["bottom"], trunc_grid=1, threads=1, filter=c["tri"])
        o = np.full((H, W, 4), CLEAR, dtype=np.float32)
        oracle.render(c["vol"], p, out=o)
        diff = (g.view(np.uint32) != o.view(np.uint32)).any(axis=2)
        pixels += W * H
        tri_frames += c["tri"]
        lit += int((g[..., 3] > 0).sum())
        nan_frames += int(np.isnan(g).any())
        if diff.any():
            bad_frames += 1
            if worst is None:
                worst = (k, int(diff.sum()), {kk: vv for kk, vv in c.items() if kk != "vol"}, c["vol"].shape, str(c["vol"].dtype))
    dt = time.time() - t0
    print(f"oracle vs executed reference ({ref_gl.gl_info()}): seed {seed}, {trials} random frames ({tri_frames} TRILINEAR), {pixels} pixels, "
          f"{lit} of them with alpha > 0, {nan_frames} frames with a NaN (both sides alike), {bad_frames} frames with a bit-different pixel ({dt:.0f} s)")
    if worst:
        print("first mismatch:", worst)
    return 1 if bad_frames else 0


if __name__ == "__main__":
    sys.exit(main())
