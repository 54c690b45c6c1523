#!/usr/bin/env python3
"""Mint regression goldens of the CPU oracle for BASELINE configs 0-2.

These do NOT pin the oracle to the reference -- tests/golden/ref_gl/ does that (the reference's shader
executed under Mesa llvmpipe, oracle/ref_gl/mint_ref_gl_goldens.py).  They freeze the oracle's own
answers, including TRILINEAR (no reference semantics) and sample counts, so that a later edit of
oracle/vr_oracle.c (or of the generators) cannot drift unnoticed, and they give the GPU tests a
fixture that travels.

cfg0: the full 256x256 frame; cfg1/cfg2 shapes: every K-th image row of the full-size frame.
Run from the repo root:  python tests/golden/make_oracle_goldens.py
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

CASES = {
    # name: (volume spec, image, params, row stride)
    "cfg0_sphere64_u8_256x256_a1": dict(vol=("sphere", 64, 28), img=(256, 256), alpha=1.0, window=(0, 255), rows=1),
    "cfg0_sphere64_u8_256x256_a0.05": dict(vol=("sphere", 64, 28), img=(256, 256), alpha=0.05, window=(0, 255), rows=1),
    "cfg0_sphere64_u8_256x256_mip": dict(vol=("sphere", 64, 28), img=(256, 256), alpha=0.5, window=(0, 255), rows=1, mip=1),
    "cfg1_sphere256_u8_1280x720_a1": dict(vol=("sphere", 256, 112), img=(1280, 720), alpha=1.0, window=(0, 255), rows=40),
    "cfg1_sphere256_u8_1280x720_a0.02": dict(vol=("sphere", 256, 112), img=(1280, 720), alpha=0.02, window=(0, 255), rows=40),
    # cfg2 shape: u16 data window with the +1000 quirk of setMinVal/setMaxVal (src/RendererCore.cpp:66-69)
    "cfg2_noise512x512x452_u16_1920x1080_window": dict(vol=("noise", (512, 512, 452), 2, 0x9E3779B9), img=(1920, 1080), alpha=0.05,
                                                      window=(0 + 1000, 4095 + 1000), rows=60),
    "cfg2_noise512x512x452_u16_1920x1080_trilinear": dict(vol=("noise", (512, 512, 452), 2, 0x9E3779B9), img=(1920, 1080), alpha=0.05,
                                                         window=(0, 4095), rows=120, filter=1),
}


def make_volume(spec):
    if spec[0] == "sphere":
        return oracle.gen_sphere_u8(spec[1], spec[2])
    return oracle.gen_noise_ball(spec[1], spec[2], spec[3])


def run_case(c, volume=None):
    vol = make_volume(c["vol"]) if volume is None else volume
    W, H = c["img"]
    p = oracle.OracleParams(W, H, alpha_scale=c["alpha"], min_val=c["window"][0], max_val=c["window"][1],
                            is_mip=c.get("mip", 0), filter=c.get("filter", 0), threads=0)
    rows = list(range(c["rows"] // 2, H, c["rows"]))
    h = hashlib.sha256()
    total = 0
    hit = 0
    probe = {}
    out = np.zeros((H, W, 4), dtype=np.float32)
    for y in rows:
        p.row_begin, p.row_end = y, y + 1
        _, s = oracle.render(vol, p, out=out)
        total += s
        h.update(out[y].tobytes())
        hit += int((out[y, :, 3] > 0).sum())
    ymid = rows[len(rows) // 2]
    for x in (W // 2, W // 2 - W // 8, W // 2 + W // 7):
        probe[f"{x},{ymid}"] = [float(v) for v in out[ymid, x]]
    return {"rows": len(rows), "samples": int(total), "pixels_with_alpha": hit, "sha256_rows": h.hexdigest(), "probe_rgba": probe}


def main():
    res = {}
    vols = {}
    for name, c in CASES.items():
        key = repr(c["vol"])
        if key not in vols:
            vols[key] = make_volume(c["vol"])
        res[name] = {"case": {k: v for k, v in c.items()}, "expect": run_case(c, vols[key])}
        print(name, res[name]["expect"]["samples"], res[name]["expect"]["sha256_rows"][:16])
    out = Path(__file__).with_name("oracle_goldens.json")
    out.write_text(json.dumps(res, indent=1) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
