#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY -- mints tests/golden/ref_gl/*.npz.

Every frame written here is the output of the reference's UNMODIFIED compute shader
(/root/reference/VolumeRenderer.cs, read where it lies) compiled and executed by Mesa 23.2.1
llvmpipe with the reference's own GL call sequence (oracle/ref_gl/ref_gl_runner.c).  These are
reference vectors: the oracle had no part in producing them.  What is stored is data only --
the inputs of each case (generator name + integer parameters, the 21 camera floats, the seven
uniforms) in ref_gl_manifest.json, and of the output frame
  * the sha-256 of the full H x W x 4 float32 frame (row 0 = bottom), and
  * the kept image rows (all rows of small frames, every K-th row of large ones) as
    (grey, alpha) float32 pairs -- the shader's output has r == g == b, asserted here.

One stated deviation from the reference's GL state: the volume texture's MIN/MAG filter is
GL_NEAREST, not the reference's GL_LINEAR (src/RendererCore.cpp:414-415), because a conformant
GL treats an integer texture with a LINEAR filter as incomplete and the reference then renders
all zeros -- `f4` below records exactly that, plus the check that Mesa's own app-compat switch
(force_integer_tex_nearest=true, reference calls untouched) yields the same frame as GL_NEAREST.

Run from the repo root:  python oracle/ref_gl/mint_ref_gl_goldens.py
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402  (volume generators + camera restatement only: INPUTS, not outputs)
from oracle import ref_gl  # noqa: E402

OUT_DIR = ROOT / "tests" / "golden" / "ref_gl"
MANIFEST = ROOT / "tests" / "golden" / "ref_gl_manifest.json"
CLEAR = -7.0   # target pre-fill, to see which pixels the dispatch never wrote (Q1)


def cam_default():
    return oracle.default_camera_block()


def cam_orbit(*moves):
    c = oracle.Camera()
    for m in moves:
        c.orient(*m)
    return c.block()


# name -> case.  vol: ("sphere", n, R) | ("noise", (nx, ny, nz), bytes, seed).  window = the GUI's min_val/max_val
# (RendererCore::min_val / max_val); `uploaded` adds the +1000 of setMinVal/setMaxVal for 16-bit data.
SPH64 = ("sphere", 64, 28)
SPH256 = ("sphere", 256, 112)
CFG2 = ("noise", (512, 512, 452), 2, 0x9E3779B9)
CFG3U8 = ("noise", (1024, 1024, 1024), 1, 0x9E3779B9)
# round 6: the largest 16-bit volume llvmpipe accepts at cfg3's x/y extent -- 1024 x 1024 x 448 uint16 = 0.94 GiB (the cap is
# 1 GiB per texture) -- and, for the filtered path, 1024 x 1024 x 224 as R32F (0.94 GiB)
CFG3SLAB = ("noise", (1024, 1024, 448), 2, 0x9E3779B9)
CFG3SLABTRI = ("noise", (1024, 1024, 224), 2, 0x9E3779B9)
SMALL16 = ("noise", (48, 40, 36), 2, 7)
ODD8 = ("noise", (37, 53, 29), 1, 3)
ORBIT_A = [(0.0, 0.06 * 7, 0.06 * 9)]
ORBIT_B = [(0.0, 0.06 * 7, 0.06 * 9), (0.0, -0.06 * 15, 0.06 * 31)]
OFFAXIS = [(0.0, np.pi / 3, np.pi / 4)]
CASES = {
    # BASELINE config 0, the full frame
    "cfg0_a1": dict(vol=SPH64, img=(256, 256), alpha=1.0, window=(0, 255), rows=1),
    "cfg0_a0.05": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1),
    "cfg0_mip": dict(vol=SPH64, img=(256, 256), alpha=0.5, window=(0, 255), rows=1, mip=1),
    "cfg0_orbit": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1, cam=ORBIT_A),
    "cfg0_view_top": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1, top=1),
    "cfg0_view_bottom": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1, bottom=1, cam=ORBIT_A),
    # Q5: eye inside the box (t_min < 0 is used unclamped, VolumeRenderer.cs:107 vs :237): an orbit pose with the eye
    # of the UBO block moved to a point inside [-0.5, 0.5]^3
    "cfg0_eye_inside": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1, cam=ORBIT_A, eye=(0.11, -0.07, 0.23)),
    # Q1: 250x250 window -> 15x15 workgroups, the last 10 columns / rows are never written
    "cfg0_q1_250": dict(vol=SPH64, img=(250, 250), alpha=0.05, window=(0, 255), rows=1),
    # 16-bit data: the +1000 window quirk (Q10), anisotropic spacing, non-cubic dims, all modes and views
    "u16_small_window": dict(vol=SMALL16, img=(320, 240), alpha=0.05, window=(0, 4095), spacing=(1.0, 1.0, 1.5), rows=1, cam=ORBIT_A),
    "u16_small_top": dict(vol=SMALL16, img=(320, 240), alpha=0.1, window=(-800, 2000), spacing=(1.0, 0.8, 1.5), rows=1, top=1, cam=ORBIT_A),
    "u16_small_mip_bottom": dict(vol=SMALL16, img=(320, 240), alpha=0.7, window=(-800, 2000), spacing=(1.0, 0.8, 1.5), rows=1, bottom=1, mip=1,
                                 cam=ORBIT_B),
    # nx % 4 != 0: the UNPACK_ALIGNMENT 1 upload path (src/RendererCore.cpp:417-418)
    "u8_odd_dims": dict(vol=ODD8, img=(304, 208), alpha=0.03, window=(0, 255), rows=1, cam=ORBIT_B),
    # BASELINE config 1 shape (Bonsai stand-in), every 40th row
    "cfg1_shape_a1": dict(vol=SPH256, img=(1280, 720), alpha=1.0, window=(0, 255), rows=40),
    "cfg1_shape_a0.02": dict(vol=SPH256, img=(1280, 720), alpha=0.02, window=(0, 255), rows=40),
    # BASELINE config 2 shape (Head-CT stand-in, 16-bit, data window + 1000), every 60th row
    "cfg2_shape_window": dict(vol=CFG2, img=(1920, 1080), alpha=0.05, window=(0, 4095), rows=60),
    "cfg2_shape_offaxis": dict(vol=CFG2, img=(1920, 1080), alpha=0.01, window=(0, 4095), rows=60, cam=OFFAXIS),
    "cfg2_shape_mip": dict(vol=CFG2, img=(1920, 1080), alpha=0.2, window=(0, 4095), rows=60, mip=1),
    # BASELINE config 3 geometry (1024^3 @ 1920x1080, step 1/1024) with 8-bit voxels: llvmpipe caps a texture at 1 GiB,
    # so the 2 GiB uint16 volume itself cannot be uploaded; ray set-up, lattice and traversal are identical
    "cfg3_geom_u8_deep": dict(vol=CFG3U8, img=(1920, 1080), alpha=0.004, window=(0, 255), rows=60),
    "cfg3_geom_u8_shallow": dict(vol=CFG3U8, img=(1920, 1080), alpha=1.0, window=(0, 255), rows=60),
    "cfg3_geom_u8_offaxis": dict(vol=CFG3U8, img=(1920, 1080), alpha=0.004, window=(0, 255), rows=60, cam=OFFAXIS),
    # cfg3's x/y extent with 16-bit voxels (u16 bricks + the 12-bit packed copy + the address tables of the HIP path at 1024 x 1024),
    # the window [0, 4095] uploaded as [1000, 5095] (Q10); three regimes like cfg3_geom_u8_*
    "cfg3_slab_u16_deep": dict(vol=CFG3SLAB, img=(1920, 1080), alpha=0.004, window=(0, 4095), rows=60),
    "cfg3_slab_u16_shallow": dict(vol=CFG3SLAB, img=(1920, 1080), alpha=1.0, window=(0, 4095), rows=60),
    "cfg3_slab_u16_offaxis": dict(vol=CFG3SLAB, img=(1920, 1080), alpha=0.004, window=(0, 4095), rows=60, cam=OFFAXIS),
}


def _random_cases(n=16, seed=20260929):
    """small whole frames over random dims / voxel types / spacings / orbit cameras / windows / opacities / modes / views:
    the combinations nobody thought of.  Deterministic (fixed seed), so re-minting reproduces the manifest."""
    rng = np.random.default_rng(seed)
    out = {}
    for k in range(n):
        b = 1 if k % 2 == 0 else 2
        dims = tuple(int(v) for v in rng.integers(5, 72, size=3))
        vmax = 255 if b == 1 else 4095
        lo = int(rng.integers(-50 if b == 2 else 0, vmax // 3))
        hi = int(rng.integers(vmax // 2, vmax + (400 if b == 2 else 1)))
        mode = int(rng.integers(0, 6))
        c = dict(vol=("noise", dims, b, int(rng.integers(1, 1 << 30))), img=(16 * int(rng.integers(6, 13)), 16 * int(rng.integers(5, 10))),
                 alpha=float(np.float32(rng.choice([1.0, 0.5, 0.11, 0.02]))), window=(lo, hi), rows=1,
                 spacing=tuple(float(np.float32(v)) for v in np.round(rng.uniform(0.4, 2.2, size=3), 2)),
                 cam=[(float(rng.choice([-1.0, 0.0, 0.0, 1.0])), float(np.float32(rng.uniform(-1.4, 1.4))), float(np.float32(rng.uniform(-3.1, 3.1))))],
                 mip=int(mode == 1 or mode == 4), top=int(mode == 2 or mode == 4), bottom=int(mode == 3))
        out[f"rnd_{k:02d}"] = c
    return out


CASES.update(_random_cases())

# ---- TRILINEAR (north-star's filter; SURVEY F1/F4: the reference's integer texture cannot be filtered, so it has no TRILINEAR
# semantics of its own).  What a real GL's linear filter computes is measured with ONE token of the shader changed at run
# time -- `usampler3D` -> `sampler3D` -- over an R32F texture holding the same voxel values, MIN/MAG GL_LINEAR (the
# reference's own filter state), everything else of the shader and the call sequence untouched.  The oracle's TRILINEAR
# (GL's linear rule, each lerp one fma) and the HIP kernels reproduce these frames bit for bit.
TRI = dict(tri=1)
CASES.update({
    "tri_cfg0_a0.05": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1, **TRI),
    "tri_cfg0_a1_orbit": dict(vol=SPH64, img=(256, 256), alpha=1.0, window=(0, 255), rows=1, cam=ORBIT_A, **TRI),
    "tri_cfg0_mip_top": dict(vol=SPH64, img=(256, 256), alpha=0.5, window=(0, 255), rows=1, mip=1, top=1, cam=ORBIT_B, **TRI),
    "tri_cfg0_bottom_inside": dict(vol=SPH64, img=(256, 256), alpha=0.05, window=(0, 255), rows=1, bottom=1, cam=ORBIT_A, eye=(0.11, -0.07, 0.23), **TRI),
    "tri_u16_small_window": dict(vol=SMALL16, img=(320, 240), alpha=0.05, window=(0, 4095), spacing=(1.0, 1.0, 1.5), rows=1, cam=ORBIT_A, **TRI),
    "tri_u16_small_mip": dict(vol=SMALL16, img=(320, 240), alpha=0.7, window=(-800, 2000), spacing=(1.0, 0.8, 1.5), rows=1, mip=1, cam=ORBIT_B, **TRI),
    "tri_u8_odd_dims": dict(vol=ODD8, img=(304, 208), alpha=0.03, window=(0, 255), rows=1, cam=ORBIT_B, **TRI),
    "tri_cfg1_shape_a0.02": dict(vol=SPH256, img=(1280, 720), alpha=0.02, window=(0, 255), rows=40, **TRI),
    "tri_cfg2_shape_window": dict(vol=CFG2, img=(1920, 1080), alpha=0.05, window=(0, 4095), rows=60, **TRI),
    "tri_cfg2_shape_offaxis": dict(vol=CFG2, img=(1920, 1080), alpha=0.01, window=(0, 4095), rows=60, cam=OFFAXIS, **TRI),
    "tri_cfg3_slab_u16_deep": dict(vol=CFG3SLABTRI, img=(1920, 1080), alpha=0.004, window=(0, 4095), rows=60, **TRI),
    "tri_cfg3_slab_u16_offaxis": dict(vol=CFG3SLABTRI, img=(1920, 1080), alpha=0.004, window=(0, 4095), rows=60, cam=OFFAXIS, **TRI),
})
for _k, _c in _random_cases(8, seed=20260930).items():
    CASES["tri_" + _k] = dict(_c, **TRI)

_TRI_SHADER = None


def tri_shader_path():
    """the reference shader with its one `usampler3D` spelled `sampler3D`, in a temporary file (deleted at exit; never in the repo)"""
    global _TRI_SHADER
    if _TRI_SHADER is None:
        import atexit
        import tempfile
        src = ref_gl.SHADER_PATH.read_text()
        assert src.count("usampler3D") == 1
        f = tempfile.NamedTemporaryFile("w", suffix=".cs", delete=False)
        f.write(src.replace("usampler3D", "sampler3D"))
        f.close()
        atexit.register(lambda: os.path.exists(f.name) and os.unlink(f.name))
        _TRI_SHADER = f.name
    return _TRI_SHADER


def make_volume(spec):
    if spec[0] == "sphere":
        return oracle.gen_sphere_u8(spec[1], spec[2])
    return oracle.gen_noise_ball(spec[1], spec[2], spec[3])


def uploaded_window(c, bytes_per_voxel):
    lo, hi = c["window"]
    return (lo + 1000, hi + 1000) if bytes_per_voxel == 2 else (lo, hi)


def run_gl(c, vol, **over):
    lo, hi = uploaded_window(c, vol.dtype.itemsize)
    cam = cam_orbit(*c["cam"]) if "cam" in c else cam_default()
    if "eye" in c:
        cam = cam.copy()
        cam[16:19] = np.asarray(c["eye"], dtype=np.float32)
    if c.get("tri"):      # a real GL's linear filter: sampler3D variant + R32F texture of the same values + the reference's GL_LINEAR
        over = dict(dict(tex_filter=ref_gl.GL_LINEAR, shader_path=tri_shader_path()), **over)
        vol_gl = vol.astype(np.float32)
    else:
        vol_gl = vol
    frame = ref_gl.render(vol_gl, c["img"], cam, alpha_scale=c["alpha"], voxel_size=c.get("spacing", (1.0, 1.0, 1.0)), min_val=lo, max_val=hi,
                          is_mip=c.get("mip", 0), view_top=c.get("top", 0), view_bottom=c.get("bottom", 0), clear_value=CLEAR, **over)
    return frame, cam


def measure_f4():
    """SURVEY F4, measured: the reference's GL_LINEAR on the integer texture, as is; GL_NEAREST; and Mesa's
    force_integer_tex_nearest with the reference's GL_LINEAR untouched (needs a fresh process: the switch is read
    when the screen is created)."""
    c = CASES["cfg0_a1"]
    vol = make_volume(c["vol"])
    linear, _ = run_gl(c, vol, tex_filter=ref_gl.GL_LINEAR)
    nearest, _ = run_gl(c, vol, tex_filter=ref_gl.GL_NEAREST)
    code = ("import sys, hashlib; sys.path.insert(0, %r); import oracle; from oracle import ref_gl\n"
            "from oracle.ref_gl import mint_ref_gl_goldens as m\n"
            "c = m.CASES['cfg0_a1']; f, _ = m.run_gl(c, m.make_volume(c['vol']), tex_filter=ref_gl.GL_LINEAR)\n"
            "print(hashlib.sha256(f.tobytes()).hexdigest())\n") % str(ROOT)
    env = dict(os.environ, force_integer_tex_nearest="true")
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    forced = proc.stdout.strip().splitlines()[-1] if proc.returncode == 0 and proc.stdout.strip() else "failed: " + proc.stderr[-300:]
    return {
        "case": "cfg0_a1",
        "GL_LINEAR_as_the_reference_sets_it": {"max": float(linear.max()), "min": float(linear.min()),
                                                 "pixels_with_alpha": int((linear[..., 3] > 0).sum())},
        "GL_NEAREST": {"max": float(nearest.max()), "pixels_with_alpha": int((nearest[..., 3] > 0).sum()),
                       "sha256": hashlib.sha256(nearest.tobytes()).hexdigest()},
        "GL_LINEAR_with_mesa_force_integer_tex_nearest": {"sha256": forced},
    }


def main():
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    # --only SUBSTRING: mint the matching cases and merge them into the committed manifest (the other entries and frames are
    # left as they are; same GL, same shader -- asserted)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    head = {"gl": ref_gl.gl_info(), "shader_sha256": hashlib.sha256(ref_gl.SHADER_PATH.read_bytes()).hexdigest(),
            "deviation": "volume texture MIN/MAG filter GL_NEAREST instead of GL_LINEAR (see f4)", "clear_value": CLEAR}
    if only is not None:
        manifest = json.loads(MANIFEST.read_text())
        for k, v in head.items():
            assert manifest[k] == v, (k, manifest[k], v)
    else:
        manifest = dict(head, cases={})
        manifest["f4"] = measure_f4()
        print("F4:", json.dumps(manifest["f4"]))
    vols = {}
    for name, c in CASES.items():
        if only is not None and only not in name:
            continue
        key = repr(c["vol"])
        if key not in vols:
            vols.clear()   # one big volume at a time
            vols[key] = make_volume(c["vol"])
        vol = vols[key]
        t0 = time.time()
        frame, cam = run_gl(c, vol)
        dt = time.time() - t0
        W, H = c["img"]
        wlim, hlim = (W // 16) * 16, (H // 16) * 16
        written = frame[:hlim, :wlim]
        assert np.isfinite(written).all(), name
        assert np.array_equal(written[..., 0], written[..., 1]) and np.array_equal(written[..., 0], written[..., 2]), name
        # Q1: nothing outside the truncated grid was touched
        untouched = np.ones((H, W), dtype=bool)
        untouched[:hlim, :wlim] = False
        assert (frame[untouched] == CLEAR).all(), name
        frame = frame.copy()
        frame[untouched] = 0.0   # canonical form of never-written pixels in the hash and the rows
        rows = np.arange(c["rows"] // 2, H, c["rows"], dtype=np.int32)
        ga = np.ascontiguousarray(frame[rows][..., [0, 3]])
        np.savez_compressed(OUT_DIR / f"{name}.npz", rows=rows, ga=ga)
        lo, hi = uploaded_window(c, vol.dtype.itemsize)
        manifest["cases"][name] = {
            "vol": [c["vol"][0]] + [list(v) if isinstance(v, tuple) else v for v in c["vol"][1:]],
            "img": list(c["img"]), "cam_f32_hex": [np.float32(v).tobytes().hex() for v in cam], "cam": [float(v) for v in cam],
            "alpha": c["alpha"], "spacing": list(c.get("spacing", (1.0, 1.0, 1.0))), "window": list(c["window"]), "uploaded_window": [lo, hi],
            "mip": c.get("mip", 0), "top": c.get("top", 0), "bottom": c.get("bottom", 0), "row_stride": c["rows"],
            "filter": "trilinear" if c.get("tri") else "nearest",
            "shader_variant": "usampler3D -> sampler3D (one token), R32F texture of the same voxel values, GL_LINEAR" if c.get("tri") else "unmodified",
            "written_extent": [wlim, hlim], "frame_sha256": hashlib.sha256(frame.tobytes()).hexdigest(),
            "pixels_with_alpha": int((frame[..., 3] > 0).sum()), "gl_seconds": round(dt, 2),
        }
        print(f"{name:26s} {dt:6.2f}s  rows {len(rows):4d}  alpha px {manifest['cases'][name]['pixels_with_alpha']:8d}  {manifest['cases'][name]['frame_sha256'][:16]}",
              flush=True)
    if only is not None:      # keep the manifest in CASES order
        manifest["cases"] = {n: manifest["cases"][n] for n in CASES if n in manifest["cases"]}
    MANIFEST.write_text(json.dumps(manifest, indent=1) + "\n")
    print("wrote", MANIFEST)


if __name__ == "__main__":
    main()
