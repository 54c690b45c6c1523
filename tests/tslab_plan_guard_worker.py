"""worker of tests/test_tslab_plan_guard.py: runs in a process of its own with VR_CORE_LIB pointing at the CHECKED build of the library
(lib/libvr_core_chk.so: vr_tslab.hip compiled with -DVR_TSLAB_CHECK), where the staged trilinear kernel reports, per pixel and in place
of the fetch count, how many of its staged taps lay outside the brick rectangle their layer was planned and loaded with."""
import importlib
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
import oracle  # cameras only (test infrastructure)

R = vra.renderer
rng = np.random.default_rng(2024)
report = {"frames": 0, "staged_pixels": 0, "violations": 0, "worst": None, "unchecked_frames": 0}


def cams(n_random):
    out = [("default", oracle.default_camera_block())]
    for name, (dz, da) in {"orbit_a": (0.42, 0.54), "orbit_b": (-0.9, 1.86), "offaxis": (-(np.pi / 6) / 0.7, (np.pi / 4) / 0.7), "behind": (0.0, np.pi / 0.7),
                           "pole": (-100.0, 0.3)}.items():
        c = oracle.Camera(); c.orient(0, dz, da); out.append((name, c.block()))
    c = oracle.Camera(); c.orient(1, 0, 0); c.orient(1, 0, 0); out.append(("close", c.block()))
    for far in (3, 12):                                        # far eye: the footprint margins grow with |E|
        c = oracle.Camera()
        for _ in range(far):
            c.orient(-1, 0, 0)
        c.orient(0, 0.3, -0.8); out.append((f"far{far}", c.block()))
    b = oracle.default_camera_block().copy(); b[12:15] = b[16:19] = (0.1, 0.05, 0.3); out.append(("inside", b))
    for k in range(n_random):
        c = oracle.Camera(); c.orient(0, float(rng.uniform(-2, 2)), float(rng.uniform(-6, 6)))
        if k % 3 == 0:
            c.orient(1, 0, 0)
        out.append((f"random{k}", c.block()))
    return out


def check(r, what):
    _, spp = r.countSamples(per_pixel=True)
    flagged = (spp & 0x40000000) != 0
    report["frames"] += 1
    if not flagged.any():
        report["unchecked_frames"] += 1
        return
    staged = flagged & ((spp & 0x20000000) != 0)
    viol = int((spp[flagged] & 0xffff).astype(np.int64).sum())
    report["staged_pixels"] += int(staged.sum())
    if viol and (report["worst"] is None or viol > report["worst"][1]):
        report["worst"] = (what, viol)
    report["violations"] += viol


def small_volumes():
    for dims, spacing, dtype in (((64, 64, 64), (1, 1, 1), np.uint16), ((96, 80, 72), (1.0, 0.8, 1.7), np.uint16), ((130, 33, 47), (1, 1, 1), np.uint8),
                                 ((40, 56, 24), (0.6, 1.0, 1.3), np.uint16), ((7, 5, 3), (1, 1, 1), np.uint16), ((200, 24, 160), (0.3, 2.5, 1.0), np.uint16),
                                 ((256, 256, 256), (1, 1, 1), np.uint16)):
        vol = rng.integers(0, 4096 if dtype == np.uint16 else 256, size=dims[::-1], dtype=np.int64).astype(dtype)
        with vra.RendererCore(0) as r:
            r.setup((200, 144) if dims[0] < 256 else (320, 200)); r.loadShader("x"); r.setQuirks(0)
            r.setVolume(vol, spacing); r.setFilter(R.FILTER_TRILINEAR); r.setWindow(0, 4095 if dtype == np.uint16 else 255); r.setAlpha(0.02)
            for name, block in cams(4):
                r.setCameraBlock(block)
                for variant in (6, 8, 9, 10, 11):
                    r.setKernelVariant(variant); r.render()
                    check(r, f"{dims} {spacing} {np.dtype(dtype).name} {name} variant {variant}")
            for top, bottom in ((True, False), (False, True)):
                r.setInitialCameraRotation(top, bottom)
                for name, block in cams(1)[:4]:
                    r.setCameraBlock(block)
                    for variant in (6, 8, 11):
                        r.setKernelVariant(variant); r.render()
                        check(r, f"{dims} view {'top' if top else 'bottom'} {name} variant {variant}")


def cfg3():
    with vra.RendererCore(0) as r:
        r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,) * 3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004); r.setFilter(R.FILTER_TRILINEAR)
        for name, block in cams(3):
            r.setCameraBlock(block)
            for variant in (6, 8, 9, 10, 11):
                r.setKernelVariant(variant); r.render()
                check(r, f"cfg3 {name} variant {variant}")


def skipping():
    """exact empty-space skipping (round 5): layers nobody samples are not requested -- every staged tap must still find its layer asked for
    (the checked build keeps a bit per requested layer) and inside the rectangle it was planned with"""
    def blobs(dims, dtype, bg):
        nx, ny, nz = dims
        hi = 255 if dtype == np.uint8 else 4095
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        v = rng.integers(0, bg + 1, size=(nz, ny, nx)).astype(np.int64)
        for _ in range(4):
            c = [rng.uniform(0.15, 0.85) * n for n in (nx, ny, nz)]
            rad = [rng.uniform(0.08, 0.3) * n + 1.5 for n in (nx, ny, nz)]
            d2 = ((x - c[0]) / rad[0]) ** 2 + ((y - c[1]) / rad[1]) ** 2 + ((z - c[2]) / rad[2]) ** 2
            v[d2 < 1.0] = np.maximum(v[d2 < 1.0], (hi * (1.0 - 0.7 * d2[d2 < 1.0])).astype(np.int64))
        return np.clip(v, 0, hi).astype(dtype)
    for dims, spacing, dtype in (((96, 80, 72), (1, 1, 1), np.uint16), ((130, 33, 47), (1.0, 0.8, 1.7), np.uint8), ((200, 168, 184), (1, 1, 1), np.uint16),
                                 ((256, 256, 256), (1, 1, 1), np.uint8)):
        bg = 6 if dtype == np.uint8 else 60
        vol = blobs(dims, dtype, bg)
        with vra.RendererCore(0) as r:
            r.setup((200, 144) if dims[0] < 200 else (320, 208)); r.loadShader("x"); r.setQuirks(0)
            r.setVolume(vol, spacing); r.setFilter(R.FILTER_TRILINEAR); r.setSkipEmpty(True)
            for lo, alpha in ((bg, 0.03), (3 * bg, 1.0)):
                r.setWindow(lo, 4095 if dtype == np.uint16 else 255); r.setAlpha(alpha)
                for name, block in cams(3):
                    r.setCameraBlock(block)
                    for variant in (6, 8, 9, 10, 11):
                        r.setKernelVariant(variant); r.render()
                        check(r, f"skipping {dims} {np.dtype(dtype).name} window lo {lo} {name} variant {variant}")


small_volumes()
skipping()
if "full" in sys.argv:
    cfg3()
print(json.dumps(report))
