"""Exact empty-space skipping on the LDS-staged TRILINEAR kernel (vr_tslab.hip: SKIP; vr_set_skip_empty with
TRILINEAR filtering -- config 4's "adaptive step" together with north-star's filter).

The kernel skips a tile's brick LAYER -- no DMA request, and the phase only steps the rays through it -- when the
dilated cell-max grid says every voxel the layer's samples could touch classifies to exactly (0,0,0,0).  Skipping is
defined as invisible: frames AND per-pixel sample counts must equal the same launch without it bit for bit, and the
oracle's (which knows nothing of skipping).  The volumes below have real empty space (zeros / background below the
window's lower end) in front of, behind and between visible regions, so phases are skipped at the start, in the middle
and at the end of a tile's march; every mode, the rotated views, both voxel types and every workgroup shape run.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TSLAB = "raymarch_tslab_kernel"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def blobs(rng, dims, dtype, background):
    """ellipsoidal blobs of smooth + noisy values in a background of noise <= `background`"""
    nx, ny, nz = dims
    hi = 255 if dtype == np.uint8 else 4095
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    v = rng.integers(0, background + 1, size=(nz, ny, nx)).astype(np.int64)
    for _ in range(4):
        c = [rng.uniform(0.15, 0.85) * n for n in (nx, ny, nz)]
        rad = [rng.uniform(0.08, 0.3) * n + 1.5 for n in (nx, ny, nz)]
        d2 = ((x - c[0]) / rad[0]) ** 2 + ((y - c[1]) / rad[1]) ** 2 + ((z - c[2]) / rad[2]) ** 2
        inside = d2 < 1.0
        v[inside] = np.maximum(v[inside], (hi * (1.0 - 0.7 * d2[inside])).astype(np.int64) - rng.integers(0, 5, size=int(inside.sum())))
    return np.clip(v, 0, hi).astype(dtype)


def camera_blocks(oracle, rng, n_random=2):
    out = [("default", oracle.default_camera_block())]
    for name, (dz, da) in {"orbit_a": (0.06 * 7, 0.06 * 9), "offaxis": (-(np.pi / 6) / 0.7, (np.pi / 4) / 0.7), "behind": (0.0, np.pi / 0.7),
                           "side": (0.0, (np.pi / 2) / 0.7)}.items():
        c = oracle.Camera()
        c.orient(0, dz, da)
        out.append((name, c.block()))
    b = oracle.default_camera_block().copy()                    # eye inside the box
    b[12:15] = b[16:19] = (0.1, 0.05, 0.3)
    out.append(("inside", b))
    for k in range(n_random):
        c = oracle.Camera()
        c.orient(0, float(rng.uniform(-2, 2)), float(rng.uniform(-6, 6)))
        out.append((f"random{k}", c.block()))
    return out


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("dims,spacing", [((96, 80, 72), (1, 1, 1)), ((130, 33, 47), (1.0, 0.8, 1.7)), ((200, 168, 184), (1, 1, 1)), ((7, 5, 3), (1, 1, 1))],
                         ids=["noncubic", "odd_aniso", "mid", "tiny"])
# (7: every tile on the path of tiles that do not fit LDS -- it skips per ray and batch of four samples)
@pytest.mark.parametrize("variant", [6, 7, 8, 9, 10, 11], ids=["staged", "unstaged", "half", "halftall", "three", "small"])
def test_skipping_is_invisible_grey(vra, oracle, dtype, dims, spacing, variant):
    rng = np.random.default_rng(sum(d * 17 ** k for k, d in enumerate(dims)) + np.dtype(dtype).itemsize + variant)
    hi = 255 if dtype == np.uint8 else 4095
    bg = 6 if dtype == np.uint8 else 60
    vol = blobs(rng, dims, dtype, bg)
    size = (200, 144) if dims[0] < 200 else (320, 208)
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.setVolume(vol, spacing)
        r.setFilter(R.FILTER_TRILINEAR)
        r.setKernelVariant(variant)
        # window's lower end: at the background (everything outside the blobs is empty), above it, and 0 (nothing is empty)
        for (lo, hi_w, alpha) in ((bg, hi, 0.03), (bg * 3, hi - hi // 4, 1.0), (0, hi, 0.02)):
            r.setWindow(lo, hi_w)
            r.setAlpha(alpha)
            for name, block in camera_blocks(oracle, rng):
                r.setCameraBlock(block)
                r.setSkipEmpty(False)
                r.render()
                assert r.last_kernel_name == TSLAB
                plain = r.readPixels()
                r.setSkipEmpty(True)
                r.render()
                assert r.last_kernel_name == TSLAB
                got = r.readPixels()
                total, spp = r.countSamples(per_pixel=True)
                what = f"{dims} {np.dtype(dtype).name} window [{lo},{hi_w}] alpha {alpha} camera {name} variant {variant}"
                assert np.array_equal(bits(plain), bits(got)), f"{what}: skipping changed {int((bits(plain) != bits(got)).sum())} words"
                p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo, max_val=hi_w, filter=1, threads=8)
                want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
                assert total == want_total, what
                assert np.array_equal(spp, want_spp), what
                assert np.array_equal(bits(got), bits(want)), f"{what}: max|diff|={np.abs(got - want).max()}"
        r.setSkipEmpty(False)


@pytest.mark.parametrize("mode", ["mip", "tf", "mip_tf", "top", "bottom", "tf_top", "tf_hole"])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("variant", [6, 7, 8, 10, 11], ids=["staged", "unstaged", "half", "three", "small"])
def test_skipping_is_invisible_modes_and_views(vra, oracle, dtype, mode, variant):
    rng = np.random.default_rng(11)
    dims = (88, 72, 96)
    hi = 255 if dtype == np.uint8 else 4095
    bg = 5 if dtype == np.uint8 else 50
    vol = blobs(rng, dims, dtype, bg)
    mip, tf = "mip" in mode, "tf" in mode
    top, bottom = "top" in mode, "bottom" in mode
    size = (176, 120)
    lo, hi_w = (bg, hi - 7) if not tf else (0, hi)
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.setVolume(vol)
        r.setFilter(R.FILTER_TRILINEAR)
        r.setWindow(lo, hi_w)
        alpha = 0.3 if mip else 0.04
        r.setAlpha(alpha)
        r.setMIP(mip)
        tf_lut = None
        if tf:
            if mode == "tf_hole":
                # invisible, VISIBLE, invisible again, visible: only the first run of invisible entries may be skipped -- and the
                # interpolated values between two integers reach table entries no voxel value maps to
                r.setTransferFunction([0, 4, 8, 40, 120, 255], [[0, 0, 0, 0], [0, 0, 0, 0], [0.9, 0.2, 0.1, 0.5], [0, 0, 0, 0], [0.2, 0.8, 0.3, 0.2], [1, 1, 1, 0.9]])
            else:
                r.setTransferFunction([0, 30, 90, 160, 255], [[0, 0, 0, 0], [0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])
            tf_lut = r.getTransferLut()
        r.setInitialCameraRotation(top, bottom)
        r.setKernelVariant(variant)
        for name, block in camera_blocks(oracle, rng, n_random=1):
            r.setCameraBlock(block)
            r.setSkipEmpty(False)
            r.render()
            plain = r.readPixels()
            r.setSkipEmpty(True)
            r.render()
            assert r.last_kernel_name == TSLAB, (mode, name)
            got = r.readPixels()
            total, spp = r.countSamples(per_pixel=True)
            assert np.array_equal(bits(plain), bits(got)), f"{mode} {name}: skipping changed {int((bits(plain) != bits(got)).sum())} words"
            p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, min_val=lo, max_val=hi_w, is_mip=int(mip),
                                    view_top=int(top), view_bottom=int(bottom), tf_rgba=tf_lut, filter=1, threads=8)
            want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
            assert total == want_total and np.array_equal(spp, want_spp), (mode, name)
            assert np.array_equal(bits(got), bits(want)), f"{mode} {name}: max|diff|={np.abs(got - want).max()}"
        r.setSkipEmpty(False)


def test_skipping_pays_and_is_invisible_at_full_size(vra, oracle):
    """cfg3's volume (1024^3 u16 noise ball) with the window raised above the background noise, TRILINEAR: the corners outside
    the ball are skipped -- the frame must not change by a bit, sparse rows must equal the oracle's, and the launch must
    get faster (the layers outside the ball are neither requested from HBM nor sampled).  Default and off-axis pose, the
    shapes the two poses run on."""
    R = vra.renderer
    W, H, N = 1920, 1080, 1024
    with vra.RendererCore(0) as r:
        r.setup((W, H))
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), 2, 0x9E3779B9)
        r.setWindow(64, 4095); r.setAlpha(0.004); r.setFilter(R.FILTER_TRILINEAR)
        vol = r.readVolume()
        for pose, variant in (("default", 6), ("default", 8), ("offaxis", 9), ("default", 7)):
            r.resetCamera()
            if pose == "offaxis":
                r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
            r.setKernelVariant(variant)

            def timed(skip):
                r.setSkipEmpty(skip)
                r.render(); r.kernelMsTake()
                for _ in range(5):
                    r.render()
                assert r.last_kernel_name == TSLAB
                return r.kernelMsTake() / 5, r.readPixels().copy(), r.countSamples()
            t_plain, plain, n_plain = timed(False)
            t_skip, skipped, n_skip = timed(True)
            print(f"cfg3 TRILINEAR window [64,4095] {pose} variant {variant}: {t_plain:.3f} ms without, {t_skip:.3f} ms with empty-space skipping")
            assert n_plain == n_skip                        # logical samples are unchanged
            assert np.array_equal(bits(plain), bits(skipped)), (pose, variant)
            if variant in (6, 9):                           # (the oracle's rows once per pose)
                rows = [140, 333, 540, 771, 939]
                p = oracle.OracleParams(W, H, cam=r.getCameraBlock(), alpha_scale=0.004, min_val=64, max_val=4095, filter=1, threads=8)
                want = np.zeros((H, W, 4), dtype=np.float32)
                for y in rows:
                    p.row_begin, p.row_end = y, y + 1
                    oracle.render(vol, p, out=want)
                assert np.array_equal(bits(skipped[rows]), bits(want[rows])), pose
            # (the default pose's tiles see the ball's outside as whole empty layers; at the off-axis pose a tile's slanted layers mostly
            # touch the ball somewhere: measured -17 ... -25 % and -7 ... -9 %)
            # (the tiles on global taps, variant 7, skip per wavefront and batch: -11 % at the default pose)
            assert t_skip < (0.9 if pose == "default" and variant != 7 else 1.0) * t_plain, (pose, variant)


def test_skipping_on_a_long_axis_unstaged_oblique(vra, oracle):
    """round-5 advisor: the tiles on global taps (variant 7) probe the skip grid at positions from a closed form; on an axis of
    thousands of voxels its error must not grow with the ray's length (the probes are re-anchored every 16 batches).  A 4096-voxel
    axis crossed obliquely, blobs with empty space between them, every batch decision checked through the frame and the counts."""
    R = vra.renderer
    rng = np.random.default_rng(4096)
    # (spacings that fatten the short axes: the box is about 1 x 0.4 x 0.45, so rays ALONG the long axis take thousands of steps)
    for dims, dtype, spacing in (((4096, 40, 48), np.uint8, (1.0, 40.0, 38.0)), ((48, 2304, 40), np.uint16, (20.0, 1.0, 26.0))):
        hi = 255 if dtype == np.uint8 else 4095
        vol = blobs(rng, dims, dtype, 3)
        # the long axis repeated: blobs() spans the whole volume, so cut gaps of background every few hundred voxels
        ax = 2 - int(np.argmax(dims))
        idx = np.arange(vol.shape[ax])
        gap = ((idx // 160) % 2 == 1)
        sl = [slice(None)] * 3; sl[ax] = gap
        vol[tuple(sl)] = np.minimum(vol[tuple(sl)], 2)
        with vra.RendererCore(0) as r:
            r.setup((160, 96))
            assert r.loadShader("VolumeRenderer.cs")
            r.setQuirks(0)
            r.setVolume(vol, spacing)
            r.setFilter(R.FILTER_TRILINEAR); r.setWindow(3, hi); r.setAlpha(0.01)
            for name, (dz, da) in {"along_x": (0.15, (np.pi / 2) / 0.7 + 0.2), "along_y": ((np.pi / 2) / 0.7 - 0.25, 0.3), "oblique": (0.5, 0.9)}.items():
                c = oracle.Camera(); c.orient(0, dz, da)
                block = c.block()
                r.setCameraBlock(block)
                p = oracle.OracleParams(160, 96, cam=block, alpha_scale=0.01, voxel_size=spacing, min_val=3, max_val=hi, filter=1, threads=8)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                assert int(want_spp.max()) > 800, (name, int(want_spp.max()))       # rays do run along the long axis (thousands of steps at the matching pose)
                for variant in (7, 6, 0):
                    r.setKernelVariant(variant)
                    r.setSkipEmpty(True); r.render()
                    got = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    what = f"{dims} {np.dtype(dtype).name} camera {name} variant {variant} kernel {r.last_kernel_name}"
                    assert r.last_kernel_name == TSLAB, what
                    assert np.array_equal(spp, want_spp), what
                    assert np.array_equal(bits(got), bits(want)), what


def run_random_skip_trials(vra, oracle, seed, n_trials, log=None):
    """randomised trials of skipping on every TRILINEAR path (and, every fourth trial, on the NEAREST kernels): random blob volumes
    (dims, voxel type, spacing, background level), windows, opacities, modes, views, workgroup shapes, image sizes, cameras; each
    frame with skipping must equal the frame without it AND the oracle's, per-pixel counts included.  tools/stress_skip.py runs
    thousands.  Returns frames checked."""
    rng = np.random.default_rng(seed)
    R = vra.renderer
    dims_pool = [(64, 64, 64), (96, 80, 72), (130, 33, 47), (40, 56, 24), (24, 24, 160), (8, 8, 8), (200, 40, 56), (72, 64, 80)]
    n_checked = 0
    for trial in range(n_trials):
        dims = dims_pool[trial % len(dims_pool)]
        dtype = np.uint8 if trial % 3 else np.uint16
        hi = 255 if dtype == np.uint8 else 4095
        bg = int(rng.integers(0, 9)) * (1 if dtype == np.uint8 else 12)
        vol = blobs(rng, dims, dtype, bg)
        spacing = (1.0, 1.0, 1.0) if trial % 3 == 0 else tuple(np.round(rng.uniform(0.4, 2.2, size=3), 3).tolist())
        lo = int(rng.choice([0, bg, bg + 1, 2 * bg + 3])); hi_w = int(rng.integers(hi // 2, hi + 1))
        alpha = float(np.float32(rng.choice([1.0, 0.3, 0.03, 0.0])))
        W, H = int(rng.integers(24, 200)), int(rng.integers(24, 150))
        mode = int(rng.integers(0, 6))
        mip, tf = mode in (1, 3), mode in (2, 3)
        top, bottom = (trial % 7 == 3), (trial % 7 == 5)
        nearest = trial % 4 == 3
        variant = 0 if nearest else int(rng.choice([0, 6, 7, 8, 9, 10, 11]))
        with vra.RendererCore(0) as r:
            r.setup((W, H))
            assert r.loadShader("VolumeRenderer.cs")
            r.setQuirks(0)
            r.setVolume(vol, spacing)
            r.setFilter(R.FILTER_NEAREST if nearest else R.FILTER_TRILINEAR)
            r.setWindow(lo, hi_w); r.setAlpha(alpha); r.setMIP(mip)
            tf_lut = None
            if tf:
                z = int(rng.integers(1, 60))
                r.setTransferFunction([0, z, z + 40, 160, 255], [[0, 0, 0, 0], [0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])
                tf_lut = r.getTransferLut()
            if top or bottom:
                r.setInitialCameraRotation(top, bottom)
            r.setKernelVariant(variant)
            for name, block in camera_blocks(oracle, rng, n_random=2)[int(rng.integers(0, 3))::3]:
                r.setCameraBlock(block)
                r.setSkipEmpty(False); r.render()
                plain = r.readPixels()
                r.setSkipEmpty(True); r.render()
                got = r.readPixels()
                _, spp = r.countSamples(per_pixel=True)
                what = (f"seed {seed} trial {trial} dims {dims} {np.dtype(dtype).name} spacing {spacing} bg {bg} window [{lo},{hi_w}] alpha {alpha} {W}x{H} mip {mip} tf {tf} "
                        f"top {top} bottom {bottom} nearest {nearest} variant {variant} camera {name} kernel {r.last_kernel_name}")
                assert np.array_equal(bits(plain), bits(got)), what
                p = oracle.OracleParams(W, H, cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo, max_val=hi_w, is_mip=int(mip), view_top=int(top),
                                        view_bottom=int(bottom), tf_rgba=tf_lut, filter=0 if nearest else 1, threads=8)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                assert np.array_equal(spp, want_spp), what
                assert np.array_equal(bits(got), bits(want)), what
                n_checked += 1
        if log and (trial + 1) % 200 == 0:
            log(f"{trial + 1} trials, {n_checked} frames bit-exact")
    return n_checked


def test_randomised_skipping(vra, oracle):
    assert run_random_skip_trials(vra, oracle, 20260930, 48) > 100
