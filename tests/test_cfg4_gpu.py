"""BASELINE config 4 at its real workload: synthetic 2048^3 uint8 (8 GiB, > 2^32 voxels: 64-bit
offsets), 3840x2160, default alpha-spline transfer function + exact empty-space skipping
("adaptive step").  The reference cannot load this volume at all (32-bit `len`,
/root/reference/src/RendererCore.cpp:327,341,361), so the 64-bit path needs its own parity
evidence: specialised kernel == generic kernel bit for bit, skipping on == off, identical
sample counts, and sparse 4K rows against the CPU oracle.

Also here: every mode of the specialised kernels (grey / MIP / TF / MIP+TF x views x skipping)
on the > 4 GiB volume against full oracle frames at a small image size, so that the
64-bit-offset template instances all run under the checker.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 2048
W, H = 3840, 2160
ISO = [0, 141, 149, 255]                      # the widget's default alpha knots (AlphaControlSplineWidget.cpp:56-59)
RGBA = [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]]
WINDOW = (8, 255)                             # voxels <= 8 classify to zero: the corners outside the ball are skippable
ALPHA = 0.004
SPECIALISED = ("raymarch_fast_kernel", "raymarch_relay_kernel")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def cfg4(vra):
    R = vra.renderer
    r = vra.RendererCore(0)
    r.setup((W, H))
    assert r.loadShader("VolumeRenderer.cs")
    r.setQuirks(0)
    r.setLayout(R.LAYOUT_BRICKED)
    r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), 1, 0x9E3779B9)
    r.setWindow(*WINDOW)
    r.setAlpha(ALPHA)
    yield r
    r.close()


@pytest.fixture(scope="module")
def cfg4_host_volume(cfg4):
    vol = cfg4.readVolume()                    # 8 GiB over PCIe, once
    assert vol.shape == (N, N, N) and vol.dtype == np.uint8
    return vol


def test_cfg4_tf_skip_fast_equals_generic_and_skip_is_invisible(vra, cfg4):
    r = cfg4
    r.setTransferFunction(ISO, RGBA)
    try:
        frames, counts, kernels = {}, {}, {}
        for name, variant, skip in (("fast+skip", 0, True), ("fast", 0, False), ("generic", 1, False)):
            r.setKernelVariant(variant)
            r.setSkipEmpty(skip)
            r.render()
            kernels[name] = r.last_kernel_name
            frames[name] = r.readPixels().copy()
            counts[name] = r.countSamples()
        assert kernels["fast+skip"] in SPECIALISED and kernels["fast"] in SPECIALISED
        assert kernels["generic"] == "raymarch_generic_kernel"
        assert counts["fast+skip"] == counts["fast"] == counts["generic"]
        assert 3.0e9 < counts["fast"] < 3.841e9      # the spline's opacity terminates the densest rays early
        assert np.array_equal(bits(frames["fast+skip"]), bits(frames["fast"]))
        assert np.array_equal(bits(frames["fast"]), bits(frames["generic"]))
        assert frames["fast"][..., :3].max() > 0.05   # the transfer function produced a picture
    finally:
        r.setKernelVariant(0); r.setSkipEmpty(False); r.setTransferFunction()


def test_cfg4_grey_fast_equals_generic(vra, cfg4):
    r = cfg4
    r.setAlpha(0.0005)                         # deep regime: no ray reaches dest.a >= 0.95
    try:
        r.setKernelVariant(0); r.render()
        assert r.last_kernel_name in SPECIALISED
        fast = r.readPixels().copy()
        total, spp = r.countSamples(per_pixel=True)
        r.setKernelVariant(1); r.render()
        generic = r.readPixels().copy()
        assert np.array_equal(bits(fast), bits(generic))
        # image facts of SURVEY section 8: 31.3 % of the pixels hit the box, rows 274..1885
        assert abs((spp > 0).mean() - 0.313) < 0.002
        rows = np.nonzero((spp > 0).any(axis=1))[0]
        assert (rows[0], rows[-1]) == (274, 1885)
        # S of SURVEY section 8 for cfg4: 3.841e9 samples without early termination
        assert abs(total / 3.841e9 - 1.0) < 2e-3, total
    finally:
        r.setKernelVariant(0); r.setAlpha(ALPHA)


def test_cfg4_sparse_4k_rows_against_oracle(vra, oracle, cfg4, cfg4_host_volume):
    r = cfg4
    vol = cfg4_host_volume
    r.setTransferFunction(ISO, RGBA)
    r.setSkipEmpty(True)
    try:
        r.render()
        assert r.last_kernel_name in SPECIALISED
        got = r.readPixels()
        _, spp = r.countSamples(per_pixel=True)
        tf = r.getTransferLut()
        p = oracle.OracleParams(W, H, cam=r.getCameraBlock(), alpha_scale=ALPHA, min_val=WINDOW[0], max_val=WINDOW[1],
                                tf_rgba=tf, threads=1)
        want = np.zeros_like(got)
        for y in (274, 275, 600, 1079, 1080, 1500, 1885):      # first / last hit rows, the middle, in between
            p.row_begin, p.row_end = y, y + 1
            _, _, want_spp = oracle.render(vol, p, want_spp=True, out=want)
            assert np.array_equal(spp[y], want_spp[y]), f"cfg4 row {y}: fetch counts differ"
            assert np.array_equal(bits(got[y]), bits(want[y])), f"cfg4 row {y}: max|diff|={np.abs(got[y] - want[y]).max()}"
        assert not got[273].any() and not got[1886].any()       # SURVEY: hit rows 274..1885
        # the off-axis pose (zenith 60 deg, azimuth 45 deg)
        r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
        r.render()
        got = r.readPixels()
        p.cam = r.getCameraBlock()
        for y in (700, 1080, 1600):
            p.row_begin, p.row_end = y, y + 1
            oracle.render(vol, p, out=want)
            assert np.array_equal(bits(got[y]), bits(want[y])), f"cfg4 off-axis row {y}"
    finally:
        r.resetCamera(); r.setSkipEmpty(False); r.setTransferFunction()


@pytest.mark.parametrize("mode", ["grey", "mip", "tf", "mip_tf"])
def test_big_offsets_every_mode_view_and_skipping_against_oracle(vra, oracle, cfg4, cfg4_host_volume, mode):
    """the 64-bit-offset instances of the specialised kernels, mode x view x skipping, full
    frames against the oracle at 384x216 (the volume stays the 8 GiB one)"""
    vol = cfg4_host_volume
    R = vra.renderer
    mip, tf = "mip" in mode, "tf" in mode
    alpha = 0.3 if mip else 0.01
    w, h = 384, 216
    r = cfg4
    try:
        r.setup((w, h))                        # same handle (the volume stays resident), small target
        r.setMIP(mip)
        r.setAlpha(alpha)
        if tf:
            r.setTransferFunction(ISO, RGBA)
        tf_lut = r.getTransferLut() if tf else None
        for view in ("default", "top", "bottom"):
            r.setInitialCameraRotation(view == "top", view == "bottom")
            cams = [r.getCameraBlock()]
            r.cameraOrient(0.0, 0.06 * 9, 0.06 * 13)
            cams.append(r.getCameraBlock())
            for ci, block in enumerate(cams):
                r.setCameraBlock(block)
                p = oracle.OracleParams(w, h, cam=block, alpha_scale=alpha, min_val=WINDOW[0], max_val=WINDOW[1], is_mip=int(mip),
                                        view_top=int(view == "top"), view_bottom=int(view == "bottom"), tf_rgba=tf_lut, threads=8)
                want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
                for skip in (False, True):
                    r.setSkipEmpty(skip)
                    r.render()
                    assert r.last_kernel_name in SPECIALISED, (mode, view, skip)
                    got = r.readPixels()
                    total, spp = r.countSamples(per_pixel=True)
                    what = f"{mode} view {view} camera {ci} skip {skip} via {r.last_kernel_name}"
                    assert total == want_total, what
                    assert np.array_equal(spp, want_spp), what
                    assert np.array_equal(bits(got), bits(want)), f"{what}: max|diff|={np.abs(got - want).max()}"
    finally:
        cfg4.setSkipEmpty(False); cfg4.setMIP(False); cfg4.setTransferFunction(); cfg4.setAlpha(ALPHA)
        cfg4.setInitialCameraRotation(False, False)
        cfg4.setup((W, H))


def test_cfg4_trilinear_staged_kernel_beyond_4gib(vra, oracle, cfg4, cfg4_host_volume):
    """TRILINEAR on the 8 GiB volume (round 2: generic kernel only, >= 3 ms class): the LDS-staged kernel takes volumes
    beyond 32-bit offsets (its apron copy is 10 GiB, DMA addresses are 64-bit) in every mode.  Full 4K frame: staged ==
    generic bit for bit (grey composite and transfer function), equal sample counts; sparse rows against the oracle;
    small full frames of every mode against the oracle."""
    R = vra.renderer
    r = cfg4
    vol = cfg4_host_volume
    try:
        r.setFilter(R.FILTER_TRILINEAR)
        for tf in (False, True):
            if tf:
                r.setTransferFunction(ISO, RGBA)
            frames, counts, ms = {}, {}, {}
            for name, variant in (("staged", 0), ("generic", 1)):
                r.setKernelVariant(variant)
                r.render(); r.kernelMsTake()
                for _ in range(2):
                    r.render()
                ms[name] = r.kernelMsTake() / 2
                assert r.last_kernel_name == ("raymarch_tslab_kernel" if name == "staged" else "raymarch_generic_kernel"), (name, r.last_kernel_name)
                frames[name] = r.readPixels().copy()
                counts[name] = r.countSamples()
            print(f"cfg4 trilinear tf={tf}: staged {ms['staged']:.3f} ms, generic {ms['generic']:.3f} ms, apron copy {r.trilinearCopyBytes() / 2**30:.1f} GiB")
            assert counts["staged"] == counts["generic"]
            assert np.array_equal(bits(frames["staged"]), bits(frames["generic"])), tf
        r.setKernelVariant(0)
        tf_lut = r.getTransferLut()
        p = oracle.OracleParams(W, H, cam=r.getCameraBlock(), alpha_scale=ALPHA, min_val=WINDOW[0], max_val=WINDOW[1], tf_rgba=tf_lut, filter=1, threads=1)
        want = np.zeros((H, W, 4), dtype=np.float32)
        got = frames["staged"]
        for y in (274, 1080, 1885):
            p.row_begin, p.row_end = y, y + 1
            oracle.render(vol, p, out=want)
            assert np.array_equal(bits(got[y]), bits(want[y])), f"cfg4 trilinear row {y}: max|diff|={np.abs(got[y] - want[y]).max()}"
        # every mode at a small size, two cameras, against full oracle frames
        w, h = 384, 216
        r.setup((w, h))
        for mode in ("grey", "mip", "tf", "mip_tf"):
            mip, tf = "mip" in mode, "tf" in mode
            alpha = 0.3 if mip else 0.01
            r.setMIP(mip); r.setAlpha(alpha)
            if tf:
                r.setTransferFunction(ISO, RGBA)
            else:
                r.setTransferFunction()
            lut = r.getTransferLut() if tf else None
            r.resetCamera()
            cams = [r.getCameraBlock()]
            r.cameraOrient(0.0, 0.06 * 9, 0.06 * 13)
            cams.append(r.getCameraBlock())
            for ci, block in enumerate(cams):
                r.setCameraBlock(block)
                r.render()
                assert r.last_kernel_name == "raymarch_tslab_kernel", (mode, ci)
                got = r.readPixels()
                total, spp = r.countSamples(per_pixel=True)
                p = oracle.OracleParams(w, h, cam=block, alpha_scale=alpha, min_val=WINDOW[0], max_val=WINDOW[1], is_mip=int(mip), tf_rgba=lut, filter=1, threads=8)
                want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
                assert total == want_total and np.array_equal(spp, want_spp), (mode, ci)
                assert np.array_equal(bits(got), bits(want)), f"{mode} camera {ci}: max|diff|={np.abs(got - want).max()}"
    finally:
        r.setFilter(R.FILTER_NEAREST); r.setKernelVariant(0); r.setMIP(False); r.setTransferFunction(); r.setAlpha(ALPHA)
        r.resetCamera(); r.setup((W, H))
