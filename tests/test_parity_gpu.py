"""GPU parity: the HIP ray-march (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): <= 1e-4 per channel.  The build's own contract is
stricter -- identical fp32 operation order -- so these tests ALSO assert bit-exact
frames and identical per-pixel fetch counts; TOL is the fallback bar quoted in
messages.  Everything here calls libvr_core.so; the oracle is only the checker.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4
FAST_KERNELS = ("raymarch_fast_kernel", "raymarch_relay_kernel")   # relay = sparse launches


def make_renderer(vra, size, **kw):
    r = vra.RendererCore(0)
    r.setup(size)
    assert r.loadShader("VolumeRenderer.cs")
    return r


def assert_same(got, want, got_spp=None, want_spp=None, exact=True, what=""):
    err = float(np.max(np.abs(got - want))) if got.size else 0.0
    assert np.isfinite(got).all(), what
    assert err <= TOL, f"{what}: max|diff|={err}"
    if exact:
        bad = got.view(np.uint32) != want.view(np.uint32)
        assert not bad.any(), f"{what}: {int(bad.sum())} words differ bitwise (max|diff|={err})"
    if got_spp is not None:
        assert np.array_equal(got_spp, want_spp), f"{what}: per-pixel fetch counts differ"


def rand_volume(rng, dims, dtype, smooth=False):
    nx, ny, nz = dims
    hi = 256 if dtype == np.uint8 else 4096
    v = rng.integers(0, hi, size=(nz, ny, nx), dtype=np.int64)
    if smooth:
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        v = ((np.sin(x * 0.3) + np.cos(y * 0.23) + np.sin(z * 0.31) + 3) / 6 * (hi - 1)).astype(np.int64)
    return v.astype(dtype)


def orbit_blocks(oracle):
    blocks = [("default", oracle.default_camera_block())]
    c = oracle.Camera()
    c.orient(0, 0.06 * 7, 0.06 * 9)
    blocks.append(("orbit_a", c.block()))
    c.orient(0, -0.06 * 15, 0.06 * 31)
    blocks.append(("orbit_b", c.block()))
    c = oracle.Camera()
    c.orient(0, 0.0, -0.06 * 5)          # negative azimuth: exercises the Q13 wrap
    blocks.append(("orbit_neg", c.block()))
    c = oracle.Camera()
    c.orient(0, -100.0, 0.3)             # zenith clamped to 0: straight down the pole
    blocks.append(("pole", c.block()))
    return blocks


def test_cfg0_sphere_default_camera(vra, oracle):
    """BASELINE config 0: 64^3 u8 sphere, 256x256."""
    vol = oracle.gen_sphere_u8(64, 28)
    with make_renderer(vra, (256, 256)) as r:
        r.setVolume(vol)
        r.render()
        got = r.readPixels()
        total, spp = r.countSamples(per_pixel=True)
        assert r.last_kernel_name in FAST_KERNELS
        assert r.kernelMsTake() > 0.0
    want, want_total, want_spp = oracle.render(vol, oracle.OracleParams(256, 256), want_spp=True)
    assert total == want_total
    assert_same(got, want, spp, want_spp, what="cfg0")
    # non-hit pixels are exactly (0,0,0,0) (Q12)
    assert not got[0, 0].any() and not got[255, 255].any()


@pytest.mark.parametrize("variant", [0, 1, 5], ids=["auto", "generic", "pipelined"])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
def test_noncubic_spacing_orbits(vra, oracle, dtype, variant):
    """non-cubic dims + anisotropic spacing + odd image + orbit cameras"""
    rng = np.random.default_rng(7)
    dims = (40, 56, 24)
    spacing = (1.0, 0.8, 1.7)
    vol = rand_volume(rng, dims, dtype)
    lo, hi = (10, 200) if dtype == np.uint8 else (100, 3000)
    with make_renderer(vra, (173, 131)) as r:
        r.setQuirks(0)
        r.setKernelVariant(variant)
        r.setVolume(vol, spacing)
        r.setWindow(lo, hi)
        for alpha in (1.0, 0.05):
            r.setAlpha(alpha)
            for name, block in orbit_blocks(oracle):
                r.setCameraBlock(block)
                r.render()
                got = r.readPixels()
                _, spp = r.countSamples(per_pixel=True)
                p = oracle.OracleParams(173, 131, cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo, max_val=hi)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                assert_same(got, want, spp, want_spp, what=f"{name} alpha={alpha}")


def test_camera_orbit_through_capi_matches_oracle_camera(vra, oracle):
    """vr_camera_orient drives the same block the oracle's Camera restatement produces"""
    with make_renderer(vra, (64, 64)) as r:
        c = oracle.Camera()
        for (zoom, dz, da) in [(0, 0.06, 0.06), (0, 0.06, -0.06), (1, 0, 0), (0, -0.12, 0.3), (-1, 0, 0), (0, -50, 0.1)]:
            r.cameraOrient(zoom, dz, da)
            c.orient(zoom, dz, da)
            assert np.array_equal(r.getCameraBlock().view(np.uint32), c.block().view(np.uint32))


@pytest.mark.parametrize("mode", ["mip", "top", "bottom", "mip_top"])
def test_mip_and_view_swizzles(vra, oracle, mode):
    rng = np.random.default_rng(11)
    dims = (33, 47, 29)
    vol = rand_volume(rng, dims, np.uint8, smooth=True)
    vol[3, 5, 7] = 255   # a bright voxel off-centre: orientation / z-flip sensitive
    mip = "mip" in mode
    top = "top" in mode
    bottom = mode == "bottom"
    with make_renderer(vra, (96, 80)) as r:
        r.setVolume(vol, (1.0, 1.0, 1.3))
        r.setMIP(mip)
        r.setInitialCameraRotation(top, bottom)
        r.setAlpha(0.6)
        r.render()
        assert r.last_kernel_name in FAST_KERNELS      # MIP and the view swizzles have fast variants
        got = r.readPixels()
        _, spp = r.countSamples(per_pixel=True)
    p = oracle.OracleParams(96, 80, alpha_scale=0.6, voxel_size=(1.0, 1.0, 1.3), is_mip=int(mip), view_top=int(top),
                            view_bottom=int(bottom))
    want, _, want_spp = oracle.render(vol, p, want_spp=True)
    assert_same(got, want, spp, want_spp, what=mode)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
def test_trilinear_filter(vra, oracle, dtype):
    rng = np.random.default_rng(3)
    dims = (31, 20, 45)
    vol = rand_volume(rng, dims, dtype)
    lo, hi = (0, 255) if dtype == np.uint8 else (0, 4095)
    with make_renderer(vra, (120, 90)) as r:
        r.setQuirks(0)
        r.setVolume(vol)
        r.setWindow(lo, hi)
        r.setFilter(1)
        r.setAlpha(0.1)
        for name, block in orbit_blocks(oracle)[:3]:
            r.setCameraBlock(block)
            r.render()
            got = r.readPixels()
            p = oracle.OracleParams(120, 90, cam=block, alpha_scale=0.1, min_val=lo, max_val=hi, filter=1)
            want, _ = oracle.render(vol, p)
            assert_same(got, want, what=f"trilinear {name}")


def test_closed_form_accumulation(vra, oracle):
    rng = np.random.default_rng(5)
    vol = rand_volume(rng, (48, 48, 48), np.uint8)
    with make_renderer(vra, (100, 100)) as r:
        r.setVolume(vol)
        r.setAccum(1)
        r.setAlpha(0.03)
        r.render()
        got = r.readPixels()
        _, spp = r.countSamples(per_pixel=True)
    want, _, want_spp = oracle.render(vol, oracle.OracleParams(100, 100, alpha_scale=0.03, accum=1), want_spp=True)
    assert_same(got, want, spp, want_spp, what="closed form")


def test_transfer_function(vra, oracle):
    rng = np.random.default_rng(9)
    vol = rand_volume(rng, (36, 36, 36), np.uint8, smooth=True)
    iso = [0, 141, 149, 255]   # the widget's default alpha knots (AlphaControlSplineWidget.cpp:56-59)
    rgba = [[0, 0, 0, 0], [0.3, 0.5, 0.1, 0.759], [0.8, 0.2, 0.4, 0.45], [1, 1, 1, 1]]
    with make_renderer(vra, (90, 70)) as r:
        r.setVolume(vol)
        r.setTransferFunction(iso, rgba)
        lut = r.getTransferLut()
        r.setAlpha(0.2)
        r.render()
        assert r.last_kernel_name in FAST_KERNELS      # LDS-resident RGBA classification table
        got = r.readPixels()
        r.setKernelVariant(1)
        r.render()
        assert np.array_equal(got.view(np.uint32), r.readPixels().view(np.uint32))
        r.setMIP(True)                                           # MIP through the transfer function (MODE 3)
        r.setKernelVariant(0)
        r.render()
        assert r.last_kernel_name in FAST_KERNELS
        mip_fast = r.readPixels()
        r.setKernelVariant(1)
        r.render()
        assert r.last_kernel_name == "raymarch_generic_kernel"
        assert np.array_equal(mip_fast.view(np.uint32), r.readPixels().view(np.uint32))
        want_mip, _ = oracle.render(vol, oracle.OracleParams(90, 70, alpha_scale=0.2, tf_rgba=lut, is_mip=1))
        assert_same(mip_fast, want_mip, what="MIP through the transfer function")
    want_lut = oracle.spline_tf(iso, rgba)
    assert np.array_equal(lut.view(np.uint32), want_lut.view(np.uint32))
    want, _ = oracle.render(vol, oracle.OracleParams(90, 70, alpha_scale=0.2, tf_rgba=want_lut))
    assert_same(got, want, what="transfer function")
    assert not np.array_equal(got[..., 0], got[..., 1])   # colour TF really applied


def test_bricked_layout_is_invisible(vra, oracle):
    rng = np.random.default_rng(13)
    dims = (37, 50, 23)          # not multiples of the 4^3 brick
    vol = rand_volume(rng, dims, np.uint16)
    with make_renderer(vra, (110, 77)) as r:
        r.setQuirks(0)
        r.setLayout(1)
        r.setVolume(vol, (1.0, 1.1, 0.9))
        r.setWindow(0, 4095)
        assert np.array_equal(r.readVolume(), vol)
        block = orbit_blocks(oracle)[1][1]
        r.setCameraBlock(block)
        r.render()
        bricked = r.readPixels()
        r.setLayout(0)                      # re-lay out in place
        assert np.array_equal(r.readVolume(), vol)
        r.render()
        linear = r.readPixels()
    p = oracle.OracleParams(110, 77, cam=block, voxel_size=(1.0, 1.1, 0.9), min_val=0, max_val=4095)
    want, _ = oracle.render(vol, p)
    assert_same(bricked, want, what="bricked")
    assert_same(linear, want, what="linear after relayout")


def test_quirks_trunc_grid_and_u16_offset(vra, oracle):
    rng = np.random.default_rng(17)
    vol = rand_volume(rng, (32, 32, 32), np.uint16)
    with make_renderer(vra, (75, 53)) as r:      # 75 = 4*16+11, 53 = 3*16+5
        r.setVolume(vol)
        assert r.dataset_range == (int(vol.min()), int(vol.max()))
        assert r.window == r.dataset_range       # default window = dataset range (RendererCore.cpp:375-378)
        r.setQuirks(vra.renderer.QUIRK_TRUNC_GRID | vra.renderer.QUIRK_U16_OFFSET)
        assert r.workgroups == (4, 3)
        r.setWindow(-500, 2000)
        r.render()
        got = r.readPixels()
    p = oracle.OracleParams(75, 53, min_val=500, max_val=3000, trunc_grid=1)   # +1000 (Q10)
    want, _ = oracle.render(vol, p)
    assert_same(got, want, what="quirks")
    assert not got[:, 64:].any() and not got[48:, :].any()   # never written (Q1)


def test_degenerate_window_and_inside_camera(vra, oracle):
    rng = np.random.default_rng(19)
    vol = rand_volume(rng, (24, 24, 24), np.uint8)
    with make_renderer(vra, (64, 64)) as r:
        r.setVolume(vol)
        # Q4: max == min -> defined as 0 contribution everywhere
        r.setWindow(77, 77)
        r.render()
        got = r.readPixels()
        want, _ = oracle.render(vol, oracle.OracleParams(64, 64, min_val=77, max_val=77))
        assert_same(got, want, what="min==max")
        assert not got.any()
        # Q5: camera dollied into the box (3 zoom-ins put the eye at the origin)
        r.setWindow(0, 255)
        r.setAlpha(0.02)
        c = oracle.Camera()
        for _ in range(3):
            r.cameraOrient(1, 0, 0)
            c.orient(1, 0, 0)
        r.render()
        got = r.readPixels()
        _, spp = r.countSamples(per_pixel=True)
        want, _, want_spp = oracle.render(vol, oracle.OracleParams(64, 64, cam=c.block(), alpha_scale=0.02), want_spp=True)
        assert_same(got, want, spp, want_spp, what="inside camera")


def test_row_shards_and_stripes_compose_to_full_frame(vra, oracle):
    """multi-GPU decomposition: shards rendered separately == the full frame"""
    vol = oracle.gen_noise_ball((48, 40, 44), 2, 0x9E3779B9)
    W, H = 133, 101
    with make_renderer(vra, (W, H)) as r:
        r.setQuirks(0)
        r.setVolume(vol)
        r.setWindow(0, 4095)
        r.setAlpha(0.05)
        r.render()
        full = r.readPixels()
        want, _ = oracle.render(vol, oracle.OracleParams(W, H, alpha_scale=0.05, min_val=0, max_val=4095))
        assert_same(full, want, what="full")
        # contiguous row blocks, 3 ranks with a ragged last block
        acc = np.zeros_like(full)
        bounds = [0, 34, 68, H]
        for k in range(3):
            r.setRowRange(bounds[k], bounds[k + 1])
            r.render()
            part = r.readPixels()
            acc[bounds[k]:bounds[k + 1]] = part[bounds[k]:bounds[k + 1]]
        assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
        r.setRowRange(0, -1)
    # cyclic stripes of 8 rows over 4 "ranks", each rank its own zeroed target
    acc = np.zeros_like(full)
    for k in range(4):
        with make_renderer(vra, (W, H)) as r:
            r.setQuirks(0)
            r.setVolume(vol)
            r.setWindow(0, 4095)
            r.setAlpha(0.05)
            r.setRowStripes(8, k, 4)
            r.render()
            part = r.readPixels()
            rows = np.array([y for y in range(H) if (y // 8) % 4 == k])
            other = np.array([y for y in range(H) if (y // 8) % 4 != k])
            assert not part[other].any()
            acc[rows] = part[rows]
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))


def test_device_generators_match_oracle_generators(vra, oracle):
    with make_renderer(vra, (32, 32)) as r:
        r.generateSynthetic(vra.renderer.SYNTH_SPHERE_U8, (64, 64, 64), 1, 28)
        assert np.array_equal(r.readVolume(), oracle.gen_sphere_u8(64, 28))
        for bpv in (1, 2):
            r.generateSynthetic(vra.renderer.SYNTH_NOISE_BALL, (70, 45, 52), bpv, 0x9E3779B9)
            assert np.array_equal(r.readVolume(), oracle.gen_noise_ball((70, 45, 52), bpv, 0x9E3779B9))
        r.setLayout(1)
        r.generateSynthetic(vra.renderer.SYNTH_NOISE_BALL, (70, 45, 52), 2, 123)
        assert np.array_equal(r.readVolume(), oracle.gen_noise_ball((70, 45, 52), 2, 123))


def test_histogram_matches_reference_formula(vra, oracle):
    rng = np.random.default_rng(23)
    vol = rand_volume(rng, (20, 30, 25), np.uint8)
    with make_renderer(vra, (32, 32)) as r:
        r.setVolume(vol)
        h = r.histogram()
    counts = np.bincount(vol.ravel(), minlength=256).astype(np.float64)
    counts[0] = 0
    want = (counts.astype(np.float32) * np.float32(100.0) / np.float32(counts.max())).astype(np.float32)
    assert np.allclose(h, want, rtol=1e-6, atol=0)


def test_render_requires_shader_and_dataset(vra):
    r = vra.RendererCore(0)
    r.setup((32, 32))
    with pytest.raises(vra.VRError):
        r.render()
    r.loadShader("VolumeRenderer.cs")
    with pytest.raises(vra.VRError):
        r.render()
    r.close()


# ---------------------------------------------------------------------------------------
# BASELINE.json's full sizes: size-independent properties + sparse oracle rows
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg3(vra):
    """config 3: synthetic 1024^3 uint16 generated in HBM, 1920x1080"""
    r = vra.RendererCore(0)
    r.setup((1920, 1080))
    r.loadShader("VolumeRenderer.cs")
    r.setQuirks(0)
    r.setLayout(vra.renderer.LAYOUT_BRICKED)
    r.generateSynthetic(vra.renderer.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
    r.setWindow(0, 4095)
    r.setAlpha(0.004)
    yield r
    r.close()


def test_cfg3_full_size_fast_equals_generic_and_layouts_agree(vra, cfg3):
    r = cfg3
    r.render()
    assert r.last_kernel_name in FAST_KERNELS
    fast = r.readPixels()
    total = r.countSamples()
    assert total == 480301385                     # S of BASELINE.md section 2 (4.803e8)
    r.setKernelVariant(1)
    r.render()
    assert r.last_kernel_name == "raymarch_generic_kernel"
    generic = r.readPixels()
    r.setKernelVariant(5)                         # the fast kernel's software-pipelined loop (under-filled launches use it)
    r.render()
    assert r.last_kernel_name == "raymarch_fast_kernel"
    assert np.array_equal(fast.view(np.uint32), r.readPixels().view(np.uint32))
    assert r.countSamples() == total
    r.setKernelVariant(0)
    assert np.array_equal(fast.view(np.uint32), generic.view(np.uint32))
    r.setLayout(vra.renderer.LAYOUT_LINEAR)       # re-bricking is invisible
    r.render()
    linear = r.readPixels()
    r.setLayout(vra.renderer.LAYOUT_BRICKED)
    assert np.array_equal(fast.view(np.uint32), linear.view(np.uint32))
    # image facts of SURVEY section 8: 31.3 % of the pixels hit the box, rows 137..942
    hit = fast[..., 3] > 0
    assert abs(hit.mean() - 0.313) < 0.002
    rows = np.nonzero(hit.any(axis=1))[0]
    assert (rows[0], rows[-1]) == (137, 942)


def test_cfg3_full_size_row_shards_compose(vra, cfg3):
    r = cfg3
    r.render()
    full = r.readPixels()
    acc = np.zeros_like(full)
    for k in range(8):                             # 8 contiguous 135-row blocks
        r.setRowRange(135 * k, 135 * (k + 1))
        r.render()
        acc[135 * k:135 * (k + 1)] = r.readPixels()[135 * k:135 * (k + 1)]
    r.setRowRange(0, -1)
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
    r.setRowStripes(16, 3, 8)                      # rank 3 of 8, 16-row stripes
    r.render()
    part = r.readPixels()
    r.setRowStripes(1, 0, 1)
    rows = np.array([y for y in range(1080) if (y // 16) % 8 == 3])
    assert np.array_equal(part[rows].view(np.uint32), full[rows].view(np.uint32))


def test_cfg3_full_size_sparse_rows_against_oracle(vra, oracle, cfg3):
    r = cfg3
    r.render()
    got = r.readPixels()
    _, spp = r.countSamples(per_pixel=True)
    vol = r.readVolume()
    p = oracle.OracleParams(1920, 1080, cam=r.getCameraBlock(), alpha_scale=0.004, min_val=0, max_val=4095, threads=8)
    want = np.zeros_like(got)
    for y in (137, 200, 411, 539, 540, 777, 942):
        p.row_begin, p.row_end = y, y + 1
        _, _, want_spp = oracle.render(vol, p, want_spp=True, out=want)
        assert_same(got[y], want[y], spp[y], want_spp[y], what=f"cfg3 row {y}")
    # an off-axis pose (zenith 60 deg, azimuth 45 deg): layout-sensitive, same contract
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    r.render()
    got = r.readPixels()
    p.cam = r.getCameraBlock()
    for y in (300, 540, 801):
        p.row_begin, p.row_end = y, y + 1
        oracle.render(vol, p, out=want)
        assert_same(got[y], want[y], what=f"cfg3 off-axis row {y}")
    r.resetCamera()


def test_cfg2_shape_u16_window_ert_nonpow2(vra, oracle):
    """config 2 shape (512x512x452 uint16, 1920x1080, ERT + data window) on sparse rows;
    452 is not a power of two and the box is not a cube: certified-division variant"""
    dims = (512, 512, 452)
    with make_renderer(vra, (1920, 1080)) as r:
        r.setQuirks(vra.renderer.QUIRK_U16_OFFSET)          # reference default: +1000 on the window
        r.generateSynthetic(vra.renderer.SYNTH_NOISE_BALL, dims, 2, 12345)
        lo, hi = r.dataset_range
        assert r.window == (lo, hi)
        r.setAlpha(0.05)
        r.render()
        assert r.last_kernel_name in FAST_KERNELS
        got = r.readPixels()
        _, spp = r.countSamples(per_pixel=True)
        vol = r.readVolume()
        block = r.getCameraBlock()
    p = oracle.OracleParams(1920, 1080, cam=block, alpha_scale=0.05, min_val=lo + 1000, max_val=hi + 1000, threads=8)
    want = np.zeros_like(got)
    for y in (150, 400, 540, 700, 930):
        p.row_begin, p.row_end = y, y + 1
        _, _, want_spp = oracle.render(vol, p, want_spp=True, out=want)
        assert_same(got[y], want[y], spp[y], want_spp[y], what=f"cfg2 row {y}")


def test_cfg4_shape_u8_beyond_4gib_offsets(vra, oracle):
    """config 4 addressing: more than 2^32 voxels (64-bit offsets), reduced to one axis so
    the test stays small in time: 2048 x 2048 x 1100 uint8 = 4.6e9 voxels"""
    dims = (2048, 2048, 1100)
    with make_renderer(vra, (640, 360)) as r:
        r.setLayout(vra.renderer.LAYOUT_BRICKED)
        r.generateSynthetic(vra.renderer.SYNTH_NOISE_BALL, dims, 1, 99)
        r.setAlpha(0.002)
        r.render()
        assert r.last_kernel_name in FAST_KERNELS
        got = r.readPixels()
        r.setKernelVariant(1)
        r.render()
        generic = r.readPixels()
        assert np.array_equal(got.view(np.uint32), generic.view(np.uint32))
        vol = r.readVolume()
        block = r.getCameraBlock()
    p = oracle.OracleParams(640, 360, cam=block, alpha_scale=0.002, threads=8)
    want = np.zeros_like(got)
    for y in (60, 180, 300):
        p.row_begin, p.row_end = y, y + 1
        oracle.render(vol, p, out=want)
        assert_same(got[y], want[y], what=f"cfg4 row {y}")


def test_full_size_mip_and_transfer_function_fast_paths(vra, oracle, cfg3):
    """cfg3-size MIP and TF frames: fast kernel == generic kernel, sparse rows == oracle"""
    r = cfg3
    vol = None
    iso = [0, 141, 149, 255]
    rgba = [[0, 0, 0, 0], [0.9, 0.4, 0.1, 0.759], [0.2, 0.6, 0.9, 0.45], [1, 1, 1, 1]]
    for mode in ("mip", "tf"):
        r.setMIP(mode == "mip")
        r.setTransferFunction(iso, rgba) if mode == "tf" else r.setTransferFunction()
        r.setAlpha(0.3 if mode == "mip" else 0.004)
        if mode == "tf":
            r.setWindow(0, 2047)                    # 2048-entry RGBA table fits LDS
        r.render()
        assert r.last_kernel_name in FAST_KERNELS, mode
        fast = r.readPixels()
        r.setKernelVariant(1)
        r.render()
        generic = r.readPixels()
        r.setKernelVariant(0)
        assert np.array_equal(fast.view(np.uint32), generic.view(np.uint32)), mode
        if vol is None:
            vol = r.readVolume()
        p = oracle.OracleParams(1920, 1080, cam=r.getCameraBlock(), alpha_scale=0.3 if mode == "mip" else 0.004,
                                min_val=0, max_val=4095 if mode == "mip" else 2047, is_mip=int(mode == "mip"),
                                tf_rgba=oracle.spline_tf(iso, rgba) if mode == "tf" else None, threads=8)
        want = np.zeros_like(fast)
        for y in (300, 540):
            p.row_begin, p.row_end = y, y + 1
            oracle.render(vol, p, out=want)
            assert_same(fast[y], want[y], what=f"cfg3 {mode} row {y}")
    r.setMIP(False); r.setTransferFunction(); r.setWindow(0, 4095); r.setAlpha(0.004)


# ---------------------------------------------------------------------------------------
# exact empty-space skipping (config 4's "adaptive step"): never changes a bit
# ---------------------------------------------------------------------------------------
def test_skip_empty_is_bit_exact_small(vra, oracle):
    vol8 = oracle.gen_sphere_u8(96, 30)                 # a ball in a sea of zeros
    vol16 = oracle.gen_noise_ball((80, 96, 72), 2, 7)   # noise 0..63 outside the ball
    iso = [0, 60, 149, 255]
    rgba = [[0, 0, 0, 0], [0.0, 0.0, 0.0, 0.0], [0.8, 0.2, 0.4, 0.45], [1, 1, 1, 1]]
    cases = [
        ("u8 grey", vol8, dict(min_val=0, max_val=255), dict()),
        ("u8 window", vol8, dict(min_val=40, max_val=200), dict()),
        ("u16 window", vol16, dict(min_val=64, max_val=4095), dict()),
        ("u16 mip", vol16, dict(min_val=64, max_val=4095, is_mip=1), dict(mip=True)),
        ("u8 tf", vol8, dict(min_val=0, max_val=255), dict(tf=True)),
        ("u8 aniso (skipping must disable itself)", vol8, dict(min_val=0, max_val=255, voxel_size=(1.0, 1.0, 4.0)), dict()),
    ]
    for name, vol, okw, opt in cases:
        with make_renderer(vra, (150, 110)) as r:
            r.setQuirks(0)
            r.setLayout(vra.renderer.LAYOUT_BRICKED)
            r.setVolume(vol, okw.get("voxel_size", (1.0, 1.0, 1.0)))
            r.setWindow(okw["min_val"], okw["max_val"])
            r.setMIP(opt.get("mip", False))
            if opt.get("tf"):
                r.setTransferFunction(iso, rgba)
            r.setAlpha(0.3)
            for cname, block in orbit_blocks(oracle)[:4]:
                r.setCameraBlock(block)
                r.setSkipEmpty(False)
                r.render()
                assert r.last_kernel_name in FAST_KERNELS, name
                plain = r.readPixels()
                r.setSkipEmpty(True)
                r.render()
                skipped = r.readPixels()
                _, spp = r.countSamples(per_pixel=True)
                assert np.array_equal(plain.view(np.uint32), skipped.view(np.uint32)), (name, cname)
                p = oracle.OracleParams(150, 110, cam=block, alpha_scale=0.3, **okw)
                if opt.get("tf"):
                    p.tf_rgba = oracle.spline_tf(iso, rgba)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                assert_same(skipped, want, spp, want_spp, what=f"skip {name} {cname}")


def test_skip_empty_full_size_windowed(vra, oracle, cfg3):
    """cfg3 volume with the window raised above the background noise: everything outside the
    ball classifies to zero and is skipped; the frame must not change by a bit"""
    r = cfg3
    r.setWindow(64, 4095)
    r.setAlpha(0.004)
    r.setSkipEmpty(False)
    r.render(); r.kernelMsTake()
    for _ in range(5):
        r.render()
    t_plain = r.kernelMsTake() / 5
    plain = r.readPixels()
    r.setSkipEmpty(True)
    r.render(); r.kernelMsTake()
    for _ in range(5):
        r.render()
    t_skip = r.kernelMsTake() / 5
    skipped = r.readPixels()
    total = r.countSamples()
    r.setSkipEmpty(False)
    r.setWindow(0, 4095)
    assert np.array_equal(plain.view(np.uint32), skipped.view(np.uint32))
    assert total == 480301385                       # logical samples are unchanged
    print(f"cfg3 window [64,4095]: {t_plain:.3f} ms without, {t_skip:.3f} ms with empty-space skipping")
    assert t_skip < t_plain


def _random_camera_block(rng, radius_lo=0.2, radius_hi=4.0):
    """random eye + orthonormal basis looking roughly at the box (also from inside / grazing)"""
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    eye = d * rng.uniform(radius_lo, radius_hi)
    target = rng.uniform(-0.45, 0.45, size=3)
    look = target - eye; look /= np.linalg.norm(look)
    up0 = rng.normal(size=3)
    side = np.cross(look, up0); side /= np.linalg.norm(side)
    up = np.cross(side, look)
    b = np.zeros(21, dtype=np.float32)
    b[0:3] = side; b[4:7] = up; b[8:11] = -look; b[12:15] = eye; b[15] = 1
    b[16:19] = eye; b[19] = 1
    b[20] = rng.uniform(1.0, 5.0)          # view_plane_dist (FOV 22..90 degrees)
    return b


def test_packed12_copy_is_lossless_and_only_used_when_the_data_allow(vra, oracle):
    """vr_set_pack12: a bricked u16 volume whose voxels are all <= 4095 is gathered from a 12-bit
    packed copy (unaligned 2-byte loads, some straddling cache lines); frames must equal the
    oracle and the unpacked path bit for bit.  Volumes with a larger voxel never use it."""
    rng = np.random.default_rng(12)
    R = vra.renderer
    for dims in ((64, 64, 64), (37, 21, 50), (128, 8, 16), (5, 3, 2)):
        vol = rand_volume(rng, dims, np.uint16)                  # values 0..4095: eligible
        vol.flat[0] = 4095; vol.flat[-1] = 0
        frames = {}
        for pack in (1, 0):
            with make_renderer(vra, (96, 80)) as r:
                r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED); r.setPack12(pack)
                r.setVolume(vol); r.setWindow(0, 4095); r.setAlpha(0.05)
                r.cameraOrient(0.0, -0.4, 0.9)
                block = r.getCameraBlock()
                r.render()
                frames[pack] = r.readPixels()
                assert (r.pack12Bytes() > 0) == bool(pack), (dims, pack)     # 1.5 bytes per storage voxel, or not used
                _, spp = r.countSamples(per_pixel=True)
                assert r.last_kernel_name in FAST_KERNELS
        want, _, want_spp = oracle.render(vol, oracle.OracleParams(96, 80, cam=block, alpha_scale=0.05, min_val=0, max_val=4095), want_spp=True)
        assert_same(frames[1], want, spp, want_spp, what=f"packed12 dims {dims}")
        assert np.array_equal(frames[1].view(np.uint32), frames[0].view(np.uint32)), dims
    # one voxel above 4095: the packed copy must not be used (it would truncate)
    vol = rand_volume(rng, (32, 32, 32), np.uint16)
    vol[7, 9, 11] = 60000
    with make_renderer(vra, (64, 64)) as r:
        r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED); r.setVolume(vol); r.setWindow(0, 65535); r.setAlpha(0.3)
        r.render()
        got = r.readPixels()
        assert r.pack12Bytes() == 0
    want, _ = oracle.render(vol, oracle.OracleParams(64, 64, alpha_scale=0.3, min_val=0, max_val=65535))
    assert_same(got, want, what="u16 volume with a voxel > 4095")


def test_packed12_copy_takes_any_offset_when_the_range_fits_12_bits(vra, oracle):
    """Round 5: eligibility is max - min <= 4095, not max <= 4095 -- CT data stored with the reference's own +1000
    convention (src/RendererCore.cpp:66-67) or any other offset packs as (voxel - min).  Clamping windows, windows wider
    than the data (no clamp), a window too wide for the LDS table (float classification), MIP, a transfer function, the
    relay kernel, and the +1000 quirk itself: all bit-exact against the oracle and against the unpacked path."""
    rng = np.random.default_rng(512)
    R = vra.renderer
    for dims, base in (((64, 64, 64), 1000), ((37, 21, 50), 61440), ((40, 48, 24), 17)):
        vol = (rand_volume(rng, dims, np.uint16, smooth=True).astype(np.int64) + base).astype(np.uint16)
        vol.flat[0] = base; vol.flat[-1] = base + 4095
        cases = [dict(win=(base + 200, base + 3000)), dict(win=(max(base - 500, 0), min(base + 4500, 65535))),
                 dict(win=(0, 65535), alpha=0.4), dict(win=(base + 100, base + 3500), mip=1, alpha=0.6),
                 dict(win=(base, base + 4095), tf=True), dict(win=(base + 200, base + 3000), variant=3)]
        if base == 1000:
            cases.append(dict(win=(0, 4095), quirks=R.QUIRK_U16_OFFSET))       # the GUI's window 0 .. 4095 uploaded as 1000 .. 5095
        for c in cases:
            frames = {}
            lo, hi = c["win"]
            alpha = c.get("alpha", 0.05)
            for pack in (1, 0):
                with make_renderer(vra, (96, 80)) as r:
                    r.setQuirks(c.get("quirks", 0)); r.setLayout(R.LAYOUT_BRICKED); r.setPack12(pack)
                    r.setKernelVariant(c.get("variant", 0))
                    r.setVolume(vol); r.setWindow(lo, hi); r.setAlpha(alpha); r.setMIP(bool(c.get("mip", 0)))
                    tf = None
                    if c.get("tf"):
                        r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [0.3, 0.5, 0.1, 0.759], [0.8, 0.2, 0.4, 0.45], [1, 1, 1, 1]])
                        tf = r.getTransferLut()
                    r.cameraOrient(0.0, -0.4, 0.9)
                    block = r.getCameraBlock()
                    r.render()
                    frames[pack] = r.readPixels()
                    assert (r.pack12Bytes() > 0) == bool(pack), (dims, base, c, pack)
                    _, spp = r.countSamples(per_pixel=True)
                    assert r.last_kernel_name in FAST_KERNELS
            off = 1000 if c.get("quirks", 0) & R.QUIRK_U16_OFFSET else 0
            p = oracle.OracleParams(96, 80, cam=block, alpha_scale=alpha, min_val=lo + off, max_val=hi + off, is_mip=c.get("mip", 0), tf_rgba=tf)
            want, _, want_spp = oracle.render(vol, p, want_spp=True)
            assert_same(frames[1], want, spp, want_spp, what=f"packed12 dims {dims} base {base} {c}")
            assert np.array_equal(frames[1].view(np.uint32), frames[0].view(np.uint32)), (dims, base, c)
    # a range of 4097 values does not pack
    vol = rand_volume(rng, (32, 32, 32), np.uint16)
    vol = (vol.astype(np.int64) + 300).astype(np.uint16)
    vol.flat[0] = 300; vol.flat[1] = 300 + 4096
    with make_renderer(vra, (64, 64)) as r:
        r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED); r.setVolume(vol); r.setWindow(300, 4396); r.setAlpha(0.3)
        r.render()
        got = r.readPixels()
        assert r.pack12Bytes() == 0
    want, _ = oracle.render(vol, oracle.OracleParams(64, 64, alpha_scale=0.3, min_val=300, max_val=4396))
    assert_same(got, want, what="u16 volume spanning 4097 values")


def test_grey_transfer_function_runs_on_the_grey_ramp_kernels(vra, oracle):
    """Round 5: a transfer function with r == g == b in every entry (the reference's own black -> white ramp; its colour
    widget is commented out, include/UI/TransferFunction.h:4,22) is folded into the (c, a) table of the grey-ramp
    instances: composite frames through the fast and the relay kernel, all three views, with empty-space skipping, on
    u8 / u16 (packed copy included), equal the oracle and the generic kernel bit for bit; r == g == b in every pixel."""
    rng = np.random.default_rng(77)
    R = vra.renderer
    iso = [0, 141, 149, 255]
    grey = [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]]
    for dtype, dims, win in ((np.uint8, (64, 48, 40), (8, 255)), (np.uint16, (48, 64, 56), (300, 3900)), (np.uint16, (64, 64, 64), (0, 4095))):
        vol = rand_volume(rng, dims, dtype, smooth=True)
        for view in ("default", "top", "bottom"):
            for variant, skip in ((0, False), (0, True), (3, False), (2, False)):
                with make_renderer(vra, (120, 96)) as r:
                    r.setQuirks(0); r.setKernelVariant(variant); r.setSkipEmpty(skip)
                    r.setVolume(vol); r.setWindow(*win); r.setAlpha(0.3)
                    r.setInitialCameraRotation(view == "top", view == "bottom")
                    r.setTransferFunction(iso, grey)
                    lut = r.getTransferLut()
                    assert np.array_equal(lut[:, 0], lut[:, 1]) and np.array_equal(lut[:, 0], lut[:, 2])
                    r.cameraOrient(0.0, 0.35, -0.8)
                    block = r.getCameraBlock()
                    r.render()
                    got = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    assert r.last_kernel_name in FAST_KERNELS
                    r.setKernelVariant(1); r.render()
                    gen = r.readPixels()
                p = oracle.OracleParams(120, 96, cam=block, alpha_scale=0.3, min_val=win[0], max_val=win[1], tf_rgba=lut,
                                        view_top=int(view == "top"), view_bottom=int(view == "bottom"))
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                what = f"grey TF {np.dtype(dtype).name} {view} variant {variant} skip {skip}"
                assert_same(got, want, spp, want_spp, what=what)
                assert np.array_equal(got.view(np.uint32), gen.view(np.uint32)), what
                assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])
    # a coloured transfer function keeps the RGBA table path (and a grey one under MIP too): covered by
    # test_transfer_function / test_mip_and_view_swizzles; here only that a window wider than the (c, a) table still renders
    vol = rand_volume(rng, (40, 40, 40), np.uint16)
    vol[3, 4, 5] = 30000
    with make_renderer(vra, (64, 64)) as r:
        r.setQuirks(0); r.setVolume(vol); r.setWindow(0, 20000); r.setAlpha(0.2); r.setTransferFunction(iso, grey)
        lut = r.getTransferLut()
        r.render()
        got = r.readPixels()
    want, _ = oracle.render(vol, oracle.OracleParams(64, 64, alpha_scale=0.2, min_val=0, max_val=20000, tf_rgba=lut))
    assert_same(got, want, what="grey TF, window wider than the (c, a) table")


def test_long_axis_volume_without_address_tables(vra, oracle):
    """nx + ny + nz > 3072: the LDS address tables (and with them the packed copy and the batched trilinear
    kernel) do not apply; the specialised kernels compute brick addresses arithmetically"""
    rng = np.random.default_rng(3072)
    R = vra.renderer
    dims = (3000, 40, 36)
    vol = rand_volume(rng, dims, np.uint16, smooth=True)
    spacing = (0.02, 1.0, 1.0)                                   # a box of sane proportions
    for layout in (R.LAYOUT_BRICKED, R.LAYOUT_LINEAR):
        for filt, mip in ((R.FILTER_NEAREST, False), (R.FILTER_NEAREST, True), (R.FILTER_TRILINEAR, False)):
            with make_renderer(vra, (96, 64)) as r:
                r.setQuirks(0); r.setLayout(layout); r.setVolume(vol, spacing)
                r.setWindow(100, 3900); r.setAlpha(0.1); r.setFilter(filt); r.setMIP(mip)
                r.cameraOrient(0.0, -0.3, 0.5)
                block = r.getCameraBlock()
                r.render()
                got = r.readPixels()
                _, spp = r.countSamples(per_pixel=True)
                kernel = r.last_kernel_name
                assert r.pack12Bytes() == 0
            p = oracle.OracleParams(96, 64, cam=block, alpha_scale=0.1, voxel_size=spacing, min_val=100, max_val=3900,
                                    filter=int(filt == R.FILTER_TRILINEAR), is_mip=int(mip))
            want, _, want_spp = oracle.render(vol, p, want_spp=True)
            assert_same(got, want, spp, want_spp, what=f"long-axis volume layout {layout} filter {filt} mip {mip} kernel {kernel}")
            if filt == R.FILTER_TRILINEAR:       # bricked: the LDS-staged kernel's plan and tables cover a tile's own index ranges only (round 3)
                assert kernel == ("raymarch_tslab_kernel" if layout == R.LAYOUT_BRICKED else "raymarch_generic_kernel")
            else:
                assert kernel in FAST_KERNELS


def test_trilinear_on_tiny_volumes_reads_the_last_voxel(vra, oracle):
    """regression (found by tools/stress_campaign.py): the x-neighbour pair load of the LAST voxel
    of the buffer reaches past it, and a partly out-of-range buffer load returns 0 for all of it"""
    rng = np.random.default_rng(66)
    R = vra.renderer
    for dims in ((2, 3, 5), (1, 1, 1), (2, 2, 2), (5, 1, 3), (4, 4, 4)):
        for dtype in (np.uint16, np.uint8):
            vol = rand_volume(rng, dims, dtype)
            vol.flat[-1] = vol.flat[0] = 255 if dtype == np.uint8 else 4000      # bright first and last voxel
            spacing = (1.686, 0.836, 1.578)
            for layout in (R.LAYOUT_LINEAR, R.LAYOUT_BRICKED):
                for variant in (0, 1):
                    with make_renderer(vra, (68, 68)) as r:
                        r.setQuirks(0); r.setLayout(layout); r.setKernelVariant(variant)
                        r.setVolume(vol, spacing); r.setFilter(R.FILTER_TRILINEAR)
                        vmax = 255 if dtype == np.uint8 else 4095
                        r.setWindow(vmax // 6, vmax); r.setAlpha(1.0)
                        for _ in range(3):
                            block = _random_camera_block(rng)
                            r.setCameraBlock(block)
                            r.render()
                            got = r.readPixels()
                            p = oracle.OracleParams(68, 68, cam=block, alpha_scale=1.0, voxel_size=spacing, min_val=vmax // 6,
                                                    max_val=vmax, filter=1)
                            want, _ = oracle.render(vol, p)
                            assert_same(got, want, what=f"tiny trilinear dims {dims} {np.dtype(dtype).name} layout {layout} variant {variant} kernel {r.last_kernel_name}")


def test_trilinear_kernel_full_size_equals_generic(vra, cfg3):
    """cfg3 with TRILINEAR: the batched trilinear kernel against the line-by-line generic one,
    composite and MIP, bit for bit (the oracle pins both at small sizes and in the goldens)"""
    r = cfg3
    R = vra.renderer
    r.setFilter(R.FILTER_TRILINEAR)
    try:
        for mip in (False, True):
            r.setMIP(mip)
            r.setKernelVariant(1); r.render()
            assert r.last_kernel_name == "raymarch_generic_kernel"
            b = r.readPixels().copy()
            nb = r.countSamples()
            # automatic choice at the (axis-aligned) default pose: the LDS-staged kernel; variant 2: the batched kernel
            for variant, name in ((0, "raymarch_tslab_kernel"), (2, "raymarch_tri_kernel")):
                r.setKernelVariant(variant); r.render()
                assert r.last_kernel_name == name
                a = r.readPixels().copy()
                na = r.countSamples()
                assert na == nb and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mip, name)
    finally:
        r.setMIP(False); r.setKernelVariant(0); r.setFilter(R.FILTER_NEAREST)


def test_packed12_full_size_equals_unpacked(cfg3):
    """cfg3 (1024^3 u16 @1080p): packed and unpacked gathers give the same frame"""
    r = cfg3
    r.setPack12(1); r.render(); a = r.readPixels().copy()
    r.setPack12(0); r.render(); b = r.readPixels().copy()
    r.setPack12(1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def run_random_trials(vra, oracle, seed, n_trials, extended=False, log=None):
    """randomised parity trials (3 cameras each); `extended` also draws MIP / transfer function /
    TRILINEAR / closed-form accumulation / forced relay kernel / row stripes.  Used by the test
    below and, with many more trials, by tools/stress_campaign.py.  Returns frames checked."""
    rng = np.random.default_rng(seed)
    dims_pool = [(1, 1, 1), (2, 3, 5), (8, 8, 8), (16, 32, 64), (31, 17, 9), (64, 64, 64), (50, 1, 50), (128, 4, 4)]
    if extended:
        dims_pool += [(32, 32, 32), (7, 64, 33), (96, 80, 72), (4, 4, 128), (256, 16, 16), (3100, 3, 4)]   # the last: no address tables
    R = vra.renderer
    n_checked = 0
    for trial in range(n_trials):
        dims = dims_pool[trial % len(dims_pool)]
        dtype = np.uint8 if trial % 3 else np.uint16
        vol = rand_volume(rng, dims, dtype, smooth=bool(trial % 2))
        spacing = (1.0, 1.0, 1.0) if trial % 4 == 0 else tuple(np.round(rng.uniform(0.3, 2.5, size=3), 3).tolist())
        vmax = 255 if dtype == np.uint8 else 4095
        lo = int(rng.integers(0, vmax // 3)); hi = int(rng.integers(vmax // 2, vmax + 1))
        alpha = float(np.float32(rng.choice([1.0, 0.3, 0.02, 0.0])))
        W, H = int(rng.integers(17, 90)), int(rng.integers(17, 70))
        mip = tf = tri = accum = relay = stripes = pipe = tslab = batched = unstaged = False
        renders = 1
        if extended:
            mode = int(rng.integers(0, 24))
            mip, tf, tri, accum = mode in (1, 8, 13, 17, 19), mode in (2, 8, 14, 18, 19), mode in (3, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23), mode == 4     # 8: MIP through the transfer function
            relay = mode in (5, 6)
            stripes = mode in (7, 15, 21)
            pipe = mode in (9, 10)                # the fast kernel's software-pipelined loop (variant 5) instead of relay / plain loop
            tslab = mode in (16, 17, 21)          # round 3: TRILINEAR on the LDS-staged layer-synchronous kernel (variant 6); 18, 19: with a transfer function
            batched = mode == 20                  # ... and the batched trilinear kernel (variant 2)
            unstaged = mode in (12, 13, 14, 15)   # ... with staging switched off (variant 7): the path of tiles whose brick layers do not fit LDS
            renders = 5 if mode in (22, 23) else 1    # the measured work model: five frames per camera try its candidate kernels
            if rng.random() < 0.2:
                W, H = int(rng.integers(90, 200)), int(rng.integers(70, 160))
        tf_lut = None
        quirks, pack = 0, 1
        if extended:
            quirks = int(rng.integers(0, 4)) if rng.random() < 0.25 else 0     # Q1 truncated grid, Q10 +1000 (u16)
            pack = 0 if rng.random() < 0.25 else 1
        win_off = 1000 if (quirks & 2) and dtype == np.uint16 else 0
        with make_renderer(vra, (W, H)) as r:
            r.setQuirks(quirks)
            r.setPack12(pack)
            r.setLayout(trial % 2)
            r.setVolume(vol, spacing)
            r.setWindow(lo, hi)
            r.setAlpha(alpha)
            r.setSkipEmpty(bool(trial % 5 == 0))
            top, bottom = (trial % 7 == 3), (trial % 7 == 5)
            if top or bottom:
                r.setInitialCameraRotation(top, bottom)
            if mip:
                r.setMIP(True)
            if tf:
                r.setTransferFunction([0, 90, 160, 255], [[0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])
                tf_lut = r.getTransferLut()
            if tri:
                r.setFilter(R.FILTER_TRILINEAR)
            if accum:
                r.setAccum(1)
            if relay:
                r.setKernelVariant(3)
            if pipe:
                r.setKernelVariant(5)
            if tslab:
                r.setKernelVariant((6, 8, 9, 10, 11)[(trial // 3) % 5])      # round 4: whole layers, or a half-layer shape (16-bit volumes; 8-bit ones run as 6); round 6: 16x16-pixel tiles
            if unstaged:
                r.setKernelVariant(7)
            if batched:
                r.setKernelVariant(2)
            rows = None
            if stripes:
                rows = (4, int(rng.integers(0, 3)), 3)
                r.setRowStripes(*rows)
            for _ in range(3):
                block = _random_camera_block(rng)
                if rng.random() < 0.25:                       # axis-parallel rays (zero direction components)
                    block = oracle.default_camera_block(); block[16:19] = block[12:15] = (rng.uniform(-0.4, 0.4), 0.0, 3.0)
                r.setCameraBlock(block)
                for _k in range(renders - 1):                  # earlier frames of the exploration: kept for the comparison below
                    r.render()
                    first = r.readPixels().copy() if _k == 0 else first
                r.render()
                got = r.readPixels()
                if renders > 1:
                    assert np.array_equal(first.view(np.uint32), got.view(np.uint32)), "frames of the work model's exploration differ"
                _, spp = r.countSamples(per_pixel=True)
                p = oracle.OracleParams(W, H, cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo + win_off, max_val=hi + win_off,
                                        view_top=int(top), view_bottom=int(bottom), is_mip=int(mip), filter=int(tri),
                                        accum=int(accum), tf_rgba=tf_lut, trunc_grid=quirks & 1)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                what = (f"seed {seed} trial {trial} dims {dims} {np.dtype(dtype).name} spacing {spacing} window [{lo},{hi}] alpha {alpha} "
                        f"{W}x{H} mip {mip} tf {tf} tri {tri} accum {accum} relay {relay} tslab {tslab} unstaged {unstaged} batched {batched} renders {renders} stripes {rows} quirks {quirks} pack12 {pack} kernel {r.last_kernel_name}")
                if rows:                                       # only this shard's rows are rendered
                    mine = np.array([y for y in range(H) if (y // rows[0]) % rows[2] == rows[1]])
                    assert_same(got[mine], want[mine], spp[mine], want_spp[mine], what=what)
                else:
                    assert_same(got, want, spp, want_spp, what=what)
                n_checked += 1
        if log and (trial + 1) % 200 == 0:
            log(f"{trial + 1} trials, {n_checked} frames bit-exact")
    return n_checked


def test_randomised_cameras_volumes_and_modes(vra, oracle):
    """stress of the safe-prefix / certified-division / batching logic: random dims (odd,
    tiny, power-of-two), spacings, cameras (inside, grazing, axis-parallel), windows, alpha"""
    assert run_random_trials(vra, oracle, 20260928, 120) == 360


def test_randomised_extended_modes(vra, oracle):
    """the same with MIP / transfer function / TRILINEAR / closed-form accumulation / forced
    relay kernel / row stripes drawn at random (tools/stress_campaign.py runs thousands)"""
    assert run_random_trials(vra, oracle, 7, 96, extended=True) == 288


def test_out_of_contract_parameters_fall_back_to_the_generic_kernel(vra, oracle):
    """reversed window (min > max), alpha_scale outside [0,1], negative window bounds:
    legal shader inputs with odd semantics -- the generic kernel repeats them literally"""
    rng = np.random.default_rng(41)
    vol = rand_volume(rng, (30, 26, 34), np.uint8)
    cases = [
        dict(min_val=200, max_val=50, alpha_scale=0.01),      # clamp(x,lo,hi) with lo > hi -> hi, not normalised
        dict(min_val=0, max_val=255, alpha_scale=1.7),        # dest.a can overshoot and oscillate
        dict(min_val=0, max_val=255, alpha_scale=-0.3),
        dict(min_val=-40, max_val=300, alpha_scale=0.2),      # window wider than the data type
    ]
    for kw in cases:
        with make_renderer(vra, (72, 60)) as r:
            r.setVolume(vol)
            r.setWindow(kw["min_val"], kw["max_val"])
            r.setAlpha(kw["alpha_scale"])
            r.render()
            got = r.readPixels()
            _, spp = r.countSamples(per_pixel=True)
            kernel = r.last_kernel_name
        want, _, want_spp = oracle.render(vol, oracle.OracleParams(72, 60, **kw), want_spp=True)
        assert_same(got, want, spp, want_spp, what=f"{kw} via {kernel}")
        if kw["min_val"] > kw["max_val"] or not (0.0 <= kw["alpha_scale"] <= 1.0):
            assert kernel == "raymarch_generic_kernel"


def test_mip_and_tf_deep_regime_timings_are_reported(vra, cfg3):
    """not a performance gate: prints the fast-path timings of the non-headline modes"""
    r = cfg3
    out = {}
    for name, setup in (("composite", lambda: None), ("mip", lambda: r.setMIP(True)),
                        ("tf", lambda: (r.setWindow(0, 2047), r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [.5, .5, .5, .759], [.6, .6, .6, .45], [1, 1, 1, 1]])))):
        setup()
        r.render(); r.kernelMsTake()
        for _ in range(5):
            r.render()
        out[name] = (round(r.kernelMsTake() / 5, 3), r.last_kernel_name)
        r.setMIP(False); r.setTransferFunction(); r.setWindow(0, 4095)
    print("cfg3 deep regime:", out)
    assert all(k in FAST_KERNELS for _, k in out.values())


@pytest.mark.parametrize("variant", [2, 3], ids=["dense-variants", "relay"])
def test_relay_kernel_equals_fast_kernel(vra, oracle, variant):
    """the 4-wavefront relay (sparse launches) against the oracle, forced on and off, across
    dtypes / layouts / window (LUT, no-LUT, clamp) / early termination / cameras"""
    rng = np.random.default_rng(77)
    for dims, dtype, win, alpha in [((64, 64, 64), np.uint8, (0, 255), 0.02), ((64, 64, 64), np.uint8, (30, 190), 1.0),
                                    ((40, 56, 24), np.uint16, (0, 4095), 0.05), ((128, 64, 32), np.uint16, (100, 60000), 0.3),
                                    ((33, 31, 65), np.uint8, (0, 255), 0.0)]:
        vol = rand_volume(rng, dims, dtype, smooth=True)
        for layout in (0, 1):
            with make_renderer(vra, (120, 88)) as r:
                r.setQuirks(0)
                r.setLayout(layout)
                r.setKernelVariant(variant)
                r.setVolume(vol, (1.0, 1.0, 1.0) if dims[0] == dims[1] == dims[2] else (1.0, 0.7, 1.9))
                r.setWindow(*win)
                r.setAlpha(alpha)
                for name, block in orbit_blocks(oracle):
                    r.setCameraBlock(block)
                    r.render()
                    assert r.last_kernel_name == "raymarch_relay_kernel" if variant == 3 else r.last_kernel_name == "raymarch_fast_kernel"
                    got = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    p = oracle.OracleParams(120, 88, cam=block, alpha_scale=alpha, min_val=win[0], max_val=win[1],
                                            voxel_size=(1.0, 1.0, 1.0) if dims[0] == dims[1] == dims[2] else (1.0, 0.7, 1.9))
                    want, _, want_spp = oracle.render(vol, p, want_spp=True)
                    assert_same(got, want, spp, want_spp, what=f"variant {variant} {dims} {dtype.__name__} layout {layout} {name}")


def test_short_batches_for_high_opacity(vra, cfg3):
    """alpha_scale >= 0.5 (rays end early): the host picks the fast kernel's 4-sample batches; the frame and the per-ray
    sample counts equal the 8-sample loop's and the generic kernel's"""
    r = cfg3
    r.setAlpha(1.0)
    frames, totals = [], []
    for variant in (0, 2, 1):
        r.setKernelVariant(variant)
        r.render()
        assert r.last_kernel_name == ("raymarch_generic_kernel" if variant == 1 else "raymarch_fast_kernel")
        frames.append(r.readPixels())
        totals.append(r.countSamples())
    assert totals[0] == totals[1] == totals[2] == 139601160          # the shallow regime of BASELINE.md
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    assert np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))
    r.setKernelVariant(0)
    r.setAlpha(0.004)


def test_pipelined_loop_full_size_shard(vra, cfg3):
    """one rank's stripes of the cfg3 frame at N = 4 (335 tiles: under-filled, not sparse): the host picks the fast kernel's
    pipelined loop; the frame equals the plain loop's bit for bit"""
    r = cfg3
    r.setRowStripes(16, 1, 4)
    r.setKernelVariant(2)
    r.render()
    assert r.last_kernel_name == "raymarch_fast_kernel"
    plain = r.readPixels()
    spp_plain = r.countSamples()
    for variant in (0, 5):
        r.setKernelVariant(variant)
        r.render()
        assert r.last_kernel_name == "raymarch_fast_kernel"
        assert np.array_equal(plain.view(np.uint32), r.readPixels().view(np.uint32))
        assert r.countSamples() == spp_plain
    # an oblique view keeps the relay kernel up to 512 tiles (its tiles are L1-bound chains: the pipelined loop would lose)
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    r.setKernelVariant(2); r.render()
    assert r.last_kernel_name == "raymarch_fast_kernel"
    oblique_plain = r.readPixels()
    r.setKernelVariant(0); r.render()
    assert r.last_kernel_name == "raymarch_relay_kernel"
    assert np.array_equal(oblique_plain.view(np.uint32), r.readPixels().view(np.uint32))
    r.resetCamera()
    r.setKernelVariant(0)
    r.setRowStripes(1, 0, 1)


def test_relay_kernel_full_size_shard(vra, cfg3):
    """one rank's stripes of the cfg3 frame at N = 8: relay kernel == fast kernel, bit for bit"""
    r = cfg3
    r.setRowStripes(16, 5, 8)
    r.setKernelVariant(2)
    r.render(); r.kernelMsTake()
    for _ in range(5):
        r.render()
    t_fast = r.kernelMsTake() / 5
    assert r.last_kernel_name == "raymarch_fast_kernel"
    fast = r.readPixels()
    r.setKernelVariant(0)
    for _ in range(12):                                       # the measured work model tries its candidates, then settles
        r.render()
    r.kernelMsTake()
    for _ in range(5):
        r.render()
    t_relay = r.kernelMsTake() / 5
    assert r.last_kernel_name == "raymarch_relay_kernel"      # chosen automatically for this sparse shard
    relay = r.readPixels()
    r.setRowStripes(1, 0, 1)
    assert np.array_equal(fast.view(np.uint32), relay.view(np.uint32))
    print(f"cfg3 shard 5/8: fast kernel {t_fast:.3f} ms, relay kernel {t_relay:.3f} ms")


@pytest.mark.parametrize("mode", ["mip", "tf", "mip_tf", "top", "bottom", "mip_top", "tf_bottom"])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
def test_relay_kernel_modes_and_views(vra, oracle, dtype, mode):
    """round 2: the relay kernel in MIP / transfer-function / view top-bottom (what a sparse multi-GPU shard
    of those modes runs): forced with variant 3, against the oracle, both layouts, cubic (voxel-unit
    marching) and anisotropic (certified division) volumes"""
    rng = np.random.default_rng(123)
    mip, tf = "mip" in mode, "tf" in mode
    top, bottom = "top" in mode, "bottom" in mode
    vmax = 255 if dtype == np.uint8 else 4095
    for dims, spacing in (((64, 64, 64), (1.0, 1.0, 1.0)), ((40, 56, 24), (1.0, 0.7, 1.9))):
        vol = rand_volume(rng, dims, dtype, smooth=True)
        for layout in (0, 1):
            with make_renderer(vra, (120, 88)) as r:
                r.setQuirks(0)
                r.setLayout(layout)
                r.setKernelVariant(3)
                r.setVolume(vol, spacing)
                r.setWindow(vmax // 20, vmax - vmax // 10)
                alpha = 0.4 if mip else 0.04
                r.setAlpha(alpha)
                r.setMIP(mip)
                tf_lut = None
                if tf:
                    r.setTransferFunction([0, 90, 160, 255], [[0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])
                    tf_lut = r.getTransferLut()
                r.setInitialCameraRotation(top, bottom)
                for name, block in orbit_blocks(oracle)[:3]:
                    r.setCameraBlock(block)
                    r.render()
                    assert r.last_kernel_name == "raymarch_relay_kernel", (mode, dims, layout, name)
                    got = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    p = oracle.OracleParams(120, 88, cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=vmax // 20, max_val=vmax - vmax // 10,
                                            is_mip=int(mip), view_top=int(top), view_bottom=int(bottom), tf_rgba=tf_lut)
                    want, _, want_spp = oracle.render(vol, p, want_spp=True)
                    assert_same(got, want, spp, want_spp, what=f"relay {mode} {dims} {np.dtype(dtype).name} layout {layout} {name}")


def test_grey_transfer_function_folded_into_the_grey_ramp_table(vra, oracle):
    """round-5 advisor: a GREY transfer function under NEAREST composite is folded into the grey-ramp instances' (c, a) table
    (FrameParams::tf_grey) -- on the fast kernel, on the relay kernel, and through the 12-bit packed copy of CT-style data whose
    minimum is 1000 (pk12_base): every one of them must equal the transfer-function kernel proper (generic, variant 1) and the
    oracle bit for bit; with TRILINEAR the flag must stay off (the staged kernel classifies through the RGBA table)."""
    R = vra.renderer
    knots = ([0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]])
    for synth, dims, b, win in ((R.SYNTH_NOISE_BALL_CT, (160, 144, 128), 2, (1000, 5095)), (R.SYNTH_NOISE_BALL, (128, 128, 128), 2, (64, 4000)),
                                (R.SYNTH_NOISE_BALL, (96, 112, 80), 1, (8, 255))):
        with make_renderer(vra, (200, 144)) as r:
            r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED)
            r.generateSynthetic(synth, dims, b, 0x9E3779B9)
            vol = r.readVolume()
            r.setWindow(*win); r.setAlpha(0.03)
            r.setTransferFunction(*knots)
            tf_lut = r.getTransferLut()
            assert np.array_equal(tf_lut[:, 0], tf_lut[:, 1]) and np.array_equal(tf_lut[:, 0], tf_lut[:, 2])     # grey: the fold applies
            for name, block in orbit_blocks(oracle)[:3]:
                r.setCameraBlock(block)
                p = oracle.OracleParams(200, 144, cam=block, alpha_scale=0.03, min_val=win[0], max_val=win[1], tf_rgba=tf_lut)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                seen = {}
                for variant in (0, 2, 3, 1):                             # measured choice, fast kernel, relay kernel, generic
                    r.setFilter(R.FILTER_NEAREST); r.setKernelVariant(variant); r.render()
                    seen[variant] = r.last_kernel_name
                    got = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    assert_same(got, want, spp, want_spp, what=f"grey TF {dims} x{b} window {win} {name} variant {variant} [{r.last_kernel_name}]")
                    if b == 2 and variant == 2:
                        assert r.pack12Bytes() > 0                        # the packed copy (voxel - minimum) was the source of the fast kernel's gathers
                assert seen[2] == "raymarch_fast_kernel" and seen[3] == "raymarch_relay_kernel" and seen[1] == "raymarch_generic_kernel", seen
                # TRILINEAR through the same grey table: the staged kernel, against the oracle's filtered frame
                r.setFilter(R.FILTER_TRILINEAR); r.setKernelVariant(0); r.render()
                p.filter = 1
                want_t, _, want_spp_t = oracle.render(vol, p, want_spp=True)
                got = r.readPixels()
                _, spp = r.countSamples(per_pixel=True)
                assert_same(got, want_t, spp, want_spp_t, what=f"grey TF TRILINEAR {dims} x{b} {name} [{r.last_kernel_name}]")


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
def test_trilinear_apron_copy_is_invisible(vra, oracle, dtype):
    """TRILINEAR on the bricked layout gathers from the apron copy (every 4^3 brick stored as 5x4x4): frames equal the
    ones without it and the oracle's, bit for bit -- dims that are not multiples of 4, rays through every face of the
    box (clamped taps at the faces), composite and MIP"""
    R = vra.renderer
    rng = np.random.default_rng(31)
    for dims, spacing, size in (((37, 22, 41), (1.0, 1.3, 0.8), (150, 110)), ((64, 64, 64), (1.0, 1.0, 1.0), (128, 96)), ((5, 3, 2), (1.0, 1.0, 1.0), (64, 48))):
        vol = rand_volume(rng, dims, dtype, smooth=True)
        vmax = 255 if dtype == np.uint8 else 4095
        lo, hi = vmax // 10, vmax - vmax // 8
        with make_renderer(vra, size) as r:
            r.setQuirks(0)
            r.setVolume(vol, spacing)
            r.setWindow(lo, hi)
            r.setFilter(R.FILTER_TRILINEAR)
            r.setKernelVariant(2)                                # the batched trilinear kernel (variant 0 may pick the LDS-staged one)
            for mip, alpha in ((False, 0.05), (True, 0.5), (False, 1.0)):
                r.setAlpha(alpha); r.setMIP(mip)
                for name, block in orbit_blocks(oracle):
                    r.setCameraBlock(block)
                    r.setTrilinearCopy(True)
                    r.render()
                    assert r.last_kernel_name == "raymarch_tri_kernel"
                    assert r.trilinearCopyBytes() > 0
                    with_copy = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    r.setTrilinearCopy(False)
                    r.render()
                    assert r.last_kernel_name == "raymarch_tri_kernel" and r.trilinearCopyBytes() == 0
                    assert np.array_equal(with_copy.view(np.uint32), r.readPixels().view(np.uint32)), (dims, name, mip)
                    p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo, max_val=hi,
                                            is_mip=int(mip), filter=1)
                    want, _, want_spp = oracle.render(vol, p, want_spp=True)
                    assert_same(with_copy, want, spp, want_spp, what=f"apron {dims} {name} mip={mip}")
            r.setLayout(R.LAYOUT_LINEAR); r.setTrilinearCopy(True); r.render()      # the linear layout has no apron copy
            assert r.trilinearCopyBytes() == 0


def far_camera_blocks(oracle):
    """eye 13..33 box sizes away (zoomed out, then orbited; the view-plane distance grows with the radius so that the
    box still fills the image): o + d*t_min is then rounded to ~1e-6 of absolute error, more than the EPSILON*|d| nudge
    of VolumeRenderer.cs:107 -- a few per cent of the HIT rays have their FIRST sample outside the box and end with
    zero samples"""
    out = []
    for nz, ze, az in ((10, -0.9, 1.86), (20, 0.42, 0.54), (20, 0.3, -0.3), (30, 0.2, 0.9)):
        c = oracle.Camera()
        for _ in range(nz):
            c.orient(-1.0, 0.0, 0.0)
        c.orient(0, ze, az)
        b = c.block().copy()
        b[20] *= float(np.linalg.norm(b[16:19])) / 3.0
        out.append((f"far{nz}", b))
    return out


def _first_sample_outside(block, W, H):
    """float32 restatement of computeRay / intersectRayAABB / the first texcoord for a unit box (test bookkeeping only:
    it says how many hit rays start outside, i.e. whether the test below exercises what it claims to)"""
    f32 = np.float32
    c = block.astype(f32)
    px = (np.arange(W, dtype=f32) + f32(0.5))[None, :].repeat(H, 0)
    py = (np.arange(H, dtype=f32) + f32(0.5))[:, None].repeat(W, 1)
    fw, fh = f32(W), f32(H)
    x = (fw / fh) * ((f32(2) * px) / fw - f32(1)); y = (f32(2) * py) / fh - f32(1); z = np.full_like(x, -c[20])
    ln = np.sqrt((x * x + y * y) + z * z)
    dx, dy, dz = x / ln, y / ln, z / ln
    mx = (c[0] * dx + c[4] * dy) + c[8] * dz; my = (c[1] * dx + c[5] * dy) + c[9] * dz; mz = (c[2] * dx + c[6] * dy) + c[10] * dz
    ln = np.sqrt((mx * mx + my * my) + mz * mz)
    d, o = [mx / ln, my / ln, mz / ln], [c[16], c[17], c[18]]
    tmin = np.full_like(x, -np.inf); tmax = np.full_like(x, np.inf)
    with np.errstate(all="ignore"):
        hit = np.ones_like(x, bool)
        for a in range(3):
            inv = f32(1) / d[a]
            t0 = (f32(-0.5) - o[a]) * inv; t1 = (f32(0.5) - o[a]) * inv
            tmin = np.maximum(tmin, np.minimum(t0, t1)); tmax = np.minimum(tmax, np.maximum(t0, t1))
            if a == 1:
                hit &= ~(tmax < tmin)
        hit &= tmax > np.maximum(tmin, f32(0))
        outside = np.zeros_like(hit)
        for a in range(3):
            u = ((o[a] + d[a] * tmin) + d[a] * f32(1e-6)) + f32(0.5)
            outside |= (u > 1) | (u < 0)
    return int((hit & outside).sum())


@pytest.mark.parametrize("variant", [0, 2, 3, 5], ids=["auto", "fast", "relay", "pipelined"])
def test_rays_that_end_inside_their_checked_head(vra, oracle, variant):
    """round-2 advisor finding: the relay kernel composited its prefix batches and tail onto rays whose first sample
    fails the bounds test (the shader, and every other kernel, end such a ray with zero samples).  Far cameras make
    those rays common; every kernel variant must give the oracle's frame and per-pixel fetch counts."""
    rng = np.random.default_rng(404)
    R = vra.renderer
    W, H = 120, 88
    blocks = far_camera_blocks(oracle)
    assert sum(_first_sample_outside(b, W, H) for _, b in blocks) > 500
    for dtype, dims, win in ((np.uint8, (64, 64, 64), (0, 255)), (np.uint16, (64, 64, 64), (0, 4095))):
        vol = rand_volume(rng, dims, dtype, smooth=True)
        vol[:] = np.maximum(vol, 40 if dtype == np.uint8 else 600)      # every sample visible: a ray with samples shows
        for filt in (R.FILTER_NEAREST, R.FILTER_TRILINEAR):
            with make_renderer(vra, (W, H)) as r:
                r.setQuirks(0); r.setKernelVariant(variant); r.setVolume(vol); r.setWindow(*win); r.setAlpha(0.05); r.setFilter(filt)
                for name, block in blocks:
                    r.setCameraBlock(block)
                    r.render()
                    got = r.readPixels()
                    _, spp = r.countSamples(per_pixel=True)
                    kernel = r.last_kernel_name
                    p = oracle.OracleParams(W, H, cam=block, alpha_scale=0.05, min_val=win[0], max_val=win[1],
                                            filter=int(filt == R.FILTER_TRILINEAR))
                    want, _, want_spp = oracle.render(vol, p, want_spp=True)
                    assert_same(got, want, spp, want_spp, what=f"variant {variant} {np.dtype(dtype).name} filter {filt} {name} kernel {kernel}")
                    if variant == 3 and filt == R.FILTER_NEAREST:
                        assert kernel == "raymarch_relay_kernel"


@pytest.mark.parametrize("variant", [0, 1, 2], ids=["auto", "generic", "fast"])
def test_step_budget_of_10000_samples(vra, oracle, variant):
    """Q6: the shader's loop runs at most 10000 iterations (VolumeRenderer.cs:115).  A 16384 x 2 x 2 volume in a unit
    box has a step of ~1e-4, so every ray that crosses the box takes the cap; the checked head + the prefix batches
    must not take one sample more (round-2 advisor finding: head + nb * BATCH could exceed max_steps)."""
    rng = np.random.default_rng(10000)
    R = vra.renderer
    dims = (16384, 2, 2)
    vol = rng.integers(0, 256, size=(dims[2], dims[1], dims[0]), dtype=np.uint8)
    spacing = (1.0, 4096.0, 4096.0)                         # box 1 x 0.5 x 0.5, step 7.5e-5: 13 000 steps along x
    W, H = 96, 64
    for alpha in (0.00005, 1.0):
        capped = 0
        with make_renderer(vra, (W, H)) as r:
            r.setQuirks(0); r.setKernelVariant(variant); r.setVolume(vol, spacing); r.setWindow(0, 255); r.setAlpha(alpha)
            for ze, az in ((0.0, 2.244), (0.3, 2.1), (-0.9, 0.7)):
                r.resetCamera(); r.cameraOrient(0.0, ze, az)
                block = r.getCameraBlock()
                r.render()
                got = r.readPixels()
                _, spp = r.countSamples(per_pixel=True)
                p = oracle.OracleParams(W, H, cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=0, max_val=255)
                want, _, want_spp = oracle.render(vol, p, want_spp=True)
                capped += int((want_spp == 10000).sum())
                assert int(want_spp.max()) <= 10000
                assert_same(got, want, spp, want_spp, what=f"10000-step cap variant {variant} alpha {alpha} pose ({ze}, {az}) kernel {r.last_kernel_name}")
        assert alpha == 1.0 or capped >= 200, capped


@pytest.mark.parametrize("filt", ["nearest", "trilinear"])
def test_measured_work_model_keeps_the_bits_and_settles(vra, oracle, filt):
    """kernel variant 0 with the measured work model (vr_set_autotune, default on): the first frames of a configuration try the
    candidate kernels in turn -- every frame must still be the oracle's frame bit for bit, more than one kernel must have
    run, and after the exploration the choice must stay put; with the model off the heuristic's kernel runs every time"""
    rng = np.random.default_rng(21)
    R = vra.renderer
    vol = rand_volume(rng, (96, 80, 72), np.uint16, smooth=True)
    W, H = 200, 144
    with make_renderer(vra, (W, H)) as r:
        r.setQuirks(0); r.setVolume(vol); r.setWindow(0, 4095); r.setAlpha(0.03)
        r.setFilter(R.FILTER_TRILINEAR if filt == "trilinear" else R.FILTER_NEAREST)
        block = r.getCameraBlock()
        p = oracle.OracleParams(W, H, cam=block, alpha_scale=0.03, min_val=0, max_val=4095, filter=int(filt == "trilinear"))
        want, _ = oracle.render(vol, p)
        seen, tail = [], []
        for k in range(24):
            if k % 2:
                r.render()
            else:
                r.renderAsync(); r.synchronize()
            seen.append(r.last_kernel_name)
            assert_same(r.readPixels(), want, what=f"{filt} frame {k} via {r.last_kernel_name}")
        assert len(set(seen[:14])) >= 2, seen                 # the exploration really ran different kernels
        assert len(set(seen[-6:])) == 1, seen                 # ... and has settled
        r.setAutotune(False)
        for k in range(6):
            r.render()
            tail.append(r.last_kernel_name)
        assert len(set(tail)) == 1, tail
