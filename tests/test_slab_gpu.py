"""The LDS-staged ray-march kernel (vr_slab.hip, opt-in: vr_set_kernel_variant(4)) against the CPU
oracle and against the fast kernel.

Only WHERE the prefix gathers read from differs (a torus of brick slots in LDS, filled by LDS-DMA,
instead of scattered global loads); every frame must stay bit-identical and every per-pixel fetch
count equal.  The cases below force the interesting shapes of its load plan: axis-parallel and
oblique entry faces, tiles straddling box edges (several entry faces in one workgroup), rays
marching against an axis, re-dimensioning of the torus as the footprint grows, the global-gather
fallback, volumes smaller than one brick layer, both staged formats (u8 bricks, 12-bit packed u16).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SLAB = "raymarch_slab_kernel"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rand_volume(rng, dims, dtype, smooth=False):
    nx, ny, nz = dims
    hi = 256 if dtype == np.uint8 else 4096
    if smooth:
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        v = ((np.sin(x * 0.3) + np.cos(y * 0.23) + np.sin(z * 0.31) + 3) / 6 * (hi - 1)).astype(np.int64)
        v += rng.integers(0, 3, size=v.shape)
        return np.clip(v, 0, hi - 1).astype(dtype)
    return rng.integers(0, hi, size=(nz, ny, nx), dtype=np.int64).astype(dtype)


def cameras(oracle, rng, n_random=3):
    out = [("default", oracle.default_camera_block())]
    for name, (dz, da) in {"orbit_a": (0.06 * 7, 0.06 * 9), "orbit_b": (-0.06 * 15, 0.06 * 31), "neg_azimuth": (0.0, -0.06 * 5),
                           "offaxis": (-(np.pi / 6) / 0.7, (np.pi / 4) / 0.7), "behind": (0.0, np.pi / 0.7), "pole": (-100.0, 0.3)}.items():
        c = oracle.Camera()
        c.orient(0, dz, da)
        out.append((name, c.block()))
    for k in range(n_random):
        c = oracle.Camera()
        c.orient(0, float(rng.uniform(-2, 2)), float(rng.uniform(-6, 6)))
        if k % 2:
            c.orient(1, 0, 0)                                   # one unit closer: wider footprint per pixel
        out.append((f"random{k}", c.block()))
    return out


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16_pack12"])
@pytest.mark.parametrize("dims,spacing", [((64, 64, 64), (1, 1, 1)), ((96, 80, 72), (1.0, 0.8, 1.7)), ((128, 32, 48), (1, 1, 1)),
                                          ((40, 56, 24), (0.6, 1.0, 1.3)), ((7, 5, 3), (1, 1, 1)), ((256, 256, 256), (1, 1, 1))],
                         ids=["cube64", "noncubic", "flat", "aniso", "tiny", "cube256"])
def test_slab_kernel_matches_oracle(vra, oracle, dtype, dims, spacing):
    rng = np.random.default_rng(abs(hash((dims, np.dtype(dtype).name))) % (2 ** 32))
    vol = rand_volume(rng, dims, dtype, smooth=dims[0] >= 96)
    vmax = 255 if dtype == np.uint8 else 4095
    size = (200, 144) if dims[0] < 256 else (320, 200)
    seen = set()
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.setVolume(vol, spacing)
        for (lo, hi, alpha) in ((0, vmax, 0.02), (vmax // 5, vmax - vmax // 4, 1.0)):
            r.setWindow(lo, hi)
            r.setAlpha(alpha)
            for name, block in cameras(oracle, rng):
                r.setCameraBlock(block)
                r.setKernelVariant(4)                            # the LDS-staged kernel wherever it is eligible
                r.render()
                seen.add(r.last_kernel_name)
                got = r.readPixels()
                total, spp = r.countSamples(per_pixel=True)
                p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo, max_val=hi, threads=8)
                want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
                what = f"{dims} {np.dtype(dtype).name} window [{lo},{hi}] alpha {alpha} camera {name} via {r.last_kernel_name}"
                assert total == want_total, what
                assert np.array_equal(spp, want_spp), what
                assert np.array_equal(bits(got), bits(want)), f"{what}: max|diff|={np.abs(got - want).max()}"
    assert SLAB in seen, seen


@pytest.mark.parametrize("mode", ["mip", "tf", "mip_tf", "top", "bottom", "mip_top"])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16_pack12"])
def test_slab_kernel_modes_and_views(vra, oracle, dtype, mode):
    rng = np.random.default_rng(5)
    dims, spacing = (72, 64, 80), (1.0, 1.0, 1.0)
    vol = rand_volume(rng, dims, dtype, smooth=True)
    vmax = 255 if dtype == np.uint8 else 4095
    mip, tf = "mip" in mode, "tf" in mode
    top, bottom = "top" in mode, "bottom" in mode
    size = (176, 120)
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.setVolume(vol, spacing)
        r.setWindow(3, vmax - 7)
        alpha = 0.3 if mip else 0.03
        r.setAlpha(alpha)
        r.setMIP(mip)
        tf_lut = None
        if tf:
            r.setTransferFunction([0, 90, 160, 255], [[0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])
            tf_lut = r.getTransferLut()
        r.setInitialCameraRotation(top, bottom)
        r.setKernelVariant(4)
        for name, block in cameras(oracle, rng, n_random=2):
            r.setCameraBlock(block)
            r.render()
            assert r.last_kernel_name == SLAB, (mode, name)
            got = r.readPixels()
            total, spp = r.countSamples(per_pixel=True)
            p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, min_val=3, max_val=vmax - 7, is_mip=int(mip),
                                    view_top=int(top), view_bottom=int(bottom), tf_rgba=tf_lut, threads=8)
            want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
            assert total == want_total and np.array_equal(spp, want_spp), (mode, name)
            assert np.array_equal(bits(got), bits(want)), f"{mode} {name}: max|diff|={np.abs(got - want).max()}"


def test_slab_equals_fast_kernel_on_shards_and_quirks(vra):
    """row stripes / contiguous shards / the truncated dispatch grid: the slab kernel and the fast kernel
    write the same pixels (and leave the same ones alone)"""
    rng = np.random.default_rng(9)
    vol = rand_volume(rng, (64, 64, 64), np.uint8, smooth=True)
    with vra.RendererCore(0) as r:
        r.setup((250, 170))
        assert r.loadShader("VolumeRenderer.cs")
        r.setVolume(vol)
        r.setAlpha(0.05)
        for quirks in (0, 1):
            r.setQuirks(quirks)
            for shard in ("full", "rows", "stripes"):
                r.setRowRange(0, -1); r.setRowStripes(1, 0, 1)
                if shard == "rows":
                    r.setRowRange(37, 121)
                elif shard == "stripes":
                    r.setRowStripes(8, 1, 3)
                frames = {}
                for variant in (4, 2):
                    r.setKernelVariant(variant)
                    r.setup((250, 170))                          # fresh (zeroed) target: untouched pixels must agree too
                    r.render()
                    frames[variant] = (r.last_kernel_name, r.readPixels().copy())
                assert frames[4][0] == SLAB and frames[2][0] == "raymarch_fast_kernel", (quirks, shard, frames[2][0], frames[4][0])
                assert np.array_equal(bits(frames[2][1]), bits(frames[4][1])), (quirks, shard)
