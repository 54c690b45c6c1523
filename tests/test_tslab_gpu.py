"""TRILINEAR on the LDS-staged kernel (vr_tslab.hip; vr_set_kernel_variant(6), and the automatic choice for
views aligned with a volume axis) against the CPU oracle, the batched trilinear kernel and the generic kernel.

What differs from every other trilinear path: the eight taps of a sample are LDS reads from a torus of apron-copy
brick slots (5x4x4 voxels each), the x1 tap is the x0 tap's neighbour in the slot, the clamps of GL's CLAMP_TO_EDGE
are replaced by u = max(f - 0.5, 0) and a duplicated last table entry -- so the cases below lean on volume faces
(cameras inside and next to the box, tiny volumes, dims that are not multiples of 4), on all modes and views, on
both voxel types, on the global-tap fallback (oblique views whose footprint does not fit the torus) and on shards.
Frames must be bit-identical and per-pixel fetch counts equal.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TSLAB = "raymarch_tslab_kernel"


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
    c = oracle.Camera()
    c.orient(1, 0, 0); c.orient(1, 0, 0)                        # eye at z = 1: close to the front face, wide footprint
    out.append(("close", c.block()))
    b = oracle.default_camera_block().copy()                    # eye inside the box (Q5): marching starts behind the eye
    b[12:15] = b[16:19] = (0.1, 0.05, 0.3)
    out.append(("inside", b))
    for k in range(n_random):
        c = oracle.Camera()
        c.orient(0, float(rng.uniform(-2, 2)), float(rng.uniform(-6, 6)))
        if k % 2:
            c.orient(1, 0, 0)
        out.append((f"random{k}", c.block()))
    return out


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("dims,spacing", [((64, 64, 64), (1, 1, 1)), ((96, 80, 72), (1.0, 0.8, 1.7)), ((130, 33, 47), (1, 1, 1)),
                                          ((40, 56, 24), (0.6, 1.0, 1.3)), ((7, 5, 3), (1, 1, 1)), ((1, 1, 1), (1, 1, 1)),
                                          ((2, 9, 4), (1, 1, 1)), ((256, 256, 256), (1, 1, 1))],
                         ids=["cube64", "noncubic", "odd_dims", "aniso", "tiny", "one_voxel", "thin", "cube256"])
# 7: every tile on the path of tiles that do not fit LDS; 8 / 9 / 10 (16-bit volumes; the 8-bit runs take variant 6's kernel): per-major-axis
# copies and half layers where whole ones do not fit, on 32x16 tiles / 32x32 tiles on a CU's whole LDS / 32x16 tiles on the whole LDS / 16x32 tiles (11)
@pytest.mark.parametrize("variant", [6, 7, 8, 9, 10, 11], ids=["staged", "unstaged", "half", "halftall", "three", "small"])
def test_tri_slab_kernel_matches_oracle(vra, oracle, dtype, dims, spacing, variant):
    rng = np.random.default_rng((sum(d * 31 ** k for k, d in enumerate(dims)) * 7 + np.dtype(dtype).itemsize) % (2 ** 32))
    vol = rand_volume(rng, dims, dtype, smooth=dims[0] >= 96)
    vmax = 255 if dtype == np.uint8 else 4095
    size = (200, 144) if dims[0] < 256 else (320, 200)
    seen = set()
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.setVolume(vol, spacing)
        r.setFilter(R.FILTER_TRILINEAR)
        for (lo, hi, alpha) in ((0, vmax, 0.02), (vmax // 5, vmax - vmax // 4, 1.0)):
            r.setWindow(lo, hi)
            r.setAlpha(alpha)
            for name, block in cameras(oracle, rng):
                r.setCameraBlock(block)
                r.setKernelVariant(variant)                      # TRILINEAR on the LDS-staged kernel wherever it is eligible
                r.render()
                seen.add(r.last_kernel_name)
                got = r.readPixels()
                total, spp = r.countSamples(per_pixel=True)
                p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, voxel_size=spacing, min_val=lo, max_val=hi, filter=1, threads=8)
                want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
                what = f"{dims} {np.dtype(dtype).name} window [{lo},{hi}] alpha {alpha} camera {name} via {r.last_kernel_name}"
                assert total == want_total, what
                assert np.array_equal(spp, want_spp), what
                assert np.array_equal(bits(got), bits(want)), f"{what}: max|diff|={np.abs(got - want).max()}, {int((bits(got) != bits(want)).sum())} words"
    assert seen == {TSLAB}, seen


@pytest.mark.parametrize("mode", ["mip", "tf", "mip_tf", "top", "bottom", "mip_top", "tf_bottom"])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("variant", [6, 7, 8, 9, 10, 11], ids=["staged", "unstaged", "half", "halftall", "three", "small"])
def test_tri_slab_kernel_modes_and_views(vra, oracle, dtype, mode, variant):
    rng = np.random.default_rng(5)
    dims, spacing = (72, 64, 80), (1.0, 1.0, 1.0)
    vol = rand_volume(rng, dims, dtype, smooth=True)
    vmax = 255 if dtype == np.uint8 else 4095
    mip, tf = "mip" in mode, "tf" in mode
    top, bottom = "top" in mode, "bottom" in mode
    size = (176, 120)
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.setVolume(vol, spacing)
        r.setFilter(R.FILTER_TRILINEAR)
        r.setWindow(3, vmax - 7)
        alpha = 0.3 if mip else 0.03
        r.setAlpha(alpha)
        r.setMIP(mip)
        tf_lut = None
        if tf:
            r.setTransferFunction([0, 90, 160, 255], [[0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])
            tf_lut = r.getTransferLut()
        r.setInitialCameraRotation(top, bottom)
        r.setKernelVariant(variant)
        for name, block in cameras(oracle, rng, n_random=2):
            r.setCameraBlock(block)
            r.render()
            assert r.last_kernel_name == TSLAB, (mode, name)
            got = r.readPixels()
            total, spp = r.countSamples(per_pixel=True)
            p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=alpha, min_val=3, max_val=vmax - 7, is_mip=int(mip),
                                    view_top=int(top), view_bottom=int(bottom), tf_rgba=tf_lut, filter=1, threads=8)
            want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
            assert total == want_total and np.array_equal(spp, want_spp), (mode, name)
            assert np.array_equal(bits(got), bits(want)), f"{mode} {name}: max|diff|={np.abs(got - want).max()}"


@pytest.mark.parametrize("dims", [(4, 6, 10400), (10400, 5, 4), (3, 10400, 8)], ids=["long_z", "long_x", "long_y"])
def test_tri_slab_unstaged_tiles_of_a_very_long_volume(vra, oracle, dims):
    """an axis so long that the unstaged path's per-axis offset tables (8 bytes per z index) do not fit LDS: those tiles
    fall back to the shader's literal taps from the resident volume; staged tiles (variant 6) keep range-restricted tables"""
    rng = np.random.default_rng(77)
    vol = rand_volume(rng, dims, np.uint8)
    spacing = tuple(float(v) for v in (64.0 / np.array(dims)))          # a cube in world units
    size = (128, 96)
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0); r.setVolume(vol, spacing); r.setFilter(R.FILTER_TRILINEAR); r.setWindow(0, 255); r.setAlpha(0.05)
        for name, block in cameras(oracle, rng, n_random=1)[:5]:
            p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=0.05, voxel_size=spacing, min_val=0, max_val=255, filter=1, threads=8)
            want, want_total, want_spp = oracle.render(vol, p, want_spp=True)
            for variant in (6, 7):
                r.setCameraBlock(block); r.setKernelVariant(variant); r.render()
                assert r.last_kernel_name == TSLAB
                got = r.readPixels()
                total, spp = r.countSamples(per_pixel=True)
                assert total == want_total and np.array_equal(spp, want_spp), (dims, name, variant)
                assert np.array_equal(bits(got), bits(want)), f"{dims} {name} variant {variant}: max|diff|={np.abs(got - want).max()}"


def test_tri_slab_equals_batched_kernel_on_shards_and_quirks(vra):
    """row stripes / contiguous shards / the truncated dispatch grid: the staged kernel and the batched trilinear kernel
    write the same pixels (and leave the same ones alone)"""
    rng = np.random.default_rng(9)
    vol = rand_volume(rng, (64, 64, 64), np.uint8, smooth=True)
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup((250, 170))
        assert r.loadShader("VolumeRenderer.cs")
        r.setVolume(vol)
        r.setAlpha(0.05)
        r.setFilter(R.FILTER_TRILINEAR)
        for quirks in (0, 1):
            r.setQuirks(quirks)
            for shard in ("full", "rows", "stripes"):
                r.setRowRange(0, -1); r.setRowStripes(1, 0, 1)
                if shard == "rows":
                    r.setRowRange(37, 121)
                elif shard == "stripes":
                    r.setRowStripes(8, 1, 3)
                frames = {}
                for variant in (6, 2):
                    r.setKernelVariant(variant)
                    r.setup((250, 170))                          # fresh (zeroed) target: untouched pixels must agree too
                    r.render()
                    frames[variant] = (r.last_kernel_name, r.readPixels().copy())
                assert frames[6][0] == TSLAB and frames[2][0] == "raymarch_tri_kernel", (quirks, shard, frames[2][0], frames[6][0])
                assert np.array_equal(bits(frames[2][1]), bits(frames[6][1])), (quirks, shard)


def test_tri_slab_full_size_cfg3(vra, oracle):
    """1024^3 u16 @ 1920x1080: staged == batched == generic frames bit for bit (default pose and the off-axis pose), equal sample
    counts, sparse rows against the oracle; the transfer-function mode no longer falls to the generic kernel"""
    R = vra.renderer
    W, H, N = 1920, 1080, 1024
    with vra.RendererCore(0) as r:
        r.setup((W, H))
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), 2, 0x9E3779B9)
        r.setWindow(0, 4095); r.setAlpha(0.004); r.setFilter(R.FILTER_TRILINEAR)
        for pose in ("default", "offaxis"):
            r.resetCamera()
            if pose == "offaxis":
                r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
            frames, counts, kernels, ms = {}, {}, {}, {}
            for name, variant in (("staged", 6), ("half", 8), ("halftall", 9), ("three", 10), ("small", 11), ("batched", 2), ("generic", 1)):
                r.setKernelVariant(variant)
                r.render(); r.kernelMsTake()
                for _ in range(3):
                    r.render()
                ms[name] = r.kernelMsTake() / 3
                kernels[name] = r.last_kernel_name
                frames[name] = r.readPixels().copy()
                counts[name] = r.countSamples()
            print(f"cfg3 trilinear {pose}: " + ", ".join(f"{k} {kernels[k]} {ms[k]:.3f} ms" for k in ms))
            assert kernels == {"staged": TSLAB, "half": TSLAB, "halftall": TSLAB, "three": TSLAB, "small": TSLAB, "batched": "raymarch_tri_kernel", "generic": "raymarch_generic_kernel"}
            assert counts["staged"] == counts["half"] == counts["halftall"] == counts["three"] == counts["batched"] == counts["generic"]
            for k in ("staged", "half", "halftall", "three"):
                assert np.array_equal(bits(frames[k]), bits(frames["generic"])), (pose, k)
            assert np.array_equal(bits(frames["batched"]), bits(frames["generic"])), pose
        # sparse rows against the oracle (default pose)
        r.resetCamera(); r.setKernelVariant(6)
        r.render()
        got = r.readPixels()
        vol = r.readVolume()
        block = r.getCameraBlock()
        rows = [140, 333, 540, 771, 939]
        p = oracle.OracleParams(W, H, cam=block, alpha_scale=0.004, min_val=0, max_val=4095, filter=1, threads=8)
        want = np.zeros((H, W, 4), dtype=np.float32)
        for y in rows:
            p.row_begin, p.row_end = y, y + 1
            oracle.render(vol, p, out=want)
        assert np.array_equal(bits(got[rows]), bits(want[rows]))
        # transfer function + trilinear: staged kernel, against the generic kernel
        r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]])
        r.setKernelVariant(6); r.render()
        assert r.last_kernel_name == TSLAB
        a = r.readPixels().copy()
        r.setKernelVariant(1); r.render()
        assert r.last_kernel_name == "raymarch_generic_kernel"
        assert np.array_equal(bits(a), bits(r.readPixels()))
