"""GPU tests of the rows either side of the ray-march: RAW/.inf and PVM files
through readVolumeData onto the device, default window / histogram, screenshots, caller-owned
targets and streams."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def make(vra, size):
    r = vra.RendererCore(0)
    r.setup(size)
    assert r.loadShader("VolumeRenderer.cs")
    return r


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
def test_raw_file_roundtrip_and_render(vra, oracle, tmp_path, dtype):
    rng = np.random.default_rng(3)
    dims = (21, 34, 13)
    hi = 256 if dtype == np.uint8 else 3000
    vol = rng.integers(0, hi, size=(dims[2], dims[1], dims[0])).astype(dtype)
    raw = tmp_path / "scan.raw"
    raw.write_bytes(vol.tobytes())
    (tmp_path / "scan.raw.inf").write_text("#dimensions\n21 34 13\n\n#voxel-spacing\n1 0.9 1.4\n")
    with make(vra, (90, 70)) as r:
        r.readVolumeData(raw, vol.dtype.itemsize)
        assert r.takeMessage() == ("File Loaded!", "File Loaded Successfully!")
        assert r.loaded_dataset == "scan.raw"
        d, s, b = r.dims
        assert d == dims and b == vol.dtype.itemsize and s == pytest.approx((1.0, 0.9, 1.4))
        assert np.array_equal(r.readVolume(), vol)
        lo, hi_ = (0, 255) if dtype == np.uint8 else (int(vol.min()), int(vol.max()))
        assert r.window == (lo, hi_) and r.dataset_range == (lo, hi_)      # RendererCore.cpp:360-384
        r.setAlpha(0.1)
        r.render()
        got = r.readPixels()
        block = r.getCameraBlock()
    off = 1000 if dtype == np.uint16 else 0                                 # Q10 is on by default
    p = oracle.OracleParams(90, 70, cam=block, alpha_scale=0.1, voxel_size=(1.0, 0.9, 1.4), min_val=lo + off, max_val=hi_ + off)
    want, _ = oracle.render(vol, p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cfg2_sized_raw_file_through_the_reference_default_pipeline(vra, oracle, tmp_path):
    """BASELINE config 2's shape end to end the way the reference does it: a 512x512x452 uint16
    .raw + .raw.inf on disk (226 MB) -> readVolumeData -> default window = dataset range, +1000
    quirk on (src/RendererCore.cpp:66-69,360-384) -> 1920x1080 frame; sparse rows against the oracle"""
    dims = (512, 512, 452)
    vol = oracle.gen_noise_ball(dims, 2, 0x9E3779B9)
    raw = tmp_path / "head.raw"
    raw.write_bytes(vol.tobytes())
    (tmp_path / "head.raw.inf").write_text("#dimensions\n512 512 452\n\n#voxel-spacing\n1 1 1\n")
    W, H = 1920, 1080
    with make(vra, (W, H)) as r:
        r.readVolumeData(raw, 2)
        assert r.takeMessage() == ("File Loaded!", "File Loaded Successfully!")
        d, _, b = r.dims
        assert d == dims and b == 2
        lo, hi = r.window
        assert (lo, hi) == (int(vol.min()), int(vol.max())) == r.dataset_range
        r.setAlpha(0.05)
        r.render()
        assert r.last_kernel_name in ("raymarch_fast_kernel", "raymarch_relay_kernel")
        got = r.readPixels()
        block = r.getCameraBlock()
    p = oracle.OracleParams(W, H, cam=block, alpha_scale=0.05, min_val=lo + 1000, max_val=hi + 1000, threads=0)
    out = np.zeros((H, W, 4), dtype=np.float32)
    for y in (150, 400, 540, 700, 930):
        p.row_begin, p.row_end = y, y + 1
        oracle.render(vol, p, out=out)
        assert np.array_equal(got[y].view(np.uint32), out[y].view(np.uint32)), y


def test_pvm_file_to_device_and_render(vra, oracle):
    cases = {c["file"]: c for c in json.loads((GOLDEN / "pvm_manifest.json").read_text())}
    c = cases["pvm2_u16_8x6x4_scaled.pvm"]
    payload = np.load(GOLDEN / c["payload"])
    with make(vra, (64, 48)) as r:
        r.readVolumeData(GOLDEN / c["file"], 2)
        d, s, b = r.dims
        assert d == (8, 6, 4) and b == 2 and s == pytest.approx((1.0, 0.5, 2.0))
        # Q9: the big-endian 16-bit payload is reinterpreted in host order, no swap
        vol = payload.view("<u2").reshape(4, 6, 8)
        assert np.array_equal(r.readVolume(), vol)
        r.setQuirks(0)
        r.setWindow(int(vol.min()), int(vol.max()))
        r.render()
        got = r.readPixels()
        block = r.getCameraBlock()
        # an 8-bit PVM, and the component/datasize mismatch the reference would read out of bounds on
        c8 = cases["pvm3_u8_10x10x3_desc.pvm"]
        r.readVolumeData(GOLDEN / c8["file"], 1)
        assert np.array_equal(r.readVolume().ravel(), np.load(GOLDEN / c8["payload"]))
        with pytest.raises(vra.VRError):
            r.readVolumeData(GOLDEN / c8["file"], 2)
        assert r.takeMessage() == ("Error!", "Error reading PVM file")
    p = oracle.OracleParams(64, 48, cam=block, voxel_size=(1.0, 0.5, 2.0), min_val=int(vol.min()), max_val=int(vol.max()))
    want, _ = oracle.render(vol, p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_histogram_u16_matches_reference_formula(vra):
    rng = np.random.default_rng(8)
    vol = rng.integers(5, 3000, size=(18, 22, 26)).astype(np.uint16)
    with make(vra, (32, 32)) as r:
        r.setVolume(vol)
        h = r.histogram()
        mx = r.dataset_range[1]
    # val = round(val * 255.0f / max_dataset_val) as uint16; bin 0 skipped; normaliser starts at max_dataset_val
    bins = np.round(vol.astype(np.float32) * np.float32(255.0) / np.float32(mx)).astype(np.int64).ravel()
    counts = np.bincount(bins, minlength=256).astype(np.float64)[:256]
    counts[0] = 0
    norm = max(float(mx), counts.max())
    want = (counts.astype(np.float32) * np.float32(100.0) / np.float32(norm)).astype(np.float32)
    assert np.allclose(h, want, rtol=1e-6)


def test_stream_read_probe_reports_a_plausible_rate(vra):
    """vr_measure_stream_read: the box's achievable HBM read rate, used by bench.py's roofline"""
    with make(vra, (32, 32)) as r:
        r.generateSynthetic(vra.renderer.SYNTH_NOISE_BALL, (512, 512, 512), 2, 1)     # 256 MiB
        gbps = r.measureStreamRead(3)
    assert 200.0 < gbps < 8000.0 * 1.05       # above PCIe-class rates, not above the HBM3E datasheet peak


def test_save_image_png_bmp_ppm(vra, oracle, tmp_path):
    from PIL import Image

    vol = oracle.gen_sphere_u8(48, 20)
    with make(vra, (77, 52)) as r:       # odd width: row stride not a multiple of 4
        r.setVolume(vol)
        r.setAlpha(0.5)
        r.render()
        frame = r.readPixels()
        rgba8 = r.readPixelsRGBA8()              # device-side conversion, glReadPixels' rule
        for ext in (".png", ".bmp", ".ppm", ".jpg"):
            assert r.saveImage(tmp_path / f"shot{ext}", ext)
        assert not r.saveImage(tmp_path / "shot.gif", ".gif")
    want = np.floor(np.clip(frame[::-1, :, :3], 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)   # top row first
    for ext in (".png", ".bmp", ".ppm"):
        img = np.asarray(Image.open(tmp_path / f"shot{ext}").convert("RGB"))
        assert img.shape == (52, 77, 3) and np.array_equal(img, want), ext
    assert np.array_equal(rgba8[::-1, :, :3], want) and np.array_equal(rgba8[..., 3], np.floor(np.clip(frame[..., 3], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))
    jpg = np.asarray(Image.open(tmp_path / "shot.jpg").convert("RGB")).astype(np.int32)      # quality 100, 4:4:4
    assert jpg.shape == (52, 77, 3) and np.abs(jpg - want.astype(np.int32)).max() <= 3


def test_grey_alpha_target_is_the_same_frame_in_half_the_bytes(vra, oracle):
    """vr_set_framebuffer_format(VR_FB_GREYALPHA32F): (grey, alpha) per pixel into an external
    target, composite and MIP, fast and generic kernels; a transfer function is refused"""
    import torch

    R = vra.renderer
    sharding = __import__("importlib").import_module("volume-renderer_amd.sharding")
    vol = oracle.gen_noise_ball((48, 40, 44), 2, 9)
    W, H = 100, 70
    with make(vra, (W, H)) as r:
        r.setQuirks(0); r.setVolume(vol); r.setWindow(0, 4095); r.setAlpha(0.05)
        for mip, variant in ((False, 0), (True, 0), (False, 1)):
            r.setMIP(mip); r.setKernelVariant(variant)
            r.setFramebufferExternal(0); r.setFramebufferFormat(R.FB_RGBA32F)
            r.render(); want = r.readPixels().copy()
            ga = torch.full((H, W, 2), -1.0, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()          # the fill runs on torch's stream, the kernel on the renderer's own (non-blocking) one
            r.setFramebufferExternal(ga.data_ptr()); r.setFramebufferFormat(R.FB_GREYALPHA32F)
            r.render(); torch.cuda.synchronize()
            got = sharding.expand_grey_alpha(ga).cpu().numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (mip, variant)
        r.setMIP(False); r.setKernelVariant(0)
        r.setTransferFunction([0, 255], [[0, 0, 0, 0], [1, 0.5, 0.2, 1]])
        with pytest.raises(Exception):
            r.render()
        r.setFramebufferExternal(0); r.setFramebufferFormat(R.FB_RGBA32F)
        r.render()                                   # the RGBA target takes any mode


def test_external_target_stream_and_compact_shard(vra, oracle):
    import torch

    vol = oracle.gen_noise_ball((40, 40, 40), 1, 5)
    W, H = 96, 64
    with make(vra, (W, H)) as r:
        r.setVolume(vol)
        r.setAlpha(0.05)
        r.render()
        full = r.readPixels()
        stream = torch.cuda.Stream()
        r.setStream(stream.cuda_stream)
        tgt = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()              # (the zero fill is on torch's default stream: it must not land after the frame)
        r.setFramebufferExternal(tgt.data_ptr())
        assert r.framebufferDevice() == tgt.data_ptr()
        r.renderAsync()
        stream.synchronize()
        assert np.array_equal(tgt.cpu().numpy().view(np.uint32), full.view(np.uint32))
        # compact shard: rows 16..47 land at local rows 0..31 of a 32-row target
        part = torch.zeros((32, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.setFramebufferExternal(part.data_ptr())
        r.setFramebufferCompact(True)
        r.setRowRange(16, 48)
        assert r.localRows() == 32
        r.renderAsync()
        r.synchronize()
        assert np.array_equal(part.cpu().numpy().view(np.uint32), full[16:48].view(np.uint32))
        r.setStream(0)
        r.setFramebufferExternal(0)
        r.setFramebufferCompact(False)
        r.setRowRange(0, -1)
        r.render()
        assert np.array_equal(r.readPixels().view(np.uint32), full.view(np.uint32))


@pytest.mark.parametrize("which,datasize,size", [("VR_DATA_BONSAI", 1, (1280, 720)), ("VR_DATA_HEAD", 2, (1920, 1080))], ids=["cfg1_bonsai_raw", "cfg2_head_pvm"])
def test_reference_datasets_when_supplied(vra, oracle, which, datasize, size):
    """BASELINE configs 1 / 2 on the reference's own datasets (Bonsai .RAW, Head-CT .PVM; upstream git-ignores
    them, README.md:6-7): set VR_DATA_BONSAI / VR_DATA_HEAD to the files to run the reference's default pipeline
    on them (readVolumeData, default window, +1000 quirk for 16 bit) against the oracle on sparse rows.
    Skipped when the files are not on the box (synthetic stand-ins of the same shape run elsewhere)."""
    import os

    path = os.environ.get(which)
    name = "Bonsai 256x256x256 uint8 .raw (+ .raw.inf)" if which == "VR_DATA_BONSAI" else "Head CT 512x512x452 uint16 .pvm"
    if not path:
        pytest.skip(f"{which} is not set: the reference does not ship its datasets (upstream .gitignore:19-24) and this box has no network; "
                    f"export {which}=/path/to/the {name} file to run this test (README.md, 'Real datasets')")
    if not os.path.exists(path):
        pytest.skip(f"{which}={path} does not exist on this box (expected the {name} file; README.md, 'Real datasets')")
    with vra.RendererCore(0) as r:
        r.setup(size)
        assert r.loadShader("VolumeRenderer.cs")
        r.readVolumeData(path, datasize)
        assert r.takeMessage()[0] == "File Loaded!"
        lo, hi = r.window
        (dims, spacing, _) = r.dims
        r.setAlpha(0.05)
        r.render()
        got = r.readPixels()
        vol = r.readVolume()
        block = r.getCameraBlock()
    off = 1000 if datasize == 2 else 0                      # Q10: the uniform carries +1000 for 16-bit data
    p = oracle.OracleParams(size[0], size[1], cam=block, alpha_scale=0.05, voxel_size=tuple(spacing), min_val=lo + off, max_val=hi + off, threads=8)
    want = np.zeros_like(got)
    for y in range(size[1] // 8, size[1], size[1] // 8):
        p.row_begin, p.row_end = y, y + 1
        oracle.render(vol, p, out=want)
        assert np.array_equal(got[y].view(np.uint32), want[y].view(np.uint32)), f"{which} row {y}"


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world,channels", [("stripes", 3, 2), ("stripes", 8, 4), ("contiguous", 4, 2), ("contiguous", 3, 4)])
def test_assemble_shards_matches_the_index_path(vra, oracle, mode, world, channels):
    """vr_assemble_shards (one kernel on the receiving rank) == sharding's de-interleave + (grey, alpha) expansion,
    and the shards it is fed are real renders: every rank's compact shard of one frame, gathered by hand."""
    import torch

    sharding = __import__("importlib").import_module("volume-renderer_amd.sharding")
    vol = oracle.gen_noise_ball((48, 40, 44), 2, 11)
    W, H = 160, 100                      # 100 rows: not a multiple of the stripe height or the world size
    with make(vra, (W, H)) as r:
        r.setVolume(vol)
        r.setAlpha(0.03)
        r.render()
        full = r.readPixels()
        R = vra.renderer
        r.setFramebufferFormat(R.FB_GREYALPHA32F if channels == 2 else R.FB_RGBA32F)
        r.setFramebufferCompact(True)
        plans = [sharding.plan_rows(H, world, k, mode, 16) for k in range(world)]
        gathered = torch.full((world * plans[0].local_rows, W, channels), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for k, plan in enumerate(plans):
            sharding.apply_plan(r, plan)
            shard = gathered[k * plan.local_rows:(k + 1) * plan.local_rows]
            r.setFramebufferExternal(shard.data_ptr())
            r.render()
        torch.cuda.synchronize()
        frame = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            r.assembleShards(gathered.data_ptr(), frame.data_ptr(), world, plans[0].local_rows,
                             0 if mode == "contiguous" else 16, channels, stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(frame.cpu().numpy().view(np.uint32), full.view(np.uint32))
        # the torch path of sharding.gather_frame's tail gives the same frame
        idx = torch.as_tensor(sharding.gather_index(plans[0]), device="cuda")
        ref = gathered[:H] if mode == "contiguous" else gathered.index_select(0, idx)
        ref = sharding.expand_grey_alpha(ref) if channels == 2 else ref
        assert torch.equal(ref, frame)
        # bad geometry is refused, not launched
        with pytest.raises(Exception):
            r.assembleShards(gathered.data_ptr(), frame.data_ptr(), 1, 10, 0, channels, 0)
        r.setFramebufferExternal(0); r.setFramebufferCompact(False); r.setFramebufferFormat(R.FB_RGBA32F)
        r.setRowRange(0, -1); r.setRowStripes(1, 0, 1)


def test_present_rgba8_is_double_buffered_and_one_frame_late(vra):
    """vr_present_rgba8 (the shim's presentation path): call k returns the frame of call k - 1 (the first call its own), equal to
    vr_read_pixels_rgba8 of that frame byte for byte; the group's variant does the same for the assembled frame"""
    rng = np.random.default_rng(8)
    vol = rng.integers(0, 256, size=(40, 48, 56), dtype=np.uint8)
    size = (203, 157)
    poses = [(0.3, 0.66), (0.12, -0.42), (-0.54, 0.18), (0.0, 0.0), (0.06, 0.06)]

    def configure(r):
        assert r.loadShader("VolumeRenderer.cs")
        r.setQuirks(0); r.setVolume(vol); r.setWindow(5, 240); r.setAlpha(0.05)

    with vra.RendererCore(0) as r:
        r.setup(size); configure(r)
        want = []
        held = None                                         # a zero-copy view handed out one call earlier: still that frame (valid until the next-but-one call)
        for k, (ze, az) in enumerate(poses):
            r.resetCamera(); r.cameraOrient(0, ze, az)
            r.render()
            want.append(r.readPixelsRGBA8())
            got = r.presentRGBA8(copy=False)
            assert np.array_equal(got, want[max(k - 1, 0)]), k
            if held is not None:
                r.synchronize()                             # this call's copy has landed: it went to another buffer
                assert np.array_equal(held, want[max(k - 2, 0)]), ("held", k)
            held = got
        assert not np.array_equal(want[0], want[2])
        r.setup((96, 64)); r.render()                      # a new size re-allocates the presentation buffers
        assert np.array_equal(r.presentRGBA8(), r.readPixelsRGBA8())
    with vra.RendererGroup([0, 0, 0]) as g:
        g.setup(size, stripe_rows=8)
        g.each(configure)
        for k, (ze, az) in enumerate(poses):
            g.each(lambda m: (m.resetCamera(), m.cameraOrient(0, ze, az)))
            g.render()
            assert np.array_equal(g.presentRGBA8(), want[max(k - 1, 0)]), ("group", k)
