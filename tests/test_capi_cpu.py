"""CPU-side tests of the drop-in boundary: libvr_core.so loads, exports every symbol the
header declares, and the host logic (camera, .raw.inf grammar, transfer function, error
paths) behaves like the reference's RendererCore.  No compute call needs a GPU; GPU
operations must FAIL LOUDLY here (there is no CPU fallback).
"""
import ctypes as C
import os

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(vra):
    lib = vra.load_library()
    names = vra.symbols_declared_in_header()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert {"vr_create", "vr_setup", "vr_render", "vr_read_volume_file", "vr_set_volume", "vr_camera_orient",
            "vr_set_alpha", "vr_set_mip", "vr_set_view", "vr_set_window", "vr_kernel_ms_take", "vr_read_pixels",
            "vr_take_message", "vr_load_shader", "vr_save_image", "vr_read_pvm_volume"} <= set(names)


def test_no_cpu_fallback_gpu_operations_fail_loudly(vra):
    r = vra.RendererCore(-1)          # host-only handle
    r.setup((64, 64))
    assert r.loadShader("VolumeRenderer.cs")
    for call in (lambda: r.render(), lambda: r.renderAsync(), lambda: r.readPixels(),
                 lambda: r.setVolume(np.zeros((4, 4, 4), dtype=np.uint8)),
                 lambda: r.generateSynthetic(0, (8, 8, 8), 1, 3), lambda: r.countSamples(), lambda: r.histogram()):
        with pytest.raises(vra.VRError) as e:
            call()
        assert e.value.code == vra.renderer.VR_E_NO_DEVICE
    r.close()


def test_copy_budget_and_resident_bytes_on_a_host_only_handle(vra):
    """vr_set_copy_budget / vr_get_resident_bytes (round 5): defaults and round trip; nothing is resident without a device"""
    r = vra.RendererCore(-1)
    assert r.copyBudget() == r.COPY_BUDGET_AUTO
    r.setCopyBudget(3 << 30)
    assert r.copyBudget() == 3 << 30
    r.setCopyBudget(r.COPY_BUDGET_AUTO)
    assert r.copyBudget() == r.COPY_BUDGET_AUTO
    assert r.residentBytes() == (0, 0, 0)
    r.close()


def test_create_on_missing_device_reports_no_device(vra):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vra.VRError) as e:
        vra.RendererCore(0)
    assert e.value.code == vra.renderer.VR_E_NO_DEVICE


def test_camera_through_capi_matches_oracle_camera(vra, oracle):
    r = vra.RendererCore(-1)
    c = oracle.Camera()
    assert np.array_equal(r.getCameraBlock().view(np.uint32), c.block().view(np.uint32))
    rng = np.random.default_rng(0)
    for _ in range(200):
        kind = rng.integers(0, 4)
        if kind == 0:
            args = (float(rng.choice([-1.0, 1.0])), 0.0, 0.0)              # scroll zoom (GlfwManager.cpp:213)
        else:
            args = (0.0, float(rng.choice([-0.06, 0.0, 0.06])), float(rng.choice([-0.06, 0.06])))   # drag (:179)
        r.cameraOrient(*args)
        c.orient(*args)
        assert np.array_equal(r.getCameraBlock().view(np.uint32), c.block().view(np.uint32)), args
    r.resetCamera(); c.reset()
    assert np.array_equal(r.getCameraBlock().view(np.uint32), c.block().view(np.uint32))
    blk = np.arange(21, dtype=np.float32)
    r.setCameraBlock(blk)
    got = r.getCameraBlock()
    assert np.array_equal(got[:19], blk[:19]) and got[19] == 1.0 and got[20] == 20.0
    r.close()


def test_shader_and_workgroup_bookkeeping(vra):
    r = vra.RendererCore(-1)
    r.setup((1920, 1080))
    assert r.loaded_shader == ""
    assert r.loadShader("shaders/VolumeRenderer.cs")
    assert r.loaded_shader == "VolumeRenderer.cs"                     # basename, RendererCore.cpp:499-500
    assert r.takeMessage() == ("Shader Loaded!", "Shader Loaded Successfully!")
    assert r.takeMessage() is None
    assert r.workgroups == (120, 68)                                  # Q1 fixed: ceil-div
    r.setQuirks(vra.renderer.QUIRK_TRUNC_GRID)
    assert r.workgroups == (120, 67)                                  # reference: 1080/16 = 67 (RendererCore.cpp:121-122)
    assert not r.loadShader("")
    r.close()


def test_raw_inf_grammar_and_sidecar_writing(vra, tmp_path):
    r = vra.RendererCore(-1)
    r.setup((32, 32))
    raw = tmp_path / "vol.raw"
    raw.write_bytes(bytes(range(24)))
    assert not r.checkRawInfFile(raw)
    # no sidecar: dims/spacing typed into the GUI are used and the sidecar is written
    r.setDims(4, 3, 2); r.setSpacing(1.0, 0.5, 2.0)
    with pytest.raises(vra.VRError) as e:
        r.readVolumeData(raw, 1)
    assert e.value.code == vra.renderer.VR_E_NO_DEVICE                # parsed fine, upload needs a GPU
    assert r.checkRawInfFile(raw)
    text = (tmp_path / "vol.raw.inf").read_text()
    assert text == "#dimensions\n4 3 2\n\n#voxel-spacing\n1 0.5 2\n"
    # malformed sidecars produce the reference's GUI messages
    cases = {
        "#dimensions\n\n#voxel-spacing\n1 1 1\n": "Dimensions for Volume Data not provided",
        "#voxel-spacing\n1 1 1\n": "Make sure the header is \"#dimesnsions\"",
        "#dimensions\n4 3 2\n": "Make sure the header is \"#voxel-spacing\"",
        "#dimensions\n4 3 2\n#voxel-spacing\n\n": "Aspect Ratio for Volume Data not provided",
        "#dimensions\n0 3 2\n#voxel-spacing\n1 1 1\n": "shouldn't contain any zeroes",
    }
    for body, expect in cases.items():
        (tmp_path / "vol.raw.inf").write_text(body)
        with pytest.raises(vra.VRError) as e:
            r.readVolumeData(raw, 1)
        assert e.value.code == vra.renderer.VR_E_IO, body
        title, msg = r.takeMessage()
        assert expect in msg, (body, msg)
    # blank lines and unknown headers are tolerated
    (tmp_path / "vol.raw.inf").write_text("\n#comment\n#dimensions\n4 3 2\n\n\n#voxel-spacing\n1 1 1\n\n")
    with pytest.raises(vra.VRError) as e:
        r.readVolumeData(raw, 1)
    assert e.value.code == vra.renderer.VR_E_NO_DEVICE
    with pytest.raises(vra.VRError) as e:
        r.readVolumeData(tmp_path / "missing.raw", 1)
    with pytest.raises(vra.VRError):
        r.readVolumeData(raw, 3)                                       # datasize_bytes must be 1|2
    r.close()


def test_transfer_function_lut_matches_oracle_spline(vra, oracle):
    r = vra.RendererCore(-1)
    grey = r.getTransferLut()
    assert np.allclose(grey[:, 0], np.arange(256) / 255.0)
    iso = [0, 141, 149, 255]        # AlphaControlSplineWidget.cpp:56-59 default alpha knots
    rgba = [[0, 0, 0, 0], [0, 0, 0, 0.759], [0, 0, 0, 0.45], [0, 0, 0, 1]]
    r.setTransferFunction(iso, rgba)
    assert np.array_equal(r.getTransferLut().view(np.uint32), oracle.spline_tf(iso, rgba).view(np.uint32))
    rng = np.random.default_rng(5)
    for n in (2, 3, 7):
        iso = np.sort(rng.choice(256, size=n, replace=False))
        rgba = rng.random((n, 4)).astype(np.float32)
        r.setTransferFunction(iso, rgba)
        assert np.array_equal(r.getTransferLut().view(np.uint32), oracle.spline_tf(iso, rgba).view(np.uint32))
    with pytest.raises(vra.VRError):
        r.setTransferFunction([5, 5], [[0, 0, 0, 0], [1, 1, 1, 1]])    # iso values must ascend
    with pytest.raises(vra.VRError):
        r.setTransferFunction([5], [[0, 0, 0, 0]])
    r.setTransferFunction()                                            # back to the grey ramp
    assert np.array_equal(r.getTransferLut(), grey)
    r.close()


def test_argument_validation(vra):
    r = vra.RendererCore(-1)
    with pytest.raises(vra.VRError):
        r.setup((0, 10))
    for bad in (lambda: r.setFilter(7), lambda: r.setAccum(-1), lambda: r.setLayout(5), lambda: r.setRowRange(5, 2),
                lambda: r.setRowStripes(0, 0, 2), lambda: r.setRowStripes(8, 3, 2), lambda: r.setKernelVariant(12), lambda: r.setKernelVariant(4)):
        with pytest.raises(vra.VRError) as e:
            bad()
        assert e.value.code == vra.renderer.VR_E_INVALID
    r.close()
    lib = vra.load_library()
    assert lib.vr_render(None) == vra.renderer.VR_E_INVALID            # null handle
    assert lib.vr_create(None, -1) == vra.renderer.VR_E_INVALID


def test_image_writers_host_only(vra, tmp_path):
    """vr_write_image_rgb8: PNG/BMP/PPM are lossless; the baseline JPEG (quality 100, 4:4:4,
    Annex K tables) decodes with an independent decoder to within a few levels.  Formats of
    saveImage, src/RendererCore.cpp:173-178."""
    from PIL import Image

    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:61, 0:83]
    img = np.stack([(xx * 3) % 256, (yy * 4) % 256, (xx + yy) % 256], axis=-1).astype(np.uint8)
    img[20:40, 30:60] = rng.integers(0, 256, size=(20, 30, 3), dtype=np.uint8)      # noise patch: worst case for the DCT
    for ext in (".png", ".bmp", ".ppm"):
        assert vra.write_image_rgb8(tmp_path / f"a{ext}", ext, img)
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"a{ext}").convert("RGB")), img), ext
    assert vra.write_image_rgb8(tmp_path / "a.jpg", ".jpg", img)
    with Image.open(tmp_path / "a.jpg") as im:
        assert im.format == "JPEG" and im.size == (83, 61)
        got = np.asarray(im.convert("RGB")).astype(np.int32)
    err = np.abs(got - img.astype(np.int32))
    assert err.max() <= 6 and err.mean() < 1.0
    # 1x1 and non-multiple-of-8 sizes, grey image
    grey = np.full((1, 1, 3), 137, dtype=np.uint8)
    assert vra.write_image_rgb8(tmp_path / "g.jpg", ".jpg", grey)
    assert np.abs(np.asarray(Image.open(tmp_path / "g.jpg").convert("RGB")).astype(int) - 137).max() <= 1
    assert not vra.write_image_rgb8(tmp_path / "a.gif", ".gif", img)


def test_camera_set_view_matrix_mirrors_the_reference(vra):
    """Camera::setViewMatrix (src/Camera.cpp:46-57): columns side, up, -look_at, eye as GIVEN
    (not normalised), eye appended, view_plane_dist kept"""
    with vra.RendererCore(-1) as r:
        d = float(r.getCameraBlock()[20])
        eye, side, up, look = (1.0, 2.0, 3.0, 1.0), (2.0, 0.0, 0.0, 0.0), (0.0, 0.5, 0.0, 0.0), (0.0, 0.0, -4.0, 0.0)
        r.setViewMatrix(eye, side, up, look)
        b = r.getCameraBlock()
    want = np.array(list(side) + list(up) + [0.0, 0.0, 4.0, -0.0] + list(eye) + list(eye) + [d], dtype=np.float32)
    assert np.array_equal(b, want)


def test_file_parsers_survive_mutated_inputs_under_sanitizers(tmp_path):
    """tools/fuzz/fuzz_volume_io.cpp: the PVM/DDS decoder and the .raw.inf parser (the code that
    reads untrusted files) on a few thousand mutated fixtures, built with ASan + UBSan"""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fuzz_volume_io"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-I", os.path.join(root, "volume-renderer_amd", "csrc"),
                            os.path.join(root, "tools", "fuzz", "fuzz_volume_io.cpp"),
                            os.path.join(root, "volume-renderer_amd", "csrc", "volume_io.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not available")
    assert build.returncode == 0, build.stderr
    inf = tmp_path / "seed.inf"
    inf.write_text("#dimensions\n64 64 32\n#voxel-spacing\n1.0 1.0 2.5\n")
    gold = os.path.join(root, "tests", "golden")
    seeds = [os.path.join(gold, f) for f in ("pvm1_u8_9x7x5_noise.pvm", "pvm2_u16_8x6x4_scaled.pvm", "pvm3_u8_10x10x3_desc.pvm",
                                              "pvm1_u8_13x11x6_ramp.pvm")] + [str(inf)]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1")
    run = subprocess.run([str(exe), "4000"] + seeds, capture_output=True, text=True, env=env, timeout=300, cwd=str(tmp_path))
    assert run.returncode == 0 and "no crash" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]


def test_choices_blob_roundtrip_without_a_device(vra):
    """vr_export_choices / vr_import_choices on a handle without a device: an empty table is an 88-byte header, it imports
    back (0 entries), and malformed blobs are refused with VR_E_INVALID"""
    r = vra.RendererCore(-1)
    blob = r.exportChoices()
    assert len(blob) == 88 and blob[:8] == b"VRCHOICE"
    assert r.importChoices(blob) == 0
    for bad in (b"", b"VRCHOICE", b"XXCHOICE" + blob[8:], blob[:87]):
        with pytest.raises(vra.VRError) as e:
            r.importChoices(bad)
        assert e.value.code == vra.renderer.VR_E_INVALID
    r.close()
