"""ctypes binding of libvr_core.so (include/vr_core.h).

``RendererCore`` mirrors the reference class of the same name
(/root/reference/include/RendererCore.h:9-44): method names and argument meaning
follow the reference (``setup``, ``render``, ``readVolumeData``, ``setAlpha`` ...),
GUI-poked fields are properties.  Every call goes through the C ABI; if the shared
library is missing the import of this module fails loudly (no fallback path).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from pathlib import Path

import numpy as np

_PKG_DIR = Path(__file__).resolve().parent
_REPO_ROOT = _PKG_DIR.parent
# VR_CORE_LIB: experiment hook -- load an alternative build of the same C ABI
LIB_PATH = Path(os.environ["VR_CORE_LIB"]).resolve() if os.environ.get("VR_CORE_LIB") else _PKG_DIR / "lib" / "libvr_core.so"
HEADER_PATH = _REPO_ROOT / "include" / "vr_core.h"

VR_OK, VR_E_INVALID, VR_E_NO_DEVICE, VR_E_HIP, VR_E_IO, VR_E_NOMEM = range(6)
FILTER_NEAREST, FILTER_TRILINEAR = 0, 1
ACCUM_ITERATIVE, ACCUM_CLOSED_FORM = 0, 1
LAYOUT_LINEAR, LAYOUT_BRICKED = 0, 1
FB_RGBA32F, FB_GREYALPHA32F = 0, 1     # vr_set_framebuffer_format
SYNTH_SPHERE_U8, SYNTH_NOISE_BALL, SYNTH_NOISE_BALL_CT = 0, 1, 2
QUIRK_TRUNC_GRID, QUIRK_U16_OFFSET = 1, 2
QUIRK_DEFAULT = QUIRK_U16_OFFSET


class VRError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"vr_core error {code}: {message}")
        self.code = code
        self.message = message


def build_library(force: bool = False) -> Path:
    """Compile every HIP/C++ source of the package for gfx950 (in-tree)."""
    if force:
        subprocess.run(["make", "-C", str(_PKG_DIR), "clean"], check=True, capture_output=True)
    proc = subprocess.run(["make", "-C", str(_PKG_DIR)], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libvr_core.so failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


def symbols_declared_in_header() -> list[str]:
    """Names of all functions declared in include/vr_core.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C volume-renderer_amd). There is no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  If this process is going to use
    # torch as well (bench.py, RCCL), torch's copy must be the first one loaded: libvr_core.so
    # then binds to it by soname and the process has ONE runtime.  The other order leaves torch
    # unable to see the device.  (Plain C/C++ clients of the library do not involve torch.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    h = C.c_void_p
    f32, i32, u32, u64 = C.c_float, C.c_int, C.c_uint32, C.c_uint64
    sigs = {
        "vr_create": (i32, [C.POINTER(h), i32]),
        "vr_destroy": (None, [h]),
        "vr_setup": (i32, [h, i32, i32, i32, i32]),
        "vr_last_error": (C.c_char_p, [h]),
        "vr_take_message": (i32, [h, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
        "vr_camera_orient": (i32, [h, f32, f32, f32]),
        "vr_camera_reset": (i32, [h]),
        "vr_camera_set_block": (i32, [h, C.POINTER(f32)]),
        "vr_camera_set_view_matrix": (i32, [h, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]),
        "vr_camera_get_block": (i32, [h, C.POINTER(f32)]),
        "vr_load_shader": (i32, [h, C.c_char_p, i32]),
        "vr_workgroups": (i32, [h, C.POINTER(i32), C.POINTER(i32)]),
        "vr_loaded_shader": (C.c_char_p, [h]),
        "vr_loaded_dataset": (C.c_char_p, [h]),
        "vr_check_raw_inf_file": (i32, [h, C.c_char_p]),
        "vr_set_dims": (i32, [h, i32, i32, i32]),
        "vr_set_spacing": (i32, [h, f32, f32, f32]),
        "vr_read_volume_file": (i32, [h, C.c_char_p, i32]),
        "vr_set_volume": (i32, [h, C.c_void_p, i32, i32, i32, i32, f32, f32, f32]),
        "vr_generate_synthetic": (i32, [h, i32, i32, i32, i32, i32, u32]),
        "vr_read_volume": (i32, [h, C.c_void_p, C.c_size_t]),
        "vr_get_dims": (i32, [h, C.POINTER(i32), C.POINTER(f32), C.POINTER(i32)]),
        "vr_get_dataset_range": (i32, [h, C.POINTER(i32), C.POINTER(i32)]),
        "vr_histogram": (i32, [h, C.POINTER(f32)]),
        "vr_measure_stream_read": (i32, [h, i32, C.POINTER(C.c_double)]),
        "vr_assemble_shards": (i32, [h, C.c_void_p, C.c_void_p, i32, i32, i32, i32, C.c_void_p]),
        "vr_set_alpha": (i32, [h, f32]),
        "vr_set_mip": (i32, [h, i32]),
        "vr_set_view": (i32, [h, i32, i32]),
        "vr_set_window": (i32, [h, i32, i32]),
        "vr_get_window": (i32, [h, C.POINTER(i32), C.POINTER(i32)]),
        "vr_set_filter": (i32, [h, i32]),
        "vr_set_accum": (i32, [h, i32]),
        "vr_set_quirks": (i32, [h, u32]),
        "vr_set_layout": (i32, [h, i32]),
        "vr_set_skip_empty": (i32, [h, i32]),
        "vr_set_kernel_variant": (i32, [h, i32]),
        "vr_set_autotune": (i32, [h, i32]),
        "vr_export_choices": (i32, [h, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
        "vr_import_choices": (i32, [h, C.c_void_p, C.c_size_t, C.POINTER(i32)]),
        "vr_get_launch_choice": (i32, [h]),
        "vr_get_resident_bytes": (i32, [h, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "vr_set_copy_budget": (i32, [h, C.c_uint64]),
        "vr_get_copy_budget": (i32, [h, C.POINTER(C.c_uint64)]),
        "vr_set_pack12": (i32, [h, i32]),
        "vr_get_pack12_bytes": (i32, [h, C.POINTER(C.c_size_t)]),
        "vr_set_trilinear_copy": (i32, [h, i32]),
        "vr_get_trilinear_copy_bytes": (i32, [h, C.POINTER(C.c_size_t)]),
        "vr_set_transfer_function": (i32, [h, C.POINTER(C.c_int32), C.POINTER(f32), i32]),
        "vr_get_transfer_lut": (i32, [h, C.POINTER(f32)]),
        "vr_set_row_range": (i32, [h, i32, i32]),
        "vr_set_row_stripes": (i32, [h, i32, i32, i32]),
        "vr_set_framebuffer_compact": (i32, [h, i32]),
        "vr_set_framebuffer_format": (i32, [h, i32]),
        "vr_local_rows": (i32, [h]),
        "vr_set_stream": (i32, [h, C.c_void_p]),
        "vr_set_framebuffer_external": (i32, [h, C.c_void_p]),
        "vr_render": (i32, [h]),
        "vr_render_async": (i32, [h]),
        "vr_synchronize": (i32, [h]),
        "vr_kernel_ms_take": (f32, [h]),
        "vr_count_samples": (i32, [h, C.POINTER(u64), C.POINTER(u32), C.c_size_t]),
        "vr_framebuffer_device": (C.c_void_p, [h]),
        "vr_read_pixels": (i32, [h, C.POINTER(f32), C.c_size_t]),
        "vr_read_pixels_rgba8": (i32, [h, C.c_void_p, C.c_size_t]),
        "vr_present_rgba8": (i32, [h, C.POINTER(C.c_void_p)]),
        "vr_group_present_rgba8": (i32, [h, C.POINTER(C.c_void_p)]),
        "vr_save_image": (i32, [h, C.c_char_p, C.c_char_p]),
        "vr_last_kernel_name": (C.c_char_p, [h]),
        "vr_read_pvm_volume": (C.c_void_p, [C.c_char_p] + [C.POINTER(C.c_uint)] * 4 + [C.POINTER(f32)] * 3),
        "vr_checksum": (C.c_uint, [C.c_void_p, C.c_uint]),
        "vr_write_image_rgb8": (i32, [C.c_char_p, C.c_char_p, i32, i32, C.c_void_p, i32]),
        "vr_free": (None, [C.c_void_p]),
        "vr_group_create": (i32, [C.POINTER(h), C.POINTER(i32), i32]),
        "vr_group_destroy": (None, [h]),
        "vr_group_size": (i32, [h]),
        "vr_group_member": (h, [h, i32]),
        "vr_group_setup": (i32, [h, i32, i32, i32, i32, i32, i32]),
        "vr_group_set_transport": (i32, [h, i32]),
        "vr_group_transport": (C.c_char_p, [h]),
        "vr_group_render": (i32, [h]),
        "vr_group_render_async": (i32, [h]),
        "vr_group_wait": (i32, [h]),
        "vr_group_kernel_ms_take": (f32, [h]),
        "vr_group_framebuffer_device": (C.c_void_p, [h]),
        "vr_group_read_pixels": (i32, [h, C.POINTER(f32), C.c_size_t]),
        "vr_group_last_error": (C.c_char_p, [h]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)   # AttributeError = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def read_pvm_volume(path):
    """readPVMvolume through the C ABI -> (payload uint8 array, (w, h, d, components), (sx, sy, sz)) or None"""
    lib = load_library()
    w, h, d, c = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
    sx, sy, sz = C.c_float(), C.c_float(), C.c_float()
    p = lib.vr_read_pvm_volume(str(path).encode(), w, h, d, c, sx, sy, sz)
    if not p:
        return None
    n = w.value * h.value * d.value * c.value
    data = np.frombuffer((C.c_ubyte * n).from_address(p), dtype=np.uint8).copy()
    lib.vr_free(p)
    return data, (w.value, h.value, d.value, c.value), (sx.value, sy.value, sz.value)


def write_image_rgb8(path, ext: str, rgb: np.ndarray) -> bool:
    """the writers behind saveImage for an [H, W, 3] uint8 image (host only)"""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = a.shape[:2]
    return load_library().vr_write_image_rgb8(str(path).encode(), ext.encode(), w, h, a.ctypes.data, w * 3) == VR_OK


def checksum(data: np.ndarray) -> int:
    d = np.ascontiguousarray(data, dtype=np.uint8)
    return int(load_library().vr_checksum(d.ctypes.data, d.size))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class RendererGroup:
    """vr_group_*: one process, one renderer per device, image rows sharded, shards gathered on
    devices[0] (RCCL over xGMI when the devices are distinct).  Members are RendererCore objects
    borrowed from the group: configure every member alike (volume, window, camera ...)."""

    def __init__(self, devices):
        self._lib = load_library()
        self._g = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.vr_group_create(C.byref(self._g), arr, len(devices))
        if rc != VR_OK:
            raise VRError(rc, "vr_group_create failed")
        self.members = [RendererCore._borrowed(self._lib, self._lib.vr_group_member(self._g, r), devices[r]) for r in range(len(devices))]
        self.framebuffer_size = (0, 0)

    def _check(self, rc):
        if rc != VR_OK:
            raise VRError(rc, (self._lib.vr_group_last_error(self._g) or b"").decode())

    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m._h = None
            self._lib.vr_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setup(self, window_size, framebuffer_size=None, partition="stripes", stripe_rows=16):
        fb = framebuffer_size or window_size
        self._check(self._lib.vr_group_setup(self._g, window_size[0], window_size[1], fb[0], fb[1], 0 if partition == "stripes" else 1, stripe_rows))
        self.framebuffer_size = (int(fb[0]), int(fb[1]))
        for m in self.members:
            m.framebuffer_size = self.framebuffer_size

    def setTransport(self, mode: int):
        """0 peer copies, 1 RCCL when the devices are distinct (default), 2 also create the communicator of a one-member group"""
        self._check(self._lib.vr_group_set_transport(self._g, int(mode)))

    @property
    def transport(self) -> str:
        return (self._lib.vr_group_transport(self._g) or b"").decode()

    def each(self, fn):
        """apply fn(member) to every member (the group keeps them in step only for what it owns: rows, targets)"""
        for m in self.members:
            fn(m)

    def render(self):
        self._check(self._lib.vr_group_render(self._g))

    def renderAsync(self):
        """enqueue the next frame (at most two in flight); wait() completes the oldest"""
        self._check(self._lib.vr_group_render_async(self._g))

    def wait(self):
        self._check(self._lib.vr_group_wait(self._g))

    def kernelMsTake(self) -> float:
        return float(self._lib.vr_group_kernel_ms_take(self._g))

    def framebufferDevice(self) -> int:
        return int(self._lib.vr_group_framebuffer_device(self._g) or 0)

    def readPixels(self) -> np.ndarray:
        w, h = self.framebuffer_size
        out = np.zeros((h, w, 4), dtype=np.float32)
        self._check(self._lib.vr_group_read_pixels(self._g, _fp(out), out.size))
        return out

    def presentRGBA8(self) -> np.ndarray:
        """vr_group_present_rgba8: like RendererCore.presentRGBA8, of the assembled frame"""
        w, h = self.framebuffer_size
        ptr = C.c_void_p()
        self._check(self._lib.vr_group_present_rgba8(self._g, C.byref(ptr)))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, w, 4)).copy()


class RendererCore:
    """Host mirror of the reference's RendererCore over the C ABI.

    device=-1 creates a host-only handle (camera / parsing / transfer function only).
    """

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.vr_create(C.byref(self._h), int(device))
        if rc != VR_OK:
            msg = self._lib.vr_last_error(None)
            raise VRError(rc, (msg or b"").decode())
        self.device = device
        self.framebuffer_size = (0, 0)

    @classmethod
    def _borrowed(cls, lib, handle, device):
        """a RendererCore around a handle somebody else owns (vr_group_member): close() leaves it alone"""
        self = cls.__new__(cls)
        self._lib = lib
        self._h = C.c_void_p(handle)
        self._owned = False
        self.device = device
        self.framebuffer_size = (0, 0)
        return self

    # -- plumbing
    def _check(self, rc: int):
        if rc != VR_OK:
            raise VRError(rc, (self._lib.vr_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._lib.vr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- RendererCore::setup (window_size / framebuffer_size are set first by the GUI)
    def setup(self, window_size, framebuffer_size=None):
        fb = framebuffer_size or window_size
        self._check(self._lib.vr_setup(self._h, window_size[0], window_size[1], fb[0], fb[1]))
        self.framebuffer_size = (int(fb[0]), int(fb[1]))

    def takeMessage(self):
        t, m = C.create_string_buffer(256), C.create_string_buffer(1024)
        if self._lib.vr_take_message(self._h, t, 256, m, 1024):
            return t.value.decode(), m.value.decode()
        return None

    # -- camera
    def cameraOrient(self, zoom, zenith, azimuth):
        self._check(self._lib.vr_camera_orient(self._h, zoom, zenith, azimuth))

    def resetCamera(self):
        self._check(self._lib.vr_camera_reset(self._h))

    def setViewMatrix(self, eye, side, up, look_at):
        """Camera::setViewMatrix: four xyzw vectors"""
        v = [np.ascontiguousarray(x, dtype=np.float32) for x in (eye, side, up, look_at)]
        assert all(x.size == 4 for x in v)
        self._check(self._lib.vr_camera_set_view_matrix(self._h, *[_fp(x) for x in v]))

    def setCameraBlock(self, block21):
        b = np.ascontiguousarray(block21, dtype=np.float32)
        assert b.size == 21
        self._check(self._lib.vr_camera_set_block(self._h, _fp(b)))

    def getCameraBlock(self) -> np.ndarray:
        b = np.zeros(21, dtype=np.float32)
        self._check(self._lib.vr_camera_get_block(self._h, _fp(b)))
        return b

    # -- shader
    def loadShader(self, fn="VolumeRenderer.cs", reload=False) -> bool:
        rc = self._lib.vr_load_shader(self._h, fn.encode(), int(reload))
        if rc == VR_E_IO:
            return False
        self._check(rc)
        return True

    @property
    def workgroups(self):
        x, y = C.c_int(), C.c_int()
        self._lib.vr_workgroups(self._h, C.byref(x), C.byref(y))
        return x.value, y.value

    @property
    def loaded_shader(self) -> str:
        return (self._lib.vr_loaded_shader(self._h) or b"").decode()

    @property
    def loaded_dataset(self) -> str:
        return (self._lib.vr_loaded_dataset(self._h) or b"").decode()

    # -- volume
    def checkRawInfFile(self, fn) -> bool:
        return bool(self._lib.vr_check_raw_inf_file(self._h, str(fn).encode()))

    def setDims(self, nx, ny, nz):
        self._check(self._lib.vr_set_dims(self._h, nx, ny, nz))

    def setSpacing(self, sx, sy, sz):
        self._check(self._lib.vr_set_spacing(self._h, sx, sy, sz))

    def readVolumeData(self, fn, datasize_bytes: int):
        self._check(self._lib.vr_read_volume_file(self._h, str(fn).encode(), datasize_bytes))

    def setVolume(self, voxels: np.ndarray, spacing=(1.0, 1.0, 1.0)):
        """voxels: array indexed [z, y, x] (x fastest), dtype uint8 or uint16."""
        v = np.ascontiguousarray(voxels)
        assert v.ndim == 3 and v.dtype in (np.uint8, np.uint16)
        nz, ny, nx = v.shape
        self._check(self._lib.vr_set_volume(self._h, v.ctypes.data_as(C.c_void_p), nx, ny, nz, v.dtype.itemsize,
                                            spacing[0], spacing[1], spacing[2]))

    def generateSynthetic(self, kind, dims, datasize_bytes, param):
        self._check(self._lib.vr_generate_synthetic(self._h, kind, dims[0], dims[1], dims[2], datasize_bytes, param))

    def readVolume(self) -> np.ndarray:
        dims, _, b = self.dims
        out = np.empty((dims[2], dims[1], dims[0]), dtype=np.uint8 if b == 1 else np.uint16)
        self._check(self._lib.vr_read_volume(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    @property
    def dims(self):
        d = (C.c_int * 3)()
        s = (C.c_float * 3)()
        b = C.c_int()
        self._lib.vr_get_dims(self._h, d, s, C.byref(b))
        return tuple(d), tuple(s), b.value

    @property
    def dataset_range(self):
        a, b = C.c_int(), C.c_int()
        self._lib.vr_get_dataset_range(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def histogram(self) -> np.ndarray:
        out = np.zeros(256, dtype=np.float32)
        self._check(self._lib.vr_histogram(self._h, _fp(out)))
        return out

    def measureStreamRead(self, reps: int = 5) -> float:
        """achievable HBM read rate on this box (GB/s): streaming read of the resident volume"""
        g = C.c_double()
        self._check(self._lib.vr_measure_stream_read(self._h, reps, C.byref(g)))
        return g.value

    def assembleShards(self, gathered_ptr, frame_ptr, n, local_rows, stripe_rows, channels, stream=0):
        """rank-major gathered shards (device) -> the W x H RGBA32F frame (device): one kernel on `stream`"""
        self._check(self._lib.vr_assemble_shards(self._h, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr), n, local_rows,
                                                 stripe_rows, channels, C.c_void_p(stream)))

    # -- uniforms
    def setAlpha(self, alpha_scale):
        self._check(self._lib.vr_set_alpha(self._h, alpha_scale))

    def setMIP(self, use_mip):
        self._check(self._lib.vr_set_mip(self._h, int(bool(use_mip))))

    def setInitialCameraRotation(self, rotate_to_top=False, rotate_to_bottom=False):
        self._check(self._lib.vr_set_view(self._h, int(bool(rotate_to_top)), int(bool(rotate_to_bottom))))

    def setWindow(self, min_val, max_val):
        self._check(self._lib.vr_set_window(self._h, int(min_val), int(max_val)))

    @property
    def window(self):
        a, b = C.c_int(), C.c_int()
        self._lib.vr_get_window(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    # -- build-defined switches
    def setFilter(self, f):
        self._check(self._lib.vr_set_filter(self._h, f))

    def setAccum(self, a):
        self._check(self._lib.vr_set_accum(self._h, a))

    def setQuirks(self, q):
        self._check(self._lib.vr_set_quirks(self._h, q))

    def setLayout(self, layout):
        self._check(self._lib.vr_set_layout(self._h, layout))

    def setSkipEmpty(self, on):
        self._check(self._lib.vr_set_skip_empty(self._h, int(bool(on))))

    @property
    def last_launch_choice(self):
        """candidate bits of the last launch: 1 relay, 2 pipelined loop, 4 short batches, staged-trilinear shape << 3,
        256 = a trial frame (the measured choice is still exploring this configuration)"""
        return int(self._lib.vr_get_launch_choice(self._h))

    def setKernelVariant(self, variant):
        self._check(self._lib.vr_set_kernel_variant(self._h, variant))

    def exportChoices(self) -> bytes:
        """settled entries of the measured launch choice (vr_export_choices)"""
        n = C.c_size_t(0)
        self._check(self._lib.vr_export_choices(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._check(self._lib.vr_export_choices(self._h, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def importChoices(self, blob: bytes) -> int:
        """entries accepted (0: measured on another device model / library build)"""
        acc = C.c_int(0)
        buf = C.create_string_buffer(bytes(blob), len(blob))
        self._check(self._lib.vr_import_choices(self._h, buf, len(blob), C.byref(acc)))
        return acc.value

    def setAutotune(self, on):
        """kernel variant 0: measure the candidate kernels on the first frames of a configuration and keep the fastest (default on)"""
        self._check(self._lib.vr_set_autotune(self._h, 1 if on else 0))

    def setPack12(self, on):
        self._check(self._lib.vr_set_pack12(self._h, int(bool(on))))

    def pack12Bytes(self) -> int:
        """bytes of the 12-bit packed copy the last launch gathered from (0 = none)"""
        n = C.c_size_t()
        self._check(self._lib.vr_get_pack12_bytes(self._h, C.byref(n)))
        return int(n.value)

    COPY_BUDGET_AUTO = 2 ** 64 - 1

    def residentBytes(self):
        """(volume, optional copies, everything else) bytes of device memory this handle holds"""
        v, k, o = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.vr_get_resident_bytes(self._h, C.byref(v), C.byref(k), C.byref(o)))
        return v.value, k.value, o.value

    def setCopyBudget(self, nbytes):
        self._check(self._lib.vr_set_copy_budget(self._h, C.c_uint64(int(nbytes))))

    def copyBudget(self) -> int:
        b = C.c_uint64()
        self._check(self._lib.vr_get_copy_budget(self._h, C.byref(b)))
        return b.value

    def setTrilinearCopy(self, on):
        self._check(self._lib.vr_set_trilinear_copy(self._h, int(bool(on))))

    def trilinearCopyBytes(self) -> int:
        """bytes of the apron copy the last TRILINEAR launch gathered from (0 = none)"""
        n = C.c_size_t()
        self._check(self._lib.vr_get_trilinear_copy_bytes(self._h, C.byref(n)))
        return int(n.value)

    def setTransferFunction(self, iso=None, rgba=None):
        if iso is None or len(iso) == 0:
            self._check(self._lib.vr_set_transfer_function(self._h, None, None, 0))
            return
        i = np.ascontiguousarray(iso, dtype=np.int32)
        c = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
        assert c.shape[0] == i.size
        self._check(self._lib.vr_set_transfer_function(self._h, i.ctypes.data_as(C.POINTER(C.c_int32)), _fp(c), i.size))

    def getTransferLut(self) -> np.ndarray:
        out = np.zeros((256, 4), dtype=np.float32)
        self._check(self._lib.vr_get_transfer_lut(self._h, _fp(out)))
        return out

    def setRowRange(self, row_begin, row_end):
        self._check(self._lib.vr_set_row_range(self._h, row_begin, row_end))

    def setRowStripes(self, stripe_rows, index, count):
        self._check(self._lib.vr_set_row_stripes(self._h, stripe_rows, index, count))

    def setFramebufferCompact(self, on):
        self._check(self._lib.vr_set_framebuffer_compact(self._h, int(bool(on))))

    def setFramebufferFormat(self, fmt):
        """FB_RGBA32F or FB_GREYALPHA32F (external targets; grey modes only)"""
        self._check(self._lib.vr_set_framebuffer_format(self._h, int(fmt)))

    def localRows(self) -> int:
        return int(self._lib.vr_local_rows(self._h))

    def setStream(self, hip_stream_ptr):
        self._check(self._lib.vr_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def setFramebufferExternal(self, device_ptr):
        self._check(self._lib.vr_set_framebuffer_external(self._h, C.c_void_p(device_ptr)))

    # -- render
    def render(self):
        self._check(self._lib.vr_render(self._h))

    def renderAsync(self):
        self._check(self._lib.vr_render_async(self._h))

    def synchronize(self):
        self._check(self._lib.vr_synchronize(self._h))

    def kernelMsTake(self) -> float:
        return float(self._lib.vr_kernel_ms_take(self._h))

    def countSamples(self, per_pixel=False):
        w, h = self.framebuffer_size
        total = C.c_uint64()
        if per_pixel:
            spp = np.zeros((h, w), dtype=np.uint32)
            self._check(self._lib.vr_count_samples(self._h, C.byref(total), spp.ctypes.data_as(C.POINTER(C.c_uint32)), spp.size))
            return total.value, spp
        self._check(self._lib.vr_count_samples(self._h, C.byref(total), None, 0))
        return total.value

    def framebufferDevice(self) -> int:
        return int(self._lib.vr_framebuffer_device(self._h) or 0)

    def readPixels(self) -> np.ndarray:
        w, h = self.framebuffer_size
        out = np.zeros((h, w, 4), dtype=np.float32)
        self._check(self._lib.vr_read_pixels(self._h, _fp(out), out.size))
        return out

    def readPixelsRGBA8(self) -> np.ndarray:
        """the frame as uint8 [H, W, 4] (row 0 = bottom), converted on the device"""
        w, h = self.framebuffer_size
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._check(self._lib.vr_read_pixels_rgba8(self._h, out.ctypes.data, out.size))
        return out

    def presentRGBA8(self, copy=True) -> np.ndarray:
        """vr_present_rgba8: enqueue this frame's RGBA8 conversion + pinned D2H copy, return the PREVIOUS call's frame as
        uint8 [H, W, 4] -- the first call returns its own frame.  copy=False: a view of the library's pinned buffer (valid
        until the next-but-one call), what a GUI would hand to glTexImage2D"""
        w, h = self.framebuffer_size
        ptr = C.c_void_p()
        self._check(self._lib.vr_present_rgba8(self._h, C.byref(ptr)))
        view = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, w, 4))
        return view.copy() if copy else view

    def saveImage(self, fn, ext) -> bool:
        rc = self._lib.vr_save_image(self._h, str(fn).encode(), ext.encode())
        if rc == VR_E_IO:
            return False
        self._check(rc)
        return True

    @property
    def last_kernel_name(self) -> str:
        return (self._lib.vr_last_kernel_name(self._h) or b"").decode()
