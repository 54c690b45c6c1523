"""TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY -- ctypes binding of oracle/_ref/libref_gl.so
(ref_gl_runner.c): the reference's unmodified shader executed by Mesa llvmpipe.

Never imported by the product, by bench.py or by anything that runs on the GPU box; the
frames it produces travel as tests/golden/ref_gl/*.npz (see mint_ref_gl_goldens.py).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR.parent / "_ref" / "libref_gl.so"
SHADER_PATH = Path("/root/reference/VolumeRenderer.cs")   # read where it lies, never copied
DRIVER_PATH = Path("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")
GL_NEAREST, GL_LINEAR = 0x2600, 0x2601
_lib = None


class _Job(C.Structure):
    _fields_ = [
        ("shader_path", C.c_char_p), ("volume", C.c_void_p),
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("bytes_per_voxel", C.c_int32),
        ("fb_w", C.c_int32), ("fb_h", C.c_int32), ("win_w", C.c_int32), ("win_h", C.c_int32),
        ("cam", C.c_float * 21), ("alpha_scale", C.c_float), ("voxel_size", C.c_float * 3),
        ("min_val", C.c_int32), ("max_val", C.c_int32),
        ("is_mip", C.c_int32), ("view_top", C.c_int32), ("view_bottom", C.c_int32),
        ("tex_filter", C.c_int32), ("clear_value", C.c_float), ("tex_float", C.c_int32),
    ]


def available() -> bool:
    return SHADER_PATH.exists() and DRIVER_PATH.exists()


def build() -> Path:
    proc = subprocess.run(["make", "-C", str(_DIR.parent), "ref_gl"], capture_output=True, text=True)
    if proc.returncode != 0 or not LIB_PATH.exists():
        raise RuntimeError("building libref_gl.so failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        lib = C.CDLL(str(LIB_PATH))
        lib.refgl_last_error.restype = C.c_char_p
        lib.refgl_init.argtypes = [C.c_char_p, C.c_int]
        lib.refgl_render.argtypes = [C.POINTER(_Job), C.POINTER(C.c_float)]
        lib.refgl_compute.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        buf = C.create_string_buffer(512)
        if lib.refgl_init(buf, 512) != 0:
            raise RuntimeError("llvmpipe context: " + lib.refgl_last_error().decode())
        lib.gl_info = buf.value.decode()
        _lib = lib
    return _lib


def gl_info() -> str:
    return load().gl_info


def render(volume: np.ndarray, img, cam, *, alpha_scale=1.0, voxel_size=(1.0, 1.0, 1.0), min_val=0, max_val=255,
           is_mip=0, view_top=0, view_bottom=0, tex_filter=GL_NEAREST, window_size=None, clear_value=0.0,
           shader_path=SHADER_PATH) -> np.ndarray:
    """One frame of the reference shader.  volume[z, y, x] u8/u16; img = framebuffer (W, H);
    min_val/max_val are the uniforms AS UPLOADED (the +1000 of RendererCore.cpp:66-67 included).
    Returns rgba[H, W, 4] float32, row 0 = bottom."""
    lib = load()
    v = np.ascontiguousarray(volume)
    assert v.ndim == 3 and v.dtype in (np.uint8, np.uint16, np.float32)
    nz, ny, nx = v.shape
    W, H = img
    j = _Job()
    j.shader_path = str(shader_path).encode()
    j.volume = v.ctypes.data
    j.nx, j.ny, j.nz, j.bytes_per_voxel = nx, ny, nz, v.dtype.itemsize
    j.tex_float = 1 if v.dtype == np.float32 else 0     # (the TRILINEAR cross-check: R32F + a `sampler3D` variant of the shader)
    j.fb_w, j.fb_h = W, H
    j.win_w, j.win_h = window_size if window_size is not None else (W, H)
    cam = np.ascontiguousarray(cam, dtype=np.float32)
    for i in range(21):
        j.cam[i] = float(cam[i])
    j.alpha_scale = alpha_scale
    for i in range(3):
        j.voxel_size[i] = voxel_size[i]
    j.min_val, j.max_val = int(min_val), int(max_val)
    j.is_mip, j.view_top, j.view_bottom = int(is_mip), int(view_top), int(view_bottom)
    j.tex_filter = tex_filter
    j.clear_value = clear_value
    out = np.zeros((H, W, 4), dtype=np.float32)
    if lib.refgl_render(C.byref(j), out.ctypes.data_as(C.POINTER(C.c_float))) != 0:
        raise RuntimeError("refgl_render: " + lib.refgl_last_error().decode())
    return out
