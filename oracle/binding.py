"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liboracle.so (vr_oracle.c).

Never imported by the product package.  See oracle/vr_oracle.h for the contract.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB_PATH = _DIR / "liboracle.so"
_lib = None


def build_oracle() -> Path:
    proc = subprocess.run(["make", "-C", str(_DIR)], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + proc.stdout + proc.stderr)
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [
        ("img_w", C.c_int32), ("img_h", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("trunc_grid", C.c_int32),
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
        ("bytes_per_voxel", C.c_int32),
        ("volume", C.c_void_p),
        ("cam", C.c_float * 21),
        ("alpha_scale", C.c_float),
        ("voxel_size", C.c_float * 3),
        ("min_val", C.c_int32), ("max_val", C.c_int32),
        ("is_mip", C.c_int32), ("view_top", C.c_int32), ("view_bottom", C.c_int32),
        ("filter", C.c_int32), ("accum", C.c_int32), ("max_steps", C.c_int32),
        ("tf_rgba", C.POINTER(C.c_float)), ("tf_len", C.c_int32),
        ("threads", C.c_int32),
        ("arith", C.c_int32),
        ("lerp_unfused", C.c_int32),
    ]


class _Camera(C.Structure):
    _fields_ = [
        ("eye", C.c_float * 4), ("side", C.c_float * 4), ("up", C.c_float * 4), ("look_at", C.c_float * 4),
        ("view2world", C.c_float * 16),
        ("view_plane_dist", C.c_float), ("y_fov", C.c_float), ("rotation_speed", C.c_float), ("mov_speed", C.c_float),
        ("zenith", C.c_float), ("azimuth", C.c_float), ("radius", C.c_float),
        ("is_changed", C.c_int32),
    ]


def _load():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build_oracle()
        lib = C.CDLL(str(_LIB_PATH))
        lib.vro_render.restype = C.c_int
        lib.vro_render.argtypes = [C.POINTER(_Params), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        lib.vro_camera_init.argtypes = [C.POINTER(_Camera), C.c_float, C.c_float, C.c_float]
        lib.vro_camera_reset.argtypes = [C.POINTER(_Camera)]
        lib.vro_camera_orient.argtypes = [C.POINTER(_Camera), C.c_float, C.c_float, C.c_float]
        lib.vro_camera_block.argtypes = [C.POINTER(_Camera), C.POINTER(C.c_float)]
        lib.vro_gen_sphere_u8.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.vro_gen_noise_ball.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
        lib.vro_spline_tf.restype = C.c_int
        lib.vro_spline_tf.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_float, C.POINTER(C.c_float), C.c_int32]
        _lib = lib
    return _lib


class Camera:
    """oracle restatement of src/Camera.cpp"""

    def __init__(self, y_fov=30.0, rot_speed=0.7, mov_speed=0.3):
        self._c = _Camera()
        _load().vro_camera_init(C.byref(self._c), y_fov, rot_speed, mov_speed)

    def reset(self):
        _load().vro_camera_reset(C.byref(self._c))

    def orient(self, zoom, zenith, azimuth):
        _load().vro_camera_orient(C.byref(self._c), zoom, zenith, azimuth)

    def block(self) -> np.ndarray:
        out = np.zeros(21, dtype=np.float32)
        _load().vro_camera_block(C.byref(self._c), out.ctypes.data_as(C.POINTER(C.c_float)))
        return out


def default_camera_block() -> np.ndarray:
    return Camera().block()


@dataclass
class OracleParams:
    img_w: int
    img_h: int
    cam: np.ndarray = field(default_factory=default_camera_block)
    alpha_scale: float = 1.0
    voxel_size: tuple = (1.0, 1.0, 1.0)
    min_val: int = 0
    max_val: int = 255
    is_mip: int = 0
    view_top: int = 0
    view_bottom: int = 0
    filter: int = 0
    accum: int = 0
    max_steps: int = 10000
    trunc_grid: int = 0
    row_begin: int = 0
    row_end: int = -1
    tf_rgba: np.ndarray | None = None
    threads: int = 1
    arith: int = 0          # 0 = MESA (the executed reference: the contract), 1 = SPEC (specification text)
    lerp_unfused: int = 0   # TRILINEAR: 1 = three roundings per lerp (rounds 1-4; measurement only), 0 = one fma (the definition)


def render(volume: np.ndarray, p: OracleParams, want_spp: bool = False, out: np.ndarray | None = None):
    """volume indexed [z, y, x]; returns (rgba[h, w, 4] float32, total_samples[, spp])."""
    lib = _load()
    v = np.ascontiguousarray(volume)
    assert v.ndim == 3 and v.dtype in (np.uint8, np.uint16)
    nz, ny, nx = v.shape
    q = _Params()
    q.img_w, q.img_h = p.img_w, p.img_h
    q.row_begin = p.row_begin
    q.row_end = p.img_h if p.row_end < 0 else p.row_end
    q.trunc_grid = p.trunc_grid
    q.nx, q.ny, q.nz = nx, ny, nz
    q.bytes_per_voxel = v.dtype.itemsize
    q.volume = v.ctypes.data
    cam = np.ascontiguousarray(p.cam, dtype=np.float32)
    for i in range(21):
        q.cam[i] = float(cam[i])
    q.alpha_scale = p.alpha_scale
    for i in range(3):
        q.voxel_size[i] = p.voxel_size[i]
    q.min_val, q.max_val = p.min_val, p.max_val
    q.is_mip, q.view_top, q.view_bottom = p.is_mip, p.view_top, p.view_bottom
    q.filter, q.accum, q.max_steps = p.filter, p.accum, p.max_steps
    tf = None
    if p.tf_rgba is not None:
        tf = np.ascontiguousarray(p.tf_rgba, dtype=np.float32).reshape(-1, 4)
        q.tf_rgba = tf.ctypes.data_as(C.POINTER(C.c_float))
        q.tf_len = tf.shape[0]
    q.threads = p.threads
    q.arith = p.arith
    q.lerp_unfused = p.lerp_unfused
    rgba = out if out is not None else np.zeros((p.img_h, p.img_w, 4), dtype=np.float32)
    spp = np.zeros((p.img_h, p.img_w), dtype=np.uint32) if want_spp else None
    total = C.c_uint64()
    rc = lib.vro_render(C.byref(q), rgba.ctypes.data_as(C.POINTER(C.c_float)),
                        spp.ctypes.data_as(C.POINTER(C.c_uint32)) if want_spp else None, C.byref(total))
    if rc != 0:
        raise RuntimeError(f"vro_render failed: {rc}")
    return (rgba, total.value, spp) if want_spp else (rgba, total.value)


def gen_sphere_u8(n: int, radius: int) -> np.ndarray:
    out = np.zeros((n, n, n), dtype=np.uint8)
    _load().vro_gen_sphere_u8(out.ctypes.data, n, radius)
    return out


def gen_noise_ball(dims, bytes_per_voxel: int, seed: int) -> np.ndarray:
    nx, ny, nz = dims
    out = np.zeros((nz, ny, nx), dtype=np.uint8 if bytes_per_voxel == 1 else np.uint16)
    _load().vro_gen_noise_ball(out.ctypes.data, nx, ny, nz, bytes_per_voxel, seed & 0xFFFFFFFF)
    return out


def spline_tf(iso, rgba) -> np.ndarray:
    iso = np.asarray(iso, dtype=np.float32).reshape(-1, 1)
    rgba = np.asarray(rgba, dtype=np.float32).reshape(-1, 4)
    knots = np.ascontiguousarray(np.concatenate([iso, rgba], axis=1), dtype=np.float32)
    lut = np.zeros((256, 4), dtype=np.float32)
    rc = _load().vro_spline_tf(knots.ctypes.data_as(C.POINTER(C.c_float)), knots.shape[0], 255.0,
                               lut.ctypes.data_as(C.POINTER(C.c_float)), 256)
    if rc != 0:
        raise RuntimeError(f"vro_spline_tf failed: {rc}")
    return lut
