#!/usr/bin/env python3
"""Mint PVM golden fixtures with the REFERENCE's own codec.

Runs only in the authoring container: needs oracle/_ref/libddsref.so, which
oracle/Makefile compiles from /root/reference/src/ddsbase.cpp where it lies.  The
outputs (small .pvm files + manifest.json with dims / sha256 / the reference's
checksum() of the decoded payload) are committed; the GPU box never sees the reference.
"""
import ctypes as C
import hashlib
import json
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = HERE.parent.parent / "oracle" / "_ref" / "libddsref.so"
lib = C.CDLL(str(REF))
lib.ddsref_write_pvm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_float,
                                 C.c_float, C.c_char_p]
lib.ddsref_read_pvm.restype = C.c_void_p
lib.ddsref_read_pvm.argtypes = [C.c_char_p] + [C.POINTER(C.c_uint)] * 4 + [C.POINTER(C.c_float)] * 3
lib.ddsref_checksum.restype = C.c_uint
lib.ddsref_checksum.argtypes = [C.c_void_p, C.c_uint]
lib.ddsref_free.argtypes = [C.c_void_p]


def ref_read(path):
    w, h, d, c = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
    sx, sy, sz = C.c_float(), C.c_float(), C.c_float()
    p = lib.ddsref_read_pvm(str(path).encode(), w, h, d, c, sx, sy, sz)
    assert p, path
    n = w.value * h.value * d.value * c.value
    data = bytes((C.c_ubyte * n).from_address(p))
    lib.ddsref_free(p)
    return data, (w.value, h.value, d.value, c.value), (sx.value, sy.value, sz.value)


def main():
    rng = np.random.default_rng(2024)
    cases = []

    def add(name, vol_bytes, w, h, d, comps, scale=(1.0, 1.0, 1.0), desc=None, keep_payload=True):
        path = HERE / f"{name}.pvm"
        buf = np.frombuffer(vol_bytes, dtype=np.uint8).copy()
        lib.ddsref_write_pvm(str(path).encode(), buf.ctypes.data, w, h, d, comps, scale[0], scale[1], scale[2],
                             desc.encode() if desc else None)
        data, dims, sc = ref_read(path)          # decode with the reference itself
        assert data == bytes(vol_bytes) and dims == (w, h, d, comps)
        entry = {
            "file": path.name, "width": w, "height": h, "depth": d, "components": comps,
            "scale": list(sc), "payload_bytes": len(data),
            "sha256": hashlib.sha256(data).hexdigest(),
            "ref_checksum": int(lib.ddsref_checksum(np.frombuffer(data, dtype=np.uint8).ctypes.data, len(data))),
            "file_bytes": path.stat().st_size,
        }
        if keep_payload:
            np.save(HERE / f"{name}.npy", np.frombuffer(data, dtype=np.uint8))
            entry["payload"] = f"{name}.npy"
        cases.append(entry)

    # PVM (v1 header), 1 component, odd sizes, noisy data (all bit widths exercised)
    v = rng.integers(0, 256, size=9 * 7 * 5, dtype=np.uint8)
    add("pvm1_u8_9x7x5_noise", v.tobytes(), 9, 7, 5, 1)
    # smooth ramp: small deltas, long runs, strip predictor active
    z, y, x = np.meshgrid(np.arange(6), np.arange(11), np.arange(13), indexing="ij")
    v = ((x * 3 + y * 5 + z * 7) % 256).astype(np.uint8)
    add("pvm1_u8_13x11x6_ramp", v.tobytes(), 13, 11, 6, 1)
    # PVM2: anisotropic scale, 2 components (16-bit, byte-interleave skip=2)
    v16 = (rng.integers(0, 4096, size=8 * 6 * 4)).astype(">u2")          # big-endian payload (V^3 convention, Q9)
    add("pvm2_u16_8x6x4_scaled", v16.tobytes(), 8, 6, 4, 2, scale=(1.0, 0.5, 2.0))
    # PVM3: description string block
    v = rng.integers(0, 50, size=10 * 10 * 3, dtype=np.uint8)
    add("pvm3_u8_10x10x3_desc", v.tobytes(), 10, 10, 3, 1, scale=(1.0, 1.0, 1.5), desc="golden fixture")
    # all zeros / all 255 (zero-bit runs)
    add("pvm1_u8_16x16x4_zero", bytes(16 * 16 * 4), 16, 16, 4, 1)
    # > 2^24 bytes: "DDS v3e" with the 2^24 block interleave (smooth so it stays small on disk)
    w, h, d = 256, 256, 130
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    v16 = (zz * 31 + yy // 32).astype(">u2")
    add("pvm1_u16_256x256x130_v3e", v16.tobytes(), w, h, d, 2, keep_payload=False)
    (HERE / "pvm_manifest.json").write_text(json.dumps(cases, indent=1))
    for c in cases:
        print(c["file"], c["file_bytes"], "bytes;", c["payload_bytes"], "payload")


if __name__ == "__main__":
    main()
