"""MI355X-native ray-march path of gallickgunner/Volume-Renderer.

The product is ``lib/libvr_core.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/vr_core.h``).  This package is the thin ctypes harness used by tests and
``bench.py``; it contains no rendering logic and no CPU fallback.

The directory name contains a hyphen, so import it as
``importlib.import_module("volume-renderer_amd")`` or through the root-level alias
module ``volume_renderer_amd``.
"""
from .renderer import (  # noqa: F401
    LIB_PATH,
    RendererCore,
    RendererGroup,
    VRError,
    build_library,
    checksum,
    write_image_rgb8,
    load_library,
    read_pvm_volume,
    symbols_declared_in_header,
)

__all__ = [
    "LIB_PATH",
    "RendererCore",
    "RendererGroup",
    "VRError",
    "build_library",
    "checksum",
    "write_image_rgb8",
    "load_library",
    "read_pvm_volume",
    "symbols_declared_in_header",
]
