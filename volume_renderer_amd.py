"""Alias so the hyphenated package directory ``volume-renderer_amd/`` can be imported
with a normal identifier: ``import volume_renderer_amd as vra``."""
import importlib
import sys

_pkg = importlib.import_module("volume-renderer_amd")
sys.modules[__name__] = _pkg
