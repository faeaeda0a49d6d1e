"""bibim_renderer_amd -- MI355X-native forward PBR path of chromedays/bibim-renderer.

Python is only the test / bench harness around the C ABI (include/bibim_hip.h); the product is
libbibim_hip.so (HIP kernels for gfx950 + the C++ Scene/Camera/drawFrame shim).
"""
from __future__ import annotations

from .renderer import Renderer, BibimError  # noqa: F401
from . import configs, textures  # noqa: F401

__all__ = ["Renderer", "BibimError", "configs", "textures"]
