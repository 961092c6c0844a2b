"""MI355X-native (gfx950, hand-written HIP) AMG V/W-cycle + ASAt KKT assembly.

Drop-in for the hot path of zihang-student/Codes-of-IPD-SsN-AMG-method behind
the reference's own MATLAB function signatures; see DESIGN.md / INTEGRATION.md.
Importing this package loads ``libipdamg.so`` and fails loudly if it is missing
(there is no CPU fallback).
"""
from .api import *  # noqa: F401,F403
from .api import __all__ as _api_all
from ._lib import LIB_PATH, get_ctx  # noqa: F401

__all__ = list(_api_all) + ["LIB_PATH", "get_ctx"]
