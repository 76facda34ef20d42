"""eventcalib_amd — MI355X-native (gfx950) implementation of the EventCalib hot path.

The product is the C-ABI library ``libecal.so`` (see ``include/ecal.h``); this package is a thin
ctypes binding used by the tests and ``bench.py``.  There is no CPU fallback: importing
``eventcalib_amd.capi`` and calling into it without the built library raises.
"""
from .capi import Context, EcalError, lib_path, load_library  # noqa: F401

__all__ = ["Context", "EcalError", "lib_path", "load_library"]
