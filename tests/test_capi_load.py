"""CPU suite: the C-ABI library builds, loads and exports every symbol include/ecal.h declares."""
import ctypes
import os
import re

import eventcalib_amd
from eventcalib_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ecal.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ecal_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(eventcalib_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(eventcalib_amd.lib_path())
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libecal.so does not export %s" % name
    assert sorted(capi.EXPORTED_SYMBOLS) == declared
    assert eventcalib_amd.load_library().ecal_abi_version() == 3


def test_strerror_and_no_device_is_loud():
    L = eventcalib_amd.load_library()
    assert L.ecal_strerror(0) == b"ok"
    assert L.ecal_strerror(-1) == b"invalid argument"
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        try:
            eventcalib_amd.Context(0)
        except eventcalib_amd.EcalError as e:
            assert e.status == -2
        else:
            raise AssertionError("Context() must fail without a GPU (no CPU fallback)")
