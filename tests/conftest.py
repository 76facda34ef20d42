import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _ref_kdtree_present():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libkdtree_ref.so"))


def pytest_report_header(config):
    """The log says which k-d tree sits under the oracle's DBSCAN when it checks the HIP kernels (-m gpu tests): the
    reference's own kdtree.cpp compiled by oracle/Makefile, or — only when that build is absent — the restated tree."""
    if _ref_kdtree_present():
        return "oracle k-d tree for the -m gpu parity checks: the REFERENCE's kdtree.cpp (oracle/_ref/libkdtree_ref.so)"
    return "oracle k-d tree for the -m gpu parity checks: RESTATED tree only (oracle/_ref/libkdtree_ref.so is absent)"


@pytest.fixture(autouse=True)
def _oracle_kd_backend(request):
    """-m gpu tests: every oracle call that clusters (dbscan, extract_candidates, the window loops) runs on the
    reference's compiled k-d tree when oracle/_ref is present (the library then stays mapped for the whole session).
    CPU tests: the restated tree, because they are the ones that compare the two trees with each other."""
    if "oracle_lib" not in sys.modules and request.node.get_closest_marker("gpu") is None:
        yield
        return
    import oracle_lib as O
    O.set_kd_backend(request.node.get_closest_marker("gpu") is not None and O.have_ref_kdtree())
    yield


@pytest.fixture(autouse=True)
def _resync_library_switches():
    """The library reads its ECAL_* debug switches once per context; a test that changed one (monkeypatch restores the
    environment at teardown, before this fixture's) leaves no live context on a stale value."""
    yield
    mod = sys.modules.get("eventcalib_amd.capi")
    if mod is not None and hasattr(mod, "sync_env"):
        try:
            mod.sync_env()
        except Exception:
            pass
