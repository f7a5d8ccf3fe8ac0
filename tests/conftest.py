import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-compression_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _cap_oracle_threads():
    """The CPU oracle's convolutions stop scaling around 8 threads and get SLOWER with all cores of a 128-core GPU host
    (bench.py measures 0.08 frames/s at 8 threads against 0.045 at 128): cap the intra-op pool for the whole session."""
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))


def pytest_sessionstart(session):
    """The shared libraries are build products (git-ignored).  A fresh checkout that runs the tests before
    ``__graft_entry__.build()`` gets them built here once (hipcc cross-compiles without a GPU, ~3 minutes)."""
    _cap_oracle_threads()
    needed = [os.path.join(ROOT, "video-compression_amd", "libvc_hip.so"), os.path.join(ROOT, "oracle", "librans_oracle.so")]
    if all(os.path.exists(p) for p in needed) or os.environ.get("VC_HIP_LIB"):
        return
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
