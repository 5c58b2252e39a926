import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared libraries are build artefacts (git-ignored): build them once if a fresh checkout lacks them
    need = [os.path.join(PKG, "csrc", "libpnode_hip.so"), os.path.join(ROOT, "oracle", "libpn_oracle_c.so")]
    if not all(os.path.exists(p) for p in need):
        sys.path.insert(0, ROOT)
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    return {name: np.load(os.path.join(d, f"{name}.npz")) for name in ("vdp", "brusselator", "rigid_body", "pleiades", "three_body")}


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; GPU tests must run the HIP path, so a missing build is an error, not a skip."""
    from probdiffeq_hip import _backend

    return _backend.load_library()
