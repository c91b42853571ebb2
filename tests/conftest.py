import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (checker)"""
    return entry.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    """the product package; on a GPU box it must load the in-tree HIP library"""
    return entry.load_package()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


@pytest.fixture(params=["v2", "v1"])
def behz_gen(request, monkeypatch):
    """both generations of the BEHZ conversion kernels (csrc/behz2_kernels.hpp, csrc/behz_kernels.hpp): the library reads
    TROYN_BEHZ on every call"""
    if request.param == "v1":
        monkeypatch.setenv("TROYN_BEHZ", "v1")
    else:
        monkeypatch.delenv("TROYN_BEHZ", raising=False)
    return request.param
