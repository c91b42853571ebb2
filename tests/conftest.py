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


@pytest.fixture(params=["v2", "v2-smallbase", "v1"])
def behz_gen(request, monkeypatch):
    """both generations of the BEHZ conversion kernels (csrc/behz2_kernels.hpp, csrc/behz_kernels.hpp; the library reads TROYN_BEHZ on
    every call) and, for the second generation, both auxiliary bases: the reference's 61-bit base (default) and the base of primes below 2^50
    (TROYN_BEHZ_BASE=small, read by troyn_behz_create; takes effect when every q_i is below 2^50: the multiply's transforms then all run on the
    FP64 butterflies)"""
    monkeypatch.delenv("TROYN_BEHZ", raising=False)
    monkeypatch.delenv("TROYN_BEHZ_BASE", raising=False)
    if request.param == "v1":
        monkeypatch.setenv("TROYN_BEHZ", "v1")
    elif request.param == "v2-smallbase":
        monkeypatch.setenv("TROYN_BEHZ_BASE", "small")
    return request.param
