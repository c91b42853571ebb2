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


@pytest.fixture(params=["v2", "v2-refbase", "v1"])
def behz_gen(request, monkeypatch):
    """both generations of the BEHZ conversion kernels (csrc/behz2_kernels.hpp, csrc/behz_kernels.hpp; TROYN_BEHZ, read when the plan is created) and, for the
    second generation, both auxiliary bases: primes below 2^50 (the default since round 5 when every q_i is below 2^50: the multiply's transforms then all run on
    the FP64 butterflies) and the reference's 61-bit base (TROYN_BEHZ_BASE=ref, read by troyn_behz_create)"""
    monkeypatch.delenv("TROYN_BEHZ", raising=False)
    monkeypatch.delenv("TROYN_BEHZ_BASE", raising=False)
    if request.param == "v1":
        monkeypatch.setenv("TROYN_BEHZ", "v1")
    elif request.param == "v2-refbase":
        monkeypatch.setenv("TROYN_BEHZ_BASE", "ref")
    return request.param
