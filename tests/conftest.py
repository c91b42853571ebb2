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


@pytest.fixture(params=["v2", "v2-refbase", "v1", "v2-small"])
def behz_gen(request, monkeypatch):
    """both generations of the BEHZ conversion kernels (csrc/behz2_kernels.hpp, csrc/behz_kernels.hpp; TROYN_BEHZ, read when the plan is created) and, for the
    second generation, both auxiliary bases: primes below 2^50 (the default since round 5 when every q_i is below 2^50: the multiply's transforms then all run on
    the FP64 butterflies) and the reference's 61-bit base (TROYN_BEHZ_BASE=ref, read by troyn_behz_create).  The tests multiply a few ciphertexts, and at
    the whole-limb sizes such a launch takes separate transform / dyadic launches by default (troyn_bfv_multiply); the first three parameters force
    tensor_core_kernel (TROYN_BFV_TENSOR=fused), "v2-small" is the library's default."""
    for k in ("TROYN_BEHZ", "TROYN_BEHZ_BASE", "TROYN_BFV_TENSOR"):
        monkeypatch.delenv(k, raising=False)
    if request.param != "v2-small":
        monkeypatch.setenv("TROYN_BFV_TENSOR", "fused")
    if request.param == "v1":
        monkeypatch.setenv("TROYN_BEHZ", "v1")
    elif request.param == "v2-refbase":
        monkeypatch.setenv("TROYN_BEHZ_BASE", "ref")
    return request.param

# Key-switch tests on a few ciphertexts: below 128 (N <= 4096) / 256 (chains with moduli >= 2^50, N >= 8192) workgroups of the one-launch inner product the
# library takes the two-launch form by itself (csrc/troyn.hip ks_small_mixed).  The modules below were written against the one-launch kernels, so each of
# their tests runs twice: TROYN_KS_MAC=fused (read when the plan is created) and the library's default.
KS_FORM_MODULES = {"test_gpu_keyswitch", "test_gpu_keyswitch_spec", "test_gpu_corners", "test_gpu_bgv"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in KS_FORM_MODULES and "ks_form" in metafunc.fixturenames:
        metafunc.parametrize("ks_form", ["one-launch", "default"], indirect=True)


@pytest.fixture(autouse=True)
def ks_form(request, monkeypatch):
    form = getattr(request, "param", None)
    if form == "one-launch":
        monkeypatch.setenv("TROYN_KS_MAC", "fused")
    elif form == "default":
        monkeypatch.delenv("TROYN_KS_MAC", raising=False)
    return form
