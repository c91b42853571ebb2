"""Host-side logic of the C++ mirror that needs no GPU (CPU suite): parameter validation and the modulus chain of HeContext::create, replaying the reference's
test/he_context.cu (BFVConstruct, ModulusChainExpansion), and the troy::bench timers (src/utils/timer.h).  The driver links libtroy_amd.so / libtroyn.so, which load without a device; nothing here launches a kernel."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_context_validation_and_chain_host():
    drv = os.path.join(ROOT, "tests", "cpp", "context_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/context_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "host"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert sum(ln.endswith(" ok") for ln in r.stdout.splitlines()) >= 36 and "FAIL" not in r.stdout, r.stdout


def test_reference_test_binary_lists_its_cases():
    """tests/_ref_tests/ref_tests (the reference's own test sources linked against the mirror, tests/build_ref_tests.sh) loads without a device and knows its cases;
    the GPU suite runs them (tests/test_gpu_ref_tests.py)"""
    exe = os.path.join(ROOT, "tests", "_ref_tests", "ref_tests")
    if not os.path.exists(exe):
        pytest.skip("tests/_ref_tests/ref_tests is not built (bash tests/build_ref_tests.sh, needs the reference tree)")
    r = subprocess.run([exe, "--list"], capture_output=True, text=True, timeout=120)
    names = r.stdout.split()
    assert r.returncode == 0 and len(names) >= 360 and sum("Device" in n for n in names) >= 180, r.stdout[-1000:] + r.stderr[-1000:]
    assert "EvaluatorTest.DeviceCKKSRelinearize" in names and "MatmulTest.DeviceBFVMatmul" in " ".join(names) or any(n.startswith("Matmul") for n in names)
