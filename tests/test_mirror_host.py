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
