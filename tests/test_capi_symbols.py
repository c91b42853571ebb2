"""The C-ABI library loads and exports every symbol include/troyn.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "troyn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(troyn_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    declared = _declared_symbols()
    assert len(declared) >= 30
    assert sorted(pkg.capi.SYMBOLS.keys()) == declared


def test_library_exports_every_symbol(pkg):
    lib = pkg.capi.lib()          # raises if libtroyn.so is missing: there is no fallback
    for name in _declared_symbols():
        assert hasattr(lib, name), "libtroyn.so does not export %s" % name
    assert lib.troyn_version() == 1


def test_host_only_helpers_match_oracle(pkg, O):
    # CoeffModulus::create / get_primes are host functions of the product; cross-check with the oracle
    for n, bits in ((8192, [40, 40, 40]), (16384, [50] * 6), (32768, [50] * 11), (8192, [60, 40, 40, 60]), (32, [30, 30, 30, 30])):
        assert pkg.capi.coeff_modulus_create(n, bits) == O.coeff_modulus_create(n, bits)
    assert pkg.capi.get_primes(2 * 32768, 61, 13) == O.get_primes(2 * 32768, 61, 13)
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        pkg.capi.coeff_modulus_create(8192, [61])
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        pkg.capi.coeff_modulus_create(1, [30])


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg.capi, "_lib", None)
    monkeypatch.setattr(pkg.capi, "LIB_PATH", os.path.join(ROOT, "does-not-exist", "libtroyn.so"))
    with pytest.raises(pkg.capi.TroynError):
        pkg.capi.lib()


def test_no_cpu_path(pkg):
    with pytest.raises(pkg.capi.TroynError):
        pkg.Plan("cpu", 10, [1099510824961])


def test_product_never_imports_oracle():
    # the product path must not route through oracle/ (tests, smoke and bench's cpu_baseline leg only)
    pkg_dir = os.path.join(ROOT, "troy-nova_amd")
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "troy_oracle" not in text and "import oracle" not in text and "from oracle" not in text, os.path.join(base, f)


def test_switches_are_read_once_at_plan_creation():
    """the library's A/B switches come from the environment in exactly one place (options_from_environment, called by troyn_plan_create): no
    getenv is reachable from any other troyn_* entry (tests/test_gpu_switches.py checks the behaviour on the GPU)"""
    csrc = os.path.join(ROOT, "troy-nova_amd", "csrc")
    hits = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".hpp", ".inl")):
            continue
        for no, line in enumerate(open(os.path.join(csrc, name)), 1):
            code = line.split("//")[0]
            if re.search(r"\bgetenv\s*\(", code):
                hits.append((name, no, code.strip()))
    assert len(hits) == 1 and hits[0][0] == "troyn.hip", hits
    text = open(os.path.join(csrc, "troyn.hip")).read()
    body = text[text.index("static int options_from_environment(TroynOptions& o)"):]
    body = body[:body.index("\n}\n")]
    assert hits[0][2] in body and "option_apply" in body
    assert text.count("options_from_environment(") == 2      # the definition and troyn_plan_create
