"""troy-nova_amd: MI355X-native hot path (NTT/INTT, RNS dyadic ops, key switching, modulus
switching, BEHZ multiply) behind troy-nova's evaluator interface.

The directory name contains a hyphen (it is fixed by the project layout), so the package is
loaded under the importable name ``troy_nova_amd`` by ``__graft_entry__.load_package()``.
"""
from . import capi  # noqa: F401
from .engine import (ASSIGN_ADD_INPLACE, ASSIGN_OVERWRITE, ASSIGN_OVERWRITE_EXCEPT_FIRST,  # noqa: F401
                     IDX_COMPONENTWISE, IDX_KS_SET_PRODUCTS, IDX_KS_SKIP_FINALS, Behz, Bgv, Plan, Prng, Ring2k, sample_uniform_multi, to_device, to_host)
