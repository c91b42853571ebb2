"""ctypes binding of libtroyn.so (the C-ABI in include/troyn.h).

PyTorch is used only as plumbing: device buffers (int64 storage reinterpreted as uint64),
streams and torch.distributed.  Nothing here computes on the CPU, and there is no fallback:
if the HIP library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TROYN_LIB", os.path.join(_HERE, "libtroyn.so"))   # TROYN_LIB: profiling builds only

u64 = C.c_uint64
u32 = C.c_uint32
sz = C.c_size_t
vp = C.c_void_p
p64 = C.POINTER(C.c_uint64)

# every symbol include/troyn.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "troyn_last_error": (C.c_char_p, []),
    "troyn_version": (C.c_int, []),
    "troyn_kernel_timer_enable": (C.c_int, [C.c_int, C.c_int]),
    "troyn_kernel_timer_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), p64]),
    "troyn_coeff_modulus_create": (C.c_int, [sz, C.POINTER(sz), sz, p64]),
    "troyn_get_primes": (C.c_int, [u64, sz, sz, p64]),
    "troyn_plan_create": (C.c_int, [C.POINTER(vp), C.c_int, u32, u32, p64, p64]),
    "troyn_plan_destroy": (C.c_int, [vp]),
    "troyn_plan_log_n": (u32, [vp]),
    "troyn_plan_n_moduli": (u32, [vp]),
    "troyn_plan_set_option": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
    "troyn_plan_get_root": (C.c_int, [vp, u32, p64]),
    "troyn_plan_get_root_powers": (C.c_int, [vp, u32, C.c_int, p64]),
    "troyn_ntt": (C.c_int, [vp, C.c_int, vp, vp, sz, sz, sz, u32, u32, C.c_int, u32, vp]),
    "troyn_add": (C.c_int, [vp, u32, u32, vp, vp, vp, sz, vp]),
    "troyn_sub": (C.c_int, [vp, u32, u32, vp, vp, vp, sz, vp]),
    "troyn_negate": (C.c_int, [vp, u32, u32, vp, vp, sz, vp]),
    "troyn_multiply_scalar": (C.c_int, [vp, u32, u32, vp, u64, vp, sz, vp]),
    "troyn_dyadic_product": (C.c_int, [vp, u32, u32, vp, vp, vp, sz, vp]),
    "troyn_modulo": (C.c_int, [vp, u32, u32, vp, vp, sz, vp]),
    "troyn_multiply_uint64operand": (C.c_int, [vp, u32, u32, vp, vp, vp, sz, vp]),
    "troyn_dyadic_convolute": (C.c_int, [vp, u32, u32, vp, sz, vp, sz, vp, sz, vp]),
    "troyn_dyadic_square": (C.c_int, [vp, u32, u32, vp, vp, sz, vp]),
    "troyn_switch_key_workspace_bytes": (sz, [vp, u32, sz]),
    "troyn_switch_key": (C.c_int, [vp, u32, C.c_int, C.c_int, vp, C.POINTER(vp), C.c_int, vp, vp, sz, sz, vp]),
    "troyn_relinearize_workspace_bytes": (sz, [vp, u32, sz]),
    "troyn_relinearize": (C.c_int, [vp, u32, C.c_int, C.c_int, vp, C.POINTER(vp), vp, vp, sz, sz, vp]),
    "troyn_ckks_multiply_relinearize_rescale_workspace_bytes": (sz, [vp, u32, sz]),
    "troyn_ckks_multiply_relinearize_rescale": (C.c_int, [vp, u32, vp, vp, C.POINTER(vp), vp, vp, sz, sz, vp]),
    "troyn_divide_and_round_q_last": (C.c_int, [vp, u32, vp, sz, vp, sz, vp]),
    "troyn_divide_and_round_q_last_ntt_workspace_bytes": (sz, [vp, u32, sz, sz]),
    "troyn_divide_and_round_q_last_ntt": (C.c_int, [vp, u32, vp, sz, vp, vp, sz, sz, vp]),
    "troyn_mod_switch_drop": (C.c_int, [vp, u32, u32, vp, sz, vp, sz, vp]),
    "troyn_behz_create": (C.c_int, [C.POINTER(vp), vp, u32, u64]),
    "troyn_behz_destroy": (C.c_int, [vp]),
    "troyn_behz_base_Bsk_size": (u32, [vp]),
    "troyn_behz_working_base_size": (u32, [vp]),
    "troyn_behz_get_base_Bsk": (C.c_int, [vp, p64]),
    "troyn_plain_centralize": (C.c_int, [vp, u32, u64, vp, sz, sz, vp, sz, vp]),
    "troyn_plain_centralize_ntt": (C.c_int, [vp, u32, u64, vp, sz, sz, vp, sz, vp]),
    "troyn_dyadic_broadcast_product": (C.c_int, [vp, u32, u32, vp, sz, vp, sz, vp, sz, vp]),
    "troyn_multiply_plain_accumulate_workspace_bytes": (sz, [sz]),
    "troyn_multiply_plain_accumulate": (C.c_int, [vp, u32, u32, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), sz, C.c_int, vp, sz, vp]),
    "troyn_apply_galois": (C.c_int, [vp, u32, u32, C.c_int, u64, vp, vp, sz, vp]),
    "troyn_apply_galois_plain": (C.c_int, [vp, u64, u64, vp, vp, sz, vp]),
    "troyn_behz_gamma": (u64, [vp]),
    "troyn_bfv_scale_up": (C.c_int, [vp, vp, sz, sz, vp, sz, vp, sz, C.c_int, sz, vp]),
    "troyn_bfv_decrypt_scale_and_round": (C.c_int, [vp, vp, vp, sz, vp]),
    "troyn_prng_block": (C.c_int, [p64, u64, p64]),
    "troyn_sample_ternary": (C.c_int, [vp, u32, p64, u64, vp, p64, vp]),
    "troyn_sample_centered_binomial": (C.c_int, [vp, u32, p64, u64, vp, p64, vp]),
    "troyn_sample_uniform": (C.c_int, [vp, u32, p64, u64, vp, p64, vp]),
    "troyn_sample_centered_binomial_strided": (C.c_int, [vp, u32, p64, u64, u64, vp, sz, vp]),
    "troyn_sample_uniform_multi": (C.c_int, [vp, u32, p64, vp, sz, vp]),
    "troyn_bgv_create": (C.c_int, [C.POINTER(vp), vp, u32, u64]),
    "troyn_bgv_destroy": (C.c_int, [vp]),
    "troyn_bgv_inv_q_last_mod_t": (u64, [vp]),
    "troyn_bgv_mod_switch_workspace_bytes": (sz, [vp, sz, sz]),
    "troyn_bgv_mod_t_and_divide_q_last_ntt": (C.c_int, [vp, vp, sz, vp, vp, sz, sz, vp]),
    "troyn_bgv_decrypt_mod_t": (C.c_int, [vp, vp, u64, vp, sz, vp]),
    "troyn_bgv_multiply_scalar_mod_t": (C.c_int, [vp, vp, u64, vp, sz, vp]),
    "troyn_bgv_switch_key": (C.c_int, [vp, u32, vp, vp, C.c_int, vp, vp, sz, sz, vp]),
    "troyn_bgv_relinearize": (C.c_int, [vp, u32, vp, vp, vp, vp, sz, sz, vp]),
    "troyn_ring2k_create": (C.c_int, [C.POINTER(vp), vp, u32, u32, u32]),
    "troyn_ring2k_destroy": (C.c_int, [vp]),
    "troyn_ring2k_gamma": (u64, [vp]),
    "troyn_ring2k_scale_up": (C.c_int, [vp, vp, sz, vp, vp]),
    "troyn_ring2k_centralize": (C.c_int, [vp, vp, sz, vp, vp]),
    "troyn_ring2k_scale_down": (C.c_int, [vp, vp, vp, vp]),
    "troyn_ring2k_decentralize": (C.c_int, [vp, vp, vp, u64, u64, vp]),
    "troyn_gather_workspace_bytes": (sz, [sz]),
    "troyn_gather": (C.c_int, [vp, sz, sz, vp, vp, sz, vp]),
    "troyn_scatter": (C.c_int, [vp, vp, sz, sz, vp, sz, vp]),
    "troyn_negacyclic_shift": (C.c_int, [vp, u32, u32, vp, vp, sz, sz, vp]),
    "troyn_multiply_inv_degree": (C.c_int, [vp, u32, u32, vp, vp, u64, sz, vp]),
    "troyn_pack_prepare_workspace_bytes": (sz, [sz]),
    "troyn_pack_prepare": (C.c_int, [vp, u32, sz, vp, sz, u64, sz, vp, vp, sz, vp]),
    "troyn_pack_layer": (C.c_int, [vp, u32, u64, sz, vp, vp, vp, sz, vp]),
    "troyn_extract_lwe_workspace_bytes": (sz, [sz]),
    "troyn_extract_lwe": (C.c_int, [vp, u32, vp, vp, vp, vp, sz, vp, sz, vp]),
    "troyn_bfv_multiply_workspace_bytes": (sz, [vp, sz, sz, sz]),
    "troyn_bfv_multiply": (C.c_int, [vp, vp, sz, vp, sz, vp, vp, sz, sz, vp]),
}

_lib = None


class TroynError(RuntimeError):
    """Runtime (HIP) failure -- the reference throws std::runtime_error (kernel_provider.h:11-16)."""


class TroynInvalidArgument(ValueError):
    """API misuse -- the reference throws std::invalid_argument("[Class::method] ...")."""


def lib():
    """Load libtroyn.so; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TroynError(
            "libtroyn.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(L, name)   # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc == 0:
        return
    msg = lib().troyn_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise TroynInvalidArgument("%s (troyn code %d)" % (msg, rc))
    raise TroynError("%s (hipError %d)" % (msg, rc))


def coeff_modulus_create(poly_modulus_degree, bit_sizes):
    """CoeffModulus::create (coeff_modulus.cu:65-108) -- host only"""
    n = len(bit_sizes)
    bs = (sz * n)(*bit_sizes)
    out = (u64 * n)()
    check(lib().troyn_coeff_modulus_create(poly_modulus_degree, bs, n, out))
    return [int(x) for x in out]


def get_primes(factor, bit_size, count):
    """utils::get_primes (utils/number_theory.cu:22-39) -- host only"""
    out = (u64 * count)()
    check(lib().troyn_get_primes(factor, bit_size, count, out))
    return [int(x) for x in out]
