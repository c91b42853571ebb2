"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (troy-nova_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtroy_oracle.so")

u64 = C.c_uint64
sz = C.c_size_t
p64 = C.POINTER(C.c_uint64)
vp = C.c_void_p


class Modulus(C.Structure):
    _fields_ = [("value", u64), ("const_ratio", u64 * 3), ("bit_count", u64),
                ("is_prime", C.c_int32), ("pad_", C.c_int32)]


class MulOp(C.Structure):
    _fields_ = [("operand", u64), ("quotient", u64)]


def build(force=False):
    src = os.path.join(_HERE, "troy_oracle.c")
    hdr = os.path.join(_HERE, "troy_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    MP = C.POINTER(Modulus)
    OP = C.POINTER(MulOp)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_modulus_init", C.c_int, MP, u64)
    sig("orc_barrett_reduce64", u64, u64, MP)
    sig("orc_barrett_reduce128", u64, u64, u64, MP)
    sig("orc_multiply_mod", u64, u64, u64, MP)
    sig("orc_add_mod", u64, u64, u64, MP)
    sig("orc_sub_mod", u64, u64, u64, MP)
    sig("orc_negate_mod", u64, u64, MP)
    sig("orc_mulop_init", None, OP, u64, MP)
    sig("orc_mulop_mod", u64, u64, OP, MP)
    sig("orc_mulop_mod_lazy", u64, u64, OP, MP)
    sig("orc_exponentiate_mod", u64, u64, u64, MP)
    sig("orc_try_invert_mod", C.c_int, u64, MP, p64)
    sig("orc_dot_product_mod", u64, p64, p64, sz, MP)
    sig("orc_is_prime", C.c_int, u64)
    sig("orc_get_primes", C.c_int, u64, sz, sz, p64)
    sig("orc_coeff_modulus_create", C.c_int, sz, C.POINTER(sz), sz, p64)
    sig("orc_try_minimal_primitive_root", C.c_int, u64, MP, p64)
    sig("orc_ntt_tables_create", vp, sz, u64)
    sig("orc_ntt_tables_destroy", None, vp)
    sig("orc_ntt_tables_root", u64, vp)
    sig("orc_ntt_tables_root_power", u64, vp, sz, C.c_int, C.c_int)
    sig("orc_ntt_tables_inv_degree", u64, vp, C.c_int)
    sig("orc_ntt_forward", None, p64, sz, sz, sz, C.POINTER(vp), sz, C.c_int, sz)
    sig("orc_ntt_inverse", None, p64, sz, sz, sz, C.POINTER(vp), sz, C.c_int, sz)
    for nm in ("orc_add_ps", "orc_sub_ps", "orc_dyadic_product_ps"):
        sig(nm, None, p64, p64, sz, sz, MP, sz, p64)
    for nm in ("orc_negate_ps", "orc_modulo_ps"):
        sig(nm, None, p64, sz, sz, MP, sz, p64)
    sig("orc_multiply_scalar_ps", None, p64, u64, sz, sz, MP, sz, p64)
    sig("orc_multiply_uint64operand_ps", None, p64, OP, sz, sz, MP, sz, p64)
    sig("orc_dyadic_convolute", None, p64, p64, sz, sz, MP, sz, sz, p64)
    sig("orc_dyadic_square", None, p64, MP, sz, sz, p64)
    sig("orc_rns_tool_create", vp, sz, p64, sz, u64)
    sig("orc_rns_tool_destroy", None, vp)
    sig("orc_rns_tool_base_B_size", sz, vp)
    sig("orc_rns_tool_base_Bsk_size", sz, vp)
    sig("orc_rns_tool_m_sk", u64, vp)
    sig("orc_rns_tool_gamma", u64, vp)
    sig("orc_rns_tool_m_tilde", u64, vp)
    sig("orc_rns_tool_base_Bsk", None, vp, p64)
    sig("orc_rns_tool_inv_q_last_mod_q", u64, vp, sz, C.c_int)
    sig("orc_fast_convert_array", None, p64, sz, p64, sz, p64, sz, p64)
    for nm in ("orc_rns_fast_b_conv_m_tilde", "orc_rns_sm_mrq", "orc_rns_fast_floor", "orc_rns_fast_b_conv_sk",
               "orc_rns_fast_b_conv_m_tilde_sm_mrq"):
        sig(nm, None, vp, p64, p64)
    sig("orc_rns_fast_floor_fast_b_conv_sk", None, vp, p64, p64, sz, p64)
    sig("orc_rns_divide_and_round_q_last", None, vp, p64, sz, p64)
    sig("orc_rns_divide_and_round_q_last_ntt", None, vp, p64, sz, p64, C.POINTER(vp))
    sig("orc_context_create", vp, C.c_int, sz, p64, sz, u64)
    sig("orc_context_destroy", None, vp)
    sig("orc_context_key_modulus_size", sz, vp)
    sig("orc_context_ntt_table", vp, vp, sz)
    sig("orc_context_rns_tool", vp, vp, sz)
    sig("orc_context_moduli", MP, vp)
    sig("orc_transform_to_ntt", None, vp, p64, sz, sz)
    sig("orc_transform_from_ntt", None, vp, p64, sz, sz)
    sig("orc_switch_key", None, vp, sz, C.c_int, p64, C.POINTER(p64), C.c_int, p64)
    sig("orc_relinearize", None, vp, sz, C.c_int, p64, C.POINTER(p64), p64)
    sig("orc_ckks_multiply", None, vp, sz, p64, sz, p64, sz, p64)
    sig("orc_bfv_multiply", None, vp, sz, p64, sz, p64, sz, p64)
    sig("orc_mod_switch_scale_to_next", None, vp, sz, p64, sz, p64)
    sig("orc_mod_switch_drop_to_next", None, vp, sz, p64, sz, p64)
    sig("orc_fill_uniform", None, u64, u64, p64, sz)
    sig("orc_fnv1a64", u64, p64, sz)
    sig("orc_aes128_encrypt_block", None, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8))
    sig("orc_rng_create", vp, u64, u64)
    sig("orc_rng_destroy", None, vp)
    sig("orc_rng_sample_uint64", u64, vp)
    sig("orc_rng_fill_uint64s", None, vp, p64, sz)
    for nm in ("orc_sample_poly_ternary", "orc_sample_poly_centered_binomial", "orc_sample_poly_uniform"):
        sig(nm, None, vp, p64, sz, MP, sz)
    sig("orc_keygen_secret_key", None, vp, vp, p64)
    sig("orc_keygen_public_key", None, vp, vp, p64, p64)
    sig("orc_batch_encode", C.c_int, vp, p64, sz, p64)
    sig("orc_encrypt_asymmetric_bfv", None, vp, vp, p64, p64, sz, p64)
    sig("orc_fnv_words", u64, p64, sz)
    sig("orc_apply_galois", None, vp, sz, C.c_int, sz, p64, sz, p64)
    sig("orc_apply_galois_ct", None, vp, sz, C.c_int, sz, p64, C.POINTER(p64), p64)
    sig("orc_rns_mod_t_and_divide_q_last_ntt", None, vp, sz, p64, sz, p64)
    sig("orc_bgv_inv_q_last_mod_t", u64, vp, sz)
    sig("orc_rns_decrypt_mod_t", C.c_int, vp, sz, p64, p64)
    sig("orc_decrypt_bgv", C.c_int, vp, p64, p64, sz, sz, u64, p64)
    sig("orc_encrypt_asymmetric_bgv", C.c_int, vp, vp, p64, p64, sz, p64)
    sig("orc_negacyclic_shift", None, vp, sz, p64, sz, sz, p64)
    sig("orc_multiply_inv_degree", None, vp, sz, p64, sz, u64)
    sig("orc_extract_lwe", None, vp, sz, p64, sz, p64, p64)
    sig("orc_keygen_galois_key", None, vp, vp, p64, sz, p64)
    sig("orc_galois_element_from_step", sz, sz, C.c_int)
    sig("orc_plain_centralize", C.c_int, vp, sz, p64, sz, p64)
    sig("orc_multiply_plain_normal", C.c_int, vp, sz, p64, sz, p64, sz, p64)
    sig("orc_multiply_plain_ntt", None, vp, sz, p64, sz, p64, p64)
    sig("orc_keygen_relin_keys", None, vp, vp, p64, p64)
    sig("orc_rns_decrypt_scale_and_round", C.c_int, vp, p64, p64)
    sig("orc_rns_tool_decrypt_mod_t", C.c_int, vp, p64, p64)
    sig("orc_rns_tool_mod_t_and_divide_q_last_inplace", None, vp, p64)
    sig("orc_decrypt_bfv", C.c_int, vp, p64, p64, sz, sz, p64)
    sig("orc_batch_decode", C.c_int, vp, p64, p64)
    _lib = L
    return L


# ---- numpy helpers -------------------------------------------------------------------

def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(p64)


def arr(x):
    return np.ascontiguousarray(np.array(x, dtype=np.uint64))


def modulus(value):
    m = Modulus()
    if lib().orc_modulus_init(C.byref(m), value) != 0:
        raise ValueError("invalid modulus %d" % value)
    return m


def moduli_array(values):
    ms = (Modulus * len(values))()
    for i, v in enumerate(values):
        if lib().orc_modulus_init(C.byref(ms[i]), int(v)) != 0:
            raise ValueError("invalid modulus %d" % v)
    return ms


def get_primes(factor, bit_size, count):
    out = np.zeros(count, dtype=np.uint64)
    if lib().orc_get_primes(factor, bit_size, count, ptr(out)) < 0:
        raise ValueError("not enough primes")
    return [int(x) for x in out]


def coeff_modulus_create(n, bit_sizes):
    bs = (sz * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    if lib().orc_coeff_modulus_create(n, bs, len(bit_sizes), ptr(out)) != 0:
        raise ValueError("coeff_modulus_create failed")
    return [int(x) for x in out]


def fill_uniform(seed, bound, n):
    out = np.empty(n, dtype=np.uint64)
    lib().orc_fill_uniform(seed, bound, ptr(out), n)
    return out


def fnv1a64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    return int(lib().orc_fnv1a64(ptr(a), a.size))


def fnv_words(a):
    """The survey probe's digest: FNV-1a step per 64-bit word."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    return int(lib().orc_fnv_words(ptr(a), a.size))


def aes128_encrypt_block(block, key):
    b = (C.c_uint8 * 16)(*block)
    k = (C.c_uint8 * 16)(*key)
    lib().orc_aes128_encrypt_block(b, k)
    return bytes(b)


class Rng:
    """AES-128-CTR generator of the reference's HeContext (utils/random_generator.cu)."""

    def __init__(self, seed_low, seed_high=0):
        self.h = lib().orc_rng_create(seed_low, seed_high)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rng_destroy(self.h)
            self.h = None

    def sample_uint64(self):
        return int(lib().orc_rng_sample_uint64(self.h))

    def _sample(self, fn, n, q):
        out = np.zeros(len(q) * n, dtype=np.uint64)
        getattr(lib(), fn)(self.h, ptr(out), n, moduli_array(q), len(q))
        return out.reshape(len(q), n)

    def ternary(self, n, q):
        return self._sample("orc_sample_poly_ternary", n, q)

    def centered_binomial(self, n, q):
        return self._sample("orc_sample_poly_centered_binomial", n, q)

    def uniform(self, n, q):
        return self._sample("orc_sample_poly_uniform", n, q)


class NTTTables:
    def __init__(self, log_n, q):
        self.h = lib().orc_ntt_tables_create(log_n, q)
        if not self.h:
            raise ValueError("cannot create NTT tables for q=%d logN=%d" % (q, log_n))
        self.log_n, self.q = log_n, q

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ntt_tables_destroy(self.h)
            self.h = None

    @property
    def root(self):
        return int(lib().orc_ntt_tables_root(self.h))

    def root_power(self, i, inverse=False, quotient=False):
        return int(lib().orc_ntt_tables_root_power(self.h, i, int(inverse), int(quotient)))

    def inv_degree(self, quotient=False):
        return int(lib().orc_ntt_tables_inv_degree(self.h, int(quotient)))


def _table_handles(tables):
    hs = (vp * len(tables))(*[t.h if isinstance(t, NTTTables) else t for t in tables])
    return hs


def ntt_forward(data, pcount, ncomp, log_n, tables, mode=0, decomp=0):
    """in-place on a flat uint64 numpy array"""
    lib().orc_ntt_forward(ptr(data), pcount, ncomp, log_n, _table_handles(tables), len(tables), mode, decomp)
    return data


def ntt_inverse(data, pcount, ncomp, log_n, tables, mode=0, decomp=0):
    lib().orc_ntt_inverse(ptr(data), pcount, ncomp, log_n, _table_handles(tables), len(tables), mode, decomp)
    return data


class RNSTool:
    def __init__(self, n, q, t):
        qa = arr(q)
        self.h = lib().orc_rns_tool_create(n, ptr(qa), len(q), t)
        if not self.h:
            raise ValueError("cannot create RNSTool")
        self.n, self.q, self.t = n, list(q), t

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rns_tool_destroy(self.h)
            self.h = None

    @property
    def base_B_size(self):
        return int(lib().orc_rns_tool_base_B_size(self.h))

    @property
    def base_Bsk_size(self):
        return int(lib().orc_rns_tool_base_Bsk_size(self.h))

    @property
    def m_sk(self):
        return int(lib().orc_rns_tool_m_sk(self.h))

    @property
    def gamma(self):
        return int(lib().orc_rns_tool_gamma(self.h))

    @property
    def m_tilde(self):
        return int(lib().orc_rns_tool_m_tilde(self.h))

    @property
    def base_Bsk(self):
        out = np.zeros(self.base_Bsk_size, dtype=np.uint64)
        lib().orc_rns_tool_base_Bsk(self.h, ptr(out))
        return [int(x) for x in out]

    def _call(self, name, inp, out_len):
        inp = arr(inp)
        out = np.zeros(out_len, dtype=np.uint64)
        getattr(lib(), name)(self.h, ptr(inp), ptr(out))
        return out

    def decrypt_scale_and_round(self, phase):
        """RNSTool::decrypt_scale_and_round on a bare tool: phase [q_size][N] -> [N] mod t"""
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_rns_decrypt_scale_and_round(self.h, ptr(arr(phase)), ptr(out)) != 0:
            raise ValueError("decrypt_scale_and_round failed")
        return out

    def decrypt_mod_t(self, phase):
        """RNSTool::decrypt_mod_t (BGV): phase [q_size][N] -> [N] mod t"""
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_rns_tool_decrypt_mod_t(self.h, ptr(arr(phase)), ptr(out)) != 0:
            raise ValueError("decrypt_mod_t needs a plain modulus")
        return out

    def mod_t_and_divide_q_last_inplace(self, data):
        """RNSTool::mod_t_and_divide_q_last_inplace (BGV mod switch, coefficient form): [q_size][N] -> the same array, rows 0 .. q_size-2 replaced"""
        buf = arr(data).copy()
        lib().orc_rns_tool_mod_t_and_divide_q_last_inplace(self.h, ptr(buf))
        return buf

    def fast_b_conv_m_tilde(self, inp):
        return self._call("orc_rns_fast_b_conv_m_tilde", inp, (self.base_Bsk_size + 1) * self.n)

    def sm_mrq(self, inp):
        return self._call("orc_rns_sm_mrq", inp, self.base_Bsk_size * self.n)

    def fast_floor(self, inp):
        return self._call("orc_rns_fast_floor", inp, self.base_Bsk_size * self.n)

    def fast_b_conv_sk(self, inp):
        return self._call("orc_rns_fast_b_conv_sk", inp, len(self.q) * self.n)


class Context:
    """orc_context: key-level chain (K primes) + per-level RNS tools."""

    def __init__(self, scheme, n, coeff_modulus, plain_modulus=0):
        self.scheme = {"bfv": 1, "ckks": 2, "bgv": 3}[scheme] if isinstance(scheme, str) else scheme
        self.n = n
        self.log_n = n.bit_length() - 1
        self.q = [int(x) for x in coeff_modulus]
        self.K = len(self.q)
        self.t = plain_modulus
        qa = arr(self.q)
        self.h = lib().orc_context_create(self.scheme, n, ptr(qa), self.K, plain_modulus)
        if not self.h:
            raise ValueError("cannot create oracle context")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_context_destroy(self.h)
            self.h = None

    def moduli(self):
        return lib().orc_context_moduli(self.h)

    def table(self, i):
        return lib().orc_context_ntt_table(self.h, i)

    def rns_tool(self, nlimbs):
        return lib().orc_context_rns_tool(self.h, nlimbs)

    # all functions take / return flat or shaped uint64 numpy arrays (copied)
    def to_ntt(self, ct, pcount, L):
        out = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        lib().orc_transform_to_ntt(self.h, ptr(out.reshape(-1)), pcount, L)
        return out

    def from_ntt(self, ct, pcount, L):
        out = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        lib().orc_transform_from_ntt(self.h, ptr(out.reshape(-1)), pcount, L)
        return out

    def _keys(self, keys):
        ks = [np.ascontiguousarray(k, dtype=np.uint64).reshape(-1) for k in keys]
        arrp = (p64 * len(ks))(*[ptr(k) for k in ks])
        return ks, arrp

    def switch_key(self, L, is_ntt, target, keys, assign=1, dest=None):
        target = np.ascontiguousarray(target, dtype=np.uint64).reshape(-1)
        ks, arrp = self._keys(keys)
        if dest is None:
            dest = np.zeros(2 * L * self.n, dtype=np.uint64)
        else:
            dest = np.ascontiguousarray(dest, dtype=np.uint64).reshape(-1).copy()
        lib().orc_switch_key(self.h, L, int(is_ntt), ptr(target), arrp, assign, ptr(dest))
        return dest.reshape(2, L, self.n)

    def relinearize(self, L, is_ntt, ct3, keys):
        ct3 = np.ascontiguousarray(ct3, dtype=np.uint64).reshape(-1)
        ks, arrp = self._keys(keys)
        out = np.zeros(2 * L * self.n, dtype=np.uint64)
        lib().orc_relinearize(self.h, L, int(is_ntt), ptr(ct3), arrp, ptr(out))
        return out.reshape(2, L, self.n)

    def ckks_multiply(self, L, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        pa, pb = a.size // (L * self.n), b.size // (L * self.n)
        out = np.zeros((pa + pb - 1) * L * self.n, dtype=np.uint64)
        lib().orc_ckks_multiply(self.h, L, ptr(a.reshape(-1)), pa, ptr(b.reshape(-1)), pb, ptr(out))
        return out.reshape(pa + pb - 1, L, self.n)

    def bfv_multiply(self, L, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        pa, pb = a.size // (L * self.n), b.size // (L * self.n)
        out = np.zeros((pa + pb - 1) * L * self.n, dtype=np.uint64)
        lib().orc_bfv_multiply(self.h, L, ptr(a.reshape(-1)), pa, ptr(b.reshape(-1)), pb, ptr(out))
        return out.reshape(pa + pb - 1, L, self.n)

    def mod_switch_scale_to_next(self, L, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        p = ct.size // (L * self.n)
        out = np.zeros(p * (L - 1) * self.n, dtype=np.uint64)
        lib().orc_mod_switch_scale_to_next(self.h, L, ptr(ct.reshape(-1)), p, ptr(out))
        return out.reshape(p, L - 1, self.n)

    def mod_switch_drop_to_next(self, L, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        p = ct.size // (L * self.n)
        out = np.zeros(p * (L - 1) * self.n, dtype=np.uint64)
        lib().orc_mod_switch_drop_to_next(self.h, L, ptr(ct.reshape(-1)), p, ptr(out))
        return out.reshape(p, L - 1, self.n)

    # ---- BASELINE config 1 host path (keygen / encode / encrypt), used to pin the oracle ----
    def secret_key(self, rng):
        sk = np.zeros(self.K * self.n, dtype=np.uint64)
        lib().orc_keygen_secret_key(self.h, rng.h, ptr(sk))
        return sk.reshape(self.K, self.n)

    def public_key(self, rng, sk):
        pk = np.zeros(2 * self.K * self.n, dtype=np.uint64)
        lib().orc_keygen_public_key(self.h, rng.h, ptr(np.ascontiguousarray(sk).reshape(-1)), ptr(pk))
        return pk.reshape(2, self.K, self.n)

    def batch_encode(self, values):
        v = arr(values)
        plain = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_batch_encode(self.h, ptr(v), v.size, ptr(plain)) != 0:
            raise ValueError("batch_encode failed")
        return plain

    def encrypt_asymmetric_bfv(self, rng, pk, plain):
        out = np.zeros(2 * (self.K - 1) * self.n, dtype=np.uint64)
        lib().orc_encrypt_asymmetric_bfv(self.h, rng.h, ptr(np.ascontiguousarray(pk).reshape(-1)), ptr(plain), plain.size, ptr(out))
        return out.reshape(2, self.K - 1, self.n)

    # -- BGV (NTT-form ciphertexts; the correction factor is carried by the caller) ----------------------
    def encrypt_asymmetric_bgv(self, rng, pk, plain):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros(2 * (self.K - 1) * self.n, dtype=np.uint64)
        if lib().orc_encrypt_asymmetric_bgv(self.h, rng.h, ptr(np.ascontiguousarray(pk).reshape(-1)), ptr(plain), plain.size, ptr(out)) != 0:
            raise ValueError("encrypt failed")
        return out.reshape(2, self.K - 1, self.n)

    def decrypt_bgv(self, sk, ct, correction_factor=1):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        pcount, L = ct.shape[0], ct.shape[1]
        plain = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_decrypt_bgv(self.h, ptr(np.ascontiguousarray(sk).reshape(-1)), ptr(ct.reshape(-1)), pcount, L, int(correction_factor), ptr(plain)) != 0:
            raise ValueError("decrypt failed")
        return plain

    def decrypt_mod_t(self, L, phase):
        phase = np.ascontiguousarray(phase, dtype=np.uint64)
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_rns_decrypt_mod_t(self.h, L, ptr(phase.reshape(-1)), ptr(out)) != 0:
            raise ValueError("no plain modulus")
        return out

    def mod_t_and_divide_q_last_ntt(self, L, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        p = ct.size // (L * self.n)
        out = np.zeros(p * (L - 1) * self.n, dtype=np.uint64)
        lib().orc_rns_mod_t_and_divide_q_last_ntt(self.h, L, ptr(ct.reshape(-1)), p, ptr(out))
        return out.reshape(p, L - 1, self.n)

    def bgv_inv_q_last_mod_t(self, L):
        return int(lib().orc_bgv_inv_q_last_mod_t(self.h, L))

    def relin_keys(self, rng, sk):
        """list of K-1 keys u64[2][K][N] (KeyGenerator::create_relin_keys)"""
        L = self.K - 1
        out = np.zeros(L * 2 * self.K * self.n, dtype=np.uint64)
        lib().orc_keygen_relin_keys(self.h, rng.h, ptr(np.ascontiguousarray(sk).reshape(-1)), ptr(out))
        return [k.copy() for k in out.reshape(L, 2, self.K, self.n)]

    def decrypt_bfv(self, sk, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        pcount, L = ct.shape[0], ct.shape[1]
        plain = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_decrypt_bfv(self.h, ptr(np.ascontiguousarray(sk).reshape(-1)), ptr(ct.reshape(-1)), pcount, L, ptr(plain)) != 0:
            raise ValueError("decrypt failed")
        return plain

    def decrypt_scale_and_round(self, L, phase):
        phase = np.ascontiguousarray(phase, dtype=np.uint64)
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_rns_decrypt_scale_and_round(self.rns_tool(L), ptr(phase.reshape(-1)), ptr(out)) != 0:
            raise ValueError("decrypt_scale_and_round failed")
        return out

    def batch_decode(self, plain):
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().orc_batch_decode(self.h, ptr(np.ascontiguousarray(plain, dtype=np.uint64)), ptr(out)) != 0:
            raise ValueError("batch_decode failed")
        return out

    def galois_element_from_step(self, step):
        return int(lib().orc_galois_element_from_step(self.n, step))

    def apply_galois(self, nmod, is_ntt_form, element, data):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        p = data.size // (nmod * self.n)
        out = np.zeros(data.size, dtype=np.uint64)
        lib().orc_apply_galois(self.h, nmod, int(is_ntt_form), element, ptr(data.reshape(-1)), p, ptr(out))
        return out.reshape(data.shape)

    def apply_galois_ct(self, L, is_ntt_form, element, ct, keys):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        keep, karr = self._keys(keys)
        out = np.zeros(ct.size, dtype=np.uint64)
        lib().orc_apply_galois_ct(self.h, L, int(is_ntt_form), element, ptr(ct.reshape(-1)), karr, ptr(out))
        return out.reshape(2, L, self.n)

    def negacyclic_shift(self, nmod, data, shift):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros(data.size, dtype=np.uint64)
        lib().orc_negacyclic_shift(self.h, nmod, ptr(data.reshape(-1)), data.size // (nmod * self.n), shift, ptr(out))
        return out.reshape(data.shape)

    def multiply_inv_degree(self, nmod, data, scalar):
        out = np.array(data, dtype=np.uint64, copy=True, order="C")
        lib().orc_multiply_inv_degree(self.h, nmod, ptr(out.reshape(-1)), out.size // (nmod * self.n), scalar)
        return out

    def extract_lwe(self, L, ct, term):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        c0, c1 = np.zeros(L, dtype=np.uint64), np.zeros(L * self.n, dtype=np.uint64)
        lib().orc_extract_lwe(self.h, L, ptr(ct.reshape(-1)), term, ptr(c0), ptr(c1))
        return c0, c1.reshape(L, self.n)

    def assemble_lwe(self, L, c0, c1):
        """LWECiphertext::assemble_lwe (lwe_ciphertext.cu:9-60): c0 becomes the constant coefficient, c1 is kept"""
        ct = np.zeros((2, L, self.n), dtype=np.uint64)
        ct[0, :, 0] = c0
        ct[1] = c1
        return ct

    def _ct_add(self, L, a, b, subtract=False):
        q = np.array(self.q[:L], dtype=np.uint64).reshape(1, L, 1)
        return (a + (q - b if subtract else b)) % q          # q < 2^61: no wrap

    def pack_rlwe_ciphertexts(self, L, cts, keys_by_element, shift, input_interval, output_interval, is_ntt_form=False, apply_field_trace=True):
        """Evaluator::pack_rlwe_ciphertexts_new, host branch (evaluator_lwes.cu:315-439, :479-489), for BFV (results stay in
        coefficient form).  cts: list of [2][L][N]; keys_by_element: {galois element: the L keys of that element}."""
        n = self.n
        maxc = input_interval // output_interval
        layers = maxc.bit_length() - 1
        assert 1 <= len(cts) <= maxc and (1 << layers) == maxc
        rl = [None] * maxc
        for i in range(maxc):
            index = int("{:0{w}b}".format(i, w=layers)[::-1], 2) if layers else 0
            if index < len(cts):
                c = np.array(cts[index], dtype=np.uint64, copy=True)
                if is_ntt_form:
                    c = self.from_ntt(c, 2, L)
                c = self.multiply_inv_degree(L, c, n // input_interval)
                if shift:
                    c = self.negacyclic_shift(L, c, shift)
                rl[i] = c
        for layer in range(layers):
            gap, sh = 1 << layer, input_interval >> (layer + 1)
            g = (n // input_interval) * (1 << (layer + 1)) + 1
            for offset in range(0, maxc, gap * 2):
                even, odd = rl[offset], rl[offset + gap]
                if even is None and odd is None:
                    continue
                temp = self.negacyclic_shift(L, odd, sh) if odd is not None else None
                if even is not None:
                    if odd is not None:
                        odd2 = self._ct_add(L, even, temp, subtract=True)
                        even = self._ct_add(L, even, temp)
                        even = self._ct_add(L, even, self.apply_galois_ct(L, False, g, odd2, keys_by_element[g]))
                    else:
                        even = self._ct_add(L, even, self.apply_galois_ct(L, False, g, even, keys_by_element[g]))
                else:
                    neg = self._ct_add(L, np.zeros_like(temp), temp, subtract=True)
                    even = self._ct_add(L, self.apply_galois_ct(L, False, g, neg, keys_by_element[g]), temp)
                rl[offset] = even
        ret = rl[0]
        if output_interval != 1 and apply_field_trace:
            logn = (n // output_interval).bit_length() - 1
            ret = self.field_trace(L, ret, keys_by_element, logn)
        return ret

    def field_trace(self, L, ct, keys_by_element, logn):
        """Evaluator::field_trace_inplace (evaluator_lwes.cu:100-109), coefficient form"""
        d = self.n
        while d > (1 << logn):
            ct = self._ct_add(L, ct, self.apply_galois_ct(L, False, d + 1, ct, keys_by_element[d + 1]))
            d >>= 1
        return ct

    def galois_key(self, rng, sk, element):
        L = self.K - 1
        out = np.zeros(L * 2 * self.K * self.n, dtype=np.uint64)
        lib().orc_keygen_galois_key(self.h, rng.h, ptr(np.ascontiguousarray(sk).reshape(-1)), element, ptr(out))
        return [k.copy() for k in out.reshape(L, 2, self.K, self.n)]

    def plain_centralize(self, L, plain):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros(L * self.n, dtype=np.uint64)
        if lib().orc_plain_centralize(self.h, L, ptr(plain), plain.size, ptr(out)) != 0:
            raise ValueError("no fast plain lift")
        return out.reshape(L, self.n)

    def multiply_plain_normal(self, L, ct, plain):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        p = ct.size // (L * self.n)
        out = np.zeros(ct.size, dtype=np.uint64)
        if lib().orc_multiply_plain_normal(self.h, L, ptr(ct.reshape(-1)), p, ptr(plain), plain.size, ptr(out)) != 0:
            raise ValueError("multiply_plain failed")
        return out.reshape(p, L, self.n)

    def multiply_plain_ntt(self, L, ct, plain_ntt):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        plain_ntt = np.ascontiguousarray(plain_ntt, dtype=np.uint64)
        p = ct.size // (L * self.n)
        out = np.zeros(ct.size, dtype=np.uint64)
        lib().orc_multiply_plain_ntt(self.h, L, ptr(ct.reshape(-1)), p, ptr(plain_ntt.reshape(-1)), ptr(out))
        return out.reshape(p, L, self.n)

    def random_ct(self, seed, pcount, L):
        """uniform residues x[p][l][i] in [0, q_l) from the shared splitmix generator"""
        out = np.empty((pcount, L, self.n), dtype=np.uint64)
        for p in range(pcount):
            for l in range(L):
                out[p, l] = fill_uniform(seed * 1000003 + p * 101 + l, self.q[l], self.n)
        return out

    def random_keys(self, seed, L):
        """L key-switching keys, each u64[2][K][N] of uniform residues (NTT form)"""
        keys = []
        for j in range(L):
            k = np.empty((2, self.K, self.n), dtype=np.uint64)
            for c in range(2):
                for l in range(self.K):
                    k[c, l] = fill_uniform(seed * 7919 + j * 257 + c * 31 + l, self.q[l], self.n)
            keys.append(k)
        return keys


class Ring2k:
    """PolynomialEncoderRNSHelper<T> of the reference's ring-2^k application (src/app/bfv_ring2k.cu:96-194 constants, :197-297
    scale_up, :488-506 centralize, :620-735 scale_down), restated on Python integers (exact multi-precision arithmetic; the
    wrap-around of the element type T is applied where the reference's code has it).  q: the level's primes, n: ring degree,
    t_bits: k, elem_bits: 32 / 64 / 128."""

    def __init__(self, n, q, t_bits, elem_bits):
        assert elem_bits in (32, 64, 128) and elem_bits // 2 < t_bits <= elem_bits
        self.n, self.q, self.k, self.bits = n, [int(v) for v in q], t_bits, elem_bits
        self.mask = (1 << t_bits) - 1
        self.t_half = 1 << (t_bits - 1)
        self.Q = 1
        for v in self.q:
            self.Q *= v
        self.gamma = int(get_primes(n, 61, 1)[0])                     # utils::get_prime(poly_degree, 61) (:107)
        self.Q_mod_t = self.Q & self.mask
        self.Q_div_t = [(self.Q >> t_bits) % v for v in self.q]
        self.punct = [self.Q // v for v in self.q]
        self.inv_punct = [pow(p % v, -1, v) if len(self.q) > 1 else 1 for p, v in zip(self.punct, self.q)]
        self.neg_inv_Q_mod_t = (-pow(self.Q, -1, 1 << t_bits)) & self.mask
        self.inv_gamma_mod_t = pow(self.gamma, -1, 1 << t_bits)
        self.neg_inv_Q_mod_gamma = (-pow(self.Q % self.gamma, -1, self.gamma)) % self.gamma
        self.gamma_t_mod_q = [(self.gamma % v) * ((1 << t_bits) % v) % v for v in self.q]

    def scale_up(self, src):
        out = np.zeros((len(self.q), self.n), dtype=np.uint64)
        tmask = (1 << self.bits) - 1
        for j, x in enumerate(int(v) for v in src):
            for i, qi in enumerate(self.q):
                u = (x % qi) * self.Q_div_t[i] % qi
                if self.bits <= 64:
                    v = (((self.Q_mod_t * x + self.t_half) & ((1 << 128) - 1)) >> self.k) & tmask     # T v = (...) >> k
                    out[i, j] = ((u + v) & ((1 << 64) - 1)) % qi                                        # reduce(u + v), 64-bit sum
                else:
                    v = ((self.Q_mod_t * x + self.t_half) >> self.k) & ((1 << 128) - 1)
                    out[i, j] = ((u + v) & ((1 << 128) - 1)) % qi
        return out

    def centralize(self, src):
        out = np.zeros((len(self.q), self.n), dtype=np.uint64)
        for j, x in enumerate(int(v) for v in src):
            for i, qi in enumerate(self.q):
                out[i, j] = (qi - ((-x) & self.mask) % qi) % qi if x > self.t_half else x % qi
        return out

    def scale_down(self, phase):
        out = []
        for c in range(self.n):
            y = [int(phase[l, c]) * self.gamma_t_mod_q[l] % ql * self.inv_punct[l] % ql for l, ql in enumerate(self.q)]
            on_gamma = sum(yl * (p % self.gamma) for yl, p in zip(y, self.punct)) % self.gamma
            on_gamma = on_gamma * self.neg_inv_Q_mod_gamma % self.gamma
            on_t = sum(yl * (p & self.mask) for yl, p in zip(y, self.punct)) * self.neg_inv_Q_mod_t & self.mask
            if on_gamma > self.gamma >> 1:
                out.append(((on_t + self.gamma - on_gamma) * self.inv_gamma_mod_t) & self.mask)
            else:
                out.append(((on_t - on_gamma) * self.inv_gamma_mod_t) & self.mask)
        return out

    def decentralize(self, plain, correction_factor=1):
        """PolynomialEncoderRNSHelper::decentralize (:752-911): x mod Q -> x mod 2^k by the fast base conversion with the quotient estimated in doubles
        (step 1: y_l = x_l (Q/q_l)^-1 mod q_l, v_l = double(y_l) / double(q_l); step 2: v = round(sum v_l) summed in the order l = 0 .. L-1,
        out = sum y_l ((Q/q_l) mod 2^k) - v (Q mod 2^k)), then times correction_factor^-1 mod 2^k (inverse_ring2k, :33-40: the factor must be odd)."""
        import math
        cf = int(correction_factor) & ((1 << self.bits) - 1)
        if cf & 1 == 0:
            raise ValueError("[bfv_ring2k::inverse_ring2k] x must be odd")
        fix = pow(cf, -1, 1 << self.bits) & self.mask
        out = []
        for c in range(self.n):
            y = [int(plain[l, c]) % ql * self.inv_punct[l] % ql for l, ql in enumerate(self.q)]
            v = 0.0
            for yl, ql in zip(y, self.q):
                v += float(yl) / float(ql)
            r = math.floor(v)
            if v - r >= 0.5:                                   # std::round: halves away from zero (v >= 0)
                r += 1
            x = (sum(yl * (p & self.mask) for yl, p in zip(y, self.punct)) - r * self.Q_mod_t) & self.mask
            out.append(x * fix & self.mask)
        return out
