"""Host-side handle objects over the C-ABI: Plan (one modulus chain on one GPU) and Behz.

Ciphertext payloads are torch int64 CUDA tensors whose bits are the reference's uint64 words,
layout [batch][poly][limb][N] (ciphertext.h:211-247).  Every method only enqueues work on the
current torch stream; nothing is computed on the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import capi

IDX_COMPONENTWISE, IDX_KS_SET_PRODUCTS, IDX_KS_SKIP_FINALS = 0, 1, 2
ASSIGN_ADD_INPLACE, ASSIGN_OVERWRITE, ASSIGN_OVERWRITE_EXCEPT_FIRST = 0, 1, 2


def to_device(a, device):
    """numpy uint64 array -> int64 CUDA tensor holding the same bits"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    """int64 tensor -> numpy uint64 array (same bits)"""
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


def _ptr(t):
    if t.dtype != torch.int64 or not t.is_contiguous() or not t.is_cuda:
        raise capi.TroynInvalidArgument("[troyn] operands must be contiguous int64 CUDA tensors")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """troyn_plan: device tables for the key-level modulus chain q_0..q_{K-1} (special prime last).

    Mirrors what ContextData::to_device_inplace (context_data.cu:34-69) uploads for every level.
    """

    def __init__(self, device, log_n, moduli, roots=None):
        self.lib = capi.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise capi.TroynError("[troyn] the hot path runs on an MI355X only; there is no CPU path")
        self.log_n, self.n = int(log_n), 1 << int(log_n)
        self.moduli = [int(q) for q in moduli]
        self.K = len(self.moduli)
        h = C.c_void_p()
        m = (C.c_uint64 * self.K)(*self.moduli)
        r = (C.c_uint64 * self.K)(*[int(x) for x in roots]) if roots is not None else None
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        capi.check(self.lib.troyn_plan_create(C.byref(h), index, self.log_n, self.K, m, r))
        self.h = h
        self._ws = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.troyn_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value=None):
        """an A/B switch of this plan (the library reads the TROYN_* environment variables once, in troyn_plan_create; never on a call)"""
        capi.check(self.lib.troyn_plan_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    # -- table inspection (known-answer hooks) --------------------------------------------
    def root(self, i):
        out = C.c_uint64()
        capi.check(self.lib.troyn_plan_get_root(self.h, i, C.byref(out)))
        return int(out.value)

    def root_powers(self, i, inverse=False):
        out = np.zeros(2 * self.n, dtype=np.uint64)
        capi.check(self.lib.troyn_plan_get_root_powers(self.h, i, int(inverse), out.ctypes.data_as(capi.p64)))
        return out.reshape(self.n, 2)

    # -- workspace (stands in for the reference's MemoryPool) -------------------------------
    def workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)
        return self._ws

    # -- NTT ---------------------------------------------------------------------------------
    def ntt(self, x, pcount, ncomp, inverse=False, out=None, table_start=0, table_count=None,
            mode=IDX_COMPONENTWISE, decomp=0):
        """x: [batch][pcount][ncomp][N] (any leading shape with that many elements)"""
        if table_count is None:
            table_count = ncomp if mode == IDX_COMPONENTWISE else self.K
        out = x if out is None else out
        batch = x.numel() // (pcount * ncomp * self.n)
        capi.check(self.lib.troyn_ntt(self.h, int(inverse), _ptr(x), _ptr(out), batch, pcount, ncomp,
                                      table_start, table_count, mode, decomp, _stream()))
        return out

    # -- element-wise ---------------------------------------------------------------------------
    def _ew(self, fn, a, b, nmod, mod_start, out):
        out = torch.empty_like(a) if out is None else out
        count = a.numel() // (nmod * self.n)
        if b is None:
            capi.check(fn(self.h, mod_start, nmod, _ptr(a), _ptr(out), count, _stream()))
        else:
            capi.check(fn(self.h, mod_start, nmod, _ptr(a), _ptr(b), _ptr(out), count, _stream()))
        return out

    def add(self, a, b, nmod, mod_start=0, out=None):
        return self._ew(self.lib.troyn_add, a, b, nmod, mod_start, out)

    def sub(self, a, b, nmod, mod_start=0, out=None):
        return self._ew(self.lib.troyn_sub, a, b, nmod, mod_start, out)

    def negate(self, a, nmod, mod_start=0, out=None):
        return self._ew(self.lib.troyn_negate, a, None, nmod, mod_start, out)

    def modulo(self, a, nmod, mod_start=0, out=None):
        """utils::modulo_ps (utils/poly_small_mod.cu:119-180): Barrett-64 reduction of arbitrary 64-bit words"""
        return self._ew(self.lib.troyn_modulo, a, None, nmod, mod_start, out)

    def multiply_uint64operand(self, a, operands, nmod, mod_start=0, out=None):
        """utils::multiply_uint64operand_ps (utils/poly_small_mod.cu:752-814); operands: int64 CUDA tensor [nmod][2] = (operand, quotient)"""
        return self._ew(self.lib.troyn_multiply_uint64operand, a, operands, nmod, mod_start, out)

    def dyadic_product(self, a, b, nmod, mod_start=0, out=None):
        return self._ew(self.lib.troyn_dyadic_product, a, b, nmod, mod_start, out)

    def multiply_scalar(self, a, scalar, nmod, mod_start=0, out=None):
        out = torch.empty_like(a) if out is None else out
        count = a.numel() // (nmod * self.n)
        capi.check(self.lib.troyn_multiply_scalar(self.h, mod_start, nmod, _ptr(a), int(scalar), _ptr(out), count, _stream()))
        return out

    def dyadic_convolute(self, a, pa, b, pb, nmod, mod_start=0, out=None):
        batch = a.numel() // (pa * nmod * self.n)
        if out is None:
            out = torch.empty((batch, pa + pb - 1, nmod, self.n), dtype=torch.int64, device=a.device)
        capi.check(self.lib.troyn_dyadic_convolute(self.h, mod_start, nmod, _ptr(a), pa, _ptr(b), pb, _ptr(out), batch, _stream()))
        return out

    def dyadic_square(self, a, nmod, mod_start=0, out=None):
        batch = a.numel() // (2 * nmod * self.n)
        if out is None:
            out = torch.empty((batch, 3, nmod, self.n), dtype=torch.int64, device=a.device)
        capi.check(self.lib.troyn_dyadic_square(self.h, mod_start, nmod, _ptr(a), _ptr(out), batch, _stream()))
        return out

    # -- ciphertext x plaintext ------------------------------------------------------------------
    def plain_centralize(self, L, t, plain, out=None):
        """plain [batch][N] mod t -> [batch][L][N] (scaling_variant::centralize)"""
        batch = plain.numel() // self.n
        if out is None:
            out = torch.empty((batch, L, self.n), dtype=torch.int64, device=plain.device)
        capi.check(self.lib.troyn_plain_centralize(self.h, L, int(t), _ptr(plain), self.n, self.n, _ptr(out), batch, _stream()))
        return out

    def plain_centralize_ntt(self, L, t, plain, coeff_count=None, out=None):
        """plain [batch][stride] mod t (coeff_count <= stride <= N words used per row) -> NTT form [batch][L][N] in one launch
        (Evaluator::transform_plain_to_ntt = scaling_variant::centralize + forward NTT)"""
        stride = plain.shape[-1]
        batch = plain.numel() // stride
        cc = stride if coeff_count is None else int(coeff_count)
        if out is None:
            out = torch.empty((batch, L, self.n), dtype=torch.int64, device=plain.device)
        capi.check(self.lib.troyn_plain_centralize_ntt(self.h, L, int(t), _ptr(plain), cc, stride, _ptr(out), batch, _stream()))
        return out

    def dyadic_broadcast_product(self, ct, pcount, pt, nmod, shared_plain=False, mod_start=0, out=None):
        """ct [batch][pcount][nmod][N] (.) pt [batch][nmod][N] (or one shared [nmod][N])"""
        batch = ct.numel() // (pcount * nmod * self.n)
        out = torch.empty_like(ct) if out is None else out
        capi.check(self.lib.troyn_dyadic_broadcast_product(self.h, mod_start, nmod, _ptr(ct), pcount, _ptr(pt),
                                                           0 if shared_plain else nmod * self.n, _ptr(out), batch, _stream()))
        return out

    def multiply_plain_accumulate(self, cts, pts, dsts, pcount, nmod, set_zero=True, mod_start=0):
        """dsts[k] (+)= cts[k] (.) pts[k]; equal destination tensors accumulate (multiply_plain_ntt_accumulate)"""
        count = len(cts)
        arr = lambda ts: (C.c_void_p * count)(*[t.data_ptr() for t in ts])
        for t in list(cts) + list(pts) + list(dsts):
            _ptr(t)
        nbytes = self.lib.troyn_multiply_plain_accumulate_workspace_bytes(count)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)
        capi.check(self.lib.troyn_multiply_plain_accumulate(self.h, mod_start, nmod, pcount, arr(cts), arr(pts), arr(dsts), count,
                                                            int(set_zero), C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
        return dsts

    # -- Galois automorphisms -----------------------------------------------------------------------
    def apply_galois_poly(self, x, nmod, element, is_ntt_form, mod_start=0, out=None):
        """X -> X^element on every polynomial of x [count][nmod][N] (GaloisTool::apply_ps / apply_ntt_ps)"""
        count = x.numel() // (nmod * self.n)
        out = torch.empty_like(x) if out is None else out
        capi.check(self.lib.troyn_apply_galois(self.h, mod_start, nmod, int(is_ntt_form), int(element), _ptr(x), _ptr(out), count, _stream()))
        return out

    def apply_galois_plain(self, x, modulus, element):
        """X -> X^element on coefficient-form polynomials [count][N] modulo one explicit modulus (GaloisTool::apply with the plain modulus)"""
        out = torch.empty_like(x)
        capi.check(self.lib.troyn_apply_galois_plain(self.h, int(modulus), int(element), _ptr(x), _ptr(out), x.numel() // self.n, _stream()))
        return out

    def apply_galois(self, L, ct, element, keys, is_ckks=True, is_ntt_form=True):
        """ct [batch][2][L][N] -> Evaluator::apply_galois: permute both polynomials, key-switch the second"""
        out = self.apply_galois_poly(ct, L, element, is_ntt_form)
        batch = ct.numel() // (2 * L * self.n)
        target = out.view(batch, 2, L, self.n)[:, 1].contiguous()
        self.switch_key(L, target, keys, dest=out.view(batch, 2, L, self.n), assign=ASSIGN_OVERWRITE_EXCEPT_FIRST,
                        is_ckks=is_ckks, is_ntt_form=is_ntt_form)
        return out

    # -- RLWE / LWE packing (evaluator_lwes.cu) -----------------------------------------------------
    def negacyclic_shift(self, x, nmod, shift, mod_start=0):
        """x [count][nmod][N] * X^shift (utils::negacyclic_shift_ps), any shift (modulo 2N)"""
        out = torch.empty_like(x)
        capi.check(self.lib.troyn_negacyclic_shift(self.h, mod_start, nmod, _ptr(x), _ptr(out), int(shift), x.numel() // (nmod * self.n), _stream()))
        return out

    def multiply_inv_degree(self, x, nmod, scalar, mod_start=0, out=None):
        """x * N^-1 * scalar (utils::ntt_multiply_inv_degree); in place when out is x"""
        out = torch.empty_like(x) if out is None else out
        capi.check(self.lib.troyn_multiply_inv_degree(self.h, mod_start, nmod, _ptr(x), _ptr(out), int(scalar), x.numel() // (nmod * self.n), _stream()))
        return out

    def extract_lwe(self, L, cts, terms):
        """Evaluator::extract_lwe_new for (cts[i], terms[i]); cts: coefficient-form tensors [2][L][N] -> c0 [count][L], c1 [count][L][N]"""
        count = len(cts)
        c0 = torch.empty((count, L), dtype=torch.int64, device=self.device)
        c1 = torch.empty((count, L, self.n), dtype=torch.int64, device=self.device)
        ptrs = (C.c_void_p * count)(*[c.data_ptr() for c in cts])
        tarr = (C.c_size_t * count)(*[int(t) for t in terms])
        ws = self.workspace(self.lib.troyn_extract_lwe_workspace_bytes(count))
        capi.check(self.lib.troyn_extract_lwe(self.h, L, ptrs, tarr, _ptr(c0), _ptr(c1), count, C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
        return c0, c1

    def pack_rlwe_ciphertexts(self, L, groups, keys_by_element, shift, input_interval, output_interval, is_ckks=False, apply_field_trace=True):
        """Evaluator::pack_rlwe_ciphertexts_new_batched (evaluator_lwes.cu:491-681) for coefficient-form two-polynomial
        ciphertexts: groups = list of lists of tensors [2][L][N]; returns [len(groups)][2][L][N] (coefficient form).
        One working buffer holds every group's slots in bit-reversed order; each layer is troyn_pack_layer + one batched
        troyn_switch_key and halves the buffer."""
        n, G = self.n, len(groups)
        maxc = input_interval // output_interval
        layers = maxc.bit_length() - 1
        slots = G * maxc
        src = (C.c_void_p * slots)()
        for g, cts in enumerate(groups):
            if not 1 <= len(cts) <= maxc:
                raise capi.TroynInvalidArgument("[Evaluator::pack_rlwe_ciphertexts_new] ciphers count must be less than input_interval / output_interval.")
            for i in range(maxc):
                index = int("{:0{w}b}".format(i, w=layers)[::-1], 2) if layers else 0
                src[g * maxc + i] = cts[index].data_ptr() if index < len(cts) else None
        cur = torch.empty((slots, 2, L, n), dtype=torch.int64, device=self.device)
        ws = self.workspace(self.lib.troyn_pack_prepare_workspace_bytes(slots))
        capi.check(self.lib.troyn_pack_prepare(self.h, L, 2, src, slots, n // input_interval, int(shift), _ptr(cur),
                                               C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
        for layer in range(layers):
            pairs = cur.shape[0] // 2
            g = (n // input_interval) * (1 << (layer + 1)) + 1
            nxt = torch.empty((pairs, 2, L, n), dtype=torch.int64, device=self.device)
            target = torch.empty((pairs, L, n), dtype=torch.int64, device=self.device)
            capi.check(self.lib.troyn_pack_layer(self.h, L, g, input_interval >> (layer + 1), _ptr(cur), _ptr(nxt), _ptr(target), pairs, _stream()))
            self.switch_key(L, target, keys_by_element[g], dest=nxt, assign=ASSIGN_ADD_INPLACE, is_ckks=is_ckks, is_ntt_form=False)
            cur = nxt
        if output_interval != 1 and apply_field_trace:
            d = n
            while d > n // output_interval:
                cur = self.add(cur, self.apply_galois(L, cur, d + 1, keys_by_element[d + 1], is_ckks=is_ckks, is_ntt_form=False), L)
                d >>= 1
        return cur

    # -- key switching -----------------------------------------------------------------------------
    def _key_ptrs(self, keys, L):
        if len(keys) < L:
            raise capi.TroynInvalidArgument("[Evaluator::switch_key_inplace_internal] Key switch keys index out of range.")
        arr = (C.c_void_p * L)(*[k.data_ptr() for k in keys[:L]])
        return arr

    def switch_key(self, L, target, keys, dest=None, assign=ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True):
        """target [batch][L][N]; keys: list of L tensors [2][K][N]; dest [batch][2][L][N]"""
        batch = target.numel() // (L * self.n)
        if dest is None:
            dest = torch.zeros((batch, 2, L, self.n), dtype=torch.int64, device=target.device)
        nbytes = self.lib.troyn_switch_key_workspace_bytes(self.h, L, batch)
        ws = self.workspace(nbytes)
        capi.check(self.lib.troyn_switch_key(self.h, L, int(is_ckks), int(is_ntt_form), _ptr(target), self._key_ptrs(keys, L),
                                             assign, _ptr(dest), C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return dest

    def relinearize(self, L, ct3, keys, out=None, is_ckks=True, is_ntt_form=True):
        """ct3 [batch][3][L][N] -> [batch][2][L][N]"""
        batch = ct3.numel() // (3 * L * self.n)
        if out is None:
            out = torch.empty((batch, 2, L, self.n), dtype=torch.int64, device=ct3.device)
        nbytes = self.lib.troyn_relinearize_workspace_bytes(self.h, L, batch)
        ws = self.workspace(nbytes)
        capi.check(self.lib.troyn_relinearize(self.h, L, int(is_ckks), int(is_ntt_form), _ptr(ct3), self._key_ptrs(keys, L),
                                              _ptr(out), C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return out

    def ckks_multiply_relinearize_rescale(self, L, a, b, keys, out=None):
        """a, b [batch][2][L][N] (NTT form) -> rescale_to_next(relinearize(multiply(a, b))) [batch][2][L-1][N], one call"""
        batch = a.numel() // (2 * L * self.n)
        if out is None:
            out = torch.empty((batch, 2, L - 1, self.n), dtype=torch.int64, device=a.device)
        nbytes = self.lib.troyn_ckks_multiply_relinearize_rescale_workspace_bytes(self.h, L, batch)
        ws = self.workspace(nbytes)
        capi.check(self.lib.troyn_ckks_multiply_relinearize_rescale(self.h, L, _ptr(a), _ptr(b), self._key_ptrs(keys, L), _ptr(out),
                                                                    C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return out

    # -- modulus switching -------------------------------------------------------------------------
    def divide_and_round_q_last(self, L, x, pcount, out=None):
        batch = x.numel() // (pcount * L * self.n)
        if out is None:
            out = torch.empty((batch, pcount, L - 1, self.n), dtype=torch.int64, device=x.device)
        capi.check(self.lib.troyn_divide_and_round_q_last(self.h, L, _ptr(x), pcount, _ptr(out), batch, _stream()))
        return out

    def divide_and_round_q_last_ntt(self, L, x, pcount, out=None):
        batch = x.numel() // (pcount * L * self.n)
        if out is None:
            out = torch.empty((batch, pcount, L - 1, self.n), dtype=torch.int64, device=x.device)
        nbytes = self.lib.troyn_divide_and_round_q_last_ntt_workspace_bytes(self.h, L, pcount, batch)
        ws = self.workspace(nbytes)
        capi.check(self.lib.troyn_divide_and_round_q_last_ntt(self.h, L, _ptr(x), pcount, _ptr(out),
                                                              C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return out

    def mod_switch_drop(self, L_in, L_out, x, pcount, out=None):
        batch = x.numel() // (pcount * L_in * self.n)
        if out is None:
            out = torch.empty((batch, pcount, L_out, self.n), dtype=torch.int64, device=x.device)
        capi.check(self.lib.troyn_mod_switch_drop(self.h, L_in, L_out, _ptr(x), pcount, _ptr(out), batch, _stream()))
        return out


class Prng:
    """The context generator (AES-128-CTR, utils/random_generator.cu): seed + block counter on the host, sampling
    kernels on the device."""

    def __init__(self, plan, seed_low, seed_high=0):
        self.plan = plan
        self.seed = np.array([seed_low, seed_high], dtype=np.uint64)
        self.counter = 0

    def sample_uint64(self):
        out = np.zeros(2, dtype=np.uint64)
        capi.check(self.plan.lib.troyn_prng_block(self.seed.ctypes.data_as(capi.p64), self.counter, out.ctypes.data_as(capi.p64)))
        self.counter += 1
        return int(out[0])

    def _sample(self, fn, nmod):
        out = torch.empty((nmod, self.plan.n), dtype=torch.int64, device=self.plan.device)
        used = np.zeros(1, dtype=np.uint64)
        capi.check(getattr(self.plan.lib, fn)(self.plan.h, nmod, self.seed.ctypes.data_as(capi.p64), self.counter, _ptr(out),
                                              used.ctypes.data_as(capi.p64), _stream()))
        self.counter += int(used[0])
        return out

    def ternary(self, nmod):
        return self._sample("troyn_sample_ternary", nmod)

    def centered_binomial(self, nmod):
        return self._sample("troyn_sample_centered_binomial", nmod)

    def uniform(self, nmod):
        return self._sample("troyn_sample_uniform", nmod)

    def centered_binomial_strided(self, nmod, count, counter_stride):
        """`count` noise polynomials; item i reads the blocks from counter + i*counter_stride (the positions `count`
        sequential encryptions would use).  The counter is not advanced: the caller owns the block range."""
        out = torch.empty((count, nmod, self.plan.n), dtype=torch.int64, device=self.plan.device)
        capi.check(self.plan.lib.troyn_sample_centered_binomial_strided(self.plan.h, nmod, self.seed.ctypes.data_as(capi.p64), self.counter,
                                                                        int(counter_stride), _ptr(out), count, _stream()))
        return out


def sample_uniform_multi(plan, nmod, seeds):
    """One uniform RNS polynomial per seed pair (low, high), each from block 0 of its own generator (the c1 generators
    of `len(seeds)` symmetric encryptions, utils/rlwe.cu:252-262)."""
    sd = np.ascontiguousarray(np.array(seeds, dtype=np.uint64).reshape(-1, 2))
    out = torch.empty((sd.shape[0], nmod, plan.n), dtype=torch.int64, device=plan.device)
    capi.check(plan.lib.troyn_sample_uniform_multi(plan.h, nmod, sd.ctypes.data_as(capi.p64), _ptr(out), sd.shape[0], _stream()))
    return out


class Bgv:
    """troyn_bgv: the BGV-only constants of one level (RNSTool, utils/rns_tool.cu:205-232) and the steps that use them."""

    def __init__(self, plan, L, plain_modulus):
        self.plan, self.L, self.t = plan, int(L), int(plain_modulus)
        h = C.c_void_p()
        capi.check(plan.lib.troyn_bgv_create(C.byref(h), plan.h, self.L, self.t))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.plan.lib.troyn_bgv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def inv_q_last_mod_t(self):
        return int(self.plan.lib.troyn_bgv_inv_q_last_mod_t(self.h))

    def mod_t_and_divide_q_last_ntt(self, x, pcount):
        """x [batch][pcount][L][N] NTT form -> [batch][pcount][L-1][N]"""
        n, L = self.plan.n, self.L
        batch = x.numel() // (pcount * L * n)
        out = torch.empty((batch, pcount, L - 1, n), dtype=torch.int64, device=x.device)
        ws = self.plan.workspace(self.plan.lib.troyn_bgv_mod_switch_workspace_bytes(self.h, pcount, batch))
        capi.check(self.plan.lib.troyn_bgv_mod_t_and_divide_q_last_ntt(self.h, _ptr(x), pcount, _ptr(out), C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return out

    def decrypt_mod_t(self, phase, correction_factor=1):
        """phase [batch][L][N] coefficient form -> [batch][N] mod t (times correction_factor^-1)"""
        n = self.plan.n
        batch = phase.numel() // (self.L * n)
        out = torch.empty((batch, n), dtype=torch.int64, device=phase.device)
        capi.check(self.plan.lib.troyn_bgv_decrypt_mod_t(self.h, _ptr(phase), int(correction_factor), _ptr(out), batch, _stream()))
        return out

    def multiply_scalar_mod_t(self, x, scalar):
        out = torch.empty_like(x)
        capi.check(self.plan.lib.troyn_bgv_multiply_scalar_mod_t(self.h, _ptr(x), int(scalar), _ptr(out), x.numel(), _stream()))
        return out

    def switch_key(self, L, target, keys, dest=None, assign=ASSIGN_OVERWRITE):
        """self must be the key level (L = plan.K); target [batch][L][N] NTT form"""
        plan = self.plan
        batch = target.numel() // (L * plan.n)
        if dest is None:
            dest = torch.zeros((batch, 2, L, plan.n), dtype=torch.int64, device=target.device)
        ws = plan.workspace(plan.lib.troyn_switch_key_workspace_bytes(plan.h, L, batch))
        capi.check(plan.lib.troyn_bgv_switch_key(self.h, L, _ptr(target), plan._key_ptrs(keys, L), assign, _ptr(dest), C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return dest

    def relinearize(self, L, ct3, keys):
        plan = self.plan
        batch = ct3.numel() // (3 * L * plan.n)
        out = torch.empty((batch, 2, L, plan.n), dtype=torch.int64, device=ct3.device)
        ws = plan.workspace(plan.lib.troyn_relinearize_workspace_bytes(plan.h, L, batch))
        capi.check(plan.lib.troyn_bgv_relinearize(self.h, L, _ptr(ct3), plan._key_ptrs(keys, L), _ptr(out), C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return out


class Ring2k:
    """troyn_ring2k: PolynomialEncoderRNSHelper<T> (src/app/bfv_ring2k.cu) of the level with the plan's first L primes; elements are
    numpy arrays of uint32 / uint64, or uint64 pairs (low, high) for 128-bit elements."""

    def __init__(self, plan, L, t_bits, elem_bits):
        self.plan, self.L, self.t_bits, self.elem_bytes = plan, int(L), int(t_bits), elem_bits // 8
        h = C.c_void_p()
        capi.check(plan.lib.troyn_ring2k_create(C.byref(h), plan.h, self.L, self.t_bits, self.elem_bytes))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.plan.lib.troyn_ring2k_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def gamma(self):
        return int(self.plan.lib.troyn_ring2k_gamma(self.h))

    def _elements(self, values):
        """list of Python ints -> device byte buffer"""
        raw = b"".join(int(v).to_bytes(self.elem_bytes, "little") for v in values) or b"\0" * 16
        host = np.frombuffer(raw + b"\0" * (-len(raw) % 8), dtype=np.int64).copy()
        return torch.from_numpy(host).to(self.plan.device)

    def _encode(self, fn, values):
        buf = self._elements(values)
        out = torch.empty((self.L, self.plan.n), dtype=torch.int64, device=self.plan.device)
        capi.check(fn(self.h, _ptr(buf), len(values), _ptr(out), _stream()))
        return out

    def scale_up(self, values):
        return self._encode(self.plan.lib.troyn_ring2k_scale_up, values)

    def centralize(self, values):
        return self._encode(self.plan.lib.troyn_ring2k_centralize, values)

    def scale_down(self, phase):
        """phase [L][N] coefficient form -> list of N Python ints"""
        words = (self.plan.n * self.elem_bytes + 7) // 8
        out = torch.empty(words, dtype=torch.int64, device=self.plan.device)
        capi.check(self.plan.lib.troyn_ring2k_scale_down(self.h, _ptr(phase), _ptr(out), _stream()))
        raw = out.cpu().numpy().tobytes()
        return [int.from_bytes(raw[i * self.elem_bytes:(i + 1) * self.elem_bytes], "little") for i in range(self.plan.n)]

    def decentralize(self, plain, correction_factor=1):
        """plain [L][N] coefficient form (x mod Q) -> list of N Python ints x * correction_factor^-1 mod 2^k (bfv_ring2k.cu:872-925)"""
        words = (self.plan.n * self.elem_bytes + 7) // 8
        out = torch.empty(words, dtype=torch.int64, device=self.plan.device)
        cf = int(correction_factor)
        capi.check(self.plan.lib.troyn_ring2k_decentralize(self.h, _ptr(plain), _ptr(out), cf & ((1 << 64) - 1), (cf >> 64) & ((1 << 64) - 1), _stream()))
        raw = out.cpu().numpy().tobytes()
        return [int.from_bytes(raw[i * self.elem_bytes:(i + 1) * self.elem_bytes], "little") for i in range(self.plan.n)]


class Behz:
    """troyn_behz: BEHZ constants (RNSTool, utils/rns_tool.cu:29-275) for level L and plain modulus t."""

    def __init__(self, plan, L, plain_modulus):
        self.plan, self.L, self.t = plan, int(L), int(plain_modulus)
        h = C.c_void_p()
        capi.check(plan.lib.troyn_behz_create(C.byref(h), plan.h, self.L, self.t))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.plan.lib.troyn_behz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def base_Bsk(self):
        n = self.plan.lib.troyn_behz_base_Bsk_size(self.h)
        out = np.zeros(n, dtype=np.uint64)
        capi.check(self.plan.lib.troyn_behz_get_base_Bsk(self.h, out.ctypes.data_as(capi.p64)))
        return [int(x) for x in out]

    @property
    def working_base_size(self):
        """primes of the auxiliary base the multiply works in (more, smaller primes than base_Bsk when every q_i is below 2^50)"""
        return int(self.plan.lib.troyn_behz_working_base_size(self.h))

    @property
    def gamma(self):
        return int(self.plan.lib.troyn_behz_gamma(self.h))

    def scale_up(self, plain, src=None, subtract=False, out=None):
        """plain [batch][N] mod t -> out [batch][L][N] = (src or 0) +/- round(q/t * plain)   (scaling_variant::scale_up)"""
        n, L = self.plan.n, self.L
        batch = plain.numel() // n
        if out is None:
            out = torch.empty((batch, L, n), dtype=torch.int64, device=plain.device)
        capi.check(self.plan.lib.troyn_bfv_scale_up(self.h, _ptr(plain), n, n, _ptr(src) if src is not None else None, L * n,
                                                    _ptr(out), L * n, int(subtract), batch, _stream()))
        return out

    def decrypt_scale_and_round(self, phase, out=None):
        """phase [batch][L][N] -> [batch][N] mod t   (RNSTool::decrypt_scale_and_round)"""
        n, L = self.plan.n, self.L
        batch = phase.numel() // (L * n)
        if out is None:
            out = torch.empty((batch, n), dtype=torch.int64, device=phase.device)
        capi.check(self.plan.lib.troyn_bfv_decrypt_scale_and_round(self.h, _ptr(phase), _ptr(out), batch, _stream()))
        return out

    def multiply(self, a, pa, b, pb, out=None):
        """a [batch][pa][L][N] x b [batch][pb][L][N] (coefficient form) -> [batch][pa+pb-1][L][N]"""
        n, L = self.plan.n, self.L
        batch = a.numel() // (pa * L * n)
        if out is None:
            out = torch.empty((batch, pa + pb - 1, L, n), dtype=torch.int64, device=a.device)
        nbytes = self.plan.lib.troyn_bfv_multiply_workspace_bytes(self.h, pa, pb, batch)
        ws = self.plan.workspace(nbytes)
        capi.check(self.plan.lib.troyn_bfv_multiply(self.h, _ptr(a), pa, _ptr(b), pb, _ptr(out),
                                                    C.c_void_p(ws.data_ptr()), ws.numel(), batch, _stream()))
        return out
