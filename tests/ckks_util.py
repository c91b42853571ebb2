"""Test-side CKKS helpers (numpy): canonical-embedding encode / decode and a symmetric encryption built from the
oracle's RLWE samples.  Test infrastructure only -- approximate arithmetic, used to check that the headline pipeline
(multiply -> relinearize -> rescale) DECRYPTS to the slot-wise product."""
import numpy as np


def encode(values, n, scale):
    """values: N/2 complex slots at the evaluation points zeta^(2k+1), k < N/2 (zeta = exp(i*pi/N)); the other half are
    their conjugates.  Returns integer coefficients (int64) of round(scale * m(X))."""
    v = np.concatenate([values, np.conj(values[::-1])])
    a = np.fft.fft(v) / n
    zeta_inv = np.exp(-1j * np.pi * np.arange(n) / n)
    return np.rint(scale * np.real(a * zeta_inv)).astype(np.int64)


def decode(coeffs, n, scale):
    zeta = np.exp(1j * np.pi * np.arange(n) / n)
    v = n * np.fft.ifft(np.asarray(coeffs, dtype=np.float64) * zeta)
    return v[: n // 2] / scale


def to_rns_ntt(ctx, coeffs, L):
    """signed integer coefficients -> [L][N] residues in NTT form"""
    out = np.empty((L, ctx.n), dtype=np.uint64)
    for l in range(L):
        q = ctx.q[l]
        out[l] = np.mod(coeffs, q).astype(np.uint64)
    return ctx.to_ntt(out[None], 1, L)[0]


def encrypt(ctx, rng, sk, coeffs, L):
    """(b + NTT(m), a) with (b, a) a fresh RLWE sample b = -(a s + e) under the first L moduli"""
    zero = ctx.public_key(rng, sk)                       # rlwe::symmetric at the key level, NTT form
    ct = np.ascontiguousarray(zero[:, :L, :]).copy()
    m = to_rns_ntt(ctx, coeffs, L)
    for l in range(L):
        ct[0, l] = (ct[0, l] + m[l]) % np.uint64(ctx.q[l])
    return ct


def decrypt_limb0(ctx, sk, ct):
    """c0 + c1*s (+ c2*s^2) in limb 0, coefficient form, centred: valid while |m| < q_0 / 2"""
    L = ct.shape[1]
    q0 = ctx.q[0]
    mods = ctx.moduli()
    import ctypes as C
    lib = __import__("oracle.oracle", fromlist=["lib"]).lib()
    ptr = __import__("oracle.oracle", fromlist=["ptr"]).ptr
    s = np.ascontiguousarray(sk[0])
    acc = np.ascontiguousarray(ct[0, 0]).copy()
    spow = s.copy()
    tmp = np.empty_like(acc)
    for p in range(1, ct.shape[0]):
        lib.orc_dyadic_product_ps(ptr(np.ascontiguousarray(ct[p, 0])), ptr(spow), 1, ctx.n, mods, 1, ptr(tmp))
        acc = (acc + tmp) % np.uint64(q0)
        nxt = np.empty_like(spow)
        lib.orc_dyadic_product_ps(ptr(spow), ptr(s), 1, ctx.n, mods, 1, ptr(nxt))
        spow = nxt
    coeff = ctx.from_ntt(acc[None, None], 1, 1)[0, 0].astype(np.int64)
    return np.where(coeff > q0 // 2, coeff - q0, coeff)
