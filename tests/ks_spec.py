"""Exact big-integer SPECIFICATION of the hybrid key switch (test infrastructure, not a restatement).

What `Evaluator::switch_key_internal` (/root/reference/src/evaluator_keyswitching_core.cu:757-1052) must compute, written
as mathematics on Python integers -- no Barrett / Shoup / lazy ranges, no butterflies, no CRT:

  * decomposition digits d_j = the target's limb j in coefficient form, as integers in [0, q_j)        (:817-821, :847-858)
  * for every key modulus m in {q_0 .. q_{L-1}, q_special} and component c in {0, 1}
        X_c[m] = sum_j  d_j (*) k_j[c][m]      (negacyclic product in Z_m[x]/(x^N + 1))                (:833-919)
  * s_c = X_c[q_special] in [0, q_special);  r_c = ((s_c + h) mod q_special) - h  with h = floor(q_special / 2): the
    representative of s_c in [-h, q_special - 1 - h]                                                     (:583-597)
  * result_c[j] = (X_c[q_j] - r_c) * q_special^-1  mod q_j                                               (:641-656)
  * AddInplace: dest += result;  Overwrite: dest = result;  OverwriteExceptFirst: dest[0] += result[0], dest[1] = result[1]

The reference stores keys and (CKKS) operands in NTT form; the transform is evaluation at the odd powers of the minimal
primitive 2N-th root psi in bit-reversed order (utils/ntt.cu:14-76): NTT(x)[i] = x(psi^(2 bitrev(i) + 1)).  `ntt_by_definition`
evaluates exactly that, O(N^2), and `minimal_primitive_root` finds psi by exhaustive search.

Negacyclic products use Kronecker substitution (one big-integer multiplication), which is exact.
"""


def bitrev(i, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def minimal_primitive_root(two_n, q):
    """smallest primitive two_n-th root of unity mod prime q (utils/number_theory.cu:41-87 finds the same value)"""
    assert (q - 1) % two_n == 0
    g = 2
    while True:
        r = pow(g, (q - 1) // two_n, q)
        if pow(r, two_n // 2, q) == q - 1:
            break
        g += 1
    best, cur, r2 = r, r, r * r % q
    for _ in range(two_n // 2 - 1):       # every primitive root is an odd power of r
        cur = cur * r2 % q
        best = min(best, cur)
    return best


def ntt_by_definition(x, q, psi=None):
    n = len(x)
    logn = n.bit_length() - 1
    psi = psi or minimal_primitive_root(2 * n, q)
    out = [0] * n
    for i in range(n):
        w = pow(psi, 2 * bitrev(i, logn) + 1, q)
        acc, p = 0, 1
        for c in x:
            acc += int(c) * p
            p = p * w % q
        out[i] = acc % q
    return out


def negacyclic(a, b, m):
    """a (*) b in Z_m[x]/(x^N + 1); operands any non-negative integers.  Kronecker substitution: both polynomials are
    packed into one integer each with slots wide enough for a whole coefficient of the product, multiplied once, unpacked."""
    n = len(a)
    bits = max(max(int(v) for v in a).bit_length(), 1) + max(max(int(v) for v in b).bit_length(), 1) + n.bit_length() + 1
    wb = (bits + 7) // 8
    pa = int.from_bytes(b"".join(int(v).to_bytes(wb, "little") for v in a), "little")
    pb = int.from_bytes(b"".join(int(v).to_bytes(wb, "little") for v in b), "little")
    raw = (pa * pb).to_bytes(2 * n * wb, "little")
    out = [0] * n
    for k in range(2 * n - 1):
        v = int.from_bytes(raw[k * wb:(k + 1) * wb], "little")
        if k < n:
            out[k] += v
        else:
            out[k - n] -= v          # x^N = -1
    return [v % m for v in out]


def switch_key_spec(q, L, digits, keys_coeff, dest, assign):
    """q: the K key-level moduli (special prime last); digits[j][i]: coefficient-form target limbs;
    keys_coeff[j][c][k][i]: key j, component c, under modulus q[k] (k = K-1: special), coefficient form;
    dest[c][j][i] canonical; returns the new destination in coefficient form"""
    K = len(q)
    qs = q[K - 1]
    h = qs // 2
    out = [[None] * L for _ in range(2)]
    for c in range(2):
        n = len(digits[0])
        s = [0] * n
        for j in range(L):
            s = [(x + y) % qs for x, y in zip(s, negacyclic(digits[j], keys_coeff[j][c][K - 1], qs))]
        r = [((v + h) % qs) - h for v in s]
        for l in range(L):
            X = [0] * n
            for j in range(L):
                X = [(x + y) % q[l] for x, y in zip(X, negacyclic(digits[j], keys_coeff[j][c][l], q[l]))]
            inv = pow(qs, -1, q[l])
            res = [((x - rr) * inv) % q[l] for x, rr in zip(X, r)]
            add = assign == 0 or (assign == 2 and c == 0)
            out[c][l] = [(int(d) + v) % q[l] for d, v in zip(dest[c][l], res)] if add else res
    return out
