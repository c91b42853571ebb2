// host_math.hpp -- host-side number theory used to build the device tables.
//
// Product code (NOT the oracle): written independently of oracle/ so that table contents
// are cross-checked by two implementations.  It reproduces the *values* the reference
// builds on the host: Barrett ratios (modulus.cu:7-32), Shoup operands
// (utils/uint_small_mod.h:92-122), minimal primitive 2N-th roots
// (utils/number_theory.cu:68-87), NTT tables (utils/ntt.cu:14-76), prime search
// (utils/number_theory.cu:22-39) and the RNSTool constants (utils/rns_tool.cu:29-275).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace troyn { namespace host {

using u64 = unsigned long long;  // same type as the device-side troyn::u64
using u128 = unsigned __int128;

inline u64 mulmod(u64 a, u64 b, u64 m) { return static_cast<u64>((static_cast<u128>(a) * b) % m); }

inline u64 powmod(u64 a, u64 e, u64 m) {
    u64 r = 1 % m;
    a %= m;
    while (e) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return r;
}

// inverse of a modulo m for gcd(a, m) = 1 (m need not be prime: m_tilde = 2^32)
inline bool invmod(u64 a, u64 m, u64& out) {
    a %= m;
    if (a == 0) return false;
    __int128 r0 = m, r1 = a, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 q = r0 / r1;
        __int128 r2 = r0 - q * r1; r0 = r1; r1 = r2;
        __int128 t2 = t0 - q * t1; t0 = t1; t1 = t2;
    }
    if (r0 != 1) return false;
    if (t0 < 0) t0 += m;
    out = static_cast<u64>(t0);
    return true;
}

// deterministic Miller-Rabin, exact for 64-bit inputs
inline bool is_prime(u64 n) {
    if (n < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1; int r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int i = 1; i < r; i++) {
            x = mulmod(x, x, n);
            if (x == n - 1) { witness = false; break; }
        }
        if (witness) return false;
    }
    return true;
}

// `count` largest primes p = 1 (mod factor) with exactly bit_size bits, descending
inline std::vector<u64> get_primes(u64 factor, size_t bit_size, size_t count) {
    std::vector<u64> out;
    u64 value = (((u64)1 << bit_size) - 1) / factor * factor + 1;
    u64 lower = (u64)1 << (bit_size - 1);
    while (out.size() < count && value > lower) {
        if (is_prime(value)) out.push_back(value);
        value -= factor;
    }
    if (out.size() < count) throw std::logic_error("[troyn::get_primes] Failed to find enough qualifying primes.");
    return out;
}

struct Shoup { u64 operand, quotient; };
inline Shoup shoup(u64 w, u64 q) { return Shoup{w, static_cast<u64>((static_cast<u128>(w) << 64) / q)}; }

struct BarrettRatio { u64 lo, hi; };
inline BarrettRatio barrett_ratio(u64 q) {
    // floor(2^128 / q) as two words
    u128 two64 = (u128)1 << 64;
    u128 hi = two64 / q, r = two64 % q;
    u128 lo = (r << 64) / q;
    return BarrettRatio{static_cast<u64>(lo), static_cast<u64>(hi)};
}

// smallest primitive `degree`-th root of unity mod prime q (degree a power of two, degree | q-1)
inline bool minimal_primitive_root(u64 degree, u64 q, u64& out) {
    if ((q - 1) % degree != 0) return false;
    u64 cofactor = (q - 1) / degree, g = 0;
    for (u64 c = 2; c < q && c < 100000; c++) {
        u64 cand = powmod(c, cofactor, q);
        if (powmod(cand, degree / 2, q) == q - 1) { g = cand; break; }
    }
    if (g == 0) return false;
    // all primitive roots are the odd powers of g; take the minimum
    u64 g2 = mulmod(g, g, q), cur = g, best = g;
    for (u64 i = 0; i < degree / 2; i++) {
        if (cur < best) best = cur;
        cur = mulmod(cur, g2, q);
    }
    out = best;
    return true;
}

inline u64 bit_reverse(u64 x, unsigned bits) {
    u64 r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

struct NttTable {
    u64 q = 0, root = 0;
    std::vector<Shoup> fwd;  // fwd[bitrev(i)] = psi^i, fwd[0] = 1
    std::vector<Shoup> inv;  // inv[bitrev(i-1)+1] = psi^-i, inv[0] = 1
    Shoup inv_degree{0, 0};
};

inline NttTable make_ntt_table(unsigned log_n, u64 q, u64 root_hint) {
    NttTable t;
    t.q = q;
    size_t n = (size_t)1 << log_n;
    u64 psi = root_hint;
    if (psi == 0) {
        if (!minimal_primitive_root(2 * (u64)n, q, psi))
            throw std::invalid_argument("[troyn::make_ntt_table] Invalid modulus, unable to find primitive root.");
    } else if (powmod(psi, n, q) != q - 1) {
        throw std::invalid_argument("[troyn::make_ntt_table] Supplied root is not a primitive 2N-th root.");
    }
    t.root = psi;
    u64 psi_inv;
    if (!invmod(psi, q, psi_inv)) throw std::invalid_argument("[troyn::make_ntt_table] Invalid modulus, unable to invert.");
    t.fwd.resize(n); t.inv.resize(n);
    u64 p = 1, pi = 1;
    for (size_t i = 0; i < n; i++) {
        if (i == 0) { t.fwd[0] = shoup(1, q); t.inv[0] = shoup(1, q); }
        else {
            t.fwd[bit_reverse(i, log_n)] = shoup(p, q);
            t.inv[bit_reverse(i - 1, log_n) + 1] = shoup(pi, q);
        }
        p = mulmod(p, psi, q);
        pi = mulmod(pi, psi_inv, q);
        if (i == 0) { /* p = psi^1, pi = psi^-1 now */ }
    }
    u64 ninv;
    if (!invmod((u64)n % q, q, ninv)) throw std::invalid_argument("[troyn::make_ntt_table] Invalid modulus, unable to invert degree.");
    t.inv_degree = shoup(ninv, q);
    return t;
}

// number of significant bits of prod(values)
inline size_t product_bit_count(const std::vector<u64>& values) {
    std::vector<u64> acc{1};
    for (u64 v : values) {
        u64 carry = 0;
        for (auto& w : acc) {
            u128 p = (u128)w * v + carry;
            w = (u64)p; carry = (u64)(p >> 64);
        }
        if (carry) acc.push_back(carry);
    }
    size_t bits = 0; u64 top = acc.back();
    while (top) { bits++; top >>= 1; }
    return bits + 64 * (acc.size() - 1);
}

inline size_t bit_count(u64 v) { size_t b = 0; while (v) { b++; v >>= 1; } return b; }

// prod_{k != except} base[k] mod p  (except = SIZE_MAX: full product)
inline u64 product_mod(const std::vector<u64>& base, size_t except, u64 p) {
    u64 acc = 1 % p;
    for (size_t k = 0; k < base.size(); k++) if (k != except) acc = mulmod(acc, base[k] % p, p);
    return acc;
}

// ---- multi-word helpers for the BFV Delta constants (context_data.cu:226-247: floor(q/t), q mod t) ----
inline std::vector<u64> big_product(const std::vector<u64>& values) {
    std::vector<u64> acc{1};
    for (u64 v : values) {
        u64 carry = 0;
        for (size_t k = 0; k < acc.size(); k++) {
            const u128 p = static_cast<u128>(acc[k]) * v + carry;
            acc[k] = static_cast<u64>(p);
            carry = static_cast<u64>(p >> 64);
        }
        if (carry) acc.push_back(carry);
    }
    return acc;
}
// a <- floor(a / d); returns a mod d
inline u64 big_divmod_small(std::vector<u64>& a, u64 d) {
    u128 rem = 0;
    for (size_t k = a.size(); k-- > 0;) {
        const u128 cur = (rem << 64) | a[k];
        a[k] = static_cast<u64>(cur / d);
        rem = cur % d;
    }
    return static_cast<u64>(rem);
}
inline u64 big_mod_small(const std::vector<u64>& a, u64 d) {
    u128 rem = 0;
    for (size_t k = a.size(); k-- > 0;) rem = ((rem << 64) | a[k]) % d;
    return static_cast<u64>(rem);
}

}}  // namespace troyn::host
