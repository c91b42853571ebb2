// behz2_kernels.hpp -- second-generation BEHZ base conversions (BFV multiply steps (1)-(2) and (6)-(8)).
//
// Same value semantics as behz_kernels.hpp (fgk/rns_tool.cu:7-100 kernel_fast_b_conv_m_tilde_sm_mrq, :147-286
// kernel_fast_floor_fast_b_conv_sk; host functions utils/rns_tool.cu:762-790, :870-905, :973-988, :1083-1094 and
// utils/rns_base.cu:350-380), rebuilt around the integer multiplier of gfx950:
//
//   * carry-free dot products.  A base conversion is  sum_i y_i * M_ib  mod p_b.  y_i and M_ib are split once into two 32-bit
//     words each (y at bit SA, M at bit SB) such that the four partial products stay below 2^(64 - log2 GROUP); each term then costs
//     exactly four v_mad_u64_u32 into four independent 64-bit accumulators -- no carries, no 128-bit bookkeeping -- and the four
//     sums are recombined with three shifted additions once per output (once per GROUP terms when the moduli are too wide).
//   * every scalar multiplication around a conversion is folded into its matrix on the host: all values are canonical residues of
//     the same modulus, so  ((x*a mod p) + (sum mod p)) * c mod p  ==  (x*(a c) + sum_i y_i (M_i c)) mod p  bit for bit.  The lift
//     costs one reduction per output (was Barrett-128 + Barrett-64 + two Shoup multiplies), the floor likewise (was Barrett-128 + two
//     Shoup multiplies, then another Shoup multiply by (B/p_b)^-1 before the second conversion), the Shenoy-Kumaresan correction
//     alpha_sk * (-B) enters the last conversion as one more term.
//   * the 61-bit auxiliary primes reduce with a one-multiply Barrett (quotient off by at most two, result canonical).
//   * the floor's second conversion is transposed: each converted residue y'_b is folded into all L output accumulators as soon as
//     it exists, so the only loop with a runtime index is the loop over rows and every register array is indexed statically.
//
// The constant tables are built by behz2_build_tables() below (host) from the moduli alone.
#pragma once
#include "dev_math.hpp"
#include "behz_kernels.hpp"

#include <cstring>
#include <vector>
#include "host_math.hpp"
#include "dev_math_f64.hpp"

namespace troyn {

typedef unsigned int u32;
typedef const u32 __attribute__((address_space(4)))* cu32p;
template <typename T> __device__ __forceinline__ cu32p as_c32(const T* p) { return (cu32p)(unsigned long long)p; }

constexpr unsigned BEHZ2_MAX_L = 16;
constexpr unsigned BEHZ2_RC = 8;   // u64 words of per-row constants
#ifndef BEHZ2_FUSED_THREADS
#define BEHZ2_FUSED_THREADS 256
#endif
constexpr unsigned BEHZ2_FUSED_MAX_ROWS = 31;   // behz2_lift_pass1.hpp: rows of both bases in LDS, 2 KB each (64 KB of dynamic LDS without an attribute)

// row constants (u64 index)
enum { B2_P = 0, B2_RLO = 1, B2_RHI = 2, B2_MU = 3, B2_C0 = 4, B2_C1 = 5, B2_W32 = 6, B2_W64 = 7 };

struct Behz2Dev {
    unsigned L, n, rs;                 // rs: u32 stride of one split row = 2 * Lp (Lp = L rounded up to even)
    unsigned NB;                       // primes of the auxiliary base B (rows 0 .. NB-1; row NB = m_sk).  NB == L for the reference's base of
                                       // 61-bit primes; larger for the base of primes below 2^50 (troyn_behz_create), "L+1" below then reads "NB+1"
    const DevModulus* q_mods;          // [L]
    const ulonglong2* q_mt_inv_punc;   // [L] m_tilde (q/q_i)^-1 mod q_i (Shoup): lift input scaling
    const ulonglong2* q_t_inv_punc;    // [L] t (q/q_i)^-1 mod q_i (Shoup): floor input scaling
    const u32* lift_mt;                // [L]  (q/q_i) * (-q^-1) mod 2^32
    const u32* lift_rows;              // [L+1][lo Lp | hi Lp]  ((q/q_i) m_tilde^-1 mod p_b), split at bit 32
    const u64* lift_rc;                // [L+1][RC]: p, ratio_lo, ratio_hi, mu, c = q m_tilde^-1 mod p, cneg = (p - m_tilde) c mod p
    const u32* fa_rows;                // [L+1][lo Lp | hi Lp]  (-(q/q_i) q^-1 (B/p_b)^-1 mod p_b; row L = m_sk without the last factor)
    const u64* fa_rc;                  // [L+1][RC]: p, ratio_lo, ratio_hi, mu, t q^-1 (B/p_b)^-1 mod p, (B/p_b) B^-1 mod m_sk (row L: -B^-1 mod m_sk)
    const u32* fb_cols;                // [L][lo Lp | hi Lp]  ((B/p_b) mod q_j), j along the row, split at bit SHQ
    const u64* fb_rc;                  // [L][RC]: q_j, ratio_lo, ratio_hi, -, B mod q_j, -B mod q_j
};

// ---- host: table construction -----------------------------------------------------------------------------------------------------
struct Behz2Offsets { size_t lift_mt, lift_rows, lift_rc, fa_rows, fa_rc, fb_cols, fb_rc; unsigned rs; };

// q: base q (all below 2^60), B: auxiliary base (|B| == |q| for wide q), m_sk, t.  smallq: every q_i < 2^50 (split point 25, else 30).
// Appends to blob (u64 words); returns false when some inverse does not exist.
inline bool behz2_build_tables(const std::vector<u64>& q, const std::vector<u64>& B, u64 m_sk, u64 t, bool smallq,
                               std::vector<u64>& blob, Behz2Offsets& o) {
    using namespace host;
    const size_t L = q.size(), NB = B.size();
    if (NB == 0 || NB > 32 || (!smallq && NB != L) || L == 0 || L > BEHZ2_MAX_L) return false;
    const size_t Lp = (L + 1) & ~(size_t)1;
    const unsigned shq = smallq ? 25 : 30;
    const u64 mt = (u64)1 << 32;
    o.rs = (unsigned)(2 * Lp);
    std::vector<u64> bsk = B; bsk.push_back(m_sk);
    auto inv_or_fail = [](u64 a, u64 m, u64& out) { return invmod(a % m, m, out); };
    auto push_row = [&](const std::vector<u64>& vals, unsigned split) {   // [lo Lp | hi Lp] as u32 pairs packed into u64 words
        std::vector<u32> w(2 * Lp, 0);
        for (size_t i = 0; i < vals.size(); i++) {
            w[i] = (u32)(vals[i] & (((u64)1 << split) - 1));
            w[Lp + i] = (u32)(vals[i] >> split);
        }
        for (size_t k = 0; k < 2 * Lp; k += 2) blob.push_back((u64)w[k] | ((u64)w[k + 1] << 32));
    };
    auto push_rc = [&](u64 p, u64 c0, u64 c1) {
        BarrettRatio r = barrett_ratio(p);
        const bool small = p < ((u64)1 << 50);
        // slot 3: one-multiply Barrett constant (61-bit moduli) or, for p < 2^50, fl(1/p) of the FP64 reduction (behz2_reduce_f64)
        u64 mu = (p >> 60) == 1 ? (u64)((((u128)1) << 124) / p) : 0, w32 = 0, w64 = 0;
        auto dbits = [](double d) { u64 b; std::memcpy(&b, &d, 8); return b; };
        if (small) {
            mu = dbits(1.0 / (double)p);
            w32 = dbits((double)((((u128)1) << 32) % p));
            w64 = dbits((double)((((u128)1) << 64) % p));
        }
        blob.push_back(p); blob.push_back(r.lo); blob.push_back(r.hi); blob.push_back(mu);
        blob.push_back(c0); blob.push_back(c1); blob.push_back(w32); blob.push_back(w64);
    };
    // ---- lift ----
    u64 neg_inv_q_mt;
    {
        u64 inv;
        if (!inv_or_fail(product_mod(q, SIZE_MAX, mt), mt, inv)) return false;
        neg_inv_q_mt = (mt - inv) % mt;
    }
    o.lift_mt = blob.size();
    {
        std::vector<u32> w(Lp, 0);
        for (size_t i = 0; i < L; i++) w[i] = (u32)((product_mod(q, i, mt) * neg_inv_q_mt) & 0xffffffffull);
        for (size_t k = 0; k < Lp; k += 2) blob.push_back((u64)w[k] | ((u64)w[k + 1] << 32));
    }
    std::vector<u64> inv_mt(NB + 1), inv_q(NB + 1);
    for (size_t b = 0; b <= NB; b++) {
        if (!inv_or_fail(mt, bsk[b], inv_mt[b])) return false;
        if (!inv_or_fail(product_mod(q, SIZE_MAX, bsk[b]), bsk[b], inv_q[b])) return false;
    }
    o.lift_rows = blob.size();
    for (size_t b = 0; b <= NB; b++) {
        std::vector<u64> row(L);
        for (size_t i = 0; i < L; i++) row[i] = mulmod(product_mod(q, i, bsk[b]), inv_mt[b], bsk[b]);
        push_row(row, 32);
    }
    o.lift_rc = blob.size();
    for (size_t b = 0; b <= NB; b++) {
        const u64 p = bsk[b];
        const u64 c = mulmod(product_mod(q, SIZE_MAX, p), inv_mt[b], p);
        push_rc(p, c, mulmod((p - mt % p) % p, c, p));
    }
    // ---- floor, first conversion (q -> Bsk) with the division by q and, for the B rows, the scaling of the second conversion ----
    std::vector<u64> B_inv_punc(NB, 1);
    for (size_t b = 0; b < NB; b++)
        if (NB > 1 && !inv_or_fail(product_mod(B, b, B[b]), B[b], B_inv_punc[b])) return false;
    u64 inv_B_msk;
    if (!inv_or_fail(product_mod(B, SIZE_MAX, m_sk), m_sk, inv_B_msk)) return false;
    o.fa_rows = blob.size();
    for (size_t b = 0; b <= NB; b++) {
        const u64 p = bsk[b];
        std::vector<u64> row(L);
        for (size_t i = 0; i < L; i++) {
            u64 v = mulmod(product_mod(q, i, p), inv_q[b], p);
            if (b < NB) v = mulmod(v, B_inv_punc[b], p);
            row[i] = (p - v) % p;
        }
        push_row(row, 32);
    }
    o.fa_rc = blob.size();
    for (size_t b = 0; b <= NB; b++) {
        const u64 p = bsk[b];
        u64 tq = mulmod(t % p, inv_q[b], p);
        if (b < NB) tq = mulmod(tq, B_inv_punc[b], p);
        const u64 mk = b < NB ? mulmod(product_mod(B, b, m_sk), inv_B_msk, m_sk) : (m_sk - inv_B_msk) % m_sk;
        push_rc(p, tq, mk);
    }
    // ---- floor, second conversion (B -> q), one column block per B prime ----
    o.fb_cols = blob.size();
    for (size_t b = 0; b < NB; b++) {
        std::vector<u64> col(L);
        for (size_t j = 0; j < L; j++) col[j] = product_mod(B, b, q[j]);
        push_row(col, shq);
    }
    o.fb_rc = blob.size();
    for (size_t j = 0; j < L; j++) {
        const u64 pb = product_mod(B, SIZE_MAX, q[j]);
        push_rc(q[j], pb, (q[j] - pb) % q[j]);
    }
    return true;
}

// ---- device --------------------------------------------------------------------------------------------------------------------------

// p in [2^60, 2^61), v < 2^124, mu = floor(2^124 / p): floor((v >> 60) mu / 2^64) is the quotient or up to two below it
__device__ __forceinline__ u64 behz2_reduce61(u128 v, u64 p, u64 mu) {
    const u64 qh = mul_hi((u64)(v >> 60), mu);
    u64 r = (u64)v - qh * p;
    r = r >= 2 * p ? r - 2 * p : r;
    return r >= p ? r - p : r;
}

__device__ __forceinline__ u64 behz2_reduce(u128 v, u64 p, u64 rlo, u64 rhi) { return barrett128((u64)v, (u64)(v >> 64), p, rlo, rhi); }

// p < 2^50, v < 2^106: v = (x3 2^32 + x2) 2^64 + x1 2^32 + x0 in 32-bit words; the two upper parts are multiplied by (2^64 mod p) and (2^32 mod p) with
// the exact FP64 modular product (f64_mulq: the quotient comes from the rounded product, error < 2^-10 here), every value an integer below 2^53.
// ~24 instructions against ~50 of the Barrett-128 form; returns the re-centred residue (|r| <= p/2 + 1) as a double.
__device__ __forceinline__ double behz2_reduce_f64(u128 v, cu64p rc) {
    const double p = f64_from_u64(rc[B2_P]), inv_p = f64_bits_to_double(rc[B2_MU]);
    const double w32 = f64_bits_to_double(rc[B2_W32]), w64 = f64_bits_to_double(rc[B2_W64]);
    const u64 lo = (u64)v, hi = (u64)(v >> 64);
    const double xh = __builtin_fma((double)(u32)(hi >> 32), 4294967296.0, (double)(u32)hi);
    const double r = f64_mulq(xh, w64, inv_p, p) + f64_mulq((double)(u32)(lo >> 32), w32, inv_p, p) + (double)(u32)lo;
    return f64_corr(r, F64Mod{p, inv_p});
}
__device__ __forceinline__ u64 behz2_canon_f64(double r, cu64p rc) { return f64_to_u64(r < 0.0 ? r + f64_from_u64(rc[B2_P]) : r); }

// reduction modulo an auxiliary prime: FAST61 = the one-multiply form for primes in [2^60, 2^61) (the reference's base under small q)
template <bool FAST61>
__device__ __forceinline__ u64 behz2_reduce_aux(u128 v, cu64p rc) {
    if (FAST61) return behz2_reduce61(v, rc[B2_P], rc[B2_MU]);
    return behz2_reduce(v, rc[B2_P], rc[B2_RLO], rc[B2_RHI]);
}

// four carry-free partial sums of  sum_i a_i * b_i  (a split at SA in VGPRs, b split at SB through the scalar cache)
struct Behz2Acc { u64 ll, lh, hl, hh; };
__device__ __forceinline__ void behz2_zero(Behz2Acc& a) { a.ll = a.lh = a.hl = a.hh = 0; }
__device__ __forceinline__ void behz2_mac(Behz2Acc& a, u32 xlo, u32 xhi, u32 blo, u32 bhi) {
    a.ll += (u64)xlo * blo;
    a.lh += (u64)xlo * bhi;
    a.hl += (u64)xhi * blo;
    a.hh += (u64)xhi * bhi;
}
template <int SA, int SB>
__device__ __forceinline__ u128 behz2_combine(const Behz2Acc& a) {
    return (u128)a.ll + ((u128)a.lh << SB) + ((u128)a.hl << SA) + ((u128)a.hh << (SA + SB));
}

template <int L, int SA, int SB, int GROUP>
__device__ __forceinline__ u128 behz2_dot(const u32 (&xlo)[L], const u32 (&xhi)[L], cu32p blo, cu32p bhi) {
    u128 v = 0;
#pragma unroll
    for (int g = 0; g < L; g += GROUP) {
        Behz2Acc a; behz2_zero(a);
#pragma unroll
        for (int i = g; i < g + GROUP && i < L; ++i) behz2_mac(a, xlo[i], xhi[i], blo[i], bhi[i]);
        v += behz2_combine<SA, SB>(a);
    }
    return v;
}

// scaled input residues of base q from any source (load(i) = canonical residue i of this coefficient)
template <int L, int SHQ, class LD>
__device__ __forceinline__ void behz2_scale_q(LD&& load, cu64x2p scale, cmodp q_mods, u32 (&ylo)[L], u32 (&yhi)[L]) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const ulonglong2 f = ld_pair(scale, i);
        const u64 y = shoup_mul(load(i), f.x, f.y, q_mods[i].q);
        ylo[i] = (u32)y & ((1u << SHQ) - 1);
        yhi[i] = (u32)(y >> SHQ);
    }
}

// BEHZ steps (1)-(2) at one coefficient: load(i) = residue of base q, store(b, word) = lifted residue of row b (0 .. NB)
template <int L, bool SMALLQ, bool AUX50, class LD, class ST>
__device__ __forceinline__ void behz2_lift_one(const Behz2Dev& c, LD&& load, ST&& store) {
    constexpr int SHQ = SMALLQ ? 25 : 30, GROUP = SMALLQ ? 64 : 4;
    const cu32p mtrow = as_c32(c.lift_mt), rows = as_c32(c.lift_rows);
    const cu64p rcs = as_c64(c.lift_rc);
    const unsigned Lp = c.rs >> 1, NB = c.NB;
    u32 ylo[L], yhi[L];
    behz2_scale_q<L, SHQ>(load, as_c128(c.q_mt_inv_punc), as_cmod(c.q_mods), ylo, yhi);
    // q -> {m_tilde = 2^32} and the multiplication by -q^-1: everything modulo 2^32
    u32 r_mt = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) r_mt += (ylo[i] | (yhi[i] << SHQ)) * mtrow[i];
    const bool neg = r_mt >= 0x80000000u;
#pragma unroll 1
    for (unsigned b = 0; b <= NB; ++b) {
        const cu32p row = rows + (size_t)b * c.rs;
        const cu64p rc = rcs + (size_t)b * BEHZ2_RC;
        u128 v = behz2_dot<L, SHQ, 32, GROUP>(ylo, yhi, row, row + Lp);
        v += (u128)rc[B2_C0] * r_mt + (neg ? rc[B2_C1] : 0ull);
        if constexpr (AUX50) store(b, behz2_reduce_f64(v, rc));      // the re-centred residue as a double (the caller canonicalises or keeps it)
        else store(b, behz2_reduce_aux<SMALLQ>(v, rc));
    }
}

// BEHZ steps (1)-(2): in [items][L][N] (coefficient form, base q) -> out [items][NB+1][N]
// AUX50: the auxiliary primes are below 2^50 (NB > L of them) instead of the reference's 61-bit primes
template <int L, bool SMALLQ, bool AUX50 = false>
__global__ __launch_bounds__(256) void behz2_lift_kernel(unsigned chunks, Behz2Dev c, const u64* in, u64* out) {
    static_assert(!AUX50 || SMALLQ, "the 50-bit auxiliary base is chosen for small q only");
    const unsigned n = c.n;
    const size_t item = blockIdx.x / chunks;
    const u64* ip = in + item * (size_t)L * n;
    u64* op = out + item * (size_t)(c.NB + 1) * n;
#pragma unroll 1
    for (unsigned x = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; x < n; x += chunks * blockDim.x)
        behz2_lift_one<L, SMALLQ, AUX50>(c, [&](int i) { return ip[(size_t)i * n + x]; }, [&](unsigned b, auto w) {
                if constexpr (AUX50) op[(size_t)b * n + x] = behz2_canon_f64(w, as_c64(c.lift_rc) + (size_t)b * BEHZ2_RC); else op[(size_t)b * n + x] = w;
            });
}

// BEHZ steps (6)-(8) at one coefficient: load_q(i) / load_b(b) = residues of the product in base q / Bsk (b = NB: m_sk), store(j, word)
template <int L, bool SMALLQ, bool AUX50, class LQ, class LB, class ST>
__device__ __forceinline__ void behz2_floor_one(const Behz2Dev& c, LQ&& load_q, LB&& load_b, ST&& store) {
    constexpr int SHQ = SMALLQ ? 25 : 30, GROUP = SMALLQ ? 64 : 4;
    const unsigned NB = c.NB;
    const cu32p fa_rows = as_c32(c.fa_rows), fb_cols = as_c32(c.fb_cols);
    const cu64p fa_rc = as_c64(c.fa_rc), fb_rc = as_c64(c.fb_rc);
    const unsigned Lp = c.rs >> 1;
    u32 ylo[L], yhi[L];
    behz2_scale_q<L, SHQ>(load_q, as_c128(c.q_t_inv_punc), as_cmod(c.q_mods), ylo, yhi);
    Behz2Acc acc[L];
    u128 wide[L];           // only live when the partial sums must be folded every GROUP rows
#pragma unroll
    for (int j = 0; j < L; ++j) { behz2_zero(acc[j]); wide[j] = 0; }
    u64 sk_lo = 0, sk_hi = 0;   // sum_b y'_b * ((B/p_b) B^-1 mod m_sk)
#pragma unroll 1
    for (unsigned b = 0; b < NB; ++b) {
        const cu32p row = fa_rows + (size_t)b * c.rs;
        const cu64p rc = fa_rc + (size_t)b * BEHZ2_RC;
        // ((x_b t - conv_b) q^-1) (B/p_b)^-1 mod p_b in one dot product
        u128 v = behz2_dot<L, SHQ, 32, GROUP>(ylo, yhi, row, row + Lp);
        const u64 xb = load_b(b), tq = rc[B2_C0];
        v += (u128)xb * tq;
        u64 yb;
        if constexpr (AUX50) yb = behz2_canon_f64(behz2_reduce_f64(v, rc), rc); else yb = behz2_reduce_aux<SMALLQ>(v, rc);
        mac128(sk_lo, sk_hi, yb, rc[B2_C1]);
        const u32 zlo = (u32)yb, zhi = (u32)(yb >> 32);
        const cu32p col = fb_cols + (size_t)b * c.rs;
#pragma unroll
        for (int j = 0; j < L; ++j) behz2_mac(acc[j], zlo, zhi, col[j], col[Lp + j]);
        if (GROUP < L && (b % GROUP) == GROUP - 1) {
#pragma unroll
            for (int j = 0; j < L; ++j) { wide[j] += behz2_combine<32, SHQ>(acc[j]); behz2_zero(acc[j]); }
        }
    }
    // m_sk row: r_sk = (x_sk t - conv_sk) q^-1 mod m_sk, then alpha_sk = (sum_b y'_b (B/p_b) - r_sk) B^-1 mod m_sk
    u64 alpha_use; bool neg;
    {
        const cu32p row = fa_rows + (size_t)NB * c.rs;
        const cu64p rc = fa_rc + (size_t)NB * BEHZ2_RC;
        u128 v = behz2_dot<L, SHQ, 32, GROUP>(ylo, yhi, row, row + Lp);
        v += (u128)load_b(NB) * rc[B2_C0];
        const u64 msk = rc[B2_P];
        u64 r_sk;
        if constexpr (AUX50) r_sk = behz2_canon_f64(behz2_reduce_f64(v, rc), rc); else r_sk = behz2_reduce_aux<SMALLQ>(v, rc);
        mac128(sk_lo, sk_hi, r_sk, rc[B2_C1]);
        const u64 alpha_sk = barrett128(sk_lo, sk_hi, msk, rc[B2_RLO], rc[B2_RHI]);
        neg = alpha_sk > (msk >> 1);
        alpha_use = neg ? msk - alpha_sk : alpha_sk;
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const cu64p rc = fb_rc + (size_t)j * BEHZ2_RC;
        u128 v = wide[j] + behz2_combine<32, SHQ>(acc[j]);
        v += (u128)alpha_use * (neg ? rc[B2_C0] : rc[B2_C1]);
        if constexpr (AUX50) store(j, behz2_canon_f64(behz2_reduce_f64(v, rc), rc));      // AUX50 implies every q_j below 2^50
        else store(j, behz2_reduce(v, rc[B2_P], rc[B2_RLO], rc[B2_RHI]));
    }
}

// BEHZ steps (6)-(8): in_q [items][L][N], in_bsk [items][NB+1][N] (coefficient form) -> out [items][L][N]
template <int L, bool SMALLQ, bool AUX50 = false>
__global__ __launch_bounds__(256) void behz2_floor_kernel(unsigned chunks, Behz2Dev c, const u64* in_q, const u64* in_bsk, u64* out) {
    static_assert(!AUX50 || SMALLQ, "the 50-bit auxiliary base is chosen for small q only");
    const unsigned n = c.n;
    const size_t item = blockIdx.x / chunks;
    const u64* qp = in_q + item * (size_t)L * n;
    const u64* bp = in_bsk + item * (size_t)(c.NB + 1) * n;
    u64* op = out + item * (size_t)L * n;
#pragma unroll 1
    for (unsigned x = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; x < n; x += chunks * blockDim.x)
        behz2_floor_one<L, SMALLQ, AUX50>(c, [&](int i) { return qp[(size_t)i * n + x]; }, [&](unsigned b) { return bp[(size_t)b * n + x]; },
                                          [&](int j, u64 w) { op[(size_t)j * n + x] = w; });
}

}  // namespace troyn
