// Decrypt-and-compare matrix through the C++ mirror -- the oracle-independent semantic pin of the hot path.
//
// Same idea and the same parameter sets as the reference's own device tests (test/evaluator.cu: N = 32 with coefficient moduli
// {40,40,40}, {30,30,30,30}, {60,60,60} and {60,40,40,60}; BFV / BGV with a 20- or 35-bit plain modulus, CKKS at scale 2^20 with
// tolerance 1e-2, seed 0x123): random messages are encoded and encrypted, the operation under test runs on the GPU, and the
// decrypted, decoded result must equal the plain computation.  Nothing here consults the oracle: a wrong key switch, rescale or
// tensor product cannot decrypt to the right message.
//
//   semantics_driver <scheme: bfv|bgv|ckks> <log_t> <q bits...>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;
using cd = std::complex<double>;

static size_t failures = 0;
static void report(const char* name, bool ok, double err = 0.0) {
    std::printf("%s %s %.3e\n", name, ok ? "pass" : "FAIL", err);
    failures += !ok;
}

struct Ctx {
    SchemeType scheme; size_t n; uint64_t t = 0; double scale = 0, tol = 0;
    HeContextPointer context;
    std::unique_ptr<BatchEncoder> batch; std::unique_ptr<CKKSEncoder> ckks;
    std::unique_ptr<KeyGenerator> keygen; std::unique_ptr<Encryptor> enc; std::unique_ptr<Decryptor> dec; std::unique_ptr<Evaluator> ev;
    std::mt19937_64 gen{0x123};

    // integer schemes: slot vectors mod t; CKKS: complex slot vectors in [-10, 10]^2 (input_max = 10 in the reference's CKKS contexts)
    std::vector<uint64_t> rand_u() { std::vector<uint64_t> v(n); for (auto& x : v) x = gen() % t; return v; }
    std::vector<cd> rand_c() { std::uniform_real_distribution<double> U(-10.0, 10.0); std::vector<cd> v(n / 2); for (auto& x : v) x = cd(U(gen), U(gen)); return v; }
    Ciphertext encrypt(const std::vector<uint64_t>& m) { return enc->encrypt_asymmetric_new(batch->encode_new(m)); }
    Ciphertext encrypt(const std::vector<cd>& m, double s) { return enc->encrypt_asymmetric_new(ckks->encode_complex64_simd_new(m, std::nullopt, s)); }
    std::vector<uint64_t> open_u(const Ciphertext& c) { return batch->decode_new(dec->decrypt_new(c)); }
    std::vector<cd> open_c(const Ciphertext& c) { return ckks->decode_complex64_simd_new(dec->decrypt_new(c)); }
};

static bool eq(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) { return a == b; }
static double maxerr(const std::vector<cd>& a, const std::vector<cd>& b) { double e = 0; for (size_t i = 0; i < b.size(); i++) e = std::max(e, std::abs(a[i] - b[i])); return e; }

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: semantics_driver scheme log_t bits...\n"); return 2; }
    try {
        Ctx c;
        const std::string sch = argv[1];
        c.scheme = sch == "bfv" ? SchemeType::BFV : sch == "bgv" ? SchemeType::BGV : SchemeType::CKKS;
        const bool ckks = c.scheme == SchemeType::CKKS;
        c.n = 32;
        const size_t log_t = std::strtoul(argv[2], nullptr, 10);
        std::vector<size_t> bits;
        for (int i = 3; i < argc; i++) bits.push_back(std::strtoul(argv[i], nullptr, 10));
        EncryptionParameters params(c.scheme);
        params.set_poly_modulus_degree(c.n);
        params.set_coeff_modulus(CoeffModulus::create(c.n, bits));
        if (!ckks) { params.set_plain_modulus(PlainModulus::batching(c.n, log_t)); c.t = params.plain_modulus().value(); }
        c.scale = 1048576.0; c.tol = 1e-2;
        c.context = HeContext::create(params, true, SecurityLevel::Nil, 0x123);
        c.context->to_device_inplace();
        if (ckks) c.ckks.reset(new CKKSEncoder(c.context)); else { c.batch.reset(new BatchEncoder(c.context)); c.batch->to_device_inplace(); }
        c.keygen.reset(new KeyGenerator(c.context));
        c.enc.reset(new Encryptor(c.context));
        c.enc->set_public_key(c.keygen->create_public_key(false));
        c.dec.reset(new Decryptor(c.context, c.keygen->secret_key()));
        c.ev.reset(new Evaluator(c.context));
        const bool has_ks = bits.size() >= 2;      // a special prime exists
        const size_t levels = bits.size() - 1;     // data limbs at the first level
        const size_t slots = ckks ? c.n / 2 : c.n;
        const size_t row = ckks ? slots : slots / 2;

        if (!ckks) {
            const uint64_t t = c.t;
            auto mulv = [&](const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) { std::vector<uint64_t> r(a.size()); for (size_t i = 0; i < a.size(); i++) r[i] = (uint64_t)((unsigned __int128)a[i] * b[i] % t); return r; };
            auto addv = [&](const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) { std::vector<uint64_t> r(a.size()); for (size_t i = 0; i < a.size(); i++) r[i] = (a[i] + b[i]) % t; return r; };
            auto m1 = c.rand_u(), m2 = c.rand_u(), m3 = c.rand_u();
            Ciphertext e1 = c.encrypt(m1), e2 = c.encrypt(m2), e3 = c.encrypt(m3);
            report("decrypt", eq(c.open_u(e1), m1));
            report("add", eq(c.open_u(c.ev->add_new(e1, e2)), addv(m1, m2)));
            Ciphertext prod = c.ev->multiply_new(e1, e2);                                        // test/evaluator.cu:276-299
            report("multiply", eq(c.open_u(prod), mulv(m1, m2)));
            report("square", eq(c.open_u(c.ev->square_new(e1)), mulv(m1, m1)));                  // :394-421
            report("multiply_plain", eq(c.open_u(c.ev->multiply_plain_new(e1, c.batch->encode_new(m2))), mulv(m1, m2)));   // :882-962
            if (has_ks) {
                RelinKeys rk = c.keygen->create_relin_keys(false);
                Ciphertext rl = c.ev->relinearize_new(prod, rk);                                 // :494-550
                report("relinearize", rl.polynomial_count() == 2 && eq(c.open_u(rl), mulv(m1, m2)));
                report("multiply_relin_add", eq(c.open_u(c.ev->add_new(rl, e3)), addv(mulv(m1, m2), m3)));
                if (levels >= 2) {
                    Ciphertext ms = c.ev->mod_switch_to_next_new(rl);                            // :581-614
                    report("mod_switch_to_next", ms.coeff_modulus_size() == levels - 1 && eq(c.open_u(ms), mulv(m1, m2)));
                }
                // key switching to another secret key (:440-475)
                KeyGenerator other(c.context);
                Encryptor enc_other(c.context);
                enc_other.set_secret_key(other.secret_key());
                KSwitchKeys ksk = c.keygen->create_keyswitching_key(other.secret_key(), false);
                Ciphertext under_other = enc_other.encrypt_symmetric_new(c.batch->encode_new(m1), false);
                report("keyswitching", eq(c.open_u(c.ev->apply_keyswitching_new(under_other, ksk)), m1));
                GaloisKeys gk = c.keygen->create_galois_keys(false);
                for (int step : {1, 3, -2}) {                                                   // :1115-1164
                    std::vector<uint64_t> want(slots);
                    for (size_t i = 0; i < row; i++) { want[i] = m1[(i + row + step) % row]; want[row + i] = m1[row + (i + row + step) % row]; }
                    report(("rotate_rows_" + std::to_string(step)).c_str(), eq(c.open_u(c.ev->rotate_rows_new(e1, step, gk)), want));
                }
                std::vector<uint64_t> sw(slots);
                for (size_t i = 0; i < row; i++) { sw[i] = m1[row + i]; sw[row + i] = m1[i]; }
                report("rotate_columns", eq(c.open_u(c.ev->rotate_columns_new(e1, gk)), sw));    // :1183-1216
            }
        } else {
            auto m1 = c.rand_c(), m2 = c.rand_c(), m3 = c.rand_c();
            std::vector<cd> pr(slots), sq(slots), sm(slots);
            for (size_t i = 0; i < slots; i++) { pr[i] = m1[i] * m2[i]; sq[i] = m1[i] * m1[i]; sm[i] = pr[i] + m3[i]; }
            Ciphertext e1 = c.encrypt(m1, c.scale), e2 = c.encrypt(m2, c.scale);
            report("decrypt", maxerr(c.open_c(e1), m1) < c.tol, maxerr(c.open_c(e1), m1));
            Ciphertext prod = c.ev->multiply_new(e1, e2);
            double e = maxerr(c.open_c(prod), pr); report("multiply", e < c.tol, e);
            e = maxerr(c.open_c(c.ev->square_new(e1)), sq); report("square", e < c.tol, e);
            Ciphertext e3 = c.encrypt(m3, prod.scale());
            e = maxerr(c.open_c(c.ev->add_new(prod, e3)), sm); report("multiply_add", e < c.tol, e);
            e = maxerr(c.open_c(c.ev->multiply_plain_new(e1, c.ckks->encode_complex64_simd_new(m2, std::nullopt, c.scale))), pr); report("multiply_plain", e < c.tol, e);
            if (has_ks) {
                RelinKeys rk = c.keygen->create_relin_keys(false);
                Ciphertext rl = c.ev->relinearize_new(prod, rk);
                e = maxerr(c.open_c(rl), pr); report("relinearize", rl.polynomial_count() == 2 && e < c.tol, e);
                if (levels >= 2) {
                    // :664-688: encode at scale * q_last so that the rescaled ciphertext is back at `scale`
                    const double q_last = (double)c.context->first_context_data().value()->parms().coeff_modulus()[levels - 1].value();
                    Ciphertext big = c.encrypt(m1, c.scale * q_last);
                    Ciphertext rs = c.ev->rescale_to_next_new(big);
                    e = maxerr(c.open_c(rs), m1); report("rescale_to_next", rs.coeff_modulus_size() == levels - 1 && e < c.tol, e);
                    // multiply -> relinearize -> rescale at a scale whose square is still a 2^20-scale after the division
                    const double s2 = std::sqrt(c.scale * q_last);
                    Ciphertext f1 = c.encrypt(m1, s2), f2 = c.encrypt(m2, s2);
                    Ciphertext chain = c.ev->rescale_to_next_new(c.ev->relinearize_new(c.ev->multiply_new(f1, f2), rk));
                    e = maxerr(c.open_c(chain), pr); report("multiply_relinearize_rescale", e < c.tol * 10, e);
                    Ciphertext ms = c.ev->mod_switch_to_next_new(e1);
                    e = maxerr(c.open_c(ms), m1); report("mod_switch_to_next", e < c.tol, e);
                }
                KeyGenerator other(c.context);
                Encryptor enc_other(c.context);
                enc_other.set_secret_key(other.secret_key());
                KSwitchKeys ksk = c.keygen->create_keyswitching_key(other.secret_key(), false);
                Ciphertext under_other = enc_other.encrypt_symmetric_new(c.ckks->encode_complex64_simd_new(m1, std::nullopt, c.scale), false);
                e = maxerr(c.open_c(c.ev->apply_keyswitching_new(under_other, ksk)), m1); report("keyswitching", e < c.tol, e);
                GaloisKeys gk = c.keygen->create_galois_keys(false);
                for (int step : {1, 3, -2}) {
                    std::vector<cd> want(slots);
                    for (size_t i = 0; i < slots; i++) want[i] = m1[(i + slots + step) % slots];
                    e = maxerr(c.open_c(c.ev->rotate_vector_new(e1, step, gk)), want); report(("rotate_vector_" + std::to_string(step)).c_str(), e < c.tol, e);
                }
                std::vector<cd> cj(slots);
                for (size_t i = 0; i < slots; i++) cj[i] = std::conj(m1[i]);
                e = maxerr(c.open_c(c.ev->complex_conjugate_new(e1, gk)), cj); report("complex_conjugate", e < c.tol, e);
            }
        }
        std::printf(failures ? "FAILED %zu\n" : "OK\n", failures);
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
