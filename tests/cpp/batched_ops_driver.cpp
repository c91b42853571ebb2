// C++ driver for tests/test_gpu_cpp_api.py::test_batched_ops_cpp_api: every Evaluator x_batched form returns, bit for bit, what
// the per-object call returns -- for operands that are scattered allocations (staged by the gather launch), for operands that are
// adjacent windows of one buffer (used in place), for in-place calls, and for mixed batches (per-object path).
// usage: batched_ops_driver <bfv|ckks> <count>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static size_t failures = 0;

static bool same(const Ciphertext& a, const Ciphertext& b) {
    return a.parms_id() == b.parms_id() && a.polynomial_count() == b.polynomial_count() && a.coeff_modulus_size() == b.coeff_modulus_size() &&
           a.is_ntt_form() == b.is_ntt_form() && a.scale() == b.scale() && a.correction_factor() == b.correction_factor() &&
           a.data().to_vector() == b.data().to_vector();
}

static void report(const char* name, const std::vector<Ciphertext>& got, const std::vector<Ciphertext>& want) {
    size_t bad = got.size() != want.size();
    for (size_t i = 0; i < got.size() && i < want.size(); i++) bad += !same(got[i], want[i]);
    std::printf("%s %zu\n", name, bad);
    failures += bad;
}

static std::vector<const Ciphertext*> cptr(const std::vector<Ciphertext>& v) { std::vector<const Ciphertext*> p; for (auto& c : v) p.push_back(&c); return p; }
static std::vector<Ciphertext*> mptr(std::vector<Ciphertext>& v) { std::vector<Ciphertext*> p; for (auto& c : v) p.push_back(&c); return p; }

int main(int argc, char** argv) {
    try {
        const bool ckks = argc > 1 && std::strcmp(argv[1], "ckks") == 0;
        const size_t count = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 6;
        const size_t n = 4096;
        const uint64_t t = 65537;
        EncryptionParameters params(ckks ? SchemeType::CKKS : SchemeType::BFV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {40, 40, 40, 40}));
        if (!ckks) params.set_plain_modulus(t);
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Nil, 0x31337);
        context->to_device_inplace();
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_secret_key(keygen.secret_key());
        Evaluator ev(context);
        RelinKeys rk = keygen.create_relin_keys(false);
        GaloisKeys gk = keygen.create_galois_keys_from_elements({3, n + 1}, false);
        std::mt19937_64 gen(3);

        std::vector<Ciphertext> a, b;
        std::vector<Plaintext> plains;
        if (ckks) {
            CKKSEncoder enc(context);
            std::uniform_real_distribution<double> U(-1.0, 1.0);
            for (size_t i = 0; i < 2 * count; i++) {
                std::vector<std::complex<double>> z(enc.slot_count());
                for (auto& v : z) v = {U(gen), U(gen)};
                Plaintext p = enc.encode_complex64_simd_new(z, std::nullopt, std::pow(2.0, 30));
                (i < count ? a : b).push_back(encryptor.encrypt_symmetric_new(p, false));
                if (i < count) plains.push_back(std::move(p));
            }
        } else {
            BatchEncoder enc(context);
            for (size_t i = 0; i < 2 * count; i++) {
                std::vector<uint64_t> v(i % 2 ? n : n / 2 + i);
                for (auto& x : v) x = gen() % t;
                Plaintext p = enc.encode_polynomial_new(v);
                (i < count ? a : b).push_back(encryptor.encrypt_symmetric_new(p, false));
                if (i < count) plains.push_back(std::move(p));
            }
        }
        std::vector<const Plaintext*> pp;
        for (auto& p : plains) pp.push_back(&p);

        // per-object results
        std::vector<Ciphertext> w_add, w_sub, w_neg, w_mul, w_relin, w_ms, w_gal, w_ntt;
        for (size_t i = 0; i < count; i++) {
            w_add.push_back(ev.add_new(a[i], b[i]));
            w_sub.push_back(ev.sub_new(a[i], b[i]));
            w_neg.push_back(ev.negate_new(a[i]));
            w_mul.push_back(ev.multiply_new(a[i], b[i]));
            w_relin.push_back(ev.relinearize_new(w_mul[i], rk));
            w_ms.push_back(ckks ? ev.rescale_to_next_new(w_relin[i]) : ev.mod_switch_to_next_new(w_relin[i]));
            w_gal.push_back(ev.apply_galois_new(a[i], 3, gk));
            w_ntt.push_back(ckks ? ev.transform_from_ntt_new(a[i]) : ev.transform_to_ntt_new(a[i]));
        }
        std::vector<Ciphertext> g(count);
        ev.add_batched(cptr(a), cptr(b), mptr(g)); report("add_batched", g, w_add);
        ev.sub_batched(cptr(a), cptr(b), mptr(g)); report("sub_batched", g, w_sub);
        ev.negate_batched(cptr(a), mptr(g)); report("negate_batched", g, w_neg);
        std::vector<Ciphertext> gm(count), gr(count), gs(count), gg(count), gn(count);
        ev.multiply_batched(cptr(a), cptr(b), mptr(gm)); report("multiply_batched", gm, w_mul);
        ev.relinearize_batched(cptr(gm), rk, mptr(gr)); report("relinearize_batched(adjacent windows)", gr, w_relin);
        ev.relinearize_batched(cptr(w_mul), rk, mptr(gr)); report("relinearize_batched(scattered)", gr, w_relin);
        if (ckks) ev.rescale_to_next_batched(cptr(gr), mptr(gs)); else ev.mod_switch_to_next_batched(cptr(gr), mptr(gs));
        report(ckks ? "rescale_to_next_batched" : "mod_switch_to_next_batched", gs, w_ms);
        ev.apply_galois_batched(cptr(a), 3, gk, mptr(gg)); report("apply_galois_batched", gg, w_gal);
        if (ckks) ev.transform_from_ntt_batched(cptr(a), mptr(gn)); else ev.transform_to_ntt_batched(cptr(a), mptr(gn));
        report("transform_ntt_batched", gn, w_ntt);
        // round trip in place: back to the inputs
        if (ckks) ev.transform_to_ntt_inplace_batched(mptr(gn)); else ev.transform_from_ntt_inplace_batched(mptr(gn));
        report("transform_ntt_inplace_batched(round trip)", gn, a);
        // in place: destination objects are the operands
        std::vector<Ciphertext> ip;
        for (auto& c : a) ip.push_back(c.clone());
        ev.negate_inplace_batched(mptr(ip)); report("negate_inplace_batched", ip, w_neg);
        ev.add_batched(cptr(ip), cptr(ip), mptr(ip));
        std::vector<Ciphertext> w_dbl;
        for (size_t i = 0; i < count; i++) w_dbl.push_back(ev.add_new(w_neg[i], w_neg[i]));
        report("add_batched(in place, aliased operands)", ip, w_dbl);
        // ciphertext (+) plaintext and ciphertext x plaintext
        std::vector<Ciphertext> w_ap, w_sp, g_ap(count), g_sp(count);
        for (size_t i = 0; i < count; i++) { w_ap.push_back(ev.add_plain_new(a[i], plains[i])); w_sp.push_back(ev.sub_plain_new(a[i], plains[i])); }
        ev.add_plain_batched(cptr(a), pp, mptr(g_ap)); report("add_plain_batched", g_ap, w_ap);
        ev.sub_plain_batched(cptr(a), pp, mptr(g_sp)); report("sub_plain_batched", g_sp, w_sp);
        {
            std::vector<Ciphertext> an;
            std::vector<Plaintext> pn;
            for (size_t i = 0; i < count; i++) {
                an.push_back(ckks ? a[i].clone() : ev.transform_to_ntt_new(a[i]));
                pn.push_back(ckks ? plains[i].clone() : ev.transform_plain_to_ntt_new(plains[i], context->first_parms_id()));
            }
            std::vector<const Plaintext*> pnp;
            for (auto& p : pn) pnp.push_back(&p);
            std::vector<Ciphertext> w_mp, g_mp(count);
            for (size_t i = 0; i < count; i++) w_mp.push_back(ev.multiply_plain_new(an[i], pn[i]));
            ev.multiply_plain_batched(cptr(an), pnp, mptr(g_mp)); report("multiply_plain_batched", g_mp, w_mp);
        }
        // ---- the remaining spellings: x_new_batched / x_inplace_batched, key switching, rotations, level switching, shifts ----
        {
            report("add_new_batched", ev.add_new_batched(cptr(a), cptr(b)), w_add);
            report("negate_new_batched", ev.negate_new_batched(cptr(a)), w_neg);
            std::vector<Ciphertext> ip2;
            for (auto& c : a) ip2.push_back(c.clone());
            ev.sub_inplace_batched(mptr(ip2), cptr(b)); report("sub_inplace_batched", ip2, w_sub);
            KeyGenerator other(context);
            KSwitchKeys ksk = keygen.create_keyswitching_key(other.secret_key(), false);
            std::vector<Ciphertext> w_ks;
            for (auto& c : a) w_ks.push_back(ev.apply_keyswitching_new(c, ksk));
            report("apply_keyswitching_new_batched", ev.apply_keyswitching_new_batched(cptr(a), ksk), w_ks);
            std::vector<Ciphertext> ip3;
            for (auto& c : a) ip3.push_back(c.clone());
            ev.apply_keyswitching_inplace_batched(mptr(ip3), ksk); report("apply_keyswitching_inplace_batched", ip3, w_ks);
            GaloisKeys gall = keygen.create_galois_keys(false);
            for (int steps : {1, 3, -5}) {                                   // 3 and -5 go through the NAF decomposition
                std::vector<Ciphertext> w_rot;
                for (auto& c : a) w_rot.push_back(ckks ? ev.rotate_vector_new(c, steps, gall) : ev.rotate_rows_new(c, steps, gall));
                char label[64];
                std::snprintf(label, sizeof label, "%s(%d)", ckks ? "rotate_vector_new_batched" : "rotate_rows_new_batched", steps);
                report(label, ckks ? ev.rotate_vector_new_batched(cptr(a), steps, gall) : ev.rotate_rows_new_batched(cptr(a), steps, gall), w_rot);
            }
            {
                std::vector<Ciphertext> w_conj, ip4;
                for (auto& c : a) { w_conj.push_back(ckks ? ev.complex_conjugate_new(c, gall) : ev.rotate_columns_new(c, gall)); ip4.push_back(c.clone()); }
                if (ckks) ev.complex_conjugate_inplace_batched(mptr(ip4), gall); else ev.rotate_columns_inplace_batched(mptr(ip4), gall);
                report(ckks ? "complex_conjugate_inplace_batched" : "rotate_columns_inplace_batched", ip4, w_conj);
            }
            const ParmsID last = context->last_parms_id();
            std::vector<Ciphertext> w_to, w_shift, w_next;
            for (auto& c : a) { w_to.push_back(ev.mod_switch_to_new(c, last)); w_next.push_back(ev.mod_switch_to_next_new(c)); }
            report("mod_switch_to_new_batched(last level)", ev.mod_switch_to_new_batched(cptr(a), last), w_to);
            report("mod_switch_to_next_new_batched", ev.mod_switch_to_next_new_batched(cptr(a)), w_next);
            if (!ckks) {                                                       // the shift works on coefficient-form ciphertexts
                for (auto& c : a) w_shift.push_back(ev.negacyclic_shift_new(c, 37));
                report("negacyclic_shift_new_batched", ev.negacyclic_shift_new_batched(cptr(a), 37), w_shift);
            }
            if (!ckks) {
                std::vector<Ciphertext> w_mpn;
                for (size_t i = 0; i < count; i++) w_mpn.push_back(ev.multiply_plain_new(a[i], plains[i]));
                report("multiply_plain_new_batched", ev.multiply_plain_new_batched(cptr(a), pp), w_mpn);
                // plaintext side: bfv_centralize + NTT == transform_plain_to_ntt; its inverse returns the centred lift; bfv_scale_up == BatchEncoder::scale_up
                BatchEncoder benc(context);
                const ParmsID first = context->first_parms_id();
                size_t bad = 0;
                std::vector<Plaintext> cen = ev.bfv_centralize_new_batched(pp, first), ntt = ev.transform_plain_to_ntt_new_batched(pp, first);
                for (size_t i = 0; i < count; i++) {
                    bad += ev.transform_plain_to_ntt_new(cen[i], first).data().to_vector() != ntt[i].data().to_vector();
                    bad += ev.transform_plain_from_ntt_new(ntt[i]).data().to_vector() != cen[i].data().to_vector();
                    Plaintext full = plains[i].clone();
                    bad += ev.bfv_scale_up_new(full, first).data().to_vector() != benc.scale_up_new(full, first).expanded_rns(3, n).to_vector();
                }
                std::printf("plaintext families %zu\n", bad);
                failures += bad;
            }
            // a caller-supplied generator decides the seed of c1: same generator state, same seed and same c1
            utils::RandomGenerator g1(99), g2(99);
            Ciphertext e1 = ckks ? encryptor.encrypt_zero_symmetric_new(true, std::nullopt, &g1) : encryptor.encrypt_symmetric_new(plains[0], true, &g1);
            Ciphertext e2 = ckks ? encryptor.encrypt_zero_symmetric_new(true, std::nullopt, &g2) : encryptor.encrypt_symmetric_new(plains[0], true, &g2);
            const bool seeded = e1.seed() != 0 && e1.seed() == e2.seed();
            std::vector<Ciphertext> zs = encryptor.encrypt_zero_symmetric_new_batched(3, true);
            const bool zeros = zs.size() == 3 && zs[0].seed() != 0 && zs[0].seed() != zs[1].seed() && zs[0].is_ntt_form() == ckks;
            std::printf("u_prng_seed %d zero_batched %d\n", seeded ? 1 : 0, zeros ? 1 : 0);
            failures += !seeded + !zeros;
        }
        // a mixed batch (one three-polynomial member) takes the per-object path
        {
            std::vector<Ciphertext> mix;
            for (size_t i = 0; i < count; i++) mix.push_back(i == 1 ? w_mul[i].clone() : a[i].clone());
            std::vector<Ciphertext> want, got(count);
            for (auto& c : mix) want.push_back(ev.negate_new(c));
            ev.negate_batched(cptr(mix), mptr(got)); report("negate_batched(mixed)", got, want);
        }
        // size mismatch throws what the reference throws
        bool threw = false;
        try { std::vector<Ciphertext> shortd(count - 1); ev.add_batched(cptr(a), cptr(b), mptr(shortd)); } catch (const std::invalid_argument&) { threw = true; }
        std::printf("size_mismatch_rejected %d\n", threw ? 1 : 0);
        failures += !threw;
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
