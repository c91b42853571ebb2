// C++ driver for tests/test_gpu_cpp_api.py::test_conv2d_ckks_ring2k_cpp_api: Conv2dHelper beyond BFV mod t --
//   (1) CKKS (examples/15_ckks_conv2d.cu): y = conv2d(x, w) + s on real tensors, N=8192, {60,40,40,60}, scale 2^20, bias at scale^2;
//   (2) ring-2^k: the same over Z_{2^k} through PolynomialEncoderRing2k<T> (uint64 k=64, uint32 k=32, uint128 k=100), N=8192, five 60-bit primes.
// Both with encrypted images (seed-compressed on the wire) and outputs through save_terms / load_terms.
#include <cmath>
#include <cstdio>
#include <random>
#include <sstream>

#include "../../troy-nova_amd/troy/conv2d.h"

using namespace troy;
using namespace troy::linear;
typedef unsigned __int128 u128;

static const size_t bs = 2, ic = 3, oc = 4, ih = 12, iw = 11, kh = 3, kw = 2, oh = ih - kh + 1, ow = iw - kw + 1;

template <typename T, typename Mul>
static std::vector<T> plain_conv(const std::vector<T>& x, const std::vector<T>& w, const std::vector<T>& s, Mul&& fix) {
    std::vector<T> want(bs * oc * oh * ow);
    for (size_t b = 0; b < bs; b++)
        for (size_t o = 0; o < oc; o++)
            for (size_t i = 0; i < oh; i++)
                for (size_t j = 0; j < ow; j++) {
                    T acc = s[b * oc * oh * ow + o * oh * ow + i * ow + j];
                    for (size_t c = 0; c < ic; c++)
                        for (size_t p = 0; p < kh; p++)
                            for (size_t q = 0; q < kw; q++) acc = fix(acc + x[b * ic * ih * iw + c * ih * iw + (i + p) * iw + (j + q)] * w[o * ic * kh * kw + c * kh * kw + p * kw + q]);
                    want[b * oc * oh * ow + o * oh * ow + i * ow + j] = acc;
                }
    return want;
}

static bool run_ckks() {
    const size_t n = 8192;
    const double scale = static_cast<double>(1 << 20);
    EncryptionParameters params(SchemeType::CKKS);
    params.set_poly_modulus_degree(n);
    params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
    HeContextPointer context = HeContext::create(params, true, SecurityLevel::Classical128, 0x15c);
    context->to_device_inplace();
    CKKSEncoder encoder(context);
    KeyGenerator keygen(context);
    Encryptor encryptor(context);
    encryptor.set_secret_key(keygen.secret_key());
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    std::mt19937_64 gen(15);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> x(bs * ic * ih * iw), w(oc * ic * kh * kw), s(bs * oc * oh * ow);
    for (auto& v : x) v = U(gen);
    for (auto& v : w) v = U(gen);
    for (auto& v : s) v = U(gen);
    const std::vector<double> want = plain_conv(x, w, s, [](double v) { return v; });
    Conv2dHelper helper(bs, ic, oc, ih, iw, kh, kw, n, MatmulObjective::EncryptLeft);
    Plain2d we = helper.encode_weights_doubles(encoder, w.data(), std::nullopt, scale);
    Plain2d se = helper.encode_outputs_doubles(encoder, s.data(), std::nullopt, scale * scale);
    Cipher2d xe = helper.encrypt_inputs_doubles(encryptor, encoder, x.data(), std::nullopt, scale);
    std::stringstream xs;
    xe.save(xs, context);
    xe = Cipher2d::load_new(xs, context);
    Cipher2d ye = helper.conv2d(evaluator, xe, we);
    ye.add_plain_inplace(evaluator, se);
    std::stringstream ys;
    helper.serialize_outputs(evaluator, ye, ys);
    Cipher2d yl = helper.deserialize_outputs(evaluator, ys);
    const std::vector<double> got = helper.decrypt_outputs_doubles(encoder, decryptor, yl);
    double err = got.size() == want.size() ? 0.0 : 1.0;
    for (size_t i = 0; i < got.size() && i < want.size(); i++) err = std::max(err, std::fabs(got[i] - want[i]));
    // the reversed objective: plaintext images, encrypted kernels
    Conv2dHelper rhelper(bs, ic, oc, ih, iw, kh, kw, n, MatmulObjective::EncryptRight);
    Plain2d xp = rhelper.encode_inputs_doubles(encoder, x.data(), std::nullopt, scale);
    Cipher2d wc = rhelper.encrypt_weights_doubles(encryptor, encoder, w.data(), std::nullopt, scale);
    wc.expand_seed(context);
    Cipher2d yr = rhelper.conv2d_reverse(evaluator, xp, wc);
    yr.add_plain_inplace(evaluator, rhelper.encode_outputs_doubles(encoder, s.data(), std::nullopt, scale * scale));
    const std::vector<double> got_r = rhelper.decrypt_outputs_doubles(encoder, decryptor, yr);
    double err_r = got_r.size() == want.size() ? 0.0 : 1.0;
    for (size_t i = 0; i < got_r.size() && i < want.size(); i++) err_r = std::max(err_r, std::fabs(got_r[i] - want[i]));
    std::printf("ckks max_error %.3e reverse %.3e\n", err, err_r);
    return err < 1e-3 && err_r < 1e-3;
}

template <typename T>
static bool run_ring2k(const HeContextPointer& context, const Encryptor& encryptor, const Decryptor& decryptor, const Evaluator& evaluator, size_t k, const char* name) {
    PolynomialEncoderRing2k<T> encoder(context, k);
    const T mask = encoder.t_mask();
    std::mt19937_64 gen(sizeof(T) * 3 + k);
    auto rnd = [&]() { T v = static_cast<T>(gen()); if (sizeof(T) == 16) v = (v << 32 << 32) | static_cast<T>(gen()); return static_cast<T>(v & mask); };
    std::vector<T> x(bs * ic * ih * iw), w(oc * ic * kh * kw), s(bs * oc * oh * ow);
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = rnd();
    for (auto& v : s) v = rnd();
    const std::vector<T> want = plain_conv(x, w, s, [mask](T v) { return static_cast<T>(v & mask); });
    Conv2dHelper helper(bs, ic, oc, ih, iw, kh, kw, encoder.slot_count(), MatmulObjective::EncryptLeft);
    Plain2d we = helper.encode_weights_ring2k(encoder, w.data(), std::nullopt);
    Cipher2d xe = helper.encrypt_inputs_ring2k(encryptor, encoder, x.data(), std::nullopt);
    std::stringstream xs;
    xe.save(xs, context);
    xe = Cipher2d::load_new(xs, context);
    Cipher2d ye = helper.conv2d(evaluator, xe, we);
    ye.mod_switch_to_next_inplace(evaluator);
    ye.add_plain_inplace(evaluator, helper.encode_outputs_ring2k(encoder, s.data(), ye[0][0].parms_id()));
    std::stringstream ys;
    helper.serialize_outputs(evaluator, ye, ys);
    Cipher2d yl = helper.deserialize_outputs(evaluator, ys);
    const std::vector<T> got = helper.decrypt_outputs_ring2k(encoder, decryptor, yl);
    size_t bad = got.size() != want.size();
    for (size_t i = 0; i < got.size() && i < want.size(); i++) bad += got[i] != want[i];
    std::printf("ring2k %s k=%zu mismatches %zu of %zu\n", name, k, bad, want.size());
    return bad == 0;
}

int main() {
    try {
        bool ok = run_ckks();
        {
            const size_t n = 8192;
            EncryptionParameters parms(SchemeType::BFV);
            parms.set_poly_modulus_degree(n);
            parms.set_plain_modulus(1 << 20);                                   // unused by the ring-2^k encoder
            parms.set_coeff_modulus(CoeffModulus::create(n, {60, 60, 60, 60, 60, 60}));
            HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil, 0x2c);   // 360 bits at N=8192: a functional test, not a secure parameter set
            context->to_device_inplace();
            KeyGenerator keygen(context);
            Encryptor encryptor(context);
            encryptor.set_secret_key(keygen.secret_key());
            Decryptor decryptor(context, keygen.secret_key());
            Evaluator evaluator(context);
            ok = run_ring2k<uint64_t>(context, encryptor, decryptor, evaluator, 64, "uint64") && ok;
            ok = run_ring2k<uint32_t>(context, encryptor, decryptor, evaluator, 32, "uint32") && ok;
            ok = run_ring2k<u128>(context, encryptor, decryptor, evaluator, 100, "uint128") && ok;
        }
        std::printf(ok ? "OK\n" : "FAIL\n");
        MemoryPool::Destroy();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
