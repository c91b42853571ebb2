// C++ driver for tests/test_gpu_cpp_api.py::test_conv2d_cpp_api: examples/14_bfv_conv2d.cu -- y = conv2d(x, w) + s (valid cross-correlation)
// with encrypted images through troy::linear::Conv2dHelper (N=8192, {60,40,40,60}, t=2^21): encode weights and bias, encrypt the image
// tiles (seed-compressed on the wire), conv2d, optional mod-switch, add the bias, outputs through save_terms / load_terms, decrypt,
// compare with the plain result mod t.
// usage: conv2d_driver <batch> <in_ch> <out_ch> <H> <W> <kh> <kw> [mod_switch 0|1] [objective left|right]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <sstream>

#include "../../troy-nova_amd/troy/conv2d.h"

using namespace troy;
using namespace troy::linear;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    try {
        auto arg = [&](int i, size_t d) { return argc > i ? std::strtoull(argv[i], nullptr, 0) : d; };
        const size_t bs = arg(1, 2), ic = arg(2, 3), oc = arg(3, 5), ih = arg(4, 15), iw = arg(5, 15), kh = arg(6, 3), kw = arg(7, 3);
        const bool mod_switch = arg(8, 0) != 0;
        const bool right = argc > 9 && std::string(argv[9]) == "right";
        const size_t n = 8192, oh = ih - kh + 1, ow = iw - kw + 1;
        const uint64_t t = 1ull << 21;
        EncryptionParameters params(SchemeType::BFV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        params.set_plain_modulus(t);
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Classical128, 0xc02d);
        context->to_device_inplace();
        BatchEncoder encoder(context);
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_secret_key(keygen.secret_key());
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator evaluator(context);

        std::mt19937_64 gen(14);
        std::vector<uint64_t> x(bs * ic * ih * iw), w(oc * ic * kh * kw), s(bs * oc * oh * ow), want(bs * oc * oh * ow, 0);
        for (auto& v : x) v = gen() % t;
        for (auto& v : w) v = gen() % t;
        for (auto& v : s) v = gen() % t;
        for (size_t b = 0; b < bs; b++)
            for (size_t o = 0; o < oc; o++)
                for (size_t i = 0; i < oh; i++)
                    for (size_t j = 0; j < ow; j++) {
                        uint64_t acc = s[b * oc * oh * ow + o * oh * ow + i * ow + j];
                        for (size_t c = 0; c < ic; c++)
                            for (size_t p = 0; p < kh; p++)
                                for (size_t q = 0; q < kw; q++)
                                    acc = (acc + x[b * ic * ih * iw + c * ih * iw + (i + p) * iw + (j + q)] * w[o * ic * kh * kw + c * kh * kw + p * kw + q]) % t;
                        want[b * oc * oh * ow + o * oh * ow + i * ow + j] = acc;
                    }

        Conv2dHelper helper(bs, ic, oc, ih, iw, kh, kw, n, right ? MatmulObjective::EncryptRight : MatmulObjective::EncryptLeft);
        {
            std::stringstream text;
            text << helper;
            if (text.str().rfind("Conv2dHelper(batch_size=", 0) != 0 || text.str().find(right ? "objective=EncryptRight)" : "objective=EncryptLeft)") == std::string::npos || !helper.batched_mul) {
                std::printf("operator<< gives %s\n", text.str().c_str());
                return 1;
            }
        }
        std::printf("block b %zu ci %zu co %zu h %zu w %zu tiles %zu objective %s\n", helper.batch_block, helper.input_channel_block, helper.output_channel_block,
                    helper.image_height_block, helper.image_width_block, helper.get_total_batch_size(), right ? "right" : "left");
        const double t0 = now();
        Plain2d se = helper.encode_outputs_uint64s(encoder, s.data());
        Cipher2d ye;
        size_t wire_in = 0;
        if (!right) {
            Plain2d we = helper.encode_weights_uint64s(encoder, w.data());
            Cipher2d xe = helper.encrypt_inputs_uint64s(encryptor, encoder, x.data());
            std::stringstream xs;
            xe.save(xs, context);
            wire_in = xs.str().size();
            xe = Cipher2d::load_new(xs, context);
            ye = helper.conv2d(evaluator, xe, we);
        } else {
            Plain2d xp = helper.encode_inputs_uint64s(encoder, x.data());
            Cipher2d wc = helper.encrypt_weights_uint64s(encryptor, encoder, w.data());
            std::stringstream ws;
            wc.save(ws, context);
            wire_in = ws.str().size();
            wc = Cipher2d::load_new(ws, context);
            ye = helper.conv2d_reverse(evaluator, xp, wc);
        }
        if (mod_switch) ye.mod_switch_to_next_inplace(evaluator);
        ye.add_plain_inplace(evaluator, se);
        std::stringstream ys;
        helper.serialize_outputs(evaluator, ye, ys);
        const size_t wire_out = ys.str().size();
        Cipher2d yl = helper.deserialize_outputs(evaluator, ys);
        const std::vector<uint64_t> got = helper.decrypt_outputs_uint64s(encoder, decryptor, yl);
        const double t1 = now();
        size_t bad = got.size() != want.size();
        for (size_t i = 0; i < got.size() && i < want.size(); i++) bad += got[i] != want[i];
        std::printf("bytes in %zu out %zu, %.2f ms end to end\n", wire_in, wire_out, (t1 - t0) * 1e3);
        std::printf("mismatches %zu of %zu\n", bad, want.size());
        std::printf(bad ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
