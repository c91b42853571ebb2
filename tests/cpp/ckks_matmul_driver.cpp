// C++ driver for tests/test_gpu_cpp_api.py::test_ckks_matmul_cpp_api: examples/11_ckks_matmul.cu -- y = x * w + s on real matrices with
// encrypted x through troy::linear::MatmulHelper and the CKKS encoder (N=8192, {60,40,40,60}, scale 2^20): encode weights and bias
// (bias at scale^2), encrypt inputs (seed-compressed on the wire), matmul, optional mod-switch and output packing, add the bias,
// outputs through save_terms / load_terms, decrypt, compare with the plain result.
// usage: ckks_matmul_driver <batch> <input_dims> <output_dims> [pack_lwe 0|1] [mod_switch 0|1]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <sstream>

#include "../../troy-nova_amd/troy/matmul.h"

using namespace troy;
using namespace troy::linear;

int main(int argc, char** argv) {
    try {
        const size_t M = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : 25, R = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 30,
                     Nn = argc > 3 ? std::strtoull(argv[3], nullptr, 0) : 35;
        const bool pack_lwe = argc > 4 && std::atoi(argv[4]) != 0, mod_switch = argc > 5 && std::atoi(argv[5]) != 0;
        const size_t n = 8192;
        const double scale = static_cast<double>(1 << 20);
        EncryptionParameters params(SchemeType::CKKS);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Classical128, 0x11c);
        context->to_device_inplace();
        CKKSEncoder encoder(context);
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_secret_key(keygen.secret_key());
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator evaluator(context);
        GaloisKeys automorphism_key;
        if (pack_lwe) automorphism_key = keygen.create_automorphism_keys(false);

        std::mt19937_64 gen(4);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        std::vector<double> x(M * R), w(R * Nn), s(M * Nn), want(M * Nn, 0.0);
        for (auto& v : x) v = U(gen);
        for (auto& v : w) v = U(gen);
        for (auto& v : s) v = U(gen);
        for (size_t i = 0; i < M; i++)
            for (size_t k = 0; k < R; k++)
                for (size_t j = 0; j < Nn; j++) want[i * Nn + j] += x[i * R + k] * w[k * Nn + j];
        for (size_t i = 0; i < M * Nn; i++) want[i] += s[i];

        MatmulHelper helper(M, R, Nn, n, MatmulObjective::EncryptLeft, pack_lwe);
        std::printf("block %zu %zu %zu pack_lwe %d mod_switch %d\n", helper.batch_block, helper.input_block, helper.output_block, pack_lwe ? 1 : 0, mod_switch ? 1 : 0);
        Plain2d we = helper.encode_weights_doubles(encoder, w.data(), std::nullopt, scale);
        Plain2d se = helper.encode_outputs_doubles(encoder, s.data(), std::nullopt, scale * scale);
        Cipher2d xe = helper.encrypt_inputs_doubles(encryptor, encoder, x.data(), std::nullopt, scale);
        std::stringstream x_serialized;
        xe.save(x_serialized, context);
        const size_t x_bytes = x_serialized.str().size();
        xe = Cipher2d::load_new(x_serialized, context);
        Cipher2d ye = helper.matmul(evaluator, xe, we);
        size_t fly_bad = 0;
        {
            Cipher2d yf = helper.matmul_fly_doubles(encoder, evaluator, xe, w.data(), std::nullopt, scale);
            for (size_t r = 0; r < ye.data().size(); r++)
                for (size_t c = 0; c < ye[r].size(); c++) fly_bad += yf[r][c].data().to_vector() != ye[r][c].data().to_vector() || yf[r][c].scale() != ye[r][c].scale();
            std::printf("fly_mismatches %zu\n", fly_bad);
        }
        if (mod_switch) ye.mod_switch_to_next_inplace(evaluator);
        if (pack_lwe) ye = helper.pack_outputs(evaluator, automorphism_key, ye);
        ye.add_plain_inplace(evaluator, se);
        std::stringstream y_serialized;
        helper.serialize_outputs(evaluator, ye, y_serialized);
        const size_t y_bytes = y_serialized.str().size();
        Cipher2d yl = helper.deserialize_outputs(evaluator, y_serialized);
        const std::vector<double> got = helper.decrypt_outputs_doubles(encoder, decryptor, yl);
        double err = 0;
        for (size_t i = 0; i < got.size(); i++) err = std::max(err, std::fabs(got[i] - want[i]));
        size_t outputs_n = 0;
        for (auto& r : ye.data()) outputs_n += r.size();
        std::printf("outputs %zu level %zu scale_log2 %.2f bytes inputs %zu outputs %zu\n", outputs_n, ye[0][0].coeff_modulus_size(), std::log2(ye[0][0].scale()), x_bytes, y_bytes);
        std::printf("max_error %.3e\n", err);
        const bool ok = err < 1e-3 && fly_bad == 0;                       // the reference's own check uses an absolute tolerance of this order at scale 2^20
        std::printf(ok ? "OK\n" : "FAIL\n");
        MemoryPool::Destroy();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
