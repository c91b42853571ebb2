// matmul.h -- troy::linear::MatmulHelper, the BFV matrix product of the reference's matmul application
// (src/app/matmul.h, matmul.cu; BASELINE config 5 = examples/10_bfv_matmul.cu at 512x512x512) on top of the mirror API.
//
// Supported here: BFV (and BGV for the ciphertext x ciphertext product) with BatchEncoder polynomial (coefficient) packing; the three
// objectives -- encrypted inputs x plaintext weights (EncryptLeft, the example's configuration), plaintext inputs x encrypted
// weights (EncryptRight), both encrypted (Crossed) -- with and without packing of the outputs (pack_lwe), output mod-switch,
// bias addition and the outputs' partial wire format.  Layout, as the reference:
//   input block  (batch rows li..ui, input columns lj..uj):   coefficient (i-li)*ib*ob + (j-lj)              = x[i][j]
//   weight block (input rows li..ui, output columns lj..uj):  coefficient (j-lj)*ib + ib - (i-li) - 1         = w[i][j]
//   output block: y[i][j] is coefficient (i-li)*ib*ob + (j-lj)*ib + ib - 1 of sum_k input[b][k] * weight[k][j-block]
// with (bb, ib, ob) = (batch_block, input_block, output_block), bb*ib*ob <= N.
#pragma once
#include <functional>
#include <ostream>

#include "ring2k.h"
#include "troy.h"

namespace troy { namespace linear {

enum class MatmulObjective : uint8_t { EncryptLeft = 0, EncryptRight = 1, Crossed = 2 };
inline std::ostream& operator<<(std::ostream& os, const MatmulObjective& obj) {        // matmul.h:165-172
    static const char* const names[] = {"EncryptLeft", "EncryptRight", "Crossed"};
    return os << names[static_cast<size_t>(obj) < 3 ? static_cast<size_t>(obj) : 0];
}

class Cipher2d;

class Plain2d {
public:
    Plain2d clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plain2d p; for (const auto& r : inner) { auto& row = p.new_row(); for (const Plaintext& x : r) row.push_back(x.clone(pool)); } return p; }
    // app/cipher2d.h:52-56: element-wise encryption
    Cipher2d encrypt_symmetric(const Encryptor& encryptor, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Cipher2d encrypt_asymmetric(const Encryptor& encryptor, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<std::vector<Plaintext>>& data() { return inner; }
    const std::vector<std::vector<Plaintext>>& data() const { return inner; }
    size_t size() const { return inner.size(); }
    size_t rows() const { return inner.size(); }
    size_t columns() const { return inner.empty() ? 0 : inner[0].size(); }
    std::vector<Plaintext>& operator[](size_t i) { return inner[i]; }
    const std::vector<Plaintext>& operator[](size_t i) const { return inner[i]; }
    std::vector<Plaintext>& new_row() { inner.emplace_back(); return inner.back(); }
    void resize(size_t rows, size_t columns) { inner.resize(rows); for (auto& r : inner) r.resize(columns); }
    // app/cipher2d.cu: [rows][row size, plaintexts...] in the Plaintext wire format
    size_t save(std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const;
    void load(std::istream& stream, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static Plain2d load_new(std::istream& stream, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { Plain2d p; p.load(stream, pool); return p; }
private:
    std::vector<std::vector<Plaintext>> inner;
};

class Cipher2d {
public:
    std::vector<std::vector<Ciphertext>>& data() { return inner; }
    const std::vector<std::vector<Ciphertext>>& data() const { return inner; }
    size_t size() const { return inner.size(); }
    size_t rows() const { return inner.size(); }
    size_t columns() const { return inner.empty() ? 0 : inner[0].size(); }
    std::vector<Ciphertext>& operator[](size_t i) { return inner[i]; }
    const std::vector<Ciphertext>& operator[](size_t i) const { return inner[i]; }
    std::vector<Ciphertext>& new_row() { inner.emplace_back(); return inner.back(); }
    void resize(size_t rows, size_t columns) { inner.resize(rows); for (auto& r : inner) r.resize(columns); }
    Cipher2d clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void expand_seed(HeContextPointer context);
    // app/cipher2d.cu:29-67: [rows][row size, ciphertexts...] in the Ciphertext wire format
    size_t save(std::ostream& stream, HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const;
    void load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static Cipher2d load_new(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { Cipher2d c; c.load(stream, context, pool); return c; }
    size_t serialized_size_upperbound(HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const;
    // app/cipher2d.h:164-230
    void mod_switch_to_next_inplace(const Evaluator& evaluator, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    Cipher2d mod_switch_to_next(const Evaluator& evaluator, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Cipher2d c = clone(pool); c.mod_switch_to_next_inplace(evaluator, pool); return c; }
    void add_inplace(const Evaluator& evaluator, const Cipher2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { translate_inplace(evaluator, other, false, pool); }
    void sub_inplace(const Evaluator& evaluator, const Cipher2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { translate_inplace(evaluator, other, true, pool); }
    Cipher2d add(const Evaluator& evaluator, const Cipher2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Cipher2d c = clone(pool); c.add_inplace(evaluator, other, pool); return c; }
    Cipher2d sub(const Evaluator& evaluator, const Cipher2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Cipher2d c = clone(pool); c.sub_inplace(evaluator, other, pool); return c; }
    void relinearize_inplace(const Evaluator& evaluator, const RelinKeys& relin_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    Cipher2d relinearize(const Evaluator& evaluator, const RelinKeys& relin_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Cipher2d c = clone(pool); c.relinearize_inplace(evaluator, relin_keys, pool); return c; }
    Plain2d decrypt(const Decryptor& decryptor, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void add_plain_inplace(const Evaluator& evaluator, const Plain2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { translate_plain_inplace(evaluator, other, false, pool); }
    void sub_plain_inplace(const Evaluator& evaluator, const Plain2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { translate_plain_inplace(evaluator, other, true, pool); }
    Cipher2d add_plain(const Evaluator& evaluator, const Plain2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Cipher2d c = clone(pool); c.add_plain_inplace(evaluator, other, pool); return c; }
    Cipher2d sub_plain(const Evaluator& evaluator, const Plain2d& other, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Cipher2d c = clone(pool); c.sub_plain_inplace(evaluator, other, pool); return c; }
private:
    void translate_inplace(const Evaluator& evaluator, const Cipher2d& other, bool subtract, MemoryPoolHandle pool);
    void translate_plain_inplace(const Evaluator& evaluator, const Plain2d& other, bool subtract, MemoryPoolHandle pool);
    std::vector<std::vector<Ciphertext>> inner;
};

// Shared by the linear-algebra helpers (matmul, conv2d): `packed` holds rows*cols coefficient vectors of length `len` back to back.
namespace detail {
// encode_for_plain + ensure_ntt_form(centralize) (app/matmul.cu:14-70): one copy, one centralize launch, one NTT launch; the
// plaintexts are windows of one buffer
Plain2d encode_blocks_for_plain(const BatchEncoder& encoder, const std::vector<uint64_t>& packed, size_t rows, size_t cols, size_t len, MemoryPoolHandle pool);
// encode_for_cipher + ensure_ntt_form(scale up) + encrypt_symmetric_batched(save_seed = true): NTT-form ciphertexts carrying the c1 seed
Cipher2d encrypt_blocks(const Encryptor& encryptor, const BatchEncoder& encoder, const std::vector<uint64_t>& packed, size_t rows, size_t cols, size_t len,
                        MemoryPoolHandle pool);
// ret[b][j] = sum_i ct(b, i, j) (.) pt(b, i, j): one multiply_plain_accumulate launch, results in one buffer, one trailing INTT (BFV)
Cipher2d accumulate_products(const Evaluator& evaluator, const Ciphertext& like, size_t batch_split, size_t input_split, size_t output_split,
                             const std::function<const Ciphertext*(size_t, size_t, size_t)>& ct_at,
                             const std::function<const Plaintext*(size_t, size_t, size_t)>& pt_at, MemoryPoolHandle pool);
// the same sum with the plaintext operand produced one input block at a time: row(i) returns the output_split NTT-form plaintexts of
// input block i, which are multiplied into the running sums and dropped (app/matmul.cu:725-748 matmul_fly)
Cipher2d accumulate_products_fly(const Evaluator& evaluator, const Cipher2d& inputs, size_t batch_split, size_t input_split, size_t output_split,
                                 const std::function<std::vector<Plaintext>(size_t)>& row, MemoryPoolHandle pool);
}  // namespace detail

class MatmulHelper {
public:
    size_t batch_size, input_dims, output_dims, slot_count;
    size_t batch_block = 0, input_block = 0, output_block = 0;
    MatmulObjective objective;
    bool pack_lwe;
    bool batched_mul = true;          // matmul.h:60-65: the products of a call are formed by batched kernels (always the case here)
    MemoryPoolHandle pool;

    void set_pool(MemoryPoolHandle p) { pool = std::move(p); }
    MatmulHelper(size_t batch_size, size_t input_dims, size_t output_dims, size_t slot_count,
                 MatmulObjective objective = MatmulObjective::EncryptLeft, bool pack_lwe = true, MemoryPoolHandle pool = MemoryPool::GlobalPool());

    // weights [input_dims][output_dims] row-major -> [ceil(in/ib)][ceil(out/ob)] NTT-form plaintexts (centred lift: they multiply
    // ciphertexts) or NTT-form ciphertexts carrying their c1 seed (scaled up by q/t; expand_seed / load before use)
    Plain2d encode_weights_uint64s(const BatchEncoder& encoder, const uint64_t* weights) const;
    Cipher2d encrypt_weights_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* weights) const;
    // inputs [batch_size][input_dims] row-major -> [ceil(batch/bb)][ceil(in/ib)], the same two forms
    Plain2d encode_inputs_uint64s(const BatchEncoder& encoder, const uint64_t* inputs) const;
    Cipher2d encrypt_inputs_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* inputs) const;
    // ret[b][j] = sum_i a[b][i] * w[i][j]: ONE multiply_plain_accumulate launch over all (i, j, b) terms
    Cipher2d matmul(const Evaluator& evaluator, const Cipher2d& a, const Plain2d& w) const;
    // the same product with the roles exchanged (MatmulObjective::EncryptRight): plaintext inputs x encrypted weights
    Cipher2d matmul_reverse(const Evaluator& evaluator, const Plain2d& a, const Cipher2d& w) const;
    // both operands encrypted (MatmulObjective::Crossed): ciphertext products; as in the reference this needs a scheme whose
    // ciphertexts multiply in NTT form (CKKS / BGV) -- BFV operands in NTT form are refused by Evaluator::multiply
    Cipher2d matmul_cipher(const Evaluator& evaluator, const Cipher2d& a, const Cipher2d& w) const;
    // outputs [batch_size][output_dims] row-major, values mod t
    std::vector<uint64_t> decrypt_outputs_uint64s(const BatchEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;
    // a bias / share [batch_size][output_dims] laid out like the (packed or unpacked) outputs, for Cipher2d::add_plain_inplace
    Plain2d encode_outputs_uint64s(const BatchEncoder& encoder, const uint64_t* outputs) const;
    // CKKS (examples/11_ckks_matmul.cu): the same layouts on real coefficients through CKKSEncoder::encode_float64_polynomial; the
    // product carries scale^2, so a bias is encoded at scale^2 (and at the level of the outputs when they were mod-switched)
    Plain2d encode_weights_doubles(const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const;
    Plain2d encode_inputs_doubles(const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const;
    Cipher2d encrypt_inputs_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const;
    Cipher2d encrypt_weights_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const;
    Plain2d encode_outputs_doubles(const CKKSEncoder& encoder, const double* outputs, std::optional<ParmsID> parms_id, double scale) const;
    std::vector<double> decrypt_outputs_doubles(const CKKSEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;
    // matmul.h:40-51, :120-128: the weights (and the bias) encoded on the fly instead of being held as a Plain2d -- one input block of weight
    // plaintexts is alive at a time
    Cipher2d matmul_fly_uint64s(const BatchEncoder& encoder, const Evaluator& evaluator, const Cipher2d& inputs, const uint64_t* weights) const;
    Cipher2d matmul_fly_doubles(const CKKSEncoder& encoder, const Evaluator& evaluator, const Cipher2d& inputs, const double* weights, std::optional<ParmsID> parms_id, double scale) const;
    template <typename T> Cipher2d matmul_fly_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Evaluator& evaluator, const Cipher2d& inputs, const T* weights, std::optional<ParmsID> parms_id) const;
    void add_bias_inplace_fly_uint64s(const BatchEncoder& encoder, const Evaluator& evaluator, Cipher2d& multiplied, const uint64_t* bias) const {
        multiplied.add_plain_inplace(evaluator, encode_outputs_uint64s(encoder, bias), pool);
    }
    void add_bias_inplace_fly_doubles(const CKKSEncoder& encoder, const Evaluator& evaluator, Cipher2d& multiplied, const double* bias, std::optional<ParmsID> parms_id, double scale) const {
        multiplied.add_plain_inplace(evaluator, encode_outputs_doubles(encoder, bias, parms_id, scale), pool);
    }
    template <typename T> void add_bias_inplace_fly_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Evaluator& evaluator, Cipher2d& multiplied, const T* bias, std::optional<ParmsID> parms_id) const {
        multiplied.add_plain_inplace(evaluator, encode_outputs_ring2k(encoder, bias, parms_id), pool);
    }
    // Z_{2^k} matrices through PolynomialEncoderRing2k<T> (encoder_adapter.h:48-67: the encrypted operand is scaled up, the plaintext
    // operand centralized, outputs are scaled down from the undivided phase); the same block layouts
    template <typename T> Plain2d encode_weights_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id) const;
    template <typename T> Plain2d encode_inputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id) const;
    template <typename T> Cipher2d encrypt_inputs_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id) const;
    template <typename T> Cipher2d encrypt_weights_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id) const;
    template <typename T> Plain2d encode_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* outputs, std::optional<ParmsID> parms_id) const;
    template <typename T> std::vector<T> decrypt_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;
    // pack_lwe: input_block output ciphertexts are merged into one (Evaluator::pack_rlwe_ciphertexts, all groups batched);
    // the result is one row of ceil(#outputs / input_block) ciphertexts.  Needs the Galois keys of
    // (N / input_block) * 2^k + 1, k = 1..log2(input_block)
    Cipher2d pack_outputs(const Evaluator& evaluator, const GaloisKeys& auto_key, const Cipher2d& cipher) const;
    // app/matmul.cu:621-720: the weights in full; of the unpacked outputs only the coefficients that carry results
    void serialize_encoded_weights(const Plain2d& w, std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const;
    Plain2d deserialize_encoded_weights(std::istream& stream) const;
    void serialize_outputs(const Evaluator& evaluator, const Cipher2d& x, std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const;
    Cipher2d deserialize_outputs(const Evaluator& evaluator, std::istream& stream) const;

private:
    void determine_block();
    // block (r, c) of the weights / inputs / outputs as a coefficient vector, any element type
    template <typename T> std::vector<T> weight_block(const T* weights, size_t li, size_t lj) const {
        std::vector<T> vec(input_block * output_block, 0);
        for (size_t j = lj; j < std::min(lj + output_block, output_dims); j++)
            for (size_t i = li; i < std::min(li + input_block, input_dims); i++) vec[(j - lj) * input_block + input_block - (i - li) - 1] = weights[i * output_dims + j];
        return vec;
    }
    template <typename T> std::vector<T> input_block_vector(const T* inputs, size_t li, size_t lj) const {
        std::vector<T> vec(slot_count, 0);
        for (size_t i = li; i < std::min(li + batch_block, batch_size); i++)
            for (size_t j = lj; j < std::min(lj + input_block, input_dims); j++) vec[(i - li) * input_block * output_block + (j - lj)] = inputs[i * input_dims + j];
        return vec;
    }
    std::vector<uint64_t> pack_weight_blocks(uint64_t t, const uint64_t* weights, size_t& rows, size_t& cols, size_t& len) const;
    std::vector<uint64_t> pack_input_blocks(uint64_t t, const uint64_t* inputs, size_t& rows, size_t& cols, size_t& len) const;
};

// ---- ring-2^k forms (templates; the element type is the encoder's) --------------------------------------------------------------
template <typename T>
Plain2d MatmulHelper::encode_weights_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id) const {
    Plain2d out;
    for (size_t li = 0; li < input_dims; li += input_block) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t lj = 0; lj < output_dims; lj += output_block) {
            Plaintext p = encoder.centralize_new(weight_block(weights, li, lj), parms_id, pool);
            Evaluator(encoder.context()).transform_plain_to_ntt_inplace(p, p.parms_id(), pool);
            row.push_back(std::move(p));
        }
    }
    return out;
}

template <typename T>
Plain2d MatmulHelper::encode_inputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id) const {
    Plain2d out;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t lj = 0; lj < input_dims; lj += input_block) {
            Plaintext p = encoder.centralize_new(input_block_vector(inputs, li, lj), parms_id, pool);
            Evaluator(encoder.context()).transform_plain_to_ntt_inplace(p, p.parms_id(), pool);
            row.push_back(std::move(p));
        }
    }
    return out;
}

template <typename T>
Cipher2d MatmulHelper::encrypt_inputs_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* inputs, std::optional<ParmsID> parms_id) const {
    Evaluator evaluator(encoder.context());
    Cipher2d out;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        std::vector<Ciphertext>& row = out.new_row();
        for (size_t lj = 0; lj < input_dims; lj += input_block) {
            Plaintext p = encoder.scale_up_new(input_block_vector(inputs, li, lj), parms_id, pool);
            evaluator.transform_plain_to_ntt_inplace(p, p.parms_id(), pool);
            row.push_back(encryptor.encrypt_symmetric_new(p, true, pool));      // NTT form, c1 kept as its seed (app/matmul.cu:311)
        }
    }
    return out;
}

template <typename T>
Cipher2d MatmulHelper::encrypt_weights_ring2k(const Encryptor& encryptor, const PolynomialEncoderRing2k<T>& encoder, const T* weights, std::optional<ParmsID> parms_id) const {
    Evaluator evaluator(encoder.context());
    Cipher2d out;
    for (size_t li = 0; li < input_dims; li += input_block) {
        std::vector<Ciphertext>& row = out.new_row();
        for (size_t lj = 0; lj < output_dims; lj += output_block) {
            Plaintext p = encoder.scale_up_new(weight_block(weights, li, lj), parms_id, pool);
            evaluator.transform_plain_to_ntt_inplace(p, p.parms_id(), pool);
            row.push_back(encryptor.encrypt_symmetric_new(p, true, pool));
        }
    }
    return out;
}

template <typename T>
Plain2d MatmulHelper::encode_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const T* outputs, std::optional<ParmsID> parms_id) const {
    const size_t n = slot_count, ocols = (output_dims + output_block - 1) / output_block, brows = (batch_size + batch_block - 1) / batch_block;
    const size_t count = pack_lwe ? (brows * ocols + input_block - 1) / input_block : brows * ocols;
    std::vector<std::vector<T>> buffers(count, std::vector<T>(n, 0));
    size_t di = 0;
    for (size_t li = 0; li < batch_size; li += batch_block, di++) {
        size_t dj = 0;
        for (size_t lj = 0; lj < output_dims; lj += output_block, dj++) {
            const size_t cipher_id = di * ocols + dj;
            std::vector<T>& buf = buffers[pack_lwe ? cipher_id / input_block : cipher_id];
            const size_t offset = pack_lwe ? cipher_id % input_block : input_block - 1;
            for (size_t i = li; i < std::min(li + batch_block, batch_size); i++)
                for (size_t j = lj; j < std::min(lj + output_block, output_dims); j++)
                    buf[(i - li) * input_block * output_block + (j - lj) * input_block + offset] = outputs[i * output_dims + j];
        }
    }
    Plain2d out;
    if (pack_lwe) {
        std::vector<Plaintext>& row = out.new_row();
        for (const auto& buf : buffers) row.push_back(encoder.scale_up_new(buf, parms_id, pool));
    } else {
        for (size_t r = 0; r < brows; r++) {
            std::vector<Plaintext>& row = out.new_row();
            for (size_t c = 0; c < ocols; c++) row.push_back(encoder.scale_up_new(buffers[r * ocols + c], parms_id, pool));
        }
    }
    return out;
}

template <typename T>
std::vector<T> MatmulHelper::decrypt_outputs_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    std::vector<std::vector<T>> coeffs;
    for (const auto& r : outputs.data())
        for (const Ciphertext& c : r) coeffs.push_back(encoder.scale_down_new(decryptor.bfv_decrypt_without_scaling_down_new(c, pool), pool));
    const size_t ocols = (output_dims + output_block - 1) / output_block, brows = (batch_size + batch_block - 1) / batch_block;
    if (coeffs.size() != (pack_lwe ? (brows * ocols + input_block - 1) / input_block : brows * ocols))
        throw std::invalid_argument("[MatmulHelper::decrypt_outputs] Output ciphertext count incorrect");
    std::vector<T> out(batch_size * output_dims, 0);
    size_t di = 0;
    for (size_t li = 0; li < batch_size; li += batch_block, di++) {
        size_t dj = 0;
        for (size_t lj = 0; lj < output_dims; lj += output_block, dj++) {
            const size_t cipher_id = di * ocols + dj;
            const std::vector<T>& cf = coeffs[pack_lwe ? cipher_id / input_block : cipher_id];
            const size_t offset = pack_lwe ? cipher_id % input_block : input_block - 1;
            for (size_t i = li; i < std::min(li + batch_block, batch_size); i++)
                for (size_t j = lj; j < std::min(lj + output_block, output_dims); j++)
                    out[i * output_dims + j] = cf[(i - li) * input_block * output_block + (j - lj) * input_block + offset];
        }
    }
    return out;
}

template <typename T>
Cipher2d MatmulHelper::matmul_fly_ring2k(const PolynomialEncoderRing2k<T>& encoder, const Evaluator& evaluator, const Cipher2d& inputs, const T* weights, std::optional<ParmsID> parms_id) const {
    const size_t batch_split = (batch_size + batch_block - 1) / batch_block, input_split = (input_dims + input_block - 1) / input_block, output_split = (output_dims + output_block - 1) / output_block;
    return detail::accumulate_products_fly(evaluator, inputs, batch_split, input_split, output_split, [&](size_t i) {
        std::vector<Plaintext> row;
        for (size_t j = 0; j < output_split; j++) {
            Plaintext p = encoder.centralize_new(weight_block(weights, i * input_block, j * output_block), parms_id, pool);
            evaluator.transform_plain_to_ntt_inplace(p, p.parms_id(), pool);
            row.push_back(std::move(p));
        }
        return row;
    }, pool);
}

inline std::ostream& operator<<(std::ostream& os, const MatmulHelper& h) {               // matmul.cu:6-10
    return os << "MatmulHelper(batch_size=" << h.batch_size << ", input_dims=" << h.input_dims << ", output_dims=" << h.output_dims << ", slot_count=" << h.slot_count
              << ", objective=" << h.objective << ", pack_lwe=" << h.pack_lwe << ")";
}

}}  // namespace troy::linear
