// matmul.h -- troy::linear::MatmulHelper, the BFV matrix product of the reference's matmul application
// (src/app/matmul.h, matmul.cu; BASELINE config 5 = examples/10_bfv_matmul.cu at 512x512x512) on top of the mirror API.
//
// Supported here: BFV with BatchEncoder polynomial (coefficient) packing, encrypted inputs x plaintext weights
// (MatmulObjective::EncryptLeft, the example's configuration), without LWE output packing.  Layout, as the reference:
//   input block  (batch rows li..ui, input columns lj..uj):   coefficient (i-li)*ib*ob + (j-lj)              = x[i][j]
//   weight block (input rows li..ui, output columns lj..uj):  coefficient (j-lj)*ib + ib - (i-li) - 1         = w[i][j]
//   output block: y[i][j] is coefficient (i-li)*ib*ob + (j-lj)*ib + ib - 1 of sum_k input[b][k] * weight[k][j-block]
// with (bb, ib, ob) = (batch_block, input_block, output_block), bb*ib*ob <= N.
#pragma once
#include "troy.h"

namespace troy { namespace linear {

enum class MatmulObjective : uint8_t { EncryptLeft = 0, EncryptRight = 1, Crossed = 2 };

class Plain2d {
public:
    std::vector<std::vector<Plaintext>>& data() { return inner; }
    const std::vector<std::vector<Plaintext>>& data() const { return inner; }
    size_t size() const { return inner.size(); }
    std::vector<Plaintext>& operator[](size_t i) { return inner[i]; }
    const std::vector<Plaintext>& operator[](size_t i) const { return inner[i]; }
private:
    std::vector<std::vector<Plaintext>> inner;
};

class Cipher2d {
public:
    std::vector<std::vector<Ciphertext>>& data() { return inner; }
    const std::vector<std::vector<Ciphertext>>& data() const { return inner; }
    size_t size() const { return inner.size(); }
    std::vector<Ciphertext>& operator[](size_t i) { return inner[i]; }
    const std::vector<Ciphertext>& operator[](size_t i) const { return inner[i]; }
private:
    std::vector<std::vector<Ciphertext>> inner;
};

class MatmulHelper {
public:
    size_t batch_size, input_dims, output_dims, slot_count;
    size_t batch_block = 0, input_block = 0, output_block = 0;
    MatmulObjective objective;
    bool pack_lwe;
    MemoryPoolHandle pool;

    MatmulHelper(size_t batch_size, size_t input_dims, size_t output_dims, size_t slot_count,
                 MatmulObjective objective = MatmulObjective::EncryptLeft, bool pack_lwe = false, MemoryPoolHandle pool = MemoryPool::GlobalPool());

    // weights [input_dims][output_dims] row-major -> NTT-form plaintexts [ceil(in/ib)][ceil(out/ob)]
    Plain2d encode_weights_uint64s(const BatchEncoder& encoder, const uint64_t* weights) const;
    // inputs [batch_size][input_dims] row-major -> plaintexts / ciphertexts [ceil(batch/bb)][ceil(in/ib)]
    Plain2d encode_inputs_uint64s(const BatchEncoder& encoder, const uint64_t* inputs) const;
    Cipher2d encrypt_inputs_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* inputs) const;
    // ret[b][j] = sum_i a[b][i] * w[i][j]: ONE multiply_plain_accumulate launch over all (i, j, b) terms
    Cipher2d matmul(const Evaluator& evaluator, const Cipher2d& a, const Plain2d& w) const;
    // outputs [batch_size][output_dims] row-major, values mod t
    std::vector<uint64_t> decrypt_outputs_uint64s(const BatchEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const;

private:
    void determine_block();
};

}}  // namespace troy::linear
