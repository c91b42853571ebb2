// matmul.cpp -- see matmul.h.  Host logic only; all arithmetic goes through the mirror's Evaluator (GPU).
#include "matmul.h"

namespace troy { namespace linear {

static size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

MatmulHelper::MatmulHelper(size_t batch_size, size_t input_dims, size_t output_dims, size_t slot_count, MatmulObjective objective, bool pack_lwe,
                           MemoryPoolHandle pool)
    : batch_size(batch_size), input_dims(input_dims), output_dims(output_dims), slot_count(slot_count), objective(objective), pack_lwe(pack_lwe),
      pool(std::move(pool)) {
    if (pack_lwe) throw std::logic_error("[MatmulHelper::MatmulHelper] LWE output packing is not part of this build.");
    if (objective != MatmulObjective::EncryptLeft) throw std::logic_error("[MatmulHelper::MatmulHelper] only MatmulObjective::EncryptLeft is part of this build.");
    determine_block();
}

void MatmulHelper::determine_block() {
    // app/matmul.cu:101-127 (no LWE packing): choose (bb, ib, ob), bb*ib*ob <= N, minimising the number of ciphertexts that
    // travel: encrypted inputs ceil(B/bb)*ceil(I/ib) plus encrypted outputs ceil(B/bb)*ceil(O/ob)
    size_t best_cost = static_cast<size_t>(-1);
    for (size_t bb = std::min(batch_size, slot_count - 1); bb >= 1; bb--) {
        const size_t bc = ceil_div(batch_size, bb);
        if (2 * bc > best_cost) continue;
        for (size_t ib = 1; ib <= input_dims && ib < slot_count / bb; ib++) {
            size_t ob = std::min(slot_count / bb / ib, output_dims);
            if (ob < 1) continue;
            const size_t cost = bc * (ceil_div(input_dims, ib) + ceil_div(output_dims, ob));
            if (cost < best_cost) { best_cost = cost; batch_block = bb; input_block = ib; output_block = ob; }
        }
    }
    if (best_cost == static_cast<size_t>(-1)) throw std::invalid_argument("[MatmulHelper::determine_block] no valid blocking for these dimensions.");
}

Plain2d MatmulHelper::encode_weights_uint64s(const BatchEncoder& encoder, const uint64_t* weights) const {
    // app/matmul.cu:160-230: coefficient packing, then centralize + NTT (the weights multiply ciphertexts).  All blocks are
    // packed on the host into one buffer and go through ONE copy, ONE centralize launch and ONE NTT launch; the plaintext
    // objects are windows of the shared result buffer.
    HeContextPointer context = encoder.context();
    if (!context->on_device()) throw std::invalid_argument("[MatmulHelper::encode_weights] HeContext is not on device (call to_device_inplace).");
    ContextDataPointer cd = context->first_context_data().value();
    const ParmsID first = context->first_parms_id();
    const size_t n = cd->parms().poly_modulus_degree(), L = cd->parms().coeff_modulus().size();
    const uint64_t t = cd->parms().plain_modulus().value();
    const size_t rows = ceil_div(input_dims, input_block), cols = ceil_div(output_dims, output_block), count = rows * cols;
    const size_t clen = input_block * output_block;
    std::vector<uint64_t> packed(count * clen, 0);
    size_t idx = 0;
    for (size_t li = 0; li < input_dims; li += input_block) {
        const size_t ui = std::min(li + input_block, input_dims);
        for (size_t lj = 0; lj < output_dims; lj += output_block, idx++) {
            const size_t uj = std::min(lj + output_block, output_dims);
            uint64_t* vec = packed.data() + idx * clen;
            for (size_t j = lj; j < uj; j++)
                for (size_t i = li; i < ui; i++) {
                    const uint64_t v = weights[i * output_dims + j];
                    if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus");
                    vec[(j - lj) * input_block + input_block - (i - li) - 1] = v;
                }
        }
    }
    utils::DynamicArray staged(packed.size(), true, pool);
    staged.copy_from(packed.data(), packed.size(), false);
    auto shared = std::make_shared<utils::DynamicArray>(count * L * n, true, pool);
    troyn_check_public(troyn_plain_centralize(context->plan(), static_cast<uint32_t>(L), t, staged.raw_pointer(), clen, clen, shared->raw_pointer(), count, troyn_current_stream()));
    troyn_check_public(troyn_ntt(context->plan(), 0, shared->raw_pointer(), shared->raw_pointer(), count, 1, L, 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0,
                                 troyn_current_stream()));
    troyn_sync_current_stream();     // `staged` returns to the pool
    Plain2d out;
    idx = 0;
    for (size_t r = 0; r < rows; r++) {
        std::vector<Plaintext> row;
        for (size_t c = 0; c < cols; c++, idx++) {
            Plaintext p;
            p.data() = utils::DynamicArray::device_view(shared->raw_pointer() + idx * L * n, L * n, shared);
            p.parms_id() = first;
            p.coeff_count() = n;
            p.coeff_modulus_size() = L;
            p.poly_modulus_degree() = n;
            p.is_ntt_form() = true;
            row.push_back(std::move(p));
        }
        out.data().push_back(std::move(row));
    }
    return out;
}

Plain2d MatmulHelper::encode_inputs_uint64s(const BatchEncoder& encoder, const uint64_t* inputs) const {
    // app/matmul.cu:245-300
    Plain2d out;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        const size_t ui = std::min(li + batch_block, batch_size);
        std::vector<Plaintext> row;
        for (size_t lj = 0; lj < input_dims; lj += input_block) {
            const size_t uj = std::min(lj + input_block, input_dims);
            std::vector<uint64_t> vec(slot_count, 0);
            for (size_t j = lj; j < uj; j++)
                for (size_t i = li; i < ui; i++) vec[(i - li) * input_block * output_block + (j - lj)] = inputs[i * input_dims + j];
            row.push_back(encoder.encode_polynomial_new(vec, pool));
        }
        out.data().push_back(std::move(row));
    }
    return out;
}

Cipher2d MatmulHelper::encrypt_inputs_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* inputs) const {
    // every block packed on the host, one copy, one batched symmetric encryption, one batched NTT (the same ciphertexts
    // as encode_inputs_uint64s + encrypt_symmetric per block, see Encryptor::encrypt_symmetric_packed)
    HeContextPointer context = encoder.context();
    const size_t n = slot_count;
    const uint64_t t = context->first_context_data().value()->parms().plain_modulus().value();
    const size_t rows = ceil_div(batch_size, batch_block), cols = ceil_div(input_dims, input_block), count = rows * cols;
    std::vector<uint64_t> packed(count * n, 0);
    size_t idx = 0;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        const size_t ui = std::min(li + batch_block, batch_size);
        for (size_t lj = 0; lj < input_dims; lj += input_block, idx++) {
            const size_t uj = std::min(lj + input_block, input_dims);
            uint64_t* vec = packed.data() + idx * n;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) {
                    const uint64_t v = inputs[i * input_dims + j];
                    if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus");
                    vec[(i - li) * input_block * output_block + (j - lj)] = v;
                }
        }
    }
    utils::DynamicArray staged(packed.size(), true, pool);
    staged.copy_from(packed.data(), packed.size(), false);
    std::vector<Ciphertext> cts = encryptor.encrypt_symmetric_packed(staged.raw_pointer(), n, n, count, pool);
    // the products are taken in NTT form: the ciphertexts are windows of one buffer, transformed in one launch
    const uint32_t L = static_cast<uint32_t>(cts[0].coeff_modulus_size());
    troyn_check_public(troyn_ntt(context->plan(), 0, cts[0].data().raw_pointer(), cts[0].data().raw_pointer(), count, 2, L, 0, L, TROYN_IDX_COMPONENTWISE, 0,
                                 troyn_current_stream()));
    for (Ciphertext& c : cts) c.is_ntt_form() = true;
    troyn_sync_current_stream();
    Cipher2d out;
    idx = 0;
    for (size_t r = 0; r < rows; r++) {
        std::vector<Ciphertext> row;
        for (size_t c = 0; c < cols; c++, idx++) row.push_back(std::move(cts[idx]));
        out.data().push_back(std::move(row));
    }
    return out;
}

Cipher2d MatmulHelper::matmul(const Evaluator& evaluator, const Cipher2d& a, const Plain2d& w) const {
    // app/matmul.cu:326-374, batched form
    const size_t batch_split = ceil_div(batch_size, batch_block), input_split = ceil_div(input_dims, input_block), output_split = ceil_div(output_dims, output_block);
    if (a.size() != batch_split) throw std::invalid_argument("[MatmulHelper::matmul] Input batch_size incorrect.");
    if (w.size() != input_split) throw std::invalid_argument("[MatmulHelper::matmul] Weight input dimension incorrect.");
    // the results are windows of one zeroed buffer, so the trailing inverse NTT is one launch
    const Ciphertext& a0 = a[0][0];
    const size_t pcnt = a0.polynomial_count(), L = a0.coeff_modulus_size(), n = a0.poly_modulus_degree(), words = pcnt * L * n;
    auto shared = std::make_shared<utils::DynamicArray>(batch_split * output_split * words, true, pool);
    shared->set_zero();
    Cipher2d ret;
    ret.data().resize(batch_split);
    for (size_t b = 0; b < batch_split; b++)
        for (size_t j = 0; j < output_split; j++)
            ret[b].push_back(Ciphertext::from_members(pcnt, L, n, a0.parms_id(), a0.scale(), true, a0.correction_factor(), 0,
                                                      utils::DynamicArray::device_view(shared->raw_pointer() + (b * output_split + j) * words, words, shared)));
    std::vector<const Ciphertext*> a_ptrs;
    std::vector<const Plaintext*> w_ptrs;
    std::vector<Ciphertext*> r_ptrs;
    for (size_t i = 0; i < input_split; i++)
        for (size_t j = 0; j < output_split; j++)
            for (size_t b = 0; b < batch_split; b++) { a_ptrs.push_back(&a[b][i]); w_ptrs.push_back(&w[i][j]); r_ptrs.push_back(&ret[b][j]); }
    evaluator.multiply_plain_accumulate(a_ptrs, w_ptrs, r_ptrs, false, pool);
    // BFV results leave in coefficient form
    troyn_check_public(troyn_ntt(evaluator.context()->plan(), 1, shared->raw_pointer(), shared->raw_pointer(), batch_split * output_split, pcnt,
                                 static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, troyn_current_stream()));
    troyn_sync_current_stream();
    for (auto& r : ret.data()) for (Ciphertext& c : r) c.is_ntt_form() = false;
    return ret;
}

std::vector<uint64_t> MatmulHelper::decrypt_outputs_uint64s(const BatchEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    // app/matmul.cu:560-640; all output ciphertexts decrypted as one batch and read back with one copy
    (void)encoder;
    std::vector<const Ciphertext*> all;
    for (const auto& r : outputs.data()) for (const Ciphertext& c : r) all.push_back(&c);
    const std::vector<uint64_t> coeffs = decryptor.bfv_decrypt_to_host(all, pool);
    const size_t n = slot_count;
    std::vector<uint64_t> out(batch_size * output_dims, 0);
    size_t idx = 0;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        const size_t ui = std::min(li + batch_block, batch_size);
        for (size_t lj = 0; lj < output_dims; lj += output_block, idx++) {
            const size_t uj = std::min(lj + output_block, output_dims);
            const uint64_t* cf = coeffs.data() + idx * n;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) out[i * output_dims + j] = cf[(i - li) * input_block * output_block + (j - lj) * input_block + input_block - 1];
        }
    }
    return out;
}

}}  // namespace troy::linear
