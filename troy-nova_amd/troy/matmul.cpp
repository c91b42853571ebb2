// matmul.cpp -- see matmul.h.  Host logic only; all arithmetic goes through the mirror's Evaluator (GPU).
#include "matmul.h"

#include <cmath>

namespace troy { namespace linear {

static size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// Plain2d / Cipher2d  (app/cipher2d.h, cipher2d.cu)
// ------------------------------------------------------------------------------------------------
namespace {
void put_size(std::ostream& os, size_t v) { os.write(reinterpret_cast<const char*>(&v), sizeof(v)); }
size_t get_size(std::istream& is) {
    size_t v = 0;
    is.read(reinterpret_cast<char*>(&v), sizeof(v));
    if (!is) throw std::runtime_error("[serialize::load_object] unexpected end of stream");
    return v;
}
}  // namespace

size_t Plain2d::save(std::ostream& stream, CompressionMode mode) const {
    size_t bytes = sizeof(size_t);
    put_size(stream, rows());
    for (const auto& row : inner) {
        put_size(stream, row.size());
        bytes += sizeof(size_t);
        for (const Plaintext& p : row) bytes += p.save(stream, mode);
    }
    return bytes;
}

void Plain2d::load(std::istream& stream, MemoryPoolHandle pool) {
    inner.clear();
    const size_t rows = get_size(stream);
    for (size_t i = 0; i < rows; i++) {
        const size_t count = get_size(stream);
        std::vector<Plaintext>& row = new_row();
        for (size_t j = 0; j < count; j++) row.push_back(Plaintext::load_new(stream, pool));
    }
}

Cipher2d Plain2d::encrypt_symmetric(const Encryptor& encryptor, MemoryPoolHandle pool) const {
    Cipher2d out;
    for (const auto& row : inner) { auto& r = out.new_row(); for (const Plaintext& p : row) r.push_back(encryptor.encrypt_symmetric_new(p, false, pool)); }
    return out;
}

Cipher2d Plain2d::encrypt_asymmetric(const Encryptor& encryptor, MemoryPoolHandle pool) const {
    Cipher2d out;
    for (const auto& row : inner) { auto& r = out.new_row(); for (const Plaintext& p : row) r.push_back(encryptor.encrypt_asymmetric_new(p, pool)); }
    return out;
}

void Cipher2d::relinearize_inplace(const Evaluator& evaluator, const RelinKeys& relin_keys, MemoryPoolHandle pool) {
    std::vector<const Ciphertext*> src;
    std::vector<Ciphertext*> dst;
    for (auto& row : inner) for (Ciphertext& c : row) { src.push_back(&c); dst.push_back(&c); }
    evaluator.relinearize_batched(src, relin_keys, dst, pool);
}

Plain2d Cipher2d::decrypt(const Decryptor& decryptor, MemoryPoolHandle pool) const {
    Plain2d out;
    for (const auto& row : inner) { auto& r = out.new_row(); for (const Ciphertext& c : row) r.push_back(decryptor.decrypt_new(c, pool)); }
    return out;
}

Cipher2d Cipher2d::clone(MemoryPoolHandle pool) const {
    Cipher2d out;
    for (const auto& row : inner) {
        std::vector<Ciphertext>& r = out.new_row();
        for (const Ciphertext& c : row) r.push_back(c.clone(pool));
    }
    return out;
}

void Cipher2d::expand_seed(HeContextPointer context) {
    for (auto& row : inner) for (Ciphertext& c : row) if (c.contains_seed()) c.expand_seed(context);
}

size_t Cipher2d::save(std::ostream& stream, HeContextPointer context, CompressionMode mode) const {
    // app/cipher2d.cu: rows, then per row its length and its ciphertexts; the ciphertexts of a row leave as one batch (Ciphertext::save_many)
    put_size(stream, rows());
    size_t bytes = sizeof(size_t);
    for (const auto& row : inner) {
        put_size(stream, row.size());
        std::vector<const Ciphertext*> ptrs;
        for (const Ciphertext& c : row) ptrs.push_back(&c);
        bytes += sizeof(size_t) + Ciphertext::save_many(stream, ptrs.data(), ptrs.size(), context, mode);
    }
    return bytes;      // what was written (with Zstd: less than serialized_size_upperbound)
}

void Cipher2d::load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool) {
    inner.clear();
    const size_t rows = get_size(stream);
    for (size_t i = 0; i < rows; i++) {
        const size_t count = get_size(stream);
        if (count > (size_t(1) << 24)) throw std::runtime_error("[Cipher2d::load] invalid row length");
        std::vector<Ciphertext>& row = new_row();
        row.resize(count);
        std::vector<Ciphertext*> ptrs;
        for (Ciphertext& c : row) ptrs.push_back(&c);
        Ciphertext::load_many(stream, ptrs.data(), ptrs.size(), context, pool);
    }
}

size_t Cipher2d::serialized_size_upperbound(HeContextPointer context, CompressionMode mode) const {
    size_t bytes = sizeof(size_t);
    for (const auto& row : inner) {
        bytes += sizeof(size_t);
        for (const Ciphertext& c : row) bytes += c.serialized_size_upperbound(context, mode);
    }
    return bytes;
}

void Cipher2d::mod_switch_to_next_inplace(const Evaluator& evaluator, MemoryPoolHandle pool) {
    std::vector<const Ciphertext*> src;
    std::vector<Ciphertext*> dst;
    for (auto& row : inner) for (Ciphertext& c : row) { src.push_back(&c); dst.push_back(&c); }
    evaluator.mod_switch_to_next_batched(src, dst, pool);
}

void Cipher2d::translate_inplace(const Evaluator& evaluator, const Cipher2d& other, bool subtract, MemoryPoolHandle pool) {
    if (size() != other.size()) throw std::runtime_error("[Cipher2d::translate_inplace] Row size mismatch.");
    std::vector<const Ciphertext*> a, b;
    std::vector<Ciphertext*> d;
    for (size_t i = 0; i < rows(); i++) {
        if (inner[i].size() != other[i].size()) throw std::runtime_error("[Cipher2d::translate_inplace] Column size mismatch.");
        for (size_t j = 0; j < inner[i].size(); j++) { a.push_back(&inner[i][j]); b.push_back(&other[i][j]); d.push_back(&inner[i][j]); }
    }
    if (subtract) evaluator.sub_batched(a, b, d, pool); else evaluator.add_batched(a, b, d, pool);
}

void Cipher2d::translate_plain_inplace(const Evaluator& evaluator, const Plain2d& other, bool subtract, MemoryPoolHandle pool) {
    if (size() != other.size()) throw std::runtime_error("[Cipher2d::translate_plain_inplace] Row size mismatch.");
    std::vector<const Ciphertext*> a;
    std::vector<const Plaintext*> b;
    std::vector<Ciphertext*> d;
    for (size_t i = 0; i < rows(); i++) {
        if (inner[i].size() != other[i].size()) throw std::runtime_error("[Cipher2d::translate_plain_inplace] Column size mismatch.");
        for (size_t j = 0; j < inner[i].size(); j++) { a.push_back(&inner[i][j]); b.push_back(&other[i][j]); d.push_back(&inner[i][j]); }
    }
    if (subtract) evaluator.sub_plain_batched(a, b, d, pool); else evaluator.add_plain_batched(a, b, d, pool);
}

// ------------------------------------------------------------------------------------------------
// MatmulHelper  (app/matmul.h, matmul.cu)
// ------------------------------------------------------------------------------------------------
MatmulHelper::MatmulHelper(size_t batch_size, size_t input_dims, size_t output_dims, size_t slot_count, MatmulObjective objective, bool pack_lwe,
                           MemoryPoolHandle pool)
    : batch_size(batch_size), input_dims(input_dims), output_dims(output_dims), slot_count(slot_count), objective(objective), pack_lwe(pack_lwe),
      pool(std::move(pool)) {
    determine_block();
}

void MatmulHelper::determine_block() {
    size_t best_cost = static_cast<size_t>(-1);
    if (!pack_lwe) {
        // app/matmul.cu:103-127: choose (bb, ib, ob), bb*ib*ob <= N, minimising the number of ciphertexts that travel (which
        // operands are encrypted depends on the objective; the outputs always are)
        for (size_t bb = std::min(batch_size, slot_count - 1); bb >= 1; bb--) {
            const size_t bc = ceil_div(batch_size, bb);
            if (2 * bc > best_cost) continue;
            for (size_t ib = 1; ib <= input_dims && ib < slot_count / bb; ib++) {
                size_t ob = std::min(slot_count / bb / ib, output_dims);
                if (ob < 1) continue;
                size_t cost;
                if (objective == MatmulObjective::EncryptLeft) cost = bc * (ceil_div(input_dims, ib) + ceil_div(output_dims, ob));
                else if (objective == MatmulObjective::EncryptRight) cost = (bc + ceil_div(input_dims, ib)) * ceil_div(output_dims, ob);
                else cost = bc * input_dims + (bc + ceil_div(input_dims, ib)) * ceil_div(output_dims, ob);
                if (cost < best_cost) { best_cost = cost; batch_block = bb; input_block = ib; output_block = ob; }
            }
        }
    } else {
        // app/matmul.cu:128-158: the input block is a power of two near N^(1/3) (the packing tree merges input_block
        // outputs into one ciphertext); the outputs then cost ceil(#outputs / ib) ciphertexts
        const double cube = std::pow(static_cast<double>(slot_count), 0.33);
        size_t ib = 1;
        while (static_cast<double>(ib * 2) < cube) ib *= 2;
        if (ib > input_dims) { ib = 1; while (ib < input_dims) ib *= 2; }
        for (size_t bb = 1; bb <= batch_size; bb++) {
            const size_t bc = ceil_div(batch_size, bb);
            if (bb > slot_count) continue;
            size_t ob = std::min(slot_count / bb / ib, output_dims);
            if (ob < 1) continue;
            const size_t packed_out = ceil_div(bc * ceil_div(output_dims, ob), ib);
            size_t cost;
            if (objective == MatmulObjective::EncryptLeft) cost = bc * ceil_div(input_dims, ib) + packed_out;
            else if (objective == MatmulObjective::EncryptRight) cost = ceil_div(output_dims, ob) * ceil_div(input_dims, ib) + packed_out;
            else cost = bc * ceil_div(input_dims, ib) + ceil_div(output_dims, ob) * ceil_div(input_dims, ib) + packed_out;
            if (cost < best_cost) { best_cost = cost; batch_block = bb; input_block = ib; output_block = ob; }
        }
    }
    if (best_cost == static_cast<size_t>(-1)) throw std::invalid_argument("[MatmulHelper::determine_block] no valid blocking for these dimensions.");
}

std::vector<uint64_t> MatmulHelper::pack_weight_blocks(uint64_t t, const uint64_t* weights, size_t& rows, size_t& cols, size_t& len) const {
    // app/matmul.cu:160-175 (encode_weights_small): coefficient (j-lj)*ib + ib - (i-li) - 1 = w[i][j]
    rows = ceil_div(input_dims, input_block); cols = ceil_div(output_dims, output_block); len = input_block * output_block;
    std::vector<uint64_t> packed(rows * cols * len, 0);
    size_t idx = 0;
    for (size_t li = 0; li < input_dims; li += input_block) {
        const size_t ui = std::min(li + input_block, input_dims);
        for (size_t lj = 0; lj < output_dims; lj += output_block, idx++) {
            const size_t uj = std::min(lj + output_block, output_dims);
            uint64_t* vec = packed.data() + idx * len;
            for (size_t j = lj; j < uj; j++)
                for (size_t i = li; i < ui; i++) {
                    const uint64_t v = weights[i * output_dims + j];
                    if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus");
                    vec[(j - lj) * input_block + input_block - (i - li) - 1] = v;
                }
        }
    }
    return packed;
}

std::vector<uint64_t> MatmulHelper::pack_input_blocks(uint64_t t, const uint64_t* inputs, size_t& rows, size_t& cols, size_t& len) const {
    // app/matmul.cu:245-258 (encode_inputs_small): coefficient (i-li)*ib*ob + (j-lj) = x[i][j]
    rows = ceil_div(batch_size, batch_block); cols = ceil_div(input_dims, input_block); len = slot_count;
    std::vector<uint64_t> packed(rows * cols * len, 0);
    size_t idx = 0;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        const size_t ui = std::min(li + batch_block, batch_size);
        for (size_t lj = 0; lj < input_dims; lj += input_block, idx++) {
            const size_t uj = std::min(lj + input_block, input_dims);
            uint64_t* vec = packed.data() + idx * len;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) {
                    const uint64_t v = inputs[i * input_dims + j];
                    if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus");
                    vec[(i - li) * input_block * output_block + (j - lj)] = v;
                }
        }
    }
    return packed;
}

Plain2d detail::encode_blocks_for_plain(const BatchEncoder& encoder, const std::vector<uint64_t>& packed, size_t rows, size_t cols, size_t len, MemoryPoolHandle pool) {
    // encode_for_plain + ensure_ntt_form(centralize = true) (app/matmul.cu:14-70,:200-203): all blocks go through ONE copy and ONE launch
    // (the forward transform centralizes while it loads; of a block only its `len` coefficients are read); the plaintext objects are windows of the shared result buffer
    HeContextPointer context = encoder.context();
    if (!context->on_device()) throw std::invalid_argument("[MatmulHelper::encode_weights] HeContext is not on device (call to_device_inplace).");
    ContextDataPointer cd = context->first_context_data().value();
    const ParmsID first = context->first_parms_id();
    const size_t n = cd->parms().poly_modulus_degree(), L = cd->parms().coeff_modulus().size(), count = rows * cols;
    const uint64_t t = cd->parms().plain_modulus().value();
    utils::DynamicArray staged(packed.size(), true, pool);
    staged.copy_from(packed.data(), packed.size(), false);
    auto shared = std::make_shared<utils::DynamicArray>(count * L * n, true, pool);
    troyn_check_public(troyn_plain_centralize_ntt(context->plan(), static_cast<uint32_t>(L), t, staged.raw_pointer(), len, len, shared->raw_pointer(), count, troyn_current_stream()));
    troyn_sync_current_stream();     // `staged` returns to the pool
    Plain2d out;
    size_t idx = 0;
    for (size_t r = 0; r < rows; r++) {
        std::vector<Plaintext>& row = out.new_row();
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
    }
    return out;
}

Cipher2d detail::encrypt_blocks(const Encryptor& encryptor, const BatchEncoder& encoder, const std::vector<uint64_t>& packed, size_t rows, size_t cols, size_t len,
                                MemoryPoolHandle pool) {
    // encode_for_cipher + ensure_ntt_form(centralize = false) + encrypt_symmetric_batched(save_seed = true) (app/matmul.cu:200-216,
    // :296-311): one copy and one batched encryption that stays in NTT form; the ciphertexts carry the seed of c1
    std::vector<Ciphertext> cts;
    if (encoder.context()->first_context_data().value()->parms().scheme() == SchemeType::BFV) {
        utils::DynamicArray staged(packed.size(), true, pool);
        staged.copy_from(packed.data(), packed.size(), false);
        cts = encryptor.encrypt_symmetric_packed(staged.raw_pointer(), len, len, rows * cols, pool, true);
    } else {
        // BGV: block by block through the encoder and Encryptor::encrypt_symmetric (NTT-form ciphertexts, no seed)
        for (size_t k = 0; k < rows * cols; k++) {
            const std::vector<uint64_t> block(packed.begin() + static_cast<std::ptrdiff_t>(k * len), packed.begin() + static_cast<std::ptrdiff_t>((k + 1) * len));
            cts.push_back(encryptor.encrypt_symmetric_new(encoder.encode_polynomial_new(block, pool), false, pool));
        }
    }
    Cipher2d out;
    size_t idx = 0;
    for (size_t r = 0; r < rows; r++) {
        std::vector<Ciphertext>& row = out.new_row();
        for (size_t c = 0; c < cols; c++, idx++) row.push_back(std::move(cts[idx]));
    }
    return out;
}

Plain2d MatmulHelper::encode_weights_uint64s(const BatchEncoder& encoder, const uint64_t* weights) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_weight_blocks(encoder.context()->first_context_data().value()->parms().plain_modulus().value(), weights, rows, cols, len);
    return detail::encode_blocks_for_plain(encoder, packed, rows, cols, len, pool);
}

Cipher2d MatmulHelper::encrypt_weights_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* weights) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_weight_blocks(encoder.context()->first_context_data().value()->parms().plain_modulus().value(), weights, rows, cols, len);
    return detail::encrypt_blocks(encryptor, encoder, packed, rows, cols, len, pool);
}

Plain2d MatmulHelper::encode_inputs_uint64s(const BatchEncoder& encoder, const uint64_t* inputs) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_input_blocks(encoder.context()->first_context_data().value()->parms().plain_modulus().value(), inputs, rows, cols, len);
    return detail::encode_blocks_for_plain(encoder, packed, rows, cols, len, pool);
}

Cipher2d MatmulHelper::encrypt_inputs_uint64s(const Encryptor& encryptor, const BatchEncoder& encoder, const uint64_t* inputs) const {
    size_t rows, cols, len;
    const std::vector<uint64_t> packed = pack_input_blocks(encoder.context()->first_context_data().value()->parms().plain_modulus().value(), inputs, rows, cols, len);
    return detail::encrypt_blocks(encryptor, encoder, packed, rows, cols, len, pool);
}

// ret[b][j] = sum_i ct(b, i, j) (.) pt(b, i, j) in one multiply_plain_accumulate launch; the results are windows of one zeroed
// buffer, so the trailing inverse NTT (BFV results leave in coefficient form) is one launch too
Cipher2d detail::accumulate_products(const Evaluator& evaluator, const Ciphertext& like, size_t batch_split, size_t input_split, size_t output_split,
                                     const std::function<const Ciphertext*(size_t, size_t, size_t)>& ct_at,
                                     const std::function<const Plaintext*(size_t, size_t, size_t)>& pt_at, MemoryPoolHandle pool) {
    const size_t pcnt = like.polynomial_count(), L = like.coeff_modulus_size(), n = like.poly_modulus_degree(), words = pcnt * L * n;
    auto shared = std::make_shared<utils::DynamicArray>(batch_split * output_split * words, true, pool);
    shared->set_zero();
    Cipher2d ret;
    ret.data().resize(batch_split);
    for (size_t b = 0; b < batch_split; b++)
        for (size_t j = 0; j < output_split; j++)
            ret[b].push_back(Ciphertext::from_members(pcnt, L, n, like.parms_id(), like.scale(), true, like.correction_factor(), 0,
                                                      utils::DynamicArray::device_view(shared->raw_pointer() + (b * output_split + j) * words, words, shared)));
    std::vector<const Ciphertext*> c_ptrs;
    std::vector<const Plaintext*> p_ptrs;
    std::vector<Ciphertext*> r_ptrs;
    for (size_t i = 0; i < input_split; i++)
        for (size_t j = 0; j < output_split; j++)
            for (size_t b = 0; b < batch_split; b++) { c_ptrs.push_back(ct_at(b, i, j)); p_ptrs.push_back(pt_at(b, i, j)); r_ptrs.push_back(&ret[b][j]); }
    evaluator.multiply_plain_accumulate(c_ptrs, p_ptrs, r_ptrs, false, pool);
    if (evaluator.context()->first_context_data().value()->parms().scheme() == SchemeType::BFV) {
        troyn_check_public(troyn_ntt(evaluator.context()->plan(), 1, shared->raw_pointer(), shared->raw_pointer(), batch_split * output_split, pcnt,
                                     static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, troyn_current_stream()));
        for (auto& r : ret.data()) for (Ciphertext& c : r) c.is_ntt_form() = false;
    }
    return ret;
}

Cipher2d detail::accumulate_products_fly(const Evaluator& evaluator, const Cipher2d& inputs, size_t batch_split, size_t input_split, size_t output_split,
                                         const std::function<std::vector<Plaintext>(size_t)>& row, MemoryPoolHandle pool) {
    if (inputs.size() != batch_split) throw std::invalid_argument("[MatmulHelper::matmul] Input batch_size incorrect.");
    const Ciphertext& like = inputs[0][0];
    const size_t pcnt = like.polynomial_count(), L = like.coeff_modulus_size(), n = like.poly_modulus_degree(), words = pcnt * L * n;
    auto shared = std::make_shared<utils::DynamicArray>(batch_split * output_split * words, true, pool);
    shared->set_zero();
    Cipher2d ret;
    ret.data().resize(batch_split);
    for (size_t b = 0; b < batch_split; b++)
        for (size_t j = 0; j < output_split; j++)
            ret[b].push_back(Ciphertext::from_members(pcnt, L, n, like.parms_id(), like.scale(), true, like.correction_factor(), 0,
                                                      utils::DynamicArray::device_view(shared->raw_pointer() + (b * output_split + j) * words, words, shared)));
    for (size_t i = 0; i < input_split; i++) {
        const std::vector<Plaintext> w_i = row(i);                      // alive for this input block only
        if (w_i.size() != output_split) throw std::invalid_argument("[MatmulHelper::matmul_fly] Weight block count incorrect.");
        std::vector<const Ciphertext*> c_ptrs;
        std::vector<const Plaintext*> p_ptrs;
        std::vector<Ciphertext*> r_ptrs;
        for (size_t j = 0; j < output_split; j++)
            for (size_t b = 0; b < batch_split; b++) {
                if (inputs[b].size() != input_split) throw std::invalid_argument("[MatmulHelper::matmul] Input input_dims incorrect.");
                c_ptrs.push_back(&inputs[b][i]); p_ptrs.push_back(&w_i[j]); r_ptrs.push_back(&ret[b][j]);
            }
        evaluator.multiply_plain_accumulate(c_ptrs, p_ptrs, r_ptrs, false, pool);
    }
    if (evaluator.context()->first_context_data().value()->parms().scheme() == SchemeType::BFV) {
        troyn_check_public(troyn_ntt(evaluator.context()->plan(), 1, shared->raw_pointer(), shared->raw_pointer(), batch_split * output_split, pcnt,
                                     static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, troyn_current_stream()));
        for (auto& r : ret.data()) for (Ciphertext& c : r) c.is_ntt_form() = false;
    }
    return ret;
}

Cipher2d MatmulHelper::matmul_fly_uint64s(const BatchEncoder& encoder, const Evaluator& evaluator, const Cipher2d& inputs, const uint64_t* weights) const {
    const size_t batch_split = ceil_div(batch_size, batch_block), input_split = ceil_div(input_dims, input_block), output_split = ceil_div(output_dims, output_block);
    const uint64_t t = encoder.context()->first_context_data().value()->parms().plain_modulus().value();
    const size_t len = input_block * output_block;
    return detail::accumulate_products_fly(evaluator, inputs, batch_split, input_split, output_split, [&](size_t i) {
        std::vector<uint64_t> packed(output_split * len);
        for (size_t j = 0; j < output_split; j++) {
            const std::vector<uint64_t> block = weight_block(weights, i * input_block, j * output_block);
            for (uint64_t v : block) if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus");
            std::copy(block.begin(), block.end(), packed.begin() + static_cast<std::ptrdiff_t>(j * len));
        }
        Plain2d encoded = detail::encode_blocks_for_plain(encoder, packed, 1, output_split, len, pool);     // one copy, one centralize, one NTT launch
        return std::move(encoded[0]);
    }, pool);
}

Cipher2d MatmulHelper::matmul_fly_doubles(const CKKSEncoder& encoder, const Evaluator& evaluator, const Cipher2d& inputs, const double* weights, std::optional<ParmsID> parms_id,
                                          double scale) const {
    const size_t batch_split = ceil_div(batch_size, batch_block), input_split = ceil_div(input_dims, input_block), output_split = ceil_div(output_dims, output_block);
    return detail::accumulate_products_fly(evaluator, inputs, batch_split, input_split, output_split, [&](size_t i) {
        std::vector<Plaintext> row;
        for (size_t j = 0; j < output_split; j++) row.push_back(encoder.encode_float64_polynomial_new(weight_block(weights, i * input_block, j * output_block), parms_id, scale, pool));
        return row;
    }, pool);
}

Cipher2d MatmulHelper::matmul(const Evaluator& evaluator, const Cipher2d& a, const Plain2d& w) const {
    // app/matmul.cu:326-374, batched form
    const size_t batch_split = ceil_div(batch_size, batch_block), input_split = ceil_div(input_dims, input_block), output_split = ceil_div(output_dims, output_block);
    if (a.size() != batch_split) throw std::invalid_argument("[MatmulHelper::matmul] Input batch_size incorrect.");
    if (w.size() != input_split) throw std::invalid_argument("[MatmulHelper::matmul] Weight input dimension incorrect.");
    return detail::accumulate_products(evaluator, a[0][0], batch_split, input_split, output_split,
                                       [&](size_t b, size_t i, size_t) { return &a[b][i]; }, [&](size_t, size_t i, size_t j) { return &w[i][j]; }, pool);
}

Cipher2d MatmulHelper::matmul_reverse(const Evaluator& evaluator, const Plain2d& a, const Cipher2d& w) const {
    // app/matmul.cu:406-454: the weights are the encrypted side
    const size_t batch_split = ceil_div(batch_size, batch_block), input_split = ceil_div(input_dims, input_block), output_split = ceil_div(output_dims, output_block);
    if (a.size() != batch_split) throw std::invalid_argument("[MatmulHelper::matmul] Input batch_size incorrect.");
    if (w.size() != input_split) throw std::invalid_argument("[MatmulHelper::matmul] Weight input dimension incorrect.");
    return detail::accumulate_products(evaluator, w[0][0], batch_split, input_split, output_split,
                                       [&](size_t, size_t i, size_t j) { return &w[i][j]; }, [&](size_t b, size_t i, size_t) { return &a[b][i]; }, pool);
}

Cipher2d MatmulHelper::matmul_cipher(const Evaluator& evaluator, const Cipher2d& a, const Cipher2d& w) const {
    // app/matmul.cu:376-404
    const size_t batch_split = ceil_div(batch_size, batch_block), input_split = ceil_div(input_dims, input_block), output_split = ceil_div(output_dims, output_block);
    if (a.size() != batch_split) throw std::invalid_argument("[MatmulHelper::matmul_cipher] Input batch_size incorrect.");
    if (w.size() != input_split) throw std::invalid_argument("[MatmulHelper::matmul_cipher] Weight input dimension incorrect.");
    Cipher2d ret;
    for (size_t b = 0; b < batch_split; b++) {
        std::vector<Ciphertext>& row = ret.new_row();
        row.resize(output_split);
        for (size_t j = 0; j < output_split; j++) {
            std::vector<const Ciphertext*> lhs, rhs;
            for (size_t i = 0; i < input_split; i++) { lhs.push_back(&a[b][i]); rhs.push_back(&w[i][j]); }
            std::vector<Ciphertext> prods(input_split);
            std::vector<Ciphertext*> pp;
            for (Ciphertext& c : prods) pp.push_back(&c);
            evaluator.multiply_batched(lhs, rhs, pp, pool);
            row[j] = std::move(prods[0]);
            for (size_t i = 1; i < input_split; i++) evaluator.add_inplace(row[j], prods[i], pool);
        }
    }
    return ret;
}

std::vector<uint64_t> MatmulHelper::decrypt_outputs_uint64s(const BatchEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    // app/matmul.cu:519-566; every output ciphertext is decrypted in one batch and read back with one copy
    (void)encoder;
    std::vector<const Ciphertext*> all;
    for (const auto& r : outputs.data()) for (const Ciphertext& c : r) all.push_back(&c);
    const std::vector<uint64_t> coeffs = decryptor.bfv_decrypt_to_host(all, pool);
    const size_t n = slot_count, ocols = ceil_div(output_dims, output_block);
    if (all.size() != (pack_lwe ? ceil_div(ceil_div(batch_size, batch_block) * ocols, input_block) : ceil_div(batch_size, batch_block) * ocols))
        throw std::invalid_argument("[MatmulHelper::decrypt_outputs] Output ciphertext count incorrect");
    std::vector<uint64_t> out(batch_size * output_dims, 0);
    // the reference's mapping (app/matmul.cu:540-562) walked row by row of the result: `out` is written front to back (the block-by-block order
    // writes with a stride of output_dims words -- one cache line and, at 512 columns, one page per word: 1 ms of the 1.5 ms this call took)
    size_t di = 0;
    for (size_t li = 0; li < batch_size; li += batch_block, di++) {
        const size_t ui = std::min(li + batch_block, batch_size);
        for (size_t i = li; i < ui; i++) {
            size_t dj = 0;
            for (size_t lj = 0; lj < output_dims; lj += output_block, dj++) {
                const size_t uj = std::min(lj + output_block, output_dims);
                const size_t cipher_id = di * ocols + dj;
                // unpacked: output k of the block sits on coefficient ..*ib + ib-1; packed: ciphertext cipher_id % ib of its group
                // was moved to offset cipher_id % ib
                const uint64_t* cf = coeffs.data() + (pack_lwe ? cipher_id / input_block : cipher_id) * n;
                const size_t offset = pack_lwe ? cipher_id % input_block : input_block - 1;
                for (size_t j = lj; j < uj; j++) out[i * output_dims + j] = cf[(i - li) * input_block * output_block + (j - lj) * input_block + offset];
            }
        }
    }
    return out;
}

Plain2d MatmulHelper::encode_outputs_uint64s(const BatchEncoder& encoder, const uint64_t* outputs) const {
    // app/matmul.cu:452-512: the same coefficient positions decrypt_outputs reads
    const size_t n = slot_count, ocols = ceil_div(output_dims, output_block), brows = ceil_div(batch_size, batch_block);
    const size_t count = pack_lwe ? ceil_div(brows * ocols, input_block) : brows * ocols;
    // all plaintexts in ONE host image, ONE copy; the plaintext objects are windows of the shared device buffer (as encode_blocks_for_plain hands out the
    // weights) -- BatchEncoder::encode_polynomial_new per plaintext was an allocation, a pageable copy and a stream wait each (32 of them at 512^3)
    HeContextPointer context = encoder.context();
    if (!context->on_device()) throw std::invalid_argument("[MatmulHelper::encode_outputs] HeContext is not on device (call to_device_inplace).");
    const auto& parms = context->first_context_data().value()->parms();
    const uint64_t t = parms.plain_modulus().value();
    std::vector<uint64_t> image(count * n, 0);
    size_t di = 0;
    for (size_t li = 0; li < batch_size; li += batch_block, di++) {
        const size_t ui = std::min(li + batch_block, batch_size);
        size_t dj = 0;
        for (size_t lj = 0; lj < output_dims; lj += output_block, dj++) {
            const size_t uj = std::min(lj + output_block, output_dims);
            const size_t cipher_id = di * ocols + dj;
            uint64_t* buf = image.data() + (pack_lwe ? cipher_id / input_block : cipher_id) * n;
            const size_t offset = pack_lwe ? cipher_id % input_block : input_block - 1;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) {
                    const uint64_t v = outputs[i * output_dims + j];
                    if (v >= t) throw std::invalid_argument("[BatchEncoder::encode_polynomial] Value is larger than plain modulus");
                    buf[(i - li) * input_block * output_block + (j - lj) * input_block + offset] = v;
                }
        }
    }
    auto shared = std::make_shared<utils::DynamicArray>(count * n, true, pool);
    shared->copy_from(image.data(), image.size(), false);
    auto window = [&](size_t idx) {
        Plaintext p;                                         // = BatchEncoder::encode_polynomial_new of n values (batch_encoder.cu encode_polynomial)
        p.data() = utils::DynamicArray::device_view(shared->raw_pointer() + idx * n, n, shared);
        p.parms_id() = parms_id_zero;
        p.coeff_count() = n;
        p.is_ntt_form() = false;
        p.poly_modulus_degree() = n;
        p.coeff_modulus_size() = parms.coeff_modulus().size();
        return p;
    };
    Plain2d out;
    if (pack_lwe) {
        std::vector<Plaintext>& row = out.new_row();
        for (size_t k = 0; k < count; k++) row.push_back(window(k));
    } else {
        for (size_t r = 0; r < brows; r++) {
            std::vector<Plaintext>& row = out.new_row();
            for (size_t c = 0; c < ocols; c++) row.push_back(window(r * ocols + c));
        }
    }
    return out;
}

Cipher2d MatmulHelper::pack_outputs(const Evaluator& evaluator, const GaloisKeys& auto_key, const Cipher2d& cipher) const {
    // app/matmul.cu:572-619: consecutive runs of input_block outputs are packed into one ciphertext.  Output k of a group has
    // its results on coefficients = ib-1 (mod ib); the shift 2N - (ib-1) moves them to multiples of ib and the packing tree
    // interleaves the members, member k landing on offset k.  All groups run through the tree together.
    if (!pack_lwe) throw std::invalid_argument("[MatmulHelper::packOutputs] PackLwe not enabled");
    Cipher2d ret;
    ret.new_row();
    if (cipher.data().empty() || cipher.data()[0].empty()) return ret;
    const size_t pack_slots = input_block;
    const size_t inherent_shift = pack_slots == 1 ? 0 : 2 * slot_count - (pack_slots - 1);
    std::vector<std::vector<const Ciphertext*>> to_pack;
    for (const auto& row : cipher.data())
        for (const Ciphertext& c : row) {
            if (to_pack.empty() || to_pack.back().size() == pack_slots) to_pack.emplace_back();
            to_pack.back().push_back(&c);
        }
    ret[0] = evaluator.pack_rlwe_ciphertexts_new_batched(to_pack, auto_key, inherent_shift, input_block, 1, pool);
    return ret;
}

void MatmulHelper::serialize_encoded_weights(const Plain2d& w, std::ostream& stream, CompressionMode mode) const {
    // app/matmul.cu:621-636
    const size_t rows = w.rows();
    if (rows == 0) throw std::invalid_argument("[MatmulHelper::serialize_encoded_weights] No rows in weight matrix.");
    const size_t cols = w[0].size();
    if (cols == 0) throw std::invalid_argument("[MatmulHelper::serialize_encoded_weights] No columns in weight matrix.");
    for (size_t i = 0; i < rows; i++)
        if (w[i].size() != cols) throw std::invalid_argument("[MatmulHelper::serialize_encoded_weights] Weight matrix is not rectangular.");
    put_size(stream, rows);
    put_size(stream, cols);
    for (size_t i = 0; i < rows; i++) for (size_t j = 0; j < cols; j++) w[i][j].save(stream, mode);
}

Plain2d MatmulHelper::deserialize_encoded_weights(std::istream& stream) const {
    const size_t rows = get_size(stream), cols = get_size(stream);
    Plain2d ret;
    for (size_t i = 0; i < rows; i++) {
        std::vector<Plaintext>& row = ret.new_row();
        for (size_t j = 0; j < cols; j++) row.push_back(Plaintext::load_new(stream, pool));
    }
    return ret;
}

static std::vector<size_t> output_terms(const MatmulHelper& h, size_t li, size_t ui, size_t lj, size_t uj) {
    std::vector<size_t> required;
    for (size_t i = li; i < ui; i++)
        for (size_t j = lj; j < uj; j++) required.push_back((i - li) * h.input_block * h.output_block + (j - lj) * h.input_block + h.input_block - 1);
    return required;
}

void MatmulHelper::serialize_outputs(const Evaluator& evaluator, const Cipher2d& x, std::ostream& stream, CompressionMode mode) const {
    // app/matmul.cu:655-688: unpacked outputs carry results on (ui-li)*(uj-lj) coefficients only, so c0 is cut down to those
    HeContextPointer context = evaluator.context();
    if (!pack_lwe) {
        size_t di = 0;
        for (size_t li = 0; li < batch_size; li += batch_block, di++) {
            const size_t ui = std::min(li + batch_block, batch_size);
            size_t dj = 0;
            for (size_t lj = 0; lj < output_dims; lj += output_block, dj++)
                x[di][dj].save_terms(stream, context, output_terms(*this, li, ui, lj, std::min(lj + output_block, output_dims)), pool, mode);
        }
    } else {
        const size_t count = ceil_div(ceil_div(batch_size, batch_block) * ceil_div(output_dims, output_block), input_block);
        if (x.data().empty() || count != x.data()[0].size()) throw std::invalid_argument("[MatmulHelper::serialize_outputs] Output ciphertext count incorrect");
        std::vector<const Ciphertext*> ptrs;
        for (const Ciphertext& c : x.data()[0]) ptrs.push_back(&c);
        Ciphertext::save_many(stream, ptrs.data(), ptrs.size(), context, mode);
    }
}

Cipher2d MatmulHelper::deserialize_outputs(const Evaluator& evaluator, std::istream& stream) const {
    // app/matmul.cu:690-720
    HeContextPointer context = evaluator.context();
    Cipher2d ret;
    if (!pack_lwe) {
        for (size_t li = 0; li < batch_size; li += batch_block) {
            const size_t ui = std::min(li + batch_block, batch_size);
            std::vector<Ciphertext>& row = ret.new_row();
            for (size_t lj = 0; lj < output_dims; lj += output_block)
                row.push_back(Ciphertext::load_terms_new(stream, context, output_terms(*this, li, ui, lj, std::min(lj + output_block, output_dims)), pool));
        }
    } else {
        const size_t count = ceil_div(ceil_div(batch_size, batch_block) * ceil_div(output_dims, output_block), input_block);
        std::vector<Ciphertext>& row = ret.new_row();
        row.resize(count);
        std::vector<Ciphertext*> ptrs;
        for (Ciphertext& c : row) ptrs.push_back(&c);
        Ciphertext::load_many(stream, ptrs.data(), ptrs.size(), context, pool);
    }
    return ret;
}

// ------------------------------------------------------------------------------------------------
// CKKS forms (encoder_adapter.h:28-46: both operand kinds are encode_float64_polynomial at the given level and scale)
// ------------------------------------------------------------------------------------------------
Plain2d MatmulHelper::encode_weights_doubles(const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const {
    Plain2d out;
    for (size_t li = 0; li < input_dims; li += input_block) {
        const size_t ui = std::min(li + input_block, input_dims);
        std::vector<Plaintext>& row = out.new_row();
        for (size_t lj = 0; lj < output_dims; lj += output_block) {
            const size_t uj = std::min(lj + output_block, output_dims);
            std::vector<double> vec(input_block * output_block, 0.0);
            for (size_t j = lj; j < uj; j++)
                for (size_t i = li; i < ui; i++) vec[(j - lj) * input_block + input_block - (i - li) - 1] = weights[i * output_dims + j];
            row.push_back(encoder.encode_float64_polynomial_new(vec, parms_id, scale, pool));
        }
    }
    return out;
}

Plain2d MatmulHelper::encode_inputs_doubles(const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const {
    Plain2d out;
    for (size_t li = 0; li < batch_size; li += batch_block) {
        const size_t ui = std::min(li + batch_block, batch_size);
        std::vector<Plaintext>& row = out.new_row();
        for (size_t lj = 0; lj < input_dims; lj += input_block) {
            const size_t uj = std::min(lj + input_block, input_dims);
            std::vector<double> vec(slot_count, 0.0);
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) vec[(i - li) * input_block * output_block + (j - lj)] = inputs[i * input_dims + j];
            row.push_back(encoder.encode_float64_polynomial_new(vec, parms_id, scale, pool));
        }
    }
    return out;
}

Cipher2d MatmulHelper::encrypt_inputs_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* inputs, std::optional<ParmsID> parms_id, double scale) const {
    // app/matmul.cu:296-311: symmetric encryption with the c1 seed kept (half the wire size; Cipher2d::load / expand_seed restores c1)
    const Plain2d plain = encode_inputs_doubles(encoder, inputs, parms_id, scale);
    Cipher2d out;
    for (const auto& prow : plain.data()) {
        std::vector<Ciphertext>& row = out.new_row();
        for (const Plaintext& p : prow) row.push_back(encryptor.encrypt_symmetric_new(p, true, pool));
    }
    return out;
}

Cipher2d MatmulHelper::encrypt_weights_doubles(const Encryptor& encryptor, const CKKSEncoder& encoder, const double* weights, std::optional<ParmsID> parms_id, double scale) const {
    const Plain2d plain = encode_weights_doubles(encoder, weights, parms_id, scale);
    Cipher2d out;
    for (const auto& prow : plain.data()) {
        std::vector<Ciphertext>& row = out.new_row();
        for (const Plaintext& p : prow) row.push_back(encryptor.encrypt_symmetric_new(p, true, pool));
    }
    return out;
}

Plain2d MatmulHelper::encode_outputs_doubles(const CKKSEncoder& encoder, const double* outputs, std::optional<ParmsID> parms_id, double scale) const {
    const size_t n = slot_count, ocols = ceil_div(output_dims, output_block), brows = ceil_div(batch_size, batch_block);
    const size_t count = pack_lwe ? ceil_div(brows * ocols, input_block) : brows * ocols;
    std::vector<std::vector<double>> buffers(count, std::vector<double>(n, 0.0));
    size_t di = 0;
    for (size_t li = 0; li < batch_size; li += batch_block, di++) {
        const size_t ui = std::min(li + batch_block, batch_size);
        size_t dj = 0;
        for (size_t lj = 0; lj < output_dims; lj += output_block, dj++) {
            const size_t uj = std::min(lj + output_block, output_dims);
            const size_t cipher_id = di * ocols + dj;
            std::vector<double>& buf = buffers[pack_lwe ? cipher_id / input_block : cipher_id];
            const size_t offset = pack_lwe ? cipher_id % input_block : input_block - 1;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) buf[(i - li) * input_block * output_block + (j - lj) * input_block + offset] = outputs[i * output_dims + j];
        }
    }
    Plain2d out;
    if (pack_lwe) {
        std::vector<Plaintext>& row = out.new_row();
        for (const auto& buf : buffers) row.push_back(encoder.encode_float64_polynomial_new(buf, parms_id, scale, pool));
    } else {
        for (size_t r = 0; r < brows; r++) {
            std::vector<Plaintext>& row = out.new_row();
            for (size_t c = 0; c < ocols; c++) row.push_back(encoder.encode_float64_polynomial_new(buffers[r * ocols + c], parms_id, scale, pool));
        }
    }
    return out;
}

std::vector<double> MatmulHelper::decrypt_outputs_doubles(const CKKSEncoder& encoder, const Decryptor& decryptor, const Cipher2d& outputs) const {
    std::vector<std::vector<double>> coeffs;
    for (const auto& r : outputs.data()) for (const Ciphertext& c : r) coeffs.push_back(encoder.decode_float64_polynomial_new(decryptor.decrypt_new(c, pool), pool));
    const size_t ocols = ceil_div(output_dims, output_block);
    if (coeffs.size() != (pack_lwe ? ceil_div(ceil_div(batch_size, batch_block) * ocols, input_block) : ceil_div(batch_size, batch_block) * ocols))
        throw std::invalid_argument("[MatmulHelper::decrypt_outputs] Output ciphertext count incorrect");
    std::vector<double> out(batch_size * output_dims, 0.0);
    size_t di = 0;
    for (size_t li = 0; li < batch_size; li += batch_block, di++) {
        const size_t ui = std::min(li + batch_block, batch_size);
        size_t dj = 0;
        for (size_t lj = 0; lj < output_dims; lj += output_block, dj++) {
            const size_t uj = std::min(lj + output_block, output_dims);
            const size_t cipher_id = di * ocols + dj;
            const std::vector<double>& cf = coeffs[pack_lwe ? cipher_id / input_block : cipher_id];
            const size_t offset = pack_lwe ? cipher_id % input_block : input_block - 1;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) out[i * output_dims + j] = cf[(i - li) * input_block * output_block + (j - lj) * input_block + offset];
        }
    }
    return out;
}

}}  // namespace troy::linear
