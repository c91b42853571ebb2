// Plaintext-side operations of the host-side mirror: partial RNS plaintexts, BatchEncoder::scale_up / scale_down / centralize /
// decentralize, Evaluator::apply_galois_plain, the integer CKKS encodings, Ciphertext::is_transparent.
// Reference: src/batch_encoder.cu:558-662, src/utils/scaling_variant.cu:326-431, src/evaluator_keyswitching.cu:235-261,
// src/ckks_encoder.cu:983-1090, src/plaintext.cu (resize_rns_partial), src/ciphertext.cu:73-77.
// Device work goes through the C-ABI (include/troyn.h); nothing here computes ring arithmetic on the host.
#include <hip/hip_runtime.h>

#include <sstream>

#include "troy.h"

// Every method here is asynchronous on the calling thread's stream, like the reference's: temporaries return to the pool, which hands a
// block back to the thread that released it in stream order (rounds 1-3 waited for the stream at the end of each method).

namespace troy {

namespace {

hipStream_t stream() { return static_cast<hipStream_t>(troyn_current_stream()); }

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("[kernel_provider::") + what + "] " + hipGetErrorString(e));
}

// [L][N] (zero-padded rows) -> [L][count]
void compact_rows(const uint64_t* full, uint64_t* partial, size_t L, size_t n, size_t count) {
    hip_ok(hipMemcpy2DAsync(partial, count * sizeof(uint64_t), full, n * sizeof(uint64_t), count * sizeof(uint64_t), L, hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
}

ContextDataPointer level_of(const char* prompt, const HeContextPointer& context, const ParmsID& id) {
    auto cd = context->get_context_data(id);
    if (!cd.has_value()) throw std::invalid_argument(std::string(prompt) + " Could not find context data.");
    return cd.value();
}

void need_device(const char* prompt, const HeContextPointer& context, const Plaintext& plain) {
    if (!context->on_device() || !plain.on_device()) throw std::invalid_argument(std::string(prompt) + " Operands must be on the device (the encoder's scaling runs on the GPU only).");
}

Plaintext mod_t_like(const Plaintext& rns, size_t L, size_t n, MemoryPoolHandle pool) {
    Plaintext d;
    d.data() = utils::DynamicArray(0, true, pool);
    d.parms_id() = parms_id_zero;
    d.resize(rns.coeff_count());
    d.coeff_modulus_size() = L;
    d.poly_modulus_degree() = n;
    d.is_ntt_form() = false;
    return d;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// partial RNS plaintexts
// ------------------------------------------------------------------------------------------------
void Plaintext::resize_rns_partial(const HeContext& context, const ParmsID& parms_id, size_t coeff_count, bool fill_extra_with_zeros, bool copy_data) {
    auto cd = context.get_context_data(parms_id);
    if (!cd.has_value()) throw std::invalid_argument("[Plaintext::resize_rns] parms_id is not valid");
    const EncryptionParameters& p = cd.value()->parms();
    // (no upper bound: the reference lets a partial plaintext hold MORE than N coefficients per limb, test/app/bfv_ring2k.cu:93-110 decodes such a one)
    parms_id_ = parms_id;
    coeff_modulus_size_ = p.coeff_modulus().size();
    poly_modulus_degree_ = p.poly_modulus_degree();
    coeff_count_ = coeff_count;
    const size_t words = coeff_modulus_size_ * coeff_count;
    if (fill_extra_with_zeros) data_.resize(words, copy_data); else data_.resize_uninitialized(words, copy_data);
}

utils::DynamicArray Plaintext::expanded_rns(size_t L, size_t n, MemoryPoolHandle pool) const {
    if (!on_device()) throw std::invalid_argument("[Plaintext::expanded_rns] Plaintext is on host.");
    if (coeff_count_ > n || data_.size() < L * coeff_count_) throw std::invalid_argument("[Plaintext::expanded_rns] Plaintext does not have the shape of this level.");
    utils::DynamicArray full(L * n, true, pool);
    if (coeff_count_ == n) {
        hip_ok(hipMemcpyAsync(full.raw_pointer(), data_.raw_pointer(), L * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    } else {
        hip_ok(hipMemsetAsync(full.raw_pointer(), 0, L * n * sizeof(uint64_t), stream()), "memset");
        if (coeff_count_ > 0)
            hip_ok(hipMemcpy2DAsync(full.raw_pointer(), n * sizeof(uint64_t), data_.raw_pointer(), coeff_count_ * sizeof(uint64_t), coeff_count_ * sizeof(uint64_t), L,
                                    hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    }
    return full;
}

std::string Plaintext::to_string() const {
    if (is_ntt_form_ || parms_id_ != parms_id_zero) throw std::invalid_argument("cannot convert NTT or RNS plaintext to string");
    const std::vector<uint64_t> c = data_.to_vector();
    std::ostringstream out;
    bool empty = true;
    for (size_t k = std::min(coeff_count_, c.size()); k-- > 0;) {
        if (c[k] == 0) continue;
        if (!empty) out << " + ";
        out << std::hex << std::uppercase << c[k] << std::dec;
        if (k) out << "x^" << k;
        empty = false;
    }
    return empty ? std::string("0") : out.str();
}

bool Ciphertext::is_transparent() const {
    if (data_.size() == 0 || polynomial_count_ < 2) return true;
    for (uint64_t w : data_.to_vector())
        if (w != 0) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// BatchEncoder: scale_up / centralize and their inverses
// ------------------------------------------------------------------------------------------------
Plaintext BatchEncoder::scale_up_new(const Plaintext& plain, std::optional<ParmsID> parms_id, MemoryPoolHandle pool) const {
    const char* P = "[BatchEncoder::scale_up_new]";
    if (context_->first_context_data().value()->parms().scheme() != SchemeType::BFV) throw std::logic_error(std::string(P) + " Only BFV scheme is supported.");
    if (plain.parms_id() != parms_id_zero) throw std::invalid_argument(std::string(P) + " Plaintext is already at the desired level.");
    need_device(P, context_, plain);
    const ParmsID pid = parms_id.value_or(context_->first_parms_id());
    ContextDataPointer cd = level_of(P, context_, pid);
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree(), cc = plain.coeff_count();
    if (cc > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
    // round(q/t * m) added onto zeros: the kernel of Encryptor::encrypt / add_plain with a zero "ciphertext polynomial"
    utils::DynamicArray full(L * n, true, pool);
    hip_ok(hipMemsetAsync(full.raw_pointer(), 0, L * n * sizeof(uint64_t), stream()), "memset");
    troyn_check_public(troyn_bfv_scale_up(context_->behz(L), plain.poly(), cc, n, full.raw_pointer(), L * n, full.raw_pointer(), L * n, 0, 1, troyn_current_stream()));
    Plaintext d;
    d.data() = utils::DynamicArray(0, true, pool);
    d.resize_rns_partial(*context_, pid, cc);
    compact_rows(full.raw_pointer(), d.poly(), L, n, cc);
    d.is_ntt_form() = false;
    return d;
}

Plaintext BatchEncoder::centralize_new(const Plaintext& plain, std::optional<ParmsID> parms_id, MemoryPoolHandle pool) const {
    const char* P = "[BatchEncoder::centralize_new]";
    const SchemeType scheme = context_->first_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::logic_error(std::string(P) + " Only BFV/BGV scheme is supported.");
    if (plain.parms_id() != parms_id_zero) throw std::invalid_argument(std::string(P) + " Plaintext is already at the desired level.");
    need_device(P, context_, plain);
    const ParmsID pid = parms_id.value_or(context_->first_parms_id());
    ContextDataPointer cd = level_of(P, context_, pid);
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree(), cc = plain.coeff_count();
    if (cc > n) throw std::invalid_argument("[scaling_variant::centralize] plain_coeff_count exceeds the polynomial degree.");
    utils::DynamicArray full(L * n, true, pool);
    troyn_check_public(troyn_plain_centralize(context_->plan(), static_cast<uint32_t>(L), cd->parms().plain_modulus().value(), plain.poly(), cc, n, full.raw_pointer(), 1,
                                              troyn_current_stream()));
    Plaintext d;
    d.data() = utils::DynamicArray(0, true, pool);
    d.resize_rns_partial(*context_, pid, cc);
    compact_rows(full.raw_pointer(), d.poly(), L, n, cc);
    d.is_ntt_form() = false;
    return d;
}

Plaintext BatchEncoder::scale_down_new(const Plaintext& plain, MemoryPoolHandle pool) const {
    const char* P = "[BatchEncoder::scale_down_new]";
    if (context_->first_context_data().value()->parms().scheme() != SchemeType::BFV) throw std::logic_error(std::string(P) + " Only BFV scheme is supported.");
    if (plain.parms_id() == parms_id_zero) throw std::invalid_argument(std::string(P) + " Plaintext not in RNS form.");
    if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is in NTT form.");
    need_device(P, context_, plain);
    ContextDataPointer cd = level_of(P, context_, plain.parms_id());
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree();
    // RNSTool::decrypt_scale_and_round is coefficient-wise: run it on the zero-padded polynomial, keep the first coeff_count results
    const utils::DynamicArray full = plain.expanded_rns(L, n, pool);
    utils::DynamicArray out(n, true, pool);
    troyn_check_public(troyn_bfv_decrypt_scale_and_round(context_->behz(L), full.raw_pointer(), out.raw_pointer(), 1, troyn_current_stream()));
    Plaintext d = mod_t_like(plain, L, n, pool);
    hip_ok(hipMemcpyAsync(d.poly(), out.raw_pointer(), plain.coeff_count() * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    return d;
}

Plaintext BatchEncoder::decentralize_new(const Plaintext& plain, uint64_t correction_factor, MemoryPoolHandle pool) const {
    const char* P = "[BatchEncoder::decentralize_new]";
    const SchemeType scheme = context_->first_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::logic_error(std::string(P) + " Only BFV/BGV scheme is supported.");
    if (plain.parms_id() == parms_id_zero) throw std::invalid_argument(std::string(P) + " Plaintext not in RNS form.");
    if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is in NTT form.");
    need_device(P, context_, plain);
    ContextDataPointer cd = level_of(P, context_, plain.parms_id());
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree();
    const utils::DynamicArray full = plain.expanded_rns(L, n, pool);
    utils::DynamicArray out(n, true, pool);
    // RNSTool::decrypt_mod_t, then times correction_factor^{-1} mod t (the C entry folds both, as Decryptor::bgv_decrypt does)
    troyn_check_public(troyn_bgv_decrypt_mod_t(context_->bgv(L), full.raw_pointer(), correction_factor, out.raw_pointer(), 1, troyn_current_stream()));
    Plaintext d = mod_t_like(plain, L, n, pool);
    hip_ok(hipMemcpyAsync(d.poly(), out.raw_pointer(), plain.coeff_count() * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    return d;
}

// ------------------------------------------------------------------------------------------------
// Evaluator::apply_galois_plain
// ------------------------------------------------------------------------------------------------
void Evaluator::apply_galois_plain(const Plaintext& plain, size_t galois_element, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::apply_galois_plain]";
    if (!context_->on_device() || !plain.on_device()) throw std::invalid_argument(std::string(P) + " Operand is on host; the evaluator runs on the GPU only.");
    ContextDataPointer key_cd = context_->key_context_data().value();
    const size_t n = key_cd->parms().poly_modulus_degree();
    if ((galois_element & 1) == 0 || galois_element > 2 * n) throw std::invalid_argument("[Evaluator::apply_galois_inplace] Galois element is not valid.");
    Plaintext out = plain;                                  // same shape and metadata (Plaintext::like)
    out.data() = utils::DynamicArray(plain.data().size(), true, pool);
    if (plain.parms_id() == parms_id_zero) {
        // mod t, coefficient form (BFV / BGV): all N coefficients take part, shorter plaintexts are zero-padded first
        if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " A plaintext modulo t cannot be in NTT form.");
        const uint64_t t = key_cd->parms().plain_modulus().value();
        if (plain.coeff_count() == n) {
            troyn_check_public(troyn_apply_galois_plain(context_->plan(), t, galois_element, plain.poly(), out.poly(), 1, troyn_current_stream()));
        } else {
            if (plain.coeff_count() > n) throw std::invalid_argument(std::string(P) + " Plaintext has too many coefficients.");
            utils::DynamicArray padded(n, true, pool);
            hip_ok(hipMemsetAsync(padded.raw_pointer(), 0, n * sizeof(uint64_t), stream()), "memset");
            hip_ok(hipMemcpyAsync(padded.raw_pointer(), plain.poly(), plain.coeff_count() * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
            out.resize(n);
            troyn_check_public(troyn_apply_galois_plain(context_->plan(), t, galois_element, padded.raw_pointer(), out.poly(), 1, troyn_current_stream()));
        }
    } else {
        // RNS plaintext of a level (CKKS, or BFV / BGV after scale_up / centralize): limb-wise, either form
        ContextDataPointer cd = level_of(P, context_, plain.parms_id());
        const size_t L = cd->parms().coeff_modulus().size();
        if (plain.coeff_count() != n) throw std::invalid_argument(std::string(P) + " The automorphism needs all N coefficients of an RNS plaintext.");
        troyn_check_public(troyn_apply_galois(context_->plan(), 0, static_cast<uint32_t>(L), plain.is_ntt_form() ? 1 : 0, galois_element, plain.poly(), out.poly(), 1,
                                              troyn_current_stream()));
    }
    destination = std::move(out);
}

// ------------------------------------------------------------------------------------------------
// CKKSEncoder: exact integers at scale 1
// ------------------------------------------------------------------------------------------------
void CKKSEncoder::encode_integer64_polynomial(const std::vector<int64_t>& values, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[CKKSEncoder::encode_internal_integer_polynomial_slice]";
    if (!context_->on_device()) throw std::invalid_argument(std::string(P) + " HeContext is not on device (call to_device_inplace).");
    const ParmsID pid = parms_id.value_or(context_->first_parms_id());
    auto cdo = context_->get_context_data(pid);
    if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " parms_id not valid for context.");
    const auto& q = cdo.value()->parms().coeff_modulus();
    const size_t n = cdo.value()->parms().poly_modulus_degree(), L = q.size();
    if (values.size() > n) throw std::invalid_argument(std::string(P) + " Too many input values.");
    // reduce_values: v >= 0 -> v mod q_i, v < 0 -> q_i - (|v| mod q_i)  (index bookkeeping, not ring arithmetic; the NTT runs on the device)
    std::vector<uint64_t> host(L * n, 0);
    for (size_t j = 0; j < values.size(); j++) {
        const bool neg = values[j] < 0;
        const uint64_t mag = neg ? static_cast<uint64_t>(0) - static_cast<uint64_t>(values[j]) : static_cast<uint64_t>(values[j]);
        for (size_t i = 0; i < L; i++) {
            const uint64_t r = mag % q[i].value();
            host[i * n + j] = (neg && r) ? q[i].value() - r : r;
        }
    }
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, pid);
    out.data().copy_from(host.data(), host.size(), false);
    troyn_check_public(troyn_ntt(context_->plan(), 0, out.poly(), out.poly(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0,
                                 troyn_current_stream()));
    out.scale() = 1.0;
    out.is_ntt_form() = true;
    destination = std::move(out);
}

}  // namespace troy

namespace troy {

// ------------------------------------------------------------------------------------------------
// Evaluator: mod-t plaintexts to full RNS plaintexts, plaintext NTT transforms (evaluator_transform_ntt.cu:131-240, :366-420)
// ------------------------------------------------------------------------------------------------
void Evaluator::bfv_centralize(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::bfv_centralize]";
    if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is in NTT form.");
    if (plain.parms_id() != parms_id_zero) throw std::invalid_argument(std::string(P) + " Plaintext is not modulo t.");
    need_device(P, context_, plain);
    ContextDataPointer cd = level_of("[Evaluator::transform_plain_to_ntt_inplace]", context_, parms_id);
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree();
    if (plain.coeff_count() > n) throw std::invalid_argument("[scaling_variant::centralize] plain_coeff_count exceeds the polynomial degree.");
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, parms_id);
    troyn_check_public(troyn_plain_centralize(context_->plan(), static_cast<uint32_t>(L), cd->parms().plain_modulus().value(), plain.poly(), plain.coeff_count(), n, out.poly(), 1,
                                              troyn_current_stream()));
    out.is_ntt_form() = false;
    out.scale() = plain.scale();
    destination = std::move(out);
}

void Evaluator::bfv_scale_up(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::bfv_centralize]";                       // the reference reuses this prompt (evaluator_transform_ntt.cu:186-189)
    if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is in NTT form.");
    if (plain.parms_id() != parms_id_zero) throw std::invalid_argument(std::string(P) + " Plaintext is not modulo t.");
    need_device(P, context_, plain);
    ContextDataPointer cd = level_of("[Evaluator::transform_plain_to_ntt_inplace]", context_, parms_id);
    if (cd->parms().scheme() != SchemeType::BFV) throw std::logic_error("[Evaluator::bfv_scale_up] Only BFV scheme is supported.");
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree();
    if (plain.coeff_count() > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, parms_id);
    hip_ok(hipMemsetAsync(out.poly(), 0, L * n * sizeof(uint64_t), stream()), "memset");
    troyn_check_public(troyn_bfv_scale_up(context_->behz(L), plain.poly(), plain.coeff_count(), n, out.poly(), L * n, out.poly(), L * n, 0, 1, troyn_current_stream()));
    out.is_ntt_form() = false;
    out.scale() = plain.scale();
    destination = std::move(out);
}

void Evaluator::bfv_centralize_batched(const std::vector<const Plaintext*>& plain, const ParmsID& parms_id, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool) const {
    if (plain.size() != destination.size()) throw std::invalid_argument("[Evaluator::transform_plain_to_ntt_batched] The number of plaintexts does not match the number of destinations.");
    for (size_t i = 0; i < plain.size(); i++) { Plaintext d; bfv_centralize(*plain[i], parms_id, d, pool); *destination[i] = std::move(d); }
}

void Evaluator::bfv_scale_up_batched(const std::vector<const Plaintext*>& plain, const ParmsID& parms_id, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool) const {
    if (plain.size() != destination.size()) throw std::invalid_argument("[Evaluator::transform_plain_to_ntt_batched] The number of plaintexts does not match the number of destinations.");
    for (size_t i = 0; i < plain.size(); i++) { Plaintext d; bfv_scale_up(*plain[i], parms_id, d, pool); *destination[i] = std::move(d); }
}

void Evaluator::transform_plain_to_ntt_batched(const std::vector<const Plaintext*>& plain, const ParmsID& parms_id, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool) const {
    if (plain.size() != destination.size()) throw std::invalid_argument("[Evaluator::transform_plain_to_ntt_batched] The number of plaintexts does not match the number of destinations.");
    for (size_t i = 0; i < plain.size(); i++) { Plaintext d; transform_plain_to_ntt(*plain[i], parms_id, d, pool); *destination[i] = std::move(d); }
}

void Evaluator::transform_plain_from_ntt(const Plaintext& plain, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::transform_plain_from_ntt_inplace]";
    if (!plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is already in NTT form.");      // (sic) the reference's wording
    if (plain.parms_id() == parms_id_zero) throw std::invalid_argument(std::string(P) + " Invalid ParmsID, but this should never be reached.");
    need_device(P, context_, plain);
    ContextDataPointer cd = level_of(P, context_, plain.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    Plaintext out = plain;
    out.data() = utils::DynamicArray(plain.data().size(), true, pool);
    troyn_check_public(troyn_ntt(context_->plan(), 1, plain.poly(), out.poly(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, troyn_current_stream()));
    out.is_ntt_form() = false;
    destination = std::move(out);
}

void Evaluator::transform_plain_from_ntt_batched(const std::vector<const Plaintext*>& plain, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool) const {
    if (plain.size() != destination.size()) throw std::invalid_argument("[Evaluator::transform_plain_from_ntt_batched] The number of plaintexts does not match the number of destinations.");
    for (size_t i = 0; i < plain.size(); i++) { Plaintext d; transform_plain_from_ntt(*plain[i], d, pool); *destination[i] = std::move(d); }
}

}  // namespace troy

extern "C" void troy_wrapper::create_memory_pool_handle(size_t device_index, troy::MemoryPoolHandle* out) {
    if (out) *out = troy::MemoryPool::create(device_index);
}
