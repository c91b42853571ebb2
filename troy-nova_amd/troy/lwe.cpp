// LWE extraction, RLWE packing, ciphertext +/- plaintext and the partial ("terms") wire format of the host-side mirror.
// Reference: src/evaluator_lwes.cu, src/lwe_ciphertext.cu, src/evaluator_translate_plain.cu, src/ciphertext.cu:213-339.
// Device work goes through the C-ABI (include/troyn.h); nothing here computes on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "troy.h"

namespace troy {

namespace {

hipStream_t stream() { return static_cast<hipStream_t>(troyn_current_stream()); }

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("[kernel_provider::") + what + "] " + hipGetErrorString(e));
}

void no_seed(const char* prompt, const Ciphertext& c) {
    if (c.contains_seed()) throw std::invalid_argument(std::string(prompt) + " Argument contains seed.");
}

void need_device(const char* prompt, const HeContextPointer& ctx, const Ciphertext& c) {
    if (!ctx->on_device() || !c.on_device()) throw std::invalid_argument(std::string(prompt) + " Operands must be on the device (the evaluator runs on the GPU only).");
}

ContextDataPointer level(const char* prompt, const HeContextPointer& ctx, const ParmsID& id) {
    auto cd = ctx->get_context_data(id);
    if (!cd.has_value()) throw std::invalid_argument(std::string(prompt) + " ParmsID is not valid for the current context.");
    return cd.value();
}

bool are_close_double(double a, double b) {   // utils/basics.h:150-160
    const double scale_factor = std::max(std::max(a, b), 1.0);
    return std::fabs(a - b) < scale_factor * 1e-8;
}

std::pair<size_t, bool> power_of_two(uint64_t r) {
    size_t p = 0;
    while ((static_cast<uint64_t>(1) << (p + 1)) <= r && p < 62) p++;
    return {p, r == (static_cast<uint64_t>(1) << p)};
}

size_t reverse_bits(size_t v, size_t bits) {
    size_t r = 0;
    for (size_t i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// LWECiphertext  (lwe_ciphertext.h, lwe_ciphertext.cu)
// ------------------------------------------------------------------------------------------------
LWECiphertext LWECiphertext::clone(MemoryPoolHandle pool) const {
    LWECiphertext r;
    r.coeff_modulus_size_ = coeff_modulus_size_;
    r.poly_modulus_degree_ = poly_modulus_degree_;
    r.c0_ = c0_.clone(pool);
    r.c1_ = c1_.clone(pool);
    r.parms_id_ = parms_id_;
    r.scale_ = scale_;
    r.correction_factor_ = correction_factor_;
    return r;
}

bool LWECiphertext::on_device() const {
    if (c0_.on_device() != c1_.on_device()) throw std::runtime_error("[LWECiphertext::on_device] c0 and c1 are not on the same device");
    return c0_.on_device();
}

static void assemble_into(const LWECiphertext& lwe, uint64_t* rlwe) {
    // lwe_ciphertext.cu:11-21: c0[l] becomes the constant coefficient of limb l, c1 is kept
    const size_t L = lwe.coeff_modulus_size(), n = lwe.poly_modulus_degree(), pc = L * n;
    hipStream_t s = stream();
    hip_ok(hipMemsetAsync(rlwe, 0, pc * 8, s), "memset");
    hip_ok(hipMemcpy2DAsync(rlwe, n * 8, lwe.c0(), 8, 8, L, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
    hip_ok(hipMemcpyAsync(rlwe + pc, lwe.c1(), pc * 8, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
}

Ciphertext LWECiphertext::assemble_lwe(MemoryPoolHandle pool) const {
    if (!on_device()) throw std::invalid_argument("[LWECiphertext::assemble_lwe] the LWE ciphertext must be on the device.");
    const size_t pc = coeff_modulus_size_ * poly_modulus_degree_;
    utils::DynamicArray data(2 * pc, true, pool);
    assemble_into(*this, data.raw_pointer());
    troyn_sync_current_stream();
    return Ciphertext::from_members(2, coeff_modulus_size_, poly_modulus_degree_, parms_id_, scale_, false, correction_factor_, 0, std::move(data));
}

std::vector<Ciphertext> LWECiphertext::assemble_lwe_batched_new(const std::vector<const LWECiphertext*>& lwes, MemoryPoolHandle pool) {
    // the assembled ciphertexts are windows of one buffer
    std::vector<Ciphertext> out;
    if (lwes.empty()) return out;
    const size_t L = lwes[0]->coeff_modulus_size(), n = lwes[0]->poly_modulus_degree(), words = 2 * L * n;
    for (const LWECiphertext* l : lwes) {
        if (!l->on_device()) throw std::invalid_argument("[LWECiphertext::assemble_lwe] the LWE ciphertext must be on the device.");
        if (l->coeff_modulus_size() != L || l->poly_modulus_degree() != n) throw std::invalid_argument("[assemble_lwe_set_batched]: invalid input sizes");
    }
    auto shared = std::make_shared<utils::DynamicArray>(lwes.size() * words, true, pool);
    for (size_t i = 0; i < lwes.size(); i++) assemble_into(*lwes[i], shared->raw_pointer() + i * words);
    troyn_sync_current_stream();
    for (size_t i = 0; i < lwes.size(); i++)
        out.push_back(Ciphertext::from_members(2, L, n, lwes[i]->parms_id(), lwes[i]->scale(), false, lwes[i]->correction_factor(), 0,
                                               utils::DynamicArray::device_view(shared->raw_pointer() + i * words, words, shared)));
    return out;
}

// ------------------------------------------------------------------------------------------------
// Evaluator: ciphertext +/- plaintext  (evaluator_translate_plain.cu:13-93)
// ------------------------------------------------------------------------------------------------
void Evaluator::translate_plain_inplace(Ciphertext& encrypted, const Plaintext& plain, bool subtract, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::translate_plain_inplace]";
    (void)pool;
    no_seed(P, encrypted);
    need_device(P, context_, encrypted);
    if (!plain.on_device()) throw std::invalid_argument(std::string(P) + " Operands must be on the device (the evaluator runs on the GPU only).");
    ContextDataPointer cd = level(P, context_, encrypted.parms_id());
    const EncryptionParameters& parms = cd->parms();
    const uint32_t L = static_cast<uint32_t>(parms.coeff_modulus().size());
    const size_t n = parms.poly_modulus_degree();
    switch (parms.scheme()) {
        case SchemeType::BFV: {
            if (encrypted.is_ntt_form() != plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext and ciphertext are not in the same NTT form.");
            if (plain.parms_id() == parms_id_zero) {
                if (encrypted.is_ntt_form()) throw std::invalid_argument(std::string(P) + " When plain is mod t, encrypted must not be in NTT form.");
                if (plain.coeff_count() > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
                // scaling_variant::multiply_add/sub_plain_inplace: c0 +/- round(q/t * m) with the constants of THIS level
                troyn_check_public(troyn_bfv_scale_up(context_->behz(L), plain.poly(), plain.coeff_count(), n, encrypted.poly(0), static_cast<size_t>(L) * n,
                                                      encrypted.poly(0), static_cast<size_t>(L) * n, subtract ? 1 : 0, 1, stream()));
            } else {
                if (plain.parms_id() != encrypted.parms_id()) throw std::invalid_argument(std::string(P) + " Plaintext and ciphertext parameters do not match.");
                if (plain.coeff_count() != n) {
                    // a partial RNS plaintext (BatchEncoder::scale_up of a short polynomial): zero-padded to the full shape first
                    const utils::DynamicArray full = plain.expanded_rns(L, n, pool);
                    troyn_check_public((subtract ? troyn_sub : troyn_add)(context_->plan(), 0, L, encrypted.poly(0), full.raw_pointer(), encrypted.poly(0), 1, stream()));
                    break;   // `full` returns to the pool; the pool hands it back to this thread in stream order (MemoryPool, troy.h)
                }
                troyn_check_public((subtract ? troyn_sub : troyn_add)(context_->plan(), 0, L, encrypted.poly(0), plain.poly(), encrypted.poly(0), 1, stream()));
            }
            break;
        }
        case SchemeType::CKKS: {
            if (!encrypted.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Ciphertext is not in NTT form.");
            if (!are_close_double(plain.scale(), encrypted.scale())) throw std::invalid_argument(std::string(P) + " Plaintext scale is not equal to the scale of the ciphertext.");
            if (!plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext and ciphertext are not in the same NTT form.");
            // the reference adds the first L limbs of the plaintext whatever its level (utils::add_inplace_p over the ciphertext's
            // moduli, evaluator_translate_plain.cu:71-75): a plaintext encoded higher up the chain is a valid operand
            if (plain.parms_id() != encrypted.parms_id()) {
                auto pcd = context_->get_context_data(plain.parms_id());
                if (!pcd.has_value() || pcd.value()->chain_index() < cd->chain_index() || pcd.value()->parms().poly_modulus_degree() != n)
                    throw std::invalid_argument(std::string(P) + " Plaintext and ciphertext parameters do not match.");
            }
            troyn_check_public((subtract ? troyn_sub : troyn_add)(context_->plan(), 0, L, encrypted.poly(0), plain.poly(), encrypted.poly(0), 1, stream()));
            break;
        }
        case SchemeType::BGV: {
            // evaluator_translate_plain.cu:78-88: the plaintext (mod t) takes the ciphertext's correction factor, is lifted
            // centrally, transformed and added
            if (!encrypted.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Ciphertext is not in NTT form.");
            if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is in NTT form.");
            if (plain.coeff_count() > n) throw std::invalid_argument("[scaling_variant::centralize] plain_coeff_count exceeds the polynomial degree.");
            utils::DynamicArray scaled(plain.coeff_count(), true, pool), lifted(static_cast<size_t>(L) * n, true, pool);
            troyn_check_public(troyn_bgv_multiply_scalar_mod_t(context_->bgv(L), plain.poly(), encrypted.correction_factor(), scaled.raw_pointer(), plain.coeff_count(), stream()));
            troyn_check_public(troyn_plain_centralize(context_->plan(), L, parms.plain_modulus().value(), scaled.raw_pointer(), plain.coeff_count(), n, lifted.raw_pointer(), 1, stream()));
            troyn_check_public(troyn_ntt(context_->plan(), 0, lifted.raw_pointer(), lifted.raw_pointer(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, stream()));
            troyn_check_public((subtract ? troyn_sub : troyn_add)(context_->plan(), 0, L, encrypted.poly(0), lifted.raw_pointer(), encrypted.poly(0), 1, stream()));
            break;
        }
        default:
            throw std::logic_error(std::string(P) + " Scheme not implemented.");
    }
    // asynchronous on the thread's stream like the reference's method; the temporaries above return to the pool, which hands a block back to
    // the thread that released it in stream order (round 4: this was a stream wait per call -- 32 of them in MatmulHelper's bias add)
}

// ------------------------------------------------------------------------------------------------
// Evaluator: LWE extraction, shifts, field trace  (evaluator_lwes.cu:52-198)
// ------------------------------------------------------------------------------------------------
LWECiphertext Evaluator::extract_lwe_new(const Ciphertext& encrypted, size_t term, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::extract_lwe_new]";
    no_seed(P, encrypted);
    need_device(P, context_, encrypted);
    if (encrypted.polynomial_count() != 2) throw std::invalid_argument(std::string(P) + " Ciphertext size must be 2.");
    if (encrypted.is_ntt_form()) {
        Ciphertext transformed;
        transform_from_ntt(encrypted, transformed, pool);
        return extract_lwe_new(transformed, term, pool);
    }
    ContextDataPointer cd = level(P, context_, encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree();
    if (term >= n) throw std::invalid_argument(std::string(P) + " term out of range");
    LWECiphertext ret;
    ret.coeff_modulus_size() = L;
    ret.poly_modulus_degree() = n;
    ret.c0_dyn() = utils::DynamicArray(L, true, pool);
    ret.c1_dyn() = utils::DynamicArray(static_cast<size_t>(L) * n, true, pool);
    const uint64_t* src = encrypted.data().raw_pointer();
    const size_t bytes = troyn_extract_lwe_workspace_bytes(1);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    troyn_check_public(troyn_extract_lwe(context_->plan(), L, &src, &term, ret.c0_dyn().raw_pointer(), ret.c1_dyn().raw_pointer(), 1, ws.raw_pointer(), bytes, stream()));
    troyn_sync_current_stream();
    ret.parms_id() = encrypted.parms_id();
    ret.scale() = encrypted.scale();
    ret.correction_factor() = encrypted.correction_factor();
    return ret;
}

void Evaluator::divide_by_poly_modulus_degree_inplace(Ciphertext& encrypted, uint64_t mul) const {
    const char* P = "[Evaluator::divide_by_poly_modulus_degree_inplace]";
    need_device(P, context_, encrypted);
    ContextDataPointer cd = level(P, context_, encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    troyn_check_public(troyn_multiply_inv_degree(context_->plan(), 0, L, encrypted.data().raw_pointer(), encrypted.data().raw_pointer(), mul,
                                                 encrypted.polynomial_count(), stream()));
    troyn_sync_current_stream();
}

void Evaluator::negacyclic_shift(const Ciphertext& encrypted, size_t shift, Ciphertext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::negacyclic_shift]";
    no_seed(P, encrypted);
    need_device(P, context_, encrypted);
    ContextDataPointer cd = level(P, context_, encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    if (encrypted.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Ciphertext is in NTT form.");
    Ciphertext out = Ciphertext::like(encrypted, false, pool);
    troyn_check_public(troyn_negacyclic_shift(context_->plan(), 0, L, encrypted.data().raw_pointer(), out.data().raw_pointer(), shift,
                                              encrypted.polynomial_count(), stream()));
    troyn_sync_current_stream();
    destination = std::move(out);
}

void Evaluator::field_trace_inplace(Ciphertext& encrypted, const GaloisKeys& automorphism_keys, size_t logn, MemoryPoolHandle pool) const {
    size_t poly_degree = encrypted.poly_modulus_degree();
    Ciphertext temp;
    while (poly_degree > (static_cast<size_t>(1) << logn)) {
        apply_galois(encrypted, poly_degree + 1, automorphism_keys, temp, pool);
        add_inplace(encrypted, temp, pool);
        poly_degree >>= 1;
    }
}

void Evaluator::field_trace_inplace_batched(const std::vector<Ciphertext*>& encrypted, const GaloisKeys& automorphism_keys, size_t logn, MemoryPoolHandle pool) const {
    // evaluator_lwes.cu:111-139
    if (encrypted.empty()) return;
    size_t poly_degree = encrypted[0]->poly_modulus_degree();
    for (Ciphertext* c : encrypted)
        if (c->poly_modulus_degree() != poly_degree) throw std::invalid_argument("[Evaluator::field_trace_inplace_batched] Mismatched poly_modulus_degree.");
    std::vector<Ciphertext> temp(encrypted.size());
    std::vector<Ciphertext*> temp_ptrs;
    std::vector<const Ciphertext*> temp_const, enc_const(encrypted.begin(), encrypted.end());
    for (Ciphertext& t : temp) { temp_ptrs.push_back(&t); temp_const.push_back(&t); }
    while (poly_degree > (static_cast<size_t>(1) << logn)) {
        apply_galois_batched(enc_const, poly_degree + 1, automorphism_keys, temp_ptrs, pool);
        add_batched(enc_const, temp_const, encrypted, pool);
        poly_degree >>= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Evaluator: rescale_to, plaintext modulus switching  (evaluator_modswitch.cu:279-316, :402-443, :463-472)
// ------------------------------------------------------------------------------------------------
void Evaluator::rescale_to(const Ciphertext& encrypted, const ParmsID& parms_id, Ciphertext& destination, MemoryPoolHandle pool) const {
    ContextDataPointer cd = level("[Evaluator::rescale_to]", context_, encrypted.parms_id());
    ContextDataPointer target = level("[Evaluator::rescale_to]", context_, parms_id);
    if (cd->chain_index() < target->chain_index()) throw std::invalid_argument("[Evaluator::rescale_to] Cannot rescale to a higher level.");
    Ciphertext cur = encrypted.clone(pool);
    while (cur.parms_id() != parms_id) { Ciphertext next; rescale_to_next(cur, next, pool); cur = std::move(next); }
    destination = std::move(cur);
}

void Evaluator::mod_switch_plain_to(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::mod_switch_plain_to_inplace]";
    if (!plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is not in NTT form.");
    if (!plain.on_device() || !context_->on_device()) throw std::invalid_argument(std::string(P) + " Operands must be on the device (the evaluator runs on the GPU only).");
    ContextDataPointer cd = level(P, context_, plain.parms_id());
    ContextDataPointer target = level(P, context_, parms_id);
    if (cd->chain_index() < target->chain_index()) throw std::invalid_argument(std::string(P) + " Cannot switch to a higher level.");
    if (plain.parms_id() == parms_id) { destination = plain.clone(pool); return; }
    // kernel_mod_switch_drop_to with one polynomial: the first L_out limbs are kept
    const uint32_t L_in = static_cast<uint32_t>(cd->parms().coeff_modulus().size()), L_out = static_cast<uint32_t>(target->parms().coeff_modulus().size());
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, parms_id);
    troyn_check_public(troyn_mod_switch_drop(context_->plan(), L_in, L_out, plain.poly(), 1, out.poly(), 1, stream()));
    troyn_sync_current_stream();
    out.is_ntt_form() = true;
    out.scale() = plain.scale();
    destination = std::move(out);
}

void Evaluator::mod_switch_plain_to_next(const Plaintext& plain, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::mod_switch_plain_to_next]";
    if (context_->last_parms_id() == plain.parms_id()) throw std::invalid_argument(std::string(P) + " End of modulus switching chain reached.");
    ContextDataPointer cd = level(P, context_, plain.parms_id());
    if (!cd->next_context_data().has_value()) throw std::invalid_argument("[Evaluator::mod_switch_drop_to_plain_internal] Next context data is not set.");
    mod_switch_plain_to(plain, cd->next_context_data().value()->parms_id(), destination, pool);
}

// ------------------------------------------------------------------------------------------------
// Evaluator: RLWE packing  (evaluator_lwes.cu:200-681)
// ------------------------------------------------------------------------------------------------
Ciphertext Evaluator::pack_lwe_ciphertexts_new(const std::vector<const LWECiphertext*>& lwes, const GaloisKeys& automorphism_keys, MemoryPoolHandle pool,
                                               bool apply_field_trace) const {
    const char* P = "[Evaluator::pack_lwe_ciphertexts_new]";
    if (lwes.empty()) throw std::invalid_argument(std::string(P) + " LWE ciphertexts must not be empty.");
    for (const LWECiphertext* l : lwes)
        if (l->parms_id() != lwes[0]->parms_id()) throw std::invalid_argument(std::string(P) + " LWE ciphertexts must have same parms id.");
    ContextDataPointer cd = level(P, context_, lwes[0]->parms_id());
    const size_t n = cd->parms().poly_modulus_degree();
    if (lwes.size() > n) throw std::invalid_argument(std::string(P) + " LWE ciphertexts count must be less than poly_modulus_degree.");
    size_t l = 0;
    while ((static_cast<size_t>(1) << l) < lwes.size()) l++;
    std::vector<Ciphertext> rlwes = LWECiphertext::assemble_lwe_batched_new(lwes, pool);
    std::vector<const Ciphertext*> ptrs;
    for (const Ciphertext& c : rlwes) ptrs.push_back(&c);
    return pack_rlwe_ciphertexts_new(ptrs, automorphism_keys, 0, n, n >> l, pool, apply_field_trace);
}

std::vector<Ciphertext> Evaluator::pack_lwe_ciphertexts_new_batched(const std::vector<std::vector<const LWECiphertext*>>& lwe_groups, const GaloisKeys& automorphism_keys,
                                                                    MemoryPoolHandle pool, bool apply_field_trace) const {
    // evaluator_lwes.cu:232-300: every group goes through the packing tree together; the tree depth is set by the largest group
    const char* P = "[Evaluator::pack_lwe_ciphertexts_new]";
    std::vector<Ciphertext> out(lwe_groups.size());
    if (lwe_groups.empty()) return out;
    size_t max_count = 0;
    std::vector<const LWECiphertext*> flat;
    for (const auto& group : lwe_groups) {
        if (group.empty()) throw std::invalid_argument(std::string(P) + " LWE ciphertexts must not be empty.");
        max_count = std::max(max_count, group.size());
        for (const LWECiphertext* l : group) {
            if (l->parms_id() != lwe_groups[0][0]->parms_id()) throw std::invalid_argument(std::string(P) + " LWE ciphertexts must have same parms id.");
            flat.push_back(l);
        }
    }
    ContextDataPointer cd = level(P, context_, lwe_groups[0][0]->parms_id());
    const size_t n = cd->parms().poly_modulus_degree();
    if (max_count > n) throw std::invalid_argument(std::string(P) + " LWE ciphertexts count must be less than poly_modulus_degree.");
    size_t l = 0;
    while ((static_cast<size_t>(1) << l) < max_count) l++;
    std::vector<Ciphertext> rlwes = LWECiphertext::assemble_lwe_batched_new(flat, pool);
    std::vector<std::vector<const Ciphertext*>> groups(lwe_groups.size());
    size_t k = 0;
    for (size_t i = 0; i < lwe_groups.size(); i++)
        for (size_t j = 0; j < lwe_groups[i].size(); j++) groups[i].push_back(&rlwes[k++]);
    std::vector<Ciphertext*> ptrs;
    for (Ciphertext& c : out) ptrs.push_back(&c);
    pack_rlwe_ciphertexts_batched(groups, automorphism_keys, 0, n, n >> l, ptrs, pool, apply_field_trace);
    return out;
}

Ciphertext Evaluator::pack_rlwe_ciphertexts_new(const std::vector<const Ciphertext*>& ciphers, const GaloisKeys& automorphism_keys, size_t shift, size_t input_interval,
                                                size_t output_interval, MemoryPoolHandle pool, bool apply_field_trace) const {
    Ciphertext out;
    pack_rlwe_ciphertexts_batched({ciphers}, automorphism_keys, shift, input_interval, output_interval, {&out}, pool, apply_field_trace);
    return out;
}

void Evaluator::pack_rlwe_ciphertexts_batched(const std::vector<std::vector<const Ciphertext*>>& cipher_groups, const GaloisKeys& automorphism_keys, size_t shift,
                                              size_t input_interval, size_t output_interval, const std::vector<Ciphertext*>& outputs, MemoryPoolHandle pool,
                                              bool apply_field_trace) const {
    // evaluator_lwes.cu:491-681.  The reference keeps max_cipher_count ciphertext objects per group and runs, per layer, a
    // shift, a subtraction, an addition, a Galois automorphism and another addition over them.  Here every slot of every group
    // lives in one buffer (bit-reversed slot order, so each layer pairs neighbours); a layer is one fused kernel
    // (troyn_pack_layer) plus one key switch batched over all pairs of all groups, and halves the buffer.
    const char* P = "[Evaluator::pack_rlwe_ciphertexts_batched]";
    if (cipher_groups.size() != outputs.size()) throw std::invalid_argument(std::string(P) + " Input groups and outputs should have same size.");
    const size_t groups = cipher_groups.size();
    if (groups == 0) return;
    if (cipher_groups[0].empty()) throw std::invalid_argument(std::string(P) + " Input group 0 is empty.");
    const Ciphertext& first = *cipher_groups[0][0];
    const ParmsID parms_id = first.parms_id();
    const bool input_ntt_form = first.is_ntt_form();
    ContextDataPointer cd = level(P, context_, parms_id);
    const EncryptionParameters& parms = cd->parms();
    const SchemeType scheme = parms.scheme();
    const bool output_ntt_form = scheme == SchemeType::CKKS || scheme == SchemeType::BGV;      // evaluator_lwes.cu:522
    const size_t n = parms.poly_modulus_degree();
    const uint32_t L = static_cast<uint32_t>(parms.coeff_modulus().size());
    if (input_interval > n) throw std::invalid_argument(std::string(P) + " input_interval must be less than poly_modulus_degree.");
    if (output_interval > input_interval) throw std::invalid_argument(std::string(P) + " output_interval must be less than input_interval.");
    if (input_interval == 0 || !power_of_two(input_interval).second) throw std::invalid_argument(std::string(P) + " input_interval must be power of two.");
    if (output_interval == 0 || !power_of_two(output_interval).second) throw std::invalid_argument(std::string(P) + " output_interval must be power of two.");
    if (automorphism_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument("[Evaluator::apply_galois_inplace] Galois keys has incorrect parms id.");
    const size_t max_cipher_count = input_interval / output_interval;
    const size_t layers = power_of_two(max_cipher_count).first;
    for (size_t i = 0; i < groups; i++) {
        const auto& group = cipher_groups[i];
        if (group.empty()) throw std::invalid_argument(std::string(P) + " Input group " + std::to_string(i) + " is empty.");
        if (group.size() > max_cipher_count) throw std::invalid_argument(std::string(P) + " Input group " + std::to_string(i) + " has more than input_interval / output_interval.");
        for (size_t j = 0; j < group.size(); j++) {
            const std::string at = " Input[" + std::to_string(i) + "][" + std::to_string(j) + "]";
            no_seed(P, *group[j]);
            need_device(P, context_, *group[j]);
            if (group[j]->parms_id() != parms_id) throw std::invalid_argument(std::string(P) + at + " has different parms_id.");
            if (group[j]->is_ntt_form() != input_ntt_form) throw std::invalid_argument(std::string(P) + at + " has different ntt_form.");
            if (group[j]->polynomial_count() != 2) throw std::invalid_argument(std::string(P) + at + " has different polynomial count.");
            if (scheme == SchemeType::CKKS && group[j]->scale() != group[0]->scale()) throw std::invalid_argument(std::string(P) + at + " has different scale.");
            if (scheme == SchemeType::BGV && group[j]->correction_factor() != group[0]->correction_factor())
                throw std::invalid_argument(std::string(P) + at + " has different correction factor.");      // evaluator_lwes.cu:566-570
        }
    }
    for (size_t layer = 0; layer < layers; layer++) {
        const size_t g = (n / input_interval) * (static_cast<size_t>(1) << (layer + 1)) + 1;
        if (!automorphism_keys.has_key(g)) throw std::invalid_argument("[Evaluator::apply_galois_inplace] Galois key not present.");
    }
    const troyn_plan* plan = context_->plan();
    hipStream_t s = stream();
    const size_t pc = static_cast<size_t>(L) * n, words = 2 * pc, slots = groups * max_cipher_count;

    // sources in coefficient form; NTT-form inputs are gathered and transformed in one launch first
    std::vector<const uint64_t*> src(slots, nullptr);
    utils::DynamicArray staged(0, true, pool);
    if (input_ntt_form) {
        size_t total = 0;
        for (const auto& group : cipher_groups) total += group.size();
        staged = utils::DynamicArray(total * words, true, pool);
        size_t k = 0;
        for (const auto& group : cipher_groups)
            for (const Ciphertext* c : group) hip_ok(hipMemcpyAsync(staged.raw_pointer() + (k++) * words, c->data().raw_pointer(), words * 8, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
        troyn_check_public(troyn_ntt(plan, 1, staged.raw_pointer(), staged.raw_pointer(), total, 2, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    }
    size_t flat = 0;
    for (size_t j = 0; j < groups; j++) {
        for (size_t i = 0; i < max_cipher_count; i++) {
            const size_t index = reverse_bits(i, layers);
            if (index < cipher_groups[j].size())
                src[j * max_cipher_count + i] = input_ntt_form ? staged.raw_pointer() + (flat + index) * words : cipher_groups[j][index]->data().raw_pointer();
        }
        flat += cipher_groups[j].size();
    }
    auto current = std::make_shared<utils::DynamicArray>(slots * words, true, pool);
    {
        const size_t bytes = troyn_pack_prepare_workspace_bytes(slots);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        troyn_check_public(troyn_pack_prepare(plan, L, 2, src.data(), slots, n / input_interval, shift, current->raw_pointer(), ws.raw_pointer(), bytes, s));
    }
    size_t count = slots;
    for (size_t layer = 0; layer < layers; layer++) {
        const size_t pairs = count / 2;
        const size_t g = (n / input_interval) * (static_cast<size_t>(1) << (layer + 1)) + 1;
        auto next = std::make_shared<utils::DynamicArray>(pairs * words, true, pool);
        utils::DynamicArray target(pairs * pc, true, pool);
        troyn_check_public(troyn_pack_layer(plan, L, g, input_interval >> (layer + 1), current->raw_pointer(), next->raw_pointer(), target.raw_pointer(), pairs, s));
        const std::vector<const uint64_t*> keys = automorphism_keys.get_data_ptrs(GaloisKeys::get_index(g));
        if (keys.size() < L) throw std::invalid_argument("[Evaluator::switch_key_inplace_internal] Key switching key has too few components for this level.");
        const size_t bytes = troyn_switch_key_workspace_bytes(plan, L, pairs);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        if (scheme == SchemeType::BGV) {
            // BGV key switching is defined on NTT-form operands (its tail removes the special prime AND keeps the result a multiple of t:
            // ski_util5, evaluator_keyswitching_core.cu:998-1030), so the reference transforms the odd member around apply_galois
            // (evaluator_lwes.cu:655-657).  Here: the permuted c1 to NTT form, the BGV key switch into a temporary, back to coefficient
            // form, added to the pair's sum -- the same words (key switching is linear in its target; the additions are exact).
            const size_t K = context_->key_context_data().value()->parms().coeff_modulus().size();
            utils::DynamicArray switched(pairs * words, true, pool);
            troyn_check_public(troyn_ntt(plan, 0, target.raw_pointer(), target.raw_pointer(), pairs, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
            troyn_check_public(troyn_bgv_switch_key(context_->bgv(K), L, target.raw_pointer(), keys.data(), TROYN_ASSIGN_OVERWRITE, switched.raw_pointer(), ws.raw_pointer(), bytes, pairs, s));
            troyn_check_public(troyn_ntt(plan, 1, switched.raw_pointer(), switched.raw_pointer(), pairs, 2, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
            troyn_check_public(troyn_add(plan, 0, L, next->raw_pointer(), switched.raw_pointer(), next->raw_pointer(), pairs * 2, s));
        } else
        troyn_check_public(troyn_switch_key(plan, L, scheme == SchemeType::CKKS, 0, target.raw_pointer(), keys.data(), TROYN_ASSIGN_ADD_INPLACE, next->raw_pointer(),
                                            ws.raw_pointer(), bytes, pairs, s));
        // `target`, `ws` and the previous buffer return to the pool here: stream order protects them (same thread, same stream), no wait per layer
        current = std::move(next);
        count = pairs;
    }
    if (output_ntt_form) troyn_check_public(troyn_ntt(plan, 0, current->raw_pointer(), current->raw_pointer(), groups, 2, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    troyn_sync_current_stream();
    // with no layers (max_cipher_count == 1) `current` still holds one slot per group
    for (size_t j = 0; j < groups; j++) {
        const Ciphertext& like = *cipher_groups[j][0];
        *outputs[j] = Ciphertext::from_members(2, L, n, parms_id, like.scale(), output_ntt_form, like.correction_factor(), 0,
                                               utils::DynamicArray::device_view(current->raw_pointer() + j * words, words, current));
    }
    if (output_interval != 1 && apply_field_trace) field_trace_inplace_batched(outputs, automorphism_keys, power_of_two(n / output_interval).first, pool);
}

}  // namespace troy
