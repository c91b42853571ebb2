// The Evaluator's x_batched(vector<const T*>, vector<T*>, pool) forms (evaluator.h; batch_utils.h in the reference).
//
// The reference builds device arrays of slice pointers per call and lets every kernel thread loop over the batch.  Here a
// uniform batch is ONE contiguous block [count][polys][limbs][N]: scattered operands are staged by a single gather launch
// (operands that are already adjacent windows of one buffer -- which is what these functions return -- are used in place),
// the C-ABI entry runs once with batch = count, and the results are windows of one shared buffer.  Item 0 goes through the
// per-object function first: it performs every argument check of the reference and fixes the result's shape and metadata.
#include <hip/hip_runtime.h>

#include "troy.h"

namespace troy {

namespace {

hipStream_t stream() { return static_cast<hipStream_t>(troyn_current_stream()); }

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("[kernel_provider::") + what + "] " + hipGetErrorString(e));
}

// same level, shape, form and scale as item 0, on the device, no seed
bool uniform(const std::vector<const Ciphertext*>& v) {
    if (v.empty()) return false;
    const Ciphertext& a = *v[0];
    for (const Ciphertext* c : v)
        if (!c->on_device() || c->contains_seed() || c->parms_id() != a.parms_id() || c->polynomial_count() != a.polynomial_count() ||
            c->coeff_modulus_size() != a.coeff_modulus_size() || c->is_ntt_form() != a.is_ntt_form() || c->scale() != a.scale() ||
            c->correction_factor() != a.correction_factor() || c->data().size() != a.data().size())
            return false;
    return true;
}

// [count][words] contiguous: the operands themselves when they are adjacent windows of one buffer, else a staged copy
const uint64_t* contiguous(const std::vector<const Ciphertext*>& v, utils::DynamicArray& staged, MemoryPoolHandle pool) {
    const size_t count = v.size(), words = v[0]->data().size();
    const utils::DynamicArray* owner = v[0]->data().view_owner();
    const uint64_t* base = v[0]->data().raw_pointer();
    bool adjacent = owner != nullptr;
    for (size_t i = 0; i < count && adjacent; i++) adjacent = v[i]->data().view_owner() == owner && v[i]->data().raw_pointer() == base + i * words;
    if (adjacent) return base;
    staged = utils::DynamicArray(count * words, true, pool);
    std::vector<const uint64_t*> src(count);
    for (size_t i = 0; i < count; i++) src[i] = v[i]->data().raw_pointer();
    const size_t bytes = troyn_gather_workspace_bytes(count);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    troyn_check_public(troyn_gather(src.data(), count, words, staged.raw_pointer(), ws.raw_pointer(), bytes, stream()));   // (`src` is consumed by the call)
    return staged.raw_pointer();
}

// the results: windows of one buffer shaped like `proto` (the per-object result of item 0)
std::shared_ptr<utils::DynamicArray> result_block(const Ciphertext& proto, size_t count, MemoryPoolHandle pool) {
    return std::make_shared<utils::DynamicArray>(count * proto.data().size(), true, pool);
}

void assign_views(const Ciphertext& proto, const std::shared_ptr<utils::DynamicArray>& block, const std::vector<Ciphertext*>& destination) {
    const size_t words = proto.data().size();
    for (size_t i = 0; i < destination.size(); i++)
        *destination[i] = Ciphertext::from_members(proto.polynomial_count(), proto.coeff_modulus_size(), proto.poly_modulus_degree(), proto.parms_id(), proto.scale(),
                                                   proto.is_ntt_form(), proto.correction_factor(), 0,
                                                   utils::DynamicArray::device_view(block->raw_pointer() + i * words, words, block));
}

std::vector<const Ciphertext*> as_const(const std::vector<Ciphertext*>& v) { return std::vector<const Ciphertext*>(v.begin(), v.end()); }

void same_size(const char* prompt, size_t a, size_t b) {
    if (a != b) throw std::invalid_argument(std::string(prompt) + " Input and destination have different sizes.");
}

}  // namespace

// -- negate ------------------------------------------------------------------------------------------------------------
void Evaluator::negate_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    same_size("[Evaluator::negate_batched]", encrypted.size(), destination.size());
    if (encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext d; negate(*encrypted[i], d, pool); *destination[i] = std::move(d); }
        return;
    }
    Ciphertext proto;
    negate(*encrypted[0], proto, pool);
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, encrypted.size(), pool);
    troyn_check_public(troyn_negate(context_->plan(), 0, static_cast<uint32_t>(proto.coeff_modulus_size()), in, block->raw_pointer(),
                                    encrypted.size() * proto.polynomial_count(), stream()));
    assign_views(proto, block, destination);
}

void Evaluator::negate_inplace_batched(const std::vector<Ciphertext*>& encrypted, MemoryPoolHandle pool) const { negate_batched(as_const(encrypted), encrypted, pool); }

// -- add / sub ---------------------------------------------------------------------------------------------------------
void Evaluator::translate_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d, bool subtract,
                                  MemoryPoolHandle pool) const {
    if (e1.size() != e2.size() || e1.size() != d.size()) throw std::invalid_argument("[Evaluator::translate_batched] Input and destination have different sizes.");
    const bool batched = e1.size() >= BATCH_OP_THRESHOLD && uniform(e1) && uniform(e2) && e1[0]->polynomial_count() == e2[0]->polynomial_count() &&
                         e1[0]->correction_factor() == e2[0]->correction_factor();   // BGV operands with different factors are balanced one by one
    if (!batched) {
        for (size_t i = 0; i < e1.size(); i++) { Ciphertext out; translate(*e1[i], *e2[i], out, subtract, pool); *d[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    translate(*e1[0], *e2[0], proto, subtract, pool);
    utils::DynamicArray s1(0, true, pool), s2(0, true, pool);
    const uint64_t* a = contiguous(e1, s1, pool);
    const uint64_t* b = contiguous(e2, s2, pool);
    auto block = result_block(proto, e1.size(), pool);
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    troyn_check_public((subtract ? troyn_sub : troyn_add)(context_->plan(), 0, L, a, b, block->raw_pointer(), e1.size() * proto.polynomial_count(), stream()));
    assign_views(proto, block, d);
}

void Evaluator::add_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    translate_batched(e1, e2, d, false, pool);
}
void Evaluator::sub_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    translate_batched(e1, e2, d, true, pool);
}

// -- multiply ----------------------------------------------------------------------------------------------------------
void Evaluator::multiply_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    if (e1.size() != e2.size() || e1.size() != d.size()) throw std::invalid_argument("[Evaluator::multiply_batched] Input and destination have different sizes.");
    if (e1.size() < BATCH_OP_THRESHOLD || !uniform(e1) || !uniform(e2)) {
        for (size_t i = 0; i < e1.size(); i++) { Ciphertext out; multiply(*e1[i], *e2[i], out, pool); *d[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    multiply_prepare(*e1[0], *e2[0], proto, pool);   // checks + shape of item 0 (a uniform batch), no device work
    const size_t count = e1.size(), p1 = e1[0]->polynomial_count(), p2 = e2[0]->polynomial_count();
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    utils::DynamicArray s1(0, true, pool), s2(0, true, pool);
    const uint64_t* a = contiguous(e1, s1, pool);
    const uint64_t* b = contiguous(e2, s2, pool);
    auto block = result_block(proto, count, pool);
    if (context_->key_context_data().value()->parms().scheme() == SchemeType::BFV) {
        const troyn_behz* bz = context_->behz(L);
        const size_t bytes = troyn_bfv_multiply_workspace_bytes(bz, p1, p2, count);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        troyn_check_public(troyn_bfv_multiply(bz, a, p1, b, p2, block->raw_pointer(), ws.raw_pointer(), bytes, count, stream()));
    } else {
        troyn_check_public(troyn_dyadic_convolute(context_->plan(), 0, L, a, p1, b, p2, block->raw_pointer(), count, stream()));
    }
    assign_views(proto, block, d);
}

// -- multiply -> relinearize -> rescale_to_next, one call for the whole batch (addition) ------------------------------------------------
void Evaluator::multiply_relinearize_rescale_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const RelinKeys& relin_keys,
                                                     const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    if (e1.size() != e2.size() || e1.size() != destination.size())
        throw std::invalid_argument("[Evaluator::multiply_relinearize_rescale_batched] Input and destination have different sizes.");
    uint32_t L = 0; ParmsID next; double scale = 1.0; std::vector<const uint64_t*> keys;
    const bool batched = e1.size() >= BATCH_OP_THRESHOLD && uniform(e1) && uniform(e2) &&
                         multiply_relinearize_rescale_prepare(*e1[0], *e2[0], relin_keys, L, next, scale, keys);   // item 0 stands for every item of a uniform batch
    if (!batched) {
        for (size_t i = 0; i < e1.size(); i++) { Ciphertext out; multiply_relinearize_rescale(*e1[i], *e2[i], relin_keys, out, pool); *destination[i] = std::move(out); }
        return;
    }
    const size_t count = e1.size();
    const size_t n = e1[0]->poly_modulus_degree();
    const uint64_t cf = e1[0]->correction_factor();
    utils::DynamicArray s1(0, true, pool), s2(0, true, pool);
    const uint64_t* a = contiguous(e1, s1, pool);
    const uint64_t* b = contiguous(e2, s2, pool);
    const size_t words = (size_t)2 * (L - 1) * n;
    auto block = std::make_shared<utils::DynamicArray>(count * words, true, pool);
    const size_t bytes = troyn_ckks_multiply_relinearize_rescale_workspace_bytes(context_->plan(), L, count);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    troyn_check_public(troyn_ckks_multiply_relinearize_rescale(context_->plan(), L, a, b, keys.data(), block->raw_pointer(), ws.raw_pointer(), bytes, count, stream()));
    // no synchronisation: the call is asynchronous on the thread's stream like the reference's evaluator methods; `ws` returns to the pool, which
    // hands a block back to the thread that released it in stream order and to any other thread only after a device synchronisation (MemoryPool).
    // "In stream order" holds by construction: every launch of this mirror goes to the calling thread's stream (troy.cpp current_stream(); with call
    // combining on, to the one shared stream, and the pool then treats every thread as the same owner); there is no
    // API through which a host thread could move its work to another stream between the release and the reuse.
    for (size_t i = 0; i < count; i++)
        *destination[i] = Ciphertext::from_members(2, L - 1, n, next, scale, true, cf, 0,
                                                   utils::DynamicArray::device_view(block->raw_pointer() + i * words, words, block));
}

// -- relinearize -------------------------------------------------------------------------------------------------------
void Evaluator::relinearize_batched(const std::vector<const Ciphertext*>& encrypted, const RelinKeys& relin_keys, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    same_size("[Evaluator::relinearize_batched]", encrypted.size(), d.size());
    const bool bgv = context_->key_context_data().value()->parms().scheme() == SchemeType::BGV;   // ski_util5 tail: per-object path
    if (bgv || encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted) || encrypted[0]->polynomial_count() != 3) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; relinearize_internal(*encrypted[i], relin_keys, 2, out, pool); *d[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    std::vector<const uint64_t*> keys;
    relinearize_prepare(*encrypted[0], relin_keys, proto, keys, pool);   // checks + shape of item 0, no device work
    const size_t count = encrypted.size();
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, count, pool);
    const size_t bytes = troyn_relinearize_workspace_bytes(context_->plan(), L, count);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    const bool ckks = context_->key_context_data().value()->parms().scheme() == SchemeType::CKKS;
    troyn_check_public(troyn_relinearize(context_->plan(), L, ckks, proto.is_ntt_form(), in, keys.data(), block->raw_pointer(), ws.raw_pointer(), bytes, count, stream()));
    assign_views(proto, block, d);
}

// -- modulus switching -------------------------------------------------------------------------------------------------
void Evaluator::mod_switch_to_next_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    same_size("[Evaluator::mod_switch_to_next_batched]", encrypted.size(), destination.size());
    const bool bgv = context_->key_context_data().value()->parms().scheme() == SchemeType::BGV;
    if (bgv || encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; mod_switch_to_next(*encrypted[i], out, pool); *destination[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    mod_switch_to_next(*encrypted[0], proto, pool);
    const size_t count = encrypted.size(), pc = proto.polynomial_count();
    const uint32_t L = static_cast<uint32_t>(encrypted[0]->coeff_modulus_size());
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, count, pool);
    if (context_->first_context_data().value()->parms().scheme() == SchemeType::BFV)
        troyn_check_public(troyn_divide_and_round_q_last(context_->plan(), L, in, pc, block->raw_pointer(), count, stream()));
    else
        troyn_check_public(troyn_mod_switch_drop(context_->plan(), L, static_cast<uint32_t>(proto.coeff_modulus_size()), in, pc, block->raw_pointer(), count, stream()));
    assign_views(proto, block, destination);
}

void Evaluator::rescale_to_next_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    same_size("[Evaluator::rescale_to_next_batched]", encrypted.size(), destination.size());
    if (encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; rescale_to_next(*encrypted[i], out, pool); *destination[i] = std::move(out); }
        return;
    }
    // rescale_to_next's own checks (evaluator_modswitch.cu:445-461), then checks + shape of item 0 without device work
    if (encrypted[0]->contains_seed()) throw std::invalid_argument("[Evaluator::rescale_to_next] Argument contains seed.");
    if (context_->last_parms_id() == encrypted[0]->parms_id()) throw std::invalid_argument("[Evaluator::rescale_to_next] End of modulus switching chain reached.");
    if (context_->first_context_data().value()->parms().scheme() != SchemeType::CKKS) throw std::invalid_argument("[Evaluator::rescale_to_next] Cannot rescale BFV/BGV ciphertext.");
    Ciphertext proto;
    mod_switch_scale_prepare(*encrypted[0], proto, pool);
    const size_t count = encrypted.size(), pc = proto.polynomial_count();
    const uint32_t L = static_cast<uint32_t>(encrypted[0]->coeff_modulus_size());
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, count, pool);
    const size_t bytes = troyn_divide_and_round_q_last_ntt_workspace_bytes(context_->plan(), L, pc, count);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    troyn_check_public(troyn_divide_and_round_q_last_ntt(context_->plan(), L, in, pc, block->raw_pointer(), ws.raw_pointer(), bytes, count, stream()));
    assign_views(proto, block, destination);
}

// -- NTT ---------------------------------------------------------------------------------------------------------------
static void ntt_batched(const Evaluator& ev, const HeContextPointer& context, bool inverse, const std::vector<const Ciphertext*>& encrypted,
                        const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) {
    same_size(inverse ? "[Evaluator::transform_from_ntt_batched]" : "[Evaluator::transform_to_ntt_batched]", encrypted.size(), destination.size());
    if (encrypted.size() < Evaluator::BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) {
            Ciphertext out;
            if (inverse) ev.transform_from_ntt(*encrypted[i], out, pool); else ev.transform_to_ntt(*encrypted[i], out, pool);
            *destination[i] = std::move(out);
        }
        return;
    }
    Ciphertext proto;
    if (inverse) ev.transform_from_ntt(*encrypted[0], proto, pool); else ev.transform_to_ntt(*encrypted[0], proto, pool);
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, encrypted.size(), pool);
    troyn_check_public(troyn_ntt(context->plan(), inverse ? 1 : 0, in, block->raw_pointer(), encrypted.size(), proto.polynomial_count(), L, 0, L, TROYN_IDX_COMPONENTWISE, 0,
                                 stream()));
    assign_views(proto, block, destination);
}

void Evaluator::transform_to_ntt_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    ntt_batched(*this, context_, false, encrypted, destination, pool);
}
void Evaluator::transform_from_ntt_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    ntt_batched(*this, context_, true, encrypted, destination, pool);
}
void Evaluator::transform_to_ntt_inplace_batched(const std::vector<Ciphertext*>& encrypted, MemoryPoolHandle pool) const {
    ntt_batched(*this, context_, false, as_const(encrypted), encrypted, pool);
}
void Evaluator::transform_from_ntt_inplace_batched(const std::vector<Ciphertext*>& encrypted, MemoryPoolHandle pool) const {
    ntt_batched(*this, context_, true, as_const(encrypted), encrypted, pool);
}

// -- Galois automorphism -----------------------------------------------------------------------------------------------
void Evaluator::apply_galois_batched(const std::vector<const Ciphertext*>& encrypted, size_t galois_element, const GaloisKeys& galois_keys,
                                     const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    same_size("[Evaluator::apply_galois_batched]", encrypted.size(), destination.size());
    const bool bgv = context_->key_context_data().value()->parms().scheme() == SchemeType::BGV;
    if (bgv || encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; apply_galois(*encrypted[i], galois_element, galois_keys, out, pool); *destination[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    std::vector<const uint64_t*> keys;
    apply_galois_prepare(*encrypted[0], galois_element, galois_keys, proto, keys, pool);   // checks + shape of item 0, no device work
    const size_t count = encrypted.size(), n = proto.poly_modulus_degree();
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    const size_t pc = static_cast<size_t>(L) * n;
    const troyn_plan* plan = context_->plan();
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, count, pool);
    // permute (c0, c1) of every item, take the permuted c1s as key-switch targets, overwrite them with the switched result
    troyn_check_public(troyn_apply_galois(plan, 0, L, proto.is_ntt_form() ? 1 : 0, galois_element, in, block->raw_pointer(), count * 2, stream()));
    utils::DynamicArray target(count * pc, true, pool);
    hip_ok(hipMemcpy2DAsync(target.raw_pointer(), pc * 8, block->raw_pointer() + pc, 2 * pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    const size_t bytes = troyn_switch_key_workspace_bytes(plan, L, count);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    const bool ckks = context_->key_context_data().value()->parms().scheme() == SchemeType::CKKS;
    troyn_check_public(troyn_switch_key(plan, L, ckks, proto.is_ntt_form(), target.raw_pointer(), keys.data(), TROYN_ASSIGN_OVERWRITE_EXCEPT_FIRST, block->raw_pointer(),
                                        ws.raw_pointer(), bytes, count, stream()));
    assign_views(proto, block, destination);
}

// -- ciphertext +/- plaintext, ciphertext x plaintext ----------------------------------------------------------------------
void Evaluator::translate_plain_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination,
                                        bool subtract, MemoryPoolHandle pool) const {
    if (encrypted.size() != plain.size() || encrypted.size() != destination.size())
        throw std::invalid_argument("[Evaluator::translate_plain_batched] Input and destination have different sizes.");
    bool batched = encrypted.size() >= BATCH_OP_THRESHOLD && uniform(encrypted) && context_->key_context_data().value()->parms().scheme() == SchemeType::BFV;
    for (const Plaintext* p : plain) batched = batched && p->parms_id() == parms_id_zero && !p->is_ntt_form() && p->on_device();
    if (!batched) {
        for (size_t i = 0; i < encrypted.size(); i++) {
            Ciphertext out = encrypted[i]->clone(pool);
            translate_plain_inplace(out, *plain[i], subtract, pool);
            *destination[i] = std::move(out);
        }
        return;
    }
    // BFV, plaintexts mod t: c0 +/- round(q/t * m) for the whole batch (scaling_variant::multiply_add_plain_inplace)
    Ciphertext proto = encrypted[0]->clone(pool);
    translate_plain_inplace(proto, *plain[0], subtract, pool);
    const size_t count = encrypted.size(), n = proto.poly_modulus_degree(), L = proto.coeff_modulus_size(), words = proto.data().size();
    utils::DynamicArray staged(0, true, pool), plains(count * n, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    plains.set_zero();
    for (size_t i = 0; i < count; i++) {
        if (plain[i]->coeff_count() > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
        hip_ok(hipMemcpyAsync(plains.raw_pointer() + i * n, plain[i]->poly(), plain[i]->coeff_count() * 8, hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    }
    auto block = result_block(proto, count, pool);
    hip_ok(hipMemcpyAsync(block->raw_pointer(), in, count * words * 8, hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    troyn_check_public(troyn_bfv_scale_up(context_->behz(L), plains.raw_pointer(), n, n, in, words, block->raw_pointer(), words, subtract ? 1 : 0, count, stream()));
    assign_views(proto, block, destination);
}

void Evaluator::multiply_plain_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination,
                                       MemoryPoolHandle pool) const {
    if (encrypted.size() != plain.size() || encrypted.size() != destination.size())
        throw std::invalid_argument("[Evaluator::multiply_plain_batched] Input and destination have different sizes.");
    bool ntt = !encrypted.empty();
    for (size_t i = 0; i < encrypted.size() && ntt; i++) ntt = encrypted[i]->is_ntt_form() && plain[i]->is_ntt_form();
    bool distinct = true;
    for (size_t i = 0; i < destination.size() && distinct; i++)
        for (size_t k = 0; k < i && distinct; k++) distinct = destination[k] != destination[i];
    if (ntt && distinct && uniform(encrypted)) {
        // evaluator_multiply_plain.cu:356-385 (multiply_plain_ntt_batched): one launch over all (ciphertext, plaintext) pairs;
        // an in-place call (destination[i] == encrypted[i]) goes through temporaries
        std::vector<Ciphertext> tmp(encrypted.size());
        std::vector<Ciphertext*> tp;
        for (Ciphertext& c : tmp) tp.push_back(&c);
        multiply_plain_accumulate(encrypted, plain, tp, true, pool);
        for (size_t i = 0; i < tmp.size(); i++) *destination[i] = std::move(tmp[i]);
        return;
    }
    // evaluator_multiply_plain.cu:70-194 (multiply_plain_normal_batched): coefficient-form ciphertexts times plaintexts modulo t (BFV) --
    // the showcase of examples/15_batched_operation.cu.  One gather, one centralize launch and one NTT launch for the plaintexts, one
    // NTT launch for the ciphertexts, one product launch, one inverse NTT launch.
    bool normal = encrypted.size() >= BATCH_OP_THRESHOLD && distinct && uniform(encrypted) && !encrypted[0]->is_ntt_form() &&
                  context_->key_context_data().value()->parms().scheme() == SchemeType::BFV;
    for (size_t i = 0; i < plain.size() && normal; i++) normal = plain[i]->parms_id() == parms_id_zero && !plain[i]->is_ntt_form() && plain[i]->on_device();
    if (normal) {
        Ciphertext proto;
        multiply_plain(*encrypted[0], *plain[0], proto, pool);           // every check of the per-object form; fixes shape and metadata
        const size_t count = encrypted.size(), n = proto.poly_modulus_degree(), pcnt = proto.polynomial_count();
        const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
        const troyn_plan* plan = context_->plan();
        const uint64_t t = context_->first_context_data().value()->parms().plain_modulus().value();
        utils::DynamicArray staged(0, true, pool), plains(count * n, true, pool), lifted(count * L * n, true, pool);
        const uint64_t* in = contiguous(encrypted, staged, pool);
        plains.set_zero();
        for (size_t i = 0; i < count; i++) {
            if (plain[i]->coeff_count() > n) throw std::invalid_argument("[scaling_variant::centralize] plain_coeff_count exceeds the polynomial degree.");
            hip_ok(hipMemcpyAsync(plains.raw_pointer() + i * n, plain[i]->poly(), plain[i]->coeff_count() * 8, hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
        }
        auto block = result_block(proto, count, pool);
        troyn_check_public(troyn_plain_centralize_ntt(plan, L, t, plains.raw_pointer(), n, n, lifted.raw_pointer(), count, stream()));
        troyn_check_public(troyn_ntt(plan, 0, in, block->raw_pointer(), count, pcnt, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, stream()));
        troyn_check_public(troyn_dyadic_broadcast_product(plan, 0, L, block->raw_pointer(), pcnt, lifted.raw_pointer(), static_cast<size_t>(L) * n, block->raw_pointer(), count, stream()));
        troyn_check_public(troyn_ntt(plan, 1, block->raw_pointer(), block->raw_pointer(), count, pcnt, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, stream()));
        assign_views(proto, block, destination);
        return;
    }
    for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; multiply_plain(*encrypted[i], *plain[i], out, pool); *destination[i] = std::move(out); }
}

}  // namespace troy

namespace troy {

// -- key switching of a batch (evaluator_keyswitching.cu:52-93) -------------------------------------------------------------------
void Evaluator::apply_keyswitching_batched(const std::vector<const Ciphertext*>& encrypted, const KSwitchKeys& kswitch_keys, const std::vector<Ciphertext*>& destination,
                                           MemoryPoolHandle pool) const {
    same_size("[Evaluator::apply_keyswitching_batched]", encrypted.size(), destination.size());
    const bool bgv = context_->key_context_data().value()->parms().scheme() == SchemeType::BGV;
    if (bgv || encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; apply_keyswitching(*encrypted[i], kswitch_keys, out, pool); *destination[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    apply_keyswitching(*encrypted[0], kswitch_keys, proto, pool);      // all the checks of the per-object form
    const size_t count = encrypted.size(), n = proto.poly_modulus_degree();
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    const size_t pc = static_cast<size_t>(L) * n;
    const troyn_plan* plan = context_->plan();
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, count, pool);
    // (c0, c1) -> (c0 + ks0, ks1): the block starts as a copy, the c1s are the key-switch targets
    hip_ok(hipMemcpyAsync(block->raw_pointer(), in, count * 2 * pc * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    utils::DynamicArray target(count * pc, true, pool);
    hip_ok(hipMemcpy2DAsync(target.raw_pointer(), pc * 8, in + pc, 2 * pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, stream()), "copy_device_to_device");
    const std::vector<const uint64_t*> keys = kswitch_keys.get_data_ptrs(0);
    const size_t bytes = troyn_switch_key_workspace_bytes(plan, L, count);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    const bool ckks = context_->key_context_data().value()->parms().scheme() == SchemeType::CKKS;
    troyn_check_public(troyn_switch_key(plan, L, ckks, proto.is_ntt_form(), target.raw_pointer(), keys.data(), TROYN_ASSIGN_OVERWRITE_EXCEPT_FIRST, block->raw_pointer(),
                                        ws.raw_pointer(), bytes, count, stream()));
    assign_views(proto, block, destination);
}

// -- rotations of a batch ------------------------------------------------------------------------------------------------------------
void Evaluator::rotate_internal_batched(const std::vector<const Ciphertext*>& encrypted, int steps, const GaloisKeys& galois_keys, const std::vector<Ciphertext*>& destination,
                                        MemoryPoolHandle pool) const {
    // evaluator_keyswitching.cu:263-294 for every member; the Galois elements depend on the step only, so each NAF term is one apply_galois_batched
    const char* P = "[Evaluator::rotate_inplace_internal]";
    same_size(P, encrypted.size(), destination.size());
    if (encrypted.empty()) return;
    if (galois_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Galois keys has incorrect parms id.");
    if (steps == 0) {
        for (size_t i = 0; i < encrypted.size(); i++) if (destination[i] != encrypted[i]) *destination[i] = encrypted[i]->clone(pool);
        return;
    }
    const size_t n = encrypted[0]->poly_modulus_degree();
    const size_t element = utils::galois_element_from_step(n, steps);
    if (galois_keys.has_key(element)) { apply_galois_batched(encrypted, element, galois_keys, destination, pool); return; }
    const std::vector<int> naf_steps = utils::naf(steps);
    if (naf_steps.size() == 1) throw std::invalid_argument(std::string(P) + " Galois key not present.");
    bool first = true;
    for (int st : naf_steps) {
        if (first) { rotate_internal_batched(encrypted, st, galois_keys, destination, pool); first = false; }
        else rotate_internal_batched(as_const(destination), st, galois_keys, destination, pool);
    }
}

void Evaluator::rotate_rows_batched(const std::vector<const Ciphertext*>& e, int steps, const GaloisKeys& k, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::invalid_argument("[Evaluator::rotate_rows_inplace] Rotate rows only applies for BFV or BGV");
    rotate_internal_batched(e, steps, k, d, pool);
}

void Evaluator::rotate_vector_batched(const std::vector<const Ciphertext*>& e, int steps, const GaloisKeys& k, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    if (context_->key_context_data().value()->parms().scheme() != SchemeType::CKKS) throw std::invalid_argument("[Evaluator::rotate_vector_inplace] Rotate vector only applies for CKKS");
    rotate_internal_batched(e, steps, k, d, pool);
}

void Evaluator::rotate_columns_batched(const std::vector<const Ciphertext*>& e, const GaloisKeys& k, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::invalid_argument("[Evaluator::rotate_columns_inplace] Rotate columns only applies for BFV or BGV");
    same_size("[Evaluator::conjugate_internal_batched]", e.size(), d.size());
    if (e.empty()) return;
    apply_galois_batched(e, utils::galois_element_from_step(e[0]->poly_modulus_degree(), 0), k, d, pool);
}

void Evaluator::complex_conjugate_batched(const std::vector<const Ciphertext*>& e, const GaloisKeys& k, const std::vector<Ciphertext*>& d, MemoryPoolHandle pool) const {
    if (context_->key_context_data().value()->parms().scheme() != SchemeType::CKKS)
        throw std::invalid_argument("[Evaluator::complex_conjugate_inplace] Complex conjugate only applies for CKKS");
    same_size("[Evaluator::conjugate_internal_batched]", e.size(), d.size());
    if (e.empty()) return;
    apply_galois_batched(e, utils::galois_element_from_step(e[0]->poly_modulus_degree(), 0), k, d, pool);
}

// -- modulus switching down to a level -------------------------------------------------------------------------------------------------
void Evaluator::mod_switch_to_batched(const std::vector<const Ciphertext*>& encrypted, const ParmsID& parms_id, const std::vector<Ciphertext*>& destination,
                                      MemoryPoolHandle pool) const {
    // evaluator_modswitch.cu:222-260 per member; a uniform batch steps down together through mod_switch_to_next_batched
    const char* P = "[Evaluator::mod_switch_to_batched]";
    same_size(P, encrypted.size(), destination.size());
    if (encrypted.empty()) return;
    auto target = context_->get_context_data(parms_id);
    if (!target.has_value()) throw std::invalid_argument("[Evaluator::mod_switch_to] ParmsID is not valid for the current context.");
    if (!uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; mod_switch_to(*encrypted[i], parms_id, out, pool); *destination[i] = std::move(out); }
        return;
    }
    auto cd = context_->get_context_data(encrypted[0]->parms_id());
    if (!cd.has_value()) throw std::invalid_argument("[Evaluator::mod_switch_to] ParmsID is not valid for the current context.");
    if (cd.value()->chain_index() < target.value()->chain_index()) throw std::invalid_argument("[Evaluator::mod_switch_to_inplace] Cannot switch to a higher level.");
    if (encrypted[0]->parms_id() == parms_id) {
        for (size_t i = 0; i < encrypted.size(); i++) if (destination[i] != encrypted[i]) *destination[i] = encrypted[i]->clone(pool);
        return;
    }
    mod_switch_to_next_batched(encrypted, destination, pool);
    while (destination[0]->parms_id() != parms_id) mod_switch_to_next_batched(as_const(destination), destination, pool);
}

// -- shifts and the 1/N factor of the packing tree ----------------------------------------------------------------------------------------
void Evaluator::negacyclic_shift_batched(const std::vector<const Ciphertext*>& encrypted, size_t shift, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    same_size("[Evaluator::negacyclic_shift_batched]", encrypted.size(), destination.size());
    if (encrypted.size() < BATCH_OP_THRESHOLD || !uniform(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) { Ciphertext out; negacyclic_shift(*encrypted[i], shift, out, pool); *destination[i] = std::move(out); }
        return;
    }
    Ciphertext proto;
    negacyclic_shift(*encrypted[0], shift, proto, pool);
    const size_t count = encrypted.size();
    const uint32_t L = static_cast<uint32_t>(proto.coeff_modulus_size());
    utils::DynamicArray staged(0, true, pool);
    const uint64_t* in = contiguous(encrypted, staged, pool);
    auto block = result_block(proto, count, pool);
    troyn_check_public(troyn_negacyclic_shift(context_->plan(), 0, L, in, block->raw_pointer(), shift, count * proto.polynomial_count(), stream()));
    assign_views(proto, block, destination);
}

void Evaluator::divide_by_poly_modulus_degree_inplace_batched(const std::vector<Ciphertext*>& encrypted, uint64_t mul, MemoryPoolHandle pool) const {
    (void)pool;
    for (Ciphertext* c : encrypted) divide_by_poly_modulus_degree_inplace(*c, mul);
}

}  // namespace troy
