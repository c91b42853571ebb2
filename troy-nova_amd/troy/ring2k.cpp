// ring2k.cpp -- see ring2k.h.  The arithmetic lives in csrc/ring2k_kernels.hpp behind troyn_ring2k_*.
#include "ring2k.h"

#include <hip/hip_runtime.h>

namespace troy { namespace linear {

template <typename T>
PolynomialEncoderRing2k<T>::PolynomialEncoderRing2k(HeContextPointer context, size_t t_bit_length) : context_(std::move(context)), t_bit_length_(t_bit_length) {
    // bfv_ring2k.cu:96-104, :880-900
    const char* P = "[PolynomialEncoderRNSHelper::PolynomialEncoderRNSHelper]";
    if (t_bit_length <= sizeof(T) * 4 || t_bit_length > sizeof(T) * 8) throw std::invalid_argument(std::string(P) + " t_bit_length must be greater than type_bits<T>() / 2");
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::invalid_argument(std::string(P) + " scheme must be BFV or BGV");
}

template <typename T>
void PolynomialEncoderRing2k<T>::release_helpers() {
    std::lock_guard<std::mutex> lock(mutex_);
    for (auto& kv : helpers_) troyn_ring2k_destroy(kv.second);
    helpers_.clear();
}

template <typename T>
PolynomialEncoderRing2k<T>::~PolynomialEncoderRing2k() { release_helpers(); }

template <typename T>
const troyn_ring2k* PolynomialEncoderRing2k<T>::helper(const ParmsID& parms_id) const {
    if (!context_->on_device()) throw std::invalid_argument("[PolynomialEncoderRing2k] HeContext is not on device (call to_device_inplace).");
    auto cd = context_->get_context_data(parms_id);
    if (!cd.has_value()) throw std::invalid_argument("[PolynomialEncoderRing2k:scale_up] No helper found for the given parms_id");
    const size_t L = cd.value()->parms().coeff_modulus().size();
    std::lock_guard<std::mutex> lock(mutex_);
    auto it = helpers_.find(L);
    if (it != helpers_.end()) return it->second;
    troyn_ring2k* h = nullptr;
    troyn_check_public(troyn_ring2k_create(&h, context_->plan(), static_cast<uint32_t>(L), static_cast<uint32_t>(t_bit_length_), static_cast<uint32_t>(sizeof(T))));
    helpers_[L] = h;
    return h;
}

template <typename T>
void PolynomialEncoderRing2k<T>::encode(const utils::ConstSlice<T>* source, size_t count, std::optional<ParmsID> parms_id, bool scale, Plaintext* const* destination, MemoryPoolHandle pool) const {
    const ParmsID pid = parms_id.value_or(context_->first_parms_id());
    const troyn_ring2k* h = helper(pid);
    const size_t n = slot_count();
    if (count == 0) return;
    // the elements travel as raw bytes in ONE word buffer (each source at an 8-byte aligned offset); device-resident sources are copied device to device
    std::vector<size_t> offset(count);
    size_t words = 0;
    for (size_t i = 0; i < count; i++) {
        if (source[i].size() > n) throw std::invalid_argument(scale ? "[PolynomialEncoderRNSHelper:scale_up] source size is larger than poly_modulus_degree"
                                                                    : "[PolynomialEncoderRNSHelper:centralize] source size is larger than poly_modulus_degree");
        offset[i] = words;
        words += (source[i].size() * sizeof(T) + 7) / 8 + 1;
    }
    utils::DynamicArray staged(words, true, pool);
    hipStream_t s = static_cast<hipStream_t>(troyn_current_stream());
    std::vector<uint64_t> raw(words, 0);
    bool any_host = false;
    for (size_t i = 0; i < count; i++) if (!source[i].on_device() && source[i].size()) { std::memcpy(raw.data() + offset[i], source[i].raw_pointer(), source[i].size() * sizeof(T)); any_host = true; }
    if (any_host) staged.copy_from(raw.data(), words, false);
    for (size_t i = 0; i < count; i++)
        if (source[i].on_device() && source[i].size())
            if (hipMemcpyAsync(staged.raw_pointer() + offset[i], source[i].raw_pointer(), source[i].size() * sizeof(T), hipMemcpyDeviceToDevice, s) != hipSuccess)
                throw std::runtime_error("[PolynomialEncoderRing2k] device copy of a source slice failed");
    // the result keeps only the source's coefficients (bfv_ring2k.cu:330-347, :540-556: resize_rns_partial(.., source.size())), data[l * count + i];
    // the kernels write full rows, so a shorter source goes through one scratch polynomial and a strided copy
    const size_t L = context_->get_context_data(pid).value()->parms().coeff_modulus().size();
    utils::DynamicArray scratch(0, true, pool);
    for (size_t i = 0; i < count; i++) {
        const size_t cc = source[i].size();
        Plaintext out;
        out.data() = utils::DynamicArray(0, true, pool);
        out.resize_rns_partial(*context_, pid, cc);
        if (cc > 0) {
            uint64_t* target = out.poly();
            if (cc < n) { if (scratch.size() == 0) scratch = utils::DynamicArray(L * n, true, pool); target = scratch.raw_pointer(); }
            if (scale) troyn_check_public(troyn_ring2k_scale_up(h, staged.raw_pointer() + offset[i], cc, target, s));
            else troyn_check_public(troyn_ring2k_centralize(h, staged.raw_pointer() + offset[i], cc, target, s));
            if (cc < n && hipMemcpy2DAsync(out.poly(), cc * sizeof(uint64_t), target, n * sizeof(uint64_t), cc * sizeof(uint64_t), L, hipMemcpyDeviceToDevice, s) != hipSuccess)
                throw std::runtime_error("[PolynomialEncoderRing2k] strided copy of a partial plaintext failed");
        }
        out.is_ntt_form() = false;
        *destination[i] = std::move(out);
    }
    troyn_sync_current_stream();
}

template <typename T>
void PolynomialEncoderRing2k<T>::decode(const Plaintext& input, bool scale, T correction_factor, utils::Slice<T> destination, MemoryPoolHandle pool) const {
    const char* P = "[PolynomialEncoderRNSHelper::scale_down]";       // the reference reports both decoders under this name
    if (input.parms_id() == parms_id_zero) throw std::invalid_argument(std::string(P) + " input is not valid");
    if (!input.on_device()) throw std::invalid_argument(std::string(P) + " self, input, destination are not in the same device");
    if (input.is_ntt_form()) throw std::invalid_argument(std::string(P) + " input is in NTT form");
    const troyn_ring2k* h = helper(input.parms_id());
    const size_t n = slot_count(), cc = input.coeff_count(), L = input.coeff_modulus_size();
    if (input.data().size() != L * cc) throw std::invalid_argument(std::string(P) + " input does not have the shape of an RNS plaintext");
    if (destination.size() != cc) throw std::invalid_argument(std::string(P) + " destination size must be the input's coeff_count");
    if (cc == 0) return;
    hipStream_t s = static_cast<hipStream_t>(troyn_current_stream());
    // Both decoders work coefficient by coefficient on data[l * coeff_count + i] (bfv_ring2k.cu:700-712, :872-911: coeff_count = destination.size(), which may be
    // SHORTER than the ring degree -- a partial plaintext -- or LONGER: the reference's test decodes 54 coefficients of an N = 32 context in one call).  The kernels
    // take full rows [L][N]: the input is walked in chunks of N coefficients, each widened (zero-padded) to full rows when it is not one already.
    const unsigned __int128 cf = static_cast<unsigned __int128>(correction_factor);
    const size_t chunk_words = (n * sizeof(T) + 7) / 8;
    utils::DynamicArray rows(cc == n ? 0 : L * n, true, pool), out(destination.on_device() && cc == n ? 0 : chunk_words, true, pool);
    std::vector<uint64_t> raw;
    for (size_t base = 0; base < cc; base += n) {
        const size_t len = std::min(n, cc - base);
        const uint64_t* in = input.poly();
        if (cc != n) {
            if (len < n && hipMemsetAsync(rows.raw_pointer(), 0, L * n * sizeof(uint64_t), s) != hipSuccess) throw std::runtime_error("[PolynomialEncoderRing2k] memset failed");
            if (hipMemcpy2DAsync(rows.raw_pointer(), n * sizeof(uint64_t), input.poly().raw_pointer() + base, cc * sizeof(uint64_t), len * sizeof(uint64_t), L, hipMemcpyDeviceToDevice, s) != hipSuccess)
                throw std::runtime_error("[PolynomialEncoderRing2k] strided copy of a plaintext chunk failed");
            in = rows.raw_pointer();
        }
        const bool direct = destination.on_device() && cc == n;
        void* dst = direct ? static_cast<void*>(destination.raw_pointer()) : static_cast<void*>(out.raw_pointer());
        if (scale) troyn_check_public(troyn_ring2k_scale_down(h, in, dst, s));
        else troyn_check_public(troyn_ring2k_decentralize(h, in, dst, static_cast<uint64_t>(cf), static_cast<uint64_t>(cf >> 64), s));
        if (direct) continue;
        if (destination.on_device()) {
            if (hipMemcpyAsync(destination.raw_pointer() + base, dst, len * sizeof(T), hipMemcpyDeviceToDevice, s) != hipSuccess)
                throw std::runtime_error("[PolynomialEncoderRing2k] device copy of the decoded elements failed");
        } else {
            troyn_sync_current_stream();
            raw = out.to_vector();
            std::memcpy(destination.raw_pointer() + base, raw.data(), len * sizeof(T));
        }
    }
    troyn_sync_current_stream();
}

template class PolynomialEncoderRing2k<uint32_t>;
template class PolynomialEncoderRing2k<uint64_t>;
template class PolynomialEncoderRing2k<unsigned __int128>;

}}  // namespace troy::linear
