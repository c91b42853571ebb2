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
PolynomialEncoderRing2k<T>::~PolynomialEncoderRing2k() {
    for (auto& kv : helpers_) troyn_ring2k_destroy(kv.second);
}

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
void PolynomialEncoderRing2k<T>::encode(const std::vector<T>& source, std::optional<ParmsID> parms_id, bool scale, Plaintext& destination, MemoryPoolHandle pool) const {
    const ParmsID pid = parms_id.value_or(context_->first_parms_id());
    const troyn_ring2k* h = helper(pid);
    const size_t n = slot_count();
    if (source.size() > n) throw std::invalid_argument("[PolynomialEncoderRNSHelper:scale_up] source size is larger than poly_modulus_degree");
    // the elements travel as raw bytes in a word buffer
    const size_t words = (source.size() * sizeof(T) + 7) / 8 + 1;
    std::vector<uint64_t> raw(words, 0);
    std::memcpy(raw.data(), source.data(), source.size() * sizeof(T));
    utils::DynamicArray staged(words, true, pool);
    staged.copy_from(raw.data(), words, false);
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, pid);
    hipStream_t s = static_cast<hipStream_t>(troyn_current_stream());
    if (scale) troyn_check_public(troyn_ring2k_scale_up(h, staged.raw_pointer(), source.size(), out.poly(), s));
    else troyn_check_public(troyn_ring2k_centralize(h, staged.raw_pointer(), source.size(), out.poly(), s));
    troyn_sync_current_stream();
    out.is_ntt_form() = false;
    destination = std::move(out);
}

template <typename T>
std::vector<T> PolynomialEncoderRing2k<T>::scale_down_new(const Plaintext& input, MemoryPoolHandle pool) const {
    const char* P = "[PolynomialEncoderRNSHelper::scale_down]";
    if (input.parms_id() == parms_id_zero) throw std::invalid_argument(std::string(P) + " input is not valid");
    if (!input.on_device()) throw std::invalid_argument(std::string(P) + " self, input, destination are not in the same device");
    if (input.is_ntt_form()) throw std::invalid_argument(std::string(P) + " input is in NTT form");
    const troyn_ring2k* h = helper(input.parms_id());
    const size_t n = slot_count();
    if (input.coeff_count() != n) throw std::invalid_argument(std::string(P) + " input is not a full RNS polynomial");
    const size_t words = (n * sizeof(T) + 7) / 8;
    utils::DynamicArray out(words, true, pool);
    troyn_check_public(troyn_ring2k_scale_down(h, input.poly(), out.raw_pointer(), troyn_current_stream()));
    troyn_sync_current_stream();
    const std::vector<uint64_t> raw = out.to_vector();
    std::vector<T> result(n);
    std::memcpy(result.data(), raw.data(), n * sizeof(T));
    return result;
}

template class PolynomialEncoderRing2k<uint32_t>;
template class PolynomialEncoderRing2k<uint64_t>;
template class PolynomialEncoderRing2k<unsigned __int128>;

}}  // namespace troy::linear
