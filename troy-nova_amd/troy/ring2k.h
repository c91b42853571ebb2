// ring2k.h -- PolynomialEncoderRing2k<T> of the reference (src/app/bfv_ring2k.h:78-330): polynomials over Z_{2^k}, k up to 128, encoded for BFV
// outside the context's own plain modulus.  T is uint32_t / uint64_t / unsigned __int128 and k > bits(T) / 2.  The arithmetic lives behind the
// C-ABI (troyn_ring2k_*, csrc/ring2k_kernels.hpp); this class keeps the reference's method surface: vector, slice (host or device memory) and
// batched forms of scale_up / centralize, scale_down and decentralize.
#pragma once
#include <cstring>
#include <map>
#include <mutex>

#include "troy.h"

namespace troy { namespace linear {

template <typename T>
class PolynomialEncoderRing2k {
    static_assert(std::is_same<T, uint32_t>::value || std::is_same<T, uint64_t>::value || std::is_same<T, unsigned __int128>::value,
                  "T must be uint32_t, uint64_t or uint128_t");
public:
    PolynomialEncoderRing2k(HeContextPointer context, size_t t_bit_length);
    ~PolynomialEncoderRing2k();
    PolynomialEncoderRing2k(const PolynomialEncoderRing2k&) = delete;
    PolynomialEncoderRing2k& operator=(const PolynomialEncoderRing2k&) = delete;
    // movable, as the reference's (test_adv.h keeps the encoders in std::optional): the per-level helpers travel, the mutex is the new object's own
    PolynomialEncoderRing2k(PolynomialEncoderRing2k&& other) noexcept : context_(std::move(other.context_)), t_bit_length_(other.t_bit_length_) {
        std::lock_guard<std::mutex> lock(other.mutex_);
        helpers_ = std::move(other.helpers_);
        other.helpers_.clear();
    }
    PolynomialEncoderRing2k& operator=(PolynomialEncoderRing2k&& other) noexcept {
        if (this != &other) {
            release_helpers();
            std::lock_guard<std::mutex> lock(other.mutex_);
            context_ = std::move(other.context_); t_bit_length_ = other.t_bit_length_; helpers_ = std::move(other.helpers_); other.helpers_.clear();
        }
        return *this;
    }

    HeContextPointer context() const noexcept { return context_; }
    size_t t_bit_length() const noexcept { return t_bit_length_; }
    T t_mask() const noexcept { return t_bit_length_ == sizeof(T) * 8 ? static_cast<T>(-1) : static_cast<T>((static_cast<T>(1) << t_bit_length_) - 1); }
    bool on_device() const noexcept { return context_->on_device(); }
    size_t slot_count() const { return context_->first_context_data().value()->parms().poly_modulus_degree(); }
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { (void)pool; }   // constants are built on the device on first use

    // ---- encode: m -> round(Q / 2^k * m) (scale_up, for encryption) or the centred lift of m (centralize, for ciphertext x plaintext) ----
    void scale_up_slice(utils::ConstSlice<T> source, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        const utils::ConstSlice<T>* one = &source; Plaintext* d = &destination; encode(one, 1, parms_id, true, &d, pool);
    }
    void scale_up_slice_batched(const utils::ConstSliceVec<T>& source, std::optional<ParmsID> parms_id, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (source.size() != destination.size()) throw std::invalid_argument("[PolynomialEncoderRNSHelper::scale_up_batched] source and destination must have the same size");
        encode(source.data(), source.size(), parms_id, true, destination.data(), pool);
    }
    void scale_up(const std::vector<T>& source, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        scale_up_slice(utils::ConstSlice<T>(source.data(), source.size(), false), parms_id, destination, pool);
    }
    Plaintext scale_up_slice_new(utils::ConstSlice<T> source, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; scale_up_slice(source, parms_id, p, pool); return p; }
    Plaintext scale_up_new(const std::vector<T>& source, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; scale_up(source, parms_id, p, pool); return p; }

    void centralize_slice(utils::ConstSlice<T> source, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        const utils::ConstSlice<T>* one = &source; Plaintext* d = &destination; encode(one, 1, parms_id, false, &d, pool);
    }
    void centralize_slice_batched(const utils::ConstSliceVec<T>& source, std::optional<ParmsID> parms_id, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (source.size() != destination.size()) throw std::invalid_argument("[PolynomialEncoderRNSHelper::centralize_batched] source and destination must have the same size");
        encode(source.data(), source.size(), parms_id, false, destination.data(), pool);
    }
    void centralize(const std::vector<T>& source, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        centralize_slice(utils::ConstSlice<T>(source.data(), source.size(), false), parms_id, destination, pool);
    }
    Plaintext centralize_slice_new(utils::ConstSlice<T> source, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; centralize_slice(source, parms_id, p, pool); return p; }
    Plaintext centralize_new(const std::vector<T>& source, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; centralize(source, parms_id, p, pool); return p; }

    // ---- decode: scale_down takes the phase c(s) of a ciphertext in coefficient form (Decryptor::bfv_decrypt_without_scaling_down); decentralize takes
    //      x mod Q (the inverse of centralize) and optionally divides by an odd correction factor ----
    void scale_down_slice(const Plaintext& input, utils::Slice<T> destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { decode(input, true, 1, destination, pool); }
    void scale_down(const Plaintext& input, std::vector<T>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        destination.resize(input.coeff_count());
        scale_down_slice(input, utils::Slice<T>(destination.data(), destination.size(), false), pool);
    }
    utils::Array<T> scale_down_slice_new(const Plaintext& input, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { utils::Array<T> a = utils::Array<T>::create_uninitialized(input.coeff_count(), on_device(), pool); scale_down_slice(input, a.reference(), pool); return a; }
    std::vector<T> scale_down_new(const Plaintext& input, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<T> v; scale_down(input, v, pool); return v; }

    void decentralize_slice(const Plaintext& input, utils::Slice<T> destination, T correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { decode(input, false, correction_factor, destination, pool); }
    void decentralize(const Plaintext& input, std::vector<T>& destination, T correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        destination.resize(input.coeff_count());
        decentralize_slice(input, utils::Slice<T>(destination.data(), destination.size(), false), correction_factor, pool);
    }
    utils::Array<T> decentralize_slice_new(const Plaintext& input, T correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { utils::Array<T> a = utils::Array<T>::create_uninitialized(input.coeff_count(), on_device(), pool); decentralize_slice(input, a.reference(), correction_factor, pool); return a; }
    std::vector<T> decentralize_new(const Plaintext& input, T correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<T> v; decentralize(input, v, correction_factor, pool); return v; }

private:
    // `count` sources (host or device memory each) -> `count` plaintexts: one staging buffer, `count` launches on the thread's stream, one wait
    void encode(const utils::ConstSlice<T>* source, size_t count, std::optional<ParmsID> parms_id, bool scale, Plaintext* const* destination, MemoryPoolHandle pool) const;
    void decode(const Plaintext& input, bool scale, T correction_factor, utils::Slice<T> destination, MemoryPoolHandle pool) const;
    const troyn_ring2k* helper(const ParmsID& parms_id) const;   // PolynomialEncoderRNSHelper of that level
    void release_helpers();
    HeContextPointer context_;
    size_t t_bit_length_;
    mutable std::mutex mutex_;
    mutable std::map<size_t, troyn_ring2k*> helpers_;            // by modulus count of the level
};

extern template class PolynomialEncoderRing2k<uint32_t>;
extern template class PolynomialEncoderRing2k<uint64_t>;
extern template class PolynomialEncoderRing2k<unsigned __int128>;

}}  // namespace troy::linear
