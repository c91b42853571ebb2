// ring2k.h -- troy::linear::PolynomialEncoderRing2k<T>, the Z_{2^k} polynomial encoder of the reference's ring-2^k application
// (src/app/bfv_ring2k.{h,cu}; examples/13_ring2k.cu): elements of Z_{2^k}, k up to 128, carried by a BFV context whose own plain
// modulus is not used.  T = uint32_t, uint64_t or unsigned __int128; bits(T)/2 < k <= bits(T).
//   scale_up     for the operand that is encrypted:            round(Q/2^k * m) mod q_l
//   centralize   for the operand that multiplies a ciphertext: the centred lift of m mod q_l
//   scale_down   on Decryptor::bfv_decrypt_without_scaling_down's output: m mod 2^k
// Plaintexts are full-size RNS polynomials of the chosen level (coefficients beyond the input are zero).  GPU only.
#pragma once
#include <map>

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

    HeContextPointer context() const noexcept { return context_; }
    size_t t_bit_length() const noexcept { return t_bit_length_; }
    T t_mask() const noexcept { return t_bit_length_ == sizeof(T) * 8 ? static_cast<T>(-1) : static_cast<T>((static_cast<T>(1) << t_bit_length_) - 1); }
    bool on_device() const noexcept { return context_->on_device(); }
    size_t slot_count() const { return context_->first_context_data().value()->parms().poly_modulus_degree(); }
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { (void)pool; }   // constants are built on the device on first use

    void scale_up(const std::vector<T>& source, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        encode(source, parms_id, true, destination, pool);
    }
    Plaintext scale_up_new(const std::vector<T>& source, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; scale_up(source, parms_id, p, pool); return p;
    }
    void centralize(const std::vector<T>& source, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        encode(source, parms_id, false, destination, pool);
    }
    Plaintext centralize_new(const std::vector<T>& source, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; centralize(source, parms_id, p, pool); return p;
    }
    // input: the phase c(s) of a ciphertext in coefficient form (Decryptor::bfv_decrypt_without_scaling_down)
    std::vector<T> scale_down_new(const Plaintext& input, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

private:
    void encode(const std::vector<T>& source, std::optional<ParmsID> parms_id, bool scale, Plaintext& destination, MemoryPoolHandle pool) const;
    const troyn_ring2k* helper(const ParmsID& parms_id) const;   // PolynomialEncoderRNSHelper of that level
    HeContextPointer context_;
    size_t t_bit_length_;
    mutable std::mutex mutex_;
    mutable std::map<size_t, troyn_ring2k*> helpers_;            // by modulus count of the level
};

extern template class PolynomialEncoderRing2k<uint32_t>;
extern template class PolynomialEncoderRing2k<uint64_t>;
extern template class PolynomialEncoderRing2k<unsigned __int128>;

}}  // namespace troy::linear
