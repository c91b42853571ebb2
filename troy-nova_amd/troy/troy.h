// troy.h -- host-side C++ mirror of troy-nova's public API for the hot path, on top of the
// C-ABI in include/troyn.h (libtroyn.so).  Same namespace, class and method names, argument
// meaning and error behaviour as the reference (src/troy.h and the headers it includes), so code
// written against `#include "troy/troy.h"` for the operations below compiles unchanged:
//
//   EncryptionParameters / CoeffModulus / PlainModulus   src/encryption_parameters.h, coeff_modulus.h
//   HeContext::create / to_device_inplace, ContextData     src/he_context.h, context_data.h
//   MemoryPool / MemoryPoolHandle                           src/utils/memory_pool.h
//   Ciphertext / Plaintext / KSwitchKeys / RelinKeys        src/ciphertext.h, plaintext.h, kswitch_keys.h
//   utils::RandomGenerator, SecretKey, KeyGenerator, Encryptor, Decryptor, BatchEncoder
//                                                          src/utils/random_generator.h, key_generator.h, encryptor.h,
//                                                          decryptor.h, batch_encoder.h   (BFV and CKKS; GPU only)
//   Evaluator (negate, add, sub, multiply, square, relinearize, apply_keyswitching,
//              mod_switch_to_next, mod_switch_to, rescale_to_next, transform_to/from_ntt,
//              x / x_inplace / x_new / x_batched)           src/evaluator.h:113-700
//
// Differences that are deliberate:
//   * the Evaluator runs on the GPU only.  Operands must be on the device; a host-resident operand
//     raises std::invalid_argument instead of silently taking a CPU path (there is none).
//   * multiply / relinearize / rescale also exist as *_batched (the reference lists them as
//     "not implemented yet", test/bench/he_operations.cpp:119-135).
#pragma once
#include <algorithm>
#include <atomic>
#include <thread>
#include <complex>
#include <cstdint>
#include <cstring>
#include <iosfwd>
#include <ostream>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/troyn.h"
#include "bench_timer.h"

namespace troy {

enum class SchemeType : uint8_t { Nil = 0, BFV = 1, CKKS = 2, BGV = 3 };
// utils/compression.h:15-18.  Zstd works when libzstd.so.1 is present at run time (the reference compiles zstd in from a submodule; see troy.cpp).
enum class CompressionMode : uint8_t { Nil = 0, Zstd = 1 };
enum class SecurityLevel : uint8_t { Nil = 0, Classical128 = 1, Classical192 = 2, Classical256 = 3 };
namespace utils { namespace compression {
bool available(CompressionMode mode);       // utils/compression.h:46-52: Nil always; Zstd when the zstd runtime library (libzstd.so.1) can be loaded (troy.cpp)
}}  // namespace utils::compression

// utils/box.h: the non-owning (pointer, length) views and the owning array the reference's public API hands out for HOST-side
// parameter data (moduli lists, ...).  Device payloads of this mirror are DynamicArray / raw device pointers; these types cover what
// user programs touch: size(), operator[], iteration, to_vector(), Array::copy_from_slice.
namespace utils {
class MemoryPool;
// bytes between a slice (host or device) and host memory: what to_vector() and the encoders' *_slice forms go through (troy.cpp)
void slice_bytes_to_host(const void* src, bool src_on_device, void* dst, size_t bytes);
void host_bytes_to_slice(void* dst, bool dst_on_device, const void* src, size_t bytes);
template <typename T>
class ConstSlice {
public:
    ConstSlice(const T* ptr, size_t len, bool on_device = false, std::nullptr_t = nullptr) : ptr_(ptr), len_(len), device_(on_device) {}
    ConstSlice(const std::vector<T>& v) : ptr_(v.data()), len_(v.size()), device_(false) {}      // NOLINT: implicit by design
    size_t size() const noexcept { return len_; }
    bool empty() const noexcept { return len_ == 0; }
    bool on_device() const noexcept { return device_; }
    const T* raw_pointer() const noexcept { return ptr_; }
    const T& operator[](size_t i) const { return ptr_[i]; }
    const T* begin() const noexcept { return ptr_; }
    const T* end() const noexcept { return ptr_ + len_; }
    ConstSlice const_slice(size_t begin, size_t end) const { return ConstSlice(ptr_ + begin, end - begin, device_); }
    std::vector<T> to_vector() const {                                                            // box.h: a host copy, wherever the view points
        if (!device_) return std::vector<T>(ptr_, ptr_ + len_);
        std::vector<T> v(len_);
        if (len_) slice_bytes_to_host(ptr_, true, v.data(), len_ * sizeof(T));
        return v;
    }
    operator std::vector<T>() const { return to_vector(); }                                       // NOLINT
    operator const T*() const noexcept { return ptr_; }                                           // NOLINT: the C-ABI takes raw device pointers
private:
    const T* ptr_;
    size_t len_;
    bool device_;
};

// utils/box.h:262-308: the mutable view (Ciphertext::poly, Plaintext::poly, ...).  Converts to ConstSlice and to a raw pointer.
template <typename T>
class Slice {
public:
    Slice(T* ptr, size_t len, bool on_device = false, std::nullptr_t = nullptr) : ptr_(ptr), len_(len), device_(on_device) {}
    size_t size() const noexcept { return len_; }
    bool on_device() const noexcept { return device_; }
    T* raw_pointer() const noexcept { return ptr_; }
    T& operator[](size_t i) const { return ptr_[i]; }
    ConstSlice<T> as_const() const { return ConstSlice<T>(ptr_, len_, device_); }
    std::vector<T> to_vector() const { return as_const().to_vector(); }
    operator ConstSlice<T>() const { return as_const(); }                                         // NOLINT
    operator T*() const noexcept { return ptr_; }                                                 // NOLINT
    ConstSlice<T> const_slice(size_t begin, size_t end) const { return ConstSlice<T>(ptr_ + begin, end - begin, device_); }
    Slice slice(size_t begin, size_t end) const { return Slice(ptr_ + begin, end - begin, device_); }
    void set_zero() const;                                                                        // box.h:298-308 (host memset / device memset)
    void copy_from_slice(ConstSlice<T> source) const;                                             // box.h:282-297 (any host / device combination)
private:
    T* ptr_;
    size_t len_;
    bool device_;
};

// raw device memory for Array<T> (any element type; the uint64_t payloads of ciphertexts / plaintexts are DynamicArray): troy.cpp
std::shared_ptr<void> device_bytes_allocate(size_t bytes, std::shared_ptr<MemoryPool> pool, bool zero);     // freed (returned to its pool) with the last owner
void device_bytes_copy(void* dst, const void* src, size_t bytes);                                             // device to device, on the calling thread's stream

// utils/box.h:300-560: the owning array.  Host-resident (a std::vector) or device-resident (pool memory); the views it hands out carry the location, and
// to_vector() / copy_from_slice() move bytes across whatever the combination.  operator[] and iteration are for host-resident arrays, as in the reference.
template <typename T>
class Array {
public:
    Array() = default;
    Array(std::vector<T> values) : host_(std::move(values)), count_(host_.size()) {}             // NOLINT: implicit by design
    Array(size_t count, bool on_device, std::shared_ptr<MemoryPool> pool = nullptr) : count_(count), device_(on_device) { allocate(pool, true); }
    static Array create_uninitialized(size_t count, bool on_device, std::shared_ptr<MemoryPool> pool = nullptr) {   // box.h:342-350
        Array a; a.count_ = count; a.device_ = on_device; a.allocate(pool, false); return a;
    }
    static Array from_vector(std::vector<T> values) { return Array(std::move(values)); }
    static Array create_and_copy_from_slice(ConstSlice<T> source, bool on_device = false, std::shared_ptr<MemoryPool> pool = nullptr) {
        Array a = create_uninitialized(source.size(), on_device, pool); a.copy_from_slice(source); return a;
    }
    operator std::vector<T>() const { return to_vector(); }                                      // NOLINT
    typename std::vector<T>::const_iterator begin() const noexcept { return host_.begin(); }
    typename std::vector<T>::const_iterator end() const noexcept { return host_.end(); }
    size_t size() const noexcept { return count_; }
    bool on_device() const noexcept { return device_; }
    T& operator[](size_t i) { return host_[i]; }
    const T& operator[](size_t i) const { return host_[i]; }
    T* raw_pointer() { return device_ ? static_cast<T*>(dev_.get()) : host_.data(); }
    const T* raw_pointer() const { return device_ ? static_cast<const T*>(dev_.get()) : host_.data(); }
    ConstSlice<T> const_reference() const { return ConstSlice<T>(raw_pointer(), count_, device_); }
    Slice<T> reference() { return Slice<T>(raw_pointer(), count_, device_); }
    ConstSlice<T> const_slice(size_t begin, size_t end) const { return ConstSlice<T>(raw_pointer() + begin, end - begin, device_); }
    Slice<T> slice(size_t begin, size_t end) { return Slice<T>(raw_pointer() + begin, end - begin, device_); }
    std::vector<T> to_vector() const { return const_reference().to_vector(); }
    void copy_from_slice(ConstSlice<T> source) {
        if (source.size() != count_) throw std::invalid_argument("[Array::copy_from_slice] Slice size does not match the array size.");
        if (!count_) return;
        if (device_ && source.on_device()) device_bytes_copy(raw_pointer(), source.raw_pointer(), count_ * sizeof(T));
        else if (device_) host_bytes_to_slice(raw_pointer(), true, source.raw_pointer(), count_ * sizeof(T));
        else slice_bytes_to_host(source.raw_pointer(), source.on_device(), host_.data(), count_ * sizeof(T));
    }
    Array clone(std::shared_ptr<MemoryPool> pool = nullptr) const { Array a = create_uninitialized(count_, device_, pool ? pool : pool_); a.copy_from_slice(const_reference()); return a; }
    Array to_host() const { Array a = create_uninitialized(count_, false); a.copy_from_slice(const_reference()); return a; }
    Array to_device(std::shared_ptr<MemoryPool> pool = nullptr) const { Array a = create_uninitialized(count_, true, pool); a.copy_from_slice(const_reference()); return a; }
    void to_host_inplace() { if (device_) *this = to_host(); }
    void to_device_inplace(std::shared_ptr<MemoryPool> pool = nullptr) { if (!device_) *this = to_device(pool); }
    void set_zero() { if (device_) { Array z(count_, true, pool_); *this = std::move(z); } else std::fill(host_.begin(), host_.end(), T()); }
private:
    void allocate(std::shared_ptr<MemoryPool> pool, bool zero) {
        if (!device_) { host_.assign(count_, T()); return; }
        pool_ = std::move(pool);
        dev_ = device_bytes_allocate(count_ * sizeof(T), pool_, zero);
    }
    std::vector<T> host_;
    std::shared_ptr<void> dev_;
    std::shared_ptr<MemoryPool> pool_;
    size_t count_ = 0;
    bool device_ = false;
};
template <typename T> std::vector<T> slice_to_vector(ConstSlice<T> s) {
    std::vector<T> v(s.size());
    if (s.size()) slice_bytes_to_host(s.raw_pointer(), s.on_device(), v.data(), s.size() * sizeof(T));
    return v;
}
template <typename T> void vector_to_slice(const std::vector<T>& v, Slice<T> d, const char* prompt) {
    if (d.size() < v.size()) throw std::invalid_argument(std::string(prompt) + " destination is too small.");
    if (!v.empty()) host_bytes_to_slice(d.raw_pointer(), d.on_device(), v.data(), v.size() * sizeof(T));
}
// the *_slice_new decoders return their array where the plaintext lives, as the reference's do (batch_encoder.h:103-135, ckks_encoder.h)
template <typename T> Array<T> located(Array<T> a, bool on_device, std::shared_ptr<MemoryPool> pool) { if (on_device) a.to_device_inplace(std::move(pool)); return a; }
template <typename T> using ConstSliceVec = std::vector<ConstSlice<T>>;
template <typename T> using SliceVec = std::vector<Slice<T>>;
// utils/box.h:562-590: "[a, b, c]" for host views, "device[a, b, c]" for device views (copied to the host to be printed)
template <typename T> std::ostream& operator<<(std::ostream& os, const ConstSlice<T>& view) {
    const std::vector<T> v = view.to_vector();
    os << (view.on_device() ? "device[" : "[");
    for (size_t i = 0; i < v.size(); i++) { if (i) os << ", "; os << v[i]; }
    return os << "]";
}
template <typename T> std::ostream& operator<<(std::ostream& os, const Slice<T>& view) { return os << view.as_const(); }
template <typename T> std::ostream& operator<<(std::ostream& os, const Array<T>& array) { return os << array.const_reference(); }
}  // namespace utils

// ----------------------------------------------------------------------------------------------
// utils: device runtime shim + memory pool (src/kernel_provider.h, src/utils/memory_pool.h)
// ----------------------------------------------------------------------------------------------
namespace utils {

size_t device_count();

class MemoryPool;
using MemoryPoolHandle = std::shared_ptr<MemoryPool>;

// Pooled device allocator: freed blocks are kept in per-size free lists and reused (the reference
// uses a best-fit multimap, memory_pool_safe.in:119-148).  Thread safe.  Every host thread works on ONE stream per device for its
// whole life -- one of a bounded set of streams per device that the threads share (default 16, 8 once more than 16 host threads use the library; TROY_STREAMS=<1..16>, or
// TROY_STREAMS=per-thread for hipStreamPerThread as in rounds 1-5) -- and a block may be released while kernels that use it are
// still queued: a block carries the tag of the STREAM it was released on, and a thread that takes a block of its own stream
// is safe by stream order; blocks of host threads that have ended (per-thread mode) are handed out after one
// device-wide synchronisation (which makes every cached block anybody's); otherwise the pool allocates fresh memory, and
// only when the device is out of memory does it hand out a block last released by another LIVE thread, again after a
// device-wide synchronisation (memory_pool_safe.in:133-143 does that with cudaDeviceSynchronize on every foreign reuse).
// ASSUMPTION the asynchronous methods rest on: a host thread always launches on the same stream -- true here because every launch of this
// library's C++ layer goes to troyn_current_stream(); a caller that drives the C-ABI (include/troyn.h) with streams of its own must order the
// release of a workspace after the work that uses it (stream-ordered free or an event), exactly as with any asynchronous HIP API.
// While call combining is on (below) there is ONE stream for all host threads, and all of them count as the same owner.
class MemoryPool {
public:
    explicit MemoryPool(size_t device = 0);
    ~MemoryPool();
    static MemoryPoolHandle create(size_t device = 0) { return std::make_shared<MemoryPool>(device); }
    static MemoryPoolHandle GlobalPool();
    static void Destroy();   // releases the global pool; call before exit (reference readme.md:129)
    size_t get_device() const { return device_; }
    // the rest of the reference's public surface (utils/memory_pool.h:91-140)
    static int implementation_type() { return 2; }                 // the thread-safe pool (memory_pool_safe.in:11)
    void set_device();                                             // makes the pool's device current
    void deny(bool set = true) { denying_.store(set, std::memory_order_relaxed); }      // debugging aid: every further allocation throws (memory_pool_safe.in:120-122)
    void destroy();                                                // frees everything the pool holds, cached or handed out (memory_pool_safe.in:168-205)
    void force_set_thread_id(std::thread::id) {}                   // (the reference re-tags its cached blocks; here blocks are tagged by stream, nothing to re-tag)
    static void* Allocate(size_t bytes) { return GlobalPool()->allocate(bytes); }
    static void Free(void* ptr) { GlobalPool()->release(ptr); }
    static void ReleaseUnused() { GlobalPool()->release_unused(); }
    void* allocate(size_t bytes);
    void release(void* ptr);
    void release_unused();
    static uint64_t device_allocations();   // hipMalloc calls of all pools so far (a steady-state loop should not add any)
    // internal: the blocks that thread released (~0: every block) at or before release number `upto` have no work pending any more
    void disown(uint64_t thread_tag, uint64_t upto = ~uint64_t(0));
    static void disown_all_pools();     // internal (call combining switched at a quiescent point): every cached block of every pool is anybody's
    // Fresh device memory is preferred over a block another live thread released (no device-wide wait) only while the pool holds less than
    // this many bytes (live + cached); above it the pool synchronises and reuses, as it does when the device is out of memory.
    // 0 = no cap.  Default: environment TROY_POOL_HIGH_WATER_MB read when the pool is created, else no cap.
    void set_high_water_bytes(size_t bytes);
    size_t held_bytes();                // live + cached bytes of this pool
private:
    // The cached blocks live in one shard per OWNER tag (the stream they were released on; shard 0: nobody has work pending on them), each with its own lock,
    // and the live blocks in a table sharded by address: threads on different streams never meet on a lock in the steady state, and a thread looks only at
    // blocks it may take (round 5: one mutex and one free list for everybody -- every allocation walked past the other streams' blocks).  Measured on one GPU:
    // no change at 16 / 64 host threads (what binds there is the dispatch path, profiles/r06_streams_ab.txt).  The slow path (fresh memory, device-wide waits) is serialised.
    struct FreeBlock { void* ptr; uint64_t owner; uint64_t seq; };   // owner = tag it was released under, seq = number of that release
    struct Shard { std::mutex m; std::map<size_t, std::vector<FreeBlock>> free_; };
    struct LiveShard { std::mutex m; std::unordered_map<void*, size_t> map; };
    static constexpr size_t LIVE_SHARDS = 64, STREAM_SHARDS = 256;
    Shard* shard_of(uint64_t tag);                       // created on first use; never removed while the pool lives (except shards of ended threads, under the table lock)
    void* take_from(Shard& s, size_t bytes);             // best fit with at most 2x slack, or null
    bool dead_owner_has(size_t bytes);                   // some ended thread's shard holds a fitting block
    LiveShard& live_shard(const void* p) { return live_[(reinterpret_cast<uintptr_t>(p) >> 8) % LIVE_SHARDS]; }
    size_t device_;
    uint64_t id_;                                        // unique per pool object (thread-local shard caches are keyed by it)
    Shard nobody_;                                       // owner 0
    std::atomic<Shard*> stream_shards_[STREAM_SHARDS];   // owner tags of the per-device stream set (troy.cpp this_thread_tag)
    std::mutex table_mutex_;                             // the shard table of every other tag (per-thread tags, the combining tag)
    std::unordered_map<uint64_t, std::unique_ptr<Shard>> other_shards_;
    LiveShard live_[LIVE_SHARDS];
    std::mutex slow_mutex_;                              // one thread at a time on the slow path
    std::atomic<uint64_t> release_seq_{0};
    std::atomic<size_t> held_bytes_{0}, high_water_{0};
    std::atomic<bool> denying_{false};
};

// Owning array of uint64_t on the host (malloc) or on a device (pool) -- src/utils/dynamic_array.h
class DynamicArray {
public:
    DynamicArray() = default;
    DynamicArray(size_t count, bool device, MemoryPoolHandle pool = nullptr);
    DynamicArray(const DynamicArray& other);
    DynamicArray(DynamicArray&& other) noexcept;
    DynamicArray& operator=(const DynamicArray& other);
    DynamicArray& operator=(DynamicArray&& other) noexcept;
    ~DynamicArray();

    static DynamicArray from_vector(const std::vector<uint64_t>& v);
    // a device array that is a window of a larger allocation kept alive by `owner` (objects produced by one batched
    // launch share their buffer; copies of the object are ordinary owning arrays)
    static DynamicArray device_view(uint64_t* ptr, size_t count, std::shared_ptr<DynamicArray> owner);
    const DynamicArray* view_owner() const noexcept { return owner_.get(); }   // the shared buffer a view points into, or null
    std::vector<uint64_t> to_vector() const;
    DynamicArray clone(MemoryPoolHandle pool = nullptr) const;

    size_t size() const noexcept { return size_; }
    bool on_device() const noexcept { return device_; }
    MemoryPoolHandle pool() const { return pool_; }
    size_t device_index() const { return pool_ ? pool_->get_device() : 0; }
    uint64_t* raw_pointer() noexcept { return data_; }
    const uint64_t* raw_pointer() const noexcept { return data_; }
    uint64_t& operator[](size_t i) { return data_[i]; }             // host only
    const uint64_t& operator[](size_t i) const { return data_[i]; } // host only
    void set_zero();
    void resize(size_t count, bool copy_data = true);                 // dynamic_array.h:183-189: what is not copied is zero
    void resize_uninitialized(size_t count, bool copy_data = true);   //   ... or left as allocated
    void to_device_inplace(MemoryPoolHandle pool);
    void to_host_inplace();
    void copy_from(const uint64_t* src, size_t count, bool src_on_device);
private:
    void free_();
    uint64_t* data_ = nullptr;
    size_t size_ = 0;
    bool device_ = false;
    MemoryPoolHandle pool_;
    std::shared_ptr<DynamicArray> owner_;   // non-null: data_ points into *owner_
};

// utils/random_generator.h: AES-128-CTR generator; the seed/counter state lives on the host, the polynomial
// samplers run on the device through the C-ABI (troyn_sample_*).
class RandomGenerator {
public:
    explicit RandomGenerator(uint64_t seed_low = 0, uint64_t seed_high = 0) { reset_seed(seed_low, seed_high); }
    void reset_seed(uint64_t seed_low, uint64_t seed_high = 0) { seed_[0] = seed_low; seed_[1] = seed_high; counter_ = 0; }
    uint64_t sample_uint64();
    // destination: device buffer of nmod*N words (the first nmod moduli of the plan's chain)
    void sample_poly_ternary(const troyn_plan* plan, size_t nmod, uint64_t* destination);
    void sample_poly_centered_binomial(const troyn_plan* plan, size_t nmod, uint64_t* destination);
    void sample_poly_uniform(const troyn_plan* plan, size_t nmod, uint64_t* destination);
    // batched encryption: take `blocks` consecutive counter values at once (returns the first); read the seed
    uint64_t reserve_blocks(uint64_t blocks);
    const uint64_t* seed() const { return seed_; }
private:
    uint64_t seed_[2] = {0, 0};
    uint64_t counter_ = 0;
    std::mutex mutex_;
};

}  // namespace utils

namespace utils { void blake2b(void* out, size_t outlen, const void* in, size_t inlen); }   // RFC 7693, unkeyed

// helpers for code layered on the mirror (matmul.cpp): C-ABI status -> the reference's exception types, the calling
// thread's stream (its slot of the per-device stream set, troy.cpp current_stream(); the shared stream while call combining is on), and a
// wait on it.  Wait with these (or hipStreamSynchronize(0) / hipDeviceSynchronize() in a translation unit compiled WITHOUT
// -fgpu-default-stream=per-thread: the streams are blocking streams), not with hipStreamSynchronize(hipStreamPerThread).
void troyn_check_public(int rc);
troyn_stream_t troyn_current_stream();
void troyn_sync_current_stream();
namespace utils { void stream_sync(); }   // utils/memory_pool.h:37: wait for the calling thread's stream

using MemoryPool = utils::MemoryPool;
using MemoryPoolHandle = utils::MemoryPoolHandle;

// ----------------------------------------------------------------------------------------------
// Modulus, CoeffModulus, PlainModulus  (src/modulus.h, coeff_modulus.h)
// ----------------------------------------------------------------------------------------------
class Modulus {
public:
    explicit Modulus(uint64_t value = 0);
    uint64_t value() const { return value_; }
    size_t bit_count() const { return bit_count_; }
    bool is_prime() const { return is_prime_; }
    bool is_zero() const { return value_ == 0; }
    utils::ConstSlice<uint64_t> const_ratio() const { return utils::ConstSlice<uint64_t>(const_ratio_, 3, false); }   // modulus.h:94-97: floor(2^128 / value) (two words) and the remainder
    uint64_t reduce(uint64_t input) const;
    uint64_t reduce_mul_uint64(uint64_t operand1, uint64_t operand2) const {                 // modulus.h:86-92 (Barrett-128; the canonical residue)
        return static_cast<uint64_t>((static_cast<unsigned __int128>(operand1) * operand2) % value_);
    }
    uint64_t reduce_uint128(unsigned __int128 value) const { return static_cast<uint64_t>(value % value_); }      // modulus.h:43-84: the canonical residue of a 128-bit value
    uint64_t reduce_uint128_limbs(utils::ConstSlice<uint64_t> input) const { return reduce_uint128((static_cast<unsigned __int128>(input[1]) << 64) | input[0]); }
    uint64_t uint64_count() const { return 1; }                                                                  // modulus.h:105-107
private:
    uint64_t value_ = 0;
    uint64_t const_ratio_[3] = {0, 0, 0};
    size_t bit_count_ = 0;
    bool is_prime_ = false;
};

// What EncryptionParameters::plain_modulus() hands out.  The reference returns utils::ConstPointer<Modulus> (encryption_parameters.h:114: callers write
// parms.plain_modulus()->value() or *parms.plain_modulus()); earlier rounds of this mirror returned const Modulus& (callers wrote .value()).  This view does both:
// -> and * as the reference's pointer, the Modulus queries forwarded, and an implicit conversion to const Modulus&.
class ModulusPointer {
public:
    explicit ModulusPointer(const Modulus* m) : m_(m) {}
    const Modulus* operator->() const { return m_; }
    const Modulus& operator*() const { return *m_; }
    operator const Modulus&() const { return *m_; }                                               // NOLINT
    bool is_null() const noexcept { return m_ == nullptr; }
    const Modulus* get() const noexcept { return m_; }
    bool on_device() const noexcept { return false; }
    uint64_t value() const { return m_->value(); }
    size_t bit_count() const { return m_->bit_count(); }
    bool is_prime() const { return m_->is_prime(); }
    bool is_zero() const { return m_->is_zero(); }
    utils::ConstSlice<uint64_t> const_ratio() const { return m_->const_ratio(); }
    uint64_t reduce(uint64_t input) const { return m_->reduce(input); }
    uint64_t reduce_mul_uint64(uint64_t a, uint64_t b) const { return m_->reduce_mul_uint64(a, b); }
    uint64_t reduce_uint128(unsigned __int128 v) const { return m_->reduce_uint128(v); }
private:
    const Modulus* m_;
};
inline std::ostream& operator<<(std::ostream& os, const ModulusPointer& modulus) { return os << "Modulus(" << modulus.value() << ")"; }

inline std::ostream& operator<<(std::ostream& os, const Modulus& modulus) { return os << "Modulus(" << modulus.value() << ")"; }   // modulus.h:126-129

class CoeffModulus {
public:
    static size_t max_bit_count(size_t poly_modulus_degree, SecurityLevel sec_level = SecurityLevel::Classical128);
    static std::vector<Modulus> create_vector(size_t poly_modulus_degree, std::vector<size_t> bit_sizes);
    // coeff_modulus.h: the reference returns utils::Array<Modulus> (callers use .size(), [i], .to_vector(), or pass it to set_coeff_modulus)
    static utils::Array<Modulus> create(size_t poly_modulus_degree, std::vector<size_t> bit_sizes) { return utils::Array<Modulus>(create_vector(poly_modulus_degree, std::move(bit_sizes))); }
    // coeff_modulus.cu:6-63: SEAL's default BFV chains per degree and security level
    static std::vector<Modulus> bfv_default_vector(size_t poly_modulus_degree, SecurityLevel sec_level = SecurityLevel::Classical128);
    static utils::Array<Modulus> bfv_default(size_t poly_modulus_degree, SecurityLevel sec_level = SecurityLevel::Classical128) {
        return utils::Array<Modulus>(bfv_default_vector(poly_modulus_degree, sec_level));
    }
};

class PlainModulus {
public:
    static Modulus batching(size_t poly_modulus_degree, size_t bit_size);
    static utils::Array<Modulus> batching_multiple(size_t poly_modulus_degree, std::vector<size_t> bit_sizes) {
        return CoeffModulus::create(poly_modulus_degree, std::move(bit_sizes));
    }
};

// batch_utils.h: pointer collections for the x_batched forms
namespace batch_utils {
template <typename T> std::vector<T*> collect_pointer(std::vector<T>& v) { std::vector<T*> r; r.reserve(v.size()); for (T& x : v) r.push_back(&x); return r; }
template <typename T> std::vector<const T*> collect_const_pointer(const std::vector<T>& v) { std::vector<const T*> r; r.reserve(v.size()); for (const T& x : v) r.push_back(&x); return r; }
template <typename T> std::vector<const T*> pcollect_const_pointer(const std::vector<T*>& v) { return std::vector<const T*>(v.begin(), v.end()); }
// the view-collecting helpers of utils/box_batch.h:256-398 (r*: from a vector of objects, p*: from a vector of pointers), written over one generic mapper
namespace detail { template <typename R, typename V, typename F> std::vector<R> mapped(V&& v, F f) { std::vector<R> r; r.reserve(v.size()); for (auto&& x : v) r.push_back(f(x)); return r; } }
template <typename T> utils::ConstSliceVec<T> rcollect_as_const(const utils::SliceVec<T>& v) { return detail::mapped<utils::ConstSlice<T>>(v, [](const utils::Slice<T>& x) { return x.as_const(); }); }
template <typename T> std::vector<T> clone(const std::vector<T>& v) { return detail::mapped<T>(v, [](const T& x) { return x.clone(); }); }
template <typename T> std::vector<T> pclone(const std::vector<T*>& v) { return detail::mapped<T>(v, [](T* x) { return x->clone(); }); }
template <typename T, typename U = uint64_t> utils::ConstSliceVec<U> pcollect_const_reference(const std::vector<const T*>& v) { return detail::mapped<utils::ConstSlice<U>>(v, [](const T* x) { return x->const_reference(); }); }
template <typename T, typename U = uint64_t> utils::ConstSliceVec<U> rcollect_const_reference(const std::vector<T>& v) { return detail::mapped<utils::ConstSlice<U>>(v, [](const T& x) { return x.const_reference(); }); }
template <typename T, typename U = uint64_t> utils::SliceVec<U> pcollect_reference(const std::vector<T*>& v) { return detail::mapped<utils::Slice<U>>(v, [](T* x) { return x->reference(); }); }
template <typename T, typename U = uint64_t> utils::SliceVec<U> rcollect_reference(std::vector<T>& v) { return detail::mapped<utils::Slice<U>>(v, [](T& x) { return x.reference(); }); }
}  // namespace batch_utils

// ----------------------------------------------------------------------------------------------
// EncryptionParameters, ParmsID  (src/encryption_parameters.h)
// ----------------------------------------------------------------------------------------------
struct ParmsID {
    uint64_t v[4] = {0, 0, 0, 0};
    bool operator==(const ParmsID& o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
    bool operator!=(const ParmsID& o) const { return !(*this == o); }
    uint64_t operator[](size_t i) const { return v[i]; }                  // ParmsID is std::array<uint64_t, 4>-like in the reference
    uint64_t& operator[](size_t i) { return v[i]; }
    bool is_zero() const { return v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0; }
};
extern const ParmsID parms_id_zero;
struct ParmsIDHash { size_t operator()(const ParmsID& p) const { return static_cast<size_t>(p.v[0] ^ p.v[3]); } };

class EncryptionParameters {
public:
    explicit EncryptionParameters(SchemeType scheme = SchemeType::Nil) : scheme_(scheme) { compute_parms_id(); }
    void set_poly_modulus_degree(size_t n) { poly_modulus_degree_ = n; compute_parms_id(); }
    void set_coeff_modulus(const std::vector<Modulus>& q) { coeff_modulus_ = q; compute_parms_id(); }
    void set_coeff_modulus(utils::ConstSlice<Modulus> q) { coeff_modulus_ = q.to_vector(); compute_parms_id(); }
    void set_coeff_modulus(const utils::Array<Modulus>& q) { coeff_modulus_ = q.to_vector(); compute_parms_id(); }
    void set_coeff_modulus(const std::vector<uint64_t>& q) { coeff_modulus_.clear(); for (uint64_t v : q) coeff_modulus_.push_back(Modulus(v)); compute_parms_id(); }
    void set_plain_modulus(const Modulus& t) { plain_modulus_ = t; compute_parms_id(); }
    void set_plain_modulus(uint64_t t) { set_plain_modulus(Modulus(t)); }
    void set_use_special_prime_for_encryption(bool f) { use_special_prime_for_encryption_ = f; }
    SchemeType scheme() const { return scheme_; }
    size_t poly_modulus_degree() const { return poly_modulus_degree_; }
    utils::ConstSlice<Modulus> coeff_modulus() const { return utils::ConstSlice<Modulus>(coeff_modulus_); }
    ModulusPointer plain_modulus() const noexcept { return ModulusPointer(&plain_modulus_); }      // encryption_parameters.h:114-116
    const Modulus& plain_modulus_host() const noexcept { return plain_modulus_; }                 // encryption_parameters.h:118-124 (the host copies; this mirror keeps
    utils::ConstSlice<Modulus> coeff_modulus_host() const noexcept { return utils::ConstSlice<Modulus>(coeff_modulus_); }    //  parameters on the host, device constants live in the troyn_plan)
    bool use_special_prime_for_encryption() const { return use_special_prime_for_encryption_; }
    const ParmsID& parms_id() const { return parms_id_; }
    // encryption_parameters.cu:53-112 (raw little-endian fields, no compression header)
    size_t save(std::ostream& stream) const;
    void load(std::istream& stream);
    static EncryptionParameters load_new(std::istream& stream) { EncryptionParameters p(SchemeType::Nil); p.load(stream); return p; }      // encryption_parameters.h:238-242
    size_t serialized_size_upperbound() const {                                                                                          // encryption_parameters.cu:70-82
        const bool has_t = scheme_ == SchemeType::BFV || scheme_ == SchemeType::BGV;
        return sizeof(SchemeType) + 2 * sizeof(size_t) + coeff_modulus_.size() * sizeof(uint64_t) + (has_t ? sizeof(uint64_t) : 0) + sizeof(bool);
    }
private:
    void compute_parms_id();
    SchemeType scheme_;
    size_t poly_modulus_degree_ = 0;
    std::vector<Modulus> coeff_modulus_;
    Modulus plain_modulus_;
    bool use_special_prime_for_encryption_ = false;
    ParmsID parms_id_;
};

// a one-line description (the reference's special_prime_for_encryption test streams its parameters to std::cerr on failure)
inline std::ostream& operator<<(std::ostream& os, const EncryptionParameters& p) {
    static const char* const schemes[] = {"Nil", "BFV", "CKKS", "BGV"};
    os << "EncryptionParameters(scheme=" << schemes[static_cast<size_t>(p.scheme()) & 3] << ", poly_modulus_degree=" << p.poly_modulus_degree() << ", coeff_modulus=" << p.coeff_modulus();
    if (p.scheme() == SchemeType::BFV || p.scheme() == SchemeType::BGV) os << ", plain_modulus=" << p.plain_modulus();
    return os << ")";
}

// ----------------------------------------------------------------------------------------------
// ContextData, HeContext  (src/context_data.h, he_context.h)
// ----------------------------------------------------------------------------------------------
class HeContext;
class ContextData;
using ContextDataPointer = std::shared_ptr<const ContextData>;
using HeContextPointer = std::shared_ptr<HeContext>;

// encryption_parameters.h:256-275: why a parameter set was refused (HeContext::create never throws for that: it records the reason)
enum class EncryptionParameterErrorType {
    Nil = -1, Success = 0, CreatedFromDeviceParms, InvalidScheme, InvalidCoeffModulusSize, InvalidCoeffModulusBitCount, InvalidCoeffModulusNoNTT,
    InvalidPolyModulusDegree, InvalidPolyModulusDegreeNonPowerOfTwo, InvalidParametersTooLarge, InvalidParametersInsecure, FailedCreatingRNSBase,
    InvalidPlainModulusBitCount, InvalidPlainModulusCoprimality, InvalidPlainModulusTooLarge, InvalidPlainModulusNonZero, FailedCreatingRNSTool,
    FailedCreatingGaloisTool,
};

// encryption_parameters.h:277-289
struct EncryptionParameterQualifiers {
    EncryptionParameterErrorType parameter_error = EncryptionParameterErrorType::Nil;
    bool using_fft = false, using_ntt = false;
    bool using_batching = false;                 // t is an NTT prime for this degree (BFV / BGV); always for CKKS
    bool using_fast_plain_lift = false;          // every q_i > t
    bool using_descending_modulus_chain = false;
    SecurityLevel security_level = SecurityLevel::Nil;
    bool parameters_set() const noexcept { return parameter_error == EncryptionParameterErrorType::Success; }
};

class ContextData {
public:
    const EncryptionParameters& parms() const { return parms_; }
    const EncryptionParameterQualifiers& qualifiers() const { return qualifiers_; }
    // context_data.h:81-111: the host-side constants of the level (the device copies live in the troyn_plan / troyn_behz / troyn_bgv handles)
    utils::ConstSlice<uint64_t> total_coeff_modulus() const { return utils::ConstSlice<uint64_t>(total_coeff_modulus_); }      // little-endian words, coeff_modulus_size of them
    size_t total_coeff_modulus_bit_count() const { return total_coeff_modulus_bit_count_; }
    uint64_t plain_upper_half_threshold() const noexcept { return plain_upper_half_threshold_; }                               // (t + 1) / 2; 2^63 for CKKS
    utils::ConstSlice<uint64_t> plain_upper_half_increment() const { return utils::ConstSlice<uint64_t>(plain_upper_half_increment_); }
    utils::ConstSlice<uint64_t> upper_half_threshold() const { return utils::ConstSlice<uint64_t>(upper_half_threshold_); }    // CKKS: (Q + 1) / 2
    utils::ConstSlice<uint64_t> upper_half_increment() const { return utils::ConstSlice<uint64_t>(upper_half_increment_); }    // BFV / BGV: (Q mod t) mod q_i
    uint64_t coeff_modulus_mod_plain_modulus() const noexcept { return coeff_modulus_mod_plain_modulus_; }                     // Q mod t
    const ParmsID& parms_id() const { return parms_.parms_id(); }
    size_t chain_index() const { return chain_index_; }
    std::optional<ContextDataPointer> next_context_data() const noexcept { return next_ ? std::optional<ContextDataPointer>(next_) : std::nullopt; }
    // context_data.h:113-135: the link towards the key level is weak, as in the reference (.value().lock())
    std::optional<std::weak_ptr<const ContextData>> prev_context_data() const noexcept {
        return prev_.expired() ? std::nullopt : std::optional<std::weak_ptr<const ContextData>>(prev_);
    }
    std::weak_ptr<const ContextData> prev_context_data_pointer() const noexcept { return prev_; }
    ContextDataPointer next_context_data_pointer() const noexcept { return next_; }
    bool is_ckks() const noexcept { return parms_.scheme() == SchemeType::CKKS; }
    bool is_bfv() const noexcept { return parms_.scheme() == SchemeType::BFV; }
    bool is_bgv() const noexcept { return parms_.scheme() == SchemeType::BGV; }
private:
    friend class HeContext;
    void validate(SecurityLevel sec_level);      // context_data.cu:71-345
    EncryptionParameters parms_;
    EncryptionParameterQualifiers qualifiers_;
    std::vector<uint64_t> total_coeff_modulus_, plain_upper_half_increment_, upper_half_threshold_, upper_half_increment_;
    size_t total_coeff_modulus_bit_count_ = 0;
    uint64_t plain_upper_half_threshold_ = 0, coeff_modulus_mod_plain_modulus_ = 0;
    size_t chain_index_ = 0;
    std::shared_ptr<const ContextData> next_;
    std::weak_ptr<const ContextData> prev_;
};

class HeContext {
public:
    // src/he_context.cu:46-132
    static HeContextPointer create(EncryptionParameters parms, bool expand_mod_chain, SecurityLevel sec_level = SecurityLevel::Classical128,
                                   uint64_t random_seed = 0);
    ~HeContext();
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool());   // src/he_context.cu:134
    bool on_device() const noexcept { return plan_ != nullptr; }
    MemoryPoolHandle pool() const { return pool_; }
    size_t device_index() const { return pool_ ? pool_->get_device() : 0; }
    ParmsID key_parms_id() const noexcept { return key_parms_id_; }
    ParmsID first_parms_id() const noexcept { return first_parms_id_; }
    ParmsID last_parms_id() const noexcept { return last_parms_id_; }
    std::optional<ContextDataPointer> get_context_data(const ParmsID& id) const noexcept {
        auto it = map_.find(id);
        return it == map_.end() ? std::nullopt : std::optional<ContextDataPointer>(it->second);
    }
    std::optional<ContextDataPointer> key_context_data() const noexcept { return get_context_data(key_parms_id_); }
    std::optional<ContextDataPointer> first_context_data() const noexcept { return get_context_data(first_parms_id_); }
    // he_context.h:60-82: the same lookups without the optional (nullptr when the level does not exist)
    ContextDataPointer get_context_data_pointer(const ParmsID& id) const noexcept { auto c = get_context_data(id); return c.has_value() ? c.value() : nullptr; }
    ContextDataPointer key_context_data_pointer() const noexcept { return get_context_data_pointer(key_parms_id_); }
    ContextDataPointer first_context_data_pointer() const noexcept { return get_context_data_pointer(first_parms_id_); }
    ContextDataPointer last_context_data_pointer() const noexcept { return get_context_data_pointer(last_parms_id_); }
    std::optional<ContextDataPointer> last_context_data() const noexcept { return get_context_data(last_parms_id_); }
    SecurityLevel security_level() const noexcept { return security_level_; }
    bool using_keyswitching() const noexcept { return using_keyswitching_; }
    bool parameters_set() const { return parameters_set_; }
    uint64_t random_seed() const { return random_seed_; }

    utils::RandomGenerator& random_generator() const { return random_generator_; }   // he_context.h:16 (mutable shared state)

    // C-ABI handles (boundary objects)
    const troyn_plan* plan() const { return plan_; }
    const troyn_behz* behz(size_t coeff_modulus_size) const;   // created on first use
    const troyn_bgv* bgv(size_t coeff_modulus_size) const;     // BGV-only constants of the level with that many primes
    const troyn_plan* plain_plan() const;                      // NTT tables mod t (ContextData::plain_ntt_tables), first use
private:
    HeContext() = default;
    std::unordered_map<ParmsID, std::shared_ptr<ContextData>, ParmsIDHash> map_;
    ParmsID key_parms_id_, first_parms_id_, last_parms_id_;
    SecurityLevel security_level_ = SecurityLevel::Nil;
    bool using_keyswitching_ = false, parameters_set_ = false;
    uint64_t random_seed_ = 0;
    MemoryPoolHandle pool_;
    troyn_plan* plan_ = nullptr;
    mutable std::mutex behz_mutex_;
    mutable std::map<size_t, troyn_behz*> behz_;
    mutable std::map<size_t, troyn_bgv*> bgv_;
    mutable troyn_plan* plain_plan_ = nullptr;
    mutable utils::RandomGenerator random_generator_;
};



// ----------------------------------------------------------------------------------------------
// Ciphertext, Plaintext  (src/ciphertext.h, plaintext.h): data[(p*L + l)*N + i]
// ----------------------------------------------------------------------------------------------
class Ciphertext {
public:
    Ciphertext() = default;
    static Ciphertext from_members(size_t polynomial_count, size_t coeff_modulus_size, size_t poly_modulus_degree, const ParmsID& parms_id,
                                   double scale, bool is_ntt_form, uint64_t correction_factor, uint64_t seed, utils::DynamicArray&& data);
    static Ciphertext like(const Ciphertext& other, size_t polynomial_count, size_t coeff_modulus_size, bool fill_zeros,
                           MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static Ciphertext like(const Ciphertext& other, size_t polynomial_count, bool fill_zeros, MemoryPoolHandle pool = MemoryPool::GlobalPool()) {
        return like(other, polynomial_count, other.coeff_modulus_size(), fill_zeros, pool);
    }
    static Ciphertext like(const Ciphertext& other, bool fill_zeros, MemoryPoolHandle pool = MemoryPool::GlobalPool()) {
        return like(other, other.polynomial_count(), other.coeff_modulus_size(), fill_zeros, pool);
    }
    Ciphertext clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

    const ParmsID& parms_id() const noexcept { return parms_id_; }
    ParmsID& parms_id() noexcept { return parms_id_; }
    size_t polynomial_count() const noexcept { return polynomial_count_; }
    size_t coeff_modulus_size() const noexcept { return coeff_modulus_size_; }
    size_t poly_modulus_degree() const noexcept { return poly_modulus_degree_; }
    double scale() const noexcept { return scale_; }
    double& scale() noexcept { return scale_; }
    bool is_ntt_form() const noexcept { return is_ntt_form_; }
    bool& is_ntt_form() noexcept { return is_ntt_form_; }
    uint64_t correction_factor() const noexcept { return correction_factor_; }
    uint64_t& correction_factor() noexcept { return correction_factor_; }
    uint64_t seed() const noexcept { return seed_; }
    uint64_t& seed() noexcept { return seed_; }
    bool contains_seed() const noexcept { return seed_ != 0; }
    bool is_transparent() const;                          // ciphertext.cu:73-77: no data, fewer than two polynomials, or all words zero
    bool on_device() const noexcept { return data_.on_device(); }
    MemoryPoolHandle pool() const { return data_.pool(); }
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { data_.to_device_inplace(pool); }
    void to_host_inplace() { data_.to_host_inplace(); }
    Ciphertext to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext c = clone(pool); c.to_device_inplace(pool); return c; }
    Ciphertext to_host() const { Ciphertext c = *this; c.to_host_inplace(); return c; }
    const utils::DynamicArray& data() const noexcept { return data_; }
    utils::DynamicArray& data() noexcept { return data_; }
    // ciphertext.h:211-252: views of polynomial p, of polynomials [lo, hi) and of one RNS component
    utils::Slice<uint64_t> poly(size_t p) { const size_t d = coeff_modulus_size_ * poly_modulus_degree_; return utils::Slice<uint64_t>(data_.raw_pointer() + p * d, d, on_device()); }
    utils::ConstSlice<uint64_t> poly(size_t p) const { const size_t d = coeff_modulus_size_ * poly_modulus_degree_; return utils::ConstSlice<uint64_t>(data_.raw_pointer() + p * d, d, on_device()); }
    utils::ConstSlice<uint64_t> const_poly(size_t p) const { return poly(p); }
    utils::ConstSlice<uint64_t> const_reference() const { return utils::ConstSlice<uint64_t>(data_.raw_pointer(), data_.size(), on_device()); }   // ciphertext.h:199-209: all polynomials
    utils::ConstSlice<uint64_t> reference() const { return const_reference(); }
    utils::Slice<uint64_t> reference() { return utils::Slice<uint64_t>(data_.raw_pointer(), data_.size(), on_device()); }
    utils::Slice<uint64_t> polys(size_t lo, size_t hi) { const size_t d = coeff_modulus_size_ * poly_modulus_degree_; return utils::Slice<uint64_t>(data_.raw_pointer() + lo * d, (hi - lo) * d, on_device()); }
    utils::ConstSlice<uint64_t> polys(size_t lo, size_t hi) const { const size_t d = coeff_modulus_size_ * poly_modulus_degree_; return utils::ConstSlice<uint64_t>(data_.raw_pointer() + lo * d, (hi - lo) * d, on_device()); }
    utils::ConstSlice<uint64_t> const_polys(size_t lo, size_t hi) const { return polys(lo, hi); }
    utils::Slice<uint64_t> poly_component(size_t p, size_t c) { return utils::Slice<uint64_t>(data_.raw_pointer() + (p * coeff_modulus_size_ + c) * poly_modulus_degree_, poly_modulus_degree_, on_device()); }
    utils::ConstSlice<uint64_t> poly_component(size_t p, size_t c) const { return utils::ConstSlice<uint64_t>(data_.raw_pointer() + (p * coeff_modulus_size_ + c) * poly_modulus_degree_, poly_modulus_degree_, on_device()); }
    utils::ConstSlice<uint64_t> const_poly_component(size_t p, size_t c) const { return poly_component(p, c); }
    // resize(context, parms_id, polynomial_count) -- src/ciphertext.cu:26-60
    // ciphertext.h:196-197 (ciphertext.cu:25-71)
    void resize(const HeContextPointer& context, const ParmsID& parms_id, size_t polynomial_count, bool fill_extra_with_zeros = true, bool copy_data = true);
    void reconfigure_like(const HeContextPointer& context, const Ciphertext& other, size_t polynomial_count, bool fill_extra_with_zeros = true);
    // ciphertext.cu:79-210, ciphertext.h:257-270: [CompressionMode][raw fields]; a seeded ciphertext stores c0 + seed only
    size_t save(std::ostream& stream, HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const;
    void load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static Ciphertext load_new(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { Ciphertext c; c.load(stream, context, pool); return c; }
    // ADDITION: `count` ciphertexts in the byte format of `count` save() / load() calls, moved as ONE batch -- the payloads cross the bus through a pinned
    // staging image with one stream wait per batch (save) or none (load: the copies and ONE seed-expansion launch are queued on the caller's stream), instead of a
    // pageable copy, a wait and a launch per ciphertext.  save() / load() are the count == 1 case; Cipher2d::save / load and MatmulHelper::(de)serialize_outputs use them.
    static size_t save_many(std::ostream& stream, const Ciphertext* const* cts, size_t count, HeContextPointer context, CompressionMode mode = CompressionMode::Nil);
    static void load_many(std::istream& stream, Ciphertext* const* cts, size_t count, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    size_t serialized_size_upperbound(HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const;
    void expand_seed(HeContextPointer context);
    // ciphertext.cu:213-339: only the listed coefficients of c0 are written (flag bit 3); the other polynomials in full
    size_t save_terms(std::ostream& stream, HeContextPointer context, const std::vector<size_t>& terms, MemoryPoolHandle pool = MemoryPool::GlobalPool(),
                      CompressionMode mode = CompressionMode::Nil) const;
    void load_terms(std::istream& stream, HeContextPointer context, const std::vector<size_t>& terms, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static Ciphertext load_terms_new(std::istream& stream, HeContextPointer context, const std::vector<size_t>& terms, MemoryPoolHandle pool = MemoryPool::GlobalPool()) {
        Ciphertext c; c.load_terms(stream, context, terms, pool); return c;
    }
    size_t serialized_terms_size_upperbound(HeContextPointer context, size_t terms_count, CompressionMode mode = CompressionMode::Nil) const;
    size_t serialized_terms_size_upperbound(HeContextPointer context, const std::vector<size_t>& terms, CompressionMode mode = CompressionMode::Nil) const {   // ciphertext.h:283-285
        return serialized_terms_size_upperbound(context, terms.size(), mode);
    }
private:
    size_t polynomial_count_ = 0, coeff_modulus_size_ = 0, poly_modulus_degree_ = 0;
    ParmsID parms_id_;
    double scale_ = 1.0;
    bool is_ntt_form_ = false;
    uint64_t correction_factor_ = 1, seed_ = 0;
    utils::DynamicArray data_;
};

class Plaintext {
public:
    Plaintext() = default;
    const ParmsID& parms_id() const noexcept { return parms_id_; }
    ParmsID& parms_id() noexcept { return parms_id_; }
    double scale() const noexcept { return scale_; }
    double& scale() noexcept { return scale_; }
    size_t coeff_count() const noexcept { return coeff_count_; }
    size_t& coeff_count() noexcept { return coeff_count_; }
    bool is_ntt_form() const noexcept { return is_ntt_form_; }
    bool& is_ntt_form() noexcept { return is_ntt_form_; }
    bool on_device() const noexcept { return data_.on_device(); }
    const utils::DynamicArray& data() const noexcept { return data_; }
    utils::DynamicArray& data() noexcept { return data_; }
    size_t coeff_modulus_size() const noexcept { return coeff_modulus_size_; }
    size_t& coeff_modulus_size() noexcept { return coeff_modulus_size_; }
    size_t poly_modulus_degree() const noexcept { return poly_modulus_degree_; }
    size_t& poly_modulus_degree() noexcept { return poly_modulus_degree_; }
    // plaintext.h:127-170
    utils::Slice<uint64_t> poly() { return utils::Slice<uint64_t>(data_.raw_pointer(), data_.size(), on_device()); }
    utils::ConstSlice<uint64_t> poly() const { return utils::ConstSlice<uint64_t>(data_.raw_pointer(), data_.size(), on_device()); }
    utils::ConstSlice<uint64_t> const_poly() const { return poly(); }
    // plaintext.h:123-170: the data array; limb `index` of an RNS plaintext (a mod-t plaintext has the one component 0); the whole polynomial
    const utils::DynamicArray& const_data() const noexcept { return data_; }
    utils::ConstSlice<uint64_t> component(size_t index) const {
        if (parms_id_ == parms_id_zero) { if (index != 0) throw std::out_of_range("[Plaintext::component] Index out of range"); return poly(); }
        return poly().const_slice(index * coeff_count_, (index + 1) * coeff_count_);
    }
    utils::Slice<uint64_t> component(size_t index) {
        if (parms_id_ == parms_id_zero) { if (index != 0) throw std::out_of_range("[Plaintext::component] Index out of range"); return poly(); }
        return poly().slice(index * coeff_count_, (index + 1) * coeff_count_);
    }
    utils::ConstSlice<uint64_t> const_component(size_t index) const { return component(index); }
    utils::ConstSlice<uint64_t> const_reference() const { return poly(); }
    utils::ConstSlice<uint64_t> reference() const { return poly(); }                                    // plaintext.h:167-169
    utils::Slice<uint64_t> reference() { return poly(); }
    // plaintext.h:171-200: what is not copied over is zero (fill_extra_with_zeros) or left as allocated
    void resize(size_t coeff_count, bool fill_extra_with_zeros = true, bool copy_data = true) {
        if (!(parms_id_ == parms_id_zero)) throw std::invalid_argument("[Plaintext::resize] Cannot resize if the plaintext is not mod t. Call resize_rns instead.");
        coeff_count_ = coeff_count;
        if (fill_extra_with_zeros) data_.resize(coeff_count, copy_data); else data_.resize_uninitialized(coeff_count, copy_data);
    }
    void resize_rns(const HeContext& context, const ParmsID& parms_id, bool fill_extra_with_zeros = true, bool copy_data = true);
    // plaintext.cu resize_rns_partial: an RNS polynomial that keeps only its first coeff_count coefficients, data[l * coeff_count + i]
    void resize_rns_partial(const HeContext& context, const ParmsID& parms_id, size_t coeff_count, bool fill_extra_with_zeros = true, bool copy_data = true);
    Plaintext clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { (void)pool; return *this; }
    // plaintext.cu:20-70, plaintext.h save/load: [CompressionMode][raw fields]; byte-compatible with the reference
    size_t save(std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const;
    void load(std::istream& stream, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static Plaintext load_new(std::istream& stream, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { Plaintext p; p.load(stream, pool); return p; }
    size_t serialized_size_upperbound(CompressionMode mode = CompressionMode::Nil) const;
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { data_.to_device_inplace(pool); }
    void to_host_inplace() { data_.to_host_inplace(); }
    Plaintext to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p = *this; p.to_device_inplace(pool); return p; }
    Plaintext to_host() const { Plaintext p = *this; p.to_host_inplace(); return p; }
    MemoryPoolHandle pool() const { return data_.pool(); }
    std::string to_string() const;                        // plaintext.h:225-232: "7FFx^3 + 1x^1 + 3" (hexadecimal coefficients, highest degree first)
    // a full-size ([L][N], zero-padded) device copy of an RNS plaintext that keeps only coeff_count coefficients per limb
    utils::DynamicArray expanded_rns(size_t coeff_modulus_size, size_t poly_modulus_degree, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
private:
    size_t coeff_count_ = 0;
    ParmsID parms_id_;
    double scale_ = 1.0;
    bool is_ntt_form_ = false;
    size_t coeff_modulus_size_ = 0, poly_modulus_degree_ = 0;
    utils::DynamicArray data_;
};

// src/key.h: the secret key is a plaintext-shaped object holding s in NTT form under the key-level moduli
class SecretKey {
public:
    SecretKey() = default;
    explicit SecretKey(Plaintext&& p) : data_(std::move(p)) {}
    const Plaintext& as_plaintext() const { return data_; }
    Plaintext& as_plaintext() { return data_; }
    const utils::DynamicArray& data() const { return data_.data(); }
    utils::DynamicArray& data() { return data_.data(); }
    bool on_device() const { return data_.on_device(); }
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { data_.to_device_inplace(pool); }
    void to_host_inplace() { data_.to_host_inplace(); }
    SecretKey clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { (void)pool; return *this; }
    SecretKey to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { SecretKey k = *this; k.to_device_inplace(pool); return k; }
    SecretKey to_host() const { SecretKey k = *this; k.to_host_inplace(); return k; }
    const ParmsID& parms_id() const { return data_.parms_id(); }
    ParmsID& parms_id() { return data_.parms_id(); }
    size_t serialized_size_upperbound(CompressionMode mode = CompressionMode::Nil) const { return data_.serialized_size_upperbound(mode); }
    size_t save(std::ostream& stream, CompressionMode mode = CompressionMode::Nil) const { return data_.save(stream, mode); }
    void load(std::istream& stream, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { data_.load(stream, pool); }
    static SecretKey load_new(std::istream& stream, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { SecretKey k; k.load(stream, pool); return k; }
private:
    Plaintext data_;
};

// ----------------------------------------------------------------------------------------------
// Keys  (src/key.h, kswitch_keys.h)
// ----------------------------------------------------------------------------------------------
class PublicKey {
public:
    PublicKey() = default;
    explicit PublicKey(Ciphertext&& c) : data_(std::move(c)) {}
    const Ciphertext& as_ciphertext() const { return data_; }
    Ciphertext& as_ciphertext() { return data_; }
    const ParmsID& parms_id() const { return data_.parms_id(); }
    ParmsID& parms_id() { return data_.parms_id(); }
    PublicKey clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { (void)pool; return *this; }
    PublicKey to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { PublicKey k = *this; k.to_device_inplace(pool); return k; }
    PublicKey to_host() const { PublicKey k = *this; k.to_host_inplace(); return k; }
    size_t serialized_size_upperbound(HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const { return data_.serialized_size_upperbound(context, mode); }
    size_t save(std::ostream& stream, HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const { return data_.save(stream, context, mode); }
    void load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { data_.load(stream, context, pool); }
    static PublicKey load_new(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { PublicKey k; k.load(stream, context, pool); return k; }
    bool contains_seed() const { return data_.contains_seed(); }
    void expand_seed(HeContextPointer context) { data_.expand_seed(context); }
    bool on_device() const { return data_.on_device(); }
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { data_.to_device_inplace(pool); }
    void to_host_inplace() { data_.to_host_inplace(); }
private:
    Ciphertext data_;
};

class KSwitchKeys {
public:
    KSwitchKeys() = default;
    KSwitchKeys(const ParmsID& parms_id, std::vector<std::vector<PublicKey>>&& keys) : parms_id_(parms_id), keys_(std::move(keys)) {}
    const ParmsID& parms_id() const { return parms_id_; }
    ParmsID& parms_id() { return parms_id_; }
    const std::vector<std::vector<PublicKey>>& data() const { return keys_; }
    std::vector<std::vector<PublicKey>>& data() { return keys_; }
    const std::vector<PublicKey>& operator[](size_t i) const { return keys_[i]; }
    bool on_device() const;
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool());
    void to_host_inplace();
    // device pointers of key_vector[index][j] (kswitch_keys.h:34-54), as a host array for the C-ABI
    std::vector<const uint64_t*> get_data_ptrs(size_t index) const;
    const KSwitchKeys& as_kswitch_keys() const { return *this; }
    size_t key_count() const { size_t c = 0; for (const auto& v : keys_) c += !v.empty(); return c; }      // kswitch_keys.h:154
    size_t serialized_size_upperbound(HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const {
        size_t total = sizeof(ParmsID) + 2 * sizeof(size_t);
        for (const auto& v : keys_) { if (v.empty()) continue; total += 2 * sizeof(size_t); for (const PublicKey& k : v) total += k.serialized_size_upperbound(context, mode); }
        return total;
    }
    MemoryPoolHandle pool() const { for (const auto& v : keys_) for (const PublicKey& k : v) return k.as_ciphertext().pool(); return nullptr; }
    KSwitchKeys clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { (void)pool; return *this; }
    KSwitchKeys to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { KSwitchKeys k = *this; k.to_device_inplace(pool); return k; }
    KSwitchKeys to_host() const { KSwitchKeys k = *this; k.to_host_inplace(); return k; }
    // kswitch_keys.cu:5-55
    size_t save(std::ostream& stream, HeContextPointer context, CompressionMode mode = CompressionMode::Nil) const;
    void load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static KSwitchKeys load_new(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { KSwitchKeys k; k.load(stream, context, pool); return k; }   // kswitch_keys.h:205-209
private:
    ParmsID parms_id_;
    std::vector<std::vector<PublicKey>> keys_;
};

class RelinKeys : public KSwitchKeys {
public:
    RelinKeys() = default;
    explicit RelinKeys(KSwitchKeys&& k) : KSwitchKeys(std::move(k)) {}
    RelinKeys clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { (void)pool; return *this; }                      // kswitch_keys.h:226-250
    RelinKeys to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { RelinKeys k = *this; k.to_device_inplace(pool); return k; }
    RelinKeys to_host() const { RelinKeys k = *this; k.to_host_inplace(); return k; }
    static RelinKeys load_new(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { RelinKeys k; k.load(stream, context, pool); return k; }       // kswitch_keys.h:300-304
    static size_t get_index(size_t key_power) {
        if (key_power < 2) throw std::invalid_argument("[RelinKeys::get_index] key_power must be at least 2.");
        return key_power - 2;
    }
    bool has_key(size_t key_power) const { size_t i = get_index(key_power); return i < data().size() && !data()[i].empty(); }
};

// kswitch_keys.h:300-370: key-switching keys indexed by Galois element, index = (element - 1) / 2
class GaloisKeys : public KSwitchKeys {
public:
    GaloisKeys() = default;
    explicit GaloisKeys(KSwitchKeys&& k) : KSwitchKeys(std::move(k)) {}
    GaloisKeys clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { (void)pool; return *this; }                     // kswitch_keys.h:322-346
    GaloisKeys to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { GaloisKeys k = *this; k.to_device_inplace(pool); return k; }
    GaloisKeys to_host() const { GaloisKeys k = *this; k.to_host_inplace(); return k; }
    static GaloisKeys load_new(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { GaloisKeys k; k.load(stream, context, pool); return k; }     // kswitch_keys.h:393-397
    static size_t get_index(size_t galois_element) {
        if ((galois_element & 1) == 0) throw std::invalid_argument("[GaloisTool::get_index_from_element] galois_element must be odd");
        return (galois_element - 1) >> 1;
    }
    bool has_key(size_t galois_element) const { size_t i = get_index(galois_element); return i < data().size() && !data()[i].empty(); }
};

namespace utils {
// utils/galois.cu:43-96 (the permutation itself runs on the device: troyn_apply_galois)
size_t galois_element_from_step(size_t poly_modulus_degree, int step);
std::vector<size_t> galois_elements_all(size_t poly_modulus_degree);
std::vector<int> naf(int value);   // utils/number_theory.cu:6-20
}  // namespace utils

// ----------------------------------------------------------------------------------------------
// KeyGenerator  (src/key_generator.h, key_generator.cu)
// ----------------------------------------------------------------------------------------------
class KeyGenerator {
public:
    explicit KeyGenerator(HeContextPointer context, MemoryPoolHandle pool = MemoryPool::GlobalPool());                 // samples s
    KeyGenerator(HeContextPointer context, const SecretKey& secret_key, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    HeContextPointer context() const { return context_; }
    bool on_device() const { return secret_key_.on_device(); }
    // key_generator.h:35-41.  Keys of this mirror are created where the context lives (on the device), so for a device context this moves nothing
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) {
        std::lock_guard<std::mutex> lock(secret_key_array_mutex_);
        if (secret_key_array_.size()) secret_key_array_.to_device_inplace(pool);
        secret_key_.to_device_inplace(pool);
    }
    const SecretKey& secret_key() const { return secret_key_; }
    PublicKey create_public_key(bool save_seed, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    // key_generator.h:65-67: the seed of the key's c1 drawn from the caller's generator (utils/rlwe.cu symmetric_with_c1_prng)
    PublicKey create_public_key_with_u_prng(bool save_seed, utils::RandomGenerator& u_prng, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    KSwitchKeys create_keyswitching_key(const SecretKey& new_key, bool save_seed, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    RelinKeys create_relin_keys(bool save_seed, size_t max_power = 2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    GaloisKeys create_galois_keys_from_elements(const std::vector<size_t>& galois_elements, bool save_seed, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    GaloisKeys create_galois_keys_from_steps(const std::vector<int>& steps, bool save_seed, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    GaloisKeys create_galois_keys(bool save_seed, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;   // all power-of-two rotations + row swap
    // key_generator.h:101-109: the elements N/2^k + 1 used by field traces and the RLWE packing tree
    GaloisKeys create_automorphism_keys(bool save_seed, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        size_t poly_degree = context_->key_context_data().value()->parms().poly_modulus_degree();
        std::vector<size_t> galois_elements;
        while (poly_degree >= 2) { galois_elements.push_back(poly_degree + 1); poly_degree >>= 1; }
        return create_galois_keys_from_elements(galois_elements, save_seed, pool);
    }
    static void compute_secret_key_powers(HeContextPointer context, size_t max_power, utils::DynamicArray& secret_key_array);
private:
    void generate_one_kswitch_key(const uint64_t* new_key, std::vector<PublicKey>& destination, bool save_seed, MemoryPoolHandle pool) const;
    HeContextPointer context_;
    SecretKey secret_key_;
    mutable std::mutex secret_key_array_mutex_;
    mutable utils::DynamicArray secret_key_array_;   // s, s^2, ... each [K][N], NTT form
};

namespace rlwe {
// utils/rlwe.h: encryptions of zero.  `parms_id` selects the level whose moduli are used.
void symmetric(const SecretKey& sk, HeContextPointer context, const ParmsID& parms_id, bool is_ntt_form, bool save_seed,
               Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool(), utils::RandomGenerator* c1_seed_prng = nullptr);
void asymmetric(const PublicKey& pk, HeContextPointer context, const ParmsID& parms_id, bool is_ntt_form,
                Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool(), utils::RandomGenerator* u_prng = nullptr);
}  // namespace rlwe

// ----------------------------------------------------------------------------------------------
// Encryptor / Decryptor  (src/encryptor.h, decryptor.h)
// ----------------------------------------------------------------------------------------------
class Encryptor {
public:
    explicit Encryptor(HeContextPointer context) : context_(std::move(context)) {}
    HeContextPointer context() const { return context_; }
    void set_public_key(const PublicKey& public_key, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { public_key_ = public_key.clone(pool); }
    void set_secret_key(const SecretKey& secret_key, MemoryPoolHandle pool = MemoryPool::GlobalPool()) { secret_key_ = secret_key.clone(pool); }
    const PublicKey& public_key() const;
    const SecretKey& secret_key() const;
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) {                      // encryptor.h:76-83
        if (public_key_.has_value()) public_key_.value().to_device_inplace(pool);
        if (secret_key_.has_value()) secret_key_.value().to_device_inplace(pool);
    }
    // encryptor.h:140-230.  The reference's argument order: (..., u_prng = nullptr, pool = GlobalPool()).  u_prng, when given, supplies the
    // ternary u of an asymmetric encryption / the seed of c1 of a symmetric one (utils/rlwe.cu asymmetric_with_u_prng,
    // symmetric_with_c1_prng); the noise always comes from the context's generator.  The (..., pool) overloads keep three-argument
    // calls with a pool working.
    void encrypt_asymmetric(const Plaintext& plain, Ciphertext& destination, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        encrypt_internal(plain, true, false, destination, pool, u_prng);
    }
    void encrypt_asymmetric(const Plaintext& plain, Ciphertext& destination, MemoryPoolHandle pool) const { encrypt_internal(plain, true, false, destination, pool, nullptr); }
    Ciphertext encrypt_asymmetric_new(const Plaintext& plain, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Ciphertext d; encrypt_internal(plain, true, false, d, pool, u_prng); return d;
    }
    Ciphertext encrypt_asymmetric_new(const Plaintext& plain, MemoryPoolHandle pool) const { Ciphertext d; encrypt_internal(plain, true, false, d, pool, nullptr); return d; }
    void encrypt_symmetric(const Plaintext& plain, bool save_seed, Ciphertext& destination, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        encrypt_internal(plain, false, save_seed, destination, pool, u_prng);
    }
    void encrypt_symmetric(const Plaintext& plain, bool save_seed, Ciphertext& destination, MemoryPoolHandle pool) const { encrypt_internal(plain, false, save_seed, destination, pool, nullptr); }
    Ciphertext encrypt_symmetric_new(const Plaintext& plain, bool save_seed, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Ciphertext d; encrypt_internal(plain, false, save_seed, d, pool, u_prng); return d;
    }
    Ciphertext encrypt_symmetric_new(const Plaintext& plain, bool save_seed, MemoryPoolHandle pool) const { Ciphertext d; encrypt_internal(plain, false, save_seed, d, pool, nullptr); return d; }
    // batched forms: a caller-supplied generator takes the per-object route (its draws interleave with the context's); without one the
    // BFV symmetric form is the batched kernel sequence below
    void encrypt_asymmetric_batched(const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination, utils::RandomGenerator* u_prng = nullptr,
                                    MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (plain.size() != destination.size()) throw std::invalid_argument("[Encryptor::encrypt_internal_batched] Input and destination have different sizes.");
        for (size_t i = 0; i < plain.size(); i++) encrypt_internal(*plain[i], true, false, *destination[i], pool, u_prng);
    }
    std::vector<Ciphertext> encrypt_asymmetric_new_batched(const std::vector<const Plaintext*>& plain, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Ciphertext> d(plain.size()); encrypt_asymmetric_batched(plain, batch_utils::collect_pointer(d), u_prng, pool); return d;
    }
    void encrypt_symmetric_batched(const std::vector<const Plaintext*>& plain, bool save_seed, const std::vector<Ciphertext*>& destination, utils::RandomGenerator* u_prng,
                                   MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (u_prng == nullptr) { encrypt_symmetric_batched(plain, save_seed, destination, pool); return; }
        if (plain.size() != destination.size()) throw std::invalid_argument("[Encryptor::encrypt_internal_batched] Input and destination have different sizes.");
        for (size_t i = 0; i < plain.size(); i++) encrypt_internal(*plain[i], false, save_seed, *destination[i], pool, u_prng);
    }
    std::vector<Ciphertext> encrypt_symmetric_new_batched(const std::vector<const Plaintext*>& plain, bool save_seed, utils::RandomGenerator* u_prng = nullptr,
                                                          MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Ciphertext> d(plain.size()); encrypt_symmetric_batched(plain, save_seed, batch_utils::collect_pointer(d), u_prng, pool); return d;
    }
    // encryptor.h encrypt_symmetric_batched: the same ciphertexts, bit for bit, as encrypt_symmetric called once per plaintext
    // in order (the generator positions are reproduced), in a constant number of launches; the results share one buffer
    void encrypt_symmetric_batched(const std::vector<const Plaintext*>& plain, bool save_seed, const std::vector<Ciphertext*>& destination,
                                   MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    // the same for `count` BFV plaintexts (parms_id_zero, coefficient form) already on the device, `stride` words apart
    // ntt_seeded: the form MatmulHelper sends (app/matmul.cu:300-311, encrypt_symmetric_batched(.., save_seed = true) of scaled-up
    // NTT-form plaintexts): NTT-form ciphertexts that carry the seed of c1 (half the wire size; expand_seed before use)
    std::vector<Ciphertext> encrypt_symmetric_packed(const uint64_t* plains, size_t coeff_count, size_t stride, size_t count,
                                                     MemoryPoolHandle pool = MemoryPool::GlobalPool(), bool ntt_seeded = false) const;
    void encrypt_zero_asymmetric(Ciphertext& destination, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void encrypt_zero_asymmetric(Ciphertext& destination, std::optional<ParmsID> parms_id, MemoryPoolHandle pool) const { encrypt_zero_asymmetric(destination, parms_id, nullptr, pool); }
    Ciphertext encrypt_zero_asymmetric_new(std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Ciphertext d; encrypt_zero_asymmetric(d, parms_id, u_prng, pool); return d;
    }
    Ciphertext encrypt_zero_asymmetric_new(std::optional<ParmsID> parms_id, MemoryPoolHandle pool) const { Ciphertext d; encrypt_zero_asymmetric(d, parms_id, nullptr, pool); return d; }
    void encrypt_zero_symmetric(bool save_seed, Ciphertext& destination, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr,
                                MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void encrypt_zero_symmetric(bool save_seed, Ciphertext& destination, std::optional<ParmsID> parms_id, MemoryPoolHandle pool) const { encrypt_zero_symmetric(save_seed, destination, parms_id, nullptr, pool); }
    Ciphertext encrypt_zero_symmetric_new(bool save_seed, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Ciphertext d; encrypt_zero_symmetric(save_seed, d, parms_id, u_prng, pool); return d;
    }
    Ciphertext encrypt_zero_symmetric_new(bool save_seed, std::optional<ParmsID> parms_id, MemoryPoolHandle pool) const { Ciphertext d; encrypt_zero_symmetric(save_seed, d, parms_id, nullptr, pool); return d; }
    void encrypt_zero_asymmetric_batched(const std::vector<Ciphertext*>& destination, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr,
                                         MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        for (Ciphertext* d : destination) encrypt_zero_asymmetric(*d, parms_id, u_prng, pool);
    }
    std::vector<Ciphertext> encrypt_zero_asymmetric_new_batched(size_t count, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr,
                                                                MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Ciphertext> d(count); encrypt_zero_asymmetric_batched(batch_utils::collect_pointer(d), parms_id, u_prng, pool); return d;
    }
    void encrypt_zero_symmetric_batched(bool save_seed, const std::vector<Ciphertext*>& destination, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr,
                                        MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        for (Ciphertext* d : destination) encrypt_zero_symmetric(save_seed, *d, parms_id, u_prng, pool);
    }
    std::vector<Ciphertext> encrypt_zero_symmetric_new_batched(size_t count, bool save_seed, std::optional<ParmsID> parms_id = std::nullopt, utils::RandomGenerator* u_prng = nullptr,
                                                               MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Ciphertext> d(count); encrypt_zero_symmetric_batched(save_seed, batch_utils::collect_pointer(d), parms_id, u_prng, pool); return d;
    }
private:
    void encrypt_zero_internal(const ParmsID& parms_id, bool is_ntt_form, bool is_asymmetric, bool save_seed, Ciphertext& destination, MemoryPoolHandle pool,
                               utils::RandomGenerator* u_prng = nullptr) const;
    void encrypt_internal(const Plaintext& plain, bool is_asymmetric, bool save_seed, Ciphertext& destination, MemoryPoolHandle pool, utils::RandomGenerator* u_prng = nullptr) const;
    HeContextPointer context_;
    std::optional<PublicKey> public_key_;
    std::optional<SecretKey> secret_key_;
};

class Decryptor {
public:
    Decryptor(HeContextPointer context, const SecretKey& secret_key, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    HeContextPointer context() const { return context_; }
    bool on_device() const { return secret_key_array_.on_device(); }
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { std::lock_guard<std::mutex> lock(secret_key_array_mutex_); if (secret_key_array_.size()) secret_key_array_.to_device_inplace(pool); }   // decryptor.h:43
    void decrypt(const Ciphertext& encrypted, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext decrypt_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; decrypt(encrypted, d, pool); return d; }
    // decryptor.cu:581-640 (BFV / BGV): bits of room left before decryption fails; the phase is formed on the device, the
    // centred infinity norm of its CRT composition (a diagnostic, outside the hot path) on the host
    size_t invariant_noise_budget(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    // decryptor.cu:268-289: the phase c(s) = Delta m + v as an RNS polynomial of the ciphertext's level (coefficient form); the
    // ring-2^k encoder scales it down itself
    void bfv_decrypt_without_scaling_down(const Ciphertext& encrypted, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext bfv_decrypt_without_scaling_down_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext d; bfv_decrypt_without_scaling_down(encrypted, d, pool); return d;
    }
    // decryptor.h decrypt_batched (BFV two-polynomial ciphertexts of one level take the batched path; anything else loops)
    void decrypt_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<Plaintext> decrypt_batched_new(const std::vector<const Ciphertext*>& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Plaintext> d(encrypted.size()); decrypt_batched(encrypted, batch_utils::collect_pointer(d), pool); return d;
    }
    void bfv_decrypt_without_scaling_down_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (encrypted.size() != destination.size()) throw std::invalid_argument("[Decryptor::bfv_decrypt_without_scaling_down_batched] Input and destination have different sizes.");
        for (size_t i = 0; i < encrypted.size(); i++) bfv_decrypt_without_scaling_down(*encrypted[i], *destination[i], pool);
    }
    std::vector<Plaintext> bfv_decrypt_without_scaling_down_batched_new(const std::vector<const Ciphertext*>& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Plaintext> d(encrypted.size()); bfv_decrypt_without_scaling_down_batched(encrypted, batch_utils::collect_pointer(d), pool); return d;
    }
    // the plaintext coefficients of every ciphertext, concatenated on the host ([count][N]); one device-to-host copy
    std::vector<uint64_t> bfv_decrypt_to_host(const std::vector<const Ciphertext*>& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
private:
    void dot_product_ct_sk_array(const Ciphertext& encrypted, uint64_t* destination, MemoryPoolHandle pool) const;
    bool bfv_batchable(const std::vector<const Ciphertext*>& encrypted) const;
    std::shared_ptr<utils::DynamicArray> bfv_decrypt_batch_device(const std::vector<const Ciphertext*>& encrypted, MemoryPoolHandle pool) const;   // [count][N] mod t
    HeContextPointer context_;
    mutable std::mutex secret_key_array_mutex_;
    mutable utils::DynamicArray secret_key_array_;
};

// ----------------------------------------------------------------------------------------------
// BatchEncoder  (src/batch_encoder.h, batch_encoder.cu)
// ----------------------------------------------------------------------------------------------
class BatchEncoder {
public:
    explicit BatchEncoder(HeContextPointer context);
    HeContextPointer context() const { return context_; }
    size_t slot_count() const { return slots_; }
    bool on_device() const { return context_->on_device(); }
    void encode(const std::vector<uint64_t>& values, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext encode_new(const std::vector<uint64_t>& values, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; encode(values, p, pool); return p; }
    void decode(const Plaintext& plain, std::vector<uint64_t>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<uint64_t> decode_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<uint64_t> v; decode(plain, v, pool); return v; }
    // coefficient ("polynomial") packing: values are the plaintext coefficients themselves (batch_encoder.cu encode_polynomial)
    Plaintext encode_polynomial_new(const std::vector<uint64_t>& values, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<uint64_t> decode_polynomial_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void encode_polynomial(const std::vector<uint64_t>& values, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = encode_polynomial_new(values, pool); }
    void decode_polynomial(const Plaintext& plain, std::vector<uint64_t>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = decode_polynomial_new(plain, pool); }
    // the *_slice forms (batch_encoder.h:55-131): the values come from / go to a (pointer, length) view that may live on the host or on the device
    void encode_slice(utils::ConstSlice<uint64_t> values, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { encode(utils::slice_to_vector(values), destination, pool); }
    Plaintext encode_slice_new(utils::ConstSlice<uint64_t> values, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; encode_slice(values, p, pool); return p; }
    void encode_slice_batched(const utils::ConstSliceVec<uint64_t>& values, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (values.size() != destination.size()) throw std::invalid_argument("[BatchEncoder::encode_slice_batched] values and destination size mismatch.");
        for (size_t i = 0; i < values.size(); i++) encode_slice(values[i], *destination[i], pool);
    }
    std::vector<Plaintext> encode_slice_new_batched(const utils::ConstSliceVec<uint64_t>& values, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<Plaintext> d(values.size());
        for (size_t i = 0; i < values.size(); i++) encode_slice(values[i], d[i], pool);
        return d;
    }
    void encode_polynomial_slice(utils::ConstSlice<uint64_t> values, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = encode_polynomial_new(utils::slice_to_vector(values), pool); }
    Plaintext encode_polynomial_slice_new(utils::ConstSlice<uint64_t> values, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { return encode_polynomial_new(utils::slice_to_vector(values), pool); }
    void decode_slice(const Plaintext& plaintext, utils::Slice<uint64_t> destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (destination.size() != slots_) throw std::invalid_argument("[BatchEncoder::decode_slice] Destination has incorrect size.");
        utils::vector_to_slice(decode_new(plaintext, pool), destination, "[BatchEncoder::decode_slice]");
    }
    utils::Array<uint64_t> decode_slice_new(const Plaintext& plaintext, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { return utils::located(utils::Array<uint64_t>(decode_new(plaintext, pool)), plaintext.on_device(), pool); }
    void decode_slice_batched(const std::vector<const Plaintext*>& plaintexts, const std::vector<utils::Slice<uint64_t>>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (plaintexts.size() != destination.size()) throw std::invalid_argument("[BatchEncoder::decode_slice_batched] plaintexts and destination size mismatch.");
        for (size_t i = 0; i < plaintexts.size(); i++) decode_slice(*plaintexts[i], destination[i], pool);
    }
    std::vector<utils::Array<uint64_t>> decode_slice_new_batched(const std::vector<const Plaintext*>& plaintexts, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<utils::Array<uint64_t>> d;
        for (const Plaintext* p : plaintexts) d.push_back(decode_slice_new(*p, pool));
        return d;
    }
    void decode_polynomial_slice(const Plaintext& plaintext, utils::Slice<uint64_t> destination) const {
        const std::vector<uint64_t> v = decode_polynomial_new(plaintext);
        if (destination.size() != v.size()) throw std::invalid_argument("[BatchEncoder::decode_polynomial_slice] Destination has incorrect size.");
        utils::vector_to_slice(v, destination, "[BatchEncoder::decode_polynomial_slice]");
    }
    utils::Array<uint64_t> decode_polynomial_slice_new(const Plaintext& plaintext, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { return utils::located(utils::Array<uint64_t>(decode_polynomial_new(plaintext, pool)), plaintext.on_device(), pool); }
    constexpr size_t row_count() const noexcept { return 2; }
    size_t column_count() const noexcept { return slots_ / 2; }
    bool simd_encoding_supported() const { return !matrix_reps_index_map_.empty(); }
    void to_device_inplace(MemoryPoolHandle = MemoryPool::GlobalPool()) {}      // the index map stays on the host; encode / decode stage through the device themselves
    // batch_encoder.cu:558-662: a mod-t plaintext to / from its RNS form at a level.  scale_up = round(q/t * m) (what encryption adds to c0),
    // centralize = the centred lift (what multiply_plain uses); scale_down / decentralize are their inverses (the final steps of BFV /
    // BGV decryption).  The RNS plaintexts keep only the source's coeff_count coefficients per limb ("partial").
    Plaintext scale_up_new(const Plaintext& plain, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void scale_up(const Plaintext& plain, Plaintext& destination, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = scale_up_new(plain, parms_id, pool); }
    void scale_up_inplace(Plaintext& plain, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { plain = scale_up_new(plain, parms_id, pool); }
    Plaintext scale_down_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void scale_down(const Plaintext& plain, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = scale_down_new(plain, pool); }
    void scale_down_inplace(Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { plain = scale_down_new(plain, pool); }
    Plaintext centralize_new(const Plaintext& plain, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void centralize(const Plaintext& plain, Plaintext& destination, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = centralize_new(plain, parms_id, pool); }
    void centralize_inplace(Plaintext& plain, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { plain = centralize_new(plain, parms_id, pool); }
    Plaintext decentralize_new(const Plaintext& plain, uint64_t correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void decentralize(const Plaintext& plain, Plaintext& destination, uint64_t correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = decentralize_new(plain, correction_factor, pool); }
    void decentralize_inplace(Plaintext& plain, uint64_t correction_factor = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { plain = decentralize_new(plain, correction_factor, pool); }
private:
    HeContextPointer context_;
    size_t slots_ = 0;
    std::vector<size_t> matrix_reps_index_map_;
};

// ----------------------------------------------------------------------------------------------
// CKKSEncoder  (src/ckks_encoder.h, ckks_encoder.cu).  The canonical-embedding FFT and the RNS conversion of the
// scaled coefficients are floating-point / host work outside the integer hot path: they run on the host in double
// precision (same slot order as the reference: Galois generator 3, bit-reversed), the NTT on the device.
// ----------------------------------------------------------------------------------------------
class CKKSEncoder {
public:
    explicit CKKSEncoder(HeContextPointer context);
    HeContextPointer context() const { return context_; }
    size_t slot_count() const { return slots_; }
    size_t polynomial_modulus_degree() const { return slots_ * 2; }
    bool on_device() const { return context_->on_device(); }
    void to_device_inplace(MemoryPoolHandle = MemoryPool::GlobalPool()) {}
    void encode_complex64_simd(const std::vector<std::complex<double>>& values, std::optional<ParmsID> parms_id, double scale, Plaintext& destination,
                               MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext encode_complex64_simd_new(const std::vector<std::complex<double>>& values, std::optional<ParmsID> parms_id, double scale,
                                        MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; encode_complex64_simd(values, parms_id, scale, p, pool); return p; }
    // the same value in every slot = a constant polynomial (ckks_encoder.h:120-135)
    void encode_float64_single(double value, std::optional<ParmsID> parms_id, double scale, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        set_plaintext(std::vector<double>{value * scale}, parms_id.value_or(context_->first_parms_id()), scale, destination, pool);
    }
    Plaintext encode_float64_single_new(double value, std::optional<ParmsID> parms_id, double scale, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; encode_float64_single(value, parms_id, scale, p, pool); return p;
    }
    void encode_float64_polynomial(const std::vector<double>& values, std::optional<ParmsID> parms_id, double scale, Plaintext& destination,
                                   MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext encode_float64_polynomial_new(const std::vector<double>& values, std::optional<ParmsID> parms_id, double scale,
                                            MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; encode_float64_polynomial(values, parms_id, scale, p, pool); return p; }
    // one complex value in every slot (ckks_encoder.h:42-48, :171-186)
    void encode_complex64_single(std::complex<double> value, std::optional<ParmsID> parms_id, double scale, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        encode_complex64_simd(std::vector<std::complex<double>>(slots_, value), parms_id, scale, destination, pool);
    }
    Plaintext encode_complex64_single_new(std::complex<double> value, std::optional<ParmsID> parms_id, double scale, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; encode_complex64_single(value, parms_id, scale, p, pool); return p;
    }
    // exact integers, scale 1 (ckks_encoder.cu:983-1090): coefficient j <- values[j] mod q_i (a single value = the constant polynomial)
    void encode_integer64_polynomial(const std::vector<int64_t>& values, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext encode_integer64_polynomial_new(const std::vector<int64_t>& values, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; encode_integer64_polynomial(values, parms_id, p, pool); return p;
    }
    void encode_integer64_single(int64_t value, std::optional<ParmsID> parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        encode_integer64_polynomial(std::vector<int64_t>{value}, parms_id, destination, pool);
    }
    Plaintext encode_integer64_single_new(int64_t value, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; encode_integer64_single(value, parms_id, p, pool); return p;
    }
    void decode_complex64_simd(const Plaintext& plain, std::vector<std::complex<double>>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<std::complex<double>> decode_complex64_simd_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<std::complex<double>> v; decode_complex64_simd(plain, v, pool); return v;
    }
    void decode_float64_polynomial(const Plaintext& plain, std::vector<double>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<double> decode_float64_polynomial_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        std::vector<double> v; decode_float64_polynomial(plain, v, pool); return v;
    }
    // the *_slice forms (ckks_encoder.h:93-283): values from / to host or device views
    void encode_complex64_simd_slice(utils::ConstSlice<std::complex<double>> values, std::optional<ParmsID> parms_id, double scale, Plaintext& destination,
                                     MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { encode_complex64_simd(utils::slice_to_vector(values), parms_id, scale, destination, pool); }
    Plaintext encode_complex64_simd_slice_new(utils::ConstSlice<std::complex<double>> values, std::optional<ParmsID> parms_id, double scale,
                                              MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; encode_complex64_simd_slice(values, parms_id, scale, p, pool); return p; }
    void encode_float64_polynomial_slice(utils::ConstSlice<double> values, std::optional<ParmsID> parms_id, double scale, Plaintext& destination,
                                         MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { encode_float64_polynomial(utils::slice_to_vector(values), parms_id, scale, destination, pool); }
    Plaintext encode_float64_polynomial_slice_new(utils::ConstSlice<double> values, std::optional<ParmsID> parms_id, double scale,
                                                  MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext p; encode_float64_polynomial_slice(values, parms_id, scale, p, pool); return p; }
    void encode_integer64_polynomial_slice(utils::ConstSlice<int64_t> values, std::optional<ParmsID> parms_id, Plaintext& destination,
                                           MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { encode_integer64_polynomial(utils::slice_to_vector(values), parms_id, destination, pool); }
    Plaintext encode_integer64_polynomial_slice_new(utils::ConstSlice<int64_t> values, std::optional<ParmsID> parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Plaintext p; encode_integer64_polynomial_slice(values, parms_id, p, pool); return p;
    }
    void decode_complex64_simd_slice(const Plaintext& plain, utils::Slice<std::complex<double>> destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (destination.size() != slots_) throw std::invalid_argument("[ckks_encoder::decode_complex64_simd_slice] destination size must be equal to slot_count.");
        utils::vector_to_slice(decode_complex64_simd_new(plain, pool), destination, "[ckks_encoder::decode_complex64_simd_slice]");
    }
    utils::Array<std::complex<double>> decode_complex64_simd_slice_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        return utils::located(utils::Array<std::complex<double>>(decode_complex64_simd_new(plain, pool)), plain.on_device(), pool);
    }
    void decode_float64_polynomial_slice(const Plaintext& plain, utils::Slice<double> destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        if (destination.size() != slots_ * 2) throw std::invalid_argument("[ckks_encoder::decode_float64_polynomial_slice] destination size must be equal to slot_count * 2.");
        utils::vector_to_slice(decode_float64_polynomial_new(plain, pool), destination, "[ckks_encoder::decode_float64_polynomial_slice]");
    }
    utils::Array<double> decode_float64_polynomial_slice_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        return utils::located(utils::Array<double>(decode_float64_polynomial_new(plain, pool)), plain.on_device(), pool);
    }
private:
    void set_plaintext(const std::vector<double>& coeffs, const ParmsID& parms_id, double scale, Plaintext& destination, MemoryPoolHandle pool) const;
    std::vector<double> plaintext_coefficients(const Plaintext& plain, MemoryPoolHandle pool) const;   // centred, as doubles (not yet / scale)
    HeContextPointer context_;
    size_t slots_ = 0;
    std::vector<size_t> matrix_reps_index_map_;
    std::vector<std::complex<double>> root_powers_, inv_root_powers_;   // bit-reversed order, as NTTTables
};

// ----------------------------------------------------------------------------------------------
// LWECiphertext  (src/lwe_ciphertext.h, lwe_ciphertext.cu): c0 u64[L], c1 u64[L][N]
// ----------------------------------------------------------------------------------------------
class LWECiphertext {
public:
    LWECiphertext() = default;
    LWECiphertext clone(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    bool on_device() const;
    MemoryPoolHandle pool() const { return c1_.pool(); }                                        // lwe_ciphertext.h:21
    void to_device_inplace(MemoryPoolHandle pool = MemoryPool::GlobalPool()) { c0_.to_device_inplace(pool); c1_.to_device_inplace(pool); }
    void to_host_inplace() { c0_.to_host_inplace(); c1_.to_host_inplace(); }
    LWECiphertext to_device(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { LWECiphertext r = clone(pool); r.to_device_inplace(pool); return r; }
    LWECiphertext to_host() const { LWECiphertext r = clone(); r.to_host_inplace(); return r; }
    size_t coeff_modulus_size() const noexcept { return coeff_modulus_size_; }
    size_t& coeff_modulus_size() noexcept { return coeff_modulus_size_; }
    size_t poly_modulus_degree() const noexcept { return poly_modulus_degree_; }
    size_t& poly_modulus_degree() noexcept { return poly_modulus_degree_; }
    const utils::DynamicArray& c0_dyn() const { return c0_; }
    utils::DynamicArray& c0_dyn() { return c0_; }
    const utils::DynamicArray& c1_dyn() const { return c1_; }
    utils::DynamicArray& c1_dyn() { return c1_; }
    const uint64_t* c0() const { return c0_.raw_pointer(); }
    const uint64_t* c1() const { return c1_.raw_pointer(); }
    utils::ConstSlice<uint64_t> const_c0() const { return utils::ConstSlice<uint64_t>(c0_.raw_pointer(), c0_.size(), c0_.on_device()); }      // lwe_ciphertext.h:96-100
    utils::ConstSlice<uint64_t> const_c1() const { return utils::ConstSlice<uint64_t>(c1_.raw_pointer(), c1_.size(), c1_.on_device()); }
    const ParmsID& parms_id() const noexcept { return parms_id_; }
    ParmsID& parms_id() noexcept { return parms_id_; }
    double scale() const noexcept { return scale_; }
    double& scale() noexcept { return scale_; }
    uint64_t correction_factor() const noexcept { return correction_factor_; }
    uint64_t& correction_factor() noexcept { return correction_factor_; }
    // lwe_ciphertext.cu:9-60: the RLWE ciphertext (c0 as constant coefficient, c1) -- coefficient form
    Ciphertext assemble_lwe(MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    static std::vector<Ciphertext> assemble_lwe_batched_new(const std::vector<const LWECiphertext*>& lwes, MemoryPoolHandle pool = MemoryPool::GlobalPool());
    static void assemble_lwe_batched(const std::vector<const LWECiphertext*>& source, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) {
        std::vector<Ciphertext> r = assemble_lwe_batched_new(source, pool);
        if (r.size() != destination.size()) throw std::invalid_argument("[LWECiphertext::assemble_lwe_batched] Input and destination have different sizes.");
        for (size_t i = 0; i < r.size(); i++) *destination[i] = std::move(r[i]);
    }
private:
    size_t coeff_modulus_size_ = 0, poly_modulus_degree_ = 0;
    utils::DynamicArray c0_, c1_;
    ParmsID parms_id_ = parms_id_zero;
    double scale_ = 1.0;
    uint64_t correction_factor_ = 1;
};

// ----------------------------------------------------------------------------------------------
// Call combining (ADDITION; off unless switched on).  The reference's concurrency model is N host threads, each issuing single-object
// Evaluator calls on its own stream (test/bench/he_operations.cu:85, :364-380 `-c N`; test/test_multithread.cu:18-37) -- it has no
// batched multiply / relinearize / rescale.  On this GPU small kernels of different streams overlap at most ~4-fold
// (tools/ubench/stream_overlap.hip), so N such threads get <= 4 / (GPU time of one op), while ONE launch sequence over 16 objects costs
// barely more than over one (tools/small_batch_sweep.py).  With combining on
//   * every host thread's calls go to ONE shared stream instead of the thread's slot of the stream set (one order for everything the threads queue: no
//     cross-stream dependencies, the pool may hand a block released by one thread to another at once);
//   * a thread that enters multiply / relinearize / rescale_to_next / multiply_relinearize_rescale / apply_galois (hence every rotation and
//     complex_conjugate) while other threads are doing the same
//     hands its (already checked) call to a rendezvous: the first thread to arrive leads, waits at most `window` for the other active
//     threads, queues the calls of ITS shape as one library call (operands staged by one gather launch, results delivered to the
//     callers' own arrays by one scatter launch) and every caller returns -- as with the uncombined call, the work is queued, not
//     finished: wait with utils::stream_sync() / troyn_sync_current_stream() (or hipDeviceSynchronize), NOT with
//     hipStreamSynchronize(hipStreamPerThread), which no longer is the stream the work is on.
// Results are the library's batched results, i.e. bit-identical to the uncombined calls.  A thread that is alone (no other thread inside
// these methods within the last few ms) is never combined.  Switch it at a quiescent point (no other thread inside the library): the
// switch waits for the device.  Environment: TROY_COMBINE=1 switches it on from the first call, TROY_COMBINE_WINDOW_US=<n> sets the
// window (default 100).  A one-device mode: the shared stream lives on the device that was current at the first call after the switch.
// ----------------------------------------------------------------------------------------------
namespace combining {
void set_enabled(bool on);
bool enabled();
void set_window_us(unsigned microseconds);
unsigned window_us();
struct Stats {
    uint64_t calls = 0, batches = 0, largest_batch = 0;   // calls that went through a batch, batches run, size of the largest
    uint64_t uncombined = 0;                              // calls that led alone and ran the ordinary way
    uint64_t gather_ns = 0, execute_ns = 0;               // leaders: time spent waiting for the others / running and waiting for the batch
    uint64_t between_ns = 0, between_calls = 0;           // callers: time from leaving one combined call to entering the next (host work of the caller)
    uint64_t between_wait_ns = 0, between_wait_calls = 0; // the same when the caller waited for the stream in between
    uint64_t spread_ns = 0;                               // first to last arrival of a batch
    uint64_t window_expired = 0, target_sum = 0;          // leaders that stopped waiting because the window passed; sum of the thread counts they waited for
};
Stats stats();
void reset_stats();
}  // namespace combining

namespace detail {
// one checked single-object call handed to the rendezvous (combine.cpp); the submitting thread has done every argument check and knows
// the result's metadata, the leader only needs shapes and pointers
enum class CombineKind : uint8_t { DyadicMultiply, BfvMultiply, Relinearize, Rescale, MultiplyRelinearizeRescale, ApplyGalois };
struct CombineRequest {
    CombineKind kind;
    const void* handle = nullptr;                 // troyn_plan* (troyn_behz* for BfvMultiply): same context + level
    uint32_t L = 0, p1 = 0, p2 = 0;               // limbs, polynomial counts of the operands (ApplyGalois: p2 = the Galois element)
    bool ckks = false, ntt_form = false;
    const std::vector<const uint64_t*>* keys = nullptr;   // key-switching forms: the L key pointers (identity: the first pointer)
    const uint64_t* in1 = nullptr; size_t words1 = 0;
    const uint64_t* in2 = nullptr; size_t words2 = 0;
    uint64_t* out = nullptr; size_t out_words = 0;   // the caller's own destination array (allocated by the calling thread from ITS share of the pool)
    // set by the leader
    std::exception_ptr error;
    static constexpr size_t FANOUT = 4;
    CombineRequest* wake[FANOUT] = {};              // the waiters this one wakes on its way out (the release fans out as a tree: a wake costs
                                                    // microseconds per sleeper, the leader alone would release 64 callers one after the other)
    std::atomic<int> state{0};
};
// true: the call ran as part of a batch (or failed there: throws); false: not combined, the caller runs it itself
bool combine_submit(CombineRequest& request, MemoryPoolHandle pool);
bool combining_wanted();   // switched on AND another thread is active in the combinable methods
bool combining_on();
void combining_switch(bool on);
int combining_stream_wait(void* stream);   // hipStreamSynchronize(shared stream), one waiter at a time (returns the hipError_t)
bool on_combining_stream();                // combining is on and this thread's calls go to the shared stream (its current device is the stream's)
}  // namespace detail

// ----------------------------------------------------------------------------------------------
// Evaluator  (src/evaluator.h)
// ----------------------------------------------------------------------------------------------
class Evaluator {
public:
    enum class SwitchKeyDestinationAssignMethod { AddInplace = 0, Overwrite = 1, OverwriteExceptFirst = 2 };

    explicit Evaluator(HeContextPointer context) : context_(std::move(context)) {}
    HeContextPointer context() const { return context_; }
    bool on_device() const { return context_->on_device(); }

    // negate -- evaluator.h:113-133
    void negate_inplace(Ciphertext& encrypted) const;
    void negate(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext negate_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; negate(encrypted, d, pool); return d; }
    void negate_inplace_batched(const std::vector<Ciphertext*>& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void negate_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

    // add / sub -- evaluator.h:138-220 (translate, evaluator_translate.cu:64-118)
    void add_inplace(Ciphertext& e1, const Ciphertext& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_inplace(e1, e2, false, pool); }
    void add(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& d, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate(e1, e2, d, false, pool); }
    Ciphertext add_new(const Ciphertext& e1, const Ciphertext& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; add(e1, e2, d, pool); return d; }
    void sub_inplace(Ciphertext& e1, const Ciphertext& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_inplace(e1, e2, true, pool); }
    void sub(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& d, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate(e1, e2, d, true, pool); }
    Ciphertext sub_new(const Ciphertext& e1, const Ciphertext& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; sub(e1, e2, d, pool); return d; }
    void add_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d,
                     MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void sub_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d,
                     MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

    // multiply / square -- evaluator.h:225-262 (evaluator.cu:29-343)
    void multiply(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void multiply_inplace(Ciphertext& e1, const Ciphertext& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; multiply(e1, e2, d, pool); e1 = std::move(d); }
    Ciphertext multiply_new(const Ciphertext& e1, const Ciphertext& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; multiply(e1, e2, d, pool); return d; }
    void square(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void square_inplace(Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; square(encrypted, d, pool); encrypted = std::move(d); }
    Ciphertext square_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; square(encrypted, d, pool); return d; }
    void multiply_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d,
                          MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;   // addition (see header comment)

    // ADDITION to the reference's interface: destination = rescale_to_next(relinearize(multiply(e1, e2), relin_keys)) as ONE library call
    // (troyn_ckks_multiply_relinearize_rescale: 5 launches instead of 8, the 3-component product never reaches HBM).  The result -- payload,
    // parms_id, scale, form -- is bit-identical to composing evaluator.h's three methods (evaluator.cu:118-145, evaluator_keyswitching.cu:119-144,
    // evaluator_modswitch.cu:445-461) and every argument check of those three is applied with the reference's messages.  Operands that are
    // not 2-component CKKS ciphertexts are evaluated by composing the three methods.
    void multiply_relinearize_rescale(const Ciphertext& e1, const Ciphertext& e2, const RelinKeys& relin_keys, Ciphertext& destination,
                                      MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void multiply_relinearize_rescale_inplace(Ciphertext& e1, const Ciphertext& e2, const RelinKeys& relin_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Ciphertext d; multiply_relinearize_rescale(e1, e2, relin_keys, d, pool); e1 = std::move(d);
    }
    Ciphertext multiply_relinearize_rescale_new(const Ciphertext& e1, const Ciphertext& e2, const RelinKeys& relin_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const {
        Ciphertext d; multiply_relinearize_rescale(e1, e2, relin_keys, d, pool); return d;
    }
    void multiply_relinearize_rescale_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const RelinKeys& relin_keys,
                                              const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

    // ciphertext x plaintext -- evaluator.h (multiply_plain*, transform_plain_to_ntt*); evaluator_multiply_plain.cu,
    // evaluator_transform_ntt.cu:35-70
    void transform_plain_to_ntt(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext transform_plain_to_ntt_new(const Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; transform_plain_to_ntt(plain, parms_id, d, pool); return d; }
    void transform_plain_to_ntt_inplace(Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; transform_plain_to_ntt(plain, parms_id, d, pool); plain = std::move(d); }
    void multiply_plain(const Ciphertext& encrypted, const Plaintext& plain, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void multiply_plain_inplace(Ciphertext& encrypted, const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; multiply_plain(encrypted, plain, d, pool); encrypted = std::move(d); }
    Ciphertext multiply_plain_new(const Ciphertext& encrypted, const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; multiply_plain(encrypted, plain, d, pool); return d; }
    // destination[i] (+)= encrypted[i] * plain[i]; equal destination pointers accumulate (MatmulHelper::matmul's call)
    void multiply_plain_accumulate(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain,
                                   const std::vector<Ciphertext*>& destination, bool set_zero = true, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;   // evaluator.h:491-497
    void multiply_plain_ntt_accumulate(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain,
                                       const std::vector<Ciphertext*>& destination, bool set_zero = true, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;   // evaluator_multiply_plain.cu:258-307

    // key switching -- evaluator.h:267-303 (evaluator_keyswitching.cu)
    void apply_keyswitching_inplace(Ciphertext& encrypted, const KSwitchKeys& kswitch_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void apply_keyswitching(const Ciphertext& encrypted, const KSwitchKeys& kswitch_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext apply_keyswitching_new(const Ciphertext& encrypted, const KSwitchKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; apply_keyswitching(encrypted, k, d, pool); return d; }
    void relinearize_inplace(Ciphertext& encrypted, const RelinKeys& relin_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { relinearize_inplace_internal(encrypted, relin_keys, 2, pool); }
    void relinearize(const Ciphertext& encrypted, const RelinKeys& relin_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { relinearize_internal(encrypted, relin_keys, 2, destination, pool); }
    Ciphertext relinearize_new(const Ciphertext& encrypted, const RelinKeys& relin_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; relinearize(encrypted, relin_keys, d, pool); return d; }
    void relinearize_batched(const std::vector<const Ciphertext*>& encrypted, const RelinKeys& relin_keys, const std::vector<Ciphertext*>& d,
                             MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;   // addition

    // Galois automorphisms / rotations -- evaluator.h:700-850 (evaluator_keyswitching.cu:147-361)
    void apply_galois(const Ciphertext& encrypted, size_t galois_element, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    // evaluator_keyswitching.cu:235-261: the automorphism on a plaintext (mod t in coefficient form, per limb for RNS plaintexts, a gather in NTT form)
    void apply_galois_plain(const Plaintext& plain, size_t galois_element, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void apply_galois_plain_inplace(Plaintext& plain, size_t galois_element, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; apply_galois_plain(plain, galois_element, d, pool); plain = std::move(d); }
    Plaintext apply_galois_plain_new(const Plaintext& plain, size_t galois_element, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; apply_galois_plain(plain, galois_element, d, pool); return d; }
    // ---- the remaining spellings of evaluator.h: x_inplace_batched / x_new_batched over the batched cores, and the plaintext-side families ----
    typedef std::vector<const Ciphertext*> ConstCiphers;
    typedef std::vector<Ciphertext*> Ciphers;
    typedef std::vector<const Plaintext*> ConstPlains;
    typedef std::vector<Plaintext*> Plains;
    std::vector<Ciphertext> negate_new_batched(const ConstCiphers& e, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); negate_batched(e, batch_utils::collect_pointer(d), pool); return d; }
    void add_inplace_batched(const Ciphers& e1, const ConstCiphers& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_batched(batch_utils::pcollect_const_pointer(e1), e2, e1, false, pool); }
    void sub_inplace_batched(const Ciphers& e1, const ConstCiphers& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_batched(batch_utils::pcollect_const_pointer(e1), e2, e1, true, pool); }
    void translate_inplace_batched(const Ciphers& e1, const ConstCiphers& e2, bool subtract, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_batched(batch_utils::pcollect_const_pointer(e1), e2, e1, subtract, pool); }
    std::vector<Ciphertext> add_new_batched(const ConstCiphers& e1, const ConstCiphers& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e1.size()); add_batched(e1, e2, batch_utils::collect_pointer(d), pool); return d; }
    std::vector<Ciphertext> sub_new_batched(const ConstCiphers& e1, const ConstCiphers& e2, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e1.size()); sub_batched(e1, e2, batch_utils::collect_pointer(d), pool); return d; }
    std::vector<Ciphertext> transform_to_ntt_new_batched(const ConstCiphers& e, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); transform_to_ntt_batched(e, batch_utils::collect_pointer(d), pool); return d; }
    std::vector<Ciphertext> transform_from_ntt_new_batched(const ConstCiphers& e, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); transform_from_ntt_batched(e, batch_utils::collect_pointer(d), pool); return d; }
    void mod_switch_to_next_inplace_batched(const Ciphers& e, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { mod_switch_to_next_batched(batch_utils::pcollect_const_pointer(e), e, pool); }
    std::vector<Ciphertext> mod_switch_to_next_new_batched(const ConstCiphers& e, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); mod_switch_to_next_batched(e, batch_utils::collect_pointer(d), pool); return d; }
    // evaluator_modswitch.cu mod_switch_to_batched: level by level down to parms_id (the batch moves together when it is uniform)
    void mod_switch_to_batched(const ConstCiphers& encrypted, const ParmsID& parms_id, const Ciphers& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void mod_switch_to_inplace_batched(const Ciphers& e, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { mod_switch_to_batched(batch_utils::pcollect_const_pointer(e), parms_id, e, pool); }
    std::vector<Ciphertext> mod_switch_to_new_batched(const ConstCiphers& e, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); mod_switch_to_batched(e, parms_id, batch_utils::collect_pointer(d), pool); return d; }
    void multiply_plain_inplace_batched(const Ciphers& e, const ConstPlains& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { multiply_plain_batched(batch_utils::pcollect_const_pointer(e), plain, e, pool); }
    std::vector<Ciphertext> multiply_plain_new_batched(const ConstCiphers& e, const ConstPlains& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); multiply_plain_batched(e, plain, batch_utils::collect_pointer(d), pool); return d; }
    // evaluator_keyswitching.cu:52-93 apply_keyswitching_batched: one gather + one key-switch launch sequence for a uniform batch
    void apply_keyswitching_batched(const ConstCiphers& encrypted, const KSwitchKeys& kswitch_keys, const Ciphers& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void apply_keyswitching_inplace_batched(const Ciphers& e, const KSwitchKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { apply_keyswitching_batched(batch_utils::pcollect_const_pointer(e), k, e, pool); }
    std::vector<Ciphertext> apply_keyswitching_new_batched(const ConstCiphers& e, const KSwitchKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); apply_keyswitching_batched(e, k, batch_utils::collect_pointer(d), pool); return d; }
    void apply_galois_inplace_batched(const Ciphers& e, size_t galois_element, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { apply_galois_batched(batch_utils::pcollect_const_pointer(e), galois_element, k, e, pool); }
    std::vector<Ciphertext> apply_galois_new_batched(const ConstCiphers& e, size_t galois_element, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); apply_galois_batched(e, galois_element, k, batch_utils::collect_pointer(d), pool); return d; }
    // rotations of a batch (evaluator_keyswitching.cu rotate_internal_batched): one apply_galois_batched per step of the NAF decomposition
    void rotate_internal_batched(const ConstCiphers& encrypted, int steps, const GaloisKeys& galois_keys, const Ciphers& destination, MemoryPoolHandle pool) const;
    void rotate_rows_batched(const ConstCiphers& e, int steps, const GaloisKeys& k, const Ciphers& d, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rotate_rows_inplace_batched(const Ciphers& e, int steps, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { rotate_rows_batched(batch_utils::pcollect_const_pointer(e), steps, k, e, pool); }
    std::vector<Ciphertext> rotate_rows_new_batched(const ConstCiphers& e, int steps, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); rotate_rows_batched(e, steps, k, batch_utils::collect_pointer(d), pool); return d; }
    void rotate_columns_batched(const ConstCiphers& e, const GaloisKeys& k, const Ciphers& d, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rotate_columns_inplace_batched(const Ciphers& e, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { rotate_columns_batched(batch_utils::pcollect_const_pointer(e), k, e, pool); }
    std::vector<Ciphertext> rotate_columns_new_batched(const ConstCiphers& e, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); rotate_columns_batched(e, k, batch_utils::collect_pointer(d), pool); return d; }
    void rotate_vector_batched(const ConstCiphers& e, int steps, const GaloisKeys& k, const Ciphers& d, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rotate_vector_inplace_batched(const Ciphers& e, int steps, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { rotate_vector_batched(batch_utils::pcollect_const_pointer(e), steps, k, e, pool); }
    std::vector<Ciphertext> rotate_vector_new_batched(const ConstCiphers& e, int steps, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); rotate_vector_batched(e, steps, k, batch_utils::collect_pointer(d), pool); return d; }
    void complex_conjugate_inplace(Ciphertext& encrypted, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; complex_conjugate(encrypted, k, d, pool); encrypted = std::move(d); }
    void complex_conjugate_batched(const ConstCiphers& e, const GaloisKeys& k, const Ciphers& d, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void complex_conjugate_inplace_batched(const Ciphers& e, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { complex_conjugate_batched(batch_utils::pcollect_const_pointer(e), k, e, pool); }
    std::vector<Ciphertext> complex_conjugate_new_batched(const ConstCiphers& e, const GaloisKeys& k, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); complex_conjugate_batched(e, k, batch_utils::collect_pointer(d), pool); return d; }
    void negacyclic_shift_batched(const ConstCiphers& encrypted, size_t shift, const Ciphers& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void negacyclic_shift_inplace_batched(const Ciphers& e, size_t shift, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { negacyclic_shift_batched(batch_utils::pcollect_const_pointer(e), shift, e, pool); }
    std::vector<Ciphertext> negacyclic_shift_new_batched(const ConstCiphers& e, size_t shift, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Ciphertext> d(e.size()); negacyclic_shift_batched(e, shift, batch_utils::collect_pointer(d), pool); return d; }
    void divide_by_poly_modulus_degree_inplace_batched(const Ciphers& encrypted, uint64_t mul = 1, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext pack_lwe_ciphertexts_new(const std::vector<LWECiphertext>& lwes, const GaloisKeys& automorphism_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool(),
                                        bool apply_field_trace = true) const {         // evaluator.h:970-973, kept by the reference for older callers
        return pack_lwe_ciphertexts_new(batch_utils::collect_const_pointer(lwes), automorphism_keys, pool, apply_field_trace);
    }
    void pack_lwe_ciphertexts(const std::vector<const LWECiphertext*>& lwes, const GaloisKeys& automorphism_keys, Ciphertext& output, MemoryPoolHandle pool = MemoryPool::GlobalPool(),
                              bool apply_field_trace = true) const { output = pack_lwe_ciphertexts_new(lwes, automorphism_keys, pool, apply_field_trace); }
    void pack_lwe_ciphertexts_batched(const std::vector<std::vector<const LWECiphertext*>>& lwe_groups, const GaloisKeys& automorphism_keys, const Ciphers& output,
                                      MemoryPoolHandle pool = MemoryPool::GlobalPool(), bool apply_field_trace = true) const {
        std::vector<Ciphertext> r = pack_lwe_ciphertexts_new_batched(lwe_groups, automorphism_keys, pool, apply_field_trace);
        if (r.size() != output.size()) throw std::invalid_argument("[Evaluator::pack_lwe_ciphertexts_batched] Input and destination have different sizes.");
        for (size_t i = 0; i < r.size(); i++) *output[i] = std::move(r[i]);
    }
    // evaluator.h:520-590: a mod-t plaintext to its RNS form at a level (what BatchEncoder::centralize / scale_up return, from the evaluator)
    void bfv_centralize(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext bfv_centralize_new(const Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; bfv_centralize(plain, parms_id, d, pool); return d; }
    void bfv_centralize_inplace(Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; bfv_centralize(plain, parms_id, d, pool); plain = std::move(d); }
    void bfv_centralize_batched(const ConstPlains& plain, const ParmsID& parms_id, const Plains& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<Plaintext> bfv_centralize_new_batched(const ConstPlains& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Plaintext> d(plain.size()); bfv_centralize_batched(plain, parms_id, batch_utils::collect_pointer(d), pool); return d; }
    void bfv_centralize_inplace_batched(const Plains& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { bfv_centralize_batched(batch_utils::pcollect_const_pointer(plain), parms_id, plain, pool); }
    void bfv_scale_up(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Plaintext bfv_scale_up_new(const Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; bfv_scale_up(plain, parms_id, d, pool); return d; }
    void bfv_scale_up_inplace(Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; bfv_scale_up(plain, parms_id, d, pool); plain = std::move(d); }
    void bfv_scale_up_batched(const ConstPlains& plain, const ParmsID& parms_id, const Plains& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    std::vector<Plaintext> bfv_scale_up_new_batched(const ConstPlains& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Plaintext> d(plain.size()); bfv_scale_up_batched(plain, parms_id, batch_utils::collect_pointer(d), pool); return d; }
    void bfv_scale_up_inplace_batched(const Plains& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { bfv_scale_up_batched(batch_utils::pcollect_const_pointer(plain), parms_id, plain, pool); }
    // evaluator.h:596-660: plaintext NTT transforms, batched and inverse
    void transform_plain_to_ntt_batched(const ConstPlains& plain, const ParmsID& parms_id, const Plains& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void transform_plain_to_ntt_inplace_batched(const Plains& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { transform_plain_to_ntt_batched(batch_utils::pcollect_const_pointer(plain), parms_id, plain, pool); }
    std::vector<Plaintext> transform_plain_to_ntt_new_batched(const ConstPlains& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Plaintext> d(plain.size()); transform_plain_to_ntt_batched(plain, parms_id, batch_utils::collect_pointer(d), pool); return d; }
    void transform_plain_from_ntt(const Plaintext& plain, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void transform_plain_from_ntt_inplace(Plaintext& plain) const { Plaintext d; transform_plain_from_ntt(plain, d); plain = std::move(d); }
    Plaintext transform_plain_from_ntt_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; transform_plain_from_ntt(plain, d, pool); return d; }
    void transform_plain_from_ntt_batched(const ConstPlains& plain, const Plains& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void transform_plain_from_ntt_inplace_batched(const Plains& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { transform_plain_from_ntt_batched(batch_utils::pcollect_const_pointer(plain), plain, pool); }
    std::vector<Plaintext> transform_plain_from_ntt_new_batched(const ConstPlains& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { std::vector<Plaintext> d(plain.size()); transform_plain_from_ntt_batched(plain, batch_utils::collect_pointer(d), pool); return d; }
    void apply_galois_inplace(Ciphertext& encrypted, size_t galois_element, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; apply_galois(encrypted, galois_element, galois_keys, d, pool); encrypted = std::move(d); }
    Ciphertext apply_galois_new(const Ciphertext& encrypted, size_t galois_element, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; apply_galois(encrypted, galois_element, galois_keys, d, pool); return d; }
    void rotate_rows(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rotate_rows_inplace(Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rotate_rows(encrypted, steps, galois_keys, d, pool); encrypted = std::move(d); }
    Ciphertext rotate_rows_new(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rotate_rows(encrypted, steps, galois_keys, d, pool); return d; }
    // CKKS: rotate_vector / complex_conjugate (evaluator.h:850-940)
    void rotate_vector(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext rotate_vector_new(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rotate_vector(encrypted, steps, galois_keys, d, pool); return d; }
    void rotate_vector_inplace(Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rotate_vector(encrypted, steps, galois_keys, d, pool); encrypted = std::move(d); }
    void complex_conjugate(const Ciphertext& encrypted, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext complex_conjugate_new(const Ciphertext& encrypted, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; complex_conjugate(encrypted, galois_keys, d, pool); return d; }
    void rotate_columns(const Ciphertext& encrypted, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rotate_columns_inplace(Ciphertext& encrypted, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rotate_columns(encrypted, galois_keys, d, pool); encrypted = std::move(d); }
    Ciphertext rotate_columns_new(const Ciphertext& encrypted, const GaloisKeys& galois_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rotate_columns(encrypted, galois_keys, d, pool); return d; }

    // The x_batched forms below run ONE launch per step over the whole batch when the operands are uniform (same level, size,
    // form, scale): scattered operands are staged into a contiguous block by one gather launch (adjacent windows of one buffer --
    // what these functions themselves return -- are used in place), and the results are windows of one shared buffer.  Mixed
    // batches and batches below BATCH_OP_THRESHOLD take the per-object path (utils/constants.h BATCH_OP_THRESHOLD).
    static constexpr size_t BATCH_OP_THRESHOLD = 4;
    void transform_to_ntt_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void transform_from_ntt_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void apply_galois_batched(const std::vector<const Ciphertext*>& encrypted, size_t galois_element, const GaloisKeys& galois_keys, const std::vector<Ciphertext*>& destination,
                              MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void add_plain_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination,
                           MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_plain_batched(encrypted, plain, destination, false, pool); }
    void sub_plain_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination,
                           MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_plain_batched(encrypted, plain, destination, true, pool); }
    void multiply_plain_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination,
                                MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

    // ciphertext +/- plaintext -- evaluator.h (add_plain*, sub_plain*); evaluator_translate_plain.cu:13-170
    void add_plain_inplace(Ciphertext& encrypted, const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_plain_inplace(encrypted, plain, false, pool); }
    void add_plain(const Ciphertext& encrypted, const Plaintext& plain, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = encrypted.clone(pool); translate_plain_inplace(destination, plain, false, pool); }
    Ciphertext add_plain_new(const Ciphertext& encrypted, const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; add_plain(encrypted, plain, d, pool); return d; }
    void sub_plain_inplace(Ciphertext& encrypted, const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { translate_plain_inplace(encrypted, plain, true, pool); }
    void sub_plain(const Ciphertext& encrypted, const Plaintext& plain, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { destination = encrypted.clone(pool); translate_plain_inplace(destination, plain, true, pool); }
    Ciphertext sub_plain_new(const Ciphertext& encrypted, const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; sub_plain(encrypted, plain, d, pool); return d; }

    // LWE extraction and RLWE packing -- evaluator.h:940-1041 (evaluator_lwes.cu)
    LWECiphertext extract_lwe_new(const Ciphertext& encrypted, size_t term, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext assemble_lwe_new(const LWECiphertext& lwe, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { return lwe.assemble_lwe(pool); }
    void field_trace_inplace(Ciphertext& encrypted, const GaloisKeys& automorphism_keys, size_t logn, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void field_trace_inplace_batched(const std::vector<Ciphertext*>& encrypted, const GaloisKeys& automorphism_keys, size_t logn, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void divide_by_poly_modulus_degree_inplace(Ciphertext& encrypted, uint64_t mul = 1) const;
    void negacyclic_shift(const Ciphertext& encrypted, size_t shift, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext negacyclic_shift_new(const Ciphertext& encrypted, size_t shift, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; negacyclic_shift(encrypted, shift, d, pool); return d; }
    void negacyclic_shift_inplace(Ciphertext& encrypted, size_t shift, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; negacyclic_shift(encrypted, shift, d, pool); encrypted = std::move(d); }
    Ciphertext pack_lwe_ciphertexts_new(const std::vector<const LWECiphertext*>& lwes, const GaloisKeys& automorphism_keys, MemoryPoolHandle pool = MemoryPool::GlobalPool(),
                                        bool apply_field_trace = true) const;
    std::vector<Ciphertext> pack_lwe_ciphertexts_new_batched(const std::vector<std::vector<const LWECiphertext*>>& lwe_groups, const GaloisKeys& automorphism_keys,
                                                             MemoryPoolHandle pool = MemoryPool::GlobalPool(), bool apply_field_trace = true) const;
    Ciphertext pack_rlwe_ciphertexts_new(const std::vector<const Ciphertext*>& ciphers, const GaloisKeys& automorphism_keys, size_t shift, size_t input_interval,
                                         size_t output_interval, MemoryPoolHandle pool = MemoryPool::GlobalPool(), bool apply_field_trace = true) const;
    // every group's packing tree advances together: one fused layer kernel and one batched key switch per layer
    void pack_rlwe_ciphertexts_batched(const std::vector<std::vector<const Ciphertext*>>& cipher_groups, const GaloisKeys& automorphism_keys, size_t shift,
                                       size_t input_interval, size_t output_interval, const std::vector<Ciphertext*>& outputs,
                                       MemoryPoolHandle pool = MemoryPool::GlobalPool(), bool apply_field_trace = true) const;
    std::vector<Ciphertext> pack_rlwe_ciphertexts_new_batched(const std::vector<std::vector<const Ciphertext*>>& cipher_groups, const GaloisKeys& automorphism_keys,
                                                              size_t shift, size_t input_interval, size_t output_interval,
                                                              MemoryPoolHandle pool = MemoryPool::GlobalPool(), bool apply_field_trace = true) const {
        std::vector<Ciphertext> out(cipher_groups.size());
        std::vector<Ciphertext*> ptrs;
        for (Ciphertext& c : out) ptrs.push_back(&c);
        pack_rlwe_ciphertexts_batched(cipher_groups, automorphism_keys, shift, input_interval, output_interval, ptrs, pool, apply_field_trace);
        return out;
    }

    // modulus switching -- evaluator.h:308-420 (evaluator_modswitch.cu)
    void mod_switch_to_next(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void mod_switch_to_next_inplace(Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; mod_switch_to_next(encrypted, d, pool); encrypted = std::move(d); }
    Ciphertext mod_switch_to_next_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; mod_switch_to_next(encrypted, d, pool); return d; }
    void mod_switch_to_next_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void mod_switch_to(const Ciphertext& encrypted, const ParmsID& parms_id, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void mod_switch_to_inplace(Ciphertext& encrypted, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; mod_switch_to(encrypted, parms_id, d, pool); encrypted = std::move(d); }
    Ciphertext mod_switch_to_new(const Ciphertext& encrypted, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; mod_switch_to(encrypted, parms_id, d, pool); return d; }
    void rescale_to_next(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rescale_to_next_inplace(Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rescale_to_next(encrypted, d, pool); encrypted = std::move(d); }
    Ciphertext rescale_to_next_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rescale_to_next(encrypted, d, pool); return d; }
    void rescale_to(const Ciphertext& encrypted, const ParmsID& parms_id, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void rescale_to_inplace(Ciphertext& encrypted, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rescale_to(encrypted, parms_id, d, pool); encrypted = std::move(d); }
    Ciphertext rescale_to_new(const Ciphertext& encrypted, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; rescale_to(encrypted, parms_id, d, pool); return d; }
    // NTT-form (CKKS) plaintexts follow the ciphertexts down the chain -- evaluator_modswitch.cu:279-316,:402-443
    void mod_switch_plain_to(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void mod_switch_plain_to_inplace(Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; mod_switch_plain_to(plain, parms_id, d, pool); plain = std::move(d); }
    Plaintext mod_switch_plain_to_new(const Plaintext& plain, const ParmsID& parms_id, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; mod_switch_plain_to(plain, parms_id, d, pool); return d; }
    void mod_switch_plain_to_next(const Plaintext& plain, Plaintext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void mod_switch_plain_to_next_inplace(Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; mod_switch_plain_to_next(plain, d, pool); plain = std::move(d); }
    Plaintext mod_switch_plain_to_next_new(const Plaintext& plain, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Plaintext d; mod_switch_plain_to_next(plain, d, pool); return d; }
    void rescale_to_next_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;   // addition

    // NTT -- evaluator.h:655-700 (evaluator_transform_ntt.cu:469-652)
    void transform_to_ntt_inplace(Ciphertext& encrypted) const;
    void transform_to_ntt(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext transform_to_ntt_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; transform_to_ntt(encrypted, d, pool); return d; }
    void transform_from_ntt_inplace(Ciphertext& encrypted) const;
    void transform_from_ntt(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    Ciphertext transform_from_ntt_new(const Ciphertext& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const { Ciphertext d; transform_from_ntt(encrypted, d, pool); return d; }
    void transform_to_ntt_inplace_batched(const std::vector<Ciphertext*>& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;
    void transform_from_ntt_inplace_batched(const std::vector<Ciphertext*>& encrypted, MemoryPoolHandle pool = MemoryPool::GlobalPool()) const;

private:
    ContextDataPointer get_context_data(const char* prompt, const ParmsID& id) const;
    void translate_inplace(Ciphertext& e1, const Ciphertext& e2, bool subtract, MemoryPoolHandle pool) const;
    void translate(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& d, bool subtract, MemoryPoolHandle pool) const;
    void translate_plain_inplace(Ciphertext& encrypted, const Plaintext& plain, bool subtract, MemoryPoolHandle pool) const;
    void translate_plain_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain, const std::vector<Ciphertext*>& destination,
                                 bool subtract, MemoryPoolHandle pool) const;
    void translate_batched(const std::vector<const Ciphertext*>& e1, const std::vector<const Ciphertext*>& e2, const std::vector<Ciphertext*>& d, bool subtract,
                           MemoryPoolHandle pool) const;
    void switch_key_checks(const Ciphertext& encrypted, const KSwitchKeys& kswitch_keys, size_t kswitch_keys_index, const Ciphertext& destination) const;
    // argument checks + result object (allocated, metadata set) of the three methods, without the device work (troy.cpp)
    SchemeType multiply_prepare(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& out, MemoryPoolHandle pool) const;
    void relinearize_prepare(const Ciphertext& encrypted, const RelinKeys& relin_keys, Ciphertext& out, std::vector<const uint64_t*>& key_ptrs, MemoryPoolHandle pool) const;
    SchemeType mod_switch_scale_prepare(const Ciphertext& encrypted, Ciphertext& out, MemoryPoolHandle pool) const;
    void apply_galois_prepare(const Ciphertext& encrypted, size_t galois_element, const GaloisKeys& galois_keys, Ciphertext& out, std::vector<const uint64_t*>& key_ptrs,
                              MemoryPoolHandle pool) const;
    void switch_key_internal(const Ciphertext& encrypted, const uint64_t* target, const KSwitchKeys& kswitch_keys, size_t kswitch_keys_index,
                             SwitchKeyDestinationAssignMethod assign_method, Ciphertext& destination, MemoryPoolHandle pool) const;
    void relinearize_inplace_internal(Ciphertext& encrypted, const RelinKeys& relin_keys, size_t destination_size, MemoryPoolHandle pool) const;
    void rotate_internal(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const;
    void relinearize_internal(const Ciphertext& encrypted, const RelinKeys& relin_keys, size_t destination_size, Ciphertext& destination, MemoryPoolHandle pool) const;
    void mod_switch_scale_to_next_internal(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const;
    void mod_switch_drop_to_internal(const Ciphertext& encrypted, Ciphertext& destination, const ParmsID& target, MemoryPoolHandle pool) const;
    // argument checks + result metadata of multiply -> relinearize -> rescale_to_next; false: the operands do not take the fused entry
    bool multiply_relinearize_rescale_prepare(const Ciphertext& e1, const Ciphertext& e2, const RelinKeys& relin_keys, uint32_t& L, ParmsID& next_parms_id, double& scale,
                                              std::vector<const uint64_t*>& key_ptrs) const;
    HeContextPointer context_;
};

// batch_utils.h:28-135: views of one polynomial (or a range of polynomials) of every ciphertext / of every plaintext of a batch
namespace batch_utils {
inline utils::ConstSliceVec<uint64_t> pcollect_const_poly(const std::vector<const Ciphertext*>& v, size_t poly_id) { return detail::mapped<utils::ConstSlice<uint64_t>>(v, [poly_id](const Ciphertext* c) { return c->const_poly(poly_id); }); }
inline utils::ConstSliceVec<uint64_t> rcollect_const_poly(const std::vector<Ciphertext>& v, size_t poly_id) { return detail::mapped<utils::ConstSlice<uint64_t>>(v, [poly_id](const Ciphertext& c) { return c.const_poly(poly_id); }); }
inline utils::ConstSliceVec<uint64_t> pcollect_const_poly(const std::vector<const Plaintext*>& v) { return detail::mapped<utils::ConstSlice<uint64_t>>(v, [](const Plaintext* p) { return p->const_poly(); }); }
inline utils::ConstSliceVec<uint64_t> rcollect_const_poly(std::vector<Plaintext>& v) { return detail::mapped<utils::ConstSlice<uint64_t>>(v, [](const Plaintext& p) { return p.const_poly(); }); }
inline utils::SliceVec<uint64_t> pcollect_poly(const std::vector<Ciphertext*>& v, size_t poly_id) { return detail::mapped<utils::Slice<uint64_t>>(v, [poly_id](Ciphertext* c) { return c->poly(poly_id); }); }
inline utils::SliceVec<uint64_t> rcollect_poly(std::vector<Ciphertext>& v, size_t poly_id) { return detail::mapped<utils::Slice<uint64_t>>(v, [poly_id](Ciphertext& c) { return c.poly(poly_id); }); }
inline utils::SliceVec<uint64_t> pcollect_poly(const std::vector<Plaintext*>& v) { return detail::mapped<utils::Slice<uint64_t>>(v, [](Plaintext* p) { return p->poly(); }); }
inline utils::SliceVec<uint64_t> rcollect_poly(std::vector<Plaintext>& v) { return detail::mapped<utils::Slice<uint64_t>>(v, [](Plaintext& p) { return p.poly(); }); }
inline utils::ConstSliceVec<uint64_t> pcollect_const_polys(const std::vector<const Ciphertext*>& v, size_t lo, size_t hi) { return detail::mapped<utils::ConstSlice<uint64_t>>(v, [lo, hi](const Ciphertext* c) { return c->const_polys(lo, hi); }); }
inline utils::ConstSliceVec<uint64_t> rcollect_const_polys(const std::vector<Ciphertext>& v, size_t lo, size_t hi) { return detail::mapped<utils::ConstSlice<uint64_t>>(v, [lo, hi](const Ciphertext& c) { return c.const_polys(lo, hi); }); }
inline utils::SliceVec<uint64_t> pcollect_polys(const std::vector<Ciphertext*>& v, size_t lo, size_t hi) { return detail::mapped<utils::Slice<uint64_t>>(v, [lo, hi](Ciphertext* c) { return c->polys(lo, hi); }); }
inline utils::SliceVec<uint64_t> rcollect_polys(std::vector<Ciphertext>& v, size_t lo, size_t hi) { return detail::mapped<utils::Slice<uint64_t>>(v, [lo, hi](Ciphertext& c) { return c.polys(lo, hi); }); }
}  // namespace batch_utils

}  // namespace troy

// The reference tree's one extern "C" entry (rustbind/wrapper.h:11-12): bindgen cannot return a std::shared_ptr, so the handle of a new
// pool on `device_index` is written through `out`.
namespace troy_wrapper {
extern "C" void create_memory_pool_handle(size_t device_index, troy::MemoryPoolHandle* out);
}
