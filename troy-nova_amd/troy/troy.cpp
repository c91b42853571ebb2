// troy.cpp -- bodies of the host-side mirror declared in troy.h.  Host logic only; every device
// operation goes through the C-ABI (include/troyn.h).  Reference file:line citations name the
// function whose behaviour (checks, metadata updates, error text) is mirrored.
#include "troy.h"

#include <atomic>
#include <unordered_set>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <istream>
#include <ostream>
#include <sstream>
#include <dlfcn.h>

namespace troy {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
static void hip_check(hipError_t e, const char* what) {
    // kernel_provider.h:11-16: runtime failures are std::runtime_error with the runtime's message
    if (e != hipSuccess) throw std::runtime_error(std::string("[kernel_provider::") + what + "] " + hipGetErrorString(e));
}

static void troyn_check(int rc) {
    if (rc == 0) return;
    if (rc < 0) throw std::invalid_argument(troyn_last_error());
    throw std::runtime_error(troyn_last_error());
}

// The stream this thread's calls launch on: its slot of the per-device stream set (below) -- or, while call combining is on (troy.h), ONE stream shared by every
// host thread: the batches of combined calls and whatever else the threads queue are then ordered by that stream alone, so a combined
// call needs no cross-stream dependency and nobody waits for the GPU inside it.
namespace detail {
static std::atomic<int> g_combining{-1};   // -1: not decided yet (environment TROY_COMBINE), 0 off, 1 on
static std::once_flag g_shared_once;
static hipStream_t g_shared_stream = nullptr;
static int g_shared_device = -1;
bool combining_on() {
    int v = g_combining.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = std::getenv("TROY_COMBINE");
        v = (e && e[0] == '1') ? 1 : 0;
        g_combining.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}
// created on the device that is current at the first call after the switch; null only if that failed
hipStream_t shared_stream() {
    std::call_once(g_shared_once, [] {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return;
        hipStream_t s = nullptr;
        if (hipStreamCreate(&s) != hipSuccess) { (void)hipGetLastError(); return; }   // a blocking stream: hipStreamSynchronize(0), the reference tool's idiom, still covers it
        g_shared_stream = s;
        g_shared_device = dev;
    });
    return g_shared_stream;   // no per-call device query (a runtime call per launch site from every thread): combining is a one-device mode
}
void combining_switch(bool on) {
    if (combining_on() == on) return;
    // a quiescent point by contract (no other thread inside the library); everything queued so far completes, every cached block is anybody's
    (void)hipDeviceSynchronize();
    g_combining.store(on ? 1 : 0, std::memory_order_release);
    utils::MemoryPool::disown_all_pools();    // user-created pools too: their blocks carry the shared tag / per-thread tags of the mode that ends
}
}  // namespace detail

// call combining is a one-device mode: a thread whose current device is not the shared stream's keeps its own stream (and its own owner tag)
static inline bool on_shared_device() {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess && dev == detail::g_shared_device;
}

// Host threads are mapped onto a BOUNDED set of streams per device.  Default: 16 streams while at most 16 host threads use the library, 8 once there are
// more (a thread re-reads the number only when its stream has just drained, see stream_wait()); TROY_STREAMS=<1..16> fixes the number, TROY_STREAMS=per-thread
// keeps one stream per host thread (hipStreamPerThread, the mapping of rounds 1-5); read once.  The reference tool's -c N mode
// (test/bench/he_operations.cu:135-147: N host threads of single-object calls) lost 60-75 % of its throughput between 16 and 64 threads with one stream per
// thread: 64 streams share the 4 hardware queues, and every switch of a queue between streams costs a barrier packet and a signal round trip.  With Q streams
// the calls of the threads that share a stream simply queue behind one another.  Measured (profiles/r06_streams_ab.txt, he_bench_driver, CKKS N = 16384 6 x 50-bit,
// three calls / fused k ops/s): per-thread 34 / 31 at 16 threads and 14 / 24 at 64; Q = 4: 25 / 35 flat; Q = 8: 31 / 44 and 22 / 42; Q = 16: 35 / 50 and 21 / 27.
// The streams are BLOCKING streams (hipStreamCreate): a caller that still
// writes hipStreamSynchronize(0) / hipDeviceSynchronize() -- the reference's idiom without --default-stream per-thread -- waits for them too;
// utils::stream_sync() (utils/memory_pool.h:37) waits for exactly the calling thread's stream.
namespace detail {
constexpr int MAX_POOL_DEVICES = 16, MAX_POOL_STREAMS = 16, ADAPTIVE_MANY_THREADS = 17, ADAPTIVE_FEW = 16, ADAPTIVE_MANY = 8;
static int stream_pool_size() {      // 0 = one stream per host thread, -1 = by the number of host threads (the default)
    static const int q = [] {
        const char* e = std::getenv("TROY_STREAMS");
        if (!e || !*e) return -1;
        if (std::string(e) == "per-thread") return 0;
        char* end = nullptr;
        const long v = std::strtol(e, &end, 10);
        if (end == e || *end || v < 1 || v > MAX_POOL_STREAMS) throw std::invalid_argument("[troy] TROY_STREAMS must be 1..16 or per-thread");
        return (int)v;
    }();
    return q;
}
struct DeviceStreams { std::once_flag once; hipStream_t s[MAX_POOL_STREAMS] = {}; };
static DeviceStreams& device_streams(int dev) { static DeviceStreams* all = new DeviceStreams[MAX_POOL_DEVICES]; return all[dev]; }   // never destroyed (threads may outlive statics)
static std::atomic<unsigned> g_next_slot{0};
static std::atomic<int> g_live_threads{0};          // host threads that have used the library and have not ended
struct ThreadSlot {
    unsigned slot; int q = 0;                        // q: the stream count this thread currently maps by (adaptive mode)
    unsigned busy_devices = 0;                       // bit d: this thread may have work in flight on device d (it took that device's stream since its last wait there)
    ThreadSlot() : slot(g_next_slot.fetch_add(1, std::memory_order_relaxed)) { g_live_threads.fetch_add(1, std::memory_order_relaxed); }
    ~ThreadSlot() { g_live_threads.fetch_sub(1, std::memory_order_relaxed); }
};
static inline ThreadSlot& this_thread_slot() { thread_local ThreadSlot ts; return ts; }   // the slot is fixed for the life of the host thread
static inline int adaptive_streams() { return g_live_threads.load(std::memory_order_relaxed) <= ADAPTIVE_MANY_THREADS ? ADAPTIVE_FEW : ADAPTIVE_MANY; }
// A thread may move to another stream only while nothing of its own is in flight: right after a wait for its stream (stream_wait) -- everything it
// queued has completed, so what it releases later is safely tagged with the new stream.
static inline void stream_drained() {
    if (stream_pool_size() >= 0) return;
    ThreadSlot& ts = this_thread_slot();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_POOL_DEVICES) return;
    ts.busy_devices &= ~(1u << dev);
    if (ts.busy_devices == 0) ts.q = adaptive_streams();     // (a thread that drives several devices moves only when all of its streams have drained)
}
// null: no pool on this device (creation failed or the device index is out of range) -> hipStreamPerThread
static inline hipStream_t pooled_stream(int q, int* slot_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_POOL_DEVICES) return nullptr;
    DeviceStreams& d = device_streams(dev);
    std::call_once(d.once, [&] {
        const int create = q < 0 ? ADAPTIVE_FEW : q;
        for (int i = 0; i < create; i++) if (hipStreamCreate(&d.s[i]) != hipSuccess) { (void)hipGetLastError(); d.s[i] = nullptr; }
    });
    ThreadSlot& ts = this_thread_slot();
    if (q < 0) { if (!ts.q) ts.q = adaptive_streams(); q = ts.q; ts.busy_devices |= 1u << dev; }
    const int slot = (int)(ts.slot % (unsigned)q);
    if (slot_out) *slot_out = dev * MAX_POOL_STREAMS + slot;
    return d.s[slot];
}
}  // namespace detail

// The threads that share a stream enter the library one at a time: a host mutex per stream around the launches of one multiply / relinearize / rescale /
// fused call.  Without it they meet on the runtime's own per-stream lock launch by launch; with it 64 host threads run 26-29 k three-call / 43 k fused
// ops/s instead of 22-25 k / 35-41 k (same session, profiles/r06_streams_ab.txt); neutral at 4 and 16 threads.  TROY_STREAM_GATE=0 switches it off.
namespace detail {
static bool stream_gate_on() { static const bool on = [] { const char* e = std::getenv("TROY_STREAM_GATE"); return !(e && e[0] == '0'); }(); return on; }
static std::mutex* stream_gates() { static std::mutex* g = new std::mutex[MAX_POOL_DEVICES * MAX_POOL_STREAMS]; return g; }
struct LaunchGate {
    std::mutex* m = nullptr;
    LaunchGate() {
        if (!stream_gate_on() || combining_on()) return;
        const int q = stream_pool_size();
        int slot = 0;
        if (q != 0 && pooled_stream(q, &slot)) { m = &stream_gates()[slot]; m->lock(); }
    }
    ~LaunchGate() { if (m) m->unlock(); }
};
}  // namespace detail

static inline hipStream_t current_stream() {
    if (detail::combining_on()) if (hipStream_t s = detail::shared_stream()) if (on_shared_device()) return s;
    if (const int q = detail::stream_pool_size()) if (hipStream_t s = detail::pooled_stream(q)) return s;
    return hipStreamPerThread;
}

namespace detail { bool on_combining_stream() { return combining_on() && g_shared_stream && current_stream() == g_shared_stream; } }

void troyn_check_public(int rc) { troyn_check(rc); }
troyn_stream_t troyn_current_stream() { return (troyn_stream_t)current_stream(); }
// every wait for the calling thread's stream goes through here: waits for the shared stream of call combining are grouped (combine.cpp)
static hipError_t stream_wait() {
    hipStream_t s = current_stream();
    if (detail::combining_on() && s == detail::g_shared_stream) return static_cast<hipError_t>(detail::combining_stream_wait(s));
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) detail::stream_drained();
    return e;
}
void troyn_sync_current_stream() { hip_check(stream_wait(), "stream_sync"); }
namespace utils { void stream_sync() { troyn_sync_current_stream(); } }   // utils/memory_pool.h:37

namespace utils {

void slice_bytes_to_host(const void* src, bool src_on_device, void* dst, size_t bytes) {
    if (!src_on_device) { std::memcpy(dst, src, bytes); return; }
    hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, current_stream()), "copy_device_to_host");
    hip_check(stream_wait(), "copy_device_to_host");
}
void host_bytes_to_slice(void* dst, bool dst_on_device, const void* src, size_t bytes) {
    if (!dst_on_device) { std::memcpy(dst, src, bytes); return; }
    hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, current_stream()), "copy_host_to_device");
    hip_check(stream_wait(), "copy_host_to_device");       // `src` is the caller's (often a temporary)
}

std::shared_ptr<void> device_bytes_allocate(size_t bytes, std::shared_ptr<MemoryPool> pool, bool zero) {
    // the block is a DynamicArray of whole words (pool memory, returned to the pool in stream order); the shared_ptr's control block owns it
    auto block = std::make_shared<DynamicArray>((bytes + 7) / 8, true, pool ? pool : MemoryPool::GlobalPool());
    if (zero) block->set_zero();
    return std::shared_ptr<void>(block, static_cast<void*>(block->raw_pointer()));
}
void device_bytes_copy(void* dst, const void* src, size_t bytes) {
    if (bytes) hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, current_stream()), "copy_device_to_device");
}

size_t device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;   // memory_pool.h:14-33: no usable device -> 0
    return n < 0 ? 0 : static_cast<size_t>(n);
}

static std::mutex g_global_pool_mutex;
static MemoryPoolHandle g_global_pool;

// every live pool (call combining disowns all of them when it is switched); leaked on purpose like the tables below
static std::mutex& g_pools_mutex() { static std::mutex* m = new std::mutex; return *m; }
static std::unordered_set<MemoryPool*>& g_pools() { static auto* s = new std::unordered_set<MemoryPool*>; return *s; }

static std::atomic<uint64_t> g_pool_ids{1};
MemoryPool::MemoryPool(size_t device) : device_(device), id_(g_pool_ids.fetch_add(1)) {
    if (device >= device_count()) throw std::runtime_error("[MemoryPool::MemoryPool] No such device.");
    for (auto& sp : stream_shards_) sp.store(nullptr, std::memory_order_relaxed);
    if (const char* e = std::getenv("TROY_POOL_HIGH_WATER_MB")) high_water_ = static_cast<size_t>(std::strtoull(e, nullptr, 0)) << 20;
    std::lock_guard<std::mutex> lock(g_pools_mutex());
    g_pools().insert(this);
}

// A block in a free list carries the tag of the host thread that released it: kernels queued on THAT thread's stream may still be using
// it.  Tag 0 = nobody's work is pending on it (it was in the free list at a device-wide synchronisation).  A thread that ends leaves its
// tag in g_dead_tags (its destructor must not call into HIP: the runtime's own thread-local state may be gone by then); blocks of dead
// threads are handed out after ONE device-wide synchronisation, which clears every tag.
// (leaked on purpose: a host thread may end after the statics of this library have been destroyed)
static std::mutex& g_dead_mutex_ref() { static std::mutex* m = new std::mutex; return *m; }
static std::unordered_set<uint64_t>& g_dead_tags_ref() { static auto* s = new std::unordered_set<uint64_t>; return *s; }
#define g_dead_mutex g_dead_mutex_ref()
#define g_dead_tags g_dead_tags_ref()
struct ThreadExit {
    uint64_t tag = 0;
    ~ThreadExit() {
        if (!tag) return;
        std::lock_guard<std::mutex> lock(g_dead_mutex);
        g_dead_tags.insert(tag);
    }
};
static uint64_t this_thread_tag() {
    // one shared stream (call combining): release and reuse are ordered by that stream whichever host thread does them
    if (detail::combining_on() && detail::shared_stream() && on_shared_device()) return uint64_t(1) << 63;
    // a pooled stream: the tag is the STREAM's (bit 62 + device / slot), so a block released by one of the threads that share it is reusable by
    // any of them at once -- release and reuse are ordered by that stream.  These tags never die.
    if (const int q = detail::stream_pool_size()) {
        int slot = 0;
        if (detail::pooled_stream(q, &slot)) return (uint64_t(1) << 62) | (uint64_t)slot;
    }
    static std::atomic<uint64_t> next{1};
    thread_local ThreadExit te;
    if (!te.tag) te.tag = next.fetch_add(1);
    return te.tag;
}
static bool tag_is_dead(uint64_t tag) {
    std::lock_guard<std::mutex> lock(g_dead_mutex);
    return g_dead_tags.count(tag) != 0;
}

static constexpr uint64_t STREAM_TAG_BIT = uint64_t(1) << 62;

MemoryPool::Shard* MemoryPool::shard_of(uint64_t tag) {
    if (tag == 0) return &nobody_;
    if ((tag & STREAM_TAG_BIT) && (tag & ~STREAM_TAG_BIT) < STREAM_SHARDS) {
        std::atomic<Shard*>& slot = stream_shards_[tag & ~STREAM_TAG_BIT];
        Shard* s = slot.load(std::memory_order_acquire);
        if (s) return s;
        Shard* fresh = new Shard;
        if (slot.compare_exchange_strong(s, fresh, std::memory_order_acq_rel)) return fresh;
        delete fresh;
        return s;
    }
    // per-thread tags (TROY_STREAMS=per-thread) and the combining tag: a table under its own lock, with this thread's last answer cached
    struct Cached { uint64_t pool_id = 0, tag = 0; Shard* s = nullptr; };
    thread_local Cached cached;
    if (cached.pool_id == id_ && cached.tag == tag) return cached.s;
    std::lock_guard<std::mutex> lock(table_mutex_);
    std::unique_ptr<Shard>& sp = other_shards_[tag];
    if (!sp) sp.reset(new Shard);
    cached.pool_id = id_; cached.tag = tag; cached.s = sp.get();      // (a shard is removed only once its thread has ended: nobody caches that tag any more)
    return sp.get();
}

void* MemoryPool::take_from(Shard& sh, size_t bytes) {
    void* found = nullptr;
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lock(sh.m);
        auto it = sh.free_.lower_bound(bytes);
        if (it == sh.free_.end() || it->first > bytes * 2) return nullptr;
        found = it->second.back().ptr;
        sz = it->first;
        it->second.pop_back();
        if (it->second.empty()) sh.free_.erase(it);
    }
    LiveShard& ls = live_shard(found);
    std::lock_guard<std::mutex> lock(ls.m);
    ls.map[found] = sz;
    return found;
}

bool MemoryPool::dead_owner_has(size_t bytes) {
    { std::lock_guard<std::mutex> dl(g_dead_mutex); if (g_dead_tags.empty()) return false; }
    std::lock_guard<std::mutex> lock(table_mutex_);
    for (auto& kv : other_shards_) {
        if (!tag_is_dead(kv.first)) continue;
        std::lock_guard<std::mutex> sl(kv.second->m);
        auto it = kv.second->free_.lower_bound(bytes);
        if (it != kv.second->free_.end() && it->first <= bytes * 2) return true;
    }
    return false;
}

// the blocks released under `tag` (~0: under any tag) at or before release number `upto` move to shard 0: nobody has work pending on them any more
void MemoryPool::disown(uint64_t tag, uint64_t upto) {
    std::vector<std::pair<size_t, FreeBlock>> moved;
    auto drain = [&](Shard& sh) {
        std::lock_guard<std::mutex> lock(sh.m);
        for (auto it = sh.free_.begin(); it != sh.free_.end();) {
            auto& v = it->second;
            for (size_t i = v.size(); i-- > 0;)
                if (v[i].seq <= upto) { moved.emplace_back(it->first, v[i]); v.erase(v.begin() + static_cast<std::ptrdiff_t>(i)); }
            it = v.empty() ? sh.free_.erase(it) : std::next(it);
        }
    };
    if (tag == ~uint64_t(0)) {
        for (auto& sp : stream_shards_) if (Shard* sh = sp.load(std::memory_order_acquire)) drain(*sh);
        std::lock_guard<std::mutex> lock(table_mutex_);
        for (auto it = other_shards_.begin(); it != other_shards_.end();) {
            drain(*it->second);
            bool empty;
            { std::lock_guard<std::mutex> sl(it->second->m); empty = it->second->free_.empty(); }
            // the shard of a host thread that has ended is dropped once it is empty (nothing refers to that tag any more)
            it = (empty && tag_is_dead(it->first)) ? other_shards_.erase(it) : std::next(it);
        }
    } else if (tag != 0) {
        drain(*shard_of(tag));
    }
    if (moved.empty()) return;
    std::lock_guard<std::mutex> lock(nobody_.m);
    for (auto& kv : moved) { kv.second.owner = 0; nobody_.free_[kv.first].push_back(kv.second); }
}
void MemoryPool::disown_all_pools() {
    std::lock_guard<std::mutex> lock(g_pools_mutex());
    for (MemoryPool* p : g_pools()) p->disown(~uint64_t(0));
}
void MemoryPool::set_high_water_bytes(size_t bytes) { high_water_.store(bytes, std::memory_order_relaxed); }
size_t MemoryPool::held_bytes() { return held_bytes_.load(std::memory_order_relaxed); }

MemoryPool::~MemoryPool() {
    { std::lock_guard<std::mutex> lock(g_pools_mutex()); g_pools().erase(this); }
    auto free_all = [](Shard& sh) { for (auto& kv : sh.free_) for (auto& blk : kv.second) (void)hipFree(blk.ptr); };
    free_all(nobody_);
    for (auto& sp : stream_shards_) if (Shard* sh = sp.load()) { free_all(*sh); delete sh; }
    for (auto& kv : other_shards_) free_all(*kv.second);
    for (auto& ls : live_) for (auto& kv : ls.map) (void)hipFree(kv.first);
}

MemoryPoolHandle MemoryPool::GlobalPool() {
    std::lock_guard<std::mutex> lock(g_global_pool_mutex);
    if (!g_global_pool) g_global_pool = std::make_shared<MemoryPool>(0);
    return g_global_pool;
}

void MemoryPool::Destroy() {
    std::lock_guard<std::mutex> lock(g_global_pool_mutex);
    g_global_pool.reset();
}

// Best fit with at most 2x slack (memory_pool_safe.in:119-148).  Order of preference: a block released on THIS thread's stream (safe by stream
// order) or one nobody has work pending on (shard 0) -- the fast path, one shard lock each; then, one thread at a time: a block of a host thread
// that has ended, behind one device-wide synchronisation (which moves every block released before it to shard 0, so this happens once per
// generation of threads); a fresh hipMalloc; and only when the device is out of memory (or the pool is above its high-water mark) a block another
// LIVE stream released, again behind a device-wide synchronisation.  Round 3 took any foreign block before trying hipMalloc: with N host threads
// working on single objects (the reference's -c N mode) blocks migrated between threads all the time and every migration was a
// hipDeviceSynchronize -- 16 threads ran at the speed of 3 (tests/cpp/he_bench_driver threads).
static std::atomic<uint64_t> g_pool_mallocs{0};
uint64_t MemoryPool::device_allocations() { return g_pool_mallocs.load(std::memory_order_relaxed); }

void MemoryPool::set_device() { hip_check(hipSetDevice(static_cast<int>(device_)), "set_device"); }

void MemoryPool::destroy() {
    // memory_pool_safe.in:168-205: cached blocks and blocks still handed out are freed (the latter become dangling in their owners, as in the reference)
    release_unused();
    (void)hipDeviceSynchronize();
    for (auto& ls : live_) {
        std::lock_guard<std::mutex> lock(ls.m);
        for (auto& kv : ls.map) { (void)hipFree(kv.first); held_bytes_.fetch_sub(std::min(kv.second, held_bytes_.load()), std::memory_order_relaxed); }
        ls.map.clear();
    }
}

void* MemoryPool::allocate(size_t bytes) {
    if (denying_.load(std::memory_order_relaxed)) throw std::runtime_error("[MemoryPool(safe)::get] DEBUG: The pool is denying allocation.");
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~size_t(255);
    Shard& mine = *shard_of(this_thread_tag());
    if (void* p = take_from(mine, bytes)) return p;
    if (void* p = take_from(nobody_, bytes)) return p;
    std::lock_guard<std::mutex> slow(slow_mutex_);
    if (void* p = take_from(nobody_, bytes)) return p;       // another thread's device-wide wait may have filled shard 0 meanwhile
    hip_check(hipSetDevice(static_cast<int>(device_)), "malloc");
    // A device-wide synchronisation covers what was queued BEFORE it: the blocks in the free lists when it STARTS are then anybody's.  A block
    // another thread releases while it drains (or after) may still have that thread's kernels pending and must keep its owner, so blocks move to
    // shard 0 only up to the release number read before the wait (round 4 cleared every tag afterwards: a block released during the wait could be
    // handed to a different stream with work still pending on it).
    auto sync_and_disown = [&] {
        const uint64_t mark = release_seq_.load(std::memory_order_acquire);
        hip_check(hipDeviceSynchronize(), "device_synchronize");
        disown(~uint64_t(0), mark);
    };
    if (dead_owner_has(bytes)) {
        sync_and_disown();
        if (void* q = take_from(nobody_, bytes)) return q;
    }
    const size_t cap = high_water_.load(std::memory_order_relaxed);
    if (cap != 0 && held_bytes_.load(std::memory_order_relaxed) + bytes > cap) {
        // above the high-water mark: reuse what other streams have released (one device-wide wait) before growing any further
        sync_and_disown();
        if (void* q = take_from(nobody_, bytes)) return q;
    }
    void* p = nullptr;
    g_pool_mallocs.fetch_add(1, std::memory_order_relaxed);
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        sync_and_disown();
        if (void* q = take_from(nobody_, bytes)) return q;      // only blocks the wait covered
        release_unused();      // nothing of a fitting size: give the cache back and retry once
        hip_check(hipMalloc(&p, bytes), "malloc");
    }
    {
        LiveShard& ls = live_shard(p);
        std::lock_guard<std::mutex> lock(ls.m);
        ls.map[p] = bytes;
    }
    held_bytes_.fetch_add(bytes, std::memory_order_relaxed);
    return p;
}

void MemoryPool::release(void* ptr) {
    if (!ptr) return;
    size_t sz;
    {
        LiveShard& ls = live_shard(ptr);
        std::lock_guard<std::mutex> lock(ls.m);
        auto it = ls.map.find(ptr);
        if (it == ls.map.end()) return;
        sz = it->second;
        ls.map.erase(it);
    }
    const uint64_t me = this_thread_tag();
    Shard& sh = *shard_of(me);
    std::lock_guard<std::mutex> lock(sh.m);
    sh.free_[sz].push_back(FreeBlock{ptr, me, release_seq_.fetch_add(1, std::memory_order_acq_rel) + 1});
}

void MemoryPool::release_unused() {
    std::vector<void*> drop;
    size_t bytes = 0;
    auto take_all = [&](Shard& sh) {
        std::lock_guard<std::mutex> lock(sh.m);
        for (auto& kv : sh.free_) { for (auto& blk : kv.second) drop.push_back(blk.ptr); bytes += kv.first * kv.second.size(); }
        sh.free_.clear();
    };
    take_all(nobody_);
    for (auto& sp : stream_shards_) if (Shard* sh = sp.load(std::memory_order_acquire)) take_all(*sh);
    { std::lock_guard<std::mutex> lock(table_mutex_); for (auto& kv : other_shards_) take_all(*kv.second); }
    if (drop.empty()) return;
    held_bytes_.fetch_sub(std::min(bytes, held_bytes_.load(std::memory_order_relaxed)), std::memory_order_relaxed);
    (void)hipDeviceSynchronize();   // queued kernels may still read blocks released a moment ago
    for (void* q : drop) (void)hipFree(q);
}

DynamicArray::DynamicArray(size_t count, bool device, MemoryPoolHandle pool) : size_(count), device_(device) {
    if (device) {
        pool_ = pool ? pool : MemoryPool::GlobalPool();
        data_ = static_cast<uint64_t*>(pool_->allocate(count * sizeof(uint64_t)));
    } else {
        data_ = count ? static_cast<uint64_t*>(std::malloc(count * sizeof(uint64_t))) : nullptr;
        if (count && !data_) throw std::bad_alloc();
    }
}

DynamicArray DynamicArray::device_view(uint64_t* ptr, size_t count, std::shared_ptr<DynamicArray> owner) {
    DynamicArray a;
    a.data_ = ptr; a.size_ = count; a.device_ = true; a.pool_ = owner->pool_; a.owner_ = std::move(owner);
    return a;
}

void DynamicArray::free_() {
    if (!data_) return;
    if (owner_) { owner_.reset(); data_ = nullptr; size_ = 0; return; }
    if (device_) {
        // the pool hands the block to the next allocation, which is ordered on the same per-thread stream
        pool_->release(data_);
    } else {
        std::free(data_);
    }
    data_ = nullptr; size_ = 0;
}

DynamicArray::~DynamicArray() { free_(); }

DynamicArray::DynamicArray(const DynamicArray& o) : DynamicArray(o.size_, o.device_, o.pool_) {
    if (size_) copy_from(o.data_, size_, o.device_);
}

DynamicArray::DynamicArray(DynamicArray&& o) noexcept
    : data_(o.data_), size_(o.size_), device_(o.device_), pool_(std::move(o.pool_)), owner_(std::move(o.owner_)) {
    o.data_ = nullptr; o.size_ = 0;
}

DynamicArray& DynamicArray::operator=(const DynamicArray& o) {
    if (this == &o) return *this;
    DynamicArray tmp(o);
    *this = std::move(tmp);
    return *this;
}

DynamicArray& DynamicArray::operator=(DynamicArray&& o) noexcept {
    if (this == &o) return *this;
    free_();
    data_ = o.data_; size_ = o.size_; device_ = o.device_; pool_ = std::move(o.pool_); owner_ = std::move(o.owner_);
    o.data_ = nullptr; o.size_ = 0;
    return *this;
}

DynamicArray DynamicArray::from_vector(const std::vector<uint64_t>& v) {
    DynamicArray a(v.size(), false);
    if (!v.empty()) std::memcpy(a.data_, v.data(), v.size() * sizeof(uint64_t));
    return a;
}

std::vector<uint64_t> DynamicArray::to_vector() const {
    std::vector<uint64_t> v(size_);
    if (!size_) return v;
    if (device_) {
        hip_check(hipSetDevice(static_cast<int>(device_index())), "copy_device_to_host");
        hip_check(hipMemcpyAsync(v.data(), data_, size_ * sizeof(uint64_t), hipMemcpyDeviceToHost, current_stream()), "copy_device_to_host");
        hip_check(stream_wait(), "copy_device_to_host");
    } else {
        std::memcpy(v.data(), data_, size_ * sizeof(uint64_t));
    }
    return v;
}

DynamicArray DynamicArray::clone(MemoryPoolHandle pool) const {
    DynamicArray a(size_, device_, pool ? pool : pool_);
    if (size_) a.copy_from(data_, size_, device_);
    return a;
}

void DynamicArray::copy_from(const uint64_t* src, size_t count, bool src_on_device) {
    if (count > size_) throw std::invalid_argument("[DynamicArray::copy_from] source is larger than the array");
    if (!count) return;
    const size_t bytes = count * sizeof(uint64_t);
    if (!device_ && !src_on_device) { std::memcpy(data_, src, bytes); return; }
    hipMemcpyKind kind = device_ ? (src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) : hipMemcpyDeviceToHost;
    hip_check(hipMemcpyAsync(data_, src, bytes, kind, current_stream()), "copy");
    // host buffers may be freed or reused by the caller right away: finish the transfer (reference: D2H is synchronous)
    if (kind != hipMemcpyDeviceToDevice) hip_check(stream_wait(), "copy");
}

// utils/box.h:282-308 for the uint64_t views the public accessors hand out
template <> void Slice<uint64_t>::set_zero() const {
    if (!len_) return;
    if (device_) hip_check(hipMemsetAsync(ptr_, 0, len_ * sizeof(uint64_t), current_stream()), "memset");
    else std::memset(ptr_, 0, len_ * sizeof(uint64_t));
}
template <> void Slice<uint64_t>::copy_from_slice(ConstSlice<uint64_t> source) const {
    if (source.size() != len_) throw std::runtime_error("[Slice::copy_from_slice] Slice size does not match array size");
    if (!len_) return;
    const size_t bytes = len_ * sizeof(uint64_t);
    if (!device_ && !source.on_device()) { std::memcpy(ptr_, source.raw_pointer(), bytes); return; }
    const hipMemcpyKind kind = device_ ? (source.on_device() ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) : hipMemcpyDeviceToHost;
    hip_check(hipMemcpyAsync(ptr_, source.raw_pointer(), bytes, kind, current_stream()), "copy");
    if (kind != hipMemcpyDeviceToDevice) hip_check(stream_wait(), "copy");
}

void DynamicArray::set_zero() {
    if (!size_) return;
    if (device_) hip_check(hipMemsetAsync(data_, 0, size_ * sizeof(uint64_t), current_stream()), "memset");
    else std::memset(data_, 0, size_ * sizeof(uint64_t));
}

static void resize_array(DynamicArray& a, size_t count, bool zero_rest, bool copy_data, size_t old_size, bool device, MemoryPoolHandle pool) {
    if (count == old_size) return;
    DynamicArray n(count, device, pool);
    const size_t kept = (copy_data && old_size && count) ? std::min(count, old_size) : 0;
    if (kept) n.copy_from(a.raw_pointer(), kept, device);
    if (zero_rest && count > kept) {
        if (device) hip_check(hipMemsetAsync(n.raw_pointer() + kept, 0, (count - kept) * sizeof(uint64_t), current_stream()), "memset");
        else std::memset(n.raw_pointer() + kept, 0, (count - kept) * sizeof(uint64_t));
    }
    a = std::move(n);
}
void DynamicArray::resize(size_t count, bool copy_data) { resize_array(*this, count, true, copy_data, size_, device_, pool_); }
void DynamicArray::resize_uninitialized(size_t count, bool copy_data) { resize_array(*this, count, false, copy_data, size_, device_, pool_); }

void DynamicArray::to_device_inplace(MemoryPoolHandle pool) {
    if (device_) return;
    DynamicArray d(size_, true, pool ? pool : MemoryPool::GlobalPool());
    if (size_) d.copy_from(data_, size_, false);
    *this = std::move(d);
}

void DynamicArray::to_host_inplace() {
    if (!device_) return;
    DynamicArray h(size_, false);
    if (size_) h.copy_from(data_, size_, true);
    *this = std::move(h);
}

}  // namespace utils

// ------------------------------------------------------------------------------------------------
// Modulus / CoeffModulus / PlainModulus
// ------------------------------------------------------------------------------------------------
static bool is_prime_u64(uint64_t n) {
    if (n < 2) return false;
    auto mulmod = [](uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)(((unsigned __int128)a * b) % m); };
    auto powmod = [&](uint64_t a, uint64_t e, uint64_t m) { uint64_t r = 1; a %= m; while (e) { if (e & 1) r = mulmod(r, a, m); a = mulmod(a, a, m); e >>= 1; } return r; };
    for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (n == p) return true; if (n % p == 0) return false; }
    uint64_t d = n - 1; int r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < r; i++) { x = mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}

Modulus::Modulus(uint64_t value) {
    // modulus.cu:7-32
    if (value == 0) return;
    if ((value >> 61) != 0 || value == 1) throw std::invalid_argument("[Modulus::set_value] Value can be at most 61-bit and cannot be 1.");
    value_ = value;
    for (uint64_t v = value; v; v >>= 1) bit_count_++;
    unsigned __int128 two64 = (unsigned __int128)1 << 64;
    unsigned __int128 hi = two64 / value, r = two64 % value;
    const_ratio_[0] = (uint64_t)((r << 64) / value);
    const_ratio_[1] = (uint64_t)hi;
    const_ratio_[2] = (uint64_t)((r << 64) % value);
    is_prime_ = is_prime_u64(value);
}

uint64_t Modulus::reduce(uint64_t input) const {
    uint64_t t = input - (uint64_t)(((unsigned __int128)input * const_ratio_[1]) >> 64) * value_;
    return t >= value_ ? t - value_ : t;
}

size_t CoeffModulus::max_bit_count(size_t n, SecurityLevel sec) {
    // utils/he_standard_params.h (HomomorphicEncryption.org tables)
    static const std::map<size_t, std::vector<size_t>> table = {
        {1024, {27, 19, 14}}, {2048, {54, 37, 29}}, {4096, {109, 75, 58}}, {8192, {218, 152, 118}}, {16384, {438, 305, 237}}, {32768, {881, 611, 476}}};
    if (sec == SecurityLevel::Nil) return static_cast<size_t>(-1) >> 1;
    auto it = table.find(n);
    if (it == table.end()) return 0;
    return it->second[static_cast<size_t>(sec) - 1];
}

std::vector<Modulus> CoeffModulus::bfv_default_vector(size_t poly_modulus_degree, SecurityLevel sec_level) {
    // coeff_modulus.cu:6-63.  The chains are SEAL's published defaults (data, not logic): rows = (security level, degree).
    struct Row { SecurityLevel sec; size_t n; std::vector<uint64_t> q; };
    static const std::vector<Row> table = {
        {SecurityLevel::Classical128, 1024, {0x7e00001}},
        {SecurityLevel::Classical128, 2048, {0x3fffffff000001}},
        {SecurityLevel::Classical128, 4096, {0xffffee001, 0xffffc4001, 0x1ffffe0001}},
        {SecurityLevel::Classical128, 8192, {0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001}},
        {SecurityLevel::Classical128, 16384, {0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001,
                                              0x1ffffffe88001, 0x1ffffffe48001}},
        {SecurityLevel::Classical128, 32768, {0x7fffffffe90001, 0x7fffffffbf0001, 0x7fffffffbd0001, 0x7fffffffba0001, 0x7fffffffaa0001, 0x7fffffffa50001,
                                              0x7fffffff9f0001, 0x7fffffff7e0001, 0x7fffffff770001, 0x7fffffff380001, 0x7fffffff330001, 0x7fffffff2d0001,
                                              0x7fffffff170001, 0x7fffffff150001, 0x7ffffffef00001, 0xfffffffff70001}},
        {SecurityLevel::Classical192, 1024, {0x7f001}},
        {SecurityLevel::Classical192, 2048, {0x1ffffc0001}},
        {SecurityLevel::Classical192, 4096, {0x1ffc001, 0x1fce001, 0x1fc0001}},
        {SecurityLevel::Classical192, 8192, {0x3ffffac001, 0x3ffff54001, 0x3ffff48001, 0x3ffff28001}},
        {SecurityLevel::Classical192, 16384, {0x3ffffffdf0001, 0x3ffffffd48001, 0x3ffffffd20001, 0x3ffffffd18001, 0x3ffffffcd0001, 0x3ffffffc70001}},
        {SecurityLevel::Classical192, 32768, {0x3fffffffd60001, 0x3fffffffca0001, 0x3fffffff6d0001, 0x3fffffff5d0001, 0x3fffffff550001, 0x7fffffffe90001,
                                              0x7fffffffbf0001, 0x7fffffffbd0001, 0x7fffffffba0001, 0x7fffffffaa0001, 0x7fffffffa50001}},
        {SecurityLevel::Classical256, 1024, {0x3001}},
        {SecurityLevel::Classical256, 2048, {0x1ffc0001}},
        {SecurityLevel::Classical256, 4096, {0x3ffffffff040001}},
        {SecurityLevel::Classical256, 8192, {0x7ffffec001, 0x7ffffb0001, 0xfffffdc001}},
        {SecurityLevel::Classical256, 16384, {0x7ffffffc8001, 0x7ffffff00001, 0x7fffffe70001, 0xfffffffd8001, 0xfffffffa0001}},
        {SecurityLevel::Classical256, 32768, {0xffffffff00001, 0x1fffffffe30001, 0x1fffffffd80001, 0x1fffffffd10001, 0x1fffffffc50001, 0x1fffffffbf0001,
                                              0x1fffffffb90001, 0x1fffffffb60001, 0x1fffffffa50001}},
    };
    if (sec_level == SecurityLevel::Nil) throw std::invalid_argument("[CoeffModulus::bfv_default_vector] No default for Nil security.");
    for (const Row& r : table)
        if (r.sec == sec_level && r.n == poly_modulus_degree) {
            std::vector<Modulus> out;
            for (uint64_t v : r.q) out.emplace_back(v);
            return out;
        }
    throw std::invalid_argument("[CoeffModulus::bfv_default_vector] Invalid poly_modulus_degree or sec_level.");
}

std::vector<Modulus> CoeffModulus::create_vector(size_t poly_modulus_degree, std::vector<size_t> bit_sizes) {
    // coeff_modulus.cu:65-108 -- prime search lives in libtroyn's host helpers
    std::vector<uint64_t> out(bit_sizes.size());
    int rc = troyn_coeff_modulus_create(poly_modulus_degree, bit_sizes.data(), bit_sizes.size(), out.data());
    if (rc != 0) throw std::invalid_argument(troyn_last_error());
    std::vector<Modulus> res;
    for (uint64_t q : out) res.emplace_back(q);
    return res;
}

Modulus PlainModulus::batching(size_t poly_modulus_degree, size_t bit_size) {
    return CoeffModulus::create(poly_modulus_degree, {bit_size})[0];
}

// ------------------------------------------------------------------------------------------------
// EncryptionParameters
// ------------------------------------------------------------------------------------------------
const ParmsID parms_id_zero{};

// BLAKE2b (RFC 7693), unkeyed, digest length `outlen` <= 64.  The reference vendors the BLAKE2 reference code
// (utils/blake2/) and uses its 32-byte digest as ParmsID (utils/hash.h:24-31); this is the published algorithm.
namespace {
inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
void blake2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t0, bool last) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    static const uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) std::memcpy(&m[i], block + 8 * i, 8);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t0;                 // message length < 2^64: the high counter word stays 0
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] = v[a] + v[b] + x; v[d] = rotr64(v[d] ^ v[a], 32);
        v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 24);
        v[a] = v[a] + v[b] + y; v[d] = rotr64(v[d] ^ v[a], 16);
        v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
        const uint8_t* s = SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
}  // namespace

void utils::blake2b(void* out, size_t outlen, const void* in, size_t inlen) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    if (outlen == 0 || outlen > 64) throw std::invalid_argument("[blake2b] invalid digest length");
    uint64_t h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010000ull ^ static_cast<uint64_t>(outlen);
    const uint8_t* p = static_cast<const uint8_t*>(in);
    uint8_t block[128];
    size_t done = 0;
    while (inlen - done > 128) { blake2b_compress(h, p + done, done + 128, false); done += 128; }
    std::memset(block, 0, 128);
    std::memcpy(block, p + done, inlen - done);
    blake2b_compress(h, block, inlen, true);
    std::memcpy(out, h, outlen);
}

void EncryptionParameters::compute_parms_id() {
    // encryption_parameters.cu:8-32: BLAKE2b-256 of the words [scheme, N, q_0..q_{K-1}, t]
    std::vector<uint64_t> words{static_cast<uint64_t>(scheme_), static_cast<uint64_t>(poly_modulus_degree_)};
    for (const Modulus& m : coeff_modulus_) words.push_back(m.value());
    words.push_back(plain_modulus_.value());
    utils::blake2b(parms_id_.v, sizeof(parms_id_.v), words.data(), words.size() * 8);
    if (parms_id_.is_zero()) throw std::logic_error("[EncryptionParameters::compute_parms_id] Computed parms_id is zero");
}

// ------------------------------------------------------------------------------------------------
// HeContext
// ------------------------------------------------------------------------------------------------
// little-endian multi-word helpers of the level constants (utils/uint_small_mod, utils/basics of the reference; a handful of words, host side)
static uint64_t words_mod_u64(const std::vector<uint64_t>& w, uint64_t m) {
    unsigned __int128 r = 0;
    for (size_t i = w.size(); i-- > 0;) r = ((r << 64) | w[i]) % m;
    return static_cast<uint64_t>(r);
}
static uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { const uint64_t r = a % b; a = b; b = r; } return a; }

void ContextData::validate(SecurityLevel sec_level) {
    // context_data.cu:71-345, in its order: the FIRST failing check is the one reported, and the qualifiers keep what was established before it
    using E = EncryptionParameterErrorType;
    EncryptionParameterQualifiers& ql = qualifiers_;
    const EncryptionParameters& p = parms_;
    ql = EncryptionParameterQualifiers();
    ql.parameter_error = E::Success;
    if (p.scheme() == SchemeType::Nil) { ql.parameter_error = E::InvalidScheme; return; }
    const auto& q = p.coeff_modulus();
    const size_t k = q.size();
    if (k > 64 || k < 1) { ql.parameter_error = E::InvalidCoeffModulusSize; return; }
    for (const Modulus& m : q) if ((m.value() >> 60) > 0 || (m.value() >> 1) == 0) { ql.parameter_error = E::InvalidCoeffModulusBitCount; return; }
    {
        std::vector<uint64_t> prod{1};
        for (const Modulus& m : q) {
            uint64_t carry = 0;
            for (uint64_t& w : prod) { const unsigned __int128 v = static_cast<unsigned __int128>(w) * m.value() + carry; w = static_cast<uint64_t>(v); carry = static_cast<uint64_t>(v >> 64); }
            if (carry) prod.push_back(carry);
        }
        size_t bits = (prod.size() - 1) * 64;
        for (uint64_t top = prod.back(); top; top >>= 1) bits++;
        total_coeff_modulus_bit_count_ = bits;
        prod.resize(k, 0);                         // the reference keeps coeff_modulus_size words
        total_coeff_modulus_ = std::move(prod);
    }
    const size_t n = p.poly_modulus_degree();
    if (n < 2 || n > 131072) { ql.parameter_error = E::InvalidPolyModulusDegree; return; }
    if ((n & (n - 1)) != 0) { ql.parameter_error = E::InvalidPolyModulusDegreeNonPowerOfTwo; return; }
    if (k * n > (size_t(1) << 32)) { ql.parameter_error = E::InvalidParametersTooLarge; return; }
    ql.using_fft = true;
    ql.security_level = sec_level;
    if (total_coeff_modulus_bit_count_ > CoeffModulus::max_bit_count(n, sec_level)) {
        ql.security_level = SecurityLevel::Nil;
        if (sec_level != SecurityLevel::Nil) { ql.parameter_error = E::InvalidParametersInsecure; return; }
    }
    // RNSBase: the moduli must be pairwise coprime (utils/rns_base.cu: the punctured products need inverses)
    for (size_t i = 0; i < k; i++) for (size_t j = 0; j < i; j++) if (gcd_u64(q[i].value(), q[j].value()) != 1) { ql.parameter_error = E::FailedCreatingRNSBase; return; }
    // NTT tables: a primitive 2N-th root of unity modulo every q_i
    ql.using_ntt = true;
    for (const Modulus& m : q) if ((m.value() - 1) % (2 * n) != 0 || !m.is_prime()) { ql.using_ntt = false; ql.parameter_error = E::InvalidCoeffModulusNoNTT; return; }
    if (p.scheme() == SchemeType::BFV || p.scheme() == SchemeType::BGV) {
        const uint64_t t = p.plain_modulus().value();
        if ((t >> 60) > 0 || (t >> 1) == 0) { ql.parameter_error = E::InvalidPlainModulusBitCount; return; }
        for (const Modulus& m : q) if (gcd_u64(m.value(), t) != 1) { ql.parameter_error = E::InvalidPlainModulusCoprimality; return; }
        bool t_below_q = false;
        for (size_t i = 1; i < k; i++) t_below_q = t_below_q || total_coeff_modulus_[i] != 0;
        t_below_q = t_below_q || t < total_coeff_modulus_[0];
        if (!t_below_q) { ql.parameter_error = E::InvalidPlainModulusTooLarge; return; }
        ql.using_batching = p.plain_modulus().is_prime() && (t - 1) % (2 * n) == 0;
        ql.using_fast_plain_lift = true;
        for (const Modulus& m : q) if (m.value() <= t) { ql.using_fast_plain_lift = false; break; }
        coeff_modulus_mod_plain_modulus_ = words_mod_u64(total_coeff_modulus_, t);
        upper_half_increment_.resize(k);
        for (size_t i = 0; i < k; i++) upper_half_increment_[i] = coeff_modulus_mod_plain_modulus_ % q[i].value();
        plain_upper_half_threshold_ = (t + 1) >> 1;
        plain_upper_half_increment_.assign(k, 0);
        if (ql.using_fast_plain_lift) {
            for (size_t i = 0; i < k; i++) plain_upper_half_increment_[i] = q[i].value() - t;
        } else {                                   // Q - t as one multi-word integer
            uint64_t borrow = t;
            for (size_t i = 0; i < k; i++) { const uint64_t w = total_coeff_modulus_[i]; plain_upper_half_increment_[i] = w - borrow; borrow = w < borrow ? 1 : 0; }
        }
    } else {
        if (!p.plain_modulus().is_zero()) { ql.parameter_error = E::InvalidPlainModulusNonZero; return; }
        ql.using_batching = true;
        ql.using_fast_plain_lift = false;
        plain_upper_half_threshold_ = 1ull << 63;
        plain_upper_half_increment_.resize(k);
        for (size_t i = 0; i < k; i++) {                                   // (2^63 mod q_i) * (q_i - 2) mod q_i  (= -2^64 mod q_i, context_data.cu:286-293)
            const uint64_t qi = q[i].value();
            plain_upper_half_increment_[i] = static_cast<uint64_t>(static_cast<unsigned __int128>((1ull << 63) % qi) * (qi - 2) % qi);
        }
        upper_half_threshold_ = total_coeff_modulus_;                      // (Q + 1) >> 1
        for (size_t i = 0; i < k; i++) if (++upper_half_threshold_[i] != 0) break;
        for (size_t i = 0; i < k; i++) upper_half_threshold_[i] = (upper_half_threshold_[i] >> 1) | (i + 1 < k ? upper_half_threshold_[i + 1] << 63 : 0);
    }
    ql.using_descending_modulus_chain = true;
    for (size_t i = 1; i < k; i++) if (q[i - 1].value() <= q[i].value()) { ql.using_descending_modulus_chain = false; break; }
}

HeContextPointer HeContext::create(EncryptionParameters parms, bool expand_mod_chain, SecurityLevel sec_level, uint64_t random_seed) {
    // he_context.cu:46-132
    HeContextPointer he(new HeContext());
    he->security_level_ = sec_level;
    auto add = [&](const EncryptionParameters& p) {
        auto cd = std::make_shared<ContextData>();
        cd->parms_ = p;
        cd->validate(sec_level);
        return cd;
    };
    auto key_cd = add(parms);                                            // recorded even when the parameters are refused (he_context.cu:56-66)
    he->map_[parms.parms_id()] = key_cd;
    he->key_parms_id_ = parms.parms_id();
    he->parameters_set_ = key_cd->qualifiers_.parameters_set();
    auto drop_last = [](const EncryptionParameters& p) {
        EncryptionParameters nx = p;
        std::vector<Modulus> q(p.coeff_modulus().begin(), p.coeff_modulus().end() - 1);
        nx.set_coeff_modulus(q);
        return nx;
    };
    std::shared_ptr<ContextData> first_cd = key_cd;
    if (he->parameters_set_ && parms.coeff_modulus().size() > 1 && !parms.use_special_prime_for_encryption()) {
        auto cd = add(drop_last(parms));
        if (cd->qualifiers_.parameters_set()) {
            first_cd = cd;
            he->map_[cd->parms_id()] = cd;
            key_cd->next_ = first_cd;
            first_cd->prev_ = key_cd;
        }
    }
    he->first_parms_id_ = first_cd->parms_id();
    he->using_keyswitching_ = he->first_parms_id_ != he->key_parms_id_;
    std::shared_ptr<ContextData> last_cd = first_cd;
    if (expand_mod_chain && he->parameters_set_) {
        while (last_cd->parms().coeff_modulus().size() > 1) {
            auto cd = add(drop_last(last_cd->parms()));
            if (!cd->qualifiers_.parameters_set()) break;
            he->map_[cd->parms_id()] = cd;
            last_cd->next_ = cd;
            cd->prev_ = last_cd;
            last_cd = cd;
        }
    }
    he->last_parms_id_ = last_cd->parms_id();
    // chain_index: key level highest, last level 0 (he_context.cu:111-123)
    size_t count = he->map_.size();
    for (std::shared_ptr<const ContextData> c = key_cd; c; c = c->next_) {
        std::const_pointer_cast<ContextData>(c)->chain_index_ = --count;
    }
    // he_context.cu:126-131: seed 0 means "seed from the clock"
    if (random_seed == 0) {
        random_seed = static_cast<uint64_t>(std::chrono::system_clock::now().time_since_epoch().count());
        if (random_seed == 0) random_seed = 1;
    }
    he->random_seed_ = random_seed;
    he->random_generator_.reset_seed(random_seed, 0);
    return he;
}

HeContext::~HeContext() {
    for (auto& kv : behz_) troyn_behz_destroy(kv.second);
    for (auto& kv : bgv_) troyn_bgv_destroy(kv.second);
    if (plain_plan_) troyn_plan_destroy(plain_plan_);
    if (plan_) troyn_plan_destroy(plan_);
}

const troyn_plan* HeContext::plain_plan() const {
    // ContextData::plain_ntt_tables (context_data.cu:189-203): one table set modulo t
    std::lock_guard<std::mutex> lock(behz_mutex_);
    if (plain_plan_) return plain_plan_;
    if (!plan_) throw std::invalid_argument("[HeContext::plain_plan] HeContext is not on device (call to_device_inplace).");
    const EncryptionParameters& kp = key_context_data().value()->parms();
    uint32_t log_n = 0;
    while ((size_t(1) << log_n) < kp.poly_modulus_degree()) log_n++;
    const uint64_t t = kp.plain_modulus().value();
    troyn_check(troyn_plan_create(&plain_plan_, static_cast<int>(pool_->get_device()), log_n, 1, &t, nullptr));
    return plain_plan_;
}

void HeContext::to_device_inplace(MemoryPoolHandle pool) {
    if (plan_) return;
    if (!parameters_set_) throw std::invalid_argument("[HeContext::to_device_inplace] Encryption parameters are not valid.");
    pool_ = pool ? pool : MemoryPool::GlobalPool();
    const EncryptionParameters& kp = key_context_data().value()->parms();
    std::vector<uint64_t> q;
    for (const Modulus& m : kp.coeff_modulus()) q.push_back(m.value());
    uint32_t log_n = 0;
    while ((size_t(1) << log_n) < kp.poly_modulus_degree()) log_n++;
    troyn_check(troyn_plan_create(&plan_, static_cast<int>(pool_->get_device()), log_n, static_cast<uint32_t>(q.size()), q.data(), nullptr));
}

const troyn_behz* HeContext::behz(size_t L) const {
    std::lock_guard<std::mutex> lock(behz_mutex_);
    auto it = behz_.find(L);
    if (it != behz_.end()) return it->second;
    troyn_behz* b = nullptr;
    const EncryptionParameters& kp = key_context_data().value()->parms();
    troyn_check(troyn_behz_create(&b, plan_, static_cast<uint32_t>(L), kp.plain_modulus().value()));
    behz_[L] = b;
    return b;
}

const troyn_bgv* HeContext::bgv(size_t L) const {
    std::lock_guard<std::mutex> lock(behz_mutex_);
    auto it = bgv_.find(L);
    if (it != bgv_.end()) return it->second;
    if (!plan_) throw std::invalid_argument("[HeContext::bgv] HeContext is not on device (call to_device_inplace).");
    troyn_bgv* b = nullptr;
    const EncryptionParameters& kp = key_context_data().value()->parms();
    troyn_check(troyn_bgv_create(&b, plan_, static_cast<uint32_t>(L), kp.plain_modulus().value()));
    bgv_[L] = b;
    return b;
}

// ------------------------------------------------------------------------------------------------
// Ciphertext
// ------------------------------------------------------------------------------------------------
Ciphertext Ciphertext::from_members(size_t polynomial_count, size_t coeff_modulus_size, size_t poly_modulus_degree, const ParmsID& parms_id,
                                    double scale, bool is_ntt_form, uint64_t correction_factor, uint64_t seed, utils::DynamicArray&& data) {
    if (data.size() != polynomial_count * coeff_modulus_size * poly_modulus_degree)
        throw std::invalid_argument("[Ciphertext::from_members] data size does not match the shape");
    Ciphertext c;
    c.polynomial_count_ = polynomial_count; c.coeff_modulus_size_ = coeff_modulus_size; c.poly_modulus_degree_ = poly_modulus_degree;
    c.parms_id_ = parms_id; c.scale_ = scale; c.is_ntt_form_ = is_ntt_form; c.correction_factor_ = correction_factor; c.seed_ = seed;
    c.data_ = std::move(data);
    return c;
}

Ciphertext Ciphertext::like(const Ciphertext& o, size_t polynomial_count, size_t coeff_modulus_size, bool fill_zeros, MemoryPoolHandle pool) {
    // ciphertext.cu:5-24
    Ciphertext c;
    c.polynomial_count_ = polynomial_count; c.coeff_modulus_size_ = coeff_modulus_size; c.poly_modulus_degree_ = o.poly_modulus_degree_;
    c.parms_id_ = o.parms_id_; c.scale_ = o.scale_; c.is_ntt_form_ = o.is_ntt_form_; c.correction_factor_ = o.correction_factor_; c.seed_ = 0;
    c.data_ = utils::DynamicArray(polynomial_count * coeff_modulus_size * o.poly_modulus_degree_, o.on_device(), pool);
    if (fill_zeros) c.data_.set_zero();
    return c;
}

Ciphertext Ciphertext::clone(MemoryPoolHandle pool) const {
    Ciphertext c = *this;   // deep copy through DynamicArray's copy constructor
    (void)pool;
    return c;
}

void Ciphertext::resize(const HeContextPointer& context, const ParmsID& parms_id, size_t polynomial_count, bool fill_extra_with_zeros, bool copy_data) {
    if (!context->parameters_set()) throw std::invalid_argument("[Ciphertext::resize] Context is not set correctly.");
    auto cd = context->get_context_data(parms_id);
    if (!cd.has_value()) throw std::invalid_argument("[Ciphertext::resize] ParmsID is not valid.");
    if (polynomial_count < 2 || polynomial_count > 16) throw std::invalid_argument("[Ciphertext::resize_internal] Polynomial count is invalid.");    // HE_CIPHERTEXT_SIZE_MIN / _MAX
    const EncryptionParameters& p = cd.value()->parms();
    parms_id_ = parms_id;
    polynomial_count_ = polynomial_count;
    coeff_modulus_size_ = p.coeff_modulus().size();
    poly_modulus_degree_ = p.poly_modulus_degree();
    const size_t words = polynomial_count_ * coeff_modulus_size_ * poly_modulus_degree_;
    if (fill_extra_with_zeros) data_.resize(words, copy_data); else data_.resize_uninitialized(words, copy_data);
}

void Ciphertext::reconfigure_like(const HeContextPointer& context, const Ciphertext& other, size_t polynomial_count, bool fill_extra_with_zeros) {
    resize(context, other.parms_id(), polynomial_count, fill_extra_with_zeros, true);
    correction_factor_ = other.correction_factor();
    scale_ = other.scale();
    is_ntt_form_ = other.is_ntt_form();
}

// ------------------------------------------------------------------------------------------------
// KSwitchKeys
// ------------------------------------------------------------------------------------------------
bool KSwitchKeys::on_device() const {
    for (const auto& v : keys_) for (const auto& k : v) return k.on_device();
    return false;
}
void KSwitchKeys::to_device_inplace(MemoryPoolHandle pool) { for (auto& v : keys_) for (auto& k : v) k.to_device_inplace(pool); }
void KSwitchKeys::to_host_inplace() { for (auto& v : keys_) for (auto& k : v) k.to_host_inplace(); }
std::vector<const uint64_t*> KSwitchKeys::get_data_ptrs(size_t index) const {
    std::vector<const uint64_t*> p;
    for (const auto& k : keys_.at(index)) p.push_back(k.as_ciphertext().data().raw_pointer());
    return p;
}

// ------------------------------------------------------------------------------------------------
// Evaluator
// ------------------------------------------------------------------------------------------------
static void check_no_seed(const char* prompt, const Ciphertext& c) {
    if (c.contains_seed()) throw std::invalid_argument(std::string(prompt) + " Argument contains seed.");
}
static void check_same_parms_id(const char* prompt, const Ciphertext& a, const Ciphertext& b) {
    if (a.parms_id() != b.parms_id()) throw std::invalid_argument(std::string(prompt) + " Arguments have different parms ID.");
}
// evaluator_utils.h:307-323
static bool is_scale_within_bounds(double scale, const ContextDataPointer& cd) {
    int bound = -1;
    switch (cd->parms().scheme()) {
        case SchemeType::BFV: case SchemeType::BGV: bound = static_cast<int>(cd->parms().plain_modulus_host().bit_count()); break;
        case SchemeType::CKKS: bound = static_cast<int>(cd->total_coeff_modulus_bit_count()); break;
        default: break;
    }
    return !(scale <= 0.0 || static_cast<int>(std::log2(scale)) >= bound);
}

static bool are_close_double(double a, double b) {
    double s = std::max(std::max(a, b), 1.0);
    return std::fabs(a - b) < s * 2.220446049250313e-16;   // basics.h:150-160
}
static void check_same_scale(const char* prompt, const Ciphertext& a, const Ciphertext& b) {
    if (!are_close_double(a.scale(), b.scale())) throw std::invalid_argument(std::string(prompt) + " Arguments have different scales.");
}
static void check_same_ntt_form(const char* prompt, const Ciphertext& a, const Ciphertext& b) {
    if (a.is_ntt_form() != b.is_ntt_form()) throw std::invalid_argument(std::string(prompt) + " Arguments have different NTT form.");
}
static void check_is_ntt_form(const char* prompt, const Ciphertext& a) {
    if (!a.is_ntt_form()) throw std::invalid_argument(std::string(prompt) + " Argument is not in NTT form.");
}
static void check_is_not_ntt_form(const char* prompt, const Ciphertext& a) {
    if (a.is_ntt_form()) throw std::invalid_argument(std::string(prompt) + " Argument is in NTT form.");
}
static void check_on_device(const char* prompt, const HeContextPointer& ctx, const Ciphertext& a) {
    // the reference dispatches host operands to its CPU branch; this build has no CPU path for the hot path
    if (!ctx->on_device()) throw std::invalid_argument(std::string(prompt) + " HeContext is not on device (call to_device_inplace).");
    if (!a.on_device()) throw std::invalid_argument(std::string(prompt) + " Operand is on host; the evaluator runs on the GPU only.");
}

static uint32_t scheme_is_ckks(const ContextDataPointer& cd) { return cd->parms().scheme() == SchemeType::CKKS ? 1u : 0u; }

ContextDataPointer Evaluator::get_context_data(const char* prompt, const ParmsID& id) const {
    auto cd = context_->get_context_data(id);
    if (!cd.has_value()) throw std::invalid_argument(std::string(prompt) + " Context data not found parms id.");
    return cd.value();
}

// -- negate ------------------------------------------------------------------------------------
void Evaluator::negate(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    check_no_seed("[Evaluator::negate]", encrypted);
    check_on_device("[Evaluator::negate]", context_, encrypted);
    auto cd = get_context_data("[Evaluator::negate]", encrypted.parms_id());
    destination = Ciphertext::like(encrypted, false, pool);
    const size_t L = cd->parms().coeff_modulus().size();
    troyn_check(troyn_negate(context_->plan(), 0, static_cast<uint32_t>(L), encrypted.data().raw_pointer(), destination.data().raw_pointer(),
                             encrypted.polynomial_count(), current_stream()));
}

void Evaluator::negate_inplace(Ciphertext& encrypted) const {
    check_no_seed("[Evaluator::negate_inplace]", encrypted);
    check_on_device("[Evaluator::negate_inplace]", context_, encrypted);
    auto cd = get_context_data("[Evaluator::negate_inplace]", encrypted.parms_id());
    const size_t L = cd->parms().coeff_modulus().size();
    troyn_check(troyn_negate(context_->plan(), 0, static_cast<uint32_t>(L), encrypted.data().raw_pointer(), encrypted.data().raw_pointer(),
                             encrypted.polynomial_count(), current_stream()));
}


// evaluator_utils.h:254-305: e1, e2 with e1 * factor1 = e2 * factor2 = prod (mod t), chosen small (extended Euclid on the ratio)
static void balance_correction_factors(uint64_t factor1, uint64_t factor2, uint64_t t, uint64_t& prod, uint64_t& e1, uint64_t& e2) {
    auto mulmod = [t](uint64_t a, uint64_t b) { return static_cast<uint64_t>((static_cast<unsigned __int128>(a) * b) % t); };
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { uint64_t r = a % b; a = b; b = r; } return a; };
    const uint64_t half_t = t >> 1;
    auto sum_abs = [half_t, t](uint64_t x, uint64_t y) -> uint64_t {
        const int64_t xb = x > half_t ? static_cast<int64_t>(x - t) : static_cast<int64_t>(x);
        const int64_t yb = y > half_t ? static_cast<int64_t>(y - t) : static_cast<int64_t>(y);
        return static_cast<uint64_t>(std::llabs(xb) + std::llabs(yb));
    };
    // factor1^-1 mod t (extended Euclid)
    int64_t r0 = static_cast<int64_t>(t), r1 = static_cast<int64_t>(factor1 % t), s0 = 0, s1 = 1;
    while (r1 != 0) { const int64_t qq = r0 / r1; int64_t tmp = r0 - qq * r1; r0 = r1; r1 = tmp; tmp = s0 - qq * s1; s0 = s1; s1 = tmp; }
    if (r0 != 1) throw std::logic_error("[balance_correction_factors] Failed to invert factor1.");
    uint64_t ratio = mulmod(static_cast<uint64_t>(s0 < 0 ? s0 + static_cast<int64_t>(t) : s0), factor2 % t);
    e1 = ratio; e2 = 1;
    uint64_t sum = sum_abs(factor1, factor2);
    int64_t prev_a = static_cast<int64_t>(t), prev_b = 0, a = static_cast<int64_t>(ratio), b = 1;
    while (a != 0) {
        const int64_t qq = prev_a / a;
        int64_t temp = prev_a % a;
        prev_a = a; a = temp;
        temp = prev_b - qq * b;
        prev_b = b; b = temp;
        uint64_t a_mod = static_cast<uint64_t>(std::llabs(a)) % t;
        if (a < 0 && a_mod != 0) a_mod = t - a_mod;
        uint64_t b_mod = static_cast<uint64_t>(std::llabs(b)) % t;
        if (b < 0 && b_mod != 0) b_mod = t - b_mod;
        if (a_mod != 0 && gcd(a_mod, t) == 1) {
            const uint64_t new_sum = sum_abs(a_mod, b_mod);
            if (new_sum < sum) { e1 = a_mod; e2 = b_mod; sum = new_sum; }
        }
    }
    prod = mulmod(e1, factor1 % t);
}

// -- add / sub (evaluator_translate.cu:12-118) ------------------------------------------------------
void Evaluator::translate(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& destination, bool subtract, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::translate_inplace]";
    check_no_seed(P, e1); check_no_seed(P, e2);
    check_same_parms_id(P, e1, e2);
    check_same_scale(P, e1, e2);
    check_same_ntt_form(P, e1, e2);
    check_on_device(P, context_, e1); check_on_device(P, context_, e2);
    auto cd = get_context_data(P, e1.parms_id());
    if (e1.correction_factor() != e2.correction_factor()) {
        // evaluator_translate.cu:84-98: bring both operands to a common correction factor first
        if (cd->parms().scheme() != SchemeType::BGV) throw std::invalid_argument(std::string(P) + " Correction factors differ outside BGV.");
        uint64_t f0 = 1, f1 = 1, f2 = 1;
        balance_correction_factors(e1.correction_factor(), e2.correction_factor(), cd->parms().plain_modulus().value(), f0, f1, f2);
        const uint32_t Lb = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
        Ciphertext a = Ciphertext::like(e1, false, pool), b = Ciphertext::like(e2, false, pool);
        troyn_check(troyn_multiply_scalar(context_->plan(), 0, Lb, e1.data().raw_pointer(), f1, a.data().raw_pointer(), e1.polynomial_count(), current_stream()));
        troyn_check(troyn_multiply_scalar(context_->plan(), 0, Lb, e2.data().raw_pointer(), f2, b.data().raw_pointer(), e2.polynomial_count(), current_stream()));
        a.correction_factor() = f0;
        b.correction_factor() = f0;
        translate(a, b, destination, subtract, pool);
        // (no stream wait: the temporaries return to the pool in stream order, the call is asynchronous like the reference's)
        return;
    }
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree();
    const size_t s1 = e1.polynomial_count(), s2 = e2.polynomial_count();
    const size_t mx = std::max(s1, s2), mn = std::min(s1, s2);
    Ciphertext out = Ciphertext::like(e1, mx, false, pool);
    const troyn_plan* plan = context_->plan();
    if (!subtract) troyn_check(troyn_add(plan, 0, L, e1.data().raw_pointer(), e2.data().raw_pointer(), out.data().raw_pointer(), mn, current_stream()));
    else troyn_check(troyn_sub(plan, 0, L, e1.data().raw_pointer(), e2.data().raw_pointer(), out.data().raw_pointer(), mn, current_stream()));
    const size_t pc = static_cast<size_t>(L) * n;
    if (s1 < s2) {
        if (!subtract) hip_check(hipMemcpyAsync(out.poly(s1), e2.poly(s1), (s2 - s1) * pc * 8, hipMemcpyDeviceToDevice, current_stream()), "copy_device_to_device");
        else troyn_check(troyn_negate(plan, 0, L, e2.poly(s1), out.poly(s1), s2 - s1, current_stream()));
    } else if (s1 > s2) {
        hip_check(hipMemcpyAsync(out.poly(s2), e1.poly(s2), (s1 - s2) * pc * 8, hipMemcpyDeviceToDevice, current_stream()), "copy_device_to_device");
    }
    destination = std::move(out);
}

void Evaluator::translate_inplace(Ciphertext& e1, const Ciphertext& e2, bool subtract, MemoryPoolHandle pool) const {
    Ciphertext d;
    translate(e1, e2, d, subtract, pool);
    e1 = std::move(d);
}


// -- multiply / square (evaluator.cu:29-343) ------------------------------------------------------------
// every argument check of multiply and the result object (allocated, metadata set) WITHOUT the product itself: shared by the method, by the
// *_batched form (item 0 stands for a uniform batch) and by the call-combining rendezvous
SchemeType Evaluator::multiply_prepare(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& out, MemoryPoolHandle pool) const {
    check_no_seed("[Evaluator::multiply]", e1); check_no_seed("[Evaluator::multiply]", e2);
    check_same_parms_id("[Evaluator::multiply]", e1, e2);
    check_on_device("[Evaluator::multiply]", context_, e1); check_on_device("[Evaluator::multiply]", context_, e2);
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    auto cd = get_context_data("[Evaluator::multiply]", e1.parms_id());
    const size_t p1 = e1.polynomial_count(), p2 = e2.polynomial_count();
    switch (scheme) {
        case SchemeType::BFV:
            check_is_not_ntt_form("[Evaluator::bfv_multiply_inplace]", e1); check_is_not_ntt_form("[Evaluator::bfv_multiply_inplace]", e2);
            out = Ciphertext::like(e1, p1 + p2 - 1, false, pool);
            break;
        case SchemeType::CKKS: case SchemeType::BGV: {
            const char* P = scheme == SchemeType::CKKS ? "[Evaluator::ckks_multiply_inplace]" : "[Evaluator::bgv_multiply]";
            check_is_ntt_form(P, e1); check_is_ntt_form(P, e2);
            out = Ciphertext::like(e1, p1 + p2 - 1, false, pool);
            if (scheme == SchemeType::CKKS) {
                out.scale() = e1.scale() * e2.scale();
                if (!is_scale_within_bounds(out.scale(), cd)) throw std::invalid_argument("[Evaluator::ckks_multiply] Scale out of bounds");   // evaluator.cu:140-143
            } else {
                const Modulus& t = cd->parms().plain_modulus();
                out.correction_factor() = (uint64_t)(((unsigned __int128)e1.correction_factor() * e2.correction_factor()) % t.value());
            }
            break;
        }
        default: throw std::logic_error("[Evaluator::multiply] Scheme not implemented.");
    }
    return scheme;
}

void Evaluator::multiply(const Ciphertext& e1, const Ciphertext& e2, Ciphertext& destination, MemoryPoolHandle pool) const {
    Ciphertext out;
    const SchemeType scheme = multiply_prepare(e1, e2, out, pool);
    const uint32_t L = static_cast<uint32_t>(out.coeff_modulus_size());
    const size_t p1 = e1.polynomial_count(), p2 = e2.polynomial_count();
    // call combining (troy.h): the batch only needs shapes and pointers
    if (detail::combining_wanted()) {
        detail::CombineRequest rq;
        rq.L = L; rq.p1 = static_cast<uint32_t>(p1); rq.p2 = static_cast<uint32_t>(p2);
        rq.in1 = e1.data().raw_pointer(); rq.words1 = e1.data().size(); rq.in2 = e2.data().raw_pointer(); rq.words2 = e2.data().size();
        rq.out = out.data().raw_pointer(); rq.out_words = out.data().size();
        if (scheme == SchemeType::BFV) { rq.kind = detail::CombineKind::BfvMultiply; rq.handle = context_->behz(L); }
        else { rq.kind = detail::CombineKind::DyadicMultiply; rq.handle = context_->plan(); rq.ntt_form = true; }
        if (detail::combine_submit(rq, pool)) { destination = std::move(out); return; }
    }
    if (scheme == SchemeType::BFV) {
        const troyn_behz* bz = context_->behz(L);
        size_t bytes = troyn_bfv_multiply_workspace_bytes(bz, p1, p2, 1);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        troyn_check(troyn_bfv_multiply(bz, e1.data().raw_pointer(), p1, e2.data().raw_pointer(), p2, out.data().raw_pointer(), ws.raw_pointer(), bytes, 1, current_stream()));
    } else {
        detail::LaunchGate gate;
        troyn_check(troyn_dyadic_convolute(context_->plan(), 0, L, e1.data().raw_pointer(), p1, e2.data().raw_pointer(), p2, out.data().raw_pointer(), 1, current_stream()));
    }
    destination = std::move(out);
}

void Evaluator::square(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    check_no_seed("[Evaluator::square]", encrypted);
    check_on_device("[Evaluator::square]", context_, encrypted);
    SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme == SchemeType::BFV || encrypted.polynomial_count() != 2) { multiply(encrypted, encrypted, destination, pool); return; }
    auto cd = get_context_data("[Evaluator::square]", encrypted.parms_id());
    check_is_ntt_form("[Evaluator::ckks_square_inplace]", encrypted);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    Ciphertext out = Ciphertext::like(encrypted, 3, false, pool);
    troyn_check(troyn_dyadic_square(context_->plan(), 0, L, encrypted.data().raw_pointer(), out.data().raw_pointer(), 1, current_stream()));
    if (scheme == SchemeType::CKKS) {
        out.scale() = encrypted.scale() * encrypted.scale();
        if (!is_scale_within_bounds(out.scale(), cd)) throw std::invalid_argument("[Evaluator::ckks_multiply_inplace] Scale out of bounds");   // evaluator.cu:308-311
    } else {
        const Modulus& t = cd->parms().plain_modulus();
        out.correction_factor() = (uint64_t)(((unsigned __int128)encrypted.correction_factor() * encrypted.correction_factor()) % t.value());
    }
    destination = std::move(out);
}


// -- key switching (evaluator_keyswitching_core.cu:757-1052, evaluator_keyswitching.cu:11-144) ---------------
// the argument checks of switch_key_internal (evaluator_keyswitching_core.cu:757-830), also run by a call that is handed to the combining rendezvous
void Evaluator::switch_key_checks(const Ciphertext& encrypted, const KSwitchKeys& kswitch_keys, size_t kswitch_keys_index, const Ciphertext& destination) const {
    const char* P = "[Evaluator::switch_key_inplace_internal]";
    check_no_seed(P, encrypted);
    if (!context_->using_keyswitching()) throw std::invalid_argument(std::string(P) + " Keyswitching is not supported.");
    if (kswitch_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Keyswitching key has incorrect parms id.");
    if (kswitch_keys_index >= kswitch_keys.data().size()) throw std::out_of_range(std::string(P) + " Key switch keys index out of range.");
    auto cd = get_context_data(P, encrypted.parms_id());
    SchemeType scheme = cd->parms().scheme();
    if (scheme == SchemeType::BGV && !encrypted.is_ntt_form()) throw std::invalid_argument(std::string(P) + " BGV ciphertexts are in NTT form.");
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const auto& key_vector = kswitch_keys.data()[kswitch_keys_index];
    if (key_vector.size() < L) throw std::invalid_argument(std::string(P) + " Key switching key has too few components for this level.");
    if (destination.polynomial_count() < 2) throw std::invalid_argument(std::string(P) + " Destination should have at least same amount of polys as the key switching key.");
    if (destination.parms_id() != encrypted.parms_id()) throw std::invalid_argument(std::string(P) + " Destination parms_id should match the input parms_id.");
    for (const auto& k : key_vector) {
        check_no_seed(P, k.as_ciphertext());
        if (!k.on_device()) throw std::invalid_argument(std::string(P) + " Incompatible encryption parameters.");
    }
    check_on_device(P, context_, encrypted); check_on_device(P, context_, destination);
}

void Evaluator::switch_key_internal(const Ciphertext& encrypted, const uint64_t* target, const KSwitchKeys& kswitch_keys, size_t kswitch_keys_index,
                                    SwitchKeyDestinationAssignMethod assign_method, Ciphertext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::switch_key_inplace_internal]";
    switch_key_checks(encrypted, kswitch_keys, kswitch_keys_index, destination);
    auto cd = get_context_data(P, encrypted.parms_id());
    SchemeType scheme = cd->parms().scheme();
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree();
    if (destination.polynomial_count() > 2 && assign_method != SwitchKeyDestinationAssignMethod::AddInplace)
        hip_check(hipMemsetAsync(destination.poly(2), 0, (destination.polynomial_count() - 2) * L * n * 8, current_stream()), "memset");
    std::vector<const uint64_t*> ptrs = kswitch_keys.get_data_ptrs(kswitch_keys_index);
    size_t bytes = troyn_switch_key_workspace_bytes(context_->plan(), L, 1);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    if (scheme == SchemeType::BGV) {
        // the ski_util5 tail needs the key level's q_special^-1 mod t (evaluator_keyswitching_core.cu:930-932)
        const size_t K = context_->key_context_data().value()->parms().coeff_modulus().size();
        troyn_check(troyn_bgv_switch_key(context_->bgv(K), L, target, ptrs.data(), static_cast<int>(assign_method), destination.data().raw_pointer(), ws.raw_pointer(), bytes, 1,
                                         current_stream()));
    } else
    troyn_check(troyn_switch_key(context_->plan(), L, scheme == SchemeType::CKKS, encrypted.is_ntt_form(), target, ptrs.data(),
                                 static_cast<int>(assign_method), destination.data().raw_pointer(), ws.raw_pointer(), bytes, 1, current_stream()));
}

void Evaluator::apply_keyswitching(const Ciphertext& encrypted, const KSwitchKeys& kswitch_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_keyswitching.cu:11-50
    if (kswitch_keys.data().size() != 1) throw std::invalid_argument("[Evaluator::apply_keyswitching_inplace] Key switch keys size must be 1.");
    if (encrypted.polynomial_count() != 2) throw std::invalid_argument("[Evaluator::apply_keyswitching_inplace] Ciphertext polynomial count must be 2.");
    auto cd = get_context_data("[Evaluator::apply_keyswitching_inplace]", encrypted.parms_id());
    Ciphertext out = Ciphertext::like(encrypted, false, pool);
    switch_key_internal(encrypted, encrypted.poly(1), kswitch_keys, 0, SwitchKeyDestinationAssignMethod::Overwrite, out, pool);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    // c0' = c0 + ks0
    troyn_check(troyn_add(context_->plan(), 0, L, out.poly(0), encrypted.poly(0), out.poly(0), 1, current_stream()));
    destination = std::move(out);
}

void Evaluator::apply_keyswitching_inplace(Ciphertext& encrypted, const KSwitchKeys& kswitch_keys, MemoryPoolHandle pool) const {
    Ciphertext d;
    apply_keyswitching(encrypted, kswitch_keys, d, pool);
    encrypted = std::move(d);
}

// 3 -> 2 components: every argument check of relinearize (evaluator_keyswitching.cu:119-144 and the key-switch checks behind it), the key
// pointers and the result object WITHOUT the key switch: shared by the method, the *_batched form and the call-combining rendezvous
void Evaluator::relinearize_prepare(const Ciphertext& encrypted, const RelinKeys& relin_keys, Ciphertext& out, std::vector<const uint64_t*>& key_ptrs, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::relinearize_inplace_internal]";
    check_no_seed(P, encrypted);
    if (relin_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Relin keys has incorrect parms id.");
    auto cd = get_context_data(P, encrypted.parms_id());
    if (encrypted.polynomial_count() != 3) throw std::invalid_argument(std::string(P) + " Destination size must be at least 2 and less/equal to the size of the encrypted polynomial.");
    check_on_device(P, context_, encrypted);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t idx = RelinKeys::get_index(2);
    if (idx >= relin_keys.data().size()) throw std::out_of_range(std::string(P) + " Key switch keys index out of range.");
    const SchemeType scheme = cd->parms().scheme();
    if (scheme == SchemeType::BGV && !encrypted.is_ntt_form()) throw std::invalid_argument(std::string(P) + " BGV ciphertexts are in NTT form.");
    if (!context_->using_keyswitching()) throw std::invalid_argument("[Evaluator::switch_key_inplace_internal] Keyswitching is not supported.");
    if (relin_keys.data()[idx].size() < L) throw std::invalid_argument(std::string(P) + " Key switching key has too few components for this level.");
    for (const auto& k : relin_keys.data()[idx]) if (!k.on_device()) throw std::invalid_argument(std::string(P) + " Incompatible encryption parameters.");
    key_ptrs = relin_keys.get_data_ptrs(idx);
    out = Ciphertext::like(encrypted, 2, false, pool);
}

void Evaluator::relinearize_internal(const Ciphertext& encrypted, const RelinKeys& relin_keys, size_t destination_size, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_keyswitching.cu:119-144
    const char* P = "[Evaluator::relinearize_inplace_internal]";
    check_no_seed(P, encrypted);
    if (relin_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Relin keys has incorrect parms id.");
    auto cd = get_context_data(P, encrypted.parms_id());
    size_t encrypted_size = encrypted.polynomial_count();
    if (encrypted_size < 2 || destination_size > encrypted_size)
        throw std::invalid_argument(std::string(P) + " Destination size must be at least 2 and less/equal to the size of the encrypted polynomial.");
    if (destination_size == encrypted_size) { destination = encrypted; return; }
    check_on_device(P, context_, encrypted);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    if (encrypted_size == 3 && destination_size == 2) {
        // one call: switch_key(c2, Overwrite) + (c0, c1), the trailing add fused into the key-switch epilogue
        Ciphertext out;
        std::vector<const uint64_t*> ptrs;
        relinearize_prepare(encrypted, relin_keys, out, ptrs, pool);
        const SchemeType scheme = cd->parms().scheme();
        if (scheme != SchemeType::BGV && detail::combining_wanted()) {
            detail::CombineRequest rq;
            rq.kind = detail::CombineKind::Relinearize; rq.handle = context_->plan(); rq.L = L; rq.p1 = 3;
            rq.ckks = scheme == SchemeType::CKKS; rq.ntt_form = encrypted.is_ntt_form(); rq.keys = &ptrs;
            rq.in1 = encrypted.data().raw_pointer(); rq.words1 = encrypted.data().size();
            rq.out = out.data().raw_pointer(); rq.out_words = out.data().size();
            if (detail::combine_submit(rq, pool)) { destination = std::move(out); return; }
        }
        size_t bytes = troyn_relinearize_workspace_bytes(context_->plan(), L, 1);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        if (scheme == SchemeType::BGV) {
            const size_t K = context_->key_context_data().value()->parms().coeff_modulus().size();
            troyn_check(troyn_bgv_relinearize(context_->bgv(K), L, encrypted.data().raw_pointer(), ptrs.data(), out.data().raw_pointer(), ws.raw_pointer(), bytes, 1, current_stream()));
        } else {
        detail::LaunchGate gate;
        troyn_check(troyn_relinearize(context_->plan(), L, scheme == SchemeType::CKKS, encrypted.is_ntt_form(), encrypted.data().raw_pointer(), ptrs.data(),
                                      out.data().raw_pointer(), ws.raw_pointer(), bytes, 1, current_stream()));
        }
        destination = std::move(out);
        return;
    }
    Ciphertext out = Ciphertext::like(encrypted, destination_size, false, pool);
    size_t relins_needed = encrypted_size - destination_size;
    for (size_t i = 0; i < relins_needed; i++) {
        switch_key_internal(encrypted, encrypted.poly(encrypted_size - 1), relin_keys.as_kswitch_keys(), RelinKeys::get_index(encrypted_size - 1),
                            i == 0 ? SwitchKeyDestinationAssignMethod::Overwrite : SwitchKeyDestinationAssignMethod::AddInplace, out, pool);
        encrypted_size -= 1;
    }
    troyn_check(troyn_add(context_->plan(), 0, L, out.data().raw_pointer(), encrypted.data().raw_pointer(), out.data().raw_pointer(), destination_size, current_stream()));
    destination = std::move(out);
}

void Evaluator::relinearize_inplace_internal(Ciphertext& encrypted, const RelinKeys& relin_keys, size_t destination_size, MemoryPoolHandle pool) const {
    Ciphertext d;
    relinearize_internal(encrypted, relin_keys, destination_size, d, pool);
    encrypted = std::move(d);
}


// -- modulus switching (evaluator_modswitch.cu) --------------------------------------------------------------
// the argument checks of mod_switch_scale_to_next_internal (evaluator_modswitch.cu:14-74) and the result object one level down (allocated,
// parms_id / form / CKKS scale set) WITHOUT the division: shared by the method, the *_batched forms and the call-combining rendezvous
SchemeType Evaluator::mod_switch_scale_prepare(const Ciphertext& encrypted, Ciphertext& out, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::mod_switch_scale_to_next_internal]";
    auto cd = get_context_data(P, encrypted.parms_id());
    const SchemeType scheme = cd->parms().scheme();
    if (scheme == SchemeType::BFV) check_is_not_ntt_form(P, encrypted);
    else if (scheme == SchemeType::CKKS || scheme == SchemeType::BGV) check_is_ntt_form(P, encrypted);
    else throw std::logic_error(std::string(P) + " Scheme not implemented.");
    if (!cd->next_context_data().has_value()) throw std::invalid_argument(std::string(P) + " Next context data is not set.");
    check_on_device(P, context_, encrypted);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    out = Ciphertext::like(encrypted, encrypted.polynomial_count(), L - 1, false, pool);
    out.parms_id() = cd->next_context_data().value()->parms_id();
    out.is_ntt_form() = encrypted.is_ntt_form();
    if (scheme == SchemeType::CKKS) out.scale() = encrypted.scale() / static_cast<double>(cd->parms().coeff_modulus()[L - 1].value());
    return scheme;
}

void Evaluator::mod_switch_scale_to_next_internal(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_modswitch.cu:14-74
    Ciphertext out;
    const SchemeType scheme = mod_switch_scale_prepare(encrypted, out, pool);
    const uint32_t L = static_cast<uint32_t>(encrypted.coeff_modulus_size());
    const size_t pc = encrypted.polynomial_count();
    if (scheme == SchemeType::CKKS && detail::combining_wanted()) {
        detail::CombineRequest rq;
        rq.kind = detail::CombineKind::Rescale; rq.handle = context_->plan(); rq.L = L; rq.p1 = static_cast<uint32_t>(pc); rq.ckks = true; rq.ntt_form = true;
        rq.in1 = encrypted.data().raw_pointer(); rq.words1 = encrypted.data().size();
        rq.out = out.data().raw_pointer(); rq.out_words = out.data().size();
        if (detail::combine_submit(rq, pool)) { destination = std::move(out); return; }
    }
    if (scheme == SchemeType::BFV) {
        troyn_check(troyn_divide_and_round_q_last(context_->plan(), L, encrypted.data().raw_pointer(), pc, out.data().raw_pointer(), 1, current_stream()));
    } else if (scheme == SchemeType::BGV) {
        // RNSTool::mod_t_and_divide_q_last_ntt; the plaintext is multiplied by q_last^-1 mod t (evaluator_modswitch.cu:62-72)
        const troyn_bgv* bg = context_->bgv(L);
        const size_t bytes = troyn_bgv_mod_switch_workspace_bytes(bg, pc, 1);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        troyn_check(troyn_bgv_mod_t_and_divide_q_last_ntt(bg, encrypted.data().raw_pointer(), pc, out.data().raw_pointer(), ws.raw_pointer(), bytes, 1, current_stream()));
        const uint64_t t = context_->get_context_data(out.parms_id()).value()->parms().plain_modulus().value();
        out.correction_factor() = static_cast<uint64_t>((static_cast<unsigned __int128>(encrypted.correction_factor()) * troyn_bgv_inv_q_last_mod_t(bg)) % t);
    } else {
        size_t bytes = troyn_divide_and_round_q_last_ntt_workspace_bytes(context_->plan(), L, pc, 1);
        utils::DynamicArray ws((bytes + 7) / 8, true, pool);
        detail::LaunchGate gate;
        troyn_check(troyn_divide_and_round_q_last_ntt(context_->plan(), L, encrypted.data().raw_pointer(), pc, out.data().raw_pointer(),
                                                      ws.raw_pointer(), bytes, 1, current_stream()));
    }
    destination = std::move(out);
}

void Evaluator::mod_switch_drop_to_internal(const Ciphertext& encrypted, Ciphertext& destination, const ParmsID& target, MemoryPoolHandle pool) const {
    // evaluator_modswitch.cu:173-220
    const char* P = "[Evaluator::mod_switch_drop_to_internal]";
    auto cd = get_context_data(P, encrypted.parms_id());
    if (cd->parms().scheme() == SchemeType::CKKS) check_is_ntt_form(P, encrypted);
    if (!cd->next_context_data().has_value()) throw std::invalid_argument("[Evaluator::mod_switch_drop_to_next_internal] Next context data is not set.");
    auto tcd = get_context_data("[Evaluator::mod_switch_drop_to_next_internal]", target);
    if (!is_scale_within_bounds(encrypted.scale(), tcd)) throw std::invalid_argument("[Evaluator::mod_switch_drop_to_internal] Scale out of bounds.");   // evaluator_modswitch.cu:186-188
    check_on_device(P, context_, encrypted);
    const uint32_t L_in = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const uint32_t L_out = static_cast<uint32_t>(tcd->parms().coeff_modulus().size());
    const size_t pc = encrypted.polynomial_count();
    Ciphertext out = Ciphertext::like(encrypted, pc, L_out, false, pool);
    out.parms_id() = target;
    troyn_check(troyn_mod_switch_drop(context_->plan(), L_in, L_out, encrypted.data().raw_pointer(), pc, out.data().raw_pointer(), 1, current_stream()));
    destination = std::move(out);
}

void Evaluator::mod_switch_to_next(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_modswitch.cu:275-300
    check_no_seed("[Evaluator::mod_switch_to_next]", encrypted);
    if (context_->last_parms_id() == encrypted.parms_id()) throw std::invalid_argument("[Evaluator::mod_switch_to_next] End of modulus switching chain reached.");
    SchemeType scheme = context_->first_context_data().value()->parms().scheme();
    switch (scheme) {
        case SchemeType::BFV: case SchemeType::BGV: mod_switch_scale_to_next_internal(encrypted, destination, pool); break;
        case SchemeType::CKKS: {
            auto cd = get_context_data("[Evaluator::mod_switch_to_next]", encrypted.parms_id());
            mod_switch_drop_to_internal(encrypted, destination, cd->next_context_data().value()->parms_id(), pool);
            break;
        }
        default: throw std::logic_error("[Evaluator::mod_switch_to_next] Scheme not implemented.");
    }
}


void Evaluator::mod_switch_to(const Ciphertext& encrypted, const ParmsID& parms_id, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_modswitch.cu:330-360
    auto cd = get_context_data("[Evaluator::mod_switch_to]", encrypted.parms_id());
    auto tcd = get_context_data("[Evaluator::mod_switch_to]", parms_id);
    if (cd->chain_index() < tcd->chain_index()) throw std::invalid_argument("[Evaluator::mod_switch_to] Cannot switch to a higher level.");
    if (encrypted.parms_id() == parms_id) { destination = encrypted; return; }
    SchemeType scheme = cd->parms().scheme();
    if (scheme == SchemeType::CKKS) { mod_switch_drop_to_internal(encrypted, destination, parms_id, pool); return; }
    Ciphertext cur = encrypted;
    while (cur.parms_id() != parms_id) { Ciphertext nx; mod_switch_to_next(cur, nx, pool); cur = std::move(nx); }
    destination = std::move(cur);
}

void Evaluator::rescale_to_next(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_modswitch.cu:445-461
    check_no_seed("[Evaluator::rescale_to_next]", encrypted);
    if (context_->last_parms_id() == encrypted.parms_id()) throw std::invalid_argument("[Evaluator::rescale_to_next] End of modulus switching chain reached.");
    SchemeType scheme = context_->first_context_data().value()->parms().scheme();
    switch (scheme) {
        case SchemeType::BFV: case SchemeType::BGV: throw std::invalid_argument("[Evaluator::rescale_to_next] Cannot rescale BFV/BGV ciphertext.");
        case SchemeType::CKKS: mod_switch_scale_to_next_internal(encrypted, destination, pool); break;
        default: throw std::logic_error("[Evaluator::rescale_to_next] Scheme not implemented.");
    }
}


// -- multiply -> relinearize -> rescale_to_next as one call (addition; evaluator.cu:118-145, evaluator_keyswitching.cu:119-144, ---------
// -- evaluator_modswitch.cu:14-74, :445-461) -------------------------------------------------------------------------------------
bool Evaluator::multiply_relinearize_rescale_prepare(const Ciphertext& e1, const Ciphertext& e2, const RelinKeys& relin_keys, uint32_t& L, ParmsID& next_parms_id,
                                                     double& scale, std::vector<const uint64_t*>& key_ptrs) const {
    // Evaluator::multiply's checks
    check_no_seed("[Evaluator::multiply]", e1); check_no_seed("[Evaluator::multiply]", e2);
    check_same_parms_id("[Evaluator::multiply]", e1, e2);
    check_on_device("[Evaluator::multiply]", context_, e1); check_on_device("[Evaluator::multiply]", context_, e2);
    SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme != SchemeType::CKKS || e1.polynomial_count() != 2 || e2.polynomial_count() != 2) return false;
    auto cd = get_context_data("[Evaluator::multiply]", e1.parms_id());
    check_is_ntt_form("[Evaluator::ckks_multiply_inplace]", e1); check_is_ntt_form("[Evaluator::ckks_multiply_inplace]", e2);
    L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    scale = e1.scale() * e2.scale();
    if (!is_scale_within_bounds(scale, cd)) throw std::invalid_argument("[Evaluator::ckks_multiply] Scale out of bounds");
    // relinearize_internal's checks (3 -> 2 components)
    const char* P = "[Evaluator::relinearize_inplace_internal]";
    if (relin_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Relin keys has incorrect parms id.");
    const size_t idx = RelinKeys::get_index(2);
    if (idx >= relin_keys.data().size()) throw std::out_of_range(std::string(P) + " Key switch keys index out of range.");
    if (!context_->using_keyswitching()) throw std::invalid_argument("[Evaluator::switch_key_inplace_internal] Keyswitching is not supported.");
    if (relin_keys.data()[idx].size() < L) throw std::invalid_argument(std::string(P) + " Key switching key has too few components for this level.");
    for (const auto& k : relin_keys.data()[idx]) {
        check_no_seed("[Evaluator::switch_key_inplace_internal]", k.as_ciphertext());
        if (!k.on_device()) throw std::invalid_argument(std::string(P) + " Incompatible encryption parameters.");
    }
    // rescale_to_next's checks
    if (context_->last_parms_id() == e1.parms_id()) throw std::invalid_argument("[Evaluator::rescale_to_next] End of modulus switching chain reached.");
    if (!cd->next_context_data().has_value()) throw std::invalid_argument("[Evaluator::mod_switch_scale_to_next_internal] Next context data is not set.");
    next_parms_id = cd->next_context_data().value()->parms_id();
    scale = scale / static_cast<double>(cd->parms().coeff_modulus()[L - 1].value());
    key_ptrs = relin_keys.get_data_ptrs(idx);
    return true;
}

void Evaluator::multiply_relinearize_rescale(const Ciphertext& e1, const Ciphertext& e2, const RelinKeys& relin_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    uint32_t L = 0; ParmsID next; double scale = 1.0; std::vector<const uint64_t*> keys;
    if (!multiply_relinearize_rescale_prepare(e1, e2, relin_keys, L, next, scale, keys)) {
        Ciphertext m, r;
        multiply(e1, e2, m, pool);
        relinearize(m, relin_keys, r, pool);
        rescale_to_next(r, destination, pool);
        return;
    }
    Ciphertext out = Ciphertext::like(e1, 2, L - 1, false, pool);
    out.parms_id() = next;
    out.scale() = scale;
    out.is_ntt_form() = true;
    if (detail::combining_wanted()) {
        detail::CombineRequest rq;
        rq.kind = detail::CombineKind::MultiplyRelinearizeRescale; rq.handle = context_->plan(); rq.L = L; rq.p1 = 2; rq.p2 = 2; rq.ckks = true; rq.ntt_form = true;
        rq.keys = &keys;
        rq.in1 = e1.data().raw_pointer(); rq.words1 = e1.data().size(); rq.in2 = e2.data().raw_pointer(); rq.words2 = e2.data().size();
        rq.out = out.data().raw_pointer(); rq.out_words = out.data().size();
        if (detail::combine_submit(rq, pool)) { destination = std::move(out); return; }
    }
    const size_t bytes = troyn_ckks_multiply_relinearize_rescale_workspace_bytes(context_->plan(), L, 1);
    utils::DynamicArray ws((bytes + 7) / 8, true, pool);
    {
        detail::LaunchGate gate;
        troyn_check(troyn_ckks_multiply_relinearize_rescale(context_->plan(), L, e1.data().raw_pointer(), e2.data().raw_pointer(), keys.data(), out.data().raw_pointer(),
                                                            ws.raw_pointer(), bytes, 1, current_stream()));
    }
    destination = std::move(out);
}


// -- NTT (evaluator_transform_ntt.cu:469-652) ----------------------------------------------------------------
void Evaluator::transform_to_ntt_inplace(Ciphertext& encrypted) const {
    check_no_seed("[Evaluator::transform_to_ntt_inplace]", encrypted);
    check_is_not_ntt_form("[Evaluator::transform_to_ntt_inplace]", encrypted);
    check_on_device("[Evaluator::transform_to_ntt_inplace]", context_, encrypted);
    auto cd = get_context_data("[Evaluator::transform_to_ntt_inplace]", encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    troyn_check(troyn_ntt(context_->plan(), 0, encrypted.data().raw_pointer(), encrypted.data().raw_pointer(), 1, encrypted.polynomial_count(), L,
                          0, L, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    encrypted.is_ntt_form() = true;
}

void Evaluator::transform_to_ntt(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    check_no_seed("[Evaluator::transform_to_ntt]", encrypted);
    check_is_not_ntt_form("[Evaluator::transform_to_ntt]", encrypted);
    check_on_device("[Evaluator::transform_to_ntt]", context_, encrypted);
    auto cd = get_context_data("[Evaluator::transform_to_ntt]", encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    Ciphertext out = Ciphertext::like(encrypted, false, pool);
    troyn_check(troyn_ntt(context_->plan(), 0, encrypted.data().raw_pointer(), out.data().raw_pointer(), 1, encrypted.polynomial_count(), L,
                          0, L, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    out.is_ntt_form() = true;
    destination = std::move(out);
}

void Evaluator::transform_from_ntt_inplace(Ciphertext& encrypted) const {
    check_no_seed("[Evaluator::transform_from_ntt_inplace]", encrypted);
    check_is_ntt_form("[Evaluator::transform_from_ntt_inplace]", encrypted);
    check_on_device("[Evaluator::transform_from_ntt_inplace]", context_, encrypted);
    auto cd = get_context_data("[Evaluator::transform_from_ntt_inplace]", encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    troyn_check(troyn_ntt(context_->plan(), 1, encrypted.data().raw_pointer(), encrypted.data().raw_pointer(), 1, encrypted.polynomial_count(), L,
                          0, L, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    encrypted.is_ntt_form() = false;
}

void Evaluator::transform_from_ntt(const Ciphertext& encrypted, Ciphertext& destination, MemoryPoolHandle pool) const {
    check_no_seed("[Evaluator::transform_from_ntt]", encrypted);
    check_is_ntt_form("[Evaluator::transform_from_ntt]", encrypted);
    check_on_device("[Evaluator::transform_from_ntt]", context_, encrypted);
    auto cd = get_context_data("[Evaluator::transform_from_ntt]", encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    Ciphertext out = Ciphertext::like(encrypted, false, pool);
    troyn_check(troyn_ntt(context_->plan(), 1, encrypted.data().raw_pointer(), out.data().raw_pointer(), 1, encrypted.polynomial_count(), L,
                          0, L, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    out.is_ntt_form() = false;
    destination = std::move(out);
}


// ------------------------------------------------------------------------------------------------
// Evaluator: ciphertext x plaintext  (evaluator_multiply_plain.cu, evaluator_transform_ntt.cu:35-70)
// ------------------------------------------------------------------------------------------------
void Evaluator::transform_plain_to_ntt(const Plaintext& plain, const ParmsID& parms_id, Plaintext& destination, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::transform_plain_to_ntt]";
    if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is already in NTT form.");
    if (!context_->on_device() || !plain.on_device()) throw std::invalid_argument(std::string(P) + " Operand is on host; the evaluator runs on the GPU only.");
    auto cd = get_context_data("[Evaluator::transform_plain_to_ntt_inplace]", parms_id);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree();
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, parms_id);
    out.scale() = plain.scale();
    hipStream_t s = current_stream();
    if (plain.parms_id() == parms_id_zero) {
        // scaling_variant::centralize in the loader of the forward transform: one launch, the centred polynomial is never written
        troyn_check(troyn_plain_centralize_ntt(context_->plan(), L, cd->parms().plain_modulus().value(), plain.poly(), plain.coeff_count(), n,
                                               out.poly(), 1, s));
    } else {
        if (plain.parms_id() != parms_id) throw std::invalid_argument(std::string(P) + " Plaintext parameters do not match.");
        if (plain.coeff_count() != n) {
            // a partial RNS plaintext: zero-padded to the full shape, then transformed in place
            utils::DynamicArray full = plain.expanded_rns(L, n, pool);
            troyn_check(troyn_ntt(context_->plan(), 0, full.raw_pointer(), out.poly(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
            hip_check(hipStreamSynchronize(s), "stream_sync");
        } else {
            troyn_check(troyn_ntt(context_->plan(), 0, plain.poly(), out.poly(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
        }
    }
    out.is_ntt_form() = true;
    destination = std::move(out);
}

void Evaluator::multiply_plain(const Ciphertext& encrypted, const Plaintext& plain, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_multiply_plain.cu:309-326 dispatch; :13-68 (normal), :196-218 (ntt)
    const char* P = "[Evaluator::multiply_plain]";
    check_no_seed(P, encrypted);
    check_on_device(P, context_, encrypted);
    auto cd = get_context_data(P, encrypted.parms_id());
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t pc = encrypted.polynomial_count();
    hipStream_t s = current_stream();
    Plaintext plain_ntt_storage;
    const Plaintext* pn = &plain;
    if (!plain.is_ntt_form()) {
        transform_plain_to_ntt(plain, encrypted.parms_id(), plain_ntt_storage, pool);
        pn = &plain_ntt_storage;
    } else if (plain.parms_id() != encrypted.parms_id()) {
        throw std::invalid_argument("[Evaluator::multiply_plain_ntt] Plaintext and ciphertext parameters do not match.");
    }
    Ciphertext out = Ciphertext::like(encrypted, false, pool);
    const uint64_t* src = encrypted.data().raw_pointer();
    if (!encrypted.is_ntt_form()) {
        troyn_check(troyn_ntt(context_->plan(), 0, src, out.data().raw_pointer(), 1, pc, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
        src = out.data().raw_pointer();
    }
    troyn_check(troyn_dyadic_broadcast_product(context_->plan(), 0, L, src, pc, pn->poly(), 0, out.data().raw_pointer(), 1, s));
    if (!encrypted.is_ntt_form())
        troyn_check(troyn_ntt(context_->plan(), 1, out.data().raw_pointer(), out.data().raw_pointer(), 1, pc, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    if (cd->parms().scheme() == SchemeType::CKKS) {
        out.scale() = encrypted.scale() * plain.scale();
        if (!is_scale_within_bounds(out.scale(), cd)) throw std::invalid_argument(encrypted.is_ntt_form() ? "[Evaluator::multiply_plain_ntt] Scale out of bounds." : "[Evaluator::multiply_plain_normal] Scale out of bounds.");   // evaluator_multiply_plain.cu:63-67, :213-217
    }
    hip_check(hipStreamSynchronize(s), "stream_sync");   // the temporary NTT plaintext returns to the pool
    destination = std::move(out);
}

void Evaluator::multiply_plain_accumulate(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain,
                                          const std::vector<Ciphertext*>& destination, bool set_zero, MemoryPoolHandle pool) const {
    // evaluator_multiply_plain.cu:356-383: by the forms of the operands (the form of the first element speaks for its vector, as get_is_ntt_form_vec does)
    if (encrypted.size() != plain.size() || encrypted.size() != destination.size()) throw std::invalid_argument("[Evaluator::multiply_plain_ntt_batched] Input vectors have different sizes.");
    if (encrypted.empty()) return;
    const bool encrypted_ntt = encrypted[0]->is_ntt_form(), plain_ntt = plain[0]->is_ntt_form();
    if (encrypted_ntt && plain_ntt) { multiply_plain_ntt_accumulate(encrypted, plain, destination, set_zero, pool); return; }
    if (!encrypted_ntt && !plain_ntt) {
        // multiply_plain_normal_accumulate (:156-193): the products are formed by multiply_plain_batched and added one by one; equal destination pointers accumulate
        const char* P = "[Evaluator::multiply_plain_normal_batched]";
        for (const Ciphertext* c : encrypted) check_no_seed(P, *c);
        if (set_zero) {
            for (size_t i = 0; i < encrypted.size(); i++) *destination[i] = Ciphertext::like(*encrypted[i], true, pool);
        } else {
            for (size_t i = 0; i < encrypted.size(); i++)
                if (destination[i]->parms_id() != encrypted[i]->parms_id() || destination[i]->is_ntt_form() != encrypted[i]->is_ntt_form())
                    throw std::invalid_argument("[Evaluator::multiply_plain_normal_accumulate] Destination parameters do not match.");
        }
        std::vector<Ciphertext> temp(encrypted.size());
        multiply_plain_batched(encrypted, plain, batch_utils::collect_pointer(temp), pool);
        for (size_t i = 0; i < encrypted.size(); i++) { destination[i]->scale() = temp[i].scale(); add_inplace(*destination[i], temp[i], pool); }
        return;
    }
    if (encrypted_ntt) {       // the plaintexts are brought to NTT form at the ciphertexts' level
        std::vector<Plaintext> moved(plain.size());
        transform_plain_to_ntt_batched(plain, encrypted[0]->parms_id(), batch_utils::collect_pointer(moved), pool);
        multiply_plain_ntt_accumulate(encrypted, batch_utils::collect_const_pointer(moved), destination, set_zero, pool);
        return;
    }
    std::vector<Ciphertext> moved(encrypted.size());       // coefficient-form ciphertexts, NTT-form plaintexts: through the NTT domain and back
    transform_to_ntt_batched(encrypted, batch_utils::collect_pointer(moved), pool);
    multiply_plain_ntt_accumulate(batch_utils::collect_const_pointer(moved), plain, destination, set_zero, pool);
    std::vector<Ciphertext*> distinct;
    for (Ciphertext* d : destination) if (std::find(distinct.begin(), distinct.end(), d) == distinct.end()) distinct.push_back(d);
    transform_from_ntt_inplace_batched(distinct, pool);
}

void Evaluator::multiply_plain_ntt_accumulate(const std::vector<const Ciphertext*>& encrypted, const std::vector<const Plaintext*>& plain,
                                              const std::vector<Ciphertext*>& destination, bool set_zero, MemoryPoolHandle pool) const {
    // evaluator_multiply_plain.cu:258-307 (multiply_plain_ntt_accumulate): all operands in NTT form, same parms_id
    const char* P = "[Evaluator::multiply_plain_ntt_batched]";
    if (encrypted.size() != plain.size() || encrypted.size() != destination.size()) throw std::invalid_argument(std::string(P) + " Input vectors have different sizes.");
    if (encrypted.empty()) return;
    const ParmsID pid = encrypted[0]->parms_id();
    const size_t pc = encrypted[0]->polynomial_count();
    auto cd = get_context_data("[Evaluator::multiply_plain_ntt]", pid);
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    for (size_t i = 0; i < encrypted.size(); i++) {
        check_no_seed(P, *encrypted[i]);
        check_on_device(P, context_, *encrypted[i]);
        if (encrypted[i]->parms_id() != pid || encrypted[i]->polynomial_count() != pc || !encrypted[i]->is_ntt_form())
            throw std::invalid_argument(std::string(P) + " Ciphertexts must share parms_id, size and NTT form.");
        if (!plain[i]->is_ntt_form() || plain[i]->parms_id() != pid || !plain[i]->on_device())
            throw std::invalid_argument("[Evaluator::multiply_plain_ntt] Plaintext and ciphertext parameters do not match.");
    }
    if (set_zero) {
        for (size_t i = 0; i < destination.size(); i++) {
            bool seen = false;
            for (size_t k = 0; k < i && !seen; k++) seen = destination[k] == destination[i];
            if (!seen) *destination[i] = Ciphertext::like(*encrypted[i], false, pool);
        }
    } else {
        for (size_t i = 0; i < destination.size(); i++)
            if (destination[i]->parms_id() != pid || !destination[i]->is_ntt_form() || destination[i]->polynomial_count() != pc)
                throw std::invalid_argument("[Evaluator::multiply_plain_normal_accumulate] Destination parameters do not match.");
    }
    std::vector<const uint64_t*> cts(encrypted.size()), pts(encrypted.size());
    std::vector<uint64_t*> dsts(encrypted.size());
    for (size_t i = 0; i < encrypted.size(); i++) {
        cts[i] = encrypted[i]->data().raw_pointer(); pts[i] = plain[i]->poly(); dsts[i] = destination[i]->data().raw_pointer();
    }
    const size_t wsb = troyn_multiply_plain_accumulate_workspace_bytes(cts.size());
    utils::DynamicArray ws((wsb + 7) / 8, true, pool);
    troyn_check(troyn_multiply_plain_accumulate(context_->plan(), 0, L, pc, cts.data(), pts.data(), dsts.data(), cts.size(), set_zero ? 1 : 0,
                                                ws.raw_pointer(), wsb, current_stream()));
    if (cd->parms().scheme() == SchemeType::CKKS)
        for (size_t i = 0; i < encrypted.size(); i++) {
            destination[i]->scale() = encrypted[i]->scale() * plain[i]->scale();
            if (!is_scale_within_bounds(destination[i]->scale(), cd)) throw std::invalid_argument("[Evaluator::multiply_plain_ntt_batched] Scale out of bounds.");   // evaluator_multiply_plain.cu:250,:301
        }
    // asynchronous like the reference's method: the pointer table went through the library's pinned ring, `ws` returns to the pool in stream order
}

// ------------------------------------------------------------------------------------------------
// Galois automorphisms  (utils/galois.cu, evaluator_keyswitching.cu:147-361, key_generator.cu:239-260)
// ------------------------------------------------------------------------------------------------
namespace utils {

size_t galois_element_from_step(size_t n, int step) {
    // utils/galois.cu:43-63
    const size_t m = n * 2;
    if (step == 0) return m - 1;
    const bool sign = step < 0;
    const size_t pos_step = static_cast<size_t>(step < 0 ? -step : step);
    if (pos_step >= (n >> 1)) throw std::invalid_argument("[GaloisTool::get_element_from_step] Step count too large");
    const size_t true_step = sign ? ((n >> 1) - pos_step) : pos_step;
    size_t e = 1;
    for (size_t i = 0; i < true_step; i++) e = (e * 3) & (m - 1);   // GALOIS_GENERATOR = 3
    return e;
}

std::vector<size_t> galois_elements_all(size_t n) {
    // utils/galois.cu:65-89
    const size_t m = n * 2;
    std::vector<size_t> out{m - 1};
    size_t logn = 0;
    while ((size_t(1) << logn) < n) logn++;
    size_t pos = 3, neg = 1;
    for (size_t x = 1; x < m; x += 2) if (((x * 3) & (m - 1)) == 1) { neg = x; break; }   // 3^-1 mod m
    for (size_t i = 0; i + 1 < logn; i++) {
        out.push_back(pos);
        out.push_back(neg);
        pos = (pos * pos) & (m - 1);
        neg = (neg * neg) & (m - 1);
    }
    return out;
}

std::vector<int> naf(int value) {
    std::vector<int> res;
    const bool sign = value < 0;
    value = std::abs(value);
    int i = 0;
    while (value > 0) {
        const int zi = ((value & 1) != 0) ? (2 - (value & 3)) : 0;
        value = (value - zi) >> 1;
        if (zi != 0) res.push_back((sign ? -zi : zi) << i);
        i++;
    }
    return res;
}

}  // namespace utils

GaloisKeys KeyGenerator::create_galois_keys_from_elements(const std::vector<size_t>& galois_elements, bool save_seed, MemoryPoolHandle pool) const {
    // key_generator.cu:239-260: key for element g encrypts the secret key rotated by g
    ContextDataPointer kcd = context_->key_context_data().value();
    const size_t n = kcd->parms().poly_modulus_degree();
    const uint32_t K = static_cast<uint32_t>(kcd->parms().coeff_modulus().size());
    std::vector<std::vector<PublicKey>> keys(n);
    utils::DynamicArray rotated(static_cast<size_t>(K) * n, true, pool);
    for (size_t g : galois_elements) {
        if (g % 2 == 0 || g >= (n << 1)) throw std::invalid_argument("[KeyGenerator::generate_galois_keys] Galois element is not valid.");
        const size_t index = GaloisKeys::get_index(g);
        if (!keys[index].empty()) continue;
        troyn_check(troyn_apply_galois(context_->plan(), 0, K, 1, g, secret_key_.data().raw_pointer(), rotated.raw_pointer(), 1, current_stream()));
        generate_one_kswitch_key(rotated.raw_pointer(), keys[index], save_seed, pool);
    }
    return GaloisKeys(KSwitchKeys(kcd->parms_id(), std::move(keys)));
}

GaloisKeys KeyGenerator::create_galois_keys_from_steps(const std::vector<int>& steps, bool save_seed, MemoryPoolHandle pool) const {
    const size_t n = context_->key_context_data().value()->parms().poly_modulus_degree();
    std::vector<size_t> elements;
    for (int st : steps) elements.push_back(utils::galois_element_from_step(n, st));
    return create_galois_keys_from_elements(elements, save_seed, pool);
}

GaloisKeys KeyGenerator::create_galois_keys(bool save_seed, MemoryPoolHandle pool) const {
    const size_t n = context_->key_context_data().value()->parms().poly_modulus_degree();
    return create_galois_keys_from_elements(utils::galois_elements_all(n), save_seed, pool);
}

// every argument check of apply_galois (evaluator_keyswitching.cu:147-179 and the key-switch checks behind it), the key pointers and the result
// object WITHOUT the device work: shared by the method, apply_galois_batched and the call-combining rendezvous
void Evaluator::apply_galois_prepare(const Ciphertext& encrypted, size_t galois_element, const GaloisKeys& galois_keys, Ciphertext& out,
                                     std::vector<const uint64_t*>& key_ptrs, MemoryPoolHandle pool) const {
    const char* P = "[Evaluator::apply_galois_inplace]";
    check_no_seed(P, encrypted);
    check_on_device(P, context_, encrypted);
    if (galois_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Galois keys has incorrect parms id.");
    auto cd = get_context_data(P, encrypted.parms_id());
    const size_t n = cd->parms().poly_modulus_degree();
    if ((galois_element & 1) == 0 || galois_element > 2 * n) throw std::invalid_argument(std::string(P) + " Galois element is not valid.");
    if (!galois_keys.has_key(galois_element)) throw std::invalid_argument(std::string(P) + " Galois key not present.");
    if (encrypted.polynomial_count() > 2) throw std::invalid_argument(std::string(P) + " Ciphertext size must be 2.");
    out = Ciphertext::like(encrypted, false, pool);
    const size_t idx = GaloisKeys::get_index(galois_element);
    switch_key_checks(encrypted, galois_keys, idx, out);
    key_ptrs = galois_keys.get_data_ptrs(idx);
}

void Evaluator::apply_galois(const Ciphertext& encrypted, size_t galois_element, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    Ciphertext out;
    std::vector<const uint64_t*> ptrs;
    apply_galois_prepare(encrypted, galois_element, galois_keys, out, ptrs, pool);
    auto cd = get_context_data("[Evaluator::apply_galois_inplace]", encrypted.parms_id());
    const size_t n = cd->parms().poly_modulus_degree();
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    if (cd->parms().scheme() != SchemeType::BGV && encrypted.polynomial_count() == 2 && detail::combining_wanted()) {
        // call combining (troy.h): the rotations of concurrent threads by the same element run as one permutation + one key switch
        detail::CombineRequest rq;
        rq.kind = detail::CombineKind::ApplyGalois; rq.handle = context_->plan(); rq.L = L; rq.p1 = 2; rq.p2 = static_cast<uint32_t>(galois_element);
        rq.ckks = cd->parms().scheme() == SchemeType::CKKS; rq.ntt_form = encrypted.is_ntt_form(); rq.keys = &ptrs;
        rq.in1 = encrypted.data().raw_pointer(); rq.words1 = encrypted.data().size();
        rq.out = out.data().raw_pointer(); rq.out_words = out.data().size();
        if (detail::combine_submit(rq, pool)) { destination = std::move(out); return; }
    }
    troyn_check(troyn_apply_galois(context_->plan(), 0, L, encrypted.is_ntt_form() ? 1 : 0, galois_element,
                                   encrypted.data().raw_pointer(), out.data().raw_pointer(), 2, current_stream()));
    // the permuted c1 is the key-switch target; the result overwrites it (c0 += ks0, c1 = ks1)
    utils::DynamicArray target(static_cast<size_t>(L) * n, true, pool);
    hip_check(hipMemcpyAsync(target.raw_pointer(), out.poly(1), static_cast<size_t>(L) * n * 8, hipMemcpyDeviceToDevice, current_stream()), "copy_device_to_device");
    switch_key_internal(encrypted, target.raw_pointer(), galois_keys, GaloisKeys::get_index(galois_element),
                        SwitchKeyDestinationAssignMethod::OverwriteExceptFirst, out, pool);
    // (no stream wait: the temporaries return to the pool in stream order, the call is asynchronous like the reference's)
    destination = std::move(out);
}

void Evaluator::rotate_internal(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    // evaluator_keyswitching.cu:263-294
    const char* P = "[Evaluator::rotate_inplace_internal]";
    auto cd = get_context_data(P, encrypted.parms_id());
    if (galois_keys.parms_id() != context_->key_parms_id()) throw std::invalid_argument(std::string(P) + " Galois keys has incorrect parms id.");
    if (steps == 0) { destination = encrypted; return; }
    const size_t n = cd->parms().poly_modulus_degree();
    const size_t element = utils::galois_element_from_step(n, steps);
    if (galois_keys.has_key(element)) { apply_galois(encrypted, element, galois_keys, destination, pool); return; }
    std::vector<int> naf_steps = utils::naf(steps);
    if (naf_steps.size() == 1) throw std::invalid_argument(std::string(P) + " Galois key not present.");
    bool first = true;
    for (int st : naf_steps) {
        if (first) { rotate_internal(encrypted, st, galois_keys, destination, pool); first = false; }
        else { Ciphertext temp; rotate_internal(destination, st, galois_keys, temp, pool); destination = std::move(temp); }
    }
}

void Evaluator::rotate_rows(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::invalid_argument("[Evaluator::rotate_rows_inplace] Rotate rows only applies for BFV or BGV");
    rotate_internal(encrypted, steps, galois_keys, destination, pool);
}

void Evaluator::rotate_columns(const Ciphertext& encrypted, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::invalid_argument("[Evaluator::rotate_columns_inplace] Rotate columns only applies for BFV or BGV");
    auto cd = get_context_data("[Evaluator::conjugate_inplace_internal]", encrypted.parms_id());
    apply_galois(encrypted, utils::galois_element_from_step(cd->parms().poly_modulus_degree(), 0), galois_keys, destination, pool);
}

// ------------------------------------------------------------------------------------------------
// Serialization  (utils/serialize.h, ciphertext.cu:93-210, plaintext.cu:20-70, kswitch_keys.cu:5-55,
// encryption_parameters.cu:53-112): raw little-endian fields in the reference's order, so files are interchangeable.
// ------------------------------------------------------------------------------------------------
namespace {
template <typename T> void put(std::ostream& os, const T& v) { os.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> void get(std::istream& is, T& v) {
    is.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!is) throw std::runtime_error("[serialize::load_object] unexpected end of stream");
}
void put_words(std::ostream& os, const std::vector<uint64_t>& v, size_t count) { os.write(reinterpret_cast<const char*>(v.data()), count * 8); }
// ---- zstd (utils/compression.h, compression_zstd.cpp, serialize.h:59-110) ---------------------------------------------------------------------------
// The reference compiles zstd in from a submodule (TROY_ZSTD).  Neither the submodule nor <zstd.h> exists in this image, but the zstd RUNTIME library does
// (libzstd.so.1): it is loaded on first use and the handful of entry points of its stable C ABI are declared here.  When it cannot be loaded, Zstd is refused at
// run time as before and compression::available(Zstd) is false.  Framing as the reference's: [Zstd][compressed size][one zstd frame] when that is shorter than the
// raw object, else [Nil][raw]; a reader accepts either.  (Frames are interchangeable, not byte-identical: the reference streams chunks through ZSTD_compressStream2,
// this writes one frame per object; any zstd decoder reads both.)
struct ZstdRuntime {
    struct InBuffer { const void* src; size_t size; size_t pos; };
    struct OutBuffer { void* dst; size_t size; size_t pos; };
    size_t (*compress_bound)(size_t) = nullptr;
    size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    const char* (*error_name)(size_t) = nullptr;
    void* (*create_dctx)() = nullptr;
    size_t (*free_dctx)(void*) = nullptr;
    size_t (*decompress_stream)(void*, OutBuffer*, InBuffer*) = nullptr;
    size_t (*dstream_out_size)() = nullptr;
    bool ok = false;
    ZstdRuntime() {
        void* h = nullptr;
        for (const char* name : {"libzstd.so.1", "libzstd.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        auto sym = [&](const char* n) { return dlsym(h, n); };
        compress_bound = reinterpret_cast<decltype(compress_bound)>(sym("ZSTD_compressBound"));
        compress = reinterpret_cast<decltype(compress)>(sym("ZSTD_compress"));
        is_error = reinterpret_cast<decltype(is_error)>(sym("ZSTD_isError"));
        error_name = reinterpret_cast<decltype(error_name)>(sym("ZSTD_getErrorName"));
        create_dctx = reinterpret_cast<decltype(create_dctx)>(sym("ZSTD_createDCtx"));
        free_dctx = reinterpret_cast<decltype(free_dctx)>(sym("ZSTD_freeDCtx"));
        decompress_stream = reinterpret_cast<decltype(decompress_stream)>(sym("ZSTD_decompressStream"));
        dstream_out_size = reinterpret_cast<decltype(dstream_out_size)>(sym("ZSTD_DStreamOutSize"));
        ok = compress_bound && compress && is_error && error_name && create_dctx && free_dctx && decompress_stream && dstream_out_size;
    }
};
const ZstdRuntime& zstd_runtime() { static const ZstdRuntime* z = new ZstdRuntime(); return *z; }
const ZstdRuntime& zstd_or_throw(const char* prompt) {
    const ZstdRuntime& z = zstd_runtime();
    if (!z.ok) throw std::invalid_argument(std::string(prompt) + " Zstd is not available: libzstd.so.1 could not be loaded.");
    return z;
}
std::string zstd_compress(const char* raw, size_t size) {
    const ZstdRuntime& z = zstd_or_throw("[serialize::compress]");
    std::string out(z.compress_bound(size), '\0');
    const size_t n = z.compress(&out[0], out.size(), raw, size, 3);      // ZSTD_CLEVEL_DEFAULT, as compression_zstd.cpp:38
    if (z.is_error(n)) throw std::runtime_error(std::string("[utils::compression::zstd::compress] ") + z.error_name(n));
    out.resize(n);
    return out;
}
std::string zstd_decompress(const char* data, size_t size) {
    // streaming: a frame written by the reference carries no content size
    const ZstdRuntime& z = zstd_or_throw("[serialize::decompress]");
    void* ctx = z.create_dctx();
    if (!ctx) throw std::runtime_error("[utils::compression::zstd::decompress] ZSTD_createDCtx() failed");
    std::string out, chunk(z.dstream_out_size(), '\0');
    ZstdRuntime::InBuffer in{data, size, 0};
    size_t state = 1;
    while (in.pos < in.size || state != 0) {
        ZstdRuntime::OutBuffer ob{&chunk[0], chunk.size(), 0};
        state = z.decompress_stream(ctx, &ob, &in);
        if (z.is_error(state)) { z.free_dctx(ctx); throw std::runtime_error(std::string("[utils::compression::zstd::decompress] ") + z.error_name(state)); }
        out.append(chunk.data(), ob.pos);
        if (in.pos == in.size && ob.pos < ob.size) {
            if (state != 0) { z.free_dctx(ctx); throw std::runtime_error("[utils::compression::zstd::decompress] truncated frame"); }
            break;
        }
    }
    z.free_dctx(ctx);
    return out;
}
// serialize.h:59-89: `save_nil` writes the object with CompressionMode::Nil (mode byte + raw fields); the Zstd form replaces that by [Zstd][size][frame] when shorter
template <typename F> size_t save_framed(std::ostream& os, CompressionMode mode, F save_nil) {
    if (mode == CompressionMode::Nil) return save_nil(os);
    std::ostringstream plain;
    save_nil(plain);
    const std::string raw = plain.str();                          // raw[0] is the Nil mode byte
    const std::string packed = zstd_compress(raw.data() + 1, raw.size() - 1);
    if (packed.size() < raw.size() - 1) {
        put(os, CompressionMode::Zstd);
        put(os, packed.size());
        os.write(packed.data(), static_cast<std::streamsize>(packed.size()));
        return packed.size() + sizeof(CompressionMode) + sizeof(size_t);
    }
    os.write(raw.data(), static_cast<std::streamsize>(raw.size()));
    return raw.size();
}
void put_mode(std::ostream& os, CompressionMode mode) {          // the Nil writers only
    if (mode != CompressionMode::Nil) throw std::logic_error("[serialize::compress] internal: a Nil writer was asked for another mode");
    put(os, mode);
}
// serialize.h:91-108: reads the mode; for a compressed object the frame is unpacked into `unpacked` (which then holds the raw fields) and that stream is returned,
// otherwise `is` itself
std::istream& get_mode(std::istream& is, std::istringstream& unpacked) {
    CompressionMode mode;
    get(is, mode);
    if (mode == CompressionMode::Nil) return is;
    if (mode != CompressionMode::Zstd) throw std::runtime_error("Invalid compression mode");
    size_t size;
    get(is, size);
    if (size > (size_t(1) << 40)) throw std::runtime_error("[serialize::decompress] invalid compressed size");
    std::string packed(size, '\0');
    is.read(&packed[0], static_cast<std::streamsize>(size));
    if (!is) throw std::runtime_error("[serialize::decompress] unexpected end of stream");
    unpacked.str(zstd_decompress(packed.data(), packed.size()));
    unpacked.clear();
    return unpacked;
}
bool get_bool(std::istream& is) {
    unsigned char c;
    get(is, c);
    if (c > 1) throw std::runtime_error("Invalid bool value");
    return c != 0;
}
}  // namespace

namespace utils { namespace compression { bool available(CompressionMode mode) { return mode == CompressionMode::Nil || (mode == CompressionMode::Zstd && zstd_runtime().ok); } } }

size_t EncryptionParameters::save(std::ostream& stream) const {
    put(stream, scheme_);
    put(stream, poly_modulus_degree_);
    put(stream, coeff_modulus_.size());
    for (const Modulus& m : coeff_modulus_) put(stream, m.value());
    size_t bytes = sizeof(SchemeType) + 2 * sizeof(size_t) + coeff_modulus_.size() * 8 + sizeof(bool);
    if (scheme_ == SchemeType::BFV || scheme_ == SchemeType::BGV) { put(stream, plain_modulus_.value()); bytes += 8; }
    put(stream, use_special_prime_for_encryption_);
    return bytes;
}

void EncryptionParameters::load(std::istream& stream) {
    get(stream, scheme_);
    size_t n, k;
    get(stream, n); get(stream, k);
    if (k > 64) throw std::runtime_error("[EncryptionParameters::load] invalid coefficient modulus count");
    poly_modulus_degree_ = n;
    coeff_modulus_.clear();
    for (size_t i = 0; i < k; i++) { uint64_t v; get(stream, v); coeff_modulus_.push_back(Modulus(v)); }
    if (scheme_ == SchemeType::BFV || scheme_ == SchemeType::BGV) { uint64_t v; get(stream, v); plain_modulus_ = Modulus(v); }
    use_special_prime_for_encryption_ = get_bool(stream);
    compute_parms_id();
}

// serialize.h:46-52: for Zstd the bound of the compressed form, or the raw form when that is what gets written
static size_t framed_size_upperbound(size_t nil_size, CompressionMode mode) {
    if (mode == CompressionMode::Nil) return nil_size;
    const size_t raw = nil_size - sizeof(CompressionMode);
    return std::max(zstd_or_throw("[serialize::serialized_size_upperbound]").compress_bound(raw) + sizeof(CompressionMode) + sizeof(size_t), nil_size);
}

size_t Plaintext::serialized_size_upperbound(CompressionMode mode) const {
    return framed_size_upperbound(sizeof(CompressionMode) + sizeof(ParmsID) + sizeof(double) + 2 * sizeof(size_t) + 2 * sizeof(bool) + data_.size() * 8 + 2 * sizeof(size_t), mode);
}

size_t Plaintext::save(std::ostream& stream, CompressionMode mode) const {
    if (mode != CompressionMode::Nil) return save_framed(stream, mode, [&](std::ostream& os) { return save(os, CompressionMode::Nil); });
    put_mode(stream, mode);
    put(stream, parms_id_);
    put(stream, scale_);
    put(stream, coeff_count_);
    put(stream, static_cast<unsigned char>(on_device()));
    put(stream, data_.size());
    put_words(stream, data_.to_vector(), data_.size());
    put(stream, static_cast<unsigned char>(is_ntt_form_));
    put(stream, poly_modulus_degree_);
    put(stream, coeff_modulus_size_);
    return serialized_size_upperbound(mode);
}

void Plaintext::load(std::istream& outer, MemoryPoolHandle pool) {
    std::istringstream unpacked;
    std::istream& stream = get_mode(outer, unpacked);
    get(stream, parms_id_);
    get(stream, scale_);
    get(stream, coeff_count_);
    const bool device = get_bool(stream);
    size_t size;
    get(stream, size);
    std::vector<uint64_t> words(size);
    stream.read(reinterpret_cast<char*>(words.data()), size * 8);
    if (!stream) throw std::runtime_error("[Plaintext::load] unexpected end of stream");
    data_ = utils::DynamicArray::from_vector(words);
    if (device) data_.to_device_inplace(pool);
    is_ntt_form_ = get_bool(stream);
    get(stream, poly_modulus_degree_);
    get(stream, coeff_modulus_size_);
}

static SchemeType context_scheme(const HeContextPointer& context) { return context->key_context_data().value()->parms().scheme(); }

size_t Ciphertext::serialized_size_upperbound(HeContextPointer context, CompressionMode mode) const {
    size_t bytes = sizeof(CompressionMode) + sizeof(ParmsID) + 3 * sizeof(size_t) + 1;
    const SchemeType scheme = context_scheme(context);
    if (scheme == SchemeType::CKKS) bytes += sizeof(double);
    if (scheme == SchemeType::BGV) bytes += sizeof(uint64_t);
    const size_t poly = poly_modulus_degree_ * coeff_modulus_size_;
    bytes += contains_seed() ? 8 + poly * 8 : poly * polynomial_count_ * 8;
    return framed_size_upperbound(bytes, mode);
}

// ---- pinned staging image of the wire path (one per host thread, grown on demand, reused) -------------------------------------------------------
// save: headers are written into it, the payloads arrive by asynchronous device-to-host copies at their final offsets, ONE wait, ONE stream.write.
// load: the payloads are read from the caller's stream straight into it and leave by asynchronous host-to-device copies; `busy` marks the last copy that
// reads it, the next user of the image waits for that event (the loading call itself does not wait: it is asynchronous like the reference's).
namespace {
struct PinnedImage {
    char* p = nullptr; size_t cap = 0; hipEvent_t busy = nullptr; bool pending = false; int busy_device = -1;   // an event belongs to the device it was created on
    ~PinnedImage() { if (p) { (void)hipHostFree(p); (void)hipGetLastError(); } if (busy) { (void)hipEventDestroy(busy); (void)hipGetLastError(); } }
    void quiesce() { if (pending) { hip_check(hipEventSynchronize(busy), "event_sync"); pending = false; } }
    char* reserve(size_t bytes) {
        quiesce();
        if (bytes > cap) {
            if (p) { hip_check(hipHostFree(p), "host_free"); p = nullptr; cap = 0; }
            const size_t want = std::max<size_t>(bytes + bytes / 4, size_t(1) << 20);
            void* q = nullptr;
            hip_check(hipHostMalloc(&q, want, hipHostMallocPortable), "host_malloc");
            p = static_cast<char*>(q); cap = want;
        }
        return p;
    }
    // the copies that read (or fill) the image were queued on `s`, a stream of the CURRENT device: the event is (re)created there if the thread moved to another device
    void mark(hipStream_t s) {
        int dev = 0;
        hip_check(hipGetDevice(&dev), "get_device");
        if (busy && dev != busy_device) { hip_check(hipEventDestroy(busy), "event_destroy"); busy = nullptr; }      // nothing pending: reserve() waited
        if (!busy) { hip_check(hipEventCreateWithFlags(&busy, hipEventDisableTiming), "event_create"); busy_device = dev; }
        hip_check(hipEventRecord(busy, s), "event_record");
        pending = true;
    }
};
PinnedImage& pinned_image() { static thread_local PinnedImage img; return img; }
}  // namespace

size_t Ciphertext::save(std::ostream& stream, HeContextPointer context, CompressionMode mode) const {
    const Ciphertext* one = this;
    return save_many(stream, &one, 1, context, mode);
}

size_t Ciphertext::save_many(std::ostream& stream, const Ciphertext* const* cts, size_t count, HeContextPointer context, CompressionMode mode) {
    if (mode != CompressionMode::Nil) {      // every object is its own frame, as when the reference saves them one by one (serialize.h:59-89)
        size_t total = 0;
        for (size_t i = 0; i < count; i++) {
            const Ciphertext* one = cts[i];
            total += save_framed(stream, mode, [&](std::ostream& os) { return save_many(os, &one, 1, context, CompressionMode::Nil); });
        }
        return total;
    }
    // ciphertext.cu:93-150 per object: mode, parms_id, the three sizes, flags, scale (CKKS) / correction factor (BGV), seed, words
    const SchemeType scheme = context_scheme(context);
    std::vector<std::string> headers(count);
    std::vector<size_t> payload(count);
    size_t total = 0, bound = 0;
    for (size_t i = 0; i < count; i++) {
        const Ciphertext& c = *cts[i];
        std::ostringstream h;
        put_mode(h, mode);
        put(h, c.parms_id_);
        put(h, c.polynomial_count_);
        put(h, c.coeff_modulus_size_);
        put(h, c.poly_modulus_degree_);
        const unsigned char flags = static_cast<unsigned char>(c.is_ntt_form_) | static_cast<unsigned char>(c.contains_seed() << 1) | static_cast<unsigned char>(c.on_device() << 2);
        put(h, flags);
        if (scheme == SchemeType::CKKS) put(h, c.scale_);
        if (scheme == SchemeType::BGV) put(h, c.correction_factor_);
        if (c.contains_seed()) {
            if (c.polynomial_count_ != 2) throw std::logic_error("[Ciphertext::save] Ciphertext contains seed but polynomial count is not 2.");
            put(h, c.seed_);
            payload[i] = c.poly_modulus_degree_ * c.coeff_modulus_size_;     // c0 only; c1 is regenerated from the seed
        } else {
            payload[i] = c.data_.size();
        }
        if (payload[i] > c.data_.size()) throw std::logic_error("[Ciphertext::save] Ciphertext data is smaller than its shape.");
        headers[i] = h.str();
        total += headers[i].size() + payload[i] * 8;
        bound += c.serialized_size_upperbound(context, mode);
    }
    if (count == 0) return 0;
    PinnedImage& img = pinned_image();
    char* base = img.reserve(total);
    size_t off = 0;
    bool queued = false;
    size_t queued_device = 0;
    for (size_t i = 0; i < count; i++) {
        const Ciphertext& c = *cts[i];
        std::memcpy(base + off, headers[i].data(), headers[i].size());
        off += headers[i].size();
        if (payload[i]) {
            if (c.on_device()) {
                // (a batch that spans devices: the copies queued on one device's stream are waited for before the thread moves to the next device)
                if (queued && c.data_.device_index() != queued_device) { hip_check(stream_wait(), "copy_device_to_host"); queued = false; }
                queued_device = c.data_.device_index();
                hip_check(hipSetDevice(static_cast<int>(c.data_.device_index())), "copy_device_to_host");
                hip_check(hipMemcpyAsync(base + off, c.data_.raw_pointer(), payload[i] * 8, hipMemcpyDeviceToHost, current_stream()), "copy_device_to_host");
                queued = true;
            } else {
                std::memcpy(base + off, c.data_.raw_pointer(), payload[i] * 8);
            }
        }
        off += payload[i] * 8;
    }
    if (queued) hip_check(stream_wait(), "copy_device_to_host");
    stream.write(base, static_cast<std::streamsize>(total));
    return bound;
}

void Ciphertext::load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool) {
    Ciphertext* one = this;
    load_many(stream, &one, 1, context, pool);
}

void Ciphertext::load_many(std::istream& outer, Ciphertext* const* cts, size_t count, HeContextPointer context, MemoryPoolHandle pool) {
    // ciphertext.cu:152-210 per object.  Device-bound payloads (flag bit 2, or seeded: the seed is expanded on the device) go through the pinned
    // image; the seeded c1 polynomials of one shape are expanded by ONE troyn_sample_uniform_multi launch into a block the ciphertexts window.
    if (count == 0) return;
    if (pool) hip_check(hipSetDevice(static_cast<int>(pool->get_device())), "copy_host_to_device");      // the pool's device: where the arrays live and the copies are queued
    const SchemeType scheme = context_scheme(context);
    struct Pending { Ciphertext* c; size_t words, img_off; bool seeded; };
    std::vector<Pending> dev;
    PinnedImage& img = pinned_image();
    size_t used = 0;
    char* base = img.reserve(size_t(1) << 20);
    auto flush = [&]() {
        // seeded ciphertexts of one shape become windows of ONE block [count][2][poly] (as Encryptor::encrypt_symmetric_packed hands them out): their c1
        // polynomials are expanded by one troyn_sample_uniform_multi launch (the c1 generators of rlwe::symmetric, utils/rlwe.cu:262-266; ciphertext.cu:79-91)
        // and dealt to the windows by one strided copy; everything else gets its own array
        std::vector<Pending*> seeded;
        for (Pending& d : dev) {
            if (d.seeded) seeded.push_back(&d);
            else d.c->data_ = utils::DynamicArray(d.words, true, pool);
        }
        while (!seeded.empty()) {
            const Ciphertext& f = *seeded.front()->c;
            std::vector<Pending*> group, rest;
            for (Pending* d : seeded) (d->c->parms_id_ == f.parms_id_ && d->c->coeff_modulus_size_ == f.coeff_modulus_size_ && d->c->poly_modulus_degree_ == f.poly_modulus_degree_ ? group : rest).push_back(d);
            auto cd = context->get_context_data(f.parms_id_);
            if (!cd.has_value()) throw std::invalid_argument("[Ciphertext::expand_seed] ParmsID is not valid.");
            if (!context->on_device()) throw std::invalid_argument("[Ciphertext::expand_seed] the seed is expanded on the GPU: context and ciphertext must be on the device.");
            const size_t poly = f.poly_modulus_degree_ * f.coeff_modulus_size_, m = group.size();
            std::vector<uint64_t> seeds(2 * m, 0);
            for (size_t g = 0; g < m; g++) seeds[2 * g] = group[g]->c->seed_;
            auto shared = std::make_shared<utils::DynamicArray>(m * 2 * poly, true, pool);
            for (size_t g = 0; g < m; g++) group[g]->c->data_ = utils::DynamicArray::device_view(shared->raw_pointer() + g * 2 * poly, 2 * poly, shared);
            if (m == 1) {
                troyn_check(troyn_sample_uniform_multi(context->plan(), static_cast<uint32_t>(f.coeff_modulus_size_), seeds.data(), shared->raw_pointer() + poly, 1, current_stream()));
            } else {
                utils::DynamicArray block(m * poly, true, pool);
                troyn_check(troyn_sample_uniform_multi(context->plan(), static_cast<uint32_t>(f.coeff_modulus_size_), seeds.data(), block.raw_pointer(), m, current_stream()));
                hip_check(hipMemcpy2DAsync(shared->raw_pointer() + poly, 2 * poly * 8, block.raw_pointer(), poly * 8, poly * 8, m, hipMemcpyDeviceToDevice, current_stream()), "copy_device_to_device");
                // (`block` returns to the pool in stream order)
            }
            for (Pending* d : group) d->c->seed_ = 0;
            seeded.swap(rest);
        }
        for (const Pending& d : dev)
            hip_check(hipMemcpyAsync(d.c->data_.raw_pointer(), base + d.img_off, d.words * 8, hipMemcpyHostToDevice, current_stream()), "copy_host_to_device");
        if (!dev.empty()) img.mark(current_stream());      // the image may be overwritten once these copies have run
        dev.clear();
        used = 0;
    };
    for (size_t i = 0; i < count; i++) {
        Ciphertext& c = *cts[i];
        std::istringstream unpacked;
        std::istream& stream = get_mode(outer, unpacked);      // a compressed object is read from its unpacked copy, the others from the caller's stream
        get(stream, c.parms_id_);
        get(stream, c.polynomial_count_);
        get(stream, c.coeff_modulus_size_);
        get(stream, c.poly_modulus_degree_);
        unsigned char flags;
        get(stream, flags);
        c.is_ntt_form_ = flags & 1;
        const bool seeded = flags & 2, device = flags & 4;
        if (flags & 8) throw std::logic_error("[Ciphertext::load] Trying to call load with ciphertext with only terms saved.");
        c.scale_ = 1.0; c.correction_factor_ = 1;
        if (scheme == SchemeType::CKKS) get(stream, c.scale_);
        if (scheme == SchemeType::BGV) get(stream, c.correction_factor_);
        const size_t poly = c.poly_modulus_degree_ * c.coeff_modulus_size_;
        if (c.poly_modulus_degree_ > 131072 || c.coeff_modulus_size_ > 64 || c.polynomial_count_ > 64) throw std::runtime_error("[Ciphertext::load] invalid shape");
        if (seeded && c.polynomial_count_ != 2) throw std::runtime_error("[Ciphertext::load] a seeded ciphertext has two polynomials");
        c.seed_ = 0;
        if (seeded) get(stream, c.seed_);
        const size_t words = seeded ? poly : poly * c.polynomial_count_;
        if (device || seeded) {
            if ((used + words * 8) > img.cap) {
                if (!dev.empty()) { flush(); }
                base = img.reserve(std::max<size_t>(words * 8, std::min<size_t>(size_t(64) << 20, (count - i) * words * 8)));
            }
            stream.read(base + used, static_cast<std::streamsize>(words * 8));
            if (!stream) throw std::runtime_error("[Ciphertext::load] unexpected end of stream");
            dev.push_back(Pending{&c, words, used, seeded});      // (its device array is made by flush())
            used += words * 8;
        } else {
            c.data_ = utils::DynamicArray(words, false);
            stream.read(reinterpret_cast<char*>(c.data_.raw_pointer()), static_cast<std::streamsize>(words * 8));
            if (!stream) throw std::runtime_error("[Ciphertext::load] unexpected end of stream");
        }
    }
    flush();
}

size_t Ciphertext::serialized_terms_size_upperbound(HeContextPointer context, size_t terms_count, CompressionMode mode) const {
    // ciphertext.cu:341-365
    size_t bytes = sizeof(CompressionMode) + sizeof(ParmsID) + 3 * sizeof(size_t) + 1;
    const SchemeType scheme = context_scheme(context);
    if (scheme == SchemeType::CKKS) bytes += sizeof(double);
    if (scheme == SchemeType::BGV) bytes += sizeof(uint64_t);
    const size_t poly = poly_modulus_degree_ * coeff_modulus_size_;
    if (contains_seed()) bytes += 8;
    bytes += terms_count * coeff_modulus_size_ * 8;
    bytes += contains_seed() ? 0 : poly * (polynomial_count_ - 1) * 8;
    return framed_size_upperbound(bytes, mode);
}

size_t Ciphertext::save_terms(std::ostream& stream, HeContextPointer context, const std::vector<size_t>& terms, MemoryPoolHandle pool, CompressionMode mode) const {
    // ciphertext.cu:213-280: as save(), flag bit 3 set, and of c0 only the coefficients `terms` (taken in coefficient form)
    if (mode != CompressionMode::Nil) return save_framed(stream, mode, [&](std::ostream& os) { return save_terms(os, context, terms, pool, CompressionMode::Nil); });
    put_mode(stream, mode);
    put(stream, parms_id_);
    put(stream, polynomial_count_);
    put(stream, coeff_modulus_size_);
    put(stream, poly_modulus_degree_);
    unsigned char flags = static_cast<unsigned char>(is_ntt_form_) | static_cast<unsigned char>(contains_seed() << 1) | static_cast<unsigned char>(on_device() << 2) | 8;
    put(stream, flags);
    const SchemeType scheme = context_scheme(context);
    if (scheme == SchemeType::CKKS) put(stream, scale_);
    if (scheme == SchemeType::BGV) put(stream, correction_factor_);
    const size_t n = poly_modulus_degree_, L = coeff_modulus_size_, poly = n * L;
    if (contains_seed()) {
        if (polynomial_count_ != 2) throw std::logic_error("[Ciphertext::save] Ciphertext contains seed but polynomial count is not 2.");
        put(stream, seed_);
    }
    for (size_t t : terms) if (t >= n) throw std::invalid_argument("[Ciphertext::save_terms] term out of range");
    std::vector<uint64_t> c0;
    if (is_ntt_form_) {
        if (!on_device() || !context->on_device()) throw std::invalid_argument("[Ciphertext::save_terms] an NTT-form ciphertext is transformed on the GPU: context and ciphertext must be on the device.");
        utils::DynamicArray tmp(poly, true, pool);
        troyn_check(troyn_ntt(context->plan(), 1, data_.raw_pointer(), tmp.raw_pointer(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, current_stream()));
        c0 = tmp.to_vector();
    } else {
        utils::DynamicArray view = data_.clone(pool);
        c0 = view.to_vector();
        c0.resize(poly);
    }
    for (size_t j = 0; j < L; j++)
        for (size_t t : terms) put(stream, c0[j * n + t]);
    const size_t start = contains_seed() ? 2 : 1;
    if (polynomial_count_ > start) {
        const std::vector<uint64_t> words = data_.to_vector();
        stream.write(reinterpret_cast<const char*>(words.data() + start * poly), (polynomial_count_ - start) * poly * 8);
    }
    return serialized_terms_size_upperbound(context, terms.size(), mode);
}

void Ciphertext::load_terms(std::istream& outer, HeContextPointer context, const std::vector<size_t>& terms, MemoryPoolHandle pool) {
    // ciphertext.cu:282-339: the coefficients of c0 that were not saved are zero
    std::istringstream unpacked;
    std::istream& stream = get_mode(outer, unpacked);
    get(stream, parms_id_);
    get(stream, polynomial_count_);
    get(stream, coeff_modulus_size_);
    get(stream, poly_modulus_degree_);
    unsigned char flags;
    get(stream, flags);
    is_ntt_form_ = flags & 1;
    const bool seeded = flags & 2, device = flags & 4;
    if (!(flags & 8)) throw std::logic_error("[Ciphertext::load_terms] Trying to call load_terms with ciphertext with all terms saved.");
    const SchemeType scheme = context_scheme(context);
    scale_ = 1.0; correction_factor_ = 1;
    if (scheme == SchemeType::CKKS) get(stream, scale_);
    if (scheme == SchemeType::BGV) get(stream, correction_factor_);
    const size_t n = poly_modulus_degree_, L = coeff_modulus_size_, poly = n * L;
    if (n > 131072 || L > 64 || polynomial_count_ > 64 || polynomial_count_ < 2) throw std::runtime_error("[Ciphertext::load] invalid shape");
    std::vector<uint64_t> words(poly * (seeded ? 2 : polynomial_count_), 0);
    seed_ = 0;
    if (seeded) get(stream, seed_);
    for (size_t t : terms) if (t >= n) throw std::invalid_argument("[Ciphertext::load_terms] term out of range");
    for (size_t j = 0; j < L; j++)
        for (size_t t : terms) get(stream, words[j * n + t]);
    const size_t start = seeded ? 2 : 1;
    if (polynomial_count_ > start) {
        stream.read(reinterpret_cast<char*>(words.data() + start * poly), (polynomial_count_ - start) * poly * 8);
        if (!stream) throw std::runtime_error("[Ciphertext::load] unexpected end of stream");
    }
    data_ = utils::DynamicArray::from_vector(words);
    if (device || seeded || is_ntt_form_) data_.to_device_inplace(pool);
    if (is_ntt_form_) {
        if (!context->on_device()) throw std::invalid_argument("[Ciphertext::load_terms] an NTT-form ciphertext is transformed on the GPU: the context must be on the device.");
        troyn_check(troyn_ntt(context->plan(), 0, data_.raw_pointer(), data_.raw_pointer(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, current_stream()));
        // (no stream wait: the temporaries return to the pool in stream order, the call is asynchronous like the reference's)
    }
    if (seeded) expand_seed(context);
}

void Ciphertext::expand_seed(HeContextPointer context) {
    // ciphertext.cu:79-91: c1 <- uniform(RandomGenerator(seed)) under the ciphertext's moduli
    if (!contains_seed()) throw std::invalid_argument("[Ciphertext::expand_seed] Ciphertext does not contain seed.");
    auto cd = context->get_context_data(parms_id_);
    if (!cd.has_value()) throw std::invalid_argument("[Ciphertext::expand_seed] ParmsID is not valid.");
    if (!context->on_device() || !on_device()) throw std::invalid_argument("[Ciphertext::expand_seed] the seed is expanded on the GPU: context and ciphertext must be on the device.");
    utils::RandomGenerator c1_prng(seed_);
    c1_prng.sample_poly_uniform(context->plan(), coeff_modulus_size_, poly(1));
    hip_check(stream_wait(), "stream_sync");
    seed_ = 0;
}

size_t KSwitchKeys::save(std::ostream& stream, HeContextPointer context, CompressionMode mode) const {
    size_t total = sizeof(ParmsID) + 2 * sizeof(size_t);
    put(stream, parms_id_);
    put(stream, keys_.size());
    size_t valid = 0;
    for (const auto& v : keys_) valid += !v.empty();
    put(stream, valid);
    for (size_t i = 0; i < keys_.size(); i++) {
        if (keys_[i].empty()) continue;
        put(stream, i);
        put(stream, keys_[i].size());
        total += 2 * sizeof(size_t);
        for (const PublicKey& k : keys_[i]) total += k.save(stream, context, mode);
    }
    return total;
}

void KSwitchKeys::load(std::istream& stream, HeContextPointer context, MemoryPoolHandle pool) {
    get(stream, parms_id_);
    size_t size1d, valid;
    get(stream, size1d); get(stream, valid);
    if (size1d > (size_t(1) << 20) || valid > size1d) throw std::runtime_error("[KSwitchKeys::load] invalid sizes");
    keys_.clear();
    keys_.resize(size1d);
    for (size_t v = 0; v < valid; v++) {
        size_t id, size2d;
        get(stream, id); get(stream, size2d);
        if (id >= size1d || size2d > 64) throw std::runtime_error("[KSwitchKeys::load] invalid key index");
        keys_[id].resize(size2d);
        for (size_t j = 0; j < size2d; j++) keys_[id][j].load(stream, context, pool);
    }
}

// ------------------------------------------------------------------------------------------------
// utils::RandomGenerator  (utils/random_generator.cu)
// ------------------------------------------------------------------------------------------------
namespace utils {

uint64_t RandomGenerator::sample_uint64() {
    std::lock_guard<std::mutex> lock(mutex_);
    uint64_t out[2];
    troyn_check(troyn_prng_block(seed_, counter_, out));
    counter_ += 1;
    return out[0];
}

uint64_t RandomGenerator::reserve_blocks(uint64_t blocks) {
    std::lock_guard<std::mutex> lock(mutex_);
    const uint64_t first = counter_;
    counter_ += blocks;
    return first;
}

void RandomGenerator::sample_poly_ternary(const troyn_plan* plan, size_t nmod, uint64_t* destination) {
    std::lock_guard<std::mutex> lock(mutex_);
    uint64_t used = 0;
    troyn_check(troyn_sample_ternary(plan, static_cast<uint32_t>(nmod), seed_, counter_, destination, &used, current_stream()));
    counter_ += used;
}

void RandomGenerator::sample_poly_centered_binomial(const troyn_plan* plan, size_t nmod, uint64_t* destination) {
    std::lock_guard<std::mutex> lock(mutex_);
    uint64_t used = 0;
    troyn_check(troyn_sample_centered_binomial(plan, static_cast<uint32_t>(nmod), seed_, counter_, destination, &used, current_stream()));
    counter_ += used;
}

void RandomGenerator::sample_poly_uniform(const troyn_plan* plan, size_t nmod, uint64_t* destination) {
    std::lock_guard<std::mutex> lock(mutex_);
    uint64_t used = 0;
    troyn_check(troyn_sample_uniform(plan, static_cast<uint32_t>(nmod), seed_, counter_, destination, &used, current_stream()));
    counter_ += used;
}

}  // namespace utils

void Plaintext::resize_rns(const HeContext& context, const ParmsID& parms_id, bool fill_extra_with_zeros, bool copy_data) {
    // plaintext.cu resize_rns: a full RNS polynomial of the level's shape
    auto cd = context.get_context_data(parms_id);
    if (!cd.has_value()) throw std::invalid_argument("[Plaintext::resize_rns] ParmsID is not valid for the current context.");
    const EncryptionParameters& p = cd.value()->parms();
    parms_id_ = parms_id;
    coeff_modulus_size_ = p.coeff_modulus().size();
    poly_modulus_degree_ = p.poly_modulus_degree();
    coeff_count_ = poly_modulus_degree_;
    const size_t words = coeff_modulus_size_ * poly_modulus_degree_;
    if (fill_extra_with_zeros) data_.resize(words, copy_data); else data_.resize_uninitialized(words, copy_data);
}

static void require_device_context(const char* prompt, const HeContextPointer& ctx) {
    if (!ctx->on_device()) throw std::invalid_argument(std::string(prompt) + " HeContext is not on device (call to_device_inplace).");
}

// ------------------------------------------------------------------------------------------------
// rlwe  (utils/rlwe.cu)
// ------------------------------------------------------------------------------------------------
namespace rlwe {

static ContextDataPointer level(const char* prompt, const HeContextPointer& context, const ParmsID& parms_id) {
    auto cd = context->get_context_data(parms_id);
    if (!cd.has_value()) throw std::invalid_argument(std::string(prompt) + " parms_id is not valid for the current context.");
    return cd.value();
}

void symmetric(const SecretKey& sk, HeContextPointer context, const ParmsID& parms_id, bool is_ntt_form, bool save_seed,
               Ciphertext& destination, MemoryPoolHandle pool, utils::RandomGenerator* c1_seed_prng) {
    // utils/rlwe.cu:218-317 (symmetric_with_c1_prng with the context generator as c1 generator)
    const char* P = "[rlwe::symmetric]";
    require_device_context(P, context);
    if (!sk.on_device()) throw std::invalid_argument(std::string(P) + " context_data and secret_key is not on the same device.");
    ContextDataPointer cd = level(P, context, parms_id);
    const EncryptionParameters& parms = cd->parms();
    const uint32_t L = static_cast<uint32_t>(parms.coeff_modulus().size());
    const size_t n = parms.poly_modulus_degree();
    const troyn_plan* plan = context->plan();
    hipStream_t s = current_stream();
    destination = Ciphertext();
    destination.data() = utils::DynamicArray(0, true, pool);
    destination.resize(context, parms_id, 2, true, false);
    destination.is_ntt_form() = is_ntt_form;
    destination.scale() = 1.0;
    destination.correction_factor() = 1;
    destination.seed() = 0;
    utils::RandomGenerator& prng = context->random_generator();
    uint64_t seed = 0;                                    // symmetric_with_c1_prng (utils/rlwe.cu:218-317): the seed of c1 may come from the caller's generator
    while (seed == 0) seed = (c1_seed_prng ? *c1_seed_prng : prng).sample_uint64();
    utils::RandomGenerator c1_prng(seed);
    c1_prng.sample_poly_uniform(plan, L, destination.poly(1));
    if (!is_ntt_form && save_seed) {
        // the seed reproduces c1 in coefficient form; the computation below needs it in NTT form
        troyn_check(troyn_ntt(plan, 0, destination.poly(1), destination.poly(1), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    }
    if (save_seed) destination.seed() = seed;
    utils::DynamicArray noise(static_cast<size_t>(L) * n, true, pool);
    prng.sample_poly_centered_binomial(plan, L, noise.raw_pointer());
    troyn_check(troyn_dyadic_product(plan, 0, L, sk.data().raw_pointer(), destination.poly(1), destination.poly(0), 1, s));
    if (is_ntt_form) troyn_check(troyn_ntt(plan, 0, noise.raw_pointer(), noise.raw_pointer(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    else troyn_check(troyn_ntt(plan, 1, destination.poly(0), destination.poly(0), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    if (parms.scheme() == SchemeType::BGV)      // -(a s + t e), utils/rlwe.cu:300-304
        troyn_check(troyn_multiply_scalar(plan, 0, L, noise.raw_pointer(), parms.plain_modulus().value(), noise.raw_pointer(), 1, s));
    troyn_check(troyn_add(plan, 0, L, destination.poly(0), noise.raw_pointer(), destination.poly(0), 1, s));
    troyn_check(troyn_negate(plan, 0, L, destination.poly(0), destination.poly(0), 1, s));
    if (!is_ntt_form && !save_seed)
        troyn_check(troyn_ntt(plan, 1, destination.poly(1), destination.poly(1), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    hip_check(hipStreamSynchronize(s), "stream_sync");   // `noise` returns to the pool
}

void asymmetric(const PublicKey& pk, HeContextPointer context, const ParmsID& parms_id, bool is_ntt_form,
                Ciphertext& destination, MemoryPoolHandle pool, utils::RandomGenerator* u_prng) {
    // utils/rlwe.cu:11-91 (asymmetric_with_u_prng with the context generator as u generator)
    const char* P = "[rlwe::asymmetric]";
    require_device_context(P, context);
    if (!pk.on_device()) throw std::invalid_argument(std::string(P) + " context_data and public_key is not on the same device.");
    ContextDataPointer cd = level(P, context, parms_id);
    const EncryptionParameters& parms = cd->parms();
    const uint32_t L = static_cast<uint32_t>(parms.coeff_modulus().size());
    const size_t n = parms.poly_modulus_degree();
    const troyn_plan* plan = context->plan();
    hipStream_t s = current_stream();
    const Ciphertext& public_key = pk.as_ciphertext();
    const size_t encrypted_size = public_key.polynomial_count();
    destination = Ciphertext();
    destination.data() = utils::DynamicArray(0, true, pool);
    destination.resize(context, parms_id, encrypted_size, true, false);
    destination.is_ntt_form() = is_ntt_form;
    destination.scale() = 1.0;
    destination.correction_factor() = 1;
    destination.seed() = 0;
    utils::RandomGenerator& prng = context->random_generator();
    utils::DynamicArray u(static_cast<size_t>(L) * n, true, pool);
    (u_prng ? *u_prng : prng).sample_poly_ternary(plan, L, u.raw_pointer());     // asymmetric_with_u_prng: u from the caller's generator, the noise from the context's
    troyn_check(troyn_ntt(plan, 0, u.raw_pointer(), u.raw_pointer(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    for (size_t j = 0; j < encrypted_size; j++)     // the key's first L limbs of polynomial j
        troyn_check(troyn_dyadic_product(plan, 0, L, u.raw_pointer(), public_key.poly(j), destination.poly(j), 1, s));
    if (!is_ntt_form)
        troyn_check(troyn_ntt(plan, 1, destination.poly(0), destination.poly(0), 1, encrypted_size, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    for (size_t j = 0; j < encrypted_size; j++) {
        prng.sample_poly_centered_binomial(plan, L, u.raw_pointer());   // u reused as e_j
        if (is_ntt_form) troyn_check(troyn_ntt(plan, 0, u.raw_pointer(), u.raw_pointer(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
        if (parms.scheme() == SchemeType::BGV)   // pk_j u + t e_j, utils/rlwe.cu:82-86
            troyn_check(troyn_multiply_scalar(plan, 0, L, u.raw_pointer(), parms.plain_modulus().value(), u.raw_pointer(), 1, s));
        troyn_check(troyn_add(plan, 0, L, destination.poly(j), u.raw_pointer(), destination.poly(j), 1, s));
    }
    hip_check(hipStreamSynchronize(s), "stream_sync");
}

}  // namespace rlwe

// ------------------------------------------------------------------------------------------------
// KeyGenerator  (key_generator.cu)
// ------------------------------------------------------------------------------------------------
static Plaintext key_level_plaintext(const HeContextPointer& context, MemoryPoolHandle pool) {
    ContextDataPointer kcd = context->key_context_data().value();
    Plaintext p;
    p.data() = utils::DynamicArray(0, true, pool);
    p.resize_rns(*context, kcd->parms_id());
    p.is_ntt_form() = true;
    return p;
}

KeyGenerator::KeyGenerator(HeContextPointer context, MemoryPoolHandle pool) : context_(std::move(context)) {
    // key_generator.cu:31-58: s <- R_3 under every key-level modulus, stored in NTT form
    require_device_context("[KeyGenerator::KeyGenerator]", context_);
    ContextDataPointer kcd = context_->key_context_data().value();
    const uint32_t K = static_cast<uint32_t>(kcd->parms().coeff_modulus().size());
    Plaintext sk = key_level_plaintext(context_, pool);
    context_->random_generator().sample_poly_ternary(context_->plan(), K, sk.poly());
    troyn_check(troyn_ntt(context_->plan(), 0, sk.poly(), sk.poly(), 1, 1, K, 0, K, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    secret_key_ = SecretKey(std::move(sk));
    secret_key_array_ = secret_key_.data().clone(pool);
}

KeyGenerator::KeyGenerator(HeContextPointer context, const SecretKey& secret_key, MemoryPoolHandle pool) : context_(std::move(context)) {
    require_device_context("[KeyGenerator::KeyGenerator]", context_);
    secret_key_ = secret_key.clone(pool);
    if (!secret_key_.on_device()) secret_key_.to_device_inplace(pool);
    secret_key_array_ = secret_key_.data().clone(pool);
}

PublicKey KeyGenerator::create_public_key(bool save_seed, MemoryPoolHandle pool) const {
    // key_generator.cu:65-84
    Ciphertext c;
    rlwe::symmetric(secret_key_, context_, context_->key_parms_id(), true, save_seed, c, pool);
    return PublicKey(std::move(c));
}

PublicKey KeyGenerator::create_public_key_with_u_prng(bool save_seed, utils::RandomGenerator& u_prng, MemoryPoolHandle pool) const {
    Ciphertext c;
    rlwe::symmetric(secret_key_, context_, context_->key_parms_id(), true, save_seed, c, pool, &u_prng);
    return PublicKey(std::move(c));
}

void KeyGenerator::compute_secret_key_powers(HeContextPointer context, size_t max_power, utils::DynamicArray& secret_key_array) {
    // key_generator.cu:86-109
    ContextDataPointer kcd = context->key_context_data().value();
    const uint32_t K = static_cast<uint32_t>(kcd->parms().coeff_modulus().size());
    const size_t poly_size = static_cast<size_t>(K) * kcd->parms().poly_modulus_degree();
    if (secret_key_array.size() % poly_size != 0 || secret_key_array.size() == 0)
        throw std::invalid_argument("[static KeyGenerator::compute_secret_key_powers] secret_key_array size must be a positive multiple of (coeff_count * coeff_modulus_size)");
    const size_t old_size = secret_key_array.size() / poly_size;
    const size_t new_size = std::max(old_size, max_power);
    if (old_size == new_size) return;
    secret_key_array.resize(new_size * poly_size, true);
    uint64_t* base = secret_key_array.raw_pointer();
    for (size_t i = old_size; i < new_size; i++)
        troyn_check(troyn_dyadic_product(context->plan(), 0, K, base + (i - 1) * poly_size, base, base + i * poly_size, 1, current_stream()));
}

void KeyGenerator::generate_one_kswitch_key(const uint64_t* new_key, std::vector<PublicKey>& destination, bool save_seed, MemoryPoolHandle pool) const {
    // key_generator.cu:136-153
    if (!context_->using_keyswitching()) throw std::logic_error("[KeyGenerator::generate_one_kswitch_key] Keyswitching is not enabled.");
    ContextDataPointer kcd = context_->key_context_data().value();
    const auto& key_modulus = kcd->parms().coeff_modulus();
    const size_t n = kcd->parms().poly_modulus_degree(), K = key_modulus.size();
    const size_t decomp_mod_count = context_->first_context_data().value()->parms().coeff_modulus().size();
    const troyn_plan* plan = context_->plan();
    hipStream_t s = current_stream();
    utils::DynamicArray temp(n, true, pool);
    destination.clear();
    for (size_t i = 0; i < decomp_mod_count; i++) {
        Ciphertext c;
        rlwe::symmetric(secret_key_, context_, kcd->parms_id(), true, save_seed, c, pool);
        const uint64_t factor = key_modulus[i].reduce(key_modulus[K - 1].value());
        troyn_check(troyn_multiply_scalar(plan, static_cast<uint32_t>(i), 1, new_key + i * n, factor, temp.raw_pointer(), 1, s));
        uint64_t* component = c.poly(0) + i * n;
        troyn_check(troyn_add(plan, static_cast<uint32_t>(i), 1, component, temp.raw_pointer(), component, 1, s));
        destination.emplace_back(std::move(c));
    }
    hip_check(hipStreamSynchronize(s), "stream_sync");
}

KSwitchKeys KeyGenerator::create_keyswitching_key(const SecretKey& new_key, bool save_seed, MemoryPoolHandle pool) const {
    std::vector<std::vector<PublicKey>> keys(1);
    generate_one_kswitch_key(new_key.data().raw_pointer(), keys[0], save_seed, pool);
    return KSwitchKeys(context_->key_parms_id(), std::move(keys));
}

RelinKeys KeyGenerator::create_relin_keys(bool save_seed, size_t max_power, MemoryPoolHandle pool) const {
    // key_generator.h:72-77, key_generator.cu:206-237 (generate_rlk)
    if (max_power < 2) throw std::invalid_argument("[KeyGenerator::create_relin_keys] max_power must be at least 2");
    const size_t count = max_power - 1;
    ContextDataPointer kcd = context_->key_context_data().value();
    const size_t d = kcd->parms().coeff_modulus().size() * kcd->parms().poly_modulus_degree();
    std::lock_guard<std::mutex> lock(secret_key_array_mutex_);
    compute_secret_key_powers(context_, count + 1, secret_key_array_);
    std::vector<std::vector<PublicKey>> keys(count);
    for (size_t i = 0; i < count; i++)
        generate_one_kswitch_key(secret_key_array_.raw_pointer() + (i + 1) * d, keys[i], save_seed, pool);
    return RelinKeys(KSwitchKeys(kcd->parms_id(), std::move(keys)));
}

// ------------------------------------------------------------------------------------------------
// Encryptor  (encryptor.cu)
// ------------------------------------------------------------------------------------------------
const PublicKey& Encryptor::public_key() const {
    if (!public_key_.has_value()) throw std::runtime_error("[Encryptor::public_key] Encryptor has no public key");
    return public_key_.value();
}
const SecretKey& Encryptor::secret_key() const {
    if (!secret_key_.has_value()) throw std::runtime_error("[Encryptor::secret_key] Encryptor has no secret key");
    return secret_key_.value();
}

void Encryptor::encrypt_zero_internal(const ParmsID& parms_id, bool is_ntt_form, bool is_asymmetric, bool save_seed,
                                      Ciphertext& destination, MemoryPoolHandle pool, utils::RandomGenerator* u_prng) const {
    // encryptor.cu:12-110
    const char* P = "[Encryptor::encrypt_zero_internal]";
    if (is_asymmetric && !public_key_.has_value()) throw std::invalid_argument(std::string(P) + " Public key not set for asymmetric encryption.");
    if (!is_asymmetric && !secret_key_.has_value()) throw std::invalid_argument(std::string(P) + " Secret key not set for symmetric encryption.");
    if (save_seed && is_asymmetric) throw std::invalid_argument(std::string(P) + " Cannot save seed when using asymmetric encryption.");
    auto cdo = context_->get_context_data(parms_id);
    if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " parms_id is not valid for encryption parameters.");
    ContextDataPointer cd = cdo.value();
    if (!is_asymmetric) {
        rlwe::symmetric(secret_key(), context_, parms_id, is_ntt_form, save_seed, destination, pool, u_prng);
        return;
    }
    ContextDataPointer pcd = cd->prev_context_data_pointer().lock();
    if (!pcd) {
        rlwe::asymmetric(public_key(), context_, parms_id, is_ntt_form, destination, pool, u_prng);
        return;
    }
    // encrypt one level up, then switch the extra prime away (encryptor.cu:44-75)
    Ciphertext temp;
    rlwe::asymmetric(public_key(), context_, pcd->parms_id(), is_ntt_form, temp, pool, u_prng);
    const uint32_t Lp = static_cast<uint32_t>(pcd->parms().coeff_modulus().size());
    const size_t n = pcd->parms().poly_modulus_degree();
    const size_t pc = temp.polynomial_count();
    Ciphertext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize(context_, parms_id, pc, true, false);
    hipStream_t s = current_stream();
    if (cd->parms().scheme() == SchemeType::BGV) {
        // encryptor.cu:65-84
        if (!is_ntt_form) throw std::invalid_argument(std::string(P) + " BGV - Plaintext is not in NTT form.");
        const troyn_bgv* bg = context_->bgv(Lp);
        const size_t wsb = troyn_bgv_mod_switch_workspace_bytes(bg, pc, 1);
        utils::DynamicArray ws((wsb + 7) / 8, true, pool);
        troyn_check(troyn_bgv_mod_t_and_divide_q_last_ntt(bg, temp.data().raw_pointer(), pc, out.data().raw_pointer(), ws.raw_pointer(), wsb, 1, s));
        hip_check(hipStreamSynchronize(s), "stream_sync");
    } else if (is_ntt_form) {
        const size_t wsb = troyn_divide_and_round_q_last_ntt_workspace_bytes(context_->plan(), Lp, pc, 1);
        utils::DynamicArray ws((wsb + 7) / 8, true, pool);
        troyn_check(troyn_divide_and_round_q_last_ntt(context_->plan(), Lp, temp.data().raw_pointer(), pc, out.data().raw_pointer(),
                                                      ws.raw_pointer(), wsb, 1, s));
        hip_check(hipStreamSynchronize(s), "stream_sync");
    } else {
        troyn_check(troyn_divide_and_round_q_last(context_->plan(), Lp, temp.data().raw_pointer(), pc, out.data().raw_pointer(), 1, s));
    }
    (void)n;
    out.is_ntt_form() = is_ntt_form;
    out.scale() = temp.scale();
    out.correction_factor() = temp.correction_factor();
    out.seed() = 0;
    hip_check(hipStreamSynchronize(s), "stream_sync");   // `temp` returns to the pool
    destination = std::move(out);
}

void Encryptor::encrypt_zero_asymmetric(Ciphertext& destination, std::optional<ParmsID> parms_id, utils::RandomGenerator* u_prng, MemoryPoolHandle pool) const {
    const SchemeType scheme = context_->first_context_data().value()->parms().scheme();      // encryptor.h:160-164: CKKS and BGV zeros are in NTT form
    encrypt_zero_internal(parms_id.value_or(context_->first_parms_id()), scheme == SchemeType::CKKS || scheme == SchemeType::BGV, true, false, destination, pool, u_prng);
}
void Encryptor::encrypt_zero_symmetric(bool save_seed, Ciphertext& destination, std::optional<ParmsID> parms_id, utils::RandomGenerator* u_prng, MemoryPoolHandle pool) const {
    const SchemeType scheme = context_->first_context_data().value()->parms().scheme();
    encrypt_zero_internal(parms_id.value_or(context_->first_parms_id()), scheme == SchemeType::CKKS || scheme == SchemeType::BGV, false, save_seed, destination, pool, u_prng);
}

void Encryptor::encrypt_internal(const Plaintext& plain, bool is_asymmetric, bool save_seed, Ciphertext& destination, MemoryPoolHandle pool, utils::RandomGenerator* u_prng) const {
    // encryptor.cu:245-330
    const char* P = "[Encryptor::encrypt_internal]";
    require_device_context(P, context_);
    if (!plain.on_device()) throw std::invalid_argument(std::string(P) + " The arguments are not on the same device.");
    const SchemeType scheme = context_->key_context_data().value()->parms().scheme();
    hipStream_t s = current_stream();
    switch (scheme) {
        case SchemeType::BFV: {
            if (plain.parms_id() == parms_id_zero) {
                if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " BFV - Plaintext is in NTT form.");
                encrypt_zero_internal(context_->first_parms_id(), false, is_asymmetric, save_seed, destination, pool, u_prng);
                // scaling_variant::multiply_add_plain_inplace: c0 += round(q/t * m)
                const size_t L = destination.coeff_modulus_size(), n = destination.poly_modulus_degree();
                if (plain.coeff_count() > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
                troyn_check(troyn_bfv_scale_up(context_->behz(L), plain.poly(), plain.coeff_count(), n, destination.poly(0), L * n,
                                               destination.poly(0), L * n, 0, 1, s));
            } else {
                auto cdo = context_->get_context_data(plain.parms_id());
                if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " BFV - Plaintext parms_id is not valid.");
                const uint32_t L = static_cast<uint32_t>(cdo.value()->parms().coeff_modulus().size());
                encrypt_zero_internal(plain.parms_id(), plain.is_ntt_form(), is_asymmetric, save_seed, destination, pool, u_prng);
                if (plain.coeff_count() != cdo.value()->parms().poly_modulus_degree()) {
                    // a partial RNS plaintext (BatchEncoder::scale_up of a short polynomial): zero-padded to the full shape first
                    const utils::DynamicArray full = plain.expanded_rns(L, cdo.value()->parms().poly_modulus_degree(), pool);
                    troyn_check(troyn_add(context_->plan(), 0, L, destination.poly(0), full.raw_pointer(), destination.poly(0), 1, s));
                    hip_check(hipStreamSynchronize(s), "stream_sync");
                } else {
                    troyn_check(troyn_add(context_->plan(), 0, L, destination.poly(0), plain.poly(), destination.poly(0), 1, s));
                }
            }
            break;
        }
        case SchemeType::CKKS: {
            auto cdo = context_->get_context_data(plain.parms_id());
            if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " CKKS - Plaintext parms_id is not valid.");
            const uint32_t L = static_cast<uint32_t>(cdo.value()->parms().coeff_modulus().size());
            encrypt_zero_internal(plain.parms_id(), plain.is_ntt_form(), is_asymmetric, save_seed, destination, pool, u_prng);
            troyn_check(troyn_add(context_->plan(), 0, L, destination.poly(0), plain.poly(), destination.poly(0), 1, s));
            destination.scale() = plain.scale();
            break;
        }
        case SchemeType::BGV: {
            // encryptor.cu:300-333: zero encryption in NTT form at the first level, plus the (centralized, NTT) plaintext
            encrypt_zero_internal(context_->first_parms_id(), true, is_asymmetric, save_seed, destination, pool, u_prng);
            ContextDataPointer fcd = context_->first_context_data().value();
            const uint32_t L = static_cast<uint32_t>(fcd->parms().coeff_modulus().size());
            const size_t n = fcd->parms().poly_modulus_degree();
            if (!plain.is_ntt_form()) {
                if (plain.coeff_count() > n) throw std::invalid_argument("[scaling_variant::centralize] plain_coeff_count exceeds the polynomial degree.");
                utils::DynamicArray lifted(static_cast<size_t>(L) * n, true, pool);
                troyn_check(troyn_plain_centralize(context_->plan(), L, fcd->parms().plain_modulus().value(), plain.poly(), plain.coeff_count(), n, lifted.raw_pointer(), 1, s));
                troyn_check(troyn_ntt(context_->plan(), 0, lifted.raw_pointer(), lifted.raw_pointer(), 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
                troyn_check(troyn_add(context_->plan(), 0, L, destination.poly(0), lifted.raw_pointer(), destination.poly(0), 1, s));
                hip_check(hipStreamSynchronize(s), "stream_sync");
            } else {
                if (plain.parms_id() != context_->first_parms_id()) throw std::invalid_argument(std::string(P) + " BGV - Plaintext parms_id is not valid.");
                troyn_check(troyn_add(context_->plan(), 0, L, destination.poly(0), plain.poly(), destination.poly(0), 1, s));
            }
            break;
        }
        default:
            throw std::logic_error("[Encryptor::encrypt_internal] Scheme not implemented.");
    }
}

std::vector<Ciphertext> Encryptor::encrypt_symmetric_packed(const uint64_t* plains, size_t coeff_count, size_t stride, size_t count, MemoryPoolHandle pool,
                                                            bool ntt_seeded) const {
    // rlwe::symmetric (utils/rlwe.cu:218-317) + multiply_add_plain for `count` BFV plaintexts, batched.  Ciphertext i uses
    // exactly the generator blocks a sequential call would: block base + i*(1 + N/2) for its c1 seed, the next N/2 blocks
    // for its noise.
    const char* P = "[Encryptor::encrypt_symmetric_batched]";
    require_device_context(P, context_);
    if (!secret_key_.has_value()) throw std::invalid_argument("[Encryptor::encrypt_zero_internal] Secret key not set for symmetric encryption.");
    ContextDataPointer cd = context_->first_context_data().value();
    if (cd->parms().scheme() != SchemeType::BFV) throw std::logic_error(std::string(P) + " the packed path is BFV only.");
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree(), pc = static_cast<size_t>(L) * n;
    if (coeff_count > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
    std::vector<Ciphertext> out;
    if (count == 0) return out;
    const troyn_plan* plan = context_->plan();
    hipStream_t s = current_stream();
    utils::RandomGenerator& prng = context_->random_generator();
    const uint64_t per_ct = 1 + (n + 1) / 2;
    const uint64_t base = prng.reserve_blocks(per_ct * count);
    std::vector<uint64_t> seeds(2 * count, 0);
    for (size_t i = 0; i < count; i++) {
        uint64_t blk[2];
        troyn_check(troyn_prng_block(prng.seed(), base + i * per_ct, blk));
        // the sequential path redraws a zero seed, which would shift every later position; probability 2^-64 per draw
        if (blk[0] == 0) throw std::runtime_error(std::string(P) + " a zero c1 seed was drawn; encrypt again.");
        seeds[2 * i] = blk[0];
    }
    utils::DynamicArray c0(count * pc, true, pool), c1(count * pc, true, pool), noise(count * pc, true, pool);
    troyn_check(troyn_sample_uniform_multi(plan, L, seeds.data(), c1.raw_pointer(), count, s));
    troyn_check(troyn_sample_centered_binomial_strided(plan, L, prng.seed(), base + 1, per_ct, noise.raw_pointer(), count, s));
    troyn_check(troyn_dyadic_broadcast_product(plan, 0, L, c1.raw_pointer(), 1, secret_key_.value().data().raw_pointer(), 0, c0.raw_pointer(), count, s));
    if (ntt_seeded) {
        // everything stays in NTT form: c0 = -(c1 (.) s + NTT(e)) + NTT(round(q/t * m)); c1 is the sampled polynomial itself and
        // is reproduced from its seed by expand_seed (rlwe::symmetric with is_ntt_form && save_seed, then the NTT-form plaintext
        // is added, encryptor.cu:269-284)
        utils::DynamicArray dm(count * pc, true, pool);
        troyn_check(troyn_bfv_scale_up(context_->behz(L), plains, coeff_count, stride, nullptr, 0, dm.raw_pointer(), pc, 0, count, s));
        troyn_check(troyn_ntt(plan, 0, dm.raw_pointer(), dm.raw_pointer(), count, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
        troyn_check(troyn_ntt(plan, 0, noise.raw_pointer(), noise.raw_pointer(), count, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
        troyn_check(troyn_add(plan, 0, L, c0.raw_pointer(), noise.raw_pointer(), c0.raw_pointer(), count, s));
        troyn_check(troyn_negate(plan, 0, L, c0.raw_pointer(), c0.raw_pointer(), count, s));
        troyn_check(troyn_add(plan, 0, L, c0.raw_pointer(), dm.raw_pointer(), c0.raw_pointer(), count, s));
        auto block = std::make_shared<utils::DynamicArray>(count * 2 * pc, true, pool);
        hip_check(hipMemcpy2DAsync(block->raw_pointer(), 2 * pc * 8, c0.raw_pointer(), pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
        hip_check(hipMemcpy2DAsync(block->raw_pointer() + pc, 2 * pc * 8, c1.raw_pointer(), pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
        hip_check(hipStreamSynchronize(s), "stream_sync");
        out.reserve(count);
        for (size_t i = 0; i < count; i++)
            out.push_back(Ciphertext::from_members(2, L, n, cd->parms_id(), 1.0, true, 1, seeds[2 * i], utils::DynamicArray::device_view(block->raw_pointer() + i * 2 * pc, 2 * pc, block)));
        return out;
    }
    // c0 = -(INTT(c1 (.) s) + e) + round(q/t * m); c1 is sampled as an NTT-form polynomial and leaves in coefficient form
    troyn_check(troyn_ntt(plan, 1, c0.raw_pointer(), c0.raw_pointer(), count, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    troyn_check(troyn_add(plan, 0, L, c0.raw_pointer(), noise.raw_pointer(), c0.raw_pointer(), count, s));
    troyn_check(troyn_negate(plan, 0, L, c0.raw_pointer(), c0.raw_pointer(), count, s));
    troyn_check(troyn_ntt(plan, 1, c1.raw_pointer(), c1.raw_pointer(), count, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    troyn_check(troyn_bfv_scale_up(context_->behz(L), plains, coeff_count, stride, c0.raw_pointer(), pc, c0.raw_pointer(), pc, 0, count, s));
    auto shared = std::make_shared<utils::DynamicArray>(count * 2 * pc, true, pool);
    hip_check(hipMemcpy2DAsync(shared->raw_pointer(), 2 * pc * 8, c0.raw_pointer(), pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
    hip_check(hipMemcpy2DAsync(shared->raw_pointer() + pc, 2 * pc * 8, c1.raw_pointer(), pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
    hip_check(hipStreamSynchronize(s), "stream_sync");
    out.reserve(count);
    for (size_t i = 0; i < count; i++)
        out.push_back(Ciphertext::from_members(2, L, n, cd->parms_id(), 1.0, false, 1, 0, utils::DynamicArray::device_view(shared->raw_pointer() + i * 2 * pc, 2 * pc, shared)));
    return out;
}

void Encryptor::encrypt_symmetric_batched(const std::vector<const Plaintext*>& plain, bool save_seed, const std::vector<Ciphertext*>& destination, MemoryPoolHandle pool) const {
    if (plain.size() != destination.size()) throw std::invalid_argument("[Encryptor::encrypt_symmetric_batched] plain and destination are not the same size.");
    if (plain.empty()) return;
    bool packable = !save_seed && context_->on_device() && context_->key_context_data().value()->parms().scheme() == SchemeType::BFV;
    for (const Plaintext* p : plain) packable = packable && p->parms_id() == parms_id_zero && !p->is_ntt_form() && p->on_device();
    if (!packable) {
        for (size_t i = 0; i < plain.size(); i++) encrypt_symmetric(*plain[i], save_seed, *destination[i], pool);
        return;
    }
    const size_t n = context_->first_context_data().value()->parms().poly_modulus_degree();
    utils::DynamicArray staged(plain.size() * n, true, pool);
    staged.set_zero();
    for (size_t i = 0; i < plain.size(); i++) {
        if (plain[i]->coeff_count() > n) throw std::invalid_argument("[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
        hip_check(hipMemcpyAsync(staged.raw_pointer() + i * n, plain[i]->poly(), plain[i]->coeff_count() * 8, hipMemcpyDeviceToDevice, current_stream()), "copy_device_to_device");
    }
    std::vector<Ciphertext> out = encrypt_symmetric_packed(staged.raw_pointer(), n, n, plain.size(), pool);
    for (size_t i = 0; i < plain.size(); i++) *destination[i] = std::move(out[i]);
}

// ------------------------------------------------------------------------------------------------
// Decryptor  (decryptor.cu)
// ------------------------------------------------------------------------------------------------
Decryptor::Decryptor(HeContextPointer context, const SecretKey& secret_key, MemoryPoolHandle pool) : context_(std::move(context)) {
    // decryptor.cu:14-25
    require_device_context("[Decryptor::Decryptor]", context_);
    const EncryptionParameters& kp = context_->key_context_data().value()->parms();
    if (secret_key.data().size() != kp.poly_modulus_degree() * kp.coeff_modulus().size())
        throw std::invalid_argument("[Decryptor::Decryptor] secret_key is not valid for encryption parameters");
    secret_key_array_ = secret_key.data().clone(pool);
    if (!secret_key_array_.on_device()) secret_key_array_.to_device_inplace(pool);
}

void Decryptor::dot_product_ct_sk_array(const Ciphertext& encrypted, uint64_t* destination, MemoryPoolHandle pool) const {
    // decryptor.cu:27-105: c_0 + <(c_1, .., c_k), (s, .., s^k)>, in the form of the ciphertext
    ContextDataPointer cd = context_->get_context_data(encrypted.parms_id()).value();
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree();
    const size_t K = context_->key_context_data().value()->parms().coeff_modulus().size();
    const size_t size = encrypted.polynomial_count();
    const size_t pc = static_cast<size_t>(L) * n, kpc = K * n;
    const troyn_plan* plan = context_->plan();
    hipStream_t s = current_stream();
    std::lock_guard<std::mutex> lock(secret_key_array_mutex_);
    KeyGenerator::compute_secret_key_powers(context_, size - 1, secret_key_array_);
    const bool ntt = encrypted.is_ntt_form();
    utils::DynamicArray copy((size - 1) * pc, true, pool);
    if (!ntt) troyn_check(troyn_ntt(plan, 0, encrypted.poly(1), copy.raw_pointer(), 1, size - 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    for (size_t i = 0; i + 1 < size; i++) {
        const uint64_t* ci = ntt ? encrypted.poly(i + 1) : copy.raw_pointer() + i * pc;
        troyn_check(troyn_dyadic_product(plan, 0, L, ci, secret_key_array_.raw_pointer() + i * kpc, copy.raw_pointer() + i * pc, 1, s));
    }
    hip_check(hipMemcpyAsync(destination, copy.raw_pointer(), pc * 8, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
    for (size_t i = 1; i + 1 < size; i++) troyn_check(troyn_add(plan, 0, L, destination, copy.raw_pointer() + i * pc, destination, 1, s));
    if (!ntt) troyn_check(troyn_ntt(plan, 1, destination, destination, 1, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    troyn_check(troyn_add(plan, 0, L, destination, encrypted.poly(0), destination, 1, s));
    hip_check(hipStreamSynchronize(s), "stream_sync");
}

void Decryptor::decrypt(const Ciphertext& encrypted, Plaintext& destination, MemoryPoolHandle pool) const {
    // decryptor.cu:227-244, :268-362 (BFV), :419-450 (CKKS)
    const char* P = "[Decryptor::decrypt]";
    require_device_context(P, context_);
    if (!encrypted.on_device()) throw std::invalid_argument(std::string(P) + " Operand is on host; the decryptor runs on the GPU only.");
    if (encrypted.contains_seed()) throw std::invalid_argument(std::string(P) + " Seed should be expanded first.");
    if (encrypted.polynomial_count() < 2) throw std::invalid_argument(std::string(P) + " Ciphertext is invalid.");
    auto cdo = context_->get_context_data(encrypted.parms_id());
    if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " Ciphertext parms_id is not valid.");
    ContextDataPointer cd = cdo.value();
    const size_t L = cd->parms().coeff_modulus().size(), n = cd->parms().poly_modulus_degree();
    switch (cd->parms().scheme()) {
        case SchemeType::BFV: {
            if (encrypted.is_ntt_form()) throw std::invalid_argument("[Decryptor::bfv_decrypt] Ciphertext is in NTT form.");
            utils::DynamicArray phase(L * n, true, pool);
            dot_product_ct_sk_array(encrypted, phase.raw_pointer(), pool);
            Plaintext out;
            out.data() = utils::DynamicArray(0, true, pool);
            out.parms_id() = parms_id_zero;
            out.resize(n);
            troyn_check(troyn_bfv_decrypt_scale_and_round(context_->behz(L), phase.raw_pointer(), out.poly(), 1, current_stream()));
            // (no stream wait: the temporaries return to the pool in stream order, the call is asynchronous like the reference's)
            out.is_ntt_form() = false;
            out.coeff_modulus_size() = L;
            out.poly_modulus_degree() = n;
            destination = std::move(out);
            break;
        }
        case SchemeType::CKKS: {
            if (!encrypted.is_ntt_form()) throw std::invalid_argument("[Decryptor::ckks_decrypt] Ciphertext is not in NTT form.");
            Plaintext out;
            out.data() = utils::DynamicArray(0, true, pool);
            out.resize_rns(*context_, encrypted.parms_id());
            dot_product_ct_sk_array(encrypted, out.poly(), pool);
            out.is_ntt_form() = true;
            out.scale() = encrypted.scale();
            destination = std::move(out);
            break;
        }
        case SchemeType::BGV: {
            // decryptor.cu:509-539: NTT-form dot product, INTT, decrypt_mod_t with the correction factor (decentralize)
            if (!encrypted.is_ntt_form()) throw std::invalid_argument("[Decryptor::bgv_decrypt] Ciphertext is not in NTT form.");
            utils::DynamicArray phase(L * n, true, pool);
            dot_product_ct_sk_array(encrypted, phase.raw_pointer(), pool);
            troyn_check(troyn_ntt(context_->plan(), 1, phase.raw_pointer(), phase.raw_pointer(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0,
                                  current_stream()));
            Plaintext out;
            out.data() = utils::DynamicArray(0, true, pool);
            out.parms_id() = parms_id_zero;
            out.resize(n);
            troyn_check(troyn_bgv_decrypt_mod_t(context_->bgv(L), phase.raw_pointer(), encrypted.correction_factor(), out.poly(), 1, current_stream()));
            // (no stream wait: the temporaries return to the pool in stream order, the call is asynchronous like the reference's)
            out.is_ntt_form() = false;
            out.coeff_modulus_size() = L;
            out.poly_modulus_degree() = n;
            destination = std::move(out);
            break;
        }
        default:
            throw std::logic_error("[Decryptor::decrypt] Scheme not implemented.");
    }
}

bool Decryptor::bfv_batchable(const std::vector<const Ciphertext*>& encrypted) const {
    if (encrypted.empty() || !context_->on_device()) return false;
    auto cd = context_->get_context_data(encrypted[0]->parms_id());
    if (!cd.has_value() || cd.value()->parms().scheme() != SchemeType::BFV) return false;
    for (const Ciphertext* c : encrypted)
        if (c->parms_id() != encrypted[0]->parms_id() || c->polynomial_count() != 2 || c->is_ntt_form() || c->contains_seed() || !c->on_device()) return false;
    return true;
}

std::shared_ptr<utils::DynamicArray> Decryptor::bfv_decrypt_batch_device(const std::vector<const Ciphertext*>& encrypted, MemoryPoolHandle pool) const {
    // decryptor.cu:107-225 (dot_product_ct_sk_array_batched) + decrypt_scale_and_round for two-polynomial BFV ciphertexts:
    // gather, then a constant number of launches
    ContextDataPointer cd = context_->get_context_data(encrypted[0]->parms_id()).value();
    const uint32_t L = static_cast<uint32_t>(cd->parms().coeff_modulus().size());
    const size_t n = cd->parms().poly_modulus_degree(), pc = static_cast<size_t>(L) * n, count = encrypted.size();
    const troyn_plan* plan = context_->plan();
    hipStream_t s = current_stream();
    utils::DynamicArray c0(count * pc, true, pool), c1(count * pc, true, pool);
    // ciphertexts that are adjacent windows of one buffer (what the batched producers return) gather in two copies
    const uint64_t* first = encrypted[0]->poly(0);
    const ptrdiff_t step = count > 1 ? encrypted[1]->poly(0) - first : static_cast<ptrdiff_t>(2 * pc);
    bool strided = step == static_cast<ptrdiff_t>(2 * pc);   // adjacent windows; anything else (unrelated allocations) is copied one by one
    const utils::DynamicArray* owner = encrypted[0]->data().view_owner();   // a 2D copy must stay inside ONE allocation
    for (size_t i = 0; i < count && strided; i++)
        strided = owner && encrypted[i]->data().view_owner() == owner && encrypted[i]->poly(0) == first + static_cast<ptrdiff_t>(i) * step;
    if (strided) {
        hip_check(hipMemcpy2DAsync(c0.raw_pointer(), pc * 8, first, static_cast<size_t>(step) * 8, pc * 8, count, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
        hip_check(hipMemcpy2DAsync(c1.raw_pointer(), pc * 8, first + pc, static_cast<size_t>(step) * 8, pc * 8, count, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
    } else {
        for (size_t i = 0; i < count; i++) {
            hip_check(hipMemcpyAsync(c0.raw_pointer() + i * pc, encrypted[i]->poly(0), pc * 8, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
            hip_check(hipMemcpyAsync(c1.raw_pointer() + i * pc, encrypted[i]->poly(1), pc * 8, hipMemcpyDeviceToDevice, s), "copy_device_to_device");
        }
    }
    std::lock_guard<std::mutex> lock(secret_key_array_mutex_);
    troyn_check(troyn_ntt(plan, 0, c1.raw_pointer(), c1.raw_pointer(), count, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    troyn_check(troyn_dyadic_broadcast_product(plan, 0, L, c1.raw_pointer(), 1, secret_key_array_.raw_pointer(), 0, c1.raw_pointer(), count, s));
    troyn_check(troyn_ntt(plan, 1, c1.raw_pointer(), c1.raw_pointer(), count, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, s));
    troyn_check(troyn_add(plan, 0, L, c1.raw_pointer(), c0.raw_pointer(), c1.raw_pointer(), count, s));
    auto out = std::make_shared<utils::DynamicArray>(count * n, true, pool);
    troyn_check(troyn_bfv_decrypt_scale_and_round(context_->behz(L), c1.raw_pointer(), out->raw_pointer(), count, s));
    hip_check(hipStreamSynchronize(s), "stream_sync");
    return out;
}

std::vector<uint64_t> Decryptor::bfv_decrypt_to_host(const std::vector<const Ciphertext*>& encrypted, MemoryPoolHandle pool) const {
    require_device_context("[Decryptor::decrypt_batched]", context_);
    if (!bfv_batchable(encrypted)) {
        std::vector<uint64_t> all;
        for (const Ciphertext* c : encrypted) {
            const std::vector<uint64_t> v = decrypt_new(*c, pool).data().to_vector();
            all.insert(all.end(), v.begin(), v.end());
        }
        return all;
    }
    return bfv_decrypt_batch_device(encrypted, pool)->to_vector();
}

void Decryptor::decrypt_batched(const std::vector<const Ciphertext*>& encrypted, const std::vector<Plaintext*>& destination, MemoryPoolHandle pool) const {
    if (encrypted.size() != destination.size()) throw std::invalid_argument("[Decryptor::decrypt_batched] encrypted and destination are not the same size.");
    if (!bfv_batchable(encrypted)) {
        for (size_t i = 0; i < encrypted.size(); i++) decrypt(*encrypted[i], *destination[i], pool);
        return;
    }
    ContextDataPointer cd = context_->get_context_data(encrypted[0]->parms_id()).value();
    const size_t n = cd->parms().poly_modulus_degree(), L = cd->parms().coeff_modulus().size();
    std::shared_ptr<utils::DynamicArray> shared = bfv_decrypt_batch_device(encrypted, pool);
    for (size_t i = 0; i < encrypted.size(); i++) {
        Plaintext out;
        out.data() = utils::DynamicArray::device_view(shared->raw_pointer() + i * n, n, shared);
        out.parms_id() = parms_id_zero;
        out.coeff_count() = n;
        out.is_ntt_form() = false;
        out.coeff_modulus_size() = L;
        out.poly_modulus_degree() = n;
        *destination[i] = std::move(out);
    }
}

void Decryptor::bfv_decrypt_without_scaling_down(const Ciphertext& encrypted, Plaintext& destination, MemoryPoolHandle pool) const {
    require_device_context("[Decryptor::bfv_decrypt]", context_);
    if (encrypted.is_ntt_form()) throw std::invalid_argument("[Decryptor::bfv_decrypt] Ciphertext is in NTT form.");
    if (!encrypted.on_device()) throw std::invalid_argument("[Decryptor::bfv_decrypt] Operand is on host; the decryptor runs on the GPU only.");
    if (encrypted.contains_seed()) throw std::invalid_argument("[Decryptor::bfv_decrypt] Seed should be expanded first.");
    if (!context_->get_context_data(encrypted.parms_id()).has_value()) throw std::invalid_argument("[Decryptor::bfv_decrypt] Ciphertext parms_id is not valid.");
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, encrypted.parms_id());
    dot_product_ct_sk_array(encrypted, out.poly(), pool);
    out.is_ntt_form() = false;
    destination = std::move(out);
}

size_t Decryptor::invariant_noise_budget(const Ciphertext& encrypted, MemoryPoolHandle pool) const {
    // decryptor.cu:581-640
    const char* P = "[Decryptor::invariant_noise_budget]";
    require_device_context(P, context_);
    if (encrypted.polynomial_count() < 2) throw std::invalid_argument(std::string(P) + " Ciphertext is invalid.");
    const SchemeType scheme = context_->first_context_data().value()->parms().scheme();
    if (scheme != SchemeType::BFV && scheme != SchemeType::BGV) throw std::invalid_argument(std::string(P) + " Unsupported scheme.");
    if (!encrypted.on_device()) throw std::invalid_argument(std::string(P) + " Operand is on host; the decryptor runs on the GPU only.");
    auto cdo = context_->get_context_data(encrypted.parms_id());
    if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " Ciphertext parms_id is not valid.");
    ContextDataPointer cd = cdo.value();
    const auto& q = cd->parms().coeff_modulus();
    const size_t L = q.size(), n = cd->parms().poly_modulus_degree();
    // c(s) = Delta m + v: the phase in coefficient form, times t for BFV (so that the message term vanishes mod q)
    utils::DynamicArray noise(L * n, true, pool);
    dot_product_ct_sk_array(encrypted, noise.raw_pointer(), pool);
    if (encrypted.is_ntt_form())
        troyn_check(troyn_ntt(context_->plan(), 1, noise.raw_pointer(), noise.raw_pointer(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    if (scheme == SchemeType::BFV)
        troyn_check(troyn_multiply_scalar(context_->plan(), 0, static_cast<uint32_t>(L), noise.raw_pointer(), cd->parms().plain_modulus().value(), noise.raw_pointer(), 1, current_stream()));
    const std::vector<uint64_t> h = noise.to_vector();
    // CRT composition as mixed-radix digits x = d_0 + q_0 (d_1 + q_1 (...)); the centred magnitude min(x, Q - x) of every
    // coefficient, the largest one as a multi-word integer, its bit length
    auto mulmod = [](uint64_t a, uint64_t b, uint64_t m) { return static_cast<uint64_t>(static_cast<unsigned __int128>(a) * b % m); };
    auto invmod = [&](uint64_t a, uint64_t m) { uint64_t r = 1, e = m - 2, b = a % m; while (e) { if (e & 1) r = mulmod(r, b, m); b = mulmod(b, b, m); e >>= 1; } return r; };
    std::vector<std::vector<uint64_t>> inv(L);
    for (size_t i = 0; i < L; i++) for (size_t j = 0; j < i; j++) inv[i].push_back(invmod(q[j].value() % q[i].value(), q[i].value()));
    std::vector<uint64_t> d(L), best(L, 0);
    for (size_t x = 0; x < n; x++) {
        for (size_t i = 0; i < L; i++) {
            const uint64_t qi = q[i].value();
            uint64_t v = h[i * n + x] % qi;
            for (size_t j = 0; j < i; j++) { const uint64_t dj = d[j] % qi; v = mulmod(v >= dj ? v - dj : v + qi - dj, inv[i][j], qi); }
            d[i] = v;
        }
        // x > Q/2  <=>  2x > Q: compare 2x with Q digit-wise from the top (Q = all digits q_i - 1, plus one)
        std::vector<uint64_t> c(L);                         // Q - x
        for (size_t i = 0; i < L; i++) c[i] = q[i].value() - 1 - d[i];
        for (size_t i = 0; i < L; i++) { if (++c[i] < q[i].value()) break; c[i] = 0; }
        bool x_is_smaller = true;                           // lexicographic compare of (d) and (c) from the most significant digit
        for (size_t i = L; i-- > 0;) { if (d[i] != c[i]) { x_is_smaller = d[i] < c[i]; break; } }
        const std::vector<uint64_t>& mag = x_is_smaller ? d : c;
        bool larger = false;
        for (size_t i = L; i-- > 0;) { if (mag[i] != best[i]) { larger = mag[i] > best[i]; break; } }
        if (larger) best = mag;
    }
    std::vector<uint64_t> words{0};
    for (size_t i = L; i-- > 0;) {                          // words = words * q_i + best_i
        uint64_t carry = best[i];
        for (uint64_t& w : words) { const unsigned __int128 v = static_cast<unsigned __int128>(w) * q[i].value() + carry; w = static_cast<uint64_t>(v); carry = static_cast<uint64_t>(v >> 64); }
        if (carry) words.push_back(carry);
    }
    while (words.size() > 1 && words.back() == 0) words.pop_back();
    size_t norm_bits = (words.size() - 1) * 64;
    for (uint64_t top = words.back(); top; top >>= 1) norm_bits++;
    const int64_t diff = static_cast<int64_t>(cd->total_coeff_modulus_bit_count()) - static_cast<int64_t>(norm_bits) - 1;
    return diff < 0 ? 0 : static_cast<size_t>(diff);
}

// ------------------------------------------------------------------------------------------------
// BatchEncoder  (batch_encoder.cu)
// ------------------------------------------------------------------------------------------------
static size_t reverse_bits_sz(size_t x, size_t bits) {
    size_t r = 0;
    for (size_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

BatchEncoder::BatchEncoder(HeContextPointer context) : context_(std::move(context)) {
    // batch_encoder.cu:14-64
    ContextDataPointer cd = context_->first_context_data().value();
    const EncryptionParameters& parms = cd->parms();
    if (parms.scheme() != SchemeType::BFV && parms.scheme() != SchemeType::BGV)
        throw std::invalid_argument("[BatchEncoder::BatchEncoder] Unsupported scheme");
    slots_ = parms.poly_modulus_degree();
    const uint64_t t = parms.plain_modulus().value();
    // qualifiers().using_batching: t prime and t = 1 mod 2N.  Without it only the polynomial (coefficient) packing is
    // available (batch_encoder.cu:33: the SIMD tables are built only when batching is possible).
    if (!parms.plain_modulus().is_prime() || (t - 1) % (2 * slots_) != 0) return;
    size_t logn = 0;
    while ((size_t(1) << logn) < slots_) logn++;
    matrix_reps_index_map_.resize(slots_);
    const size_t row = slots_ >> 1, m = slots_ << 1;
    size_t pos = 1;
    for (size_t i = 0; i < row; i++) {
        matrix_reps_index_map_[i] = reverse_bits_sz((pos - 1) >> 1, logn);
        matrix_reps_index_map_[i + row] = reverse_bits_sz((m - pos - 1) >> 1, logn);
        pos = (pos * 3) & (m - 1);     // GALOIS_GENERATOR = 3
    }
}

void BatchEncoder::encode(const std::vector<uint64_t>& values, Plaintext& destination, MemoryPoolHandle pool) const {
    // batch_encoder.cu:169-226: scatter through the index map, inverse NTT modulo t
    const char* P = "[BatchEncoder::encode]";
    require_device_context(P, context_);
    if (matrix_reps_index_map_.empty()) throw std::invalid_argument(std::string(P) + " The parameters do not support vector batching.");
    if (values.size() > slots_) throw std::invalid_argument(std::string(P) + " Values has size larger than the number of slots");
    const uint64_t t = context_->first_context_data().value()->parms().plain_modulus().value();
    std::vector<uint64_t> buf(slots_, 0);
    for (size_t i = 0; i < values.size(); i++) {
        if (values[i] >= t) throw std::invalid_argument(std::string(P) + " Value is larger than plain modulus");
        buf[matrix_reps_index_map_[i]] = values[i];
    }
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.parms_id() = parms_id_zero;
    out.resize(slots_);
    out.data().copy_from(buf.data(), slots_, false);
    troyn_check(troyn_ntt(context_->plain_plan(), 1, out.poly(), out.poly(), 1, 1, 1, 0, 1, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    // (no stream wait: the temporaries return to the pool in stream order, the call is asynchronous like the reference's)
    out.is_ntt_form() = false;
    out.poly_modulus_degree() = slots_;
    out.coeff_modulus_size() = context_->first_context_data().value()->parms().coeff_modulus().size();
    destination = std::move(out);
}

Plaintext BatchEncoder::encode_polynomial_new(const std::vector<uint64_t>& values, MemoryPoolHandle pool) const {
    const char* P = "[BatchEncoder::encode_polynomial]";
    require_device_context(P, context_);
    if (values.size() > slots_) throw std::invalid_argument(std::string(P) + " Values has size larger than the number of slots");
    const uint64_t t = context_->first_context_data().value()->parms().plain_modulus().value();
    for (uint64_t v : values) if (v >= t) throw std::invalid_argument(std::string(P) + " Value is larger than plain modulus");
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.parms_id() = parms_id_zero;
    out.resize(values.size());
    if (!values.empty()) out.data().copy_from(values.data(), values.size(), false);
    out.is_ntt_form() = false;
    out.poly_modulus_degree() = slots_;
    out.coeff_modulus_size() = context_->first_context_data().value()->parms().coeff_modulus().size();
    return out;
}

std::vector<uint64_t> BatchEncoder::decode_polynomial_new(const Plaintext& plain, MemoryPoolHandle) const {
    if (plain.is_ntt_form()) throw std::invalid_argument("[BatchEncoder::decode_polynomial] Plaintext is in NTT form");
    std::vector<uint64_t> v = plain.data().to_vector();
    v.resize(plain.coeff_count());
    return v;
}

void BatchEncoder::decode(const Plaintext& plain, std::vector<uint64_t>& destination, MemoryPoolHandle pool) const {
    // batch_encoder.cu decode: forward NTT modulo t of a copy, gather through the index map
    const char* P = "[BatchEncoder::decode]";
    require_device_context(P, context_);
    if (matrix_reps_index_map_.empty()) throw std::invalid_argument(std::string(P) + " The parameters do not support vector batching.");
    if (plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is in NTT form");
    if (plain.coeff_count() > slots_) throw std::invalid_argument(std::string(P) + " Plaintext is not valid");
    utils::DynamicArray tmp(slots_, true, pool);
    tmp.set_zero();
    hip_check(hipMemcpyAsync(tmp.raw_pointer(), plain.poly(), plain.coeff_count() * 8,
                             plain.on_device() ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, current_stream()), "copy");
    troyn_check(troyn_ntt(context_->plain_plan(), 0, tmp.raw_pointer(), tmp.raw_pointer(), 1, 1, 1, 0, 1, TROYN_IDX_COMPONENTWISE, 0, current_stream()));
    std::vector<uint64_t> host = tmp.to_vector();
    destination.assign(slots_, 0);
    for (size_t i = 0; i < slots_; i++) destination[i] = host[matrix_reps_index_map_[i]];
}

// ------------------------------------------------------------------------------------------------
// CKKSEncoder  (ckks_encoder.cu)
// ------------------------------------------------------------------------------------------------
void Evaluator::rotate_vector(const Ciphertext& encrypted, int steps, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    if (context_->key_context_data().value()->parms().scheme() != SchemeType::CKKS)
        throw std::invalid_argument("[Evaluator::rotate_vector_inplace] Rotate vector only applies for CKKS");
    rotate_internal(encrypted, steps, galois_keys, destination, pool);
}

void Evaluator::complex_conjugate(const Ciphertext& encrypted, const GaloisKeys& galois_keys, Ciphertext& destination, MemoryPoolHandle pool) const {
    if (context_->key_context_data().value()->parms().scheme() != SchemeType::CKKS)
        throw std::invalid_argument("[Evaluator::complex_conjugate_inplace] Complex conjugate only applies for CKKS");
    auto cd = get_context_data("[Evaluator::conjugate_inplace_internal]", encrypted.parms_id());
    apply_galois(encrypted, utils::galois_element_from_step(cd->parms().poly_modulus_degree(), 0), galois_keys, destination, pool);
}

CKKSEncoder::CKKSEncoder(HeContextPointer context) : context_(std::move(context)) {
    // ckks_encoder.cu:121-183 (the reference wants a host context here and moves the encoder later; this build is
    // device-only, so either order is accepted)
    if (!context_->parameters_set()) throw std::invalid_argument("[CKKSEncoder::CKKSEncoder] Encryption parameters are not set correctly.");
    const EncryptionParameters& parms = context_->first_context_data().value()->parms();
    if (parms.scheme() != SchemeType::CKKS) throw std::invalid_argument("[CKKSEncoder::CKKSEncoder] Unsupported scheme.");
    const size_t n = parms.poly_modulus_degree();
    if (n < 4) throw std::invalid_argument("[CKKSEncoder::CKKSEncoder] Poly modulus degree is too small.");
    slots_ = n / 2;
    size_t logn = 0;
    while ((size_t(1) << logn) < n) logn++;
    matrix_reps_index_map_.resize(n);
    const size_t m = n << 1;
    size_t pos = 1;
    for (size_t i = 0; i < slots_; i++) {
        matrix_reps_index_map_[i] = reverse_bits_sz((pos - 1) >> 1, logn);
        matrix_reps_index_map_[i + slots_] = reverse_bits_sz((m - pos - 1) >> 1, logn);
        pos = (pos * 3) & (m - 1);
    }
    // psi = exp(2 pi i / m); root_powers[i] = psi^bitrev(i), inv_root_powers[i] = conj(psi^(bitrev(i-1)+1))
    const double pi = 3.14159265358979323846264338327950288;
    auto root = [&](size_t k) { const double ang = 2.0 * pi * static_cast<double>(k % m) / static_cast<double>(m); return std::complex<double>(std::cos(ang), std::sin(ang)); };
    root_powers_.assign(n, {0, 0});
    inv_root_powers_.assign(n, {0, 0});
    for (size_t i = 1; i < n; i++) {
        root_powers_[i] = root(reverse_bits_sz(i, logn));
        inv_root_powers_[i] = std::conj(root(reverse_bits_sz(i - 1, logn) + 1));
    }
}

void CKKSEncoder::set_plaintext(const std::vector<double>& coeffs, const ParmsID& parms_id, double scale, Plaintext& destination, MemoryPoolHandle pool) const {
    // set_plaintext_value_array (ckks_encoder.cu:454-690): scaled real coefficients -> residues, then NTT
    const char* P = "[CKKSEncoder::encode_internal]";
    if (!context_->on_device()) throw std::invalid_argument(std::string(P) + " HeContext is not on device (call to_device_inplace).");
    auto cdo = context_->get_context_data(parms_id);
    if (!cdo.has_value()) throw std::invalid_argument("[CKKSEncoder::encode_internal_complex_array] parms_id not valid for context.");
    const EncryptionParameters& parms = cdo.value()->parms();
    const auto& q = parms.coeff_modulus();
    const size_t n = parms.poly_modulus_degree(), L = q.size();
    size_t total_bits = 0;
    for (const Modulus& mdl : q) total_bits += mdl.bit_count();
    if (scale <= 0 || std::log2(scale) + 1.0 >= static_cast<double>(total_bits))
        throw std::invalid_argument("[CKKSEncoder::encode_internal_complex_array] scale out of bounds.");
    // the residues are written straight into the thread's pinned staging image and leave by one asynchronous copy (a pageable source would be staged by the
    // runtime and waited for); the next user of the image waits for that copy, this call does not
    PinnedImage& img = pinned_image();
    uint64_t* host = reinterpret_cast<uint64_t*>(img.reserve(L * n * sizeof(uint64_t)));
    if (coeffs.size() < n) for (size_t i = 0; i < L; i++) std::memset(host + i * n + coeffs.size(), 0, (n - coeffs.size()) * sizeof(uint64_t));
    const double two64 = 18446744073709551616.0;
    for (size_t j = 0; j < n && j < coeffs.size(); j++) {
        const double v = std::nearbyint(coeffs[j]);
        const bool neg = v < 0;
        double a = std::fabs(v);
        if (a >= two64 * two64) throw std::invalid_argument("[CKKSEncoder::encode_internal_complex_array] encoded values are too large.");
        const uint64_t hi = static_cast<uint64_t>(a / two64);
        const uint64_t lo = static_cast<uint64_t>(a - static_cast<double>(hi) * two64);
        for (size_t i = 0; i < L; i++) {
            const uint64_t qi = q[i].value();
            // below 2^64 (every practical scale): one Barrett reduction instead of a 128-bit division
            const uint64_t r = hi ? static_cast<uint64_t>(((static_cast<unsigned __int128>(hi) << 64) | lo) % qi) : q[i].reduce(lo);
            host[i * n + j] = (neg && r) ? qi - r : r;
        }
    }
    Plaintext out;
    out.data() = utils::DynamicArray(0, true, pool);
    out.resize_rns(*context_, parms_id, false, false);        // every word is written by the copy below
    hipStream_t s = current_stream();
    hip_check(hipMemcpyAsync(out.poly().raw_pointer(), host, L * n * sizeof(uint64_t), hipMemcpyHostToDevice, s), "copy_host_to_device");
    img.mark(s);
    troyn_check(troyn_ntt(context_->plan(), 0, out.poly(), out.poly(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, s));
    out.scale() = scale;
    out.is_ntt_form() = true;
    destination = std::move(out);
}

void CKKSEncoder::encode_complex64_simd(const std::vector<std::complex<double>>& values, std::optional<ParmsID> parms_id, double scale,
                                        Plaintext& destination, MemoryPoolHandle pool) const {
    // ckks_encoder.cu:693-776
    if (values.size() > slots_) throw std::invalid_argument("[CKKSEncoder::encode_internal_complex_array] Too many input values.");
    const size_t n = slots_ * 2;
    std::vector<std::complex<double>> a(n, {0, 0});
    for (size_t i = 0; i < values.size(); i++) {                      // set_conjugate_values
        a[matrix_reps_index_map_[i]] = values[i];
        a[matrix_reps_index_map_[i + slots_]] = std::conj(values[i]);
    }
    // fft_transform_from_rev: Gentleman-Sande layers with the inverse root powers, then scale / n
    size_t root_index = 1;
    for (size_t mm = n >> 1, gap = 1; mm >= 1; mm >>= 1, gap <<= 1) {
        size_t offset = 0;
        for (size_t i = 0; i < mm; i++) {
            const std::complex<double> r = inv_root_powers_[root_index++];
            // products written out: std::complex's operator* goes through __muldc3 (the C99 Annex G NaN recovery), several times the cost of these
            // four multiplications, and returns the same bits whenever the result is finite
            const double rr = r.real(), ri = r.imag();
            for (size_t j = offset; j < offset + gap; j++) {
                const std::complex<double> u = a[j], v = a[j + gap];
                const double dr = u.real() - v.real(), di = u.imag() - v.imag();
                a[j] = std::complex<double>(u.real() + v.real(), u.imag() + v.imag());
                a[j + gap] = std::complex<double>(dr * rr - di * ri, dr * ri + di * rr);
            }
            offset += gap << 1;
        }
        if (mm == 1) break;
    }
    const double fix = scale / static_cast<double>(n);
    std::vector<double> coeffs(n);
    for (size_t i = 0; i < n; i++) coeffs[i] = a[i].real() * fix;
    set_plaintext(coeffs, parms_id.value_or(context_->first_parms_id()), scale, destination, pool);
}

void CKKSEncoder::encode_float64_polynomial(const std::vector<double>& values, std::optional<ParmsID> parms_id, double scale, Plaintext& destination,
                                            MemoryPoolHandle pool) const {
    // ckks_encoder.cu:777-848: coefficients directly
    if (values.size() > slots_ * 2) throw std::invalid_argument("[CKKSEncoder::encode_internal_double_polynomial] Too many input values.");
    std::vector<double> coeffs(values.size());
    for (size_t i = 0; i < values.size(); i++) coeffs[i] = values[i] * scale;
    set_plaintext(coeffs, parms_id.value_or(context_->first_parms_id()), scale, destination, pool);
}

std::vector<double> CKKSEncoder::plaintext_coefficients(const Plaintext& plain, MemoryPoolHandle pool) const {
    // decode_internal (ckks_encoder.cu:947-1290): INTT, CRT-compose, centre.  Mixed-radix (Garner) digits instead of the
    // reference's multi-word compose: x = d_0 + q_0 (d_1 + q_1 (d_2 + ...)), evaluated in long double from the top.
    const char* P = "[CKKSEncoder::decode_internal]";
    if (!plain.is_ntt_form()) throw std::invalid_argument(std::string(P) + " Plaintext is not in NTT form.");
    auto cdo = context_->get_context_data(plain.parms_id());
    if (!cdo.has_value()) throw std::invalid_argument(std::string(P) + " Plaintext parms_id is not valid.");
    const auto& q = cdo.value()->parms().coeff_modulus();
    const size_t n = cdo.value()->parms().poly_modulus_degree(), L = q.size();
    utils::DynamicArray tmp(L * n, true, pool);
    hipStream_t s = current_stream();
    troyn_check(troyn_ntt(context_->plan(), 1, plain.poly(), tmp.raw_pointer(), 1, 1, static_cast<uint32_t>(L), 0, static_cast<uint32_t>(L), TROYN_IDX_COMPONENTWISE, 0, s));
    PinnedImage& img = pinned_image();                        // a pageable destination would be staged by the runtime: read the residues from pinned memory
    const uint64_t* h = reinterpret_cast<const uint64_t*>(img.reserve(L * n * sizeof(uint64_t)));
    hip_check(hipMemcpyAsync(const_cast<uint64_t*>(h), tmp.raw_pointer(), L * n * sizeof(uint64_t), hipMemcpyDeviceToHost, s), "copy_device_to_host");
    hip_check(stream_wait(), "copy_device_to_host");
    auto mulmod = [](uint64_t a, uint64_t b, uint64_t m) { return static_cast<uint64_t>(static_cast<unsigned __int128>(a) * b % m); };
    auto invmod = [&](uint64_t a, uint64_t m) {   // m prime
        uint64_t r = 1, e = m - 2, b = a % m;
        while (e) { if (e & 1) r = mulmod(r, b, m); b = mulmod(b, b, m); e >>= 1; }
        return r;
    };
    // inv[i][j] = q_j^-1 mod q_i (j < i)
    // ... with its Shoup quotient floor(inv 2^64 / q_i): a product by a constant is two multiplications and a conditional subtraction instead of a 128-bit division
    std::vector<std::vector<uint64_t>> inv(L), inv_quo(L);
    for (size_t i = 0; i < L; i++) for (size_t j = 0; j < i; j++) {
        const uint64_t w = invmod(q[j].value() % q[i].value(), q[i].value());
        inv[i].push_back(w);
        inv_quo[i].push_back(static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 64) / q[i].value()));
    }
    std::vector<bool> digit_fits(L * L, false);      // q_j <= q_i: a digit below q_j needs no reduction modulo q_i
    for (size_t i = 0; i < L; i++) for (size_t j = 0; j < i; j++) digit_fits[i * L + j] = q[j].value() <= q[i].value();
    std::vector<double> out(n);
    std::vector<uint64_t> d(L);
    for (size_t x = 0; x < n; x++) {
        for (size_t i = 0; i < L; i++) {
            const uint64_t qi = q[i].value();
            uint64_t v = h[i * n + x];                // the inverse transform leaves canonical residues
            if (v >= qi) v = q[i].reduce(v);
            for (size_t j = 0; j < i; j++) {
                const uint64_t dj = digit_fits[i * L + j] ? d[j] : q[i].reduce(d[j]);
                const uint64_t t = v >= dj ? v - dj : v + qi - dj;
                const uint64_t est = static_cast<uint64_t>((static_cast<unsigned __int128>(t) * inv_quo[i][j]) >> 64);
                uint64_t r = t * inv[i][j] - est * qi;
                v = r >= qi ? r - qi : r;
            }
            d[i] = v;
        }
        // x > Q/2  <=>  the top digit decides (ties resolved by lower digits; a tie is measure-zero for real data)
        bool negative = 2 * static_cast<unsigned __int128>(d[L - 1]) >= q[L - 1].value();
        if (negative) {                      // Q - x: complement the digits, add one
            for (size_t i = 0; i < L; i++) d[i] = q[i].value() - 1 - d[i];
            for (size_t i = 0; i < L; i++) { if (++d[i] < q[i].value()) break; d[i] = 0; }
        }
        long double acc = 0;
        for (size_t i = L; i-- > 0;) acc = acc * static_cast<long double>(q[i].value()) + static_cast<long double>(d[i]);
        out[x] = static_cast<double>(negative ? -acc : acc);
    }
    return out;
}

void CKKSEncoder::decode_float64_polynomial(const Plaintext& plain, std::vector<double>& destination, MemoryPoolHandle pool) const {
    destination = plaintext_coefficients(plain, pool);
    for (double& v : destination) v /= plain.scale();
}

void CKKSEncoder::decode_complex64_simd(const Plaintext& plain, std::vector<std::complex<double>>& destination, MemoryPoolHandle pool) const {
    const std::vector<double> c = plaintext_coefficients(plain, pool);
    const size_t n = slots_ * 2;
    std::vector<std::complex<double>> a(n);
    for (size_t i = 0; i < n; i++) a[i] = std::complex<double>(c[i] / plain.scale(), 0.0);
    // fft_transform_to_rev: Cooley-Tukey layers with the root powers (natural -> bit-reversed)
    size_t root_index = 1;
    for (size_t mm = 1, gap = n >> 1; mm < n; mm <<= 1, gap >>= 1) {
        size_t offset = 0;
        for (size_t i = 0; i < mm; i++) {
            const std::complex<double> r = root_powers_[root_index++];
            const double rr = r.real(), ri = r.imag();
            for (size_t j = offset; j < offset + gap; j++) {
                const std::complex<double> u = a[j], w = a[j + gap];
                const double vr = w.real() * rr - w.imag() * ri, vi = w.real() * ri + w.imag() * rr;      // w * r without __muldc3 (see encode)
                a[j] = std::complex<double>(u.real() + vr, u.imag() + vi);
                a[j + gap] = std::complex<double>(u.real() - vr, u.imag() - vi);
            }
            offset += gap << 1;
        }
    }
    destination.resize(slots_);
    for (size_t i = 0; i < slots_; i++) destination[i] = a[matrix_reps_index_map_[i]];
}

}  // namespace troy
