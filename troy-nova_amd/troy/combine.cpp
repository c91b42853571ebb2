// Call combining: single-object Evaluator calls of concurrent host threads run as ONE batched library call (troy.h, "Call combining").
//
// The reference's multi-thread model is N host threads x single-object calls on per-thread streams (test/bench/he_operations.cu:85,
// :364-380; test/test_multithread.cu:18-37).  Here the GPU time of a launch sequence over 16 ciphertexts is ~1.6x that over one
// (tools/small_batch_sweep.py) while streams overlap ~4-fold at most, so the calls of concurrent threads are worth gathering.
//
// While combining is on every host thread launches on ONE shared stream (troy.cpp current_stream()), so a batch is ordered behind the
// producers of its operands and ahead of the consumers of its results by the stream itself: nobody waits for the GPU in here.
//
// Protocol (one mutex; a waiting caller sleeps on a futex on its own request -- the GPU boxes give a process a CPU quota, spinning threads
// would spend it):
//   submit: append the request; if nobody leads, lead.
//   leader: wait until the requests of ITS shape number the threads that were active in the last few ms, or the window has passed; take
//           them; hand the lead to the first request left behind, if any; queue gather + the batched library call + scatter; release the
//           waiters (or hand them the error).
// A leader that ends up alone returns "not combined" and the caller runs its call the ordinary way.
#include <hip/hip_runtime.h>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <thread>

#include "troy.h"

namespace troy {

namespace {

using clk = std::chrono::steady_clock;

inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now().time_since_epoch()).count(); }

enum : int { WAITING = 0, PROMOTED = 1, DONE = 2 };

thread_local int64_t tl_left = 0;      // when this thread left its last combined call
thread_local bool tl_waited = false;   // ... and whether it has waited for the stream since

constexpr size_t MAX_SLOTS = 512;
constexpr int64_t ACTIVE_HORIZON_NS = 3'000'000;   // a thread counts as active for 3 ms after its last combinable call
constexpr size_t MAX_BATCH = 64;                   // the pointers of <= 64 buffers travel in the gather / scatter kernels' arguments (no table upload)

struct Entry { detail::CombineRequest* r; MemoryPool* pool; int64_t pushed = 0; };   // the pool is part of a call's shape (the batch allocates from ONE pool)

struct Combiner {
    std::mutex m;
    std::vector<Entry> pending;
    bool leader = false;
    std::atomic<uint64_t> arrivals{0};               // bumped after every append: a waiting leader re-reads the list only when it moved
    std::atomic<unsigned> window_us{100};
    std::atomic<int64_t> last_seen[MAX_SLOTS];
    std::atomic<size_t> slots{0};
    combining::Stats st;
    Combiner() {
        for (auto& a : last_seen) a.store(0, std::memory_order_relaxed);
        if (const char* e = std::getenv("TROY_COMBINE_WINDOW_US")) { const long v = std::strtol(e, nullptr, 10); if (v >= 0 && v < 100000) window_us.store((unsigned)v); }
    }
};

// leaked on purpose: host threads may still be inside a call when static destructors run
Combiner& combiner() { static Combiner* c = new Combiner; return *c; }

size_t my_slot(Combiner& c) {
    thread_local size_t slot = c.slots.fetch_add(1) % MAX_SLOTS;   // more than MAX_SLOTS threads share slots: the count of active threads saturates
    return slot;
}

size_t active_threads(Combiner& c, int64_t now) {
    const size_t n = std::min(c.slots.load(std::memory_order_relaxed), MAX_SLOTS);
    size_t a = 0;
    for (size_t i = 0; i < n; i++) a += (now - c.last_seen[i].load(std::memory_order_relaxed)) < ACTIVE_HORIZON_NS;
    return a;
}

bool same_shape(const detail::CombineRequest& a, const detail::CombineRequest& b, const MemoryPool* pa, const MemoryPool* pb) {
    if (a.kind != b.kind || a.handle != b.handle || a.L != b.L || a.p1 != b.p1 || a.p2 != b.p2 || a.ckks != b.ckks || a.ntt_form != b.ntt_form ||
        a.words1 != b.words1 || a.words2 != b.words2 || a.out_words != b.out_words || pa != pb) return false;
    if ((a.keys == nullptr) != (b.keys == nullptr)) return false;
    if (a.keys && (a.keys->size() != b.keys->size() || *a.keys != *b.keys)) return false;   // the same key set: the same device pointers
    return true;
}

void lib_ok(int code) {
    if (code != TROYN_OK) {
        const char* m = troyn_last_error();
        throw std::runtime_error(m && *m ? m : "[troyn] library call failed");
    }
}

static_assert(sizeof(std::atomic<int>) == sizeof(int), "futex word");
void futex_wait(std::atomic<int>* a, int expected) { (void)syscall(SYS_futex, reinterpret_cast<int*>(a), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0); }
void futex_wake(std::atomic<int>* a) { (void)syscall(SYS_futex, reinterpret_cast<int*>(a), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

// operands of the batch as [count][words]: in place when they happen to be consecutive windows of one buffer, else one gather launch
const uint64_t* stage(const std::vector<detail::CombineRequest*>& batch, bool second, utils::DynamicArray& staged, void* table, size_t table_bytes,
                      MemoryPoolHandle pool, hipStream_t s) {
    const size_t count = batch.size();
    const size_t words = second ? batch[0]->words2 : batch[0]->words1;
    const uint64_t* base = second ? batch[0]->in2 : batch[0]->in1;
    bool adjacent = true;
    for (size_t i = 0; i < count && adjacent; i++) adjacent = (second ? batch[i]->in2 : batch[i]->in1) == base + i * words;
    if (adjacent) return base;
    staged = utils::DynamicArray(count * words, true, pool);
    std::vector<const uint64_t*> src(count);
    for (size_t i = 0; i < count; i++) src[i] = second ? batch[i]->in2 : batch[i]->in1;
    lib_ok(troyn_gather(src.data(), count, words, staged.raw_pointer(), table, table_bytes, (troyn_stream_t)s));
    return staged.raw_pointer();
}

void execute(std::vector<detail::CombineRequest*>& batch, MemoryPoolHandle pool) {
    using detail::CombineKind;
    const detail::CombineRequest& h = *batch[0];
    const size_t count = batch.size();
    hipStream_t s = (hipStream_t)troyn_current_stream();   // the shared stream
    {
        utils::DynamicArray s1, s2, block, ws;
        const size_t table_bytes = troyn_gather_workspace_bytes(count);
        utils::DynamicArray table((table_bytes + 7) / 8, true, pool);
        const uint64_t* a = stage(batch, false, s1, table.raw_pointer(), table_bytes, pool, s);
        const uint64_t* b = h.in2 ? stage(batch, true, s2, table.raw_pointer(), table_bytes, pool, s) : nullptr;
        block = utils::DynamicArray(count * h.out_words, true, pool);
        uint64_t* out = block.raw_pointer();
        switch (h.kind) {
            case CombineKind::DyadicMultiply: {
                const troyn_plan* plan = static_cast<const troyn_plan*>(h.handle);
                lib_ok(troyn_dyadic_convolute(plan, 0, h.L, a, h.p1, b, h.p2, out, count, (troyn_stream_t)s));
                break;
            }
            case CombineKind::BfvMultiply: {
                const troyn_behz* bz = static_cast<const troyn_behz*>(h.handle);
                const size_t bytes = troyn_bfv_multiply_workspace_bytes(bz, h.p1, h.p2, count);
                ws = utils::DynamicArray((bytes + 7) / 8, true, pool);
                lib_ok(troyn_bfv_multiply(bz, a, h.p1, b, h.p2, out, ws.raw_pointer(), bytes, count, (troyn_stream_t)s));
                break;
            }
            case CombineKind::Relinearize: {
                const troyn_plan* plan = static_cast<const troyn_plan*>(h.handle);
                const size_t bytes = troyn_relinearize_workspace_bytes(plan, h.L, count);
                ws = utils::DynamicArray((bytes + 7) / 8, true, pool);
                lib_ok(troyn_relinearize(plan, h.L, h.ckks, h.ntt_form, a, h.keys->data(), out, ws.raw_pointer(), bytes, count, (troyn_stream_t)s));
                break;
            }
            case CombineKind::Rescale: {
                const troyn_plan* plan = static_cast<const troyn_plan*>(h.handle);
                const size_t bytes = troyn_divide_and_round_q_last_ntt_workspace_bytes(plan, h.L, h.p1, count);
                ws = utils::DynamicArray((bytes + 7) / 8, true, pool);
                lib_ok(troyn_divide_and_round_q_last_ntt(plan, h.L, a, h.p1, out, ws.raw_pointer(), bytes, count, (troyn_stream_t)s));
                break;
            }
            case CombineKind::ApplyGalois: {
                // Evaluator::apply_galois (evaluator_keyswitching.cu:147-179) over the batch: permute (c0, c1) of every item, take the permuted c1s
                // as key-switch targets, overwrite them with the switched result (c0 += ks0, c1 = ks1)
                const troyn_plan* plan = static_cast<const troyn_plan*>(h.handle);
                const size_t pc = h.words1 / 2;
                lib_ok(troyn_apply_galois(plan, 0, h.L, h.ntt_form ? 1 : 0, h.p2, a, out, count * 2, (troyn_stream_t)s));
                s2 = utils::DynamicArray(count * pc, true, pool);
                if (hipMemcpy2DAsync(s2.raw_pointer(), pc * 8, out + pc, 2 * pc * 8, pc * 8, count, hipMemcpyDeviceToDevice, s) != hipSuccess)
                    throw std::runtime_error("[kernel_provider::copy_device_to_device] failed");
                const size_t bytes = troyn_switch_key_workspace_bytes(plan, h.L, count);
                ws = utils::DynamicArray((bytes + 7) / 8, true, pool);
                lib_ok(troyn_switch_key(plan, h.L, h.ckks, h.ntt_form, s2.raw_pointer(), h.keys->data(), TROYN_ASSIGN_OVERWRITE_EXCEPT_FIRST, out, ws.raw_pointer(), bytes, count,
                                        (troyn_stream_t)s));
                break;
            }
            case CombineKind::MultiplyRelinearizeRescale: {
                const troyn_plan* plan = static_cast<const troyn_plan*>(h.handle);
                const size_t bytes = troyn_ckks_multiply_relinearize_rescale_workspace_bytes(plan, h.L, count);
                ws = utils::DynamicArray((bytes + 7) / 8, true, pool);
                lib_ok(troyn_ckks_multiply_relinearize_rescale(plan, h.L, a, b, h.keys->data(), out, ws.raw_pointer(), bytes, count, (troyn_stream_t)s));
                break;
            }
        }
        std::vector<uint64_t*> dst(count);
        for (size_t i = 0; i < count; i++) dst[i] = batch[i]->out;
        lib_ok(troyn_scatter(out, dst.data(), count, h.out_words, table.raw_pointer(), table_bytes, (troyn_stream_t)s));
    }   // staging, block, workspace go back to the pool here: whoever takes them next uses them behind this batch (one stream)
}

}  // namespace

namespace combining {
void set_enabled(bool on) { detail::combining_switch(on); }
bool enabled() { return detail::combining_on(); }
void set_window_us(unsigned us) { combiner().window_us.store(us); }
unsigned window_us() { return combiner().window_us.load(); }
Stats stats() { Combiner& c = combiner(); std::lock_guard<std::mutex> g(c.m); return c.st; }
void reset_stats() { Combiner& c = combiner(); std::lock_guard<std::mutex> g(c.m); c.st = Stats(); }
}  // namespace combining

namespace detail {

// Waiting for the shared stream, grouped: with N threads on one stream N waits would be N threads polling the same queue (HIP's wait
// spins; the GPU boxes give a process a CPU quota).  One waiter at a time calls hipStreamSynchronize; a thread is satisfied by a wait
// that STARTED after it asked (that wait covers everything the thread had queued), the others sleep on a futex meanwhile.
namespace {
struct Waiter { std::atomic<int> word{0}; Waiter* wake[4] = {}; int err = 0; };   // word: 0 waiting, 1 covered by a finished wait (err = that wait's hipError_t), 2 lead the next one
struct WaitGroup {
    std::mutex m;
    bool running = false;
    std::vector<Waiter*> pending;   // asked while a wait was running: covered by the NEXT one
};
WaitGroup& wait_group() { static WaitGroup* w = new WaitGroup; return *w; }
}  // namespace

int combining_stream_wait(void* stream) {
    tl_waited = true;
    WaitGroup& w = wait_group();
    Waiter me;
    std::unique_lock<std::mutex> lk(w.m);
    if (w.running) {
        w.pending.push_back(&me);
        lk.unlock();
        int s;
        while ((s = me.word.load(std::memory_order_acquire)) == 0) futex_wait(&me.word, 0);
        if (s == 1) {
            for (Waiter* k : me.wake) if (k) futex_wake(&k->word);   // by address only
            return me.err;      // the result of the wait that covered this thread: a faulted stream is everybody's error
        }
        lk.lock();   // 2: lead the next wait (`running` stayed set)
    }
    w.running = true;
    std::vector<Waiter*> covered;
    covered.swap(w.pending);   // everyone who asked before this wait starts is covered by it
    lk.unlock();
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    lk.lock();
    std::atomic<int>* next = nullptr;
    if (!w.pending.empty()) { next = &w.pending.front()->word; w.pending.erase(w.pending.begin()); }
    else w.running = false;
    lk.unlock();
    constexpr size_t F = 4;
    for (size_t k = 0; k < covered.size(); k++)
        for (size_t i = 0; i < F; i++) covered[k]->wake[i] = F * (k + 1) + i < covered.size() ? covered[F * (k + 1) + i] : nullptr;
    std::atomic<int>* root[F];
    for (size_t i = 0; i < F; i++) root[i] = i < covered.size() ? &covered[i]->word : nullptr;
    for (Waiter* x : covered) { x->err = (int)e; x->word.store(1, std::memory_order_release); }   // from here on `x` may be gone
    for (std::atomic<int>* r : root) if (r) futex_wake(r);
    if (next) { next->store(2, std::memory_order_release); futex_wake(next); }
    return (int)e;
}

bool combining_wanted() {
    if (!combining_on() || !on_combining_stream()) return false;   // off, or this thread is on another device (it keeps its own stream)
    Combiner& c = combiner();
    const int64_t t = now_ns();
    c.last_seen[my_slot(c)].store(t, std::memory_order_relaxed);
    return active_threads(c, t) >= 2;
}

namespace {
struct Leave { ~Leave() { tl_left = now_ns(); } };
}  // namespace

bool combine_submit(CombineRequest& request, MemoryPoolHandle pool) {
    Combiner& c = combiner();
    const int64_t entered = now_ns();
    const int64_t between = tl_left ? entered - tl_left : 0;
    Leave leave;
    request.state.store(WAITING, std::memory_order_relaxed);
    bool lead = false;
    {
        std::lock_guard<std::mutex> g(c.m);
        c.pending.push_back({&request, pool.get(), now_ns()});
        if (!c.leader) { c.leader = true; lead = true; }
        if (between > 0 && between < ACTIVE_HORIZON_NS) {
            if (tl_waited) { c.st.between_wait_ns += (uint64_t)between; c.st.between_wait_calls += 1; }
            else { c.st.between_ns += (uint64_t)between; c.st.between_calls += 1; }
        }
        tl_waited = false;
    }
    c.arrivals.fetch_add(1, std::memory_order_release);
    if (!lead) {
        // poll briefly, then sleep: a leader gathers for tens of microseconds and queues for ~30; polling callers would spend the process's
        // CPU quota (the GPU boxes give one), and measured with 16 threads the poll length (0 ... 1000 us) does not move the throughput.
        // TROY_COMBINE_POLL_US=<n> (read once) for machines where it does.
        static const long poll_env = std::getenv("TROY_COMBINE_POLL_US") ? std::strtol(std::getenv("TROY_COMBINE_POLL_US"), nullptr, 10) : 10;
        const int64_t poll_until = now_ns() + (poll_env > 0 ? poll_env : 0) * 1000;
        int s;
        while ((s = request.state.load(std::memory_order_acquire)) == WAITING) {
            if (now_ns() < poll_until) { __builtin_ia32_pause(); continue; }
            futex_wait(&request.state, WAITING);
        }
        if (s == DONE) {
            for (CombineRequest* k : request.wake) if (k) futex_wake(&k->state);   // by address only: `k` may have left already
            if (request.error) std::rethrow_exception(request.error);
            return true;
        }
        // PROMOTED: the previous leader left this request (and maybe others) behind
    }
    // ---- leading ----
    std::vector<CombineRequest*> batch;
    const int64_t t0 = now_ns();
    // the callers of the previous batch come back ~3 us apart per thread: the window grows with the number of threads to wait for
    const int64_t window = (int64_t)c.window_us.load(std::memory_order_relaxed) * 1000 * (int64_t)std::max<size_t>(1, active_threads(c, t0) / 16);
    uint64_t seen = ~uint64_t(0);
    for (;;) {
        const uint64_t v = c.arrivals.load(std::memory_order_acquire);
        if (v == seen && now_ns() - t0 < window) { __builtin_ia32_pause(); continue; }   // nothing new: stay off the mutex the arriving threads need
        seen = v;
        {
            std::lock_guard<std::mutex> g(c.m);
            auto& en = c.pending;
            size_t same = 0;
            for (const Entry& e : en) same += same_shape(*e.r, request, e.pool, pool.get());
            const int64_t t = now_ns();
            const size_t active = active_threads(c, t);
            // wait for EVERY active thread, also those whose previous batch is still being queued: two half-size groups that alternate cost
            // the GPU nearly twice the time of one full group (a launch sequence over 8 objects takes ~0.8x that over 16; measured with
            // "active minus the threads inside a batch" as the target: the threads settle into two alternating halves)
            // (leading off at 88 % / 75 % of the active threads was measured too: the stragglers form a permanent second group, 16 threads drop
            // from 42 k to 33-36 k three-call ops/s)
            const size_t target = active;
            // every active thread is here (whatever it is calling) -- nobody else can join -- or this shape's batch is full, or time is up
            if (en.size() >= target || same >= MAX_BATCH || t - t0 >= window) {
                c.st.gather_ns += (uint64_t)(t - t0);
                if (en.size() > 1) {
                    int64_t first = en[0].pushed, last = en[0].pushed;
                    for (const Entry& e : en) { first = std::min(first, e.pushed); last = std::max(last, e.pushed); }
                    c.st.spread_ns += (uint64_t)(last - first);
                }
                c.st.window_expired += en.size() < target && same < MAX_BATCH;
                c.st.target_sum += target;
                std::vector<Entry> rest;
                for (const Entry& e : en) {
                    if (batch.size() < MAX_BATCH && same_shape(*e.r, request, e.pool, pool.get())) batch.push_back(e.r); else rest.push_back(e);
                }
                // this thread's own request is always part of its batch
                if (std::find(batch.begin(), batch.end(), &request) == batch.end()) {
                    for (auto it = rest.begin(); it != rest.end(); ++it) if (it->r == &request) { rest.erase(it); break; }
                    rest.push_back({batch.back(), pool.get()});
                    batch.back() = &request;
                }
                en.swap(rest);
                if (!en.empty()) {   // the lead stays taken
                    std::atomic<int>* w = &en.front().r->state;
                    w->store(PROMOTED, std::memory_order_release);
                    futex_wake(w);
                }
                else c.leader = false;

                if (batch.size() > 1) {
                    c.st.calls += batch.size(); c.st.batches += 1; c.st.largest_batch = std::max<uint64_t>(c.st.largest_batch, batch.size());
                } else c.st.uncombined += 1;
                break;
            }
        }
        __builtin_ia32_pause();
    }
    if (batch.size() == 1) return false;   // alone: the caller runs the ordinary asynchronous call
    std::exception_ptr err;
    const int64_t e0 = now_ns();
    try { execute(batch, pool); } catch (...) { err = std::current_exception(); }
    {
        std::lock_guard<std::mutex> g(c.m);
        c.st.execute_ns += (uint64_t)(now_ns() - e0);
    }
    // release the callers as a tree: this thread wakes FANOUT of them, each of those FANOUT more on its way out
    std::vector<CombineRequest*> f;
    for (CombineRequest* r : batch) if (r != &request) f.push_back(r);
    constexpr size_t F = CombineRequest::FANOUT;
    for (size_t k = 0; k < f.size(); k++) {
        f[k]->error = err;
        for (size_t i = 0; i < F; i++) f[k]->wake[i] = F * (k + 1) + i < f.size() ? f[F * (k + 1) + i] : nullptr;
    }
    std::atomic<int>* root[F];
    for (size_t i = 0; i < F; i++) root[i] = i < f.size() ? &f[i]->state : nullptr;
    for (CombineRequest* r : f) r->state.store(DONE, std::memory_order_release);   // from here on `r` may be gone
    for (std::atomic<int>* w : root) if (w) futex_wake(w);
    if (err) std::rethrow_exception(err);
    return true;
}

}  // namespace detail

}  // namespace troy
