// troyn.hip -- libtroyn.so: plan management, launch logic and the extern "C" boundary
// declared in include/troyn.h.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/troyn.h"
#include "behz_kernels.hpp"
#include "behz2_kernels.hpp"
#include "launch.hpp"
#include "crypto_kernels.hpp"
#include "bgv_kernels.hpp"
#include "ring2k_kernels.hpp"
#include "host_math.hpp"
#include "ntt_kernels.hpp"
#include "ksmac_kernels.hpp"
#include "ksmaci_kernels.hpp"
#include "poly_kernels.hpp"

using namespace troyn;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail((int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Small host tables (pointer lists, offsets) for kernels that index them travel through a per-thread ring of pinned slots: the copy is
// then truly asynchronous and the caller's array may go away at once.  (Rounds 1-3 copied from the caller's pageable array and waited for
// the STREAM before returning -- i.e. for every kernel queued before: MatmulHelper's 32 multiply-accumulate calls ran one at a time.)
// A slot is reused after the copy that last read it has completed (its event); tables above the slot size take the old path.
namespace {
struct PinnedSlot { void* host = nullptr; hipEvent_t done = nullptr; };
struct PinnedRing {
    static constexpr size_t SLOTS = 8, BYTES = 256u << 10;
    PinnedSlot slot[SLOTS];
    size_t next = 0;
};   // never destroyed: a host thread may end after the runtime has shut down
}  // namespace
static int upload_host_table(hipStream_t s, void* dev, const void* host, size_t bytes) {
    if (bytes == 0) return TROYN_OK;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    constexpr int MAX_DEVICES = 16;
    static thread_local PinnedRing* rings[MAX_DEVICES] = {};
    if (bytes > PinnedRing::BYTES || device < 0 || device >= MAX_DEVICES) {
        HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));   // `host` belongs to the caller
        return TROYN_OK;
    }
    if (!rings[device]) rings[device] = new PinnedRing;
    PinnedRing& ring = *rings[device];
    PinnedSlot& sl = ring.slot[ring.next];
    if (!sl.host) {
        // a slot is published only when it is complete (buffer AND event): a failure leaves it empty for the next call to retry
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        void* host_buf = nullptr;
        const hipError_t me = hipHostMalloc(&host_buf, PinnedRing::BYTES, hipHostMallocPortable);
        if (me != hipSuccess) { (void)hipEventDestroy(ev); return fail((int)me, std::string("hipHostMalloc: ") + hipGetErrorString(me)); }
        sl.done = ev; sl.host = host_buf;
    } else {
        HIP_TRY(hipEventSynchronize(sl.done));
    }
    ring.next = (ring.next + 1) % PinnedRing::SLOTS;
    std::memcpy(sl.host, host, bytes);
    HIP_TRY(hipMemcpyAsync(dev, sl.host, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(sl.done, s));
    return TROYN_OK;
}

#define LAUNCH_CHECK()                                                                     \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) return fail((int)e_, std::string("kernel launch: ") + hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* troyn_last_error(void) { return g_last_error.c_str(); }
extern "C" int troyn_version(void) { return TROYN_VERSION; }

extern "C" int troyn_get_primes(uint64_t factor, size_t bit_size, size_t count, uint64_t* out) {
    if (!out || factor == 0 || bit_size < 2 || bit_size > 62) return fail(TROYN_E_INVALID, "[get_primes] bad argument");
    try {
        std::vector<u64> v = host::get_primes(factor, bit_size, count);
        for (size_t i = 0; i < count; i++) out[i] = v[i];
    } catch (const std::exception& e) {
        return fail(TROYN_E_MODULUS, e.what());
    }
    return TROYN_OK;
}

extern "C" int troyn_coeff_modulus_create(size_t poly_modulus_degree, const size_t* bit_sizes, size_t n, uint64_t* out) {
    // coeff_modulus.cu:65-108
    if (!bit_sizes || !out) return fail(TROYN_E_INVALID, "[CoeffModulus::create] null argument");
    if (poly_modulus_degree > 131072 || poly_modulus_degree < 2) return fail(TROYN_E_INVALID, "[CoeffModulus::create] Invalid poly_modulus_degree.");
    if (n > 64 || n < 1) return fail(TROYN_E_INVALID, "[CoeffModulus::create] Invalid bit_sizes length.");
    for (size_t i = 0; i < n; i++)
        if (bit_sizes[i] > 60 || bit_sizes[i] < 2) return fail(TROYN_E_INVALID, "[CoeffModulus::create] Invalid max_bit_size.");
    std::vector<size_t> handed(61, 0);
    try {
        for (size_t i = 0; i < n; i++) {
            const size_t s = bit_sizes[i];
            size_t total = 0;
            for (size_t k = 0; k < n; k++) total += (bit_sizes[k] == s);
            std::vector<u64> primes = host::get_primes(2 * (u64)poly_modulus_degree, s, total);
            out[i] = primes[total - 1 - handed[s]++];   // prime_table[size].back(); pop_back()
        }
    } catch (const std::exception& e) {
        return fail(TROYN_E_MODULUS, e.what());
    }
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// in-library kernel timer (measurement hook, include/troyn.h): hipEvent pairs recorded on the launch stream
// around one named launch, so that a caller can time that kernel inside its own timed region
// ---------------------------------------------------------------------------------------
namespace {
// events belong to the device that was current when they were created: the pool and the recorded spans are kept per device, so one
// host thread that alternates between devices (select_device) never records a pair on a foreign stream
struct TimerSpan { hipEvent_t e0, e1; int device; };
struct KernelTimer {
    std::atomic<bool> enabled[TROYN_TIMER_REGIONS];
    std::vector<TimerSpan> spans[TROYN_TIMER_REGIONS];   // recorded, not yet read
    std::vector<TimerSpan> free_pairs;                   // reusable pairs of any device
    std::mutex mu;
    KernelTimer() { for (auto& e : enabled) e.store(false); }
};
KernelTimer g_ktimer;

struct TimerScope {
    int region; hipStream_t s; TimerSpan sp{nullptr, nullptr, -1};
    TimerScope(int region_, hipStream_t s_) : region(region_), s(s_) {
        if (!g_ktimer.enabled[region].load(std::memory_order_relaxed)) return;
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return;
        {
            std::lock_guard<std::mutex> g(g_ktimer.mu);
            for (size_t i = 0; i < g_ktimer.free_pairs.size(); i++)
                if (g_ktimer.free_pairs[i].device == dev) { sp = g_ktimer.free_pairs[i]; g_ktimer.free_pairs.erase(g_ktimer.free_pairs.begin() + i); break; }
        }
        if (!sp.e0) {
            sp.device = dev;
            if (hipEventCreate(&sp.e0) != hipSuccess) { sp.e0 = nullptr; return; }
            if (hipEventCreate(&sp.e1) != hipSuccess) { (void)hipEventDestroy(sp.e0); sp.e0 = sp.e1 = nullptr; return; }
        }
        if (hipEventRecord(sp.e0, s) != hipSuccess) { release(); }
    }
    void release() {
        std::lock_guard<std::mutex> g(g_ktimer.mu);
        g_ktimer.free_pairs.push_back(sp);
        sp.e0 = sp.e1 = nullptr;
    }
    ~TimerScope() {
        if (!sp.e0) return;
        if (hipEventRecord(sp.e1, s) != hipSuccess) { release(); return; }
        std::lock_guard<std::mutex> g(g_ktimer.mu);
        g_ktimer.spans[region].push_back(sp);
    }
};
}  // namespace

extern "C" int troyn_kernel_timer_enable(int region, int on) {
    if (region < 0 || region >= TROYN_TIMER_REGIONS) return fail(TROYN_E_INVALID, "[troyn_kernel_timer_enable] unknown region");
    g_ktimer.enabled[region].store(on != 0);
    return TROYN_OK;
}

extern "C" int troyn_kernel_timer_read(int region, double* total_ms, uint64_t* launches) {
    if (region < 0 || region >= TROYN_TIMER_REGIONS || !total_ms || !launches) return fail(TROYN_E_INVALID, "[troyn_kernel_timer_read] bad argument");
    std::vector<TimerSpan> spans;
    { std::lock_guard<std::mutex> g(g_ktimer.mu); spans.swap(g_ktimer.spans[region]); }
    double total = 0.0;
    hipError_t err = hipSuccess;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto& sp : spans) {
        if (err != hipSuccess) break;
        if (sp.device != cur) { if ((err = hipSetDevice(sp.device)) != hipSuccess) break; }
        float ms = 0.f;
        if ((err = hipEventSynchronize(sp.e1)) == hipSuccess && (err = hipEventElapsedTime(&ms, sp.e0, sp.e1)) == hipSuccess) total += ms;
        if (sp.device != cur) (void)hipSetDevice(cur);
    }
    {   // the pairs go back to the pool whatever happened: nothing leaks, no sample is half-read
        std::lock_guard<std::mutex> g(g_ktimer.mu);
        for (auto& sp : spans) g_ktimer.free_pairs.push_back(sp);
    }
    if (err != hipSuccess) return fail((int)err, std::string("[troyn_kernel_timer_read] ") + hipGetErrorString(err));
    *total_ms = total; *launches = spans.size();
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------
// A/B switches.  Rounds 1-4 read them from the environment on EVERY library call (getenv is not safe against a concurrent setenv, and
// host threads of the C++ mirror run the library concurrently); since round 5 they are read ONCE, when a plan is created, into the plan
// (troyn_plan_create), and changed per plan with troyn_plan_set_option(plan, "TROYN_...", "value") -- no environment access on any call path.
struct TroynOptions {
    bool ntt_u64 = false;            // TROYN_NTT_ARITH=u64: integer butterflies (and the integer inner product) for every modulus
    int ntt_split = -1;              // TROYN_NTT_SPLIT=0|1: launches over limbs of both classes as one integer launch / always split by class
    bool tensor_split = false;       // TROYN_BFV_TENSOR=split: separate transform and dyadic launches in the BFV multiply
    bool tensor_fused = false;       // TROYN_BFV_TENSOR=fused: tensor_core_kernel also for launches of a few ciphertexts (whole-limb sizes, see troyn_bfv_multiply)
    int ks_order = -1;               // TROYN_KS_ORDER=plain|item|row|band (0..3): workgroup order of the inner product kernels
    int ks_split = -1;               // TROYN_KS_SPLIT=0|1: digit-parallel inner product off / forced on
    bool ks_tail_split = false;      // TROYN_KS_TAIL=split: coefficient-form tail as separate launches
    bool ks_mac_split = false;       // TROYN_KS_MAC=split: decomposition NTT and inner product in two launches (the path of N < 1024 / N > 32768)
    bool ks_mac_fused = false;       // TROYN_KS_MAC=fused: the one-launch inner product also for small launches of chains with moduli >= 2^50 (see ks_small_mixed)
    bool ks_diag_loop = false;       // TROYN_KS_DIAG=loop: diagonal digit as an iteration of ksmac2's digit loop
    int ks_rows = 0;                 // TROYN_KS_ROWS=<r>: rows co-scheduled per XCD by the whole-limb inner product of N < 8192
    bool ks_mac_shoup_off = false;   // TROYN_KS_MAC_SHOUP=0: Barrett-128 terms in that kernel's integer form
    bool mrr_mixed_off = false;      // TROYN_MRR_MIXED=0: chains with moduli >= 2^50 compose the three calls inside the fused entry
    bool mrr_small_off = false;      // TROYN_MRR_SMALL=0: single objects at N = 16384 keep the six-launch tail of the fused chain
    bool mrr_small_serial = false;   // TROYN_MRR_SMALL=serial: one thread per quartet loops over the output limbs (no recomputation of the inverse tails)
    bool mrr_calls = false;          // TROYN_MRR=calls: the fused entry composes the three public calls
    int mrr_chunk = 0, mrr_streams = 2;   // TROYN_MRR_CHUNK=<items>, TROYN_MRR_STREAMS=<1..4>
    bool behz_v1 = false;            // TROYN_BEHZ=v1: first-generation conversion kernels (they stay the path of L > 16 and N < 1024)
    bool behz_base_ref = false;      // TROYN_BEHZ_BASE=ref: the reference's auxiliary base of 61-bit primes also when every q_i is below 2^50 (read by
                                     // troyn_behz_create; default since round 5: a base of primes below 2^50 there, "small" = that default)
    bool behz_lift_split = false;    // TROYN_BEHZ_LIFT=split: N = 32768 lifts an operand in three launches (first pass, lift, first pass) instead of one
    int plain_mac = 0;               // TROYN_PLAIN_MAC=v1|single|dual|quad (1..4): grouping of the ct x pt multiply-accumulate
    int ntt_half = -1;               // TROYN_NTT_HALF=<mask>
    bool ntt_overlap_off = false;    // TROYN_NTT_OVERLAP=0: the per-class launches of a chain with wide moduli run one after the other on the caller's stream
    bool ntt_small_two_pass_off = false;   // TROYN_NTT_SMALL_TWO_PASS=0
    int tensor_wgs = 8;              // TROYN_TENSOR_WGS=8|3|2
};
static const char* const TROYN_OPTION_NAMES[] = {"TROYN_NTT_ARITH", "TROYN_NTT_SPLIT", "TROYN_BFV_TENSOR", "TROYN_KS_ORDER", "TROYN_KS_SPLIT", "TROYN_KS_TAIL", "TROYN_KS_MAC",
    "TROYN_KS_DIAG", "TROYN_KS_ROWS", "TROYN_KS_MAC_SHOUP", "TROYN_MRR_MIXED", "TROYN_MRR_SMALL", "TROYN_MRR", "TROYN_MRR_CHUNK", "TROYN_MRR_STREAMS", "TROYN_BEHZ", "TROYN_BEHZ_BASE", "TROYN_BEHZ_LIFT",
    "TROYN_PLAIN_MAC", "TROYN_NTT_HALF", "TROYN_NTT_SMALL_TWO_PASS", "TROYN_TENSOR_WGS", "TROYN_NTT_OVERLAP"};
// value == nullptr or "": the option's default.  Returns 1 = applied, 0 = unknown name, -1 = a value this option does not have (nothing is changed:
// a misspelt value must not silently select a variant -- TROYN_TENSOR_WGS=foo used to parse as 0 and pick the two-workgroup kernel).
static int option_apply(TroynOptions& o, const char* name, const char* value) {
    const TroynOptions d;
    const std::string n = name ? name : "", v = value ? value : "";
    bool bad = false;
    // one of the listed words (index), or the default for ""; anything else is invalid
    auto word = [&](std::initializer_list<const char*> words, int dflt) {
        if (v.empty()) return dflt;
        int i = 0;
        for (const char* w : words) { if (v == w) return i; i++; }
        bad = true;
        return dflt;
    };
    auto num = [&](long lo, long hi, int dflt) {
        if (v.empty()) return dflt;
        char* end = nullptr;
        const long x = strtol(v.c_str(), &end, 0);
        if (end == v.c_str() || *end || x < lo || x > hi) { bad = true; return dflt; }
        return (int)x;
    };
    TroynOptions t = o;
    if (n == "TROYN_NTT_ARITH") t.ntt_u64 = word({"auto", "u64"}, 0) == 1;
    else if (n == "TROYN_NTT_SPLIT") t.ntt_split = num(0, 1, d.ntt_split);
    else if (n == "TROYN_BFV_TENSOR") { const int w = word({"auto", "split", "fused"}, 0); t.tensor_split = w == 1; t.tensor_fused = w == 2; }
    else if (n == "TROYN_KS_ORDER") t.ks_order = word({"plain", "item", "row", "band"}, d.ks_order);
    else if (n == "TROYN_KS_SPLIT") t.ks_split = num(0, 1, d.ks_split);
    else if (n == "TROYN_KS_TAIL") t.ks_tail_split = word({"fused", "split"}, 0) == 1;
    else if (n == "TROYN_KS_MAC") { const int w = word({"auto", "split", "fused"}, 0); t.ks_mac_split = w == 1; t.ks_mac_fused = w == 2; }
    else if (n == "TROYN_KS_DIAG") t.ks_diag_loop = word({"epilogue", "loop"}, 0) == 1;
    else if (n == "TROYN_KS_ROWS") t.ks_rows = num(0, 64, d.ks_rows);
    else if (n == "TROYN_KS_MAC_SHOUP") t.ks_mac_shoup_off = num(0, 1, 1) == 0;
    else if (n == "TROYN_MRR_MIXED") t.mrr_mixed_off = num(0, 1, 1) == 0;
    else if (n == "TROYN_MRR_SMALL") { const int w = word({"1", "0", "serial"}, 0); t.mrr_small_off = w == 1; t.mrr_small_serial = w == 2; }
    else if (n == "TROYN_MRR") t.mrr_calls = word({"fused", "calls"}, 0) == 1;
    else if (n == "TROYN_MRR_CHUNK") t.mrr_chunk = num(0, 1 << 20, d.mrr_chunk);
    else if (n == "TROYN_MRR_STREAMS") t.mrr_streams = num(1, 4, d.mrr_streams);
    else if (n == "TROYN_BEHZ") t.behz_v1 = word({"v2", "v1"}, 0) == 1;
    else if (n == "TROYN_BEHZ_BASE") t.behz_base_ref = word({"small", "ref"}, 0) == 1;
    else if (n == "TROYN_BEHZ_LIFT") t.behz_lift_split = word({"fused", "split"}, 0) == 1;
    else if (n == "TROYN_PLAIN_MAC") t.plain_mac = word({"auto", "v1", "single", "dual", "quad"}, 0);
    else if (n == "TROYN_NTT_HALF") t.ntt_half = num(0, 0xffff, d.ntt_half);
    else if (n == "TROYN_NTT_SMALL_TWO_PASS") t.ntt_small_two_pass_off = num(0, 1, 1) == 0;
    else if (n == "TROYN_NTT_OVERLAP") t.ntt_overlap_off = num(0, 1, 1) == 0;
    else if (n == "TROYN_TENSOR_WGS") { t.tensor_wgs = num(2, 8, d.tensor_wgs); if (t.tensor_wgs != 8 && t.tensor_wgs != 3 && t.tensor_wgs != 2) bad = true; }
    else return 0;
    if (bad) return -1;
    o = t;
    return 1;
}
// the ONLY place the library reads its switches from the environment (troyn_plan_create); a value an option does not have fails the creation
static int options_from_environment(TroynOptions& o) {
    for (const char* name : TROYN_OPTION_NAMES)
        if (const char* e = getenv(name))
            if (option_apply(o, name, e) < 0) return fail(TROYN_E_INVALID, std::string("[troyn_plan_create] environment: ") + name + "=" + e + " is not a value of this option");
    return TROYN_OK;
}

struct troyn_plan {
    int device = 0;
    TroynOptions opt;
    std::atomic<uint64_t> opt_gen{0};     // bumped by troyn_plan_set_option: handles that own a second plan (troyn_behz: the auxiliary base) re-read `opt` when it moved
    unsigned log_n = 0, n = 0, K = 0;
    std::vector<u64> moduli;
    std::vector<host::NttTable> tables;   // host copies (KAT hooks, BEHZ construction)
    DevModulus* d_mods = nullptr;         // [K]
    ulonglong2* d_fwd = nullptr;          // [K][N]
    ulonglong2* d_inv = nullptr;          // [K][N]
    double* d_fwd_f64 = nullptr;          // [K][N] w as double for moduli < 2^50 (zeros otherwise)
    double* d_inv_f64 = nullptr;          // [K][N]
    std::vector<char> small_modulus;      // [K] 1 iff q < 2^50 (FP64 fast path usable)
    ulonglong2* d_inv_last = nullptr;     // [(K+1)][K]: row L holds q_{L-1}^-1 mod q_i, i < L-1
    // ksmac2_kernel (N = 8192 / 16384, moduli < 2^50): the forward twiddles in the order its register rounds read them
    double* d_fwd_r1 = nullptr;           // [K][N/1024][32]
    double* d_fwd_r2 = nullptr;           // [K][N], lane-interleaved (ksm_perm)
    // ksmaci_kernel (same sizes, any modulus): the same two copies as (operand, quotient) pairs
    ulonglong2* d_fwd_r1i = nullptr;      // [K][N/1024][32]
    ulonglong2* d_fwd_r2i = nullptr;      // [K][N], lane-interleaved (ksm_perm)
};

// One host thread may drive several devices (the reference calls utils::set_device before every launch, fgk/ntt_grouped.cu:286):
// every entry point that takes a handle makes that handle's device current first.
static inline void select_device_index(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) (void)hipSetDevice(device);
}
static inline void select_device(const troyn_plan* p) { if (p) select_device_index(p->device); }
template <typename H> static inline void select_device(const H* h) { if (h && h->plan) select_device_index(h->plan->device); }


static DevModulus make_dev_modulus(u64 q, unsigned log_n, bool with_inv_n) {
    DevModulus m;
    std::memset(&m, 0, sizeof(m));
    m.q = q;
    host::BarrettRatio r = host::barrett_ratio(q);
    m.ratio_lo = r.lo; m.ratio_hi = r.hi;
    if (with_inv_n) {
        u64 ninv = 0;
        if (host::invmod(((u64)1 << log_n) % q, q, ninv)) {
            host::Shoup s = host::shoup(ninv, q);
            m.inv_n_op = s.operand; m.inv_n_quo = s.quotient;
            if (q < F64_MODULUS_LIMIT) { m.inv_n_d = (double)ninv; m.inv_n_pd = (double)ninv * (1.0 / (double)q); }
        }
    }
    if (q < F64_MODULUS_LIMIT) { m.pd = (double)q; m.inv_pd = 1.0 / (double)q; }
    return m;
}

static int plan_upload(troyn_plan* p) {
    const size_t K = p->K, n = p->n;
    std::vector<DevModulus> mods(K);
    for (size_t i = 0; i < K; i++) {
        mods[i] = make_dev_modulus(p->moduli[i], p->log_n, true);
        mods[i].inv_n_op = p->tables[i].inv_degree.operand;
        mods[i].inv_n_quo = p->tables[i].inv_degree.quotient;
        if (p->moduli[i] < F64_MODULUS_LIMIT && n >= 2) {
            // the final Gentleman-Sande layer has one twiddle (table index N-1); N^-1 is folded into it
            const u64 nw = host::mulmod(p->tables[i].inv[n - 1].operand, p->tables[i].inv_degree.operand, p->moduli[i]);
            mods[i].inv_n_w_d = (double)nw;
            mods[i].inv_n_w_pd = (double)nw * (1.0 / (double)p->moduli[i]);
        }
    }
    std::vector<ulonglong2> inv_last((K + 1) * K, make_ulonglong2(0, 0));
    for (size_t L = 2; L <= K; L++) {
        for (size_t i = 0; i + 1 < L; i++) {
            u64 inv = 0;
            if (!host::invmod(p->moduli[L - 1] % p->moduli[i], p->moduli[i], inv))
                return fail(TROYN_E_MODULUS, "[troyn_plan_create] Unable to invert q[last] mod q[i].");
            host::Shoup s = host::shoup(inv, p->moduli[i]);
            inv_last[L * K + i] = make_ulonglong2(s.operand, s.quotient);
        }
    }
    HIP_TRY(hipMalloc(&p->d_mods, K * sizeof(DevModulus)));
    HIP_TRY(hipMalloc(&p->d_fwd, K * n * sizeof(ulonglong2)));
    HIP_TRY(hipMalloc(&p->d_inv, K * n * sizeof(ulonglong2)));
    HIP_TRY(hipMalloc(&p->d_inv_last, inv_last.size() * sizeof(ulonglong2)));
    HIP_TRY(hipMemcpy(p->d_mods, mods.data(), K * sizeof(DevModulus), hipMemcpyHostToDevice));
    for (size_t i = 0; i < K; i++) {
        static_assert(sizeof(host::Shoup) == sizeof(ulonglong2), "Shoup layout");
        HIP_TRY(hipMemcpy(p->d_fwd + i * n, p->tables[i].fwd.data(), n * sizeof(ulonglong2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_inv + i * n, p->tables[i].inv.data(), n * sizeof(ulonglong2), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(p->d_inv_last, inv_last.data(), inv_last.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
    // FP64 twiddles (w as an exact double) for the moduli below 2^50
    p->small_modulus.assign(K, 0);
    HIP_TRY(hipMalloc(&p->d_fwd_f64, K * n * sizeof(double)));
    HIP_TRY(hipMalloc(&p->d_inv_f64, K * n * sizeof(double)));
    HIP_TRY(hipMemset(p->d_fwd_f64, 0, K * n * sizeof(double)));
    HIP_TRY(hipMemset(p->d_inv_f64, 0, K * n * sizeof(double)));
    std::vector<double> tmp(n);
    for (size_t i = 0; i < K; i++) {
        const u64 q = p->moduli[i];
        if (q >= F64_MODULUS_LIMIT) continue;
        p->small_modulus[i] = 1;
        for (size_t x = 0; x < n; x++) tmp[x] = (double)p->tables[i].fwd[x].operand;
        HIP_TRY(hipMemcpy(p->d_fwd_f64 + i * n, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
        for (size_t x = 0; x < n; x++) tmp[x] = (double)p->tables[i].inv[x].operand;
        HIP_TRY(hipMemcpy(p->d_inv_f64 + i * n, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    if (p->log_n >= 13 && p->log_n <= 15) {
        // round vectors of ksmac2_kernel: slot s = (1 << lvl) + g of the thread (round 2) / of the index bits above
        // bit 9 (round 1) holds the twiddle of butterfly group g of the round's layer lvl (ksmac_kernels.hpp)
        const size_t r1n = (n >> 10) * 32;
        std::vector<double> r1(K * r1n, 0.0), r2(K * n, 0.0);
        for (size_t i = 0; i < K; i++) {
            if (!p->small_modulus[i]) continue;
            const auto& fw = p->tables[i].fwd;
            for (unsigned s = 1; s < 32; s++) {
                unsigned lvl = 0;
                while ((2u << lvl) <= s) lvl++;
                const unsigned g = s - (1u << lvl);
                for (size_t th = 0; th < (n >> 10); th++) r1[i * r1n + th * 32 + s] = (double)fw[(((n >> 10) + th) << lvl) + g].operand;
                for (size_t T = 0; T < (n >> 5); T++) r2[i * n + ksm_perm((unsigned)(T * 32 + s))] = (double)fw[(((n >> 5) + T) << lvl) + g].operand;
            }
        }
        HIP_TRY(hipMalloc(&p->d_fwd_r1, r1.size() * sizeof(double)));
        HIP_TRY(hipMalloc(&p->d_fwd_r2, r2.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(p->d_fwd_r1, r1.data(), r1.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_fwd_r2, r2.data(), r2.size() * sizeof(double), hipMemcpyHostToDevice));
        // the integer copies (ksmaci_kernel), every modulus: a chain of narrow moduli can be forced onto the integer kernels (TROYN_NTT_ARITH=u64)
        std::vector<ulonglong2> r1i(K * r1n, make_ulonglong2(0, 0)), r2i(K * n, make_ulonglong2(0, 0));
        for (size_t i = 0; i < K; i++) {
            const auto& fw = p->tables[i].fwd;
            for (unsigned s = 1; s < 32; s++) {
                unsigned lvl = 0;
                while ((2u << lvl) <= s) lvl++;
                const unsigned g = s - (1u << lvl);
                for (size_t th = 0; th < (n >> 10); th++) { const auto& w = fw[(((n >> 10) + th) << lvl) + g]; r1i[i * r1n + th * 32 + s] = make_ulonglong2(w.operand, w.quotient); }
                for (size_t T = 0; T < (n >> 5); T++) { const auto& w = fw[(((n >> 5) + T) << lvl) + g]; r2i[i * n + ksm_perm((unsigned)(T * 32 + s))] = make_ulonglong2(w.operand, w.quotient); }
            }
        }
        HIP_TRY(hipMalloc(&p->d_fwd_r1i, r1i.size() * sizeof(ulonglong2)));
        HIP_TRY(hipMalloc(&p->d_fwd_r2i, r2i.size() * sizeof(ulonglong2)));
        HIP_TRY(hipMemcpy(p->d_fwd_r1i, r1i.data(), r1i.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_fwd_r2i, r2i.data(), r2i.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
    }
    return TROYN_OK;
}

static void plan_free(troyn_plan* p) {
    if (!p) return;
    if (p->d_mods) (void)hipFree(p->d_mods);
    if (p->d_fwd) (void)hipFree(p->d_fwd);
    if (p->d_inv) (void)hipFree(p->d_inv);
    if (p->d_inv_last) (void)hipFree(p->d_inv_last);
    if (p->d_fwd_f64) (void)hipFree(p->d_fwd_f64);
    if (p->d_inv_f64) (void)hipFree(p->d_inv_f64);
    if (p->d_fwd_r1) (void)hipFree(p->d_fwd_r1);
    if (p->d_fwd_r2) (void)hipFree(p->d_fwd_r2);
    if (p->d_fwd_r1i) (void)hipFree(p->d_fwd_r1i);
    if (p->d_fwd_r2i) (void)hipFree(p->d_fwd_r2i);
    delete p;
}

extern "C" int troyn_plan_create(troyn_plan** plan, int device, uint32_t log_n, uint32_t n_moduli,
                                 const uint64_t* moduli, const uint64_t* roots) {
    if (!plan || !moduli) return fail(TROYN_E_INVALID, "[troyn_plan_create] null argument");
    *plan = nullptr;
    if (log_n < 1 || log_n > 17) return fail(TROYN_E_INVALID, "[troyn_plan_create] Invalid poly_modulus_degree.");
    if (n_moduli < 1 || n_moduli > 64) return fail(TROYN_E_INVALID, "[troyn_plan_create] Invalid coeff modulus count.");
    std::unique_ptr<troyn_plan, void (*)(troyn_plan*)> p(new troyn_plan, plan_free);
    p->device = device; p->log_n = log_n; p->n = 1u << log_n; p->K = n_moduli;
    { const int orc = options_from_environment(p->opt); if (orc != TROYN_OK) return orc; }
    p->moduli.assign(moduli, moduli + n_moduli);
    for (uint32_t i = 0; i < n_moduli; i++) {
        u64 q = moduli[i];
        if ((q >> 61) != 0 || q < 2) return fail(TROYN_E_MODULUS, "[Modulus::set_value] Value can be at most 61-bit and cannot be 1.");
        for (uint32_t j = 0; j < i; j++)
            if (moduli[j] == q) return fail(TROYN_E_MODULUS, "[troyn_plan_create] coeff_modulus must be pairwise coprime.");
        try {
            p->tables.push_back(host::make_ntt_table(log_n, q, roots ? roots[i] : 0));
        } catch (const std::exception& e) {
            return fail(TROYN_E_MODULUS, e.what());
        }
    }
    HIP_TRY(hipSetDevice(device));
    int rc = plan_upload(p.get());
    if (rc != TROYN_OK) return rc;
    *plan = p.release();
    return TROYN_OK;
}

extern "C" int troyn_plan_destroy(troyn_plan* plan) {
    plan_free(plan);
    return TROYN_OK;
}

extern "C" int troyn_plan_set_option(troyn_plan* plan, const char* name, const char* value) {
    if (!plan || !name) return fail(TROYN_E_INVALID, "[troyn_plan_set_option] null argument");
    // (a handle created from this plan -- troyn_behz, troyn_bgv, ... -- reads the plan's options at call time; TROYN_BEHZ_BASE is read by troyn_behz_create)
    const int rc = option_apply(plan->opt, name, value);
    if (rc == 0) return fail(TROYN_E_INVALID, std::string("[troyn_plan_set_option] unknown option ") + name);
    if (rc < 0) return fail(TROYN_E_INVALID, std::string("[troyn_plan_set_option] ") + name + "=" + value + " is not a value of this option");
    plan->opt_gen.fetch_add(1, std::memory_order_release);
    return TROYN_OK;
}

extern "C" uint32_t troyn_plan_log_n(const troyn_plan* plan) { return plan ? plan->log_n : 0; }
extern "C" uint32_t troyn_plan_n_moduli(const troyn_plan* plan) { return plan ? plan->K : 0; }

extern "C" int troyn_plan_get_root(const troyn_plan* plan, uint32_t mi, uint64_t* root) {
    select_device(plan);
    if (!plan || !root || mi >= plan->K) return fail(TROYN_E_INVALID, "[troyn_plan_get_root] bad argument");
    *root = plan->tables[mi].root;
    return TROYN_OK;
}

extern "C" int troyn_plan_get_root_powers(const troyn_plan* plan, uint32_t mi, int inverse, uint64_t* out) {
    select_device(plan);
    if (!plan || !out || mi >= plan->K) return fail(TROYN_E_INVALID, "[troyn_plan_get_root_powers] bad argument");
    const auto& v = inverse ? plan->tables[mi].inv : plan->tables[mi].fwd;
    for (size_t i = 0; i < v.size(); i++) { out[2 * i] = v[i].operand; out[2 * i + 1] = v[i].quotient; }
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// NTT launch
// ---------------------------------------------------------------------------------------
// (the A/B switches live in the plan: TroynOptions above)
static inline bool force_integer_ntt(const troyn_plan* p) { return p->opt.ntt_u64; }
static inline LaunchCtx launch_ctx(const troyn_plan* p, hipStream_t s) { return LaunchCtx{s, p->opt.ntt_half, p->opt.ntt_small_two_pass_off, p->opt.tensor_wgs}; }

// Launches over limbs of BOTH arithmetic classes are issued once per run of limbs of one class (limbs are independent).  The integer runs are bound
// by vector-ALU issue, the FP64 runs of the memory-side kernels by memory latency / bandwidth, so two such runs in flight fill each other's idle
// resource: every other run goes to a side stream (one per host thread and device, non-blocking), forked from the caller's stream by an event and
// joined back to it before the call returns -- the caller still sees one stream order.  Only for launches large enough to fill the chip on their own
// (a handful of workgroups gains nothing from a second queue and pays two event waits).  TROYN_NTT_OVERLAP=0: one run after the other.
namespace {
struct SideStream { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
// one side stream per host thread and device, created on first use and destroyed when the host thread ends (the reference tool's -c N mode spawns such
// threads per run: round 5 leaked a stream and two events per thread).  Fixed slots: a pointer handed out stays valid while other devices are added.
struct SideStreams {
    static constexpr int MAX_DEVICES = 16;
    SideStream slot[MAX_DEVICES];
    ~SideStreams() {
        for (SideStream& x : slot) {
            if (!x.s) continue;
            (void)hipEventDestroy(x.fork); (void)hipEventDestroy(x.join); (void)hipStreamDestroy(x.s);
            (void)hipGetLastError();       // (a thread that ends while the runtime shuts down: nothing left to release)
        }
    }
};
SideStream* side_stream(int device) {
    static thread_local SideStreams pool;
    if (device < 0 || device >= SideStreams::MAX_DEVICES) return nullptr;
    SideStream& x = pool.slot[device];
    if (x.s) return &x;
    hipStream_t st = nullptr; hipEvent_t f = nullptr, j = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&j, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (f) (void)hipEventDestroy(f);
        (void)hipStreamDestroy(st);
        return nullptr;
    }
    x.fork = f; x.join = j; x.s = st;      // published complete
    return &x;
}
// runs 0, 2, 4, ... of a split launch go to the side stream, the others stay on the caller's; join() orders the caller's stream behind the side stream
struct RunOverlap {
    SideStream* ss = nullptr; hipStream_t main; bool forked = false; unsigned idx = 0;
    RunOverlap(const troyn_plan* p, hipStream_t s, bool enable) : main(s) { if (enable && !p->opt.ntt_overlap_off) ss = side_stream(p->device); }
    hipStream_t next() {
        const bool side = ss && (idx++ & 1u) == 0u;
        if (!side) return main;
        if (!forked) {
            if (hipEventRecord(ss->fork, main) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) { (void)hipGetLastError(); ss = nullptr; return main; }
            forked = true;
        }
        return ss->s;
    }
    int join() {       // also on an error path: whatever was queued on the side stream keeps writing the caller's buffers
        if (!forked) return TROYN_OK;
        forked = false;
        if (hipEventRecord(ss->join, ss->s) == hipSuccess && hipStreamWaitEvent(main, ss->join, 0) == hipSuccess) return TROYN_OK;
        (void)hipGetLastError();
        const hipError_t e = hipStreamSynchronize(ss->s);      // fallback: the host waits for the side stream instead of the caller's stream
        if (e != hipSuccess) return fail((int)e, std::string("[RunOverlap::join] ") + hipGetErrorString(e));
        return TROYN_OK;
    }
    ~RunOverlap() { (void)join(); }
};
constexpr size_t OVERLAP_MIN_LIMB_POLYS = 512;      // per split launch (all runs together)
}  // namespace

// CUs of the current device (cached per host thread; the same figure ntt_launch.inl sizes its small launches by)
static unsigned device_cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local unsigned cached = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached = (unsigned)cus; cached_dev = dev;
    }
    return cached;
}

static bool use_f64(const troyn_plan* p, unsigned table_start, unsigned table_count) {
    // FP64 butterflies when every modulus this launch can touch is below 2^50
    bool f64 = !force_integer_ntt(p) && p->log_n >= 10;
    for (unsigned i = 0; f64 && i < table_count; i++) f64 = p->small_modulus[table_start + i] != 0;
    return f64;
}

static int launch_ntt(const troyn_plan* p, NttArgs a, size_t batch, bool inverse, hipStream_t s, u64* two_pass_scratch = nullptr) {
    a.mods = p->d_mods;
    // inputs shared by several limb-polynomials of the launch (stride 0) should stay cached
    a.stream_loads = ((a.pcount <= 1 || a.in_pstride != 0) && (a.ncomp <= 1 || a.in_cstride != 0)) ? 1u : 0u;
    const size_t lp = batch * a.pcount * a.ncomp;
    if (lp == 0) return TROYN_OK;
    // limbs that share one input row (component stride 0) are co-located on an XCD by the fused forward kernels
    a.xcd_groups = ((a.load_mode != NTT_LOAD_PLAIN || a.fused_mode == NTT_FUSED_TAIL_RESCALE || a.fused_mode == NTT_FUSED_TAIL_RESCALE_W) && a.in_cstride == 0 && a.ncomp > 1) ? (unsigned)(batch * a.pcount) : 0u;
    if (lp * ((size_t)1 << (p->log_n > 12 ? p->log_n - 12 : 0)) > 0x7fffffffull)
        return fail(TROYN_E_INVALID, "[troyn_ntt] batch too large for one launch");
    bool f64 = use_f64(p, a.table_start, a.table_count);
    if (!f64 && !force_integer_ntt(p) && p->log_n >= 10 && a.mode == TROYN_IDX_COMPONENTWISE && a.ncomp > 1 && a.ncomp <= a.table_count &&
        a.fused_mode == 0 && p->opt.ntt_split != 0 &&
        // (a launch that cannot fill the chip is latency-bound: one integer launch beats two half-empty ones -- one ciphertext at N = 16384
        // {60,50,50,50,50,60}: relinearize 102 -> 80 us, rescale 51 -> 40 us)
        (p->opt.ntt_split == 1 || lp * TROYN_SMALL_LP_FACTOR > device_cu_count()) &&
        ((a.load_mode == NTT_LOAD_PLAIN && a.store_mode == NTT_STORE_PLAIN && !two_pass_scratch) || p->log_n >= 14 || p->opt.ntt_split == 1 ||
         (!p->opt.ntt_overlap_off && lp >= OVERLAP_MIN_LIMB_POLYS))) {
        // A component-wise launch over limbs of both size classes ({60,40,40,60}: the reference's default chain): split it into
        // runs of one class, so that the limbs below 2^50 take the FP64 butterflies instead of following the 60-bit limbs into the
        // integer ones.  Limbs are independent; results are unchanged.  The launches with a fused prologue / epilogue (key-switch tail,
        // rescale) can split the same way (TROYN_NTT_SPLIT=1: their per-component operands move with the run, and the FP64 loader
        // reduces the word of a wide dropped prime with integer arithmetic first, ArithF64::load_io), but they run at the same ~3 TB/s
        // under either policy at N = 8192, so the extra launch only costs there: relinearize {60,40,40,60} 731 k split vs 749 k unsplit -- not the
        // default at N <= 8192.  N >= 16384 (round 5): the integer kernels are one 1024-thread workgroup per CU (whole-limb tiles) or two strided
        // passes, the FP64 ones run on half-word tiles: {60,50,50,50,50,60} key-switch tail 835 us unsplit for 512 items, i.e. 163 ns per row
        // against 83 ns -- split by default (TROYN_NTT_SPLIT=0: one integer launch).
        bool mixed = false;
        for (unsigned j = 1; j < a.ncomp && !mixed; j++) mixed = p->small_modulus[a.table_start + j] != p->small_modulus[a.table_start];
        if (mixed) {
            RunOverlap ov(p, s, lp >= OVERLAP_MIN_LIMB_POLYS);
            unsigned j0 = 0;
            while (j0 < a.ncomp) {
                unsigned j1 = j0 + 1;
                while (j1 < a.ncomp && p->small_modulus[a.table_start + j1] == p->small_modulus[a.table_start + j0]) j1++;
                NttArgs r = a;
                r.in = a.in + (long long)j0 * a.in_cstride; r.out = a.out + (long long)j0 * a.out_cstride;
                r.ncomp = j1 - j0; r.table_start = a.table_start + j0; r.table_count = j1 - j0;
                if (a.ext0) r.ext0 = a.ext0 + (long long)j0 * a.ext0_cstride;
                if (a.ext1) r.ext1 = a.ext1 + (long long)j0 * a.ext1_cstride;
                if (a.inv_table) r.inv_table = a.inv_table + j0;
                // (two-pass sizes: every run keeps its own part of the scratch -- the runs may be in flight together)
                u64* run_scratch = two_pass_scratch ? two_pass_scratch + batch * (size_t)a.pcount * j0 * p->n : nullptr;
                if (int rc = launch_ntt(p, r, batch, inverse, ov.next(), run_scratch)) return rc;
                j0 = j1;
            }
            return ov.join();
        }
    }
    bool done;
    if (f64) {
        a.tw = inverse ? (const void*)p->d_inv_f64 : (const void*)p->d_fwd_f64;
        done = launch_ntt_f64(p->log_n, a, lp, inverse, launch_ctx(p, s), two_pass_scratch);
    } else {
        a.tw = inverse ? (const void*)p->d_inv : (const void*)p->d_fwd;
        done = launch_ntt_u64(p->log_n, a, lp, inverse, launch_ctx(p, s), two_pass_scratch);
    }
    if (!done) {
        a.tw = inverse ? (const void*)p->d_inv : (const void*)p->d_fwd;
        launch_ntt_generic(a, p->log_n, inverse, lp, launch_ctx(p, s));
    }
    LAUNCH_CHECK();
    return TROYN_OK;
}

// ---- tensor product of two 2-component ciphertexts fused with the transforms around it (tensor_core_kernel) ----
// two-pass sizes (N = 32768, 65536): stage 0 = first forward pass alone, stage 1 = last forward pass + tensor product + first inverse
// pass, stage 2 = last inverse pass.  Whole-limb sizes (N = 1024 .. 8192): stage 1 is everything.
// 0: limbs [0, ncomp) of plan p cannot take the fused tensor path; 1: whole-limb tiles; 2: two-pass transforms.  One arithmetic class.
static int tensor_path_kind(const troyn_plan* p, unsigned ncomp) {
    if (p->opt.tensor_split) return 0;   // TROYN_BFV_TENSOR=split: separate transform and dyadic launches (A/B runs, tests of that path)
    // whole-limb tiles; limbs of both classes: one launch per run of one class (tensor_stage).  N = 16384 holds three polynomials of
    // 16 coefficients per thread under the 128-register cap of a 1024-thread workgroup only with 78-93 spilled registers, and is
    // still 8 % faster than the separate launches (83.8 k vs 77.2 k products/s at 6 x 50-bit)
    if (p->log_n >= 10 && p->log_n <= 14) return 1;
    for (unsigned j = 1; j < ncomp; j++) if (p->small_modulus[j] != p->small_modulus[0]) return 0;
    if (p->log_n == 15 || p->log_n == 16) return 2;
    return 0;
}
static int tensor_stage(const troyn_plan* p, int stage, NttArgs a, NttArgs b, NttArgs d, size_t batch, hipStream_t s) {
    // limbs of both arithmetic classes ({60,40,40,60}): one launch per run of limbs of one class (limbs are independent)
    for (unsigned j0 = 0, j1; j0 < a.ncomp; j0 = j1) {
        for (j1 = j0 + 1; j1 < a.ncomp && p->small_modulus[a.table_start + j1] == p->small_modulus[a.table_start + j0]; j1++) {}
        if (j0 == 0 && j1 == a.ncomp) break;   // one class: fall through to the single launch
        auto sub = [&](NttArgs x) {
            x.in += (long long)j0 * x.in_cstride; if (x.out) x.out += (long long)j0 * x.out_cstride;
            x.ncomp = j1 - j0; x.table_start += j0; x.table_count = j1 - j0;
            return x;
        };
        if (int rc = tensor_stage(p, stage, sub(a), sub(b), sub(d), batch, s)) return rc;
        if (j1 == a.ncomp) return TROYN_OK;
    }
    const bool f64 = use_f64(p, a.table_start, a.ncomp);
    auto prep = [&](NttArgs& x, bool inverse) {
        x.mods = p->d_mods; x.stream_loads = 1u; x.xcd_groups = 0u;
        x.tw = f64 ? (inverse ? (const void*)p->d_inv_f64 : (const void*)p->d_fwd_f64) : (inverse ? (const void*)p->d_inv : (const void*)p->d_fwd);
    };
    prep(a, stage == 2); prep(b, false); prep(d, true);
    if ((batch * a.pcount * a.ncomp) << (p->log_n > 12 ? p->log_n - 12 : 0) > 0x7fffffffull) return fail(TROYN_E_INVALID, "[troyn_ntt] batch too large for one launch");
    if (!(f64 ? launch_tensor_f64(p->log_n, stage, a, b, d, batch, launch_ctx(p, s)) : launch_tensor_u64(p->log_n, stage, a, b, d, batch, launch_ctx(p, s))))
        return fail(TROYN_E_INVALID, "[troyn_bfv_multiply] no fused tensor kernel for this size");
    LAUNCH_CHECK();
    return TROYN_OK;
}

static NttArgs contiguous_args(const troyn_plan* p, const u64* in, u64* out, size_t pcount, size_t ncomp,
                               unsigned table_start, unsigned table_count, int mode, unsigned decomp) {
    NttArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out;
    a.in_cstride = a.out_cstride = p->n;
    a.in_pstride = a.out_pstride = (long long)ncomp * p->n;
    a.in_bstride = a.out_bstride = (long long)pcount * ncomp * p->n;
    a.pcount = (unsigned)pcount; a.ncomp = (unsigned)ncomp;
    a.table_start = table_start; a.table_count = table_count; a.mode = (unsigned)mode; a.decomp = decomp;
    return a;
}

extern "C" int troyn_ntt(const troyn_plan* plan, int inverse, const uint64_t* in, uint64_t* out,
                         size_t batch, size_t pcount, size_t ncomp,
                         uint32_t table_start, uint32_t table_count, int indexer_mode, uint32_t decomp_size,
                         troyn_stream_t stream) {
    select_device(plan);
    if (!plan || !in || !out) return fail(TROYN_E_INVALID, "[troyn_ntt] null argument");
    if (table_count == 0 || table_start + table_count > plan->K)
        return fail(TROYN_E_INVALID, "[troyn_ntt] table slice out of range");
    if (indexer_mode == TROYN_IDX_COMPONENTWISE && ncomp > table_count)
        return fail(TROYN_E_INVALID, "[troyn_ntt] more components than tables");
    if (indexer_mode == TROYN_IDX_KS_SET_PRODUCTS && (pcount > decomp_size + 1 || decomp_size >= table_count + 0u + 1u))
        return fail(TROYN_E_INVALID, "[troyn_ntt] bad KeySwitchingSetProducts shape");
    if (indexer_mode == TROYN_IDX_KS_SKIP_FINALS && (ncomp > decomp_size + 1 || decomp_size > table_count))
        return fail(TROYN_E_INVALID, "[troyn_ntt] bad KeySwitchingSkipFinals shape");
    NttArgs a = contiguous_args(plan, (const u64*)in, (u64*)out, pcount, ncomp, table_start, table_count, indexer_mode, decomp_size);
    return launch_ntt(plan, a, batch, inverse != 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// streaming ops
// ---------------------------------------------------------------------------------------
static inline unsigned chunks_pairs(unsigned n) { unsigned c = (n / 2 + POLY_BLOCK - 1) / POLY_BLOCK; return c ? c : 1; }
static inline unsigned chunks_single(unsigned n) { unsigned c = (n + 255) / 256; return c ? c : 1; }

static int check_rows(size_t rows, unsigned chunks) {
    if (rows * chunks > 0x7fffffffull) return fail(TROYN_E_INVALID, "[troyn] batch too large for one launch");
    return TROYN_OK;
}

template <int OP>
static int launch_elementwise(const troyn_plan* p, uint32_t mod_start, uint32_t nmod, const uint64_t* a, const uint64_t* b,
                              uint64_t scalar, uint64_t* out, size_t count, troyn_stream_t stream) {
    if (!p || !a || !out || (OP != EW_NEG && OP != EW_MULS && OP != EW_MOD && !b)) return fail(TROYN_E_INVALID, "[troyn elementwise] null argument");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, "[troyn elementwise] modulus slice out of range");
    const size_t rows = count * nmod;
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_pairs(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL((elementwise_kernel<OP>), dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, mod_start, nmod, p->n, (const u64*)a, (const u64*)b, (u64)scalar, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_add(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_ADD>(p, ms, nm, a, b, 0, out, count, s);
}
extern "C" int troyn_sub(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_SUB>(p, ms, nm, a, b, 0, out, count, s);
}
extern "C" int troyn_negate(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_NEG>(p, ms, nm, a, nullptr, 0, out, count, s);
}
extern "C" int troyn_multiply_scalar(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, uint64_t scalar, uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_MULS>(p, ms, nm, a, nullptr, scalar, out, count, s);
}
extern "C" int troyn_dyadic_product(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_MUL>(p, ms, nm, a, b, 0, out, count, s);
}

extern "C" int troyn_modulo(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_MOD>(p, ms, nm, a, nullptr, 0, out, count, s);
}
extern "C" int troyn_multiply_uint64operand(const troyn_plan* p, uint32_t ms, uint32_t nm, const uint64_t* a, const uint64_t* operands,
                                            uint64_t* out, size_t count, troyn_stream_t s) {
    select_device(p);
    return launch_elementwise<EW_MULOP>(p, ms, nm, a, operands, 0, out, count, s);
}

static int launch_convolute(const DevModulus* mods, unsigned n, uint32_t mod_start, uint32_t nmod,
                            const u64* a, size_t pa, const u64* b, size_t pb, u64* out, size_t batch, hipStream_t s) {
    const size_t rows = batch * nmod;
    if (rows == 0) return TROYN_OK;
    if (pa == 2 && pb == 2) {
        const unsigned ch = chunks_pairs(n);
        if (int rc = check_rows(rows, ch)) return rc;
        hipLaunchKernelGGL((dyadic_convolute_kernel<2, 2>), dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s, ch, mods, mod_start, nmod, n, a, b, out);
    } else if (pa == 3 && pb == 2) {
        const unsigned ch = chunks_pairs(n);
        if (int rc = check_rows(rows, ch)) return rc;
        hipLaunchKernelGGL((dyadic_convolute_kernel<3, 2>), dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s, ch, mods, mod_start, nmod, n, a, b, out);
    } else if (pa == 2 && pb == 3) {
        const unsigned ch = chunks_pairs(n);
        if (int rc = check_rows(rows, ch)) return rc;
        hipLaunchKernelGGL((dyadic_convolute_kernel<2, 3>), dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s, ch, mods, mod_start, nmod, n, a, b, out);
    } else {
        const unsigned ch = chunks_single(n);
        if (int rc = check_rows(rows, ch)) return rc;
        hipLaunchKernelGGL(dyadic_convolute_generic_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s, ch, mods, mod_start, nmod, n,
                           a, (unsigned)pa, b, (unsigned)pb, out);
    }
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_dyadic_convolute(const troyn_plan* p, uint32_t mod_start, uint32_t nmod,
                                      const uint64_t* a, size_t pa, const uint64_t* b, size_t pb, uint64_t* out,
                                      size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !a || !b || !out) return fail(TROYN_E_INVALID, "[fgk::dyadic_convolute] null argument");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, "[fgk::dyadic_convolute] modulus slice out of range");
    if (pa < 1 || pb < 1 || pa > 16 || pb > 16) return fail(TROYN_E_INVALID, "[fgk::dyadic_convolute] Result size mismatch");
    return launch_convolute(p->d_mods, p->n, mod_start, nmod, (const u64*)a, pa, (const u64*)b, pb, (u64*)out, batch, (hipStream_t)stream);
}

extern "C" int troyn_dyadic_square(const troyn_plan* p, uint32_t mod_start, uint32_t nmod,
                                   const uint64_t* a, uint64_t* out, size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !a || !out) return fail(TROYN_E_INVALID, "[fgk::dyadic_square] null argument");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, "[fgk::dyadic_square] modulus slice out of range");
    const size_t rows = batch * nmod;
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_pairs(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(dyadic_square_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, mod_start, nmod, p->n, (const u64*)a, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// key switching
// ---------------------------------------------------------------------------------------
// Shoup quotients of the evaluation keys for the integer inner product (ks_mac_kernel<ArithU64>, NttArgs::key_quo):
// out[j][row][i] = floor(keys[j][row][i] * 2^64 / q_row), EXACT (estimate from the Barrett ratio, corrected with the 128-bit remainder), so
// that a <digit, key> term is a lazy Shoup product in [0, 4q) instead of a 128-bit product and its Barrett reduction.  Once per call.
static __global__ __launch_bounds__(256) void ks_key_quotients_kernel(KeyPtrs keys, unsigned L, unsigned K, unsigned n, const DevModulus* mods, u64* out) {
    const size_t per_key = (size_t)2 * K * n, total = (size_t)L * per_key;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned j = (unsigned)(p / per_key);
        const size_t w = p % per_key;
        const DevModulus md = mods[(w / n) % K];
        const u64 k = keys.p[j][w];
        u64 e = k * md.ratio_hi + mul_hi(k, md.ratio_lo);            // floor(k floor(2^128/q) / 2^64): the quotient or up to two below it (k < q)
        u128 rem = ((u128)k << 64) - (u128)e * md.q;
        while (rem >= md.q) { ++e; rem -= md.q; }
        out[p] = e;
    }
}

// workgroup order of ksmac2_kernel (TROYN_KS_ORDER=plain|item|row|band):
//   0 plain: workgroups of an item dealt round-robin to the XCDs
//   1 item:  all workgroups of an item on one XCD (digits L2-resident, keys from the Infinity Cache)
//   2 row:   one output row at a time on the whole chip (keys L2-resident, digits re-fetched per row)
//   3 band:  per XCD, two rows x as many items as fill its 64 workgroup slots (keys of the band L2-resident, digits fetched once per band);
//            needs the batch to be a multiple of 8 x 64 / (2 tiles per row) items.  Default for N >= 16384 where it applies.
static unsigned ksmac_order(const troyn_plan* p, size_t batch, unsigned log_n) {
    const int want = p->opt.ks_order;
    if (want == 0) return 0u;
    if (batch % 8 != 0) return 0u;
    if (want == 2) return 2u;
    if (want == 1) return 1u;
    const size_t tiles = log_n >= 13 ? (size_t)1 << (log_n - 13) : 1, band_items = 8 * (64 / (2 * tiles));
    if (batch % band_items != 0) return 1u;
    // measured (bench.py other_configs, band vs item): N = 32768 relinearize +8 %, N = 16384 +2 % (the launch itself -4 %), N = 8192 -2..-7 %
    // (a row's keys are 0.4 MB there: every row already fits in L2 under the item order)
    return (log_n >= 14 || want == 3) ? 3u : 1u;
}

// The digit-parallel form of the inner product (ksmac2 SPLITJ + ksmac_split_reduce_kernel) pays when the plain form would leave most of the
// chip idle: batch * rows * tiles workgroups walking L digits one after the other (a single N = 16384 ciphertext: 12 workgroups, 85 us).
// TROYN_KS_SPLIT=0 / 1 forces it off / on (A/B runs, tests); it needs L >= 2 and its slots in the workspace.
static bool ksmac_split_wanted(const troyn_plan* p, size_t batch, unsigned L, unsigned log_n) {
    const int e = p->opt.ks_split;
    // L <= 15: the reducer adds the L re-centred slots (|slot| <= p/2 + 1, p < 2^50) in plain doubles, exact below 2^53 (ksmac_split_reduce_kernel)
    if (e == 0 || L < 2 || L > 15 || log_n < 13 || log_n > 15) return false;
    const size_t wgs = batch * (L + 1) << (log_n - 13);
    if (batch > 64) return false;                      // small batches only (the slots cost L times the inner product's output)
    return e == 1 || wgs <= 128;
}
// slots of the digit-parallel form: provisioned exactly when a call of this shape would take it (both read the same switch)
static size_t ks_split_words(const troyn_plan* p, size_t batch, unsigned L, unsigned log_n) {
    return ksmac_split_wanted(p, batch, L, log_n) ? batch * L * 2 * (size_t)(L + 1) * ((size_t)1 << log_n) : 0;
}


// Single objects at N = 16384 under the FP64 policy: "INTT of one row, then a forward transform of `fw.ncomp` limbs whose loader reads that row"
// in three launches instead of four -- first inverse pass, mrr_quartet_load_kernel (last inverse layers + loader + first forward layers on the
// shared quartets, troyn_mrr_small.hip), last forward pass with the epilogue of `fw`.  pa: the first inverse pass (in -> out); iv.in = pa.out.
static bool small_tail_wanted(const troyn_plan* p, size_t limb_polys) {
    // (N = 32768 transforms are two-pass at every size; the merged form is taken for the same small launches)
    return p->log_n >= 13 && p->log_n <= 15 && !p->opt.mrr_small_off && (p->log_n == 15 || !p->opt.ntt_small_two_pass_off) && limb_polys * TROYN_SMALL_LP_FACTOR <= device_cu_count();
}
static int small_tail(const troyn_plan* p, bool f64, NttArgs pa, size_t pa_limb_polys, NttArgs fw, u64* between, size_t groups, hipStream_t s) {
    const LaunchCtx lc = launch_ctx(p, s);
    auto prep = [&](NttArgs& x, bool inverse) {
        x.mods = p->d_mods; x.stream_loads = 1u; x.xcd_groups = 0u;
        x.tw = f64 ? (inverse ? (const void*)p->d_inv_f64 : (const void*)p->d_fwd_f64) : (inverse ? (const void*)p->d_inv : (const void*)p->d_fwd);
    };
    auto pass = [&](int which, const NttArgs& x, size_t lp) {
        if (f64) launch_ntt_f64_small_pass(p->log_n, which, x, lp, lc); else launch_ntt_u64_small_pass(p->log_n, which, x, lp, lc);
    };
    prep(pa, true);
    pass(0, pa, pa_limb_polys);
    LAUNCH_CHECK();
    NttArgs iv = pa;
    iv.in = pa.out; iv.in_bstride = pa.out_bstride; iv.in_pstride = pa.out_pstride; iv.in_cstride = pa.out_cstride;
    prep(fw, false);
    NttArgs first = fw;                 // the pass in between: [group][limb][N], contiguous (the layout launch_two_pass gives a scratch buffer)
    const long long N = (long long)p->n;
    first.out = between; first.out_cstride = N; first.out_pstride = (long long)fw.ncomp * N; first.out_bstride = (long long)fw.pcount * fw.ncomp * N;
    launch_mrr_quartet_load(p->log_n, groups, iv, first, s, f64);
    LAUNCH_CHECK();
    NttArgs second = fw;
    second.in = first.out; second.in_bstride = first.out_bstride; second.in_pstride = first.out_pstride; second.in_cstride = first.out_cstride;
    second.reduce_input = 0;
    pass(1, second, groups * fw.ncomp);
    LAUNCH_CHECK();
    return TROYN_OK;
}

// A chain with moduli of 2^50 and more, a few ciphertexts: its one-launch inner products (ksmac2<WIDE> + ksmaci) loop over the L digits inside
// 2 .. 4 workgroups per item and have no digit-parallel form, so a launch that cannot fill the chip is one long chain (one ciphertext at
// N = 16384 {60,50,50,50,50,60}: 148 us of inner product).  The two-launch form -- (L + 1) L digit transforms, then the multiply-accumulate -- spreads
// the same work over L times the workgroups: relinearize of one ciphertext 214 -> 84 us (N = 16384), 140 -> 78 us (N = 8192 {60,40,40,60}); equal at
// 256 workgroups of the one-launch form (N = 16384: 32 items; N = 8192: ~100), which is where this rule hands over.  TROYN_KS_MAC=fused / split force either.
// The first-generation kernel of N = 1024 .. 4096 (one workgroup per output row looping over the L digits, no digit-parallel form either) hands over
// the same way below 128 workgroups: one ciphertext at N = 4096, relinearize 40 -> 31 us (4 x 36-bit), 59 -> 42 us ({50,55,50}).
static bool ks_small_mixed(const troyn_plan* p, unsigned L, size_t batch) {
    if (p->opt.ks_mac_fused) return false;
    if (p->log_n >= 10 && p->log_n <= 12) return batch * (size_t)(L + 1) <= 128;
    if (p->log_n < 13 || p->log_n > 15) return false;
    if (use_f64(p, 0, L) && use_f64(p, p->K - 1, 1)) return false;
    return (batch * (size_t)(L + 1) << (p->log_n - 13)) <= (p->log_n == 13 ? 384u : 256u);      // (N = 8192: equal at ~400, 96 items of {60,40,40,60})
}

struct KsLayout {
    size_t target_intt, temp_ntt, poly_prod, prod_intt, temp_last, keys_f64, split, keys_quo, total;  // element offsets
};

static KsLayout ks_layout(const troyn_plan* p, unsigned L, size_t batch) {
    const size_t n = p->n;
    KsLayout w;
    size_t off = 0;
    w.target_intt = off; off += batch * L * n;
    w.temp_ntt = off;    off += batch * (size_t)(L + 1) * L * n;
    w.poly_prod = off;   off += batch * 2 * (size_t)(L + 1) * n;
    w.prod_intt = off;   off += batch * 2 * (size_t)(L + 1) * n;
    w.temp_last = off;   off += batch * 2 * (size_t)L * n;
    w.keys_f64 = off;    off += (size_t)L * 2 * p->K * n + (size_t)L * 2 * n;     // prepared keys of ksmac2_kernel + the diagonal blocks in natural order
    w.split = off;       off += ks_split_words(p, batch, L, p->log_n);                        // slots of the digit-parallel inner product (small batches)
    // Shoup quotients of the keys for the integer inner product: chains with a modulus of 2^50 or more on the whole-limb sizes
    // (N >= 8192: (key, quotient) pairs of the wide rows in the accumulators' layout + the diagonal blocks in natural order, ksmaci_kernel)
    w.keys_quo = off;    off += (p->log_n >= 10 && p->log_n <= 15 && !use_f64(p, 0, p->K)) ? (size_t)L * 2 * p->K * n * (p->log_n >= 13 ? 2 : 1) + (p->log_n >= 13 ? (size_t)p->K * 2 * n * 2 : 0) : 0;
    w.total = off;
    return w;
}

extern "C" size_t troyn_switch_key_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t batch) {
    if (!plan) return 0;
    return ks_layout(plan, L, batch).total * sizeof(u64);
}
extern "C" size_t troyn_relinearize_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t batch) {
    return troyn_switch_key_workspace_bytes(plan, L, batch);
}

// BGV divides by the special prime with a correction computed mod t (ski_util5); everything before the tail is shared.
struct BgvTail { DevModulus t; u64 inv_special_mod_t; };

// target: [batch] items of L limbs, `target_bstride` elements apart.
static bool coeff_tail_fused(const troyn_plan* p) { return !p->opt.ks_tail_split; }   // TROYN_KS_TAIL=split: inverse transforms and ski_util7 in separate launches

static int switch_key_impl(const troyn_plan* p, unsigned L, int is_ckks, int is_ntt_form,
                           const u64* target, size_t target_bstride, const uint64_t* const* keys, int assign_method,
                           u64* dest, const u64* addend, size_t addend_bstride,
                           void* workspace, size_t workspace_bytes, size_t batch, hipStream_t s, const BgvTail* bgv = nullptr) {
    const unsigned K = p->K, n = p->n;
    if (K < 2) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] Keyswitching is not supported.");
    if (L < 1 || L > K - 1) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] Invalid target size.");
    if (!target || !keys || !dest || !workspace) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] null argument");
    if (assign_method < 0 || assign_method > 2) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] bad assign method");
    const KsLayout w = ks_layout(p, L, batch);
    if (workspace_bytes < w.total * sizeof(u64)) return fail(TROYN_E_WORKSPACE, "[troyn_switch_key] workspace too small");
    if (batch == 0) return TROYN_OK;
    u64* ws = (u64*)workspace;
    KeyPtrs kp;
    std::memset(&kp, 0, sizeof(kp));
    for (unsigned j = 0; j < L; j++) {
        if (!keys[j]) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] null key pointer");
        kp.p[j] = (const u64*)keys[j];
    }
    int rc;
    const u64* digits_src = target;
    size_t digits_bstride = target_bstride;
    // the optimised NTT kernels carry fused prologues / epilogues; tiny rings (generic kernel) use the unfused chain
    const bool fused = is_ntt_form && p->log_n >= 10 && !bgv;

    // (1) NTT form: bring the target back to coefficient form (evaluator_keyswitching_core.cu:817-821)
    if (is_ntt_form) {
        NttArgs a = contiguous_args(p, target, ws + w.target_intt, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0);
        a.in_bstride = (long long)target_bstride;
        if ((rc = launch_ntt(p, a, batch, true, s))) return rc;
        digits_src = ws + w.target_intt;
        digits_bstride = (size_t)L * n;
    }
    const bool ks_unfused_mac = p->opt.ks_mac_split || ks_small_mixed(p, L, batch);
    // (2)+(3) in ONE launch for whole-limb rings (N <= 16384): every workgroup owns one output row of one item,
    //     transforms that row's L digits one after the other and multiplies them into register accumulators with
    //     the key (kernel_set_accumulate + ntt + kernel_accumulate_products, fgk/switch_key.cu:6-154); the
    //     (L+1)*L transformed digits never reach HBM.
    // (N >= 8192: the half-tile kernels, whose grid is up to 4 (L + 2) workgroups per item; the first-generation whole-limb kernel stays the path
    // of N = 1024 .. 4096 -- its N = 8192 / 16384 instantiations left the library in round 5)
    const bool mac_fused = !ks_unfused_mac && p->log_n >= 10 && p->log_n <= 15 && batch * (size_t)(L + 1) <= 0x7fffffffull &&
                           (p->log_n <= 12 || (p->d_fwd_r2 && p->d_fwd_r2i && L + 1 <= 64 && batch * (size_t)(L + 2) * 4 <= 0x7fffffffull));
    const int ks_mac_gen = p->log_n >= 13 ? 2 : 1;
    // (the band order pads an odd row count with one row of workgroups that exit: the grid guard counts L + 2 rows)
    if (mac_fused && ks_mac_gen == 2 && use_f64(p, 0, K)) {
        // ksmac2_kernel: tiles of 2^13 outputs, two workgroups per CU, keys prepared once per call (ksmac_kernels.hpp)
        double* kf = reinterpret_cast<double*>(ws + w.keys_f64);
        // NTT-form target: the block (key j, modulus j) a second time in natural order -- the diagonal digit is applied in the kernel's
        // epilogue (DG).  The switch is read ONCE per call: the preparation and the instantiation choice must agree.
        const bool dg = is_ntt_form && !p->opt.ks_diag_loop;
        // the digit-parallel form (small launches) reads the caller's keys as they are: no preparation pass
        const bool split = ksmac_split_wanted(p, batch, L, p->log_n) && (dg || !is_ntt_form);
        if (!split) {
            const size_t pairs = (size_t)L * 2 * K * (n / 2);
            const unsigned blocks = (unsigned)std::min<size_t>((pairs + 255) / 256, 4096);
            launch_ksmac_prepare_keys(kp, L, 2 * K, n, kf, blocks, s, nullptr, nullptr, 0, dg ? kf + (size_t)L * 2 * K * n : nullptr);
            LAUNCH_CHECK();
        }
        KsMacArgs a;
        std::memset(&a, 0, sizeof(a));
        if (dg) a.diag_keys = kf + (size_t)L * 2 * K * n;
        a.digits = digits_src; a.dig_bstride = (long long)digits_bstride; a.dig_cstride = n;
        a.diag = is_ntt_form ? target : nullptr; a.diag_bstride = (long long)target_bstride; a.diag_cstride = n;
        a.out = ws + w.poly_prod; a.out_bstride = 2ll * (L + 1) * n; a.out_pstride = (long long)(L + 1) * n; a.out_cstride = n;
        a.mods = p->d_mods; a.tw = p->d_fwd_f64; a.tw_r1 = p->d_fwd_r1; a.tw_r2 = p->d_fwd_r2;
        a.keys = kf; a.key_jstride = 2ll * K * n; a.key_pstride = (long long)K * n;
        a.L = L; a.table_start = 0; a.table_count = K; a.batch = (unsigned)batch;
        a.grouped = ksmac_order(p, batch, p->log_n);
        {
            TimerScope ts(TROYN_TIMER_KS_INNER_PRODUCT, s);
            if (split) {
                a.grouped = 0;
                a.part = reinterpret_cast<double*>(ws + w.split); a.part_jstride = (long long)batch * a.out_bstride;
                a.split_skip_diag = dg ? 1u : 0u;
                a.raw = kp; a.raw_pstride = (long long)K * n;
                launch_ksmac2_split(p->log_n, batch, a, s, false, dg ? 2 : 0);
            } else launch_ksmac2(p->log_n, batch, L + 1, a, s);
        }
        LAUNCH_CHECK();
    } else if (mac_fused) {
        // A chain with moduli of 2^50 and more (the reference's default {60,40,40,60}; {60,50,...,60} CKKS chains): the output rows of the
        // moduli below 2^50 take ksmac2_kernel (exact FP64 butterflies; digits of wider limbs are reduced while loading), the rows of the wide
        // moduli the integer kernel of the same shape (ksmaci_kernel, N = 8192 / 16384 / 32768; round 5).  Rows are independent; results are
        // unchanged.  TROYN_NTT_ARITH=u64 sends every row to the integer kernel; N < 8192 keeps the first-generation kernel.
        unsigned long long small_rows = 0, wide_rows = 0;
        bool wide_digits = false;
        const bool all_integer = force_integer_ntt(p);
        for (unsigned k = 0; k <= L; k++) {
            const unsigned mrow = (k == L) ? K - 1 : k;
            if (p->small_modulus[mrow] && !all_integer) small_rows |= 1ull << k; else wide_rows |= 1ull << k;
            if (k < L && !p->small_modulus[k]) wide_digits = true;
        }
        const bool gen2 = ks_mac_gen == 2;
        const bool mixed = gen2 && small_rows != 0;
        TimerScope ts(TROYN_TIMER_KS_INNER_PRODUCT, s);
        if (mixed) {
            double* kf = reinterpret_cast<double*>(ws + w.keys_f64);
            const size_t pairs = (size_t)L * 2 * K * (n / 2);
            // NTT-form target: the diagonal digit in the epilogue (DG) as in the all-FP64 path -- needs the diagonal key blocks in natural order
            const bool dg = is_ntt_form && !p->opt.ks_diag_loop;
            launch_ksmac_prepare_keys(kp, L, 2 * K, n, kf, (unsigned)std::min<size_t>((pairs + 255) / 256, 4096), s, nullptr, nullptr, 0,
                                      dg ? kf + (size_t)L * 2 * K * n : nullptr);
            LAUNCH_CHECK();
            KsMacArgs m;
            std::memset(&m, 0, sizeof(m));
            if (dg) m.diag_keys = kf + (size_t)L * 2 * K * n;
            m.digits = digits_src; m.dig_bstride = (long long)digits_bstride; m.dig_cstride = n;
            m.diag = is_ntt_form ? target : nullptr; m.diag_bstride = (long long)target_bstride; m.diag_cstride = n;
            m.out = ws + w.poly_prod; m.out_bstride = 2ll * (L + 1) * n; m.out_pstride = (long long)(L + 1) * n; m.out_cstride = n;
            m.mods = p->d_mods; m.tw = p->d_fwd_f64; m.tw_r1 = p->d_fwd_r1; m.tw_r2 = p->d_fwd_r2;
            m.keys = kf; m.key_jstride = 2ll * K * n; m.key_pstride = (long long)K * n;
            m.L = L; m.table_start = 0; m.table_count = K; m.batch = (unsigned)batch;
            m.grouped = ksmac_order(p, batch, p->log_n);
            m.row_mask = small_rows;
            launch_ksmac2(p->log_n, batch, (unsigned)__builtin_popcountll(small_rows), m, s, false, wide_digits);
            LAUNCH_CHECK();
        }
        if (gen2 && wide_rows == 0) {
            // (every row this level touches is narrow although the chain holds a wide modulus elsewhere: ksmac2 above took all of them)
        } else if (gen2) {
            // integer rows: (key, Shoup quotient) pairs of exactly these rows, in the accumulators' layout, once per call
            const unsigned slots = (unsigned)__builtin_popcountll(wide_rows);
            ulonglong2* ki = reinterpret_cast<ulonglong2*>(ws + w.keys_quo);
            ulonglong2* kdiag = ki + (size_t)L * 2 * slots * n;
            const size_t words = (size_t)L * 2 * slots * n;
            launch_ksmaci_prepare_keys(kp, L, K, n, wide_rows, ki, (unsigned)std::min<size_t>((words + 255) / 256, 4096), s, nullptr, p->d_mods, 0,
                                       is_ntt_form ? kdiag : nullptr);
            LAUNCH_CHECK();
            KsMacIArgs m;
            std::memset(&m, 0, sizeof(m));
            m.digits = digits_src; m.dig_bstride = (long long)digits_bstride; m.dig_cstride = n;
            m.diag = is_ntt_form ? target : nullptr; m.diag_bstride = (long long)target_bstride; m.diag_cstride = n;
            m.out = ws + w.poly_prod; m.out_bstride = 2ll * (L + 1) * n; m.out_pstride = (long long)(L + 1) * n; m.out_cstride = n;
            m.mods = p->d_mods; m.tw = p->d_fwd; m.tw_r1 = p->d_fwd_r1i; m.tw_r2 = p->d_fwd_r2i;
            m.keys = ki; m.key_jstride = 2ll * slots * n; m.key_pstride = (long long)slots * n;
            m.diag_keys = kdiag;
            m.L = L; m.table_start = 0; m.table_count = K; m.batch = (unsigned)batch;
            m.grouped = (batch % 8 == 0 && p->opt.ks_order != 0) ? 1u : 0u;
            m.row_mask = wide_rows;
            launch_ksmaci(p->log_n, batch, m, s, is_ntt_form ? 1 : 0);
            LAUNCH_CHECK();
        } else {
        NttArgs a = contiguous_args(p, digits_src, ws + w.poly_prod, 1, L + 1, 0, K, TROYN_IDX_KS_SET_PRODUCTS, L);
        a.in_bstride = (long long)digits_bstride; a.in_pstride = 0; a.in_cstride = n;
        a.out_bstride = 2ll * (L + 1) * n; a.out_pstride = (long long)(L + 1) * n; a.out_cstride = n;
        a.reduce_input = 1;
        a.stream_loads = 0;                      // the L+1 rows of an item re-read the same digits
        a.skip_diag = is_ntt_form ? 1 : 0;       // digit k of row k is the NTT-form input limb itself
        a.ext0 = target; a.ext0_bstride = (long long)target_bstride; a.ext0_cstride = n;
        a.batch = (unsigned)batch; a.key_pstride = (long long)K * n;
        {
            // rows of an item co-scheduled per XCD: all L+1 by default (the item's digits are then fetched once per XCD instead of
            // once per row; measured 1265 vs 1310 us per 512-item launch at cfg3), TROYN_KS_ROWS=1 restores plain row-major order
            const int ks_rows = p->opt.ks_rows;
            const unsigned R = ks_rows > 0 ? (unsigned)ks_rows : L + 1;
            a.xcd_groups = (R > 1 && batch % 8 == 0 && (L + 1) % R == 0) ? R : 0u;
        }
        a.mods = p->d_mods;
        const bool f64 = use_f64(p, 0, K);
        a.tw = f64 ? (const void*)p->d_fwd_f64 : (const void*)p->d_fwd;
        if (!f64 && !p->opt.ks_mac_shoup_off) {
            // integer policy: the keys' Shoup quotients, once per call (TROYN_KS_MAC_SHOUP=0: Barrett-128 terms as in rounds 1-3; A/B, tests)
            u64* kq = ws + w.keys_quo;
            const size_t words = (size_t)L * 2 * K * n;
            hipLaunchKernelGGL(ks_key_quotients_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 4096)), dim3(256), 0, s, kp, L, K, n, p->d_mods, kq);
            LAUNCH_CHECK();
            a.key_quo = kq; a.key_quo_jstride = 2ll * K * n;
        }
        if (f64) launch_ks_mac_f64(p->log_n, a, kp, batch * (size_t)(L + 1), launch_ctx(p, s));
        else launch_ks_mac_u64(p->log_n, a, kp, batch * (size_t)(L + 1), launch_ctx(p, s));
        LAUNCH_CHECK();
        }
    } else {
    // (2) digit decomposition fused into the forward NTT (replaces kernel_set_accumulate, fgk/switch_key.cu:6-54,
    //     + ntt_inplace_ps with key_switching_set_products, :907-908): row i = digits reduced mod q_key(i)
    {
        NttArgs a = contiguous_args(p, digits_src, ws + w.temp_ntt, L + 1, L, 0, K, TROYN_IDX_KS_SET_PRODUCTS, L);
        a.in_bstride = (long long)digits_bstride;
        a.in_pstride = 0;            // every row re-reads the same L digits
        a.reduce_input = 1;
        a.skip_diag = fused ? 1 : 0; // row i, digit i is the NTT-form input itself: not recomputed
        if ((rc = launch_ntt(p, a, batch, false, s))) return rc;
    }
    // (3) <digits, key> inner product (fgk/switch_key.cu:83-154)
    {
        const unsigned ch = chunks_pairs(n);
        const size_t rows = batch * (L + 1);
        if ((rc = check_rows(rows, ch))) return rc;
        hipLaunchKernelGGL(ks_accumulate_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, K, L, n, ws + w.temp_ntt, kp, ws + w.poly_prod,
                           fused ? target : (const u64*)nullptr, target_bstride);
        LAUNCH_CHECK();
    }
    }
    // (4) INTT: only the special-prime rows when the result stays in NTT form; all rows otherwise (:991-996)
    const u64* last_src;
    size_t last_stride;
    const u64* prod_for_util7;
    // (5)-(7) in ONE launch: the forward NTT reads the INTT'd special rows through the rounding-fix prologue (ski_util6_merged) and finishes
    // with the divide-by-special-prime / assign epilogue (ski_util7_merged)
    auto fused_tail_args = [&](const u64* src) {
        NttArgs a = contiguous_args(p, src, dest, 2, L, 0, L, TROYN_IDX_COMPONENTWISE, 0);
        a.in_bstride = 2ll * n; a.in_pstride = n; a.in_cstride = 0;
        a.load_mode = NTT_LOAD_KS_ROUND; a.aux_mod = K - 1;
        a.store_mode = NTT_STORE_KS_FINISH;
        a.flags = (is_ckks ? 1u : 0u) | ((unsigned)assign_method << 1);
        a.ext0 = ws + w.poly_prod; a.ext0_bstride = 2ll * (L + 1) * n; a.ext0_pstride = (long long)(L + 1) * n; a.ext0_cstride = n;
        a.ext1 = addend; a.ext1_bstride = (long long)addend_bstride; a.ext1_pstride = (long long)L * n; a.ext1_cstride = n;
        a.inv_table = p->d_inv_last + (size_t)K * K;
        return a;
    };
    if (fused && small_tail_wanted(p, batch * 2 * L)) {
        const bool tail_f64 = use_f64(p, 0, L) && use_f64(p, K - 1, 1);      // else both transforms on the integer kernels (small launches are not split by class)
        // a few ciphertexts at N = 16384: first inverse pass of the special rows in place (nothing else reads them), then the strided passes of both
        // transforms as one launch (small_tail)
        NttArgs pa = contiguous_args(p, ws + w.poly_prod + (size_t)L * n, ws + w.poly_prod + (size_t)L * n, 2, 1, K - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        pa.in_pstride = pa.out_pstride = (long long)(L + 1) * n;
        pa.in_bstride = pa.out_bstride = 2ll * (L + 1) * n;
        return small_tail(p, tail_f64, pa, batch * 2, fused_tail_args(ws + w.prod_intt), ws + w.temp_last, batch * 2, s);
    }
    if (is_ntt_form) {
        NttArgs a = contiguous_args(p, ws + w.poly_prod + (size_t)L * n, ws + w.prod_intt, 2, 1, K - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        a.in_pstride = (long long)(L + 1) * n;
        a.in_bstride = 2ll * (L + 1) * n;
        if ((rc = launch_ntt(p, a, batch, true, s))) return rc;
        last_src = ws + w.prod_intt; last_stride = n;
        prod_for_util7 = ws + w.poly_prod;
    } else if (!bgv && p->log_n >= 10 && p->log_n <= 17 && coeff_tail_fused(p)) {
        // coefficient form: INTT of the two special-prime rows, then the INTT of the 2L data rows finishes the key switch in its epilogue
        // ((5) + (7), ski_util6_merged / ski_util7_merged) and writes the destination -- the INTT'd rows never reach HBM
        NttArgs a = contiguous_args(p, ws + w.poly_prod + (size_t)L * n, ws + w.prod_intt, 2, 1, K - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        a.in_pstride = (long long)(L + 1) * n;
        a.in_bstride = 2ll * (L + 1) * n;
        if ((rc = launch_ntt(p, a, batch, true, s))) return rc;
        NttArgs f = contiguous_args(p, ws + w.poly_prod, dest, 2, L, 0, K, TROYN_IDX_COMPONENTWISE, 0);
        f.in_pstride = (long long)(L + 1) * n;
        f.in_bstride = 2ll * (L + 1) * n;
        f.load_mode = NTT_LOAD_KS_ROUND; f.aux_mod = K - 1;      // constants of the rounding fix (no forward prologue runs)
        f.store_mode = NTT_STORE_KS_FINISH;
        f.flags = (is_ckks ? 1u : 0u) | ((unsigned)assign_method << 1);
        f.in2 = ws + w.prod_intt; f.in2_bstride = 2ll * n; f.in2_pstride = n;
        f.ext0 = dest; f.ext0_bstride = 2ll * L * n; f.ext0_pstride = (long long)L * n; f.ext0_cstride = n;   // unused by this epilogue
        f.ext1 = addend; f.ext1_bstride = (long long)addend_bstride; f.ext1_pstride = (long long)L * n; f.ext1_cstride = n;
        f.inv_table = p->d_inv_last + (size_t)K * K;
        // two-pass sizes: the pass in between goes to the free part of prod_intt, never to dest (AddInplace reads the old destination)
        return launch_ntt(p, f, batch, true, s, ws + w.prod_intt + batch * 2 * (size_t)n);
    } else {
        NttArgs a = contiguous_args(p, ws + w.poly_prod, ws + w.prod_intt, 2, L + 1, 0, K, TROYN_IDX_KS_SKIP_FINALS, L);
        if ((rc = launch_ntt(p, a, batch, true, s))) return rc;
        last_src = ws + w.prod_intt + (size_t)L * n; last_stride = (size_t)(L + 1) * n;
        prod_for_util7 = ws + w.prod_intt;
    }
    if (fused) {
        // N >= 32768 transforms in two passes: the pass in between goes to temp_last (unused on this path), never to dest
        return launch_ntt(p, fused_tail_args(last_src), batch, false, s, ws + w.temp_last);
    }
    // (5) rounding fix of the special-prime component, per data limb (:570-598).  In NTT form the result goes to
    //     the unused tail of the prod_intt region so that step (6) can transform out of place.
    u64* util6_out = is_ntt_form ? ws + w.prod_intt + batch * 2 * (size_t)n : ws + w.temp_last;
    if (bgv) {
        // kernel_ski_util5_merged_step1 (:436-476): delta_j = (k mod q_j) * q_special + c mod q_j, k = -c * q_special^-1 mod t
        const unsigned ch = chunks_single(n);
        const size_t rows = batch * 2 * L;
        if ((rc = check_rows(rows, ch))) return rc;
        hipLaunchKernelGGL(bgv_delta_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, L, n, bgv->t, bgv->inv_special_mod_t, p->moduli[K - 1], last_src, last_stride, util6_out);
        LAUNCH_CHECK();
    } else if (is_ntt_form) {
        const unsigned ch = chunks_pairs(n);
        const size_t rows = batch * 2 * L;
        if ((rc = check_rows(rows, ch))) return rc;
        hipLaunchKernelGGL(ks_util6_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, K, L, n, last_src, last_stride, util6_out);
        LAUNCH_CHECK();
    }   // coefficient form: ks_util7_kernel forms the fix itself from the special-prime row
    // (6) back to NTT form when needed (:1033-1036)
    if (is_ntt_form) {
        NttArgs a = contiguous_args(p, util6_out, ws + w.temp_last, 2, L, 0, L, TROYN_IDX_COMPONENTWISE, 0);
        if ((rc = launch_ntt(p, a, batch, false, s))) return rc;
    }
    // (7) divide by the special prime and assign (:625-658; BGV: kernel_ski_util5_merged_step2 :506-538)
    if (bgv) {
        const unsigned ch = chunks_single(n);
        const size_t rows = batch * 2 * L;
        hipLaunchKernelGGL(bgv_finish_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, L, L + 1, n, prod_for_util7, ws + w.temp_last, p->d_inv_last + (size_t)K * K, assign_method, dest, addend, addend_bstride);
        LAUNCH_CHECK();
    } else {
        const unsigned ch = chunks_pairs(n);
        const size_t rows = batch * 2 * L;
        if ((rc = check_rows(rows, ch))) return rc;
        hipLaunchKernelGGL(ks_util7_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, L, L + 1, n, prod_for_util7, ws + w.temp_last,
                           p->d_inv_last + (size_t)K * K, is_ckks, assign_method, dest, addend, addend_bstride,
                           is_ntt_form ? (const u64*)nullptr : last_src, last_stride, K);
        LAUNCH_CHECK();
    }
    return TROYN_OK;
}

extern "C" int troyn_switch_key(const troyn_plan* plan, uint32_t L, int is_ckks, int is_ntt_form,
                                const uint64_t* target, const uint64_t* const* keys, int assign_method,
                                uint64_t* destination, void* workspace, size_t workspace_bytes,
                                size_t batch, troyn_stream_t stream) {
    select_device(plan);
    if (!plan) return fail(TROYN_E_INVALID, "[troyn_switch_key] null plan");
    return switch_key_impl(plan, L, is_ckks, is_ntt_form, (const u64*)target, (size_t)L * plan->n, keys, assign_method,
                           (u64*)destination, nullptr, 0, workspace, workspace_bytes, batch, (hipStream_t)stream);
}

extern "C" int troyn_relinearize(const troyn_plan* plan, uint32_t L, int is_ckks, int is_ntt_form,
                                 const uint64_t* ct3, const uint64_t* const* keys, uint64_t* out2,
                                 void* workspace, size_t workspace_bytes, size_t batch, troyn_stream_t stream) {
    select_device(plan);
    if (!plan || !ct3) return fail(TROYN_E_INVALID, "[Evaluator::relinearize_inplace_internal] null argument");
    const size_t pc = (size_t)L * plan->n;
    // relinearize_internal (evaluator_keyswitching.cu:119-144): switch_key(target = c2, Overwrite) then += (c0, c1)
    return switch_key_impl(plan, L, is_ckks, is_ntt_form, (const u64*)ct3 + 2 * pc, 3 * pc, keys, TROYN_ASSIGN_OVERWRITE,
                           (u64*)out2, (const u64*)ct3, 3 * pc, workspace, workspace_bytes, batch, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// modulus switching
// ---------------------------------------------------------------------------------------
extern "C" int troyn_divide_and_round_q_last(const troyn_plan* p, uint32_t L, const uint64_t* in, size_t pcount,
                                             uint64_t* out, size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !in || !out) return fail(TROYN_E_INVALID, "[RNSTool::divide_and_round_q_last] null argument");
    if (L < 2 || L > p->K) return fail(TROYN_E_INVALID, "[Evaluator::mod_switch_scale_to_next_internal] Next context data is not set.");
    const size_t items = batch * pcount, rows = items * (L - 1);
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_pairs(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(divide_round_q_last_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, L, p->n, (const u64*)in, p->d_inv_last + (size_t)L * p->K, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" size_t troyn_divide_and_round_q_last_ntt_workspace_bytes(const troyn_plan* p, uint32_t L, size_t pcount, size_t batch) {
    if (!p || L < 1) return 0;
    return batch * pcount * (size_t)(2 * L - 1) * p->n * sizeof(u64);   // last_intt [items][N] + 2 x temp [items][L-1][N]
}

extern "C" int troyn_divide_and_round_q_last_ntt(const troyn_plan* p, uint32_t L, const uint64_t* in, size_t pcount,
                                                 uint64_t* out, void* workspace, size_t workspace_bytes,
                                                 size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !in || !out || !workspace) return fail(TROYN_E_INVALID, "[RNSTool::divide_and_round_q_last_ntt] null argument");
    if (L < 2 || L > p->K) return fail(TROYN_E_INVALID, "[Evaluator::mod_switch_scale_to_next_internal] Next context data is not set.");
    if (workspace_bytes < troyn_divide_and_round_q_last_ntt_workspace_bytes(p, L, pcount, batch))
        return fail(TROYN_E_WORKSPACE, "[troyn_divide_and_round_q_last_ntt] workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t items = batch * pcount, n = p->n;
    if (items == 0) return TROYN_OK;
    u64* last_intt = (u64*)workspace;
    u64* temp0 = last_intt + items * n;               // step1 output (coefficient form)
    u64* temp = temp0 + items * (size_t)(L - 1) * n;  // its NTT (out of place: lets the NTT use half-limb workgroups)
    int rc;
    // INTT of the last limb only (the reference's device branch transforms all L limbs, utils/rns_tool.cu:675)
    NttArgs li = contiguous_args(p, (const u64*)in + (size_t)(L - 1) * n, last_intt, 1, 1, L - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
    li.in_bstride = (long long)L * n;
    // step1 -> NTT -> step2 in one launch (prologue / epilogue of the forward transform)
    NttArgs fw = contiguous_args(p, last_intt, (u64*)out, 1, L - 1, 0, L - 1, TROYN_IDX_COMPONENTWISE, 0);
    fw.in_bstride = (long long)n; fw.in_pstride = 0; fw.in_cstride = 0;
    fw.load_mode = NTT_LOAD_RESCALE; fw.aux_mod = L - 1;
    fw.store_mode = NTT_STORE_RESCALE;
    fw.ext0 = (const u64*)in; fw.ext0_bstride = (long long)L * n; fw.ext0_pstride = 0; fw.ext0_cstride = n;
    fw.inv_table = p->d_inv_last + (size_t)L * p->K;
    // a few ciphertexts at N = 16384: three launches instead of four (small_tail; the pass in between lives in `out`, as in the two-pass form)
    if (small_tail_wanted(p, items * (L - 1))) return small_tail(p, use_f64(p, 0, L), li, items, fw, (u64*)out, items, s);
    if ((rc = launch_ntt(p, li, items, true, s))) return rc;
    if (p->log_n >= 10) return launch_ntt(p, fw, items, false, s);
    const unsigned ch = chunks_pairs(p->n);
    const size_t rows = items * (L - 1);
    if ((rc = check_rows(rows, ch))) return rc;
    hipLaunchKernelGGL(rescale_step1_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s, ch, p->d_mods, L, p->n, last_intt, temp0);
    LAUNCH_CHECK();
    {
        NttArgs a = contiguous_args(p, temp0, temp, 1, L - 1, 0, L - 1, TROYN_IDX_COMPONENTWISE, 0);
        if ((rc = launch_ntt(p, a, items, false, s))) return rc;
    }
    hipLaunchKernelGGL(rescale_step2_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s, ch, p->d_mods, L, p->n,
                       (const u64*)in, temp, p->d_inv_last + (size_t)L * p->K, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// fused CKKS multiply -> relinearize -> rescale_to_next
// ---------------------------------------------------------------------------------------
struct MrrLayout { size_t digits, poly_prod, spec_intt, last_intt, keys_f64, keys_int, split, fast_total, prod3, relin2, sub, total; };

// the moduli the chain of level L touches (data limbs 0 .. L-1 and the special prime K-1) are all below 2^50 and the FP64 policy is not switched off
static bool mrr_all_f64(const troyn_plan* p, uint32_t L) { return use_f64(p, 0, L) && use_f64(p, p->K - 1, 1); }

static bool mrr_fast_path(const troyn_plan* p, uint32_t L) {
    // the chain's kernels exist for N = 8192 / 16384 / 32768 (d_fwd_r2); every modulus below 2^50: the FP64 kernels of rounds 2-4; chains with
    // moduli of 2^50 and more (round 5): every launch runs per modulus class, integer kernels for the wide limbs (ksmaci_kernel, the integer
    // forms of the fused transforms).  TROYN_MRR_MIXED=0 composes the three public calls for such chains as rounds 2-4 did (A/B runs, tests).
    if (!(p->d_fwd_r2 && L >= 2 && L + 1 <= p->K)) return false;
    if (mrr_all_f64(p, L)) return true;
    return p->d_fwd_r2i != nullptr && L + 1 <= 64 && !p->opt.mrr_mixed_off;
}

static MrrLayout mrr_layout(const troyn_plan* p, uint32_t L, size_t batch) {
    const size_t n = p->n;
    MrrLayout w;
    size_t off = 0;
    w.digits = off;    off += batch * L * n;
    w.poly_prod = off; off += batch * 2 * (size_t)(L + 1) * n;
    w.spec_intt = off; off += batch * 2 * n;
    w.last_intt = off; off += batch * 2 * n;
    w.keys_f64 = off;  off += (size_t)L * 2 * p->K * n + (size_t)L * 2 * n;     // prepared keys + the diagonal blocks in natural order
    // integer rows of a chain with wide moduli: (key, quotient) pairs [L][2][rows][N] + the diagonal blocks [rows][2][N]
    w.keys_int = off;  off += (L >= 1 && L + 1 <= p->K && !mrr_all_f64(p, L)) ? ((size_t)L * 2 * (L + 1) * n + (size_t)(L + 1) * 2 * n) * 2 : 0;
    w.split = off;     off += ks_split_words(p, batch, L, p->log_n);                        // slots of the digit-parallel inner product (small batches)
    w.fast_total = off;
    // composition of the three public calls (any other shape)
    off = 0;
    w.prod3 = off;  off += batch * 3 * (size_t)L * n;
    w.relin2 = off; off += batch * 2 * (size_t)L * n;
    w.sub = off;
    const size_t sub_bytes = std::max(troyn_relinearize_workspace_bytes(p, L, batch), troyn_divide_and_round_q_last_ntt_workspace_bytes(p, L, 2, batch));
    off += (sub_bytes + 7) / 8;
    w.total = std::max(off, w.fast_total);
    return w;
}

// internal streams of the chunked chain: one pair per (host thread, device), created on first use and kept for the life of the thread
constexpr int MRR_MAX_STREAMS = 4;
struct MrrStreams { hipStream_t s[MRR_MAX_STREAMS] = {}; hipEvent_t fork = nullptr, join[MRR_MAX_STREAMS] = {}; int device = -1; };
static MrrStreams* mrr_streams(int device) {
    static thread_local std::vector<MrrStreams> pool;
    for (auto& m : pool) if (m.device == device) return &m;
    MrrStreams m;
    m.device = device;
    bool ok = true;
    for (int q = 0; q < MRR_MAX_STREAMS && ok; q++) {
        ok = hipStreamCreateWithFlags(&m.s[q], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&m.join[q], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&m.fork, hipEventDisableTiming) == hipSuccess;
    if (!ok) {      // give back whatever was created before the failure
        for (int q = 0; q < MRR_MAX_STREAMS; q++) {
            if (m.join[q]) (void)hipEventDestroy(m.join[q]);
            if (m.s[q]) (void)hipStreamDestroy(m.s[q]);
        }
        if (m.fork) (void)hipEventDestroy(m.fork);
        return nullptr;
    }
    pool.push_back(m);      // kept for the life of the host thread (a handful of streams per thread and device; the runtime reclaims them at exit)
    return &pool.back();
}

// launches (1)-(5) of the fused chain for `batch` items whose intermediates live in `ws` (layout w); kf: the prepared keys
// raw != nullptr: digit-parallel inner product on the caller's own keys (small launches; the workspace has its slots and kf was not prepared)
// Chains with moduli of 2^50 and more: every launch is issued once per run of limbs of one arithmetic class (limbs are independent), the rows
// that cross classes are canonical u64 words (digits; T rows written by an integer kernel) and each consumer reduces what it reads.
// ksmac2's NLC instantiation (half tiles, N = 16384, all-FP64 fused chain): the words of a digit enter layer 0 as they are (u, v < max q_j), leave it
// as |x| <= max q_j + 0.875 p and run the three layers of round 0 without a re-centring in between.  Growth per layer: |x'| <= |x| + (0.5 + 1.5 |x| 2^-52) p
// (dev_math_f64.hpp); the chain qualifies when every value stays below 2^53 with a 2 % margin for EVERY row modulus p of the launch.
static bool ksmac_no_load_corr_ok(const troyn_plan* p, unsigned L) {
    if (p->log_n != 14 || p->opt.ntt_u64) return false;
    double qmax = 0.0;
    for (unsigned j = 0; j < L; j++) qmax = std::max(qmax, (double)p->moduli[j]);
    for (unsigned k = 0; k <= L; k++) {
        const double pm = (double)p->moduli[k == L ? p->K - 1 : k];
        double a = qmax + 0.875 * pm;
        for (int layer = 0; layer < 3; layer++) a += (0.5 + 1.5 * a * 0x1p-52) * pm;
        if (!(a < 0.98 * 0x1p53)) return false;
    }
    return true;
}

// ki: the integer rows' prepared keys (chains with wide moduli; nullptr otherwise)
static int mrr_chain(const troyn_plan* p, uint32_t L, const u64* a, const u64* b, const double* kf, const ulonglong2* ki, u64* out, u64* ws, const MrrLayout& w,
                     size_t batch, hipStream_t s, const KeyPtrs* raw = nullptr) {
    const unsigned K = p->K, n = p->n;
    int rc;
    const long long ct_b = 2ll * L * n, ct_p = (long long)L * n;           // strides of a, b
    const long long pp_b = 2ll * (L + 1) * n, pp_p = (long long)(L + 1) * n;   // strides of poly_prod
    auto mul_operands = [&](NttArgs& x, unsigned limb0) { x.mul_a = a; x.mul_b = b; x.mul_bstride = ct_b; x.mul_pstride = ct_p; x.mul_limb0 = limb0; };
    const bool all_f64 = mrr_all_f64(p, L);
    auto small = [&](unsigned mi) { return use_f64(p, mi, 1); };          // this modulus takes the FP64 kernels
    // runs of data limbs [j0, j1) of one class within [0, count)
    // (runs of different classes in flight together: RunOverlap above)
    auto for_runs = [&](unsigned count, unsigned polys, auto&& f) -> int {
        RunOverlap ov(p, s, !all_f64 && batch * (size_t)polys * count >= OVERLAP_MIN_LIMB_POLYS);
        for (unsigned j0 = 0, j1; j0 < count; j0 = j1) {
            for (j1 = j0 + 1; j1 < count && small(j1) == small(j0); j1++) {}
            if (int r = f(j0, j1, (j0 == 0 && j1 == count) ? s : ov.next())) return r;
        }
        return ov.join();
    };
    const bool special_wide = !small(K - 1), last_wide = !small(L - 1);
    bool wide_digits = false;
    unsigned long long small_rows = 0, wide_rows = 0;
    for (unsigned k = 0; k <= L; k++) {
        const unsigned mrow = (k == L) ? K - 1 : k;
        if (small(mrow)) small_rows |= 1ull << k; else wide_rows |= 1ull << k;
        if (k < L && !small(k)) wide_digits = true;
    }
    // (1) digits = INTT(c2), c2 = a1 (.) b1 formed in the loader (kernel_dyadic_convolute's third output + transform_from_ntt, :817-821)
    if ((rc = for_runs(L, 1, [&](unsigned j0, unsigned j1, hipStream_t rs) {
        NttArgs x = contiguous_args(p, a + (size_t)j0 * n, ws + w.digits + (size_t)j0 * n, 1, j1 - j0, j0, j1 - j0, TROYN_IDX_COMPONENTWISE, 0);
        x.in_bstride = ct_b; x.in_pstride = ct_p;                          // (the loader reads mul_a / mul_b; `in` only anchors the shapes)
        x.out_bstride = (long long)L * n; x.out_pstride = (long long)L * n;
        mul_operands(x, j0);
        x.fused_mode = NTT_FUSED_MULPAIR;
        if (all_f64) x.flags = NTT_FLAG_STORE_F64;            // the digits go to ksmac2 as doubles (one conversion here instead of L + 1 there)
        return launch_ntt(p, x, batch, true, rs);
    }))) return rc;
    // (2) key-switch inner product; the digit of row k under its own modulus is a1 (.) b1 again
    {
        TimerScope ts(TROYN_TIMER_KS_INNER_PRODUCT, s);
        if (small_rows) {
            KsMacArgs m;
            std::memset(&m, 0, sizeof(m));
            m.digits = ws + w.digits; m.dig_bstride = (long long)L * n; m.dig_cstride = n;
            m.diag = a + ct_p; m.diag_b = b + ct_p; m.diag_bstride = ct_b; m.diag_cstride = n;
            m.ten_a = a; m.ten_b = b; m.ten_bstride = ct_b; m.ten_pstride = ct_p;      // data rows leave as Q = P qk^-1 + c (keys prepared times qk^-1)
            m.diag_keys = kf + (size_t)L * 2 * K * n;
            m.out = ws + w.poly_prod; m.out_bstride = pp_b; m.out_pstride = pp_p; m.out_cstride = n;
            m.mods = p->d_mods; m.tw = p->d_fwd_f64; m.tw_r1 = p->d_fwd_r1; m.tw_r2 = p->d_fwd_r2;
            m.keys = kf; m.key_jstride = 2ll * K * n; m.key_pstride = (long long)K * n;
            m.L = L; m.table_start = 0; m.table_count = K; m.batch = (unsigned)batch; m.grouped = ksmac_order(p, batch, p->log_n);
            if (wide_rows) m.row_mask = small_rows;
            m.no_load_corr = (all_f64 && ksmac_no_load_corr_ok(p, L)) ? 1u : 0u;
            if (raw) {
                m.grouped = 0;
                m.part = reinterpret_cast<double*>(ws + w.split); m.part_jstride = (long long)batch * pp_b;
                m.split_skip_diag = 1;
                m.raw = *raw; m.raw_pstride = (long long)K * n; m.split_scale = p->d_inv_last + (size_t)K * K;
                launch_ksmac2_split(p->log_n, batch, m, s, true, 1);
            } else launch_ksmac2(p->log_n, batch, (unsigned)__builtin_popcountll(small_rows), m, s, all_f64, wide_digits);
            LAUNCH_CHECK();
        }
        if (wide_rows) {
            const unsigned slots = (unsigned)__builtin_popcountll(wide_rows);
            KsMacIArgs m;
            std::memset(&m, 0, sizeof(m));
            m.digits = ws + w.digits; m.dig_bstride = (long long)L * n; m.dig_cstride = n;
            m.ten_a = a; m.ten_b = b; m.ten_bstride = ct_b; m.ten_pstride = ct_p;
            m.out = ws + w.poly_prod; m.out_bstride = pp_b; m.out_pstride = pp_p; m.out_cstride = n;
            m.mods = p->d_mods; m.tw = p->d_fwd; m.tw_r1 = p->d_fwd_r1i; m.tw_r2 = p->d_fwd_r2i;
            m.keys = ki; m.key_jstride = 2ll * slots * n; m.key_pstride = (long long)slots * n;
            m.diag_keys = ki + (size_t)L * 2 * slots * n;
            m.L = L; m.table_start = 0; m.table_count = K; m.batch = (unsigned)batch;
            m.grouped = (batch % 8 == 0 && p->opt.ks_order != 0) ? 1u : 0u;
            m.row_mask = wide_rows;
            launch_ksmaci(p->log_n, batch, m, s, 2);
            LAUNCH_CHECK();
        }
    }
    const unsigned t_flags = (special_wide ? NTT_FLAG_TS_U64 : 0u) | (last_wide ? NTT_FLAG_TL_U64 : 0u);
    // arguments of steps (4) and (5) below (also read by the single-object form of the tail)
    auto last_args = [&]() {
        NttArgs x = contiguous_args(p, ws + w.poly_prod + (size_t)(L - 1) * n, ws + w.last_intt, 2, 1, L - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        x.in_pstride = pp_p; x.in_bstride = pp_b;
        x.in2 = ws + w.spec_intt; x.in2_bstride = 2ll * n; x.in2_pstride = n;
        x.aux_mod = K - 1; x.inv_table = p->d_inv_last + (size_t)K * K + (L - 1);
        x.fused_mode = (!last_wide && special_wide) ? NTT_FUSED_LAST_LIMB_W : NTT_FUSED_LAST_LIMB;
        x.flags = t_flags;
        return x;
    };
    auto tail_args = [&](unsigned j0, unsigned j1) {
        NttArgs x = contiguous_args(p, ws + w.spec_intt, out + (size_t)j0 * n, 2, j1 - j0, j0, j1 - j0, TROYN_IDX_COMPONENTWISE, 0);
        x.in_bstride = 2ll * n; x.in_pstride = n; x.in_cstride = 0;
        x.out_bstride = 2ll * (L - 1) * n; x.out_pstride = (long long)(L - 1) * n;
        x.in2 = ws + w.last_intt; x.in2_bstride = 2ll * n; x.in2_pstride = n;
        x.aux_mod = K - 1; x.inv_table = p->d_inv_last + (size_t)K * K + j0;
        x.aux2_mod = L - 1; x.inv_table2 = p->d_inv_last + (size_t)L * K + j0;
        x.ext0 = ws + w.poly_prod + (size_t)j0 * n; x.ext0_bstride = pp_b; x.ext0_pstride = pp_p; x.ext0_cstride = n;
        x.fused_mode = (small(j0) && t_flags) ? NTT_FUSED_TAIL_RESCALE_W : NTT_FUSED_TAIL_RESCALE;     // (launch_ntt co-locates the limbs that share the two input rows on one XCD)
        x.flags = t_flags;
        return x;
    };
    // Single objects at N = 16384 (every launch of the tail in its two-pass form, FP64 policy): the three strided passes between the first
    // inverse pass of {limb L-1, special rows} and the last forward pass of the output limbs run as ONE launch with T_s and T_l in registers
    // (troyn_mrr_small.hip): 3 launches instead of 6.  TROYN_MRR_SMALL=0 keeps the six.
    // N = 32768 (two-pass transforms at every size): the merged form at EVERY batch -- T_s and T_l never reach memory, three strided passes
    // become one (one thread per octet loops over the output limbs when the launch is not small): fused chain 6 x 50-bit, 256 items 117.5 k -> 126 k ops/s
    if (all_f64 && (small_tail_wanted(p, batch * 2 * (size_t)(L - 1)) || (p->log_n == 15 && !p->opt.mrr_small_off))) {
        const LaunchCtx lc = launch_ctx(p, s);
        auto prep = [&](NttArgs& x, bool inverse) {
            x.mods = p->d_mods; x.stream_loads = 1u; x.xcd_groups = 0u;
            x.tw = inverse ? (const void*)p->d_inv_f64 : (const void*)p->d_fwd_f64;
        };
        // first inverse pass of rows L-1 and L (special) of both polynomials, in place in poly_prod (nothing else reads those rows)
        NttArgs pa = contiguous_args(p, ws + w.poly_prod + (size_t)(L - 1) * n, ws + w.poly_prod + (size_t)(L - 1) * n, 2, 2, L - 1, K - (L - 1), TROYN_IDX_KS_SKIP_FINALS, 1);
        pa.in_pstride = pa.out_pstride = pp_p; pa.in_bstride = pa.out_bstride = pp_b;
        prep(pa, true);
        launch_ntt_f64_small_pass(p->log_n, 0, pa, batch * 4, lc);
        LAUNCH_CHECK();
        NttArgs sp = contiguous_args(p, ws + w.poly_prod + (size_t)L * n, nullptr, 2, 1, K - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        sp.in_pstride = pp_p; sp.in_bstride = pp_b;
        prep(sp, true);
        NttArgs la = last_args();
        prep(la, true);
        NttArgs ta = tail_args(0, L - 1);
        prep(ta, false);
        launch_mrr_quartet(p->log_n, batch, sp, la, ta, s, !p->opt.mrr_small_serial && batch * 2 * (size_t)(L - 1) * TROYN_SMALL_LP_FACTOR <= device_cu_count());
        LAUNCH_CHECK();
        // last forward pass of the output limbs, in place in `out`, with step (5)'s epilogue
        ta.in = ta.out; ta.in_bstride = ta.out_bstride; ta.in_pstride = ta.out_pstride; ta.in_cstride = ta.out_cstride;
        ta.reduce_input = 0;
        launch_ntt_f64_small_pass(p->log_n, 1, ta, batch * 2 * (L - 1), lc);
        LAUNCH_CHECK();
        return TROYN_OK;
    }
    // (3) s = INTT of the special-prime rows (:991-996, only the two rows the NTT-form tail needs)
    {
        NttArgs x = contiguous_args(p, ws + w.poly_prod + (size_t)L * n, ws + w.spec_intt, 2, 1, K - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        x.in_pstride = pp_p; x.in_bstride = pp_b;
        x.flags = NTT_FLAG_STORE_ROUND_HALF;     // stored as (s + qk/2) mod qk, the limb-independent part of the key switch's rounding fix
        if ((rc = launch_ntt(p, x, batch, true, s))) return rc;
    }
    // (4) l = INTT(relin_{L-1}) = INTT(Q_{L-1}) - r(s) qk^-1 with Q = P qk^-1 + c as ksmac2 left it   (divide_and_round_q_last_ntt's INTT of the last limb, :675)
    if ((rc = launch_ntt(p, last_args(), batch, true, s))) return rc;
    // (5) out_j = (Q_j - NTT_j(r_j(s) qk^-1 + f_j(l))) ql^-1, Q_j = P_j qk^-1 + c_j, for the L-1 remaining limbs: ski_util6/7 (:570-658), the
    //     trailing add of relinearize (:143) and both steps of divide_and_round_q_last_ntt (utils/rns_tool.cu:523-627) around ONE transform
    return for_runs(L - 1, 2, [&](unsigned j0, unsigned j1, hipStream_t rs) { return launch_ntt(p, tail_args(j0, j1), batch, false, rs); });
}

extern "C" size_t troyn_ckks_multiply_relinearize_rescale_workspace_bytes(const troyn_plan* plan, uint32_t L, size_t batch) {
    if (!plan || L < 2 || L > plan->K) return 0;
    return mrr_layout(plan, L, batch).total * sizeof(u64);
}

extern "C" int troyn_ckks_multiply_relinearize_rescale(const troyn_plan* p, uint32_t L, const uint64_t* a_, const uint64_t* b_,
                                                       const uint64_t* const* keys, uint64_t* out_, void* workspace, size_t workspace_bytes,
                                                       size_t batch, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[troyn_ckks_multiply_relinearize_rescale]";
    if (!p || !a_ || !b_ || !keys || !out_ || !workspace) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    const unsigned K = p->K, n = p->n;
    if (K < 2) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] Keyswitching is not supported.");
    if (L < 2 || L > K - 1) return fail(TROYN_E_INVALID, "[Evaluator::mod_switch_scale_to_next_internal] Next context data is not set.");
    const MrrLayout w = mrr_layout(p, L, batch);
    if (workspace_bytes < w.total * sizeof(u64)) return fail(TROYN_E_WORKSPACE, std::string(P) + " workspace too small");
    if (batch == 0) return TROYN_OK;
    hipStream_t s = (hipStream_t)stream;
    const u64* a = (const u64*)a_; const u64* b = (const u64*)b_;
    u64* out = (u64*)out_;
    u64* ws = (u64*)workspace;
    int rc;
    const bool unfused = p->opt.mrr_calls;    // TROYN_MRR=calls: composes the three public calls (A/B testing)
    // (a few ciphertexts of a chain with moduli >= 2^50: the three calls, whose key switch then takes the two-launch inner product -- ks_small_mixed)
    if (!mrr_fast_path(p, L) || unfused || ks_small_mixed(p, L, batch) || batch * (size_t)(L + 1) * 4 > 0x7fffffffull) {
        // Evaluator::multiply (evaluator.cu:118-145) -> relinearize (evaluator_keyswitching.cu:119-144) -> rescale_to_next
        if ((rc = launch_convolute(p->d_mods, n, 0, L, a, 2, b, 2, ws + w.prod3, batch, s))) return rc;
        const size_t sub_bytes = (w.total - w.sub) * sizeof(u64);
        if ((rc = troyn_relinearize(p, L, 1, 1, (const uint64_t*)(ws + w.prod3), keys, (uint64_t*)(ws + w.relin2), ws + w.sub, sub_bytes, batch, stream))) return rc;
        return troyn_divide_and_round_q_last_ntt(p, L, (const uint64_t*)(ws + w.relin2), 2, out_, ws + w.sub, sub_bytes, batch, stream);
    }
    KeyPtrs kp;
    std::memset(&kp, 0, sizeof(kp));
    for (unsigned j = 0; j < L; j++) {
        if (!keys[j]) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] null key pointer");
        kp.p[j] = (const u64*)keys[j];
    }
    const int chunk_env = p->opt.mrr_chunk;
    size_t chunk = batch;
    if (chunk_env > 0 && batch >= 2 * (size_t)chunk_env && (chunk_env % 8) == 0) chunk = (size_t)chunk_env;
    const bool all_f64 = mrr_all_f64(p, L);
    // small launches: the digit-parallel inner product reads the caller's keys as they are (chains of moduli below 2^50)
    const bool split = chunk == batch && all_f64 && ksmac_split_wanted(p, batch, L, p->log_n);
    // otherwise: keys prepared once per call (converted to exact doubles in the accumulators' layout), shared by every chunk
    double* kf = reinterpret_cast<double*>(ws + w.keys_f64);
    ulonglong2* ki = reinterpret_cast<ulonglong2*>(ws + w.keys_int);
    unsigned long long wide_rows = 0, small_rows = 0;
    for (unsigned k = 0; k <= L; k++) { if (use_f64(p, k == L ? K - 1 : k, 1)) small_rows |= 1ull << k; else wide_rows |= 1ull << k; }
    if (!split && small_rows) {
        const size_t pairs = (size_t)L * 2 * K * (n / 2);
        // the rows of the data moduli carry the factor qk^-1: the inner product leaves ksmac2 as P qk^-1 (+ the tensor term, KsMacArgs::ten_a)
        launch_ksmac_prepare_keys(kp, L, 2 * K, n, kf, (unsigned)std::min<size_t>((pairs + 255) / 256, 4096), s,
                                  p->d_inv_last + (size_t)K * K, p->d_mods, L, kf + (size_t)L * 2 * K * n);
        LAUNCH_CHECK();
    }
    if (wide_rows) {
        // the integer rows' (key qk^-1, Shoup quotient) pairs in the accumulators' layout + their diagonal blocks in natural order
        const unsigned slots = (unsigned)__builtin_popcountll(wide_rows);
        const size_t words = (size_t)L * 2 * slots * n;
        launch_ksmaci_prepare_keys(kp, L, K, n, wide_rows, ki, (unsigned)std::min<size_t>((words + 255) / 256, 4096), s,
                                   p->d_inv_last + (size_t)K * K, p->d_mods, L, ki + words);
        LAUNCH_CHECK();
    }
    // Chunked execution of the 5-launch chain on internal streams (round 3, an option: TROYN_MRR_CHUNK=<items, multiple of 8>,
    // TROYN_MRR_STREAMS=<1..4>, default 2): the batch is cut into chunks that alternate on internal streams, forked from and joined to
    // the caller's stream by events; chunks are independent (the batch is), results are unchanged.  The kernels of one chunk bound
    // differently (the inner product by FP64 issue, the transforms by memory), so two chunks in flight fill each other's idle resource:
    // +3 % when it was introduced.  Since then the inner product and the tail place the workgroups that share rows on one XCD (band
    // order; polynomials of a limb back to back) and a second chunk's kernels evict exactly those rows from the L2: one chunk on the
    // caller's stream is faster at every batch size measured (1024 items: 263.6 - 265.2 k vs 260.4 - 260.8 k ops/s with two halves,
    // 263.2 - 264.1 k with three thirds; 2048: 264.1 - 264.6 k vs 260.1 - 262.7 k; 512: 262.3 - 262.9 k vs 257.4 - 258.1 k) and is the default.
    const int ns = std::min(std::max(p->opt.mrr_streams, 1), MRR_MAX_STREAMS);
    if (chunk == batch) return mrr_chain(p, L, a, b, kf, ki, out, ws, w, batch, s, split ? &kp : nullptr);
    MrrStreams* ms = mrr_streams(p->device);
    if (!ms) return fail(TROYN_E_INVALID, std::string(P) + " cannot create the internal streams");
    const MrrLayout wc = mrr_layout(p, L, chunk);        // two chunk-sized workspaces side by side in the caller's workspace
    const size_t slot_words = wc.keys_f64;               // a chunk's intermediates end where its (unused) key area would start
    if ((size_t)ns * slot_words > w.keys_f64) {
        if (split) return fail(TROYN_E_INVALID, std::string(P) + " internal: chunk layout");      // unreachable: split implies one chunk
        return mrr_chain(p, L, a, b, kf, ki, out, ws, w, batch, s);
    }
    HIP_TRY(hipEventRecord(ms->fork, s));
    for (int q = 0; q < ns; q++) HIP_TRY(hipStreamWaitEvent(ms->s[q], ms->fork, 0));
    size_t done = 0, idx = 0;
    rc = TROYN_OK;
    while (done < batch) {
        const size_t c = std::min(chunk, batch - done);
        const int q = (int)(idx % (size_t)ns);
        if ((rc = mrr_chain(p, L, a + done * 2 * (size_t)L * n, b + done * 2 * (size_t)L * n, kf, ki, out + done * 2 * (size_t)(L - 1) * n,
                            ws + (size_t)q * slot_words, wc, c, ms->s[q]))) break;
        done += c; idx++;
    }
    // join ALSO on an error: the chunks already queued keep writing the caller's workspace / output, so the caller's stream must not
    // run ahead of them (the caller is free to release both as soon as this returns)
    bool joined = true;
    for (int q = 0; q < ns; q++)
        joined = hipEventRecord(ms->join[q], ms->s[q]) == hipSuccess && hipStreamWaitEvent(s, ms->join[q], 0) == hipSuccess && joined;
    if (!joined) { for (int q = 0; q < ns; q++) (void)hipStreamSynchronize(ms->s[q]); }
    return rc;
}

extern "C" int troyn_mod_switch_drop(const troyn_plan* p, uint32_t L_in, uint32_t L_out, const uint64_t* in, size_t pcount,
                                     uint64_t* out, size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !in || !out) return fail(TROYN_E_INVALID, "[Evaluator::mod_switch_drop_to_internal] null argument");
    if (L_out < 1 || L_out > L_in || L_in > p->K) return fail(TROYN_E_INVALID, "[Evaluator::mod_switch_drop_to_next_internal] Next context data is not set.");
    const size_t items = batch * pcount, rows = items * L_out;
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_pairs(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(copy_limbs_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->n, (const u64*)in, (size_t)L_in * p->n, 0u, (u64*)out, (size_t)L_out * p->n, L_out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// BGV: constants of one level's RNSTool that only BGV reads (utils/rns_tool.cu:205-232)
// ---------------------------------------------------------------------------------------
struct troyn_bgv {
    const troyn_plan* plan = nullptr;
    unsigned L = 0;
    u64 t = 0, inv_q_last_mod_t = 1, q_mod_t = 0;
    DevModulus t_mod;
    u64* d_consts = nullptr;            // [L] Shoup pairs (q/q_i)^-1 mod q_i, then [L] (q/q_i) mod t
    bool can_divide = false;            // q_last invertible mod t
};

extern "C" int troyn_bgv_destroy(troyn_bgv* b) {
    if (!b) return TROYN_OK;
    if (b->d_consts) (void)hipFree(b->d_consts);
    delete b;
    return TROYN_OK;
}

extern "C" int troyn_bgv_create(troyn_bgv** out, const troyn_plan* plan, uint32_t L, uint64_t t) {
    if (!out || !plan) return fail(TROYN_E_INVALID, "[troyn_bgv_create] null argument");
    *out = nullptr;
    if (L < 1 || L > plan->K) return fail(TROYN_E_INVALID, "[RNSTool::RNSTool] RNSBase length is invalid.");
    if (t < 2 || (t >> 61) != 0) return fail(TROYN_E_MODULUS, "[troyn_bgv_create] BGV needs a plain modulus in [2, 2^61).");
    std::unique_ptr<troyn_bgv, int (*)(troyn_bgv*)> b(new troyn_bgv, troyn_bgv_destroy);
    b->plan = plan; b->L = L; b->t = t;
    std::vector<u64> q(plan->moduli.begin(), plan->moduli.begin() + L);
    std::vector<u64> blob;
    for (size_t i = 0; i < L; i++) {
        u64 inv = 1;
        if (L > 1 && !host::invmod(host::product_mod(q, i, q[i]) % q[i], q[i], inv)) return fail(TROYN_E_MODULUS, "[RNSBase::initialize] RNSBase product is not invertible.");
        host::Shoup sh = host::shoup(inv % q[i], q[i]);
        blob.push_back(sh.operand); blob.push_back(sh.quotient);
    }
    for (size_t i = 0; i < L; i++) blob.push_back(host::product_mod(q, i, t));
    b->q_mod_t = host::product_mod(q, SIZE_MAX, t);
    u64 inv = 1;
    b->can_divide = host::invmod(q[L - 1] % t, t, inv);       // "[RNSTool::RNSTool] Unable to invert q[last] mod t."
    b->inv_q_last_mod_t = b->can_divide ? inv : 1;
    b->t_mod = make_dev_modulus(t, plan->log_n, false);
    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipMalloc(&b->d_consts, blob.size() * sizeof(u64)));
    HIP_TRY(hipMemcpy(b->d_consts, blob.data(), blob.size() * sizeof(u64), hipMemcpyHostToDevice));
    *out = b.release();
    return TROYN_OK;
}

extern "C" uint64_t troyn_bgv_inv_q_last_mod_t(const troyn_bgv* b) { return b ? b->inv_q_last_mod_t : 0; }

extern "C" size_t troyn_bgv_mod_switch_workspace_bytes(const troyn_bgv* b, size_t pcount, size_t batch) {
    return b ? troyn_divide_and_round_q_last_ntt_workspace_bytes(b->plan, b->L, pcount, batch) : 0;
}

extern "C" int troyn_bgv_mod_t_and_divide_q_last_ntt(const troyn_bgv* b, const uint64_t* in, size_t pcount, uint64_t* out, void* workspace, size_t workspace_bytes,
                                                     size_t batch, troyn_stream_t stream) {
    select_device(b);
    const char* P = "[RNSTool::mod_t_and_divide_q_last_ntt]";
    if (!b || !in || !out || !workspace) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    const troyn_plan* p = b->plan;
    const unsigned L = b->L;
    if (L < 2) return fail(TROYN_E_INVALID, "[Evaluator::mod_switch_scale_to_next_internal] Next context data is not set.");
    if (!b->can_divide) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert q[last] mod t.");
    if (workspace_bytes < troyn_bgv_mod_switch_workspace_bytes(b, pcount, batch)) return fail(TROYN_E_WORKSPACE, "[troyn_bgv_mod_t_and_divide_q_last_ntt] workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t items = batch * pcount, n = p->n;
    if (items == 0) return TROYN_OK;
    u64* last_intt = (u64*)workspace;
    u64* delta = last_intt + items * n;
    u64* delta_ntt = delta + items * (size_t)(L - 1) * n;
    int rc;
    {   // INTT of the last limb only (the reference's device branch transforms all of them, utils/rns_tool.cu:1757)
        NttArgs a = contiguous_args(p, (const u64*)in + (size_t)(L - 1) * n, last_intt, 1, 1, L - 1, 1, TROYN_IDX_COMPONENTWISE, 0);
        a.in_bstride = (long long)L * n;
        if ((rc = launch_ntt(p, a, items, true, s))) return rc;
    }
    const unsigned ch = chunks_single(p->n);
    const size_t rows = items * (L - 1);
    if ((rc = check_rows(rows, ch))) return rc;
    hipLaunchKernelGGL(bgv_delta_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                       ch, p->d_mods, L - 1, p->n, b->t_mod, b->inv_q_last_mod_t, p->moduli[L - 1], last_intt, n, delta);
    LAUNCH_CHECK();
    {
        NttArgs a = contiguous_args(p, delta, delta_ntt, 1, L - 1, 0, L - 1, TROYN_IDX_COMPONENTWISE, 0);
        if ((rc = launch_ntt(p, a, items, false, s))) return rc;
    }
    hipLaunchKernelGGL(bgv_finish_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                       ch, p->d_mods, L - 1, L, p->n, (const u64*)in, delta_ntt, p->d_inv_last + (size_t)L * p->K, -1, (u64*)out, (const u64*)nullptr, (size_t)0);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_bgv_decrypt_mod_t(const troyn_bgv* b, const uint64_t* phase, uint64_t correction_factor, uint64_t* dest, size_t batch, troyn_stream_t stream) {
    select_device(b);
    const char* P = "[scaling_variant::decentralize]";
    if (!b || !phase || !dest) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (batch == 0) return TROYN_OK;
    if (batch > 65535) return fail(TROYN_E_INVALID, std::string(P) + " batch too large for one launch");
    u64 fix = 1;
    if (correction_factor != 1 && !host::invmod(correction_factor % b->t, b->t, fix)) return fail(TROYN_E_INVALID, std::string(P) + " Correction factor is not invertible.");
    BgvDecryptArgs a;
    a.mods = b->plan->d_mods;
    a.inv_punctured = reinterpret_cast<const ulonglong2*>(b->d_consts);
    a.punctured_mod_t = b->d_consts + 2 * (size_t)b->L;
    a.t = b->t_mod; a.q_mod_t = b->q_mod_t; a.fix = fix; a.L = b->L; a.n = b->plan->n;
    hipLaunchKernelGGL(bgv_decrypt_mod_t_kernel, dim3((b->plan->n + 255) / 256, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, a, (const u64*)phase, (u64*)dest);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_bgv_multiply_scalar_mod_t(const troyn_bgv* b, const uint64_t* in, uint64_t scalar, uint64_t* out, size_t count, troyn_stream_t stream) {
    select_device(b);
    if (!b || !in || !out) return fail(TROYN_E_INVALID, "[utils::multiply_scalar] null argument");
    if (count == 0) return TROYN_OK;
    hipLaunchKernelGGL(scalar_mod_t_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b->t_mod, (u64)(scalar % b->t), (const u64*)in, (u64*)out, count);
    LAUNCH_CHECK();
    return TROYN_OK;
}

// Key switching for BGV: `key_level` = troyn_bgv_create(plan, n_moduli, t), whose last prime is the special prime
static int bgv_tail_from(const troyn_bgv* key_level, BgvTail& tail) {
    if (!key_level) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] null BGV constants");
    if (key_level->L != key_level->plan->K) return fail(TROYN_E_INVALID, "[Evaluator::switch_key_inplace_internal] BGV key switching needs the key level's constants");
    if (!key_level->can_divide) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert q[last] mod t.");
    tail.t = key_level->t_mod; tail.inv_special_mod_t = key_level->inv_q_last_mod_t;
    return TROYN_OK;
}

extern "C" int troyn_bgv_switch_key(const troyn_bgv* key_level, uint32_t L, const uint64_t* target, const uint64_t* const* keys, int assign_method,
                                    uint64_t* destination, void* workspace, size_t workspace_bytes, size_t batch, troyn_stream_t stream) {
    select_device(key_level);
    BgvTail tail;
    if (int rc = bgv_tail_from(key_level, tail)) return rc;
    const troyn_plan* plan = key_level->plan;
    return switch_key_impl(plan, L, 0, 1, (const u64*)target, (size_t)L * plan->n, keys, assign_method, (u64*)destination, nullptr, 0, workspace, workspace_bytes, batch,
                           (hipStream_t)stream, &tail);
}

extern "C" int troyn_bgv_relinearize(const troyn_bgv* key_level, uint32_t L, const uint64_t* ct3, const uint64_t* const* keys, uint64_t* out2, void* workspace,
                                     size_t workspace_bytes, size_t batch, troyn_stream_t stream) {
    select_device(key_level);
    BgvTail tail;
    if (int rc = bgv_tail_from(key_level, tail)) return rc;
    if (!ct3) return fail(TROYN_E_INVALID, "[Evaluator::relinearize_inplace_internal] null argument");
    const troyn_plan* plan = key_level->plan;
    const size_t pc = (size_t)L * plan->n;
    return switch_key_impl(plan, L, 0, 1, (const u64*)ct3 + 2 * pc, 3 * pc, keys, TROYN_ASSIGN_OVERWRITE, (u64*)out2, (const u64*)ct3, 3 * pc, workspace, workspace_bytes,
                           batch, (hipStream_t)stream, &tail);
}

// ---------------------------------------------------------------------------------------
// BEHZ
// ---------------------------------------------------------------------------------------
struct troyn_behz {
    const troyn_plan* plan = nullptr;   // base q (first L moduli)
    troyn_plan* aux = nullptr;          // Bsk primes (NTT tables for base Bsk, rns_tool.cu:97-101)
    unsigned L = 0, Bn = 0, Bsk = 0;
    u64 t = 0;
    std::vector<u64> bsk_values;
    u64* d_consts = nullptr;
    BehzDev dev;
    Behz2Dev dev2;                      // second-generation conversion kernels (behz2_kernels.hpp)
    bool have2 = false, smallq = false; // have2: |B| == L <= 16 and every q_i < 2^60; smallq: every q_i < 2^50
    bool aux50 = false;                 // the multiply works in an auxiliary base of primes below 2^50 (troyn_behz_create); bsk_values stays the reference's
    // encrypt / decrypt side constants of the same RNSTool / ContextData (context_data.cu:226-247, rns_tool.cu:168-211)
    u64 gamma = 0, q_mod_t = 0;
    DevModulus t_mod, gamma_mod;
    const ulonglong2* d_delta = nullptr;               // [L] floor(q/t) mod q_j
    const ulonglong2* d_prod_t_gamma_mod_q = nullptr;  // [L]
    const u64* d_q_to_t = nullptr;                     // [L]
    const u64* d_q_to_gamma = nullptr;                 // [L]
    ulonglong2 neg_inv_q_mod_t, neg_inv_q_mod_gamma, inv_gamma_mod_t;
    bool decrypt_ready = false;
    // options of the auxiliary plan = the plan's (both bases of one multiply on the same transform policies); followed at call time (below)
    mutable std::atomic<uint64_t> aux_gen{0};
    mutable std::mutex aux_mutex;
};
// A troyn_plan_set_option on the plan after troyn_behz_create used to leave base q and base Bsk on different policies inside one multiply:
// the auxiliary plan's options are brought up to date at the first call after the change (double-checked; set_option itself is a quiescent-point call).
static inline void behz_follow_options(const troyn_behz* b) {
    const uint64_t g = b->plan->opt_gen.load(std::memory_order_acquire);
    if (b->aux_gen.load(std::memory_order_acquire) == g || !b->aux) return;
    std::lock_guard<std::mutex> lock(b->aux_mutex);
    if (b->aux_gen.load(std::memory_order_relaxed) == g) return;
    b->aux->opt = b->plan->opt;
    b->aux_gen.store(g, std::memory_order_release);
}

extern "C" int troyn_behz_destroy(troyn_behz* b) {
    if (!b) return TROYN_OK;
    if (b->aux) plan_free(b->aux);
    if (b->d_consts) (void)hipFree(b->d_consts);
    delete b;
    return TROYN_OK;
}

extern "C" int troyn_behz_create(troyn_behz** out, const troyn_plan* plan, uint32_t L, uint64_t t) {
    if (!out || !plan) return fail(TROYN_E_INVALID, "[troyn_behz_create] null argument");
    *out = nullptr;
    if (L < 1 || L > plan->K) return fail(TROYN_E_INVALID, "[RNSTool::RNSTool] RNSBase length is invalid.");
    if (t == 0 || (t >> 61) != 0 || t == 1) return fail(TROYN_E_MODULUS, "[troyn_behz_create] BFV needs a plain modulus in [2, 2^61).");
    std::unique_ptr<troyn_behz, int (*)(troyn_behz*)> b(new troyn_behz, troyn_behz_destroy);
    b->plan = plan; b->L = L; b->t = t;
    const unsigned n = plan->n;
    std::vector<u64> q(plan->moduli.begin(), plan->moduli.begin() + L);
    // auxiliary base sizes and primes: utils/rns_tool.cu:52-80
    size_t total_bits = host::product_bit_count(q);
    size_t Bn = L;
    if (32 + host::bit_count(t) + total_bits >= 61 * (size_t)L + 61) Bn++;
    const size_t Bsk_ref = Bn + 1;
    std::vector<u64> primes;
    try {
        primes = host::get_primes(2 * (u64)n, 61, Bsk_ref + 1);
    } catch (const std::exception& e) {
        return fail(TROYN_E_MODULUS, e.what());
    }
    u64 m_sk = primes[0];   // primes[1] = gamma (decrypt side only)
    std::vector<u64> B(primes.begin() + 2, primes.begin() + 2 + Bn);
    std::vector<u64> bsk = B; bsk.push_back(m_sk);
    const u64 mt = (u64)1 << 32;
    b->bsk_values = bsk;          // the reference's base (troyn_behz_get_base_Bsk: known-answer hook), whatever base the multiply works in
    // ---- working base.  BEHZ's result does not depend on the auxiliary primes: the conversion q -> Bsk yields the integer x + alpha q
    // (alpha < L, a function of the residues mod q only), the small Montgomery reduction works modulo m_tilde = 2^32, the floor is an exact
    // integer division in base Bsk and the Shenoy-Kumaresan conversion back to q is exact as long as the value fits B -- all statements about
    // integers, true for ANY base whose product is at least the reference's (utils/rns_tool.cu:52-80 sizes B as |q| or |q| + 1 primes of 61 bits).
    // When every q_i is below 2^50 the 61-bit base is the only reason half of the multiply's transforms, its tensor product and its two
    // conversions run on the integer butterflies (27.5 issue slots against 8 FP64 instructions): take instead NB primes BELOW 2^50 with
    //     prod(B') >= 2^(61 Bn)   and   prod(B') m_sk' >= 2^(61 (Bn + 1)),
    // i.e. at least the capacity of any base the reference could have picked, so that every transform of the multiply takes the exact-FP64
    // policy.  The oracle keeps the reference's base; equality of the final residues on every BEHZ test is the proof.
    // MEASURED (round 4, BASELINE config 4, profiles/r04_cfg4_ab.txt): bit-identical results, and no gain -- the tensor kernels drop from 1.33 to
    // 1.01 ms per 64 products, but 13 + 1 primes instead of 10 + 1 make both conversions 29 % longer (0.66 -> 0.83 ms for the floor alone) and
    // add three limbs to every strided pass: 14.0 k against 14.2 k mul+relin ops/s.  ROUND 5: 13 + 1 came from asking for the capacity of the
    // reference's base itself; the reference's own size criterion (below) is met by 11 + 1 primes -- one row more than the 61-bit base instead of
    // three: tensor launches 1.33 -> 0.95 ms per 64 products, 14.15 k -> 14.7 k ops/s same box.  It is the DEFAULT whenever it applies (every q_i
    // below 2^50, second-generation conversions); TROYN_BEHZ_BASE=ref keeps the reference's base (read here).
    bool aux50 = !plan->opt.behz_base_ref;
    for (u64 v : q) if (v >= F64_MODULUS_LIMIT) aux50 = false;
    if (plan->opt.behz_v1 || L > BEHZ2_MAX_L || plan->log_n < 10) aux50 = false;
    if (aux50) {
        try {
            std::vector<u64> cand = host::get_primes(2 * (u64)n, 50, plan->K + 2 * Bsk_ref + 8);
            std::vector<u64> fresh;
            for (u64 c : cand) if (std::find(plan->moduli.begin(), plan->moduli.end(), c) == plan->moduli.end()) fresh.push_back(c);
            std::vector<u64> B2;
            size_t next = 1;                                  // fresh[0] becomes m_sk'
            // Size of the base: the reference's own criterion (utils/rns_tool.cu:50-62: K n t q^2 < q prod(B) m_sk with 32 bits reserved for K n, i.e.
            // bits(prod(B) m_sk) > 32 + bits(t) + bits(q); it states it for 61-bit primes as 61 (#B + 1)) evaluated on the ACTUAL product of the primes
            // taken, with two more bits of margin.  (Round 4 asked for the capacity of the reference's base itself, 2^(61 (Bn + 1)): 13 + 1 primes at
            // BASELINE config 4 where 11 + 1 satisfy the criterion -- every conversion and strided pass scales with the number of rows.)
            const size_t need_bits = 32 + (size_t)host::product_bit_count(std::vector<u64>{t}) + (size_t)host::product_bit_count(q) + 2;
            auto enough = [&](const std::vector<u64>& base) {
                std::vector<u64> with_sk = base; with_sk.push_back(fresh[0]);
                return (size_t)host::product_bit_count(with_sk) > need_bits;
            };
            while (next < fresh.size()) {
                B2.push_back(fresh[next++]);
                if (enough(B2)) break;
            }
            if (!fresh.empty() && !B2.empty() && enough(B2) && B2.size() <= 32) {
                B = B2; m_sk = fresh[0]; Bn = B.size();
                bsk = B; bsk.push_back(m_sk);
            } else aux50 = false;
        } catch (const std::exception&) { aux50 = false; }
    }
    // Shenoy-Kumaresan (rns_tool.cu:1000-1036): the correction term alpha_sk of the conversion B -> q is recovered modulo m_sk and satisfies
    // |alpha_sk| <= |B| (+ the lambda of the fast floor); it is read off a centred residue, so m_sk must exceed twice that -- any prime of 50 or
    // 61 bits does by ~45 bits, but the working base is chosen above, so state what it relies on
    if (m_sk < 2 * ((u64)Bn + 2) + 1) return fail(TROYN_E_MODULUS, "[troyn_behz_create] m_sk is too small for the Shenoy-Kumaresan correction of this base");
    const size_t Bsk = Bn + 1;          // working base from here on
    b->Bn = (unsigned)Bn; b->Bsk = (unsigned)Bsk; b->aux50 = aux50;

    int rc = troyn_plan_create(&b->aux, plan->device, plan->log_n, (uint32_t)Bsk, reinterpret_cast<const uint64_t*>(bsk.data()), nullptr);
    if (rc != TROYN_OK) return rc;
    b->aux->opt = plan->opt;            // the auxiliary base takes the options its plan has now and follows later changes (behz_follow_options)
    b->aux_gen.store(plan->opt_gen.load(std::memory_order_acquire), std::memory_order_release);

    // constant block: all tables in one allocation
    std::vector<u64> blob;
    auto push_shoup = [&](u64 w, u64 m) { host::Shoup s = host::shoup(w % m, m); blob.push_back(s.operand); blob.push_back(s.quotient); };
    auto need_inv = [&](u64 a, u64 m, u64& o) { return host::invmod(a % m, m, o); };
    size_t off_q_inv_punc = blob.size();
    for (size_t i = 0; i < L; i++) {
        u64 inv = 1;
        if (L > 1 && !need_inv(host::product_mod(q, i, q[i]), q[i], inv)) return fail(TROYN_E_MODULUS, "[RNSBase::initialize] RNSBase product is not invertible.");
        push_shoup(inv, q[i]);
    }
    // the scalar factor that precedes a base conversion folded into its first step: (x * c mod q_i) * inv_punc_i mod q_i
    // = x * (c * inv_punc_i mod q_i) mod q_i -- one Shoup multiply instead of a Barrett-128 product followed by one (c = m_tilde for
    // the lift, evaluator.cu:52-56 + rns_tool.cu:1083-1094; c = t for the floor, evaluator.cu:95-100)
    size_t off_q_mt_inv_punc = blob.size();
    for (size_t i = 0; i < L; i++) {
        u64 inv = 1;
        if (L > 1) need_inv(host::product_mod(q, i, q[i]), q[i], inv);
        push_shoup(host::mulmod(mt % q[i], inv, q[i]), q[i]);
    }
    size_t off_q_t_inv_punc = blob.size();
    for (size_t i = 0; i < L; i++) {
        u64 inv = 1;
        if (L > 1) need_inv(host::product_mod(q, i, q[i]), q[i], inv);
        push_shoup(host::mulmod(t % q[i], inv, q[i]), q[i]);
    }
    size_t off_q_to_bsk = blob.size();
    for (size_t bi = 0; bi < Bsk; bi++) for (size_t i = 0; i < L; i++) blob.push_back(host::product_mod(q, i, bsk[bi]));
    size_t off_q_to_mt = blob.size();
    for (size_t i = 0; i < L; i++) blob.push_back(host::product_mod(q, i, mt));
    if (blob.size() & 1) blob.push_back(0);
    size_t off_prod_q_mod_bsk = blob.size();
    for (size_t bi = 0; bi < Bsk; bi++) push_shoup(host::product_mod(q, SIZE_MAX, bsk[bi]), bsk[bi]);
    size_t off_inv_mt_mod_bsk = blob.size();
    for (size_t bi = 0; bi < Bsk; bi++) {
        u64 inv;
        if (!need_inv(mt, bsk[bi], inv)) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert m_tilde.");
        push_shoup(inv, bsk[bi]);
    }
    size_t off_inv_prod_q_mod_bsk = blob.size();
    for (size_t bi = 0; bi < Bsk; bi++) {
        u64 inv;
        if (!need_inv(host::product_mod(q, SIZE_MAX, bsk[bi]), bsk[bi], inv)) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert base_q product.");
        push_shoup(inv, bsk[bi]);
    }
    // (x * t - f) * q^-1 = x * (t q^-1) - f * q^-1 mod p_b: the floor's multiplication by t folded into the division by q
    size_t off_t_inv_prod_q_mod_bsk = blob.size();
    for (size_t bi = 0; bi < Bsk; bi++) {
        u64 inv = 0;
        need_inv(host::product_mod(q, SIZE_MAX, bsk[bi]), bsk[bi], inv);
        push_shoup(host::mulmod(t % bsk[bi], inv, bsk[bi]), bsk[bi]);
    }
    size_t off_B_inv_punc = blob.size();
    for (size_t bi = 0; bi < Bn; bi++) {
        u64 inv = 1;
        if (Bn > 1 && !need_inv(host::product_mod(B, bi, B[bi]), B[bi], inv)) return fail(TROYN_E_MODULUS, "[RNSBase::initialize] RNSBase product is not invertible.");
        push_shoup(inv, B[bi]);
    }
    size_t off_B_to_q = blob.size();
    for (size_t i = 0; i < L; i++) for (size_t bi = 0; bi < Bn; bi++) blob.push_back(host::product_mod(B, bi, q[i]));
    size_t off_B_to_msk = blob.size();
    for (size_t bi = 0; bi < Bn; bi++) blob.push_back(host::product_mod(B, bi, m_sk));
    if (blob.size() & 1) blob.push_back(0);
    size_t off_prod_B_mod_q = blob.size();
    for (size_t i = 0; i < L; i++) push_shoup(host::product_mod(B, SIZE_MAX, q[i]), q[i]);
    size_t off_neg_prod_B_mod_q = blob.size();
    for (size_t i = 0; i < L; i++) { u64 v = host::product_mod(B, SIZE_MAX, q[i]); push_shoup(q[i] - v, q[i]); }

    // second-generation conversion tables (split matrices with the scalar factors folded in)
    Behz2Offsets o2;
    bool have2 = (Bn == L || aux50) && L <= BEHZ2_MAX_L, smallq = true;
    for (u64 v : q) { if (v >> 60) have2 = false; if (v >= F64_MODULUS_LIMIT) smallq = false; }
    if (have2) {
        if (blob.size() & 1) blob.push_back(0);
        have2 = behz2_build_tables(q, B, m_sk, t, smallq, blob, o2);
    }

    // ---- encrypt / decrypt side ----
    const u64 gamma = primes[1];
    b->gamma = gamma;
    std::vector<u64> q_over_t;                       // floor(q / t) as a multi-word integer
    {
        std::vector<u64> big = host::big_product(q);
        b->q_mod_t = host::big_divmod_small(big, t);  // big <- floor(q/t)
        q_over_t = big;
    }
    if (blob.size() & 1) blob.push_back(0);
    size_t off_delta = blob.size();
    for (size_t i = 0; i < L; i++) push_shoup(host::big_mod_small(q_over_t, q[i]), q[i]);
    size_t off_ptg = blob.size();
    for (size_t i = 0; i < L; i++) push_shoup(host::mulmod(t % q[i], gamma % q[i], q[i]), q[i]);
    size_t off_q_to_t = blob.size();
    for (size_t i = 0; i < L; i++) blob.push_back(host::product_mod(q, i, t));
    size_t off_q_to_gamma = blob.size();
    for (size_t i = 0; i < L; i++) blob.push_back(host::product_mod(q, i, gamma));
    {
        u64 inv;
        b->decrypt_ready = true;
        if (need_inv(gamma % t, t, inv)) { host::Shoup s = host::shoup(inv, t); b->inv_gamma_mod_t = make_ulonglong2(s.operand, s.quotient); }
        else b->decrypt_ready = false;                // "[RNSTool::RNSTool] Unable to invert gamma mod t."
        if (need_inv(host::product_mod(q, SIZE_MAX, t), t, inv)) { host::Shoup s = host::shoup((t - inv) % t, t); b->neg_inv_q_mod_t = make_ulonglong2(s.operand, s.quotient); }
        else b->decrypt_ready = false;
        if (need_inv(host::product_mod(q, SIZE_MAX, gamma), gamma, inv)) { host::Shoup s = host::shoup((gamma - inv) % gamma, gamma); b->neg_inv_q_mod_gamma = make_ulonglong2(s.operand, s.quotient); }
        else b->decrypt_ready = false;
    }
    b->t_mod = make_dev_modulus(t, plan->log_n, false);
    b->gamma_mod = make_dev_modulus(gamma, plan->log_n, false);

    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipMalloc(&b->d_consts, blob.size() * sizeof(u64)));
    HIP_TRY(hipMemcpy(b->d_consts, blob.data(), blob.size() * sizeof(u64), hipMemcpyHostToDevice));

    BehzDev& d = b->dev;
    std::memset(&d, 0, sizeof(d));
    d.L = L; d.Bn = (unsigned)Bn; d.Bsk = (unsigned)Bsk; d.n = n; d.t = t;
    d.q_mods = plan->d_mods;
    d.bsk_mods = b->aux->d_mods;
    d.m_tilde = make_dev_modulus(mt, plan->log_n, false);
    auto P2 = [&](size_t off) { return reinterpret_cast<const ulonglong2*>(b->d_consts + off); };
    d.q_inv_punc = P2(off_q_inv_punc);
    d.q_mt_inv_punc = P2(off_q_mt_inv_punc);
    d.q_t_inv_punc = P2(off_q_t_inv_punc);
    d.t_inv_prod_q_mod_bsk = P2(off_t_inv_prod_q_mod_bsk);
    d.q_to_bsk = b->d_consts + off_q_to_bsk;
    d.q_to_mt = b->d_consts + off_q_to_mt;
    {
        u64 inv;
        if (!need_inv(host::product_mod(q, SIZE_MAX, mt), mt, inv)) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert base_q product.");
        host::Shoup s = host::shoup((mt - inv) % mt, mt);
        d.neg_inv_prod_q_mod_mt = make_ulonglong2(s.operand, s.quotient);
        if (!need_inv(host::product_mod(B, SIZE_MAX, m_sk), m_sk, inv)) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert base_B product.");
        s = host::shoup(inv, m_sk);
        d.inv_prod_B_mod_msk = make_ulonglong2(s.operand, s.quotient);
    }
    d.prod_q_mod_bsk = P2(off_prod_q_mod_bsk);
    d.inv_mt_mod_bsk = P2(off_inv_mt_mod_bsk);
    d.inv_prod_q_mod_bsk = P2(off_inv_prod_q_mod_bsk);
    d.B_inv_punc = P2(off_B_inv_punc);
    d.B_to_q = b->d_consts + off_B_to_q;
    d.B_to_msk = b->d_consts + off_B_to_msk;
    d.prod_B_mod_q = P2(off_prod_B_mod_q);
    d.neg_prod_B_mod_q = P2(off_neg_prod_B_mod_q);
    b->have2 = have2; b->smallq = smallq;
    std::memset(&b->dev2, 0, sizeof(b->dev2));
    if (have2) {
        Behz2Dev& e = b->dev2;
        e.L = L; e.n = n; e.rs = o2.rs; e.NB = (unsigned)Bn;
        e.q_mods = plan->d_mods; e.q_mt_inv_punc = d.q_mt_inv_punc; e.q_t_inv_punc = d.q_t_inv_punc;
        e.lift_mt = reinterpret_cast<const u32*>(b->d_consts + o2.lift_mt);
        e.lift_rows = reinterpret_cast<const u32*>(b->d_consts + o2.lift_rows);
        e.lift_rc = b->d_consts + o2.lift_rc;
        e.fa_rows = reinterpret_cast<const u32*>(b->d_consts + o2.fa_rows);
        e.fa_rc = b->d_consts + o2.fa_rc;
        e.fb_cols = reinterpret_cast<const u32*>(b->d_consts + o2.fb_cols);
        e.fb_rc = b->d_consts + o2.fb_rc;
    }
    b->d_delta = P2(off_delta);
    b->d_prod_t_gamma_mod_q = P2(off_ptg);
    b->d_q_to_t = b->d_consts + off_q_to_t;
    b->d_q_to_gamma = b->d_consts + off_q_to_gamma;
    *out = b.release();
    return TROYN_OK;
}

extern "C" uint32_t troyn_behz_base_Bsk_size(const troyn_behz* b) { return b ? (uint32_t)b->bsk_values.size() : 0; }
extern "C" uint32_t troyn_behz_working_base_size(const troyn_behz* b) { return b ? b->Bsk : 0; }
extern "C" int troyn_behz_get_base_Bsk(const troyn_behz* b, uint64_t* out) {
    select_device(b);
    if (!b || !out) return fail(TROYN_E_INVALID, "[troyn_behz_get_base_Bsk] null argument");
    for (size_t i = 0; i < b->bsk_values.size(); i++) out[i] = b->bsk_values[i];
    return TROYN_OK;
}

struct BehzLayout { size_t a_q, a_bsk, b_q, b_bsk, d_q, d_bsk, total; };
static BehzLayout behz_layout(const troyn_behz* b, size_t pa, size_t pb, size_t batch) {
    const size_t n = b->plan->n, L = b->L, S = b->Bsk, po = pa + pb - 1;
    BehzLayout w; size_t off = 0;
    w.a_q = off;   off += batch * pa * L * n;
    w.a_bsk = off; off += batch * pa * S * n;
    w.b_q = off;   off += batch * pb * L * n;
    w.b_bsk = off; off += batch * pb * S * n;
    w.d_q = off;   off += batch * po * L * n;
    w.d_bsk = off; off += batch * po * S * n;
    w.total = off;
    return w;
}

extern "C" size_t troyn_bfv_multiply_workspace_bytes(const troyn_behz* b, size_t pa, size_t pb, size_t batch) {
    if (!b || pa < 1 || pb < 1) return 0;
    return behz_layout(b, pa, pb, batch).total * sizeof(u64);
}

template <typename F4, typename F8, typename F16, typename F64>
static void dispatch_bound(unsigned v, F4 f4, F8 f8, F16 f16, F64 f64) {
    if (v <= 4) f4(); else if (v <= 8) f8(); else if (v <= 16) f16(); else f64();
}

static bool behz2_enabled(const troyn_behz* b) {
    return b->have2 && !b->plan->opt.behz_v1;   // TROYN_BEHZ=v1: first-generation kernels (A/B runs and the tests of that path)
}

extern "C" int troyn_bfv_multiply(const troyn_behz* b, const uint64_t* a_, size_t pa, const uint64_t* b_, size_t pb,
                                  uint64_t* out, void* workspace, size_t workspace_bytes, size_t batch, troyn_stream_t stream) {
    select_device(b);
    if (!b || !a_ || !b_ || !out || !workspace) return fail(TROYN_E_INVALID, "[Evaluator::bfv_multiply_inplace] null argument");
    if (pa < 1 || pb < 1 || pa > 16 || pb > 16) return fail(TROYN_E_INVALID, "[Evaluator::bfv_multiply_inplace] invalid ciphertext size");
    BehzLayout w = behz_layout(b, pa, pb, batch);
    if (workspace_bytes < w.total * sizeof(u64)) return fail(TROYN_E_WORKSPACE, "[troyn_bfv_multiply] workspace too small");
    if (batch == 0) return TROYN_OK;
    behz_follow_options(b);
    hipStream_t s = (hipStream_t)stream;
    const troyn_plan* pq = b->plan;
    const troyn_plan* px = b->aux;
    const unsigned n = pq->n, L = b->L, S = b->Bsk;
    const size_t po = pa + pb - 1;
    u64* ws = (u64*)workspace;
    int rc;
    const unsigned ch1 = chunks_single(n);
    const bool gen2 = behz2_enabled(b);
    // 2 x 2 components at a two-pass size: the forward transforms stop after their first pass, tensor_core_kernel finishes them, forms the
    // product and starts the inverse transforms
    int tkind = (pa == 2 && pb == 2 && tensor_path_kind(pq, L) == tensor_path_kind(px, S)) ? tensor_path_kind(pq, L) : 0;
    // Whole-limb sizes, a few ciphertexts: tensor_core_kernel puts seven transforms of a limb on ONE workgroup (L + S workgroups per item) -- a launch
    // that cannot fill the chip is one long chain (one product at N = 16384 6 x 50-bit: 240 us).  The separate launches (two-pass transforms of many
    // small workgroups, dyadic product) finish in 92 us (N = 8192 {60,40,40,60}: 245 -> 78 us for one product).  TROYN_BFV_TENSOR=fused / split force either.
    // (the hand-over point, measured: ~200 workgroups where tensor_core_kernel is at its best -- N <= 8192, FP64 policy -- and ~350 where it spills
    // (N = 16384) or runs the integer butterflies)
    const size_t tensor_small = (pq->log_n <= 13 && use_f64(pq, 0, L) && use_f64(px, 0, S)) ? 192 : 352;
    if (tkind == 1 && !pq->opt.tensor_fused && batch * (size_t)(L + S) <= tensor_small) tkind = 0;
    const bool tensor = tkind != 0, whole = tkind == 1;
    // N = 32768 under the FP64 policy: first pass of base q, lift and first pass of the lifted rows as one launch, and the last inverse pass of
    // both bases inside the floor launch (behz2_lift_pass1.hpp)
    const bool lift_fused = tkind == 2 && gen2 && b->aux50 && b->smallq && pq->log_n == 15 && !pq->opt.behz_lift_split
                            && use_f64(pq, 0, L) && use_f64(px, 0, S) && L + S <= BEHZ2_FUSED_MAX_ROWS;
    auto lift = [&](const u64* src, size_t pcount, u64* dst_q, u64* dst_bsk) -> int {
        // steps (1)-(3) of evaluator.cu:50-60 for one operand
        if (lift_fused) {
            if (batch * pcount * 512u > 0x7fffffffull) return fail(TROYN_E_INVALID, "[troyn_bfv_multiply] batch too large for one launch");
            // (lift_fused holds only for shapes both fused launches cover: the floor below skips the last inverse pass on the same condition)
            if (!launch_behz2_lift_pass1(L, batch * pcount, s, b->dev2, src, dst_q, dst_bsk, (const double*)pq->d_fwd_f64, (const double*)px->d_fwd_f64, pq->d_mods, px->d_mods))
                return fail(TROYN_E_INVALID, "[troyn_bfv_multiply] no fused lift kernel for this shape");
            LAUNCH_CHECK();
            return TROYN_OK;
        }
        NttArgs a = contiguous_args(pq, src, dst_q, pcount, L, 0, L, TROYN_IDX_COMPONENTWISE, 0);
        int r = whole ? TROYN_OK : tensor ? tensor_stage(pq, 0, a, a, a, batch, s) : launch_ntt(pq, a, batch, false, s);   // whole: the tensor kernel reads src
        if (r) return r;
        const size_t items = batch * pcount;
        if ((r = check_rows(items, ch1))) return r;
        dim3 grid((unsigned)(items * ch1)), block(256);
        if (gen2) {
            launch_behz2_lift(L, b->smallq, grid.x, s, ch1, b->dev2, src, dst_bsk, b->aux50);
        } else dispatch_bound(L,
            [&] { hipLaunchKernelGGL((behz_lift_kernel<4>), grid, block, 0, s, ch1, b->dev, src, dst_bsk); },
            [&] { hipLaunchKernelGGL((behz_lift_kernel<8>), grid, block, 0, s, ch1, b->dev, src, dst_bsk); },
            [&] { hipLaunchKernelGGL((behz_lift_kernel<16>), grid, block, 0, s, ch1, b->dev, src, dst_bsk); },
            [&] { hipLaunchKernelGGL((behz_lift_kernel<64>), grid, block, 0, s, ch1, b->dev, src, dst_bsk); });
        LAUNCH_CHECK();
        NttArgs ab = contiguous_args(px, dst_bsk, dst_bsk, pcount, S, 0, S, TROYN_IDX_COMPONENTWISE, 0);
        return whole ? TROYN_OK : tensor ? tensor_stage(px, 0, ab, ab, ab, batch, s) : launch_ntt(px, ab, batch, false, s);
    };
    // squaring (a ciphertext multiplied with itself): one lift, and the staged copies of b are those of a
    const bool square = a_ == b_ && pa == pb;
    if (square) { w.b_q = w.a_q; w.b_bsk = w.a_bsk; }
    if ((rc = lift((const u64*)a_, pa, ws + w.a_q, ws + w.a_bsk))) return rc;
    if (!square && (rc = lift((const u64*)b_, pb, ws + w.b_q, ws + w.b_bsk))) return rc;
    if (tensor) {
        // steps (4)-(5)
        for (int base = 0; base < 2; base++) {
            const troyn_plan* p = base ? px : pq;
            const unsigned nc = base ? S : L;
            // whole-limb tiles: base q is read straight from the operands (coefficient form), nothing of it is staged
            const u64* xa = (whole && !base) ? (const u64*)a_ : ws + (base ? w.a_bsk : w.a_q);
            const u64* xb = (whole && !base) ? (const u64*)b_ : ws + (base ? w.b_bsk : w.b_q);
            u64* xd = ws + (base ? w.d_bsk : w.d_q);
            NttArgs fa = contiguous_args(p, xa, nullptr, 2, nc, 0, nc, TROYN_IDX_COMPONENTWISE, 0);
            NttArgs fb = contiguous_args(p, xb, nullptr, 2, nc, 0, nc, TROYN_IDX_COMPONENTWISE, 0);
            NttArgs id = contiguous_args(p, xd, xd, 3, nc, 0, nc, TROYN_IDX_COMPONENTWISE, 0);
            {
                TimerScope ts(TROYN_TIMER_BFV_TENSOR, s);
                if ((rc = tensor_stage(p, 1, fa, fb, id, batch, s))) return rc;
            }
            if (!whole && !lift_fused && (rc = tensor_stage(p, 2, id, id, id, batch, s))) return rc;      // lift_fused: the floor launch runs the last pass
        }
    } else {
    // step (4)
    if ((rc = launch_convolute(pq->d_mods, n, 0, L, ws + w.a_q, pa, ws + w.b_q, pb, ws + w.d_q, batch, s))) return rc;
    if ((rc = launch_convolute(px->d_mods, n, 0, S, ws + w.a_bsk, pa, ws + w.b_bsk, pb, ws + w.d_bsk, batch, s))) return rc;
    // step (5)
    {
        NttArgs a = contiguous_args(pq, ws + w.d_q, ws + w.d_q, po, L, 0, L, TROYN_IDX_COMPONENTWISE, 0);
        if ((rc = launch_ntt(pq, a, batch, true, s))) return rc;
        NttArgs ab = contiguous_args(px, ws + w.d_bsk, ws + w.d_bsk, po, S, 0, S, TROYN_IDX_COMPONENTWISE, 0);
        if ((rc = launch_ntt(px, ab, batch, true, s))) return rc;
    }
    }
    // steps (6)-(8)
    {
        const size_t items = batch * po;
        if ((rc = check_rows(items, ch1))) return rc;
        dim3 grid((unsigned)(items * ch1)), block(256);
        TimerScope ts(TROYN_TIMER_BEHZ_FLOOR, s);
        if (lift_fused) {
            if (items * 512u > 0x7fffffffull) return fail(TROYN_E_INVALID, "[troyn_bfv_multiply] batch too large for one launch");
            if (!launch_behz2_floor_pass2(L, items, s, b->dev2, ws + w.d_q, ws + w.d_bsk, (u64*)out, (const double*)pq->d_inv_f64, (const double*)px->d_inv_f64, pq->d_mods, px->d_mods))
                return fail(TROYN_E_INVALID, "[troyn_bfv_multiply] no fused floor kernel for this shape");
        } else if (gen2) {
            launch_behz2_floor(L, b->smallq, grid.x, s, ch1, b->dev2, ws + w.d_q, ws + w.d_bsk, (u64*)out, b->aux50);
        } else dispatch_bound(S,
            [&] { hipLaunchKernelGGL((behz_floor_kernel<4>), grid, block, 0, s, ch1, b->dev, ws + w.d_q, ws + w.d_bsk, (u64*)out); },
            [&] { hipLaunchKernelGGL((behz_floor_kernel<8>), grid, block, 0, s, ch1, b->dev, ws + w.d_q, ws + w.d_bsk, (u64*)out); },
            [&] { hipLaunchKernelGGL((behz_floor_kernel<16>), grid, block, 0, s, ch1, b->dev, ws + w.d_q, ws + w.d_bsk, (u64*)out); },
            [&] { hipLaunchKernelGGL((behz_floor_kernel<66>), grid, block, 0, s, ch1, b->dev, ws + w.d_q, ws + w.d_bsk, (u64*)out); });
        LAUNCH_CHECK();
    }
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// context PRNG samplers, BFV plaintext scaling, BFV decryption rounding
// ---------------------------------------------------------------------------------------
extern "C" int troyn_prng_block(const uint64_t seed[2], uint64_t counter, uint64_t out[2]) {
    if (!seed || !out) return fail(TROYN_E_INVALID, "[troyn_prng_block] null argument");
    const AesRoundKeys k = aes128_expand(seed[0], seed[1]);
    u64 lo, hi;
    aes128_encrypt_counter(k, counter, 0, lo, hi);
    out[0] = lo; out[1] = hi;
    return TROYN_OK;
}

template <typename K>
static int launch_sampler(K kernel, const char* who, const troyn_plan* p, uint32_t nmod, const uint64_t seed[2], uint64_t counter,
                          uint64_t* out, size_t blocks, uint64_t* blocks_used, troyn_stream_t stream, size_t threads = 0) {
    if (!p || !seed || !out) return fail(TROYN_E_INVALID, std::string(who) + " null argument");
    if (nmod == 0 || nmod > p->K) return fail(TROYN_E_INVALID, std::string(who) + " modulus count out of range");
    const AesRoundKeys k = aes128_expand(seed[0], seed[1]);
    if (!threads) threads = blocks;        // one thread per AES block unless the kernel says otherwise
    hipLaunchKernelGGL(kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       k, (u64)counter, p->d_mods, (unsigned)nmod, p->n, (u64*)out);
    LAUNCH_CHECK();
    if (blocks_used) *blocks_used = blocks;
    return TROYN_OK;
}

extern "C" int troyn_sample_ternary(const troyn_plan* p, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t* out,
                                    uint64_t* blocks_used, troyn_stream_t stream) {
    select_device(p);
    return launch_sampler(sample_ternary_kernel, "[RandomGenerator::sample_poly_ternary]", p, nmod, seed, counter, out,
                          p ? ((size_t)p->n + 15) / 16 : 0, blocks_used, stream, p ? (size_t)p->n : 0);      // (a thread per coefficient)
}
extern "C" int troyn_sample_centered_binomial(const troyn_plan* p, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t* out,
                                              uint64_t* blocks_used, troyn_stream_t stream) {
    select_device(p);
    return launch_sampler(sample_cbd_kernel, "[RandomGenerator::sample_poly_centered_binomial]", p, nmod, seed, counter, out,
                          p ? ((size_t)p->n + 1) / 2 : 0, blocks_used, stream);
}
extern "C" int troyn_sample_uniform(const troyn_plan* p, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t* out,
                                    uint64_t* blocks_used, troyn_stream_t stream) {
    select_device(p);
    return launch_sampler(sample_uniform_kernel, "[RandomGenerator::sample_poly_uniform]", p, nmod, seed, counter, out,
                          p ? ((size_t)p->n * nmod + 1) / 2 : 0, blocks_used, stream);
}

extern "C" int troyn_sample_centered_binomial_strided(const troyn_plan* p, uint32_t nmod, const uint64_t seed[2], uint64_t counter, uint64_t counter_stride,
                                                      uint64_t* out, size_t count, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[RandomGenerator::sample_poly_centered_binomial]";
    if (!p || !seed || !out) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (nmod == 0 || nmod > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus count out of range");
    if (count == 0) return TROYN_OK;
    if (count > 65535) return fail(TROYN_E_INVALID, std::string(P) + " batch too large for one launch");
    const AesRoundKeys k = aes128_expand(seed[0], seed[1]);
    const unsigned blocks = (unsigned)((((size_t)p->n + 1) / 2 + 255) / 256);
    hipLaunchKernelGGL(sample_cbd_strided_kernel, dim3(blocks, (unsigned)count), dim3(256), 0, (hipStream_t)stream,
                       k, (u64)counter, (u64)counter_stride, p->d_mods, (unsigned)nmod, p->n, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_sample_uniform_multi(const troyn_plan* p, uint32_t nmod, const uint64_t* seeds, uint64_t* out, size_t count, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[RandomGenerator::sample_poly_uniform]";
    if (!p || !seeds || !out) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (nmod == 0 || nmod > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus count out of range");
    const size_t per_item = (size_t)nmod * p->n;
    const unsigned blocks = (unsigned)(((per_item + 1) / 2 + 255) / 256);
    for (size_t base = 0; base < count; base += SAMPLE_MULTI_KEYS) {
        const size_t m = std::min<size_t>(SAMPLE_MULTI_KEYS, count - base);
        AesRoundKeysMulti keys;
        std::memset(&keys, 0, sizeof(keys));
        for (size_t i = 0; i < m; i++) keys.k[i] = aes128_expand(seeds[2 * (base + i)], seeds[2 * (base + i) + 1]);
        hipLaunchKernelGGL(sample_uniform_multi_kernel, dim3(blocks, (unsigned)m), dim3(256), 0, (hipStream_t)stream,
                           keys, p->d_mods, (unsigned)nmod, p->n, (u64*)out + base * per_item);
        LAUNCH_CHECK();
    }
    return TROYN_OK;
}

extern "C" int troyn_bfv_scale_up(const troyn_behz* b, const uint64_t* plain, size_t plain_coeff_count, size_t plain_bstride,
                                  const uint64_t* from, size_t from_bstride, uint64_t* dest, size_t dest_bstride,
                                  int subtract, size_t batch, troyn_stream_t stream) {
    select_device(b);
    if (!b || !plain || !dest) return fail(TROYN_E_INVALID, "[scaling_variant::scale_up] null argument");
    const unsigned n = b->plan->n;
    if (plain_coeff_count > n) return fail(TROYN_E_INVALID, "[scaling_variant::scale_up] destination_coeff_count should no less than plain_coeff_count.");
    if (batch == 0) return TROYN_OK;
    ScaleUpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.L = b->L; a.n = n; a.plain_coeff_count = (unsigned)plain_coeff_count; a.subtract = subtract ? 1u : 0u;
    a.mods = b->plan->d_mods; a.delta = b->d_delta; a.t = b->t_mod; a.q_mod_t = b->q_mod_t; a.threshold = (b->t + 1) >> 1;
    a.plain = (const u64*)plain; a.plain_bstride = (long long)plain_bstride;
    a.from = (const u64*)from; a.from_bstride = (long long)from_bstride;
    a.dest = (u64*)dest; a.dest_bstride = (long long)dest_bstride;
    const unsigned ch = chunks_single(n);
    const size_t rows = batch * b->L;
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(bfv_scale_up_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream, ch, a);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_bfv_decrypt_scale_and_round(const troyn_behz* b, const uint64_t* phase, uint64_t* dest, size_t batch, troyn_stream_t stream) {
    select_device(b);
    if (!b || !phase || !dest) return fail(TROYN_E_INVALID, "[RNSTool::decrypt_scale_and_round] null argument");
    if (!b->decrypt_ready) return fail(TROYN_E_MODULUS, "[RNSTool::RNSTool] Unable to invert gamma mod t.");
    if (batch == 0) return TROYN_OK;
    DecryptArgs c;
    std::memset(&c, 0, sizeof(c));
    c.L = b->L; c.n = b->plan->n; c.mods = b->plan->d_mods; c.t = b->t_mod; c.gamma = b->gamma_mod;
    c.prod_t_gamma_mod_q = b->d_prod_t_gamma_mod_q; c.q_inv_punc = b->dev.q_inv_punc;
    c.q_to_t = b->d_q_to_t; c.q_to_gamma = b->d_q_to_gamma;
    c.neg_inv_q_mod_t = b->neg_inv_q_mod_t; c.neg_inv_q_mod_gamma = b->neg_inv_q_mod_gamma; c.inv_gamma_mod_t = b->inv_gamma_mod_t;
    const unsigned ch = chunks_single(c.n);
    if (int rc = check_rows(batch, ch)) return rc;
    hipLaunchKernelGGL(bfv_decrypt_round_kernel, dim3((unsigned)(batch * ch)), dim3(256), 0, (hipStream_t)stream, ch, c, (const u64*)phase, (u64*)dest);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" uint64_t troyn_behz_gamma(const troyn_behz* b) { return b ? b->gamma : 0; }

// ---------------------------------------------------------------------------------------
// ciphertext x plaintext
// ---------------------------------------------------------------------------------------
extern "C" int troyn_plain_centralize(const troyn_plan* p, uint32_t L, uint64_t t, const uint64_t* plain, size_t plain_coeff_count,
                                      size_t plain_bstride, uint64_t* dest, size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !plain || !dest) return fail(TROYN_E_INVALID, "[scaling_variant::centralize] null argument");
    if (L < 1 || L > p->K) return fail(TROYN_E_INVALID, "[scaling_variant::centralize] Destination has incorrect size.");
    if (plain_coeff_count > p->n) return fail(TROYN_E_INVALID, "[scaling_variant::centralize] destination_coeff_count should no less than plain_coeff_count.");
    if (batch == 0) return TROYN_OK;
    const unsigned ch = chunks_single(p->n);
    const size_t rows = batch * L;
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(plain_centralize_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, (unsigned)L, p->n, (u64)t, (const u64*)plain, (unsigned)plain_coeff_count, (long long)plain_bstride, (u64*)dest);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_plain_centralize_ntt(const troyn_plan* p, uint32_t L, uint64_t t, const uint64_t* plain, size_t plain_coeff_count,
                                          size_t plain_bstride, uint64_t* dest, size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !plain || !dest) return fail(TROYN_E_INVALID, "[Evaluator::transform_plain_to_ntt] null argument");
    if (L < 1 || L > p->K) return fail(TROYN_E_INVALID, "[scaling_variant::centralize] Destination has incorrect size.");
    if (plain_coeff_count > p->n) return fail(TROYN_E_INVALID, "[scaling_variant::centralize] destination_coeff_count should no less than plain_coeff_count.");
    if (batch == 0) return TROYN_OK;
    bool fast = p->log_n >= 10 && t >= 2;
    for (uint32_t j = 0; fast && j < L; j++) fast = t < p->moduli[j];
    if (!fast) {
        // the general lift (t not below every q_j) and the rings the optimised transforms do not cover: the two launches
        if (int rc = troyn_plain_centralize(p, L, t, plain, plain_coeff_count, plain_bstride, dest, batch, stream)) return rc;
        return troyn_ntt(p, 0, dest, dest, batch, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0, stream);
    }
    // one forward launch whose loader reads the plaintext row (shared by the L limbs: component stride 0, the limbs of an item co-located on an XCD)
    NttArgs a = contiguous_args(p, (const u64*)plain, (u64*)dest, 1, L, 0, L, TROYN_IDX_COMPONENTWISE, 0);
    a.in_cstride = 0; a.in_pstride = 0; a.in_bstride = (long long)plain_bstride;
    a.load_mode = NTT_LOAD_CENTRALIZE; a.cz_t = t; a.cz_count = (unsigned)plain_coeff_count;
    return launch_ntt(p, a, batch, false, (hipStream_t)stream);
}

extern "C" int troyn_dyadic_broadcast_product(const troyn_plan* p, uint32_t mod_start, uint32_t nmod, const uint64_t* ct, size_t pcount,
                                              const uint64_t* pt, size_t pt_bstride, uint64_t* out, size_t batch, troyn_stream_t stream) {
    select_device(p);
    if (!p || !ct || !pt || !out) return fail(TROYN_E_INVALID, "[dyadic_product_ps] null argument");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, "[dyadic_product_ps] modulus slice out of range");
    if (batch == 0 || pcount == 0) return TROYN_OK;
    const unsigned ch = chunks_pairs(p->n);
    const size_t rows = batch * nmod;
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(dyadic_broadcast_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, mod_start, nmod, p->n, (unsigned)pcount, (const u64*)ct, (const u64*)pt, (long long)pt_bstride, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_apply_galois(const troyn_plan* p, uint32_t mod_start, uint32_t nmod, int is_ntt_form, uint64_t galois_element,
                                  const uint64_t* in, uint64_t* out, size_t count, troyn_stream_t stream) {
    select_device(p);
    if (!p || !in || !out) return fail(TROYN_E_INVALID, "[GaloisTool::apply] null argument");
    if (in == out) return fail(TROYN_E_INVALID, "[GaloisTool::apply] the permutation cannot run in place");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, "[GaloisTool::apply] modulus slice out of range");
    if ((galois_element & 1) == 0 || galois_element >= 2ull * p->n) return fail(TROYN_E_INVALID, "[Evaluator::apply_galois_inplace] Galois element is not valid.");
    if (p->log_n < 1) return fail(TROYN_E_INVALID, "[GaloisTool::GaloisTool] coeff_count_power is invalid");
    const size_t rows = count * nmod;
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_single(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(galois_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, mod_start, nmod, p->log_n, (unsigned)galois_element, is_ntt_form ? 1 : 0, (const u64*)in, (u64*)out, (u64)0);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_apply_galois_plain(const troyn_plan* p, uint64_t modulus, uint64_t galois_element, const uint64_t* in, uint64_t* out, size_t count,
                                        troyn_stream_t stream) {
    select_device(p);
    if (!p || !in || !out) return fail(TROYN_E_INVALID, "[GaloisTool::apply] null argument");
    if (in == out) return fail(TROYN_E_INVALID, "[GaloisTool::apply] the permutation cannot run in place");
    if (modulus < 2) return fail(TROYN_E_INVALID, "[GaloisTool::apply] modulus is invalid");
    if ((galois_element & 1) == 0 || galois_element >= 2ull * p->n) return fail(TROYN_E_INVALID, "[Evaluator::apply_galois_inplace] Galois element is not valid.");
    if (count == 0) return TROYN_OK;
    const unsigned ch = chunks_single(p->n);
    if (int rc = check_rows(count, ch)) return rc;
    hipLaunchKernelGGL(galois_kernel, dim3((unsigned)(count * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, 0u, 1u, p->log_n, (unsigned)galois_element, 0, (const u64*)in, (u64*)out, (u64)modulus);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_negacyclic_shift(const troyn_plan* p, uint32_t mod_start, uint32_t nmod, const uint64_t* in, uint64_t* out, size_t shift,
                                      size_t count, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[negacyclic_shift_ps]";
    if (!p || !in || !out) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (in == out) return fail(TROYN_E_INVALID, std::string(P) + " the shift cannot run in place");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus slice out of range");
    shift %= 2ull * p->n;      // the reference takes any shift: index (shift + k) & (N - 1), sign from bit log2 N of shift + k (utils/poly_small_mod.cu:927-944)
    const size_t rows = count * nmod;
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_single(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(negacyclic_shift_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, mod_start, nmod, p->log_n, (unsigned)shift, (const u64*)in, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_multiply_inv_degree(const troyn_plan* p, uint32_t mod_start, uint32_t nmod, const uint64_t* in, uint64_t* out, uint64_t scalar,
                                         size_t count, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[ntt_multiply_inv_degree]";
    if (!p || !in || !out) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus slice out of range");
    const size_t rows = count * nmod;
    if (rows == 0) return TROYN_OK;
    const unsigned ch = chunks_single(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(multiply_inv_degree_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, mod_start, nmod, p->n, (u64)scalar, (const u64*)in, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" size_t troyn_pack_prepare_workspace_bytes(size_t slots) { return (slots + 1) * sizeof(u64); }

extern "C" int troyn_pack_prepare(const troyn_plan* p, uint32_t L, size_t pcount, const uint64_t* const* src, size_t slots, uint64_t mul, size_t shift,
                                  uint64_t* out, void* workspace, size_t workspace_bytes, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[Evaluator::pack_rlwe_ciphertexts_new]";
    if (!p || !src || !out || !workspace) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (L == 0 || L > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus count out of range");
    shift %= 2ull * p->n;      // the reference takes any shift: index (shift + k) & (N - 1), sign from bit log2 N of shift + k (utils/poly_small_mod.cu:927-944)
    if (slots == 0 || pcount == 0) return TROYN_OK;
    if (workspace_bytes < troyn_pack_prepare_workspace_bytes(slots)) return fail(TROYN_E_WORKSPACE, "[troyn_pack_prepare] workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = upload_host_table(s, workspace, src, slots * sizeof(u64))) return rc;
    const size_t rows = slots * pcount * L;
    const unsigned ch = chunks_single(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(pack_prepare_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                       ch, p->d_mods, (unsigned)L, p->log_n, (unsigned)pcount, (u64)mul, (unsigned)shift, (const u64* const*)workspace, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_pack_layer(const troyn_plan* p, uint32_t L, uint64_t galois_element, size_t shift, const uint64_t* in, uint64_t* out,
                                uint64_t* target, size_t pairs, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[Evaluator::pack_rlwe_ciphertexts_new]";
    if (!p || !in || !out || !target) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (L == 0 || L > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus count out of range");
    if ((galois_element & 1) == 0 || galois_element >= 2ull * p->n) return fail(TROYN_E_INVALID, "[Evaluator::apply_galois_inplace] Galois element is not valid.");
    shift %= 2ull * p->n;      // the reference takes any shift: index (shift + k) & (N - 1), sign from bit log2 N of shift + k (utils/poly_small_mod.cu:927-944)
    if (pairs == 0) return TROYN_OK;
    // g^-1 mod 2N (g odd): Newton iteration on the 2-adic inverse
    const u64 two_n = 2ull * p->n;
    u64 inv = galois_element;
    for (int it = 0; it < 6; it++) inv = inv * (2 - galois_element * inv);
    inv &= two_n - 1;
    const size_t rows = pairs * 2 * L;
    const unsigned ch = chunks_single(p->n);
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(pack_layer_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, (hipStream_t)stream,
                       ch, p->d_mods, (unsigned)L, p->log_n, (unsigned)shift, (unsigned)inv, (const u64*)in, (u64*)out, (u64*)target);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" size_t troyn_extract_lwe_workspace_bytes(size_t count) { return (count + (count + 1) / 2 + 1) * sizeof(u64); }

extern "C" int troyn_extract_lwe(const troyn_plan* p, uint32_t L, const uint64_t* const* ct, const size_t* terms, uint64_t* c0, uint64_t* c1, size_t count,
                                 void* workspace, size_t workspace_bytes, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[Evaluator::extract_lwe_new]";
    if (!p || !ct || !terms || !c0 || !c1 || !workspace) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (L == 0 || L > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus count out of range");
    if (count == 0) return TROYN_OK;
    if (count * L > 0x7fffffffull) return fail(TROYN_E_INVALID, std::string(P) + " batch too large for one launch");
    if (workspace_bytes < troyn_extract_lwe_workspace_bytes(count)) return fail(TROYN_E_WORKSPACE, "[troyn_extract_lwe] workspace too small");
    const size_t n = p->n, pc = (size_t)L * n;
    std::vector<u64> tab(count + (count + 1) / 2, 0);
    unsigned* t32 = reinterpret_cast<unsigned*>(tab.data() + count);
    for (size_t i = 0; i < count; i++) {
        if (!ct[i]) return fail(TROYN_E_INVALID, std::string(P) + " null pointer in the batch");
        if (terms[i] >= n) return fail(TROYN_E_INVALID, std::string(P) + " term out of range");
        tab[i] = (u64)(uintptr_t)ct[i];
        t32[i] = (unsigned)terms[i];
    }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = upload_host_table(s, workspace, tab.data(), tab.size() * sizeof(u64))) return rc;
    // c1 of the LWE = c1 of the RLWE shifted by 2N - term (term 0: unshifted), evaluator_lwes.cu:72-79
    const unsigned ch = chunks_single(p->n);
    for (size_t i = 0; i < count; i++) {
        const size_t shift = terms[i] == 0 ? 0 : 2 * n - terms[i];
        hipLaunchKernelGGL(negacyclic_shift_kernel, dim3((unsigned)(L * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, 0u, (unsigned)L, p->log_n, (unsigned)shift, (const u64*)ct[i] + pc, (u64*)c1 + i * pc);
    }
    hipLaunchKernelGGL(extract_lwe_c0_kernel, dim3((unsigned)((count * L + 255) / 256)), dim3(256), 0, s,
                       (unsigned)L, (unsigned)n, (const u64* const*)workspace, (const unsigned*)((const u64*)workspace + count), (u64*)c0, (unsigned)count);
    LAUNCH_CHECK();
    return TROYN_OK;
}

// ---------------------------------------------------------------------------------------
// ring-2^k encoder (src/app/bfv_ring2k.cu): constants of PolynomialEncoderRNSHelper for one level
// ---------------------------------------------------------------------------------------
struct troyn_ring2k {
    const troyn_plan* plan = nullptr;
    u64* d_consts = nullptr;
    Ring2kDev dev;
    u64 gamma = 0;
};

extern "C" int troyn_ring2k_destroy(troyn_ring2k* h) {
    if (!h) return TROYN_OK;
    if (h->d_consts) (void)hipFree(h->d_consts);
    delete h;
    return TROYN_OK;
}

extern "C" int troyn_ring2k_create(troyn_ring2k** out, const troyn_plan* plan, uint32_t L, uint32_t t_bits, uint32_t elem_bytes) {
    const char* P = "[PolynomialEncoderRNSHelper::PolynomialEncoderRNSHelper]";
    if (!out || !plan) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    *out = nullptr;
    if (L < 1 || L > plan->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus count out of range");
    if (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return fail(TROYN_E_INVALID, std::string(P) + " T must be uint32_t, uint64_t or uint128_t");
    if (t_bits <= elem_bytes * 4 || t_bits > elem_bytes * 8) return fail(TROYN_E_INVALID, std::string(P) + " t_bit_length must be greater than type_bits<T>() / 2");
    typedef unsigned __int128 u128h;
    std::unique_ptr<troyn_ring2k, int (*)(troyn_ring2k*)> h(new troyn_ring2k, troyn_ring2k_destroy);
    h->plan = plan;
    const std::vector<u64> q(plan->moduli.begin(), plan->moduli.begin() + L);
    u64 gamma;
    try { gamma = host::get_primes((u64)plan->n, 61, 1)[0]; } catch (const std::exception& e) { return fail(TROYN_E_MODULUS, e.what()); }
    for (u64 v : q) if (v == gamma) return fail(TROYN_E_MODULUS, std::string(P) + " gamma is in coeff_modulus");
    h->gamma = gamma;
    const u128h mask = t_bits == 128 ? ~(u128h)0 : (((u128h)1 << t_bits) - 1);
    auto wrap_product = [&](size_t except) { u128h acc = 1; for (size_t k = 0; k < q.size(); k++) if (k != except) acc *= q[k]; return acc; };   // mod 2^128
    auto inverse_2k = [](u128h x) { u128h inv = x; for (int it = 0; it < 8; it++) inv *= 2 - x * inv; return inv; };                            // x odd
    const u128h Q128 = wrap_product(SIZE_MAX);
    const u128h q_mod_t = Q128 & mask, neg_inv_q = (0 - inverse_2k(Q128)) & mask, inv_gamma = inverse_2k((u128h)gamma) & mask, t_half = (u128h)1 << (t_bits - 1);
    // floor(Q / 2^k) mod q_l from the multi-word product
    std::vector<u64> big = host::big_product(q), shifted(big.size(), 0);
    for (size_t w = 0; w < big.size(); w++) {
        const size_t src = w + t_bits / 64;
        u64 v = src < big.size() ? big[src] >> (t_bits % 64) : 0;
        if ((t_bits % 64) && src + 1 < big.size()) v |= big[src + 1] << (64 - t_bits % 64);
        shifted[w] = v;
    }
    std::vector<u64> blob;
    auto push_shoup = [&](u64 w, u64 m) { host::Shoup s = host::shoup(w % m, m); blob.push_back(s.operand); blob.push_back(s.quotient); };
    const size_t off_qdt = blob.size();
    for (size_t l = 0; l < L; l++) push_shoup(host::big_mod_small(shifted, q[l]), q[l]);
    const size_t off_gt = blob.size();
    for (size_t l = 0; l < L; l++) {
        // gamma * 2^k mod q_l, with 2^k split as 2^(k/2) * 2^(k - k/2) (bfv_ring2k.cu:183-191)
        const u64 t0 = (u64)(((u128h)1 << (t_bits / 2)) % q[l]), t1 = (u64)(((u128h)1 << (t_bits - t_bits / 2)) % q[l]);
        push_shoup(host::mulmod(gamma % q[l], host::mulmod(t0, t1, q[l]), q[l]), q[l]);
    }
    const size_t off_ip = blob.size();
    for (size_t l = 0; l < L; l++) {
        u64 inv = 1;
        if (L > 1 && !host::invmod(host::product_mod(q, l, q[l]), q[l], inv)) return fail(TROYN_E_MODULUS, "[RNSBase::initialize] RNSBase product is not invertible.");
        push_shoup(inv, q[l]);
    }
    const size_t off_pg = blob.size();
    for (size_t l = 0; l < L; l++) blob.push_back(host::product_mod(q, l, gamma));
    if (blob.size() & 1) blob.push_back(0);
    const size_t off_pt = blob.size();
    for (size_t l = 0; l < L; l++) { const u128h v = wrap_product(l) & mask; blob.push_back((u64)v); blob.push_back((u64)(v >> 64)); }
    u64 inv_q_gamma = 0;
    if (!host::invmod(host::product_mod(q, SIZE_MAX, gamma), gamma, inv_q_gamma)) return fail(TROYN_E_MODULUS, std::string(P) + " failed to invert Q_mod_gamma");
    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipMalloc(&h->d_consts, blob.size() * sizeof(u64)));
    HIP_TRY(hipMemcpy(h->d_consts, blob.data(), blob.size() * sizeof(u64), hipMemcpyHostToDevice));
    Ring2kDev& d = h->dev;
    std::memset(&d, 0, sizeof(d));
    d.mods = plan->d_mods;
    d.q_div_t_mod_q = reinterpret_cast<const ulonglong2*>(h->d_consts + off_qdt);
    d.gamma_t_mod_q = reinterpret_cast<const ulonglong2*>(h->d_consts + off_gt);
    d.inv_punctured = reinterpret_cast<const ulonglong2*>(h->d_consts + off_ip);
    d.punctured_mod_gamma = h->d_consts + off_pg;
    d.punctured_mod_t = h->d_consts + off_pt;
    d.gamma = make_dev_modulus(gamma, plan->log_n, false);
    { host::Shoup s = host::shoup((gamma - inv_q_gamma) % gamma, gamma); d.neg_inv_q_mod_gamma = make_ulonglong2(s.operand, s.quotient); }
    auto set2 = [](u64* dst, u128h v) { dst[0] = (u64)v; dst[1] = (u64)(v >> 64); };
    set2(d.q_mod_t, q_mod_t); set2(d.t_half, t_half); set2(d.mask, mask); set2(d.neg_inv_q_mod_t, neg_inv_q); set2(d.inv_gamma_mod_t, inv_gamma);
    d.L = L; d.n = plan->n; d.t_bits = t_bits; d.elem_bytes = elem_bytes;
    *out = h.release();
    return TROYN_OK;
}

extern "C" uint64_t troyn_ring2k_gamma(const troyn_ring2k* h) { return h ? h->gamma : 0; }

static int ring2k_encode(const troyn_ring2k* h, bool scale, const void* src, size_t count, uint64_t* out, troyn_stream_t stream) {
    const char* P = scale ? "[PolynomialEncoderRNSHelper:scale_up]" : "[PolynomialEncoderRNSHelper:centralize]";
    if (!h || !out || (!src && count)) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (count > h->dev.n) return fail(TROYN_E_INVALID, std::string(P) + " source size is larger than poly_modulus_degree");
    const dim3 grid((h->dev.n + 255) / 256), block(256);
    if (scale) hipLaunchKernelGGL(ring2k_scale_up_kernel, grid, block, 0, (hipStream_t)stream, h->dev, src, (unsigned)count, (u64*)out);
    else hipLaunchKernelGGL(ring2k_centralize_kernel, grid, block, 0, (hipStream_t)stream, h->dev, src, (unsigned)count, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_ring2k_scale_up(const troyn_ring2k* h, const void* src, size_t count, uint64_t* out, troyn_stream_t stream) {
    select_device(h); return ring2k_encode(h, true, src, count, out, stream); }
extern "C" int troyn_ring2k_centralize(const troyn_ring2k* h, const void* src, size_t count, uint64_t* out, troyn_stream_t stream) {
    select_device(h); return ring2k_encode(h, false, src, count, out, stream); }

extern "C" int troyn_ring2k_scale_down(const troyn_ring2k* h, const uint64_t* in, void* dst, troyn_stream_t stream) {
    select_device(h);
    if (!h || !in || !dst) return fail(TROYN_E_INVALID, "[PolynomialEncoderRNSHelper::scale_down] null argument");
    hipLaunchKernelGGL(ring2k_scale_down_kernel, dim3((h->dev.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dev, (const u64*)in, dst);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_ring2k_decentralize(const troyn_ring2k* h, const uint64_t* in, void* dst, uint64_t correction_lo, uint64_t correction_hi, troyn_stream_t stream) {
    select_device(h);
    const char* P = "[PolynomialEncoderRNSHelper::decentralize]";
    if (!h || !in || !dst) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    typedef unsigned __int128 u128h;
    u128h cf = ((u128h)correction_hi << 64) | correction_lo;
    if (h->dev.elem_bytes == 4) cf &= 0xffffffffull; else if (h->dev.elem_bytes == 8) cf &= ~0ull;
    if ((cf & 1) == 0) return fail(TROYN_E_INVALID, "[bfv_ring2k::inverse_ring2k] x must be odd");
    u128h fix = cf;                                                  // Newton: the inverse modulo 2^128 (its low k bits are the inverse modulo 2^k)
    for (int it = 0; it < 8; it++) fix *= 2 - cf * fix;
    const u128h mask = h->dev.t_bits == 128 ? ~(u128h)0 : (((u128h)1 << h->dev.t_bits) - 1);
    fix &= mask;
    hipLaunchKernelGGL(ring2k_decentralize_kernel, dim3((h->dev.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dev, (const u64*)in, dst, (u64)fix, (u64)(fix >> 64));
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" size_t troyn_gather_workspace_bytes(size_t count) { return (count + 1) * sizeof(u64); }

extern "C" int troyn_gather(const uint64_t* const* src, size_t count, size_t words, uint64_t* out, void* workspace, size_t workspace_bytes, troyn_stream_t stream) {
    if (!src || !out || !workspace) return fail(TROYN_E_INVALID, "[troyn_gather] null argument");
    if (count == 0 || words == 0) return TROYN_OK;
    if (count > 65535) return fail(TROYN_E_INVALID, "[troyn_gather] batch too large for one launch");
    if (workspace_bytes < troyn_gather_workspace_bytes(count)) return fail(TROYN_E_WORKSPACE, "[troyn_gather] workspace too small");
    for (size_t i = 0; i < count; i++)
        if (!src[i] || ((uintptr_t)src[i] & 15)) return fail(TROYN_E_INVALID, "[troyn_gather] null or misaligned pointer in the batch");
    if ((uintptr_t)out & 15) return fail(TROYN_E_INVALID, "[troyn_gather] misaligned destination");
    hipStream_t s = (hipStream_t)stream;
    const unsigned bx = (unsigned)std::min<size_t>((words / 2 + 255) / 256 + 1, 64);
    if (count <= 64) {   // pointers by value: no upload, no host wait
        GatherPtrs g;
        for (size_t i = 0; i < 64; i++) g.p[i] = reinterpret_cast<const u64*>(src[i < count ? i : 0]);
        hipLaunchKernelGGL(gather_small_kernel, dim3(bx, (unsigned)count), dim3(256), 0, s, g, words, (u64*)out);
        LAUNCH_CHECK();
        return TROYN_OK;
    }
    if (int rc = upload_host_table(s, workspace, src, count * sizeof(u64))) return rc;
    hipLaunchKernelGGL(gather_kernel, dim3(bx, (unsigned)count), dim3(256), 0, s, (const u64* const*)workspace, words, (u64*)out);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" int troyn_scatter(const uint64_t* in, uint64_t* const* dst, size_t count, size_t words, void* workspace, size_t workspace_bytes, troyn_stream_t stream) {
    if (!in || !dst || !workspace) return fail(TROYN_E_INVALID, "[troyn_scatter] null argument");
    if (count == 0 || words == 0) return TROYN_OK;
    if (count > 65535) return fail(TROYN_E_INVALID, "[troyn_scatter] batch too large for one launch");
    if (workspace_bytes < troyn_gather_workspace_bytes(count)) return fail(TROYN_E_WORKSPACE, "[troyn_scatter] workspace too small");
    for (size_t i = 0; i < count; i++)
        if (!dst[i] || ((uintptr_t)dst[i] & 15)) return fail(TROYN_E_INVALID, "[troyn_scatter] null or misaligned pointer in the batch");
    if ((uintptr_t)in & 15) return fail(TROYN_E_INVALID, "[troyn_scatter] misaligned source");
    hipStream_t s = (hipStream_t)stream;
    const unsigned bx = (unsigned)std::min<size_t>((words / 2 + 255) / 256 + 1, 64);
    if (count <= 64) {
        ScatterPtrs g;
        for (size_t i = 0; i < 64; i++) g.p[i] = reinterpret_cast<u64*>(dst[i < count ? i : 0]);
        hipLaunchKernelGGL(scatter_small_kernel, dim3(bx, (unsigned)count), dim3(256), 0, s, (const u64*)in, g, words);
        LAUNCH_CHECK();
        return TROYN_OK;
    }
    if (int rc = upload_host_table(s, workspace, dst, count * sizeof(u64))) return rc;
    hipLaunchKernelGGL(scatter_kernel, dim3(bx, (unsigned)count), dim3(256), 0, s, (const u64*)in, (u64* const*)workspace, words);
    LAUNCH_CHECK();
    return TROYN_OK;
}

extern "C" size_t troyn_multiply_plain_accumulate_workspace_bytes(size_t count) { return (4 * count + 2) * sizeof(u64); }

extern "C" int troyn_multiply_plain_accumulate(const troyn_plan* p, uint32_t mod_start, uint32_t nmod, size_t pcount,
                                               const uint64_t* const* ct, const uint64_t* const* pt, uint64_t* const* dst, size_t count,
                                               int set_zero, void* workspace, size_t workspace_bytes, troyn_stream_t stream) {
    select_device(p);
    const char* P = "[Evaluator::multiply_plain_ntt_batched]";
    if (!p || !ct || !pt || !dst || !workspace) return fail(TROYN_E_INVALID, std::string(P) + " null argument");
    if (nmod == 0 || mod_start + nmod > p->K) return fail(TROYN_E_INVALID, std::string(P) + " modulus slice out of range");
    if (count == 0 || pcount == 0) return TROYN_OK;
    if (workspace_bytes < troyn_multiply_plain_accumulate_workspace_bytes(count)) return fail(TROYN_E_WORKSPACE, "[troyn_multiply_plain_accumulate] workspace too small");
    if (count > 0x7fffffffull) return fail(TROYN_E_INVALID, std::string(P) + " too many terms");
    // group the terms by destination (stable), so that each destination is produced by one pass over its terms
    std::vector<size_t> order(count);
    for (size_t i = 0; i < count; i++) {
        if (!ct[i] || !pt[i] || !dst[i]) return fail(TROYN_E_INVALID, std::string(P) + " null pointer in the batch");
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return (uintptr_t)dst[a] < (uintptr_t)dst[b]; });
    std::vector<u64> tab(2 * count);
    std::vector<u64> gdst, gstart;
    for (size_t i = 0; i < count; i++) {
        const size_t k = order[i];
        tab[i] = (u64)(uintptr_t)ct[k];
        tab[count + i] = (u64)(uintptr_t)pt[k];
        if (i == 0 || dst[k] != dst[order[i - 1]]) { gdst.push_back((u64)(uintptr_t)dst[k]); gstart.push_back((u64)i); }
    }
    gstart.push_back((u64)count);
    const size_t groups = gdst.size();
    tab.insert(tab.end(), gdst.begin(), gdst.end());
    tab.insert(tab.end(), gstart.begin(), gstart.end());
    hipStream_t s = (hipStream_t)stream;
    if (int rc = upload_host_table(s, workspace, tab.data(), tab.size() * sizeof(u64))) return rc;
    const unsigned ch = chunks_pairs(p->n);
    const int pm = p->opt.plain_mac;          // TROYN_PLAIN_MAC: 1 v1 (the kernel of pcount != 2, one polynomial per thread), 2 single, 3 dual, 4 quad
    const int mac_gen = pm == 1 ? 1 : 2;
    if (pcount == 2 && mac_gen == 2) {
        const size_t rows2 = groups * nmod;
        if (int rc = check_rows(rows2, ch)) return rc;
        TimerScope ts(TROYN_TIMER_PLAIN_MAC, s);
        // ND consecutive destinations with the same ciphertext operands term by term (the columns of a matmul row) share one workgroup: every
        // ciphertext word is loaded once for all of them (TROYN_PLAIN_MAC=single / dual / quad forces the grouping; A/B, tests)
        auto shared_operands = [&](size_t nd) {
            if (groups < nd || groups % nd != 0) return false;
            for (size_t g = 0; g < groups; g += nd) {
                const size_t a0 = (size_t)gstart[g], len = (size_t)gstart[g + 1] - a0;
                for (size_t d = 1; d < nd; d++) {
                    const size_t b0 = (size_t)gstart[g + d];
                    if ((size_t)gstart[g + d + 1] - b0 != len || b0 != a0 + d * len) return false;
                    for (size_t j = 0; j < len; j++) if (tab[a0 + j] != tab[b0 + j]) return false;
                }
            }
            return true;
        };
        size_t nd = 1;
        if (pm != 2) {
            if (pm != 3 && shared_operands(4)) nd = 4;
            else if (pm != 4 && shared_operands(2)) nd = 2;
        }
        if (nd > 1) {
            const size_t rowsd = (groups / nd) * nmod;
            if (nd == 4)
                hipLaunchKernelGGL((plain_mac2_multi_kernel<2, 4>), dim3((unsigned)(rowsd * ch)), dim3(POLY_BLOCK), 0, s,
                                   ch, p->d_mods, mod_start, nmod, p->n, (const u64*)workspace, (unsigned)count, (unsigned)groups, set_zero ? 1 : 0);
            else
                hipLaunchKernelGGL((plain_mac2_multi_kernel<2, 2>), dim3((unsigned)(rowsd * ch)), dim3(POLY_BLOCK), 0, s,
                                   ch, p->d_mods, mod_start, nmod, p->n, (const u64*)workspace, (unsigned)count, (unsigned)groups, set_zero ? 1 : 0);
            LAUNCH_CHECK();
            return TROYN_OK;
        }
        // (the packed weight layout of round 4 -- plain_mac2_kernel<2, true>, 0.506 -> 0.537 of HBM, profiles/r04_plain_mac_ab.txt -- was a
        // development experiment with a second plaintext representation; its instantiation left the library in round 5)
        hipLaunchKernelGGL((plain_mac2_kernel<2>), dim3((unsigned)(rows2 * ch)), dim3(POLY_BLOCK), 0, s,
                           ch, p->d_mods, mod_start, nmod, p->n, (const u64*)workspace, (unsigned)count, (unsigned)groups, set_zero ? 1 : 0);
        LAUNCH_CHECK();
        return TROYN_OK;
    }
    const size_t rows = groups * pcount * nmod;
    if (int rc = check_rows(rows, ch)) return rc;
    hipLaunchKernelGGL(plain_mac_kernel, dim3((unsigned)(rows * ch)), dim3(POLY_BLOCK), 0, s,
                       ch, p->d_mods, mod_start, nmod, p->n, (unsigned)pcount, (const u64*)workspace, (unsigned)count, (unsigned)groups, set_zero ? 1 : 0);
    LAUNCH_CHECK();
    return TROYN_OK;
}
