// C++ driver for tests/test_mirror_host.py (CPU) and tests/test_gpu_cpp_api.py::test_context_cpp_api: the reference's test/he_context.cu replayed through the
// mirror -- HeContext::create records WHY a parameter set is refused (EncryptionParameterQualifiers::parameter_error) instead of throwing, the qualifiers
// keep what was established before the failing check, the modulus chain stops where a level becomes invalid, total_coeff_modulus / chain_index /
// prev_context_data of every level.  `context_driver host` touches no device; `context_driver device` adds to_device_inplace for the three schemes.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;
using E = EncryptionParameterErrorType;

static int failures = 0;
static void check(bool ok, const char* what) {
    std::printf("%-84s %s\n", what, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

static std::vector<Modulus> to_moduli(std::initializer_list<uint64_t> v) { std::vector<Modulus> m; for (uint64_t x : v) m.push_back(Modulus(x)); return m; }

// he_context.cu:20-58: the first level's qualifiers, except the descending-chain flag which is read from the key level when `key_descend`
static bool qualifiers_are(HeContextPointer context, E result, bool parameters_set, bool fft, bool ntt, bool batching, bool fast_plain_lift, bool descending_chain,
                           SecurityLevel sec_level, bool keyswitching, bool key_descend) {
    const EncryptionParameterQualifiers& q = context->first_context_data().value()->qualifiers();
    const bool descending = key_descend ? context->key_context_data().value()->qualifiers().using_descending_modulus_chain : q.using_descending_modulus_chain;
    const bool ok = q.parameter_error == result && q.parameters_set() == parameters_set && q.using_fft == fft && q.using_ntt == ntt && q.using_batching == batching &&
                    q.using_fast_plain_lift == fast_plain_lift && descending == descending_chain && q.security_level == sec_level && context->using_keyswitching() == keyswitching;
    if (!ok) std::printf("   error %d set %d fft %d ntt %d batching %d lift %d descending %d sec %d keyswitching %d\n", int(q.parameter_error), int(q.parameters_set()), int(q.using_fft),
                         int(q.using_ntt), int(q.using_batching), int(q.using_fast_plain_lift), int(descending), int(q.security_level), int(context->using_keyswitching()));
    return ok;
}

static void run_construct() {
    const SecurityLevel nil = SecurityLevel::Nil;
    EncryptionParameters parms(SchemeType::BFV);
    auto set = [&](size_t n, std::vector<Modulus> q, uint64_t t) { parms.set_poly_modulus_degree(n); parms.set_coeff_modulus(q); parms.set_plain_modulus(Modulus(t)); };
    check(qualifiers_are(HeContext::create(parms, false, nil), E::InvalidCoeffModulusSize, false, false, false, false, false, false, nil, false, false), "no coefficient modulus: InvalidCoeffModulusSize");
    set(4, to_moduli({2, 30}), 2);
    check(qualifiers_are(HeContext::create(parms, false, nil), E::FailedCreatingRNSBase, false, true, false, false, false, false, nil, false, false), "q = {2, 30} (not coprime): FailedCreatingRNSBase, fft only");
    set(4, to_moduli({17, 41}), 34);
    check(qualifiers_are(HeContext::create(parms, false, nil), E::InvalidPlainModulusCoprimality, false, true, true, false, false, false, nil, false, false), "q = {17, 41}, t = 34: InvalidPlainModulusCoprimality");
    set(4, to_moduli({17}), 41);
    check(qualifiers_are(HeContext::create(parms, false, nil), E::InvalidPlainModulusTooLarge, false, true, true, false, false, false, nil, false, false), "q = {17}, t = 41: InvalidPlainModulusTooLarge");
    set(4, to_moduli({3}), 2);
    check(qualifiers_are(HeContext::create(parms, false, nil), E::InvalidCoeffModulusNoNTT, false, true, false, false, false, false, nil, false, false), "q = {3}: InvalidCoeffModulusNoNTT");
    set(4, to_moduli({17, 41}), 18);
    HeContextPointer c = HeContext::create(parms, false, nil);
    check(c->first_context_data().value()->total_coeff_modulus()[0] == 697 && qualifiers_are(c, E::Success, true, true, true, false, false, false, nil, false, false),
          "q = {17, 41}, t = 18: one level (17 alone is below t), Q = 697");
    set(4, to_moduli({17, 41}), 16);
    c = HeContext::create(parms, false, nil);
    check(c->first_context_data().value()->total_coeff_modulus()[0] == 17 && c->key_context_data().value()->total_coeff_modulus()[0] == 697 &&
          qualifiers_are(c, E::Success, true, true, true, false, true, false, nil, true, true), "q = {17, 41}, t = 16: key level 697, first level 17, fast plain lift");
    set(4, to_moduli({17, 41}), 49);
    c = HeContext::create(parms, false, nil);
    check(c->first_context_data().value()->total_coeff_modulus()[0] == 697 && qualifiers_are(c, E::Success, true, true, true, false, false, false, nil, false, false), "q = {17, 41}, t = 49");
    set(4, to_moduli({17, 41}), 73);
    c = HeContext::create(parms, false, nil);
    check(c->first_context_data().value()->total_coeff_modulus()[0] == 697 && qualifiers_are(c, E::Success, true, true, true, true, false, false, nil, false, false), "q = {17, 41}, t = 73: batching");
    set(4, to_moduli({137, 193}), 73);
    c = HeContext::create(parms, false, nil);
    check(c->first_context_data().value()->total_coeff_modulus()[0] == 137 && c->key_context_data().value()->total_coeff_modulus()[0] == 26441 &&
          qualifiers_are(c, E::Success, true, true, true, true, true, false, nil, true, true), "q = {137, 193}, t = 73: two levels, batching, fast plain lift");
    check(qualifiers_are(HeContext::create(parms, false, SecurityLevel::Classical128), E::InvalidParametersInsecure, false, true, false, false, false, false, nil, false, false),
          "the same under Classical128: InvalidParametersInsecure (N = 4 has no secure modulus)");
    parms.set_poly_modulus_degree(2048);
    parms.set_coeff_modulus(CoeffModulus::bfv_default(4096, SecurityLevel::Classical128));
    check(qualifiers_are(HeContext::create(parms, false, SecurityLevel::Classical128), E::InvalidParametersInsecure, false, true, false, false, false, false, nil, false, false),
          "N = 2048 with the default modulus of N = 4096: InvalidParametersInsecure");
    set(4096, to_moduli({0xffffee001, 0xffffc4001}), 73);
    check(qualifiers_are(HeContext::create(parms, false, SecurityLevel::Classical128), E::Success, true, true, true, false, true, true, SecurityLevel::Classical128, true, false),
          "N = 4096, two 36-bit primes, Classical128: Success, descending chain");
    set(2048, to_moduli({0x1ffffe0001, 0xffffee001, 0xffffc4001}), 73);
    check(qualifiers_are(HeContext::create(parms, false, nil), E::Success, true, true, true, false, true, true, nil, true, true), "N = 2048, three primes, no security level: Success");
    parms.set_poly_modulus_degree(2048);
    parms.set_coeff_modulus(CoeffModulus::create(2048, {40}));
    parms.set_plain_modulus(Modulus(65537));
    check(qualifiers_are(HeContext::create(parms, false, nil), E::Success, true, true, true, true, true, true, nil, false, false), "N = 2048, one 40-bit prime, t = 65537: batching, no key switching");
    // further refusals, each reported by its own code
    check(HeContext::create(EncryptionParameters(SchemeType::Nil), false, nil)->key_context_data().value()->qualifiers().parameter_error == E::InvalidScheme, "no scheme: InvalidScheme");
    set(6, to_moduli({17, 41}), 16);
    check(HeContext::create(parms, false, nil)->key_context_data().value()->qualifiers().parameter_error == E::InvalidPolyModulusDegreeNonPowerOfTwo, "N = 6: InvalidPolyModulusDegreeNonPowerOfTwo");
    set(1, to_moduli({17, 41}), 16);
    check(HeContext::create(parms, false, nil)->key_context_data().value()->qualifiers().parameter_error == E::InvalidPolyModulusDegree, "N = 1: InvalidPolyModulusDegree");
    set(4, to_moduli({17, 41}), 0);
    check(HeContext::create(parms, false, nil)->key_context_data().value()->qualifiers().parameter_error == E::InvalidPlainModulusBitCount, "t = 0: InvalidPlainModulusBitCount");
    {
        EncryptionParameters ck(SchemeType::CKKS);
        ck.set_poly_modulus_degree(4);
        ck.set_coeff_modulus(to_moduli({41, 137}));
        ck.set_plain_modulus(Modulus(73));
        check(HeContext::create(ck, false, nil)->key_context_data().value()->qualifiers().parameter_error == E::InvalidPlainModulusNonZero, "CKKS with a plain modulus: InvalidPlainModulusNonZero");
    }
    bool threw = false;
    set(4, to_moduli({17, 41}), 34);
    try { KeyGenerator kg(HeContext::create(parms, false, nil)); } catch (const std::exception&) { threw = true; }
    check(threw, "objects refuse to work on a context whose parameters are not set");
}

static void run_chain(SchemeType scheme) {
    const bool ckks = scheme == SchemeType::CKKS;
    const char* name = ckks ? "CKKS" : scheme == SchemeType::BGV ? "BGV" : "BFV";
    EncryptionParameters parms(scheme);
    parms.set_poly_modulus_degree(4);
    parms.set_coeff_modulus(to_moduli({41, 137, 193, 65537}));
    if (!ckks) parms.set_plain_modulus(Modulus(73));
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil);
    // BFV / BGV stop at {41, 137}: 41 alone is below t = 73.  CKKS goes down to {41}.
    const std::vector<uint64_t> totals = ckks ? std::vector<uint64_t>{71047416497ull, 1084081, 5617, 41} : std::vector<uint64_t>{71047416497ull, 1084081, 5617};
    ContextDataPointer cd = context->key_context_data().value(), prev;
    bool ok = true;
    for (size_t i = 0; i < totals.size(); i++) {
        ok = ok && cd->chain_index() == totals.size() - 1 - i && cd->total_coeff_modulus()[0] == totals[i] && cd->total_coeff_modulus().size() == 4 - i;
        if (i) { auto up = cd->prev_context_data().has_value() ? cd->prev_context_data().value().lock() : nullptr; ok = ok && up && up->parms_id() == prev->parms_id(); }
        else ok = ok && !cd->prev_context_data().has_value();
        prev = cd;
        if (i + 1 < totals.size()) { ok = ok && cd->next_context_data().has_value(); if (!ok) break; cd = cd->next_context_data().value(); }
    }
    ok = ok && !cd->next_context_data().has_value() && cd->parms_id() == context->last_parms_id();
    char label[128];
    std::snprintf(label, sizeof label, "%s chain of {41, 137, 193, 65537}: %zu levels, totals, chain_index, prev links", name, totals.size());
    check(ok, label);
    context = HeContext::create(parms, false, SecurityLevel::Nil);
    std::snprintf(label, sizeof label, "%s without expansion: key level (index 1) and first level (index 0) only", name);
    check(context->key_context_data().value()->chain_index() == 1 && context->first_context_data().value()->chain_index() == 0 && context->first_parms_id() == context->last_parms_id() &&
          context->key_context_data().value()->total_coeff_modulus()[0] == 71047416497ull && context->first_context_data().value()->total_coeff_modulus()[0] == 1084081, label);
}

static void run_constants() {
    // the level constants (context_data.h:81-111) against their definitions, with a two-word Q
    EncryptionParameters parms(SchemeType::BFV);
    parms.set_poly_modulus_degree(8192);
    parms.set_coeff_modulus(CoeffModulus::create(8192, {50, 50, 50}));
    parms.set_plain_modulus(PlainModulus::batching(8192, 20));
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Classical128);
    ContextDataPointer cd = context->first_context_data().value();
    const uint64_t t = parms.plain_modulus().value(), q0 = parms.coeff_modulus()[0].value(), q1 = parms.coeff_modulus()[1].value();
    const unsigned __int128 Q = static_cast<unsigned __int128>(q0) * q1;
    bool ok = cd->total_coeff_modulus().size() == 2 && cd->total_coeff_modulus()[0] == uint64_t(Q) && cd->total_coeff_modulus()[1] == uint64_t(Q >> 64) && cd->total_coeff_modulus_bit_count() == 100;
    ok = ok && cd->coeff_modulus_mod_plain_modulus() == uint64_t(Q % t) && cd->plain_upper_half_threshold() == (t + 1) / 2;
    ok = ok && cd->upper_half_increment()[0] == uint64_t(Q % t) % q0 && cd->upper_half_increment()[1] == uint64_t(Q % t) % q1;
    ok = ok && cd->plain_upper_half_increment()[0] == q0 - t && cd->plain_upper_half_increment()[1] == q1 - t && cd->qualifiers().using_fast_plain_lift;
    check(ok, "BFV level constants: Q (two words), Q mod t, (t + 1) / 2, q_i - t");
    EncryptionParameters ck(SchemeType::CKKS);
    ck.set_poly_modulus_degree(8192);
    ck.set_coeff_modulus(CoeffModulus::create(8192, {50, 50, 50}));
    context = HeContext::create(ck, true, SecurityLevel::Classical128);
    cd = context->first_context_data().value();
    const uint64_t c0 = ck.coeff_modulus()[0].value(), c1 = ck.coeff_modulus()[1].value();
    const unsigned __int128 QC = static_cast<unsigned __int128>(c0) * c1, half = (QC + 1) >> 1;
    const uint64_t minus_two64 = uint64_t((static_cast<unsigned __int128>(c0) - ((static_cast<unsigned __int128>(1) << 64) % c0)) % c0);
    ok = cd->plain_upper_half_threshold() == (1ull << 63) && cd->upper_half_threshold()[0] == uint64_t(half) && cd->upper_half_threshold()[1] == uint64_t(half >> 64);
    ok = ok && cd->plain_upper_half_increment()[0] == minus_two64 && cd->qualifiers().using_batching && !cd->qualifiers().using_fast_plain_lift;
    check(ok, "CKKS level constants: 2^63, (Q + 1) / 2, -2^64 mod q_i");
}

// GaloisTool known answers of the reference (test/utils/galois.cu:20-44, N = 8): elements from steps, the generating set, key indices
static void run_galois() {
    const int steps[] = {0, 1, -3, 2, -2, 3, -1};
    const size_t want[] = {15, 3, 3, 9, 9, 11, 11};
    bool ok = true;
    for (size_t i = 0; i < 7; i++) ok = ok && utils::galois_element_from_step(8, steps[i]) == want[i];
    check(ok, "galois_element_from_step at N = 8: 15, 3, 3, 9, 9, 11, 11");
    check(utils::galois_elements_all(8) == std::vector<size_t>({15, 3, 11, 9, 9}), "galois_elements_all at N = 8: {15, 3, 11, 9, 9}");
    check(GaloisKeys::get_index(15) == 7 && GaloisKeys::get_index(3) == 1 && GaloisKeys::get_index(11) == 5 && GaloisKeys::get_index(9) == 4, "GaloisKeys::get_index: 15 -> 7, 3 -> 1, 11 -> 5, 9 -> 4");
    bool threw = false;
    try { GaloisKeys::get_index(4); } catch (const std::invalid_argument&) { threw = true; }
    check(threw, "an even Galois element has no key index");
}

// troy::bench timers (src/utils/timer.h, timer.cpp): the merged view over host threads, the default divisor, reset, the byte formatter
static void run_timers() {
    std::vector<bench::Timer> per_thread(3);
    for (size_t t = 0; t < per_thread.size(); t++) {
        per_thread[t].tab(1);
        const size_t a = per_thread[t].register_timer("encrypt"), b = per_thread[t].register_timer(t == 2 ? "only-here" : "multiply");
        for (size_t rep = 0; rep <= t; rep++) { per_thread[t].tick(a); per_thread[t].tock(a); }
        per_thread[t].tick(b); per_thread[t].tock(b);
    }
    std::stringstream captured;
    std::streambuf* old = std::cout.rdbuf(captured.rdbuf());
    bench::TimerThreaded::Print(per_thread);
    bench::TimerThreaded::PrintDivided(per_thread, 10);
    per_thread[2].print_divided();                       // the default divisor is the number of tick / tock pairs (3 for "encrypt")
    bench::print_communication("ciphertexts", 1, 3 * 1024 * 1024, 3);
    bench::print_communication(512); std::cout << std::endl;
    std::cout.rdbuf(old);
    const std::string text = captured.str();
    size_t lines = 0;
    for (char c : text) lines += c == '\n';
    const bool merged = text.find("  encrypt") != std::string::npos && text.find("  multiply") != std::string::npos && text.find("  only-here") != std::string::npos &&
                        text.find("/ thread, avg") != std::string::npos && text.find("/ op (total max") != std::string::npos && text.find(", 10 times)") != std::string::npos;
    const bool divisor = text.find(", 3 times)") != std::string::npos && text.find("1.000 MB (total     3.000 MB, 3 times)") != std::string::npos && text.find("512 B") != std::string::npos;
    check(merged && lines == 3 + 3 + 2 + 1 + 1, "TimerThreaded: one line per distinct name, max and mean over the threads that carry it");
    check(divisor, "print_divided() divides by the tick count; print_communication formats bytes");
    per_thread[2].reset();
    check(per_thread[2].get()[0] == bench::Duration(0) && per_thread[2].names().size() == 2, "Timer::reset clears the accumulations and keeps the names");
}

int main(int argc, char** argv) {
    try {
        const bool device = argc > 1 && !std::strcmp(argv[1], "device");
        run_construct();
        run_chain(SchemeType::BFV);
        run_chain(SchemeType::BGV);
        run_chain(SchemeType::CKKS);
        run_constants();
        run_timers();
        run_galois();
        if (device) {                                  // he_context.cu:323-360
            for (SchemeType scheme : {SchemeType::BFV, SchemeType::BGV, SchemeType::CKKS}) {
                EncryptionParameters parms(scheme);
                parms.set_poly_modulus_degree(4);
                parms.set_coeff_modulus(to_moduli({41, 137, 193, 65537}));
                if (scheme != SchemeType::CKKS) parms.set_plain_modulus(Modulus(73));
                HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil);
                context->to_device_inplace();
                check(context->on_device(), "to_device_inplace of the N = 4 chain");
            }
            MemoryPool::Destroy();
        }
        std::printf(failures ? "FAIL\n" : "OK\n");
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
