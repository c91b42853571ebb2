// Runner of tests/_ref_tests/ref_tests: the reference's own test sources linked against this repository's mirror (tests/build_ref_tests.sh).
//   ref_tests [filter]      runs the tests whose "Suite.Name" contains `filter` (default "Device": the mirror has no host path), prints one line per test
//   ref_tests --list        the names
#include <cstring>
#include <exception>
#include <iostream>

#include "gtest/gtest.h"

int main(int argc, char** argv) {
    const bool list = argc > 1 && !std::strcmp(argv[1], "--list");
    const std::string filter = (argc > 1 && !list) ? argv[1] : "Device";
    int ran = 0, failed = 0, skipped = 0;
    for (const auto& c : mini_gtest::registry()) {
        if (list) { std::cout << c.name << std::endl; continue; }
        if (c.name.find(filter) == std::string::npos) continue;
        mini_gtest::state() = mini_gtest::State();
        try { c.fn(); }
        catch (const std::exception& e) { mini_gtest::state().failures++; std::cout << "exception: " << e.what() << std::endl; }
        ran++;
        const char* verdict = mini_gtest::state().failures ? "FAILED" : mini_gtest::state().skipped ? "SKIPPED" : "OK";
        failed += mini_gtest::state().failures != 0;
        skipped += mini_gtest::state().skipped && !mini_gtest::state().failures;
        std::cout << "[ " << verdict << " ] " << c.name << std::endl;
    }
    if (!list) std::cout << "ran " << ran << " failed " << failed << " skipped " << skipped << std::endl;
    return failed ? 1 : 0;
}
