// A few dozen lines standing where <gtest/gtest.h> would be (googletest is an un-vendored submodule of the reference and absent from this image): just enough
// of its macro surface to RUN the reference's own test sources against this repository's mirror (tests/build_ref_tests.sh).  Written for this repository; the
// semantics follow googletest's documented behaviour: TEST registers a void function; ASSERT_* records a failure and returns from the CURRENT function; EXPECT_*
// records and continues; GTEST_SKIP() marks the test skipped and returns.  The runner (ref_tests_main.cpp) runs the tests whose full name contains a filter string.
#pragma once
#include <cmath>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace mini_gtest {
struct Case { std::string name; void (*fn)(); };
inline std::vector<Case>& registry() { static std::vector<Case> r; return r; }
struct State { int failures = 0; bool skipped = false; };
inline State& state() { static State s; return s; }
struct Registrar { Registrar(const char* suite, const char* name, void (*fn)()) { registry().push_back({std::string(suite) + "." + name, fn}); } };
// swallows what the test streams after a macro (ASSERT_TRUE(x) << "message";)
struct Sink { template <typename T> Sink& operator<<(const T&) { return *this; } Sink& operator<<(std::ostream& (*)(std::ostream&)) { return *this; } };
inline void fail(const char* file, int line, const char* what) { state().failures++; std::cout << file << ":" << line << ": Failure: " << what << std::endl; }
// `return Voidify() & (sink << ...)`: lets a macro both return from a void function and accept a streamed message
struct Voidify { void operator&(const Sink&) const {} };
}  // namespace mini_gtest

#define TEST(suite, name) \
    static void suite##_##name##_body(); \
    static ::mini_gtest::Registrar suite##_##name##_registrar(#suite, #name, &suite##_##name##_body); \
    static void suite##_##name##_body()

#define MINI_GTEST_ASSERT_(cond, text) \
    if (cond) {} else return ::mini_gtest::fail(__FILE__, __LINE__, text), ::mini_gtest::Voidify() & ::mini_gtest::Sink()
#define MINI_GTEST_EXPECT_(cond, text) \
    if (cond) {} else ::mini_gtest::fail(__FILE__, __LINE__, text), ::mini_gtest::Sink()

#define ASSERT_TRUE(x) MINI_GTEST_ASSERT_(static_cast<bool>(x), "ASSERT_TRUE(" #x ")")
#define ASSERT_FALSE(x) MINI_GTEST_ASSERT_(!static_cast<bool>(x), "ASSERT_FALSE(" #x ")")
#define ASSERT_EQ(a, b) MINI_GTEST_ASSERT_((a) == (b), "ASSERT_EQ(" #a ", " #b ")")
#define ASSERT_NE(a, b) MINI_GTEST_ASSERT_(!((a) == (b)), "ASSERT_NE(" #a ", " #b ")")
#define ASSERT_LT(a, b) MINI_GTEST_ASSERT_((a) < (b), "ASSERT_LT(" #a ", " #b ")")
#define ASSERT_LE(a, b) MINI_GTEST_ASSERT_((a) <= (b), "ASSERT_LE(" #a ", " #b ")")
#define ASSERT_GT(a, b) MINI_GTEST_ASSERT_((a) > (b), "ASSERT_GT(" #a ", " #b ")")
#define ASSERT_GE(a, b) MINI_GTEST_ASSERT_((a) >= (b), "ASSERT_GE(" #a ", " #b ")")
#define ASSERT_NEAR(a, b, tol) MINI_GTEST_ASSERT_(std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol), "ASSERT_NEAR(" #a ", " #b ", " #tol ")")
#define EXPECT_TRUE(x) MINI_GTEST_EXPECT_(static_cast<bool>(x), "EXPECT_TRUE(" #x ")")
#define EXPECT_FALSE(x) MINI_GTEST_EXPECT_(!static_cast<bool>(x), "EXPECT_FALSE(" #x ")")
#define EXPECT_EQ(a, b) MINI_GTEST_EXPECT_((a) == (b), "EXPECT_EQ(" #a ", " #b ")")
#define EXPECT_NE(a, b) MINI_GTEST_EXPECT_(!((a) == (b)), "EXPECT_NE(" #a ", " #b ")")
#define EXPECT_NEAR(a, b, tol) MINI_GTEST_EXPECT_(std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol), "EXPECT_NEAR(" #a ", " #b ", " #tol ")")
#define GTEST_SKIP() return (::mini_gtest::state().skipped = true), ::mini_gtest::Voidify() & ::mini_gtest::Sink()
#define GTEST_SKIP_(message) GTEST_SKIP()
