// bench_timer.h -- troy::bench timers with the interface of the reference's src/utils/timer.h (TimerOnce, TimerSingle, Timer): wall-clock
// accumulation around host calls.  Device work of this mirror is synchronised before each public call returns, so tick()/tock()
// around a call measures the call.
#pragma once
#include <chrono>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

namespace troy { namespace bench {

using Instant = std::chrono::time_point<std::chrono::high_resolution_clock>;
using Duration = std::chrono::nanoseconds;
const size_t PROMPT_LENGTH = 20;

inline void print_duration(size_t nanoseconds) {
    std::cout << std::right << std::setw(9) << std::setprecision(3) << std::fixed;
    if (nanoseconds < 1000) std::cout << nanoseconds << " ns";
    else if (nanoseconds < 1000000ull) std::cout << nanoseconds / 1e3 << " us";
    else if (nanoseconds < 1000000000ull) std::cout << nanoseconds / 1e6 << " ms";
    else std::cout << nanoseconds / 1e9 << " s ";
}

inline void print_duration(const std::string& prompt, size_t tabs, const Duration& duration, size_t divide) {
    for (size_t i = 0; i < tabs; i++) std::cout << "  ";
    std::cout << prompt;
    const size_t used = prompt.length() + tabs * 2;
    for (size_t i = used; i < PROMPT_LENGTH; i++) std::cout << " ";
    std::cout << ": ";
    print_duration(static_cast<size_t>(duration.count()) / (divide ? divide : 1));
    if (divide > 1) { std::cout << " (total "; print_duration(static_cast<size_t>(duration.count())); std::cout << ", " << divide << " times)"; }
    std::cout << std::endl;
}

class TimerOnce {
public:
    TimerOnce() : start_(std::chrono::high_resolution_clock::now()) {}
    TimerOnce& tab(size_t tabs) { tabs_ = tabs; return *this; }
    void finish(const std::string& prompt) { print_duration(prompt, tabs_, get_finish(), 1); }
    Duration get_finish() const { return std::chrono::duration_cast<Duration>(std::chrono::high_resolution_clock::now() - start_); }
    void restart() { start_ = std::chrono::high_resolution_clock::now(); }
private:
    Instant start_;
    size_t tabs_ = 0;
};

class TimerSingle {
public:
    TimerSingle& tab(size_t tabs) { tabs_ = tabs; return *this; }
    void tick() { last_ = std::chrono::high_resolution_clock::now(); }
    void tock() { accumulated_ += std::chrono::duration_cast<Duration>(std::chrono::high_resolution_clock::now() - last_); count_++; }
    void print(const std::string& name) const { print_duration(name, tabs_, accumulated_, count_); }
    void print_divided(const std::string& name, size_t divide) const { print_duration(name, tabs_, accumulated_, divide); }
    Duration get() const { return accumulated_; }
    size_t count() const { return count_; }
    void clear() { accumulated_ = Duration(0); count_ = 0; }
private:
    Instant last_;
    Duration accumulated_{0};
    size_t tabs_ = 0, count_ = 0;
};

class Timer {
public:
    Timer& tab(size_t tabs) { tabs_ = tabs; for (auto& t : timers_) t.tab(tabs); return *this; }
    size_t register_timer(const std::string& name) { names_.push_back(name); timers_.emplace_back(); timers_.back().tab(tabs_); return timers_.size() - 1; }
    void tick(size_t handle = 0) { timers_.at(handle).tick(); }
    void tock(size_t handle = 0) { timers_.at(handle).tock(); }
    void clear() { names_.clear(); timers_.clear(); }
    void print() const { for (size_t i = 0; i < timers_.size(); i++) timers_[i].print(names_[i]); }
    void print_divided(size_t divide) const { for (size_t i = 0; i < timers_.size(); i++) timers_[i].print_divided(names_[i], divide); }
    std::vector<Duration> get() const { std::vector<Duration> r; for (const auto& t : timers_) r.push_back(t.get()); return r; }
private:
    std::vector<std::string> names_;
    std::vector<TimerSingle> timers_;
    size_t tabs_ = 0;
};

}}  // namespace troy::bench
