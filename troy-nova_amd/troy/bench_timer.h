// bench_timer.h -- troy::bench timers with the interface of the reference's src/utils/timer.h (TimerOnce, TimerSingle, Timer, TimerThreaded, print_communication): wall-clock
// accumulation around host calls.  Device work of this mirror is synchronised before each public call returns, so tick()/tock()
// around a call measures the call.
#pragma once
#include <algorithm>
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

// timer.h:64-108: a byte count in B / KB / MB / GB, alone or as a labelled line (per-repetition figure first when `divide` > 1)
inline void print_communication(size_t bytes) {
    static const char* const units[] = {" B", " KB", " MB", " GB"};
    std::cout << std::right << std::setw(9) << std::setprecision(3) << std::fixed;
    if (bytes < 1024) { std::cout << bytes << units[0]; return; }
    double v = static_cast<double>(bytes);
    size_t u = 0;
    while (u < 3 && v >= 1024.0) { v /= 1024.0; u++; }
    std::cout << v << units[u];
}
inline void print_communication(const std::string& prompt, size_t tabs, size_t bytes, size_t divide) {
    for (size_t i = 0; i < tabs; i++) std::cout << "  ";
    std::cout << prompt;
    for (size_t i = prompt.length() + tabs * 2; i < PROMPT_LENGTH; i++) std::cout << " ";
    std::cout << ": ";
    print_communication(bytes / (divide ? divide : 1));
    if (divide > 1) { std::cout << " (total "; print_communication(bytes); std::cout << ", " << divide << " times)"; }
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
    void print_divided(const std::string& name, size_t divide_override = 0) const { print_duration(name, tabs_, accumulated_, divide_override ? divide_override : count_); }   // timer.h:167-170
    Duration get() const { return accumulated_; }
    size_t count() const { return count_; }
    void clear() { accumulated_ = Duration(0); count_ = 0; }
    void reset() { clear(); last_ = std::chrono::high_resolution_clock::now(); }                      // timer.h:174-178
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
    void print_divided(size_t divide_override = 0) const { for (size_t i = 0; i < timers_.size(); i++) timers_[i].print_divided(names_[i], divide_override); }
    void reset() { for (auto& t : timers_) t.reset(); }                                                // timer.h:230-234: the names stay
    const std::vector<std::string>& names() const { return names_; }
    size_t tabs() const { return tabs_; }
    std::vector<Duration> get() const { std::vector<Duration> r; for (const auto& t : timers_) r.push_back(t.get()); return r; }
private:
    std::vector<std::string> names_;
    std::vector<TimerSingle> timers_;
    size_t tabs_ = 0;
};

// timer.h:237-253, timer.cpp: the timers of N host threads merged by name -- the slowest thread and the mean over the threads that carry the name
class TimerThreaded {
public:
    explicit TimerThreaded(const std::vector<Timer>& timers) {
        std::vector<size_t> seen;
        for (const Timer& t : timers) {
            tabs_ = std::max(tabs_, t.tabs());
            const std::vector<Duration> d = t.get();
            for (size_t i = 0; i < d.size(); i++) {
                size_t at = 0;
                while (at < names_.size() && names_[at] != t.names()[i]) at++;
                if (at == names_.size()) { names_.push_back(t.names()[i]); max_.push_back(d[i]); sum_.push_back(Duration::zero()); seen.push_back(0); }
                sum_[at] += d[i];
                seen[at]++;
                if (d[i] > max_[at]) max_[at] = d[i];
            }
        }
        for (size_t i = 0; i < sum_.size(); i++) sum_[i] /= static_cast<long>(seen[i]);       // now the mean
    }
    void print() const { print_divided(1); }
    void print_divided(size_t divide) const {
        for (size_t i = 0; i < names_.size(); i++) {
            for (size_t k = 0; k < tabs_; k++) std::cout << "  ";
            std::cout << names_[i];
            for (size_t k = names_[i].length() + tabs_ * 2; k < PROMPT_LENGTH; k++) std::cout << " ";
            const size_t mx = static_cast<size_t>(max_[i].count()), mean = static_cast<size_t>(sum_[i].count());
            std::cout << ": max ";
            if (divide <= 1) {
                print_duration(mx); std::cout << " / thread, avg "; print_duration(mean); std::cout << " / thread ";
            } else {
                print_duration(mx / divide); std::cout << " / op, avg "; print_duration(mean / divide); std::cout << " / op (total max ";
                print_duration(mx); std::cout << " / thread, avg "; print_duration(mean); std::cout << " / thread, " << divide << " times)";
            }
            std::cout << std::endl;
        }
    }
    static void Print(const std::vector<Timer>& timers) { TimerThreaded(timers).print(); }
    static void PrintDivided(const std::vector<Timer>& timers, size_t divide) { TimerThreaded(timers).print_divided(divide); }
private:
    std::vector<std::string> names_;
    std::vector<Duration> max_, sum_;
    size_t tabs_ = 0;
};

}}  // namespace troy::bench
