// Small persistent worker pool for the host tail of an MSM (per-window chunk combine).
//   * Spawning threads per call cost more than the work itself (measured 1.4 ms for ~0.2 ms of arithmetic), so the
//     context keeps workers parked on a condition variable and the calling thread takes part in the loop.
//   * Waking parked threads costs tens to hundreds of microseconds on a busy host, so the driver calls prewake()
//     when it launches the last GPU kernel: the workers then spin (bounded) until the work is published.
// Work distribution is a single 64-bit ticket (generation << 32 | next index): a worker that was descheduled across
// two parallel_for calls simply adopts whatever generation its ticket belongs to.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hostpool {

class Pool {
public:
    explicit Pool(unsigned workers) {
        for (unsigned i = 0; i < workers; i++) threads_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_.store(true);
            wake_++;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;

    // Hint that a parallel_for follows within ~a millisecond: parked workers wake up and spin for it.
    void prewake() {
        if (threads_.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            wake_++;
        }
        cv_.notify_all();
    }

    // runs fn(i) for i in [0, n); returns when all are done.  One parallel_for at a time (callers hold the ctx lock).
    void parallel_for(unsigned n, const std::function<void(unsigned)>& fn) {
        if (n == 0) return;
        if (threads_.empty() || n == 1) {
            for (unsigned i = 0; i < n; i++) fn(i);
            return;
        }
        // A worker descheduled inside run_items() since an earlier generation could still read the descriptor slot
        // that is about to be rewritten: wait until nobody is inside (normally immediate).
        while (inside_.load(std::memory_order_acquire) != 0) cpu_relax();
        uint64_t gen = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
        Desc& d = desc_[gen & 1];
        d.fn = &fn;
        d.n = n;
        done_.store(0, std::memory_order_relaxed);
        ticket_.store(gen << 32, std::memory_order_release);   // publishes the descriptor
        prewake();                                              // in case nobody is spinning yet
        run_items();
        while (done_.load(std::memory_order_acquire) < n) cpu_relax();
    }

    // Split form of parallel_for: start() publishes the loop and returns at once (the workers run it), finish() lets the
    // caller help with what is left and waits for the end.  Between the two the caller may do work that consumes the
    // loop's results as they appear (the Horner fold over window sums).  False from start(): no workers — run it inline.
    bool start(unsigned n, const std::function<void(unsigned)>& fn) {
        if (threads_.empty() || n == 0) return false;
        while (inside_.load(std::memory_order_acquire) != 0) cpu_relax();
        uint64_t gen = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
        Desc& d = desc_[gen & 1];
        d.fn = &fn;
        d.n = n;
        done_.store(0, std::memory_order_relaxed);
        ticket_.store(gen << 32, std::memory_order_release);
        prewake();
        return true;
    }
    void finish(unsigned n) {
        run_items();
        while (done_.load(std::memory_order_acquire) < n) cpu_relax();
    }

private:
    struct Desc {
        const std::function<void(unsigned)>* fn = nullptr;
        unsigned n = 0;
    };
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    // returns true if it executed at least one item
    bool run_items() {
        bool any = false;
        inside_.fetch_add(1, std::memory_order_acq_rel);
        for (;;) {
            uint64_t t = ticket_.fetch_add(1, std::memory_order_acq_rel);
            uint64_t gen = t >> 32;
            unsigned idx = (unsigned)(t & 0xffffffffu);
            if (gen == 0) break;
            const Desc& d = desc_[gen & 1];
            if (idx >= d.n) break;
            (*d.fn)(idx);
            done_.fetch_add(1, std::memory_order_release);
            any = true;
        }
        inside_.fetch_sub(1, std::memory_order_acq_rel);
        return any;
    }
    bool work_available() const {
        uint64_t t = ticket_.load(std::memory_order_acquire);
        uint64_t gen = t >> 32;
        return gen != 0 && (unsigned)(t & 0xffffffffu) < desc_[gen & 1].n;
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return wake_ != seen; });
                seen = wake_;
            }
            if (stop_.load()) return;
            // spin (bounded) for the announced work
            auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                if (work_available()) {
                    run_items();
                    t0 = std::chrono::steady_clock::now();  // stay warm briefly: a second stage may follow at once
                    continue;
                }
                if (stop_.load()) return;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500)) break;
                cpu_relax();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    Desc desc_[2];
    std::atomic<uint64_t> ticket_{0};
    std::atomic<unsigned> done_{0}, inside_{0};
    unsigned long wake_ = 0;
    std::atomic<bool> stop_{false};
};

}  // namespace hostpool
