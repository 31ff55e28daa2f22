// Small persistent worker pool for the host tail of an MSM (per-window chunk combine).  Spawning threads per call
// cost more than the work itself (measured 1.4 ms for ~0.2 ms of arithmetic), so the context keeps workers parked
// on a condition variable; the calling thread takes part in the loop.
#pragma once
#include <atomic>
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
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;

    // runs fn(i) for i in [0, n); returns when all are done.  One parallel_for at a time (callers hold the ctx lock).
    void parallel_for(unsigned n, const std::function<void(unsigned)>& fn) {
        if (n == 0) return;
        if (threads_.empty() || n == 1) {
            for (unsigned i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            done_.store(0, std::memory_order_relaxed);
            gen_++;
        }
        cv_.notify_all();
        run_items();
        // wait for stragglers: short spin, then yield
        while (done_.load(std::memory_order_acquire) < n_) std::this_thread::yield();
        std::lock_guard<std::mutex> lk(mu_);
        fn_ = nullptr;
    }

private:
    void run_items() {
        for (;;) {
            unsigned i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) break;
            (*fn_)(i);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (!fn_) continue;
                active_++;
            }
            run_items();
            {
                std::lock_guard<std::mutex> lk(mu_);
                active_--;
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned n_ = 0;
    std::atomic<unsigned> next_{0}, done_{0};
    unsigned long gen_ = 0;
    unsigned active_ = 0;
    bool stop_ = false;
};

}  // namespace hostpool
