// Waiting for a collective with a deadline (libarkblst_amd_rccl.so, multi_rccl.hip): host-only, no HIP or RCCL in here, so that
// tests/host/workers_test.cpp can run it under ThreadSanitizer.  ncclAllGather followed by hipStreamSynchronize waits for ever when a
// peer never arrives (a rank that crashed, or one that returned early from the call); the exchange instead polls
//   done()    hipStreamQuery(stream) == hipSuccess            -> the gathered block has landed
//   failed()  ncclCommGetAsyncError(comm) != ncclSuccess, or a stream error   -> the communicator is broken
// against a deadline, and the caller aborts the communicator (ncclCommAbort) when it passes.
#pragma once
#include <chrono>
#include <thread>

namespace mi {

enum class WaitResult { Done, Failed, TimedOut };

// Polls until done() is true (Done), failed() is true (Failed) or timeout_ms have passed (TimedOut; timeout_ms <= 0 waits for ever).
// The first 200 us are a busy poll (the exchange of a few KB takes tens of microseconds), then the thread sleeps 50 us .. 1 ms between polls.
template <class Done, class Failed>
WaitResult wait_deadline(Done done, Failed failed, double timeout_ms, double* waited_ms = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    WaitResult r;
    double nap_us = 50;
    for (;;) {
        if (done()) { r = WaitResult::Done; break; }
        if (failed()) { r = WaitResult::Failed; break; }
        const double us = elapsed_us();
        if (timeout_ms > 0 && us > timeout_ms * 1e3) { r = WaitResult::TimedOut; break; }
        if (us > 200) {
            std::this_thread::sleep_for(std::chrono::microseconds((long)nap_us));
            nap_us = nap_us < 1000 ? nap_us * 1.5 : 1000;
        }
    }
    if (waited_ms) *waited_ms = elapsed_us() * 1e-3;
    return r;
}

}  // namespace mi
