// Host-only test of the library's threading and error plumbing (ark-blst_amd/csrc/common.hpp): the persistent per-device
// workers, the lane lock, and the rule that nothing crosses the C ABI as an exception — an allocation failure inside a worker
// thread comes back as MI_E_NOMEM instead of std::terminate.  Needs no GPU: without a device hipMalloc itself fails (MI_E_HIP),
// and the test build's hook injects the out-of-memory case.  Built three times by tests/test_host_threads.py: plain,
// -fsanitize=thread, -fsanitize=address,undefined.
#define MI_TEST_HOOKS 1
#include "../../ark-blst_amd/csrc/common.hpp"
#include "../../ark-blst_amd/csrc/deadline.hpp"

#include <cassert>
#include <stdexcept>

namespace mi { std::atomic<int> g_fail_allocs{0}; }
using namespace mi;

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); return 1; } } while (0)

int main() {
    // 1. persistent workers: every job runs exactly once per round on its own thread, rounds do not overlap
    {
        const size_t G = 8;
        DeviceWorkers w(G);
        std::vector<long> count(G, 0);
        std::atomic<int> inside{0};
        std::atomic<bool> overlap{false};
        for (int round = 0; round < 2000; round++) {
            std::function<void(size_t)> fn = [&](size_t k) {
                inside.fetch_add(1);
                count[k]++;   // slot k is only ever touched by worker k inside a round, and by this thread between rounds
                if (inside.load() > (int)G) overlap = true;
                inside.fetch_sub(1);
            };
            w.run(fn);
            for (size_t k = 0; k < G; k++) CHECK(count[k] == round + 1);
        }
        CHECK(!overlap.load());
    }
    // 2. failures inside workers become codes: injected out-of-memory -> MI_E_NOMEM, bad_alloc -> MI_E_NOMEM, anything else -> MI_E_HIP
    {
        const size_t G = 6;
        DeviceWorkers w(G);
        for (int rep = 0; rep < 50; rep++) {
            std::vector<PartErr> errs(G);
            g_fail_allocs.store(3);
            std::function<void(size_t)> fn = [&](size_t k) {
                guarded_part(errs[k], [&] {
                    DevBuf b;
                    b.ensure(1 << 20);   // injected failure for three of the workers; the others reach hipMalloc
                    b.release();
                });
            };
            w.run(fn);
            int nomem = 0;
            for (auto& e : errs) {
                if (e.code == MI_E_NOMEM) { nomem++; CHECK(e.msg.find("out of memory") != std::string::npos); }
                else CHECK(e.code == MI_OK || e.code == MI_E_HIP);   // OK with a GPU, "no device" without
            }
            CHECK(nomem == 3);
            g_fail_allocs.store(0);
            std::vector<PartErr> e2(G);
            std::function<void(size_t)> fn2 = [&](size_t k) {
                guarded_part(e2[k], [&] {
                    if (k == 0) throw std::bad_alloc();
                    if (k == 1) throw std::runtime_error("boom");
                    if (k == 2) throw 42;
                    if (k == 3) throw HipFail{"hipFoo failed: out of memory", false};
                });
            };
            w.run(fn2);
            CHECK(e2[0].code == MI_E_NOMEM && e2[1].code == MI_E_HIP && e2[2].code == MI_E_HIP && e2[3].code == MI_E_NOMEM && e2[4].code == MI_OK);
            CHECK(e2[1].msg.find("boom") != std::string::npos);
        }
    }
    // 3. guarded(): the same mapping on the calling thread, text kept per thread and per context
    {
        mi_ctx ctx;
        CHECK(guarded(&ctx, []() -> int { throw std::bad_alloc(); }) == MI_E_NOMEM);
        CHECK(guarded(&ctx, []() -> int { throw HipFail{"x failed", true}; }) == MI_E_NOMEM);
        CHECK(guarded(&ctx, []() -> int { throw HipFail{"y failed", false}; }) == MI_E_HIP);
        CHECK(tls_error() == "y failed" && ctx.err == "y failed");
        CHECK(guarded(&ctx, []() -> int { return MI_OK; }) == MI_OK);
        std::thread t([&] { (void)guarded(&ctx, []() -> int { throw HipFail{"from another thread", false}; }); });
        t.join();
        CHECK(tls_error() == "y failed");   // this thread's text is its own
    }
    // 4. lane lock: at most two shared holders, an exclusive holder is alone
    {
        mi_ctx ctx;
        std::atomic<int> shared_in{0}, excl_in{0}, bad{0};
        auto body = [&](int t) {
            for (int i = 0; i < 3000; i++) {
                if ((i + t) % 7 == 0) {
                    LaneLock lk(&ctx, true);
                    if (excl_in.fetch_add(1) != 0 || shared_in.load() != 0) bad++;
                    excl_in.fetch_sub(1);
                } else {
                    LaneLock lk(&ctx, false);
                    int s = shared_in.fetch_add(1) + 1;
                    if (s > NLANES || excl_in.load() != 0 || lk.lane < 0 || lk.lane >= NLANES) bad++;
                    shared_in.fetch_sub(1);
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < 6; t++) th.emplace_back(body, t);
        for (auto& t : th) t.join();
        CHECK(bad.load() == 0);
    }
    // 5. shards cover [0, n) exactly once
    for (size_t n : {0ul, 1ul, 7ul, 1000ul, 16777216ul}) {
        for (size_t g : {1ul, 2ul, 3ul, 8ul}) {
            size_t next = 0;
            for (size_t k = 0; k < g; k++) {
                size_t lo, hi;
                shard_range(n, g, k, lo, hi);
                CHECK(lo == std::min(next, n) && hi >= lo);
                next = hi;
            }
            CHECK(next == n);
        }
    }
    // 6. base-set cache: the fingerprint (round 6: EVERY byte) and the bookkeeping that msm_impl drives from two lanes at once
    {
        // equal content -> equal value; the length and ANY single byte change it (a change inside one 8-byte word: with certainty)
        const size_t n = 50000, aff = 96;
        std::vector<uint8_t> a(n * aff), b;
        for (size_t i = 0; i < a.size(); i++) a[i] = (uint8_t)(i * 131 + (i >> 7));
        b = a;
        for (size_t threads : {1ul, 3ul, 6ul}) {
            const Fp128 fa = content_fingerprint(a.data(), n * aff, threads);
            CHECK(fa == content_fingerprint(b.data(), n * aff, threads));
            CHECK(fa != content_fingerprint(a.data(), (n - 1) * aff, threads));
            // one bit of one limb of points all over the vector, sampled by the old fingerprint or not (index 1 of 50000 was not)
            for (size_t i : {0ul, 1ul, 2ul, 777ul, 24999ul, 25000ul, 33333ul, 49998ul, 49999ul}) {
                for (size_t off : {0ul, 47ul, 48ul, 95ul}) {
                    b[i * aff + off] ^= 0x10;
                    CHECK(fa != content_fingerprint(b.data(), n * aff, threads));
                    b[i * aff + off] ^= 0x10;
                }
            }
            CHECK(fa == content_fingerprint(b.data(), n * aff, threads));
        }
        // two points exchanged: the same multiset of words in other positions
        memcpy(&b[10 * aff], &a[20 * aff], aff); memcpy(&b[20 * aff], &a[10 * aff], aff);
        CHECK(content_fingerprint(a.data(), n * aff, 4) != content_fingerprint(b.data(), n * aff, 4));
        b = a;
        CHECK(content_fingerprint(a.data(), aff, 4) != content_fingerprint(a.data() + aff, aff, 4));
        std::vector<uint8_t> odd(1000 * 192 + 13, 7);   // a length that is not a multiple of 32: the tail is read, not overrun (ASan)
        const Fp128 fo = content_fingerprint(odd.data(), odd.size(), 4);
        odd.back() ^= 1;
        CHECK(fo != content_fingerprint(odd.data(), odd.size(), 4));
        // the persistent helper pool computes the same value, job after job, from whichever thread holds it (TSan: no race on its state)
        {
            HashPool pool(5);
            std::vector<uint8_t> big(((size_t)7 << 20) + 96 * 5, 0);
            for (size_t i = 0; i < big.size(); i++) big[i] = (uint8_t)(i * 2654435761u >> 13);
            for (int round = 0; round < 40; round++) {
                const size_t bytes = round % 3 == 0 ? big.size() : round % 3 == 1 ? (size_t)96 * 4096 : n * aff;
                const uint8_t* src = round % 3 == 2 ? a.data() : big.data();
                CHECK(!pool.busy());
                pool.start(src, bytes);
                CHECK(pool.busy());
                const Fp128 got = pool.finish();
                CHECK(got == content_fingerprint(src, bytes, pool.threads()));
                big[(size_t)round * 1000] ^= 1;   // the caller may rewrite the vector between two jobs
            }
            std::thread other([&] { pool.start(a.data(), n * aff); (void)pool.finish(); });   // handed to another thread between jobs
            other.join();
        }
        // bookkeeping under contention: four threads, three base vectors, a two-entry cache, entries without device memory
        mi_ctx ctx;
        ctx.cache_entries = 2;
        std::vector<std::vector<uint8_t>> sets(3, std::vector<uint8_t>(8192 * aff));
        for (size_t k = 0; k < 3; k++)
            for (size_t i = 0; i < sets[k].size(); i++) sets[k][i] = (uint8_t)(i * (k + 3) + k);
        std::atomic<int> bad{0};
        auto body = [&](int t) {
            for (int i = 0; i < 4000; i++) {
                const size_t k = (size_t)((i * 7 + t) % 3);
                const uint8_t* ptr = sets[k].data();
                std::shared_ptr<BaseCacheEntry> hit, fill;
                if (!cache_begin(&ctx, 0, ptr, 8192, hit)) { bad++; continue; }
                const Fp128 fp = content_fingerprint(ptr, 8192 * aff, 4);
                if (hit && hit->fp != fp) hit = cache_find(&ctx, 0, ptr, 8192, fp);
                if (hit && (hit->ptr != ptr || hit->fp != fp)) bad++;
                if (!hit) {
                    fill = std::make_shared<BaseCacheEntry>();
                    fill->ptr = ptr; fill->n = 8192;
                    fill->shard.resize(1);
                    fill->devs.push_back(0);
                }
                cache_finish(&ctx, 0, hit, fill, fp);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < 4; t++) th.emplace_back(body, t);
        for (auto& t : th) t.join();
        CHECK(bad.load() == 0);
        CHECK(ctx.cache[0].size() <= 2 && ctx.cache_hits + ctx.cache_misses == 16000 && ctx.cache_misses >= 3);
        for (size_t x = 0; x < ctx.cache[0].size(); x++)
            for (size_t y = x + 1; y < ctx.cache[0].size(); y++) CHECK(ctx.cache[0][x]->ptr != ctx.cache[0][y]->ptr);   // no duplicate keys
        // an in-place rewrite: the candidate is found by (pointer, n) but its fingerprint no longer matches
        sets[0][5] ^= 0xff;
        std::shared_ptr<BaseCacheEntry> cand;
        CHECK(cache_begin(&ctx, 0, sets[0].data(), 8192, cand));
        if (cand) CHECK(cand->fp != content_fingerprint(sets[0].data(), 8192 * aff, 4));
        ctx.cache_entries = 0;
        std::shared_ptr<BaseCacheEntry> none;
        CHECK(!cache_begin(&ctx, 0, sets[1].data(), 8192, none));
    }
    // 7. the exchange's wait with a deadline (multi_rccl.hip): a collective whose peer never arrives ends in TimedOut, an asynchronous error in
    //    Failed, a completion signalled from another thread in Done — never in a hang
    {
        double waited = 0;
        std::atomic<bool> never{false};
        auto t0 = std::chrono::steady_clock::now();
        CHECK(wait_deadline([&] { return never.load(); }, [] { return false; }, 30.0, &waited) == WaitResult::TimedOut);   // the never-signalled event
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CHECK(waited >= 30.0 && wall >= 30.0 && wall < 500.0);
        std::atomic<int> polls{0};
        CHECK(wait_deadline([&] { return false; }, [&] { return ++polls >= 5; }, 1000.0, &waited) == WaitResult::Failed && waited < 500.0);
        std::atomic<bool> flag{false};
        std::thread setter([&] { std::this_thread::sleep_for(std::chrono::milliseconds(5)); flag.store(true); });
        CHECK(wait_deadline([&] { return flag.load(); }, [] { return false; }, 2000.0, &waited) == WaitResult::Done);
        setter.join();
        CHECK(waited >= 4.0 && waited < 1000.0);
        CHECK(wait_deadline([] { return true; }, [] { return true; }, 1.0) == WaitResult::Done);        // completion wins over a late error
        std::atomic<bool> late{false};
        std::thread setter2([&] { std::this_thread::sleep_for(std::chrono::milliseconds(20)); late.store(true); });
        CHECK(wait_deadline([&] { return late.load(); }, [] { return false; }, 0.0) == WaitResult::Done);   // 0 = no deadline
        setter2.join();
    }
    printf("workers OK\n");
    return 0;
}
