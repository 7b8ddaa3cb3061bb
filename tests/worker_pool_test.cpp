// Stress test of stark_verifier_amd/csrc/worker_pool.hpp (pure C++; built and run by tests/test_cabi_and_host.py).
#include <atomic>
#include <cstdio>
#include <numeric>
#include <thread>
#include <vector>

#include "../stark_verifier_amd/csrc/worker_pool.hpp"

int main() {
    // every index exactly once, for task counts around the worker count, many generations in a row
    for (unsigned workers : {0u, 1u, 3u, 7u, 15u}) {
        WorkerPool pool(workers);
        for (size_t tasks : {0, 1, 2, 3, 7, 8, 9, 16, 100, 1000}) {
            for (int rep = 0; rep < 200; rep++) {
                std::vector<std::atomic<int>> hit(tasks);
                for (auto& h : hit) h.store(0);
                pool.run(tasks, [&](size_t i) { hit[i].fetch_add(1); });
                for (size_t i = 0; i < tasks; i++)
                    if (hit[i].load() != 1) {
                        std::printf("FAIL workers %u tasks %zu index %zu hit %d times\n", workers, tasks, i, hit[i].load());
                        return 1;
                    }
            }
        }
    }
    // the closure of one run is gone before the next starts: sums through per-run locals
    {
        WorkerPool pool(7);
        for (int rep = 0; rep < 2000; rep++) {
            std::vector<long> part(16, 0);
            pool.run(16, [&](size_t i) { part[i] = (long)i * rep; });
            const long want = 120L * rep;
            if (std::accumulate(part.begin(), part.end(), 0L) != want) {
                std::printf("FAIL sum at rep %d\n", rep);
                return 1;
            }
        }
    }
    // pools used from different threads at the same time (one run at a time per pool)
    {
        WorkerPool a(5), b(5);
        std::atomic<long> total{0};
        auto use = [&](WorkerPool& p) {
            for (int rep = 0; rep < 500; rep++) p.run(32, [&](size_t i) { total.fetch_add((long)i); });
        };
        std::thread t1([&] { use(a); }), t2([&] { use(b); });
        t1.join();
        t2.join();
        if (total.load() != 2L * 500 * 496) {
            std::printf("FAIL concurrent pools: %ld\n", total.load());
            return 1;
        }
    }
    std::printf("worker pool ok\n");
    return 0;
}
