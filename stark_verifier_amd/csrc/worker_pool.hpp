// A fixed set of host threads kept waiting between uses (pure C++: no HIP here, so that it can be tested on its own).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// A fixed set of host threads kept waiting between uses: starting eight std::threads for every 64 MiB staging copy or
// every piece of a batch costs a quarter of a millisecond each time, a fifth of the copy itself.
// run(tasks, f) calls f(i) for every i in [0, tasks), the caller included, and returns when all calls are done.
// One run at a time per pool (callers that may overlap use a pool each).
class WorkerPool {
   public:
    explicit WorkerPool(unsigned workers) {
        for (unsigned i = 0; i < workers; i++) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        work_.notify_all();
        for (auto& t : threads_) t.join();
    }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
    unsigned workers() const { return (unsigned)threads_.size(); }

    void run(size_t tasks, const std::function<void(size_t)>& f) {
        if (tasks == 0) return;
        if (threads_.empty() || tasks == 1) {
            for (size_t i = 0; i < tasks; i++) f(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &f;
            tasks_ = tasks;
            next_.store(0);
            pending_ = threads_.size();
            ++gen_;
        }
        work_.notify_all();
        for (size_t i; (i = next_.fetch_add(1)) < tasks;) f(i);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });  // every worker has left f: it may go out of scope
        fn_ = nullptr;
    }

   private:
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            work_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            const std::function<void(size_t)>* f = fn_;
            const size_t n = tasks_;
            lk.unlock();
            for (size_t i; (i = next_.fetch_add(1)) < n;) (*f)(i);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable work_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t tasks_ = 0, pending_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};
