// medgp_workpool.hpp -- persistent host worker threads for the per-patient host work of medgp_train / medgp_test
// (file loading, optimiser state machines, the small conditionals of the imputation pass).
//
// The reference runs one patient per process and uses OpenMP inside the objective (ref: inference/c_inference_exact.cpp:55-57,
// kernel/c_kernel_LMC_SM.cpp:219-226).  Here the device evaluates; the host threads only prepare and consume batches.  Workers
// BLOCK on a condition variable between jobs -- no spinning: an OpenMP team's spinning workers starved the HIP runtime under
// the GPU box's cgroup CPU quota.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace medgp {

// host cores this process may use: hardware threads capped by the cgroup CPU quota (a box with many more hardware threads
// than quota must not get one worker per hardware thread)
inline int usable_cores() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q;
    long long per = 0;
    if (f >> q >> per && q != "max" && per > 0) n = (int)std::min<long long>(n, std::max<long long>(1, atoll(q.c_str()) / per));
    return std::max(1, std::min(n, 64));
}

class WorkPool {
public:
    explicit WorkPool(int nthreads) {
        for (int i = 1; i < nthreads; i++) workers.emplace_back([this] { loop(); });
    }
    ~WorkPool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
    WorkPool(const WorkPool &) = delete;
    WorkPool &operator=(const WorkPool &) = delete;
    // fn(i) for i in [0, n), dynamically dealt in chunks; returns when all are done (the caller works too).  An exception thrown
    // by fn on any thread (bad_alloc, a throwing loader) is caught there, the remaining items are skipped, and the FIRST one is
    // rethrown here on the calling thread -- a worker never reaches std::terminate.
    void parallel_for(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        if (workers.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        {
            std::lock_guard<std::mutex> l(m);
            job = &fn; total = n; next.store(0); pending = (int)workers.size(); gen++;
            failed.store(false); error = nullptr;
        }
        cv.notify_all();
        run();
        std::unique_lock<std::mutex> l(m);
        done_cv.wait(l, [this] { return pending == 0; });
        job = nullptr;
        if (error) { std::exception_ptr e = error; error = nullptr; l.unlock(); std::rethrow_exception(e); }
    }
    int size() const { return (int)workers.size() + 1; }

private:
    void run() {
        const int chunk = std::max(1, total / (8 * ((int)workers.size() + 1)));
        while (!failed.load(std::memory_order_relaxed)) {
            const int i0 = next.fetch_add(chunk);
            if (i0 >= total) break;
            try {
                for (int i = i0; i < std::min(total, i0 + chunk); i++) (*job)(i);
            } catch (...) {
                std::lock_guard<std::mutex> l(m);
                if (!error) error = std::current_exception();
                failed.store(true);
            }
        }
    }
    void loop() {
        unsigned long long seen = 0;
        while (true) {
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            run();
            { std::lock_guard<std::mutex> l(m); pending--; }
            done_cv.notify_one();
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv, done_cv;
    const std::function<void(int)> *job = nullptr;
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr error = nullptr;
    int total = 0, pending = 0;
    unsigned long long gen = 0;
    bool stop = false;
};

}  // namespace medgp
