// medgp_loader.hpp -- the read-ahead side of medgp_train's continuous admission (round 5): a shared work counter and a background
// reader that turns its tickets into loaded items.  Header only and free of the trainer's types, so that host_logic_test can run it
// on the CPU (also under ThreadSanitizer: `make -C medgp_amd/host tsan`) -- its first version had three start-up races that only showed
// on the GPU box as a trainer that never ended.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

namespace medgp {

// the shared work counter: atomic fetch-and-increment of the integer in `path` (the same file protocol as
// medgp_amd/train_cohort.py take_ticket); without a path, a counter of this process.  -1: the file could not be opened / locked / rewritten
struct Tickets {
    std::string path;
    std::atomic<long long> local{0};
    long long take() {
        if (path.empty()) return local.fetch_add(1);
        const int fd = open(path.c_str(), O_RDWR | O_CREAT, 0644);
        if (fd < 0) return -1;
        long long k = -1;
        if (flock(fd, LOCK_EX) == 0) {
            char buf[40] = {0};
            const ssize_t r = read(fd, buf, sizeof buf - 1);
            k = (r > 0) ? atoll(buf) : 0;
            const std::string s = std::to_string(k + 1);
            if (lseek(fd, 0, SEEK_SET) != 0 || ftruncate(fd, 0) != 0 || write(fd, s.data(), s.size()) != (ssize_t)s.size()) k = -1;
            flock(fd, LOCK_UN);
        }
        close(fd);
        return k;
    }
};

// Background reader: `nthreads` threads take tickets 0, 1, 2, ... and call load(ticket) for every ticket below `count`, keeping up to
// `cap` loaded items ready, so that an admission never waits for the file system while the device idles.  (The cap matters with a
// shared counter: items held ready here are items an idle trainer elsewhere cannot take.)
template <typename Item>
class ReadAhead {
public:
    ReadAhead(size_t count_, Tickets &tk_, size_t cap_, int nthreads, std::function<std::unique_ptr<Item>(size_t)> load_)
        : count(count_), tk(tk_), cap(std::max<size_t>(cap_, 1)), load(std::move(load_)) {
        const int n = std::max(1, nthreads);
        live = n;                            // (before the first reader starts: a reader that finds the list exhausted decrements it)
        for (int i = 0; i < n; i++) th.emplace_back([this] { run(); });
    }
    ~ReadAhead() {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        cv_space.notify_all();
        for (auto &t : th) t.join();
    }
    ReadAhead(const ReadAhead &) = delete;
    ReadAhead &operator=(const ReadAhead &) = delete;
    // up to `want` loaded items; with `block` it waits for a full wave (min(want, cap) items, or everything that is left) -- the
    // caller's device is idle then, and one large admission beats many small ones
    std::vector<std::unique_ptr<Item>> take(size_t want, bool block) {
        std::vector<std::unique_ptr<Item>> out;
        std::unique_lock<std::mutex> l(mu);
        if (block) cv_ready.wait(l, [&] { return ready.size() >= std::min(want, cap) || live == 0; });
        while (!ready.empty() && out.size() < want) { out.push_back(std::move(ready.front())); ready.pop_front(); }
        l.unlock();
        cv_space.notify_all();
        return out;
    }
    void set_cap(size_t c) {   // (smaller: readers finish what they hold and then wait)
        { std::lock_guard<std::mutex> l(mu); cap = std::max<size_t>(c, 1); }
        cv_space.notify_all();
    }
    bool exhausted() {   // nothing ready and nothing will come
        std::lock_guard<std::mutex> l(mu);
        return ready.empty() && live == 0;
    }
    long long taken() const { return n_taken.load(); }
    bool failed() const { return counter_failed.load(); }

private:
    void run() {
        while (true) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv_space.wait(l, [&] { return stop || ready.size() + inflight < cap; });
                if (stop) break;
                inflight++;
            }
            const long long k = tk.take();
            if (k < 0) counter_failed.store(true);   // the shared counter file failed: the run must not end as a success
            std::unique_ptr<Item> p;
            if (k >= 0 && k < (long long)count) {
                n_taken.fetch_add(1);
                p = load((size_t)k);
            }
            std::unique_lock<std::mutex> l(mu);
            inflight--;
            const bool more = (bool)p;
            if (more) ready.push_back(std::move(p));
            else live--;                  // the list is exhausted (or the counter file failed)
            l.unlock();
            cv_ready.notify_all();
            cv_space.notify_all();        // (a reader waiting for room must see that this one is no longer in flight)
            if (!more) return;
        }
        std::lock_guard<std::mutex> l(mu);
        live--;
        cv_ready.notify_all();
    }
    const size_t count;
    Tickets &tk;
    size_t cap, inflight = 0;
    int live = 0;
    bool stop = false;
    std::function<std::unique_ptr<Item>(size_t)> load;
    std::atomic<long long> n_taken{0};
    std::atomic<bool> counter_failed{false};
    std::deque<std::unique_ptr<Item>> ready;
    std::mutex mu;
    std::condition_variable cv_ready, cv_space;
    std::vector<std::thread> th;
};

}  // namespace medgp
