// group_sched.h -- the host-side scheduling of the multi-device group (group.cpp), free of any device API so that it can be
// compiled and tested on a machine without a GPU (tests/cpu/group_sched_test.cpp): the contiguous shard split, one
// persistent worker thread per rank, and "run this on every rank, report the first failure".  Internal to libmiunet.so.
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace miunet {

// rank r of `world` owns items [lo, hi) of n: contiguous, in rank order, the first n % world ranks one item more
inline void shard_range(int n, int r, int world, int &lo, int &hi)
{
    const int q = n / world, rem = n % world;
    lo = r * q + (r < rem ? r : rem);
    hi = lo + q + (r < rem ? 1 : 0);
}

// One persistent worker: submit() hands it a job, wait() returns when that job is done.  One job at a time.
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    Worker(const Worker &) = delete;
    Worker &operator=(const Worker &) = delete;
    void submit(std::function<void()> job)
    {
        { std::lock_guard<std::mutex> lk(m_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !busy_; });
    }

private:
    void loop()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [this] { return stop_ || busy_; });
            if (stop_) return;
            std::function<void()> job = std::move(job_);
            lk.unlock();
            job();
            lk.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false;
    std::thread th_;
};

// fn(rank) on every rank's worker, all at once; returns 0 or the code of the lowest failing rank, whose message -- fetched
// by `last_error` ON the failing worker's thread (error strings are thread-local) -- comes back in `msg`.
inline int run_on_all_ranks(std::vector<std::unique_ptr<Worker>> &workers, const std::function<int(int)> &fn,
                            const std::function<std::string()> &last_error, int &bad_rank, std::string &msg)
{
    const int R = (int)workers.size();
    std::vector<int> rc(R, 0);
    std::vector<std::string> m(R);
    for (int r = 0; r < R; ++r)
        workers[r]->submit([&, r] {
            rc[r] = fn(r);
            if (rc[r]) m[r] = last_error();
        });
    for (int r = 0; r < R; ++r) workers[r]->wait();
    for (int r = 0; r < R; ++r)
        if (rc[r]) { bad_rank = r; msg = m[r]; return rc[r]; }
    bad_rank = -1;
    return 0;
}

}  // namespace miunet
