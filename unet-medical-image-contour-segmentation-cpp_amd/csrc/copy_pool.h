// copy_pool.h -- the pageable -> pinned staging copy of the RAW-in entry points (engine.cpp), free of any device API so that
// it can be compiled and tested without a GPU (tests/cpu/copy_pool_test.cpp).  Internal to libmiunet.so.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace miunet {

// A few persistent helper threads for the staging copy of a RAW image (pageable source -> the engine's pinned ring): one
// thread moves 6 MB at 10 - 20 GB/s, less when the source lies on the other socket, and sixteen such copies in a row are the
// upload stage of a chunk.  Four threads keep that stage shorter than the first micro-batch's network, so it stays hidden.
class CopyPool {
public:
    explicit CopyPool(int helpers) : n_(helpers)
    {
        for (int i = 0; i < n_; ++i) th_.emplace_back([this, i] { run(i); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        const int parts = n_ + 1;
        // CEILING of bytes / parts, rounded up to a page: parts * piece >= bytes for every size (the floored quotient dropped the
        // tail whenever it was already a multiple of 4096 and bytes % parts != 0 -- e.g. 1 048 578 bytes over 4 parts)
        const size_t piece = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        { std::lock_guard<std::mutex> lk(m_); dst_ = static_cast<char *>(dst); src_ = static_cast<const char *>(src); bytes_ = bytes; piece_ = piece; pending_ = n_; ++gen_; }
        cv_.notify_all();
        part(n_);                                      // the caller takes the last piece
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void part(int i)
    {
        const size_t lo = std::min(bytes_, piece_ * (size_t)i), hi = std::min(bytes_, lo + piece_);
        if (hi > lo) memcpy(dst_ + lo, src_ + lo, hi - lo);
    }
    void run(int i)
    {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            part(i);
            { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned long gen_ = 0;
    bool stop_ = false;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0, piece_ = 0;
    int pending_ = 0;
};

}  // namespace miunet
