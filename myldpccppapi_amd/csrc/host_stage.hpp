/*
 * host_stage.hpp -- what the host-buffer entry point (ldpc_decode, the reference's
 * Coder::decode signature, MyLdpc.cpp:571-618 with its blocking copies at :796 / :988) needs
 * besides a decoder: persistent worker threads, the page arithmetic of the opt-in
 * "lock the caller's pages" mode, and a process-wide record of every range this library has
 * page-locked.
 *
 * Rules this file exists to keep (round 2 ended with a silent abort of a test process right
 * after a host-buffer decode that had page-locked caller pages from three short-lived threads):
 *   - no thread is created or destroyed per call: a handle owns its workers from creation to
 *     destruction (HIP keeps per-thread state; creating and dropping it per call is churn the
 *     runtime was never asked to be good at);
 *   - every hipHostRegister / hipHostUnregister goes through ONE mutex and ONE table, its result
 *     is checked, and a range that could not be released stays on record (and is reported);
 *   - a registered block never leaves the byte range of the call that registered it.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ldpc {

/* ------------------------------------------------------------------ Worker
 * One persistent thread with a FIFO of jobs.  A Job lives in the submitter's frame (or a
 * container it owns) until wait() has returned for it. */
struct Job {
    std::function<int()> fn;
    int rc = 0;
    std::string err;            /* the worker thread's last-error text when rc != 0 */
    bool done = false;
};

class Worker {
public:
    /* err_text: reads the calling (= worker) thread's last-error string after a job failed */
    explicit Worker(std::function<std::string()> err_text) : err_text_(std::move(err_text)) {}
    ~Worker() { stop(); }
    Worker(const Worker &) = delete;
    Worker &operator=(const Worker &) = delete;

    /* false: the thread could not be started (the handle's creation then fails) */
    bool start()
    {
        if (th_.joinable()) return true;
        try {
            th_ = std::thread([this] { loop(); });
        } catch (...) {
            return false;
        }
        return true;
    }

    void submit(Job *j)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            j->done = false;
            q_.push_back(j);
        }
        cv_.notify_all();
    }

    /* blocks until the job has run; returns its rc */
    int wait(Job *j)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return j->done; });
        return j->rc;
    }

    void stop()
    {
        if (!th_.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }

private:
    void loop()
    {
        for (;;) {
            Job *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || !q_.empty(); });
                if (q_.empty()) return;          /* quit_ and nothing left to run */
                j = q_.front();
                q_.pop_front();
            }
            int rc;
            std::string err;
            try {
                rc = j->fn();
                if (rc) err = err_text_();
            } catch (const std::exception &e) {
                rc = -1;
                err = std::string("exception in a worker thread: ") + e.what();
            } catch (...) {
                rc = -1;
                err = "unknown exception in a worker thread";
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                j->rc = rc;
                j->err = std::move(err);
                j->done = true;
            }
            cv_.notify_all();
        }
    }

    std::function<std::string()> err_text_;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job *> q_;
    bool quit_ = false;
};

/* ---------------------------------------------------- page arithmetic (lock mode)
 * Group k of a call that decodes `frames` frames of N floats in launch groups of B frames, the
 * first float at byte address call_s0.  In lock mode the group's bytes [s0, s1) travel as
 *     head [s0, b0)        copied by the CPU into pinned scratch (less than a page),
 *     body [b0, body_end)  DMA-read in place from the page-locked block [b0, b1),
 *     tail [body_end, s1)  copied by the CPU (less than a page; only where the block had to stop
 *                          short of the group's end),
 * or, when no whole page lies inside the group (whole_by_cpu), entirely by the CPU.
 * Guarantees (tests/test_host_cpu.py checks them on the exported ldpc_host_block_plan):
 *     blocks of different groups are page-disjoint; every block consists of whole pages and lies
 *     inside [call_s0, call_s1) -- a block never reaches past the call's own bytes, not even by
 *     the rest of a page (the block of a non-last group used to be rounded UP past a
 *     short last group, into whatever follows the caller's buffer or a neighbouring thread's range). */
struct GroupBlocks {
    uintptr_t s0 = 0, s1 = 0;        /* the group's bytes */
    uintptr_t b0 = 0, b1 = 0;        /* the page-locked block (b0 == b1: none) */
    uintptr_t body_end = 0;          /* DMA covers [b0, body_end) */
    bool whole_by_cpu = false;
};

constexpr uintptr_t kPage = 4096;

inline GroupBlocks plan_group_blocks(uintptr_t call_s0, int64_t frames, int64_t N, int64_t B, int64_t k)
{
    GroupBlocks g;
    const uintptr_t call_s1 = call_s0 + (uintptr_t)frames * (uintptr_t)N * sizeof(float);
    const int64_t off = k * B, n = (frames - off < B) ? frames - off : B;
    g.s0 = call_s0 + (uintptr_t)off * (uintptr_t)N * sizeof(float);
    g.s1 = g.s0 + (uintptr_t)n * (uintptr_t)N * sizeof(float);
    const uintptr_t up = (g.s0 + kPage - 1) & ~(kPage - 1);
    /* a non-last group owns the pages up to the next group's first page boundary -- but never a
     * page that reaches past the call's last byte */
    const uintptr_t call_down = call_s1 & ~(kPage - 1);
    uintptr_t end = (g.s1 + kPage - 1) & ~(kPage - 1);
    if (end > call_down) end = call_down;
    if (g.s1 == call_s1) end = call_down;
    g.b0 = up;
    g.b1 = end;
    if (g.b0 >= g.b1) {
        g.b0 = g.b1 = 0;
        g.whole_by_cpu = true;
        g.body_end = 0;
        return g;
    }
    g.body_end = g.b1 < g.s1 ? g.b1 : g.s1;
    return g;
}

/* -------------------------------------------------- registry of page-locked ranges
 * Process-wide.  lock(): refuses a range that overlaps one already on record (another call of this
 * library on the same buffer), otherwise hipHostRegister under the mutex.  unlock(): hipHostUnregister
 * under the mutex; on failure the range moves to the stale list, which covers() still answers for
 * and stale_count() reports.  covers(): is this address inside a range this library locked (then the
 * memory is NOT "locked by the caller", whatever hipPointerGetAttributes says). */
class PageLockRegistry {
public:
    static PageLockRegistry &instance()
    {
        static PageLockRegistry r;
        return r;
    }

    hipError_t lock(void *p, size_t n, bool *overlap)
    {
        std::lock_guard<std::mutex> lk(m_);
        *overlap = false;
        const uintptr_t a = (uintptr_t)p, b = a + n;
        for (const Range &r : live_)
            if (a < r.b && r.a < b) { *overlap = true; return hipErrorHostMemoryAlreadyRegistered; }
        for (const Range &r : stale_)
            if (a < r.b && r.a < b) { *overlap = true; return hipErrorHostMemoryAlreadyRegistered; }
        const hipError_t e = hipHostRegister(p, n, hipHostRegisterPortable);
        if (e == hipSuccess) live_.push_back(Range{a, b});
        else (void)hipGetLastError();
        return e;
    }

    hipError_t unlock(void *p)
    {
        std::lock_guard<std::mutex> lk(m_);
        const uintptr_t a = (uintptr_t)p;
        size_t i = 0;
        for (; i < live_.size(); ++i)
            if (live_[i].a == a) break;
        if (i == live_.size()) return hipErrorHostMemoryNotRegistered;
        const hipError_t e = hipHostUnregister(p);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            stale_.push_back(live_[i]);
        }
        live_.erase(live_.begin() + (long)i);
        return e;
    }

    bool covers(const void *p)
    {
        std::lock_guard<std::mutex> lk(m_);
        return covers_unlocked(p);
    }

    /* Has the CALLER page-locked [first, last] itself (hipHostMalloc, hipHostRegister, a framework's
     * pinned allocator)?  Both ends must be known to the runtime and neither may lie in a range this
     * library locked -- a temporary lock of another in-flight call is not the caller's.
     * Asked under the registry's mutex, so no lock()/unlock() of this library runs in between.
     * Memory the runtime knows for other reasons (managed, device) also answers true: it is left to
     * the runtime's copy engine as it is. */
    bool caller_locked(const void *first, const void *last)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (covers_unlocked(first) || covers_unlocked(last)) return false;
        auto known = [](const void *q) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, q) == hipSuccess) return at.type != hipMemoryTypeUnregistered;
            (void)hipGetLastError();
            return false;
        };
        return known(first) && known(last);
    }

    size_t live_count() { std::lock_guard<std::mutex> lk(m_); return live_.size(); }
    size_t stale_count() { std::lock_guard<std::mutex> lk(m_); return stale_.size(); }

private:
    struct Range { uintptr_t a, b; };
    bool covers_unlocked(const void *p) const
    {
        const uintptr_t a = (uintptr_t)p;
        for (const Range &r : live_) if (a >= r.a && a < r.b) return true;
        for (const Range &r : stale_) if (a >= r.a && a < r.b) return true;
        return false;
    }
    std::mutex m_;
    std::vector<Range> live_, stale_;
};

}  // namespace ldpc
