// Unit test of csrc/host_stage.hpp on the CPU: the persistent worker (FIFO order, return codes, error text,
// exceptions, jobs pending at stop), the page arithmetic of the lock-pages mode, and the registry's refusal
// paths that need no device.  Prints "ok" and exits 0, or says what failed.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../myldpccppapi_amd/csrc/host_stage.hpp"

static thread_local std::string t_err;
#define CHECK(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main()
{
    using namespace ldpc;
    {   // FIFO order, return codes, the worker thread's own error text
        Worker w([] { return t_err; });
        CHECK(w.start() && w.start());
        std::vector<int> order;
        std::vector<Job> jobs(200);
        for (int i = 0; i < 200; ++i) {
            jobs[i].fn = [i, &order]() -> int {
                order.push_back(i);                       // one worker thread: no race
                if (i % 7 == 3) { t_err = "job " + std::to_string(i) + " failed"; return 40 + i; }
                return 0;
            };
            w.submit(&jobs[i]);
        }
        for (int i = 199; i >= 0; --i) {                  // waiting in any order is fine
            const int rc = w.wait(&jobs[i]);
            CHECK(rc == (i % 7 == 3 ? 40 + i : 0));
            CHECK((i % 7 == 3) == (jobs[i].err == "job " + std::to_string(i) + " failed"));
        }
        for (int i = 0; i < 200; ++i) CHECK(order[i] == i);
        // a job may be reused once waited for
        jobs[0].fn = [] { return 5; };
        w.submit(&jobs[0]);
        CHECK(w.wait(&jobs[0]) == 5);
    }
    {   // an exception inside a job does not end the process
        Worker w([] { return std::string("unused"); });
        CHECK(w.start());
        Job a, b, c;
        a.fn = []() -> int { throw std::runtime_error("boom"); };
        b.fn = []() -> int { throw 42; };
        c.fn = [] { return 0; };
        w.submit(&a); w.submit(&b); w.submit(&c);
        CHECK(w.wait(&a) == -1 && a.err.find("boom") != std::string::npos);
        CHECK(w.wait(&b) == -1 && !b.err.empty());
        CHECK(w.wait(&c) == 0);
    }
    {   // stop() runs what is still queued, then joins; a never-started worker stops trivially
        std::atomic<int> ran{0};
        std::vector<Job> jobs(50);
        {
            Worker w([] { return std::string(); });
            CHECK(w.start());
            for (auto &j : jobs) { j.fn = [&ran] { ++ran; return 0; }; w.submit(&j); }
        }                                                 // destructor = stop()
        CHECK(ran == 50);
        for (auto &j : jobs) CHECK(j.done);
        Worker idle([] { return std::string(); });
        idle.stop();
    }
    {   // page arithmetic: ADVICE r2's example -- (648, 324), max_batch 2048, 2049 frames: the 2592-byte last group
        const uintptr_t base = 0x7f0000001000ull + 12;
        const GroupBlocks g0 = plan_group_blocks(base, 2049, 648, 2048, 0), g1 = plan_group_blocks(base, 2049, 648, 2048, 1);
        const uintptr_t end = base + 2049ull * 648 * 4;
        CHECK(g0.s0 == base && g0.s1 == g1.s0 && g1.s1 == end);
        CHECK(g0.b0 % kPage == 0 && g0.b1 % kPage == 0 && g0.b0 >= base && g0.b1 <= end);   // never past the call's bytes
        CHECK(g0.body_end <= g0.s1 && g0.s1 - g0.body_end < kPage);
        CHECK(g1.whole_by_cpu || (g1.b0 >= g0.b1 && g1.b1 <= end));
    }
    {   // registry without a device: a failed hipHostRegister leaves no record; unknown pointers are refused
        PageLockRegistry &r = PageLockRegistry::instance();
        static char buf[3 * 4096];
        void *p = (void *)(((uintptr_t)buf + 4095) & ~(uintptr_t)4095);
        bool overlap = true;
        const hipError_t e = r.lock(p, 4096, &overlap);
        if (e != hipSuccess) CHECK(r.live_count() == 0 && !overlap && !r.covers(p));
        else {                                            // a device is present: the full cycle
            CHECK(r.live_count() == 1 && r.covers(p) && !r.caller_locked(p, (char *)p + 4095));
            bool ov2 = false;
            CHECK(r.lock((char *)p + 0, 4096, &ov2) != hipSuccess && ov2);
            CHECK(r.unlock(p) == hipSuccess);
        }
        CHECK(r.unlock(p) == hipErrorHostMemoryNotRegistered);
        CHECK(r.live_count() == 0 && r.stale_count() == 0);
    }
    printf("ok\n");
    return 0;
}
