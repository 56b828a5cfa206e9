#include "hostpool.hpp"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <pthread.h>

namespace feasthip {

// CPUs this process may actually use: the hardware count capped by the cgroup CPU quota (a container with a 16-CPU quota on
// a 256-CPU host reports 256 hardware threads; running 32+ busy threads there only gets the process throttled)
int host_cpu_budget() {
    static const int budget = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw <= 0) hw = 4;
        long long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
            fclose(f);
        } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {    // cgroup v1
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &period) != 1) period = 100000; fclose(f2); }
        }
        if (quota > 0 && period > 0) hw = std::min<long long>(hw, std::max<long long>(1, quota / period));
        return hw;
    }();
    return budget;
}
// ---- the pool (hostpool.hpp: run_parts, pool_submit) ----
namespace {
struct HostPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    int nthreads = 0;
    HostPool() {
        nthreads = std::max(0, std::min(15, host_cpu_budget() - 1));
        if (const char* e = getenv("FEASTHIP_HOST_THREADS")) nthreads = std::max(0, std::min(63, atoi(e) - 1));
        for (int i = 0; i < nthreads; ++i)
            std::thread([this]() {
                for (;;) {
                    std::function<void()> task;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return !q.empty(); });
                        task = std::move(q.front());
                        q.pop_front();
                    }
                    task();
                }
            }).detach();
    }
};
// never destroyed (its threads wait on it until the process ends).  A forked child has the object but not the threads: the atfork
// handler makes pool_size() say 0 there, so run_parts hands nothing to a queue nobody drains and its caller does every part itself
std::atomic<int> g_pool_gone{0};
HostPool& host_pool() {
    static HostPool* p = [] { HostPool* q = new HostPool(); (void)pthread_atfork(nullptr, nullptr, [] { g_pool_gone.store(1); }); return q; }();
    return *p;
}
}  // namespace

void pool_submit(std::function<void()> task) {
    HostPool& P = host_pool();
    { std::lock_guard<std::mutex> lk(P.mu); P.q.push_back(std::move(task)); }
    P.cv.notify_one();
}
int pool_size() { HostPool& P = host_pool(); return g_pool_gone.load(std::memory_order_relaxed) ? 0 : P.nthreads; }

}  // namespace feasthip
