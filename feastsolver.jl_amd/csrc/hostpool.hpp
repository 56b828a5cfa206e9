// Host threads of the library: the analysis (ordering, symbolic structures, launch plans) and the reduced eigensolver run their
// parallel regions on ONE process-wide pool of sleeping threads instead of spawning std::threads per region.
#pragma once
#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <thread>
#include <type_traits>

namespace feasthip {

int host_cpu_budget();      // hardware threads capped by the cgroup CPU quota

// Runs fn(0) .. fn(nparts - 1), part 0 on the calling thread and the others on whichever threads get to them first: helper tasks on
// the process-wide pool of host threads (created at first use, host_cpu_budget() - 1 of them, never joined: they sleep between
// analyses) claim parts from a shared counter, and so does the caller once it is done with part 0 -- a busy (or, after fork(), absent)
// pool only means that the caller does more of the work itself.  Against a std::thread per part: 160 us per region of 8 parts on the
// GPU box's host, ~50 regions on the critical path of C3's ordering.
void pool_submit(std::function<void()> task);      // hostpool.cpp
int pool_size();

template <class F>
inline void run_parts(int nparts, F&& fn) {
    if (nparts <= 1) { fn(0); return; }
    struct State { std::atomic<int> next{1}, done{0}; };
    auto st = std::make_shared<State>();
    const int helpers = std::min(nparts - 1, pool_size());
    std::remove_reference_t<F>* fp = &fn;
    for (int h = 0; h < helpers; ++h)
        pool_submit([st, fp, nparts]() {
            for (;;) {
                const int part = st->next.fetch_add(1, std::memory_order_relaxed);
                if (part >= nparts) return;          // (fn is only touched after a successful claim: the caller is still waiting then)
                (*fp)(part);
                st->done.fetch_add(1, std::memory_order_release);
            }
        });
    fn(0);
    for (;;) {
        const int part = st->next.fetch_add(1, std::memory_order_relaxed);
        if (part >= nparts) break;
        fn(part);
        st->done.fetch_add(1, std::memory_order_release);
    }
    for (int spin = 0; st->done.load(std::memory_order_acquire) < nparts - 1; ++spin) if (spin > 64) std::this_thread::yield();
}

}  // namespace feasthip
