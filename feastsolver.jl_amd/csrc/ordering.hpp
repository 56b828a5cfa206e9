// Multilevel vertex separator (see ordering.cpp).
#pragma once
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace feasthip {

// vector<int> whose resize() leaves new elements uninitialised: the big adjacency arrays are written once, by several threads, right
// after they are sized -- a zero fill would touch (page-fault) all of it serially first
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using ivec = std::vector<int, DefaultInitAlloc<int>>;

struct WGraph {            // undirected, symmetric adjacency, no self loops
    int n = 0;
    std::vector<int> xadj, vwgt;
    ivec adj, ewgt;
};

// Fixed partition of a large graph's vertices into contiguous ranges that separate host threads work on: the number of ranges
// does not depend on the machine, so neither does any result.
constexpr int ND_NPART = 8;
constexpr int ND_PAR_MIN_N = 2000;      // (16384 .. 500 measured on the GPU box host, 16-CPU quota: C3 analysis 102 / 96 / 90 / 88 / 101 ms at 16384 / 8000 / 2000 / 1000 / 500)

// Runs fn(0) .. fn(nparts - 1), part 0 on the calling thread and the others on whichever threads get to them first: helper tasks on
// the process-wide pool of host threads (created at first use, host_cpu_budget() - 1 of them, never joined: they sleep between
// analyses) claim parts from a shared counter, and so does the caller once it is done with part 0 -- a busy (or, after fork(), absent)
// pool only means that the caller does more of the work itself.  Against a std::thread per part: 160 us per region of 8 parts on the
// GPU box's host, ~50 regions on the critical path of C3's ordering.
void pool_submit(std::function<void()> task);      // ordering.cpp
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

// label[v] = 0 / 1 (the two sides) or 2 (separator).  No edge joins side 0 and side 1.
// Returns false if no useful separator was found.
bool multilevel_separator(const WGraph& g, std::vector<int>& label, uint64_t seed);

}  // namespace feasthip
