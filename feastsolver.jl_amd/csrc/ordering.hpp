// Multilevel vertex separator (see ordering.cpp).
#pragma once
#include <cstdint>
#include <memory>
#include <utility>
#include "hostpool.hpp"
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

// label[v] = 0 / 1 (the two sides) or 2 (separator).  No edge joins side 0 and side 1.
// Returns false if no useful separator was found.
bool multilevel_separator(const WGraph& g, std::vector<int>& label, uint64_t seed);

}  // namespace feasthip
