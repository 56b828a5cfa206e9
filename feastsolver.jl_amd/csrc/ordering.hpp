// Multilevel vertex separator (see ordering.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace feasthip {

struct WGraph {            // undirected, symmetric adjacency, no self loops
    int n = 0;
    std::vector<int> xadj, adj, ewgt, vwgt;
};

// label[v] = 0 / 1 (the two sides) or 2 (separator).  No edge joins side 0 and side 1.
// Returns false if no useful separator was found.
bool multilevel_separator(const WGraph& g, std::vector<int>& label, uint64_t seed);

}  // namespace feasthip
