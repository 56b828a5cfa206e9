// Host-side symbolic analysis for the batched multifrontal LU that backs the
// reference's `factorizer` seam (src/feast.jl:28-38, 104-114; src/nlfeast.jl:17-28).
// The sparsity pattern of T(z_k) is the same for every contour node and every
// FEAST iteration, so ordering + symbolic factorisation run ONCE per problem
// and are shared by all nodes; only values differ per node.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace feasthip {

// (host_cpu_budget(): hostpool.hpp)

struct Symbolic {
    int n = 0;
    bool dense = false;
    std::vector<int> perm;    // perm[new] = old
    std::vector<int> iperm;   // iperm[old] = new
    int nsuper = 0;
    std::vector<int> sn_start;       // nsuper+1, columns of supernode s are [sn_start[s], sn_start[s+1])
    std::vector<int> sn_of;          // n, supernode of each (new) column
    std::vector<int> sn_parent;      // nsuper, -1 for roots
    std::vector<int> depth;          // nsuper, roots = 0
    std::vector<int64_t> struct_ptr; // nsuper+1
    std::vector<int> struct_idx;     // sorted update rows (new indices) of each supernode
    std::vector<int64_t> child_ptr;  // nsuper+1
    std::vector<int> child_idx;
    std::vector<int> rel;            // same indexing as struct_idx: position of each update row in the PARENT front
    std::vector<int64_t> lp_off;     // per supernode: offset of the (p+u) x p L panel (col-major, ld p+u)
    std::vector<int64_t> f12_off;    // per supernode: offset of the p x u U12 panel (col-major, ld p)
    int64_t fact_size = 0;           // complex elements per contour node
    std::vector<int64_t> cb_off;     // offset of the u x u contribution block in arena depth%2
    int64_t cb_size[2] = {0, 0};
    std::vector<int64_t> crow_off;   // offset (in rows) of the u-row solve contribution in arena depth%2
    int64_t crow_size[2] = {0, 0};
    std::vector<int64_t> cb_used, crow_used;   // per depth: elements / rows actually used in the arena
    int64_t sum_u = 0;
    int maxdepth = 0;
    std::vector<std::vector<int>> levels;  // supernodes by depth
    // statistics
    int64_t nnz_l = 0, nnz_u = 0;    // incl. diagonal block once in L
    double flops_factor = 0;         // complex multiply-adds * 8 real flops
    int max_p = 0, max_front = 0;
    int p(int s) const { return sn_start[s + 1] - sn_start[s]; }
    int u(int s) const { return (int)(struct_ptr[s + 1] - struct_ptr[s]); }
};

struct SymbolicOptions {
    int leaf_size = 64;       // nested dissection stops below this many vertices
    int relax_small = 32;     // always merge a child if the merged supernode has <= this many columns
    int relax_mid = 64;       // merge up to this many columns if the zero fraction stays below zmid
    double zmid = 0.35;
    double zbig = 0.05;       // larger merges only below this zero fraction
    int ordering = 0;         // 0 = nested dissection, 1 = natural (identity)
};

// Pattern given as CSC/CSR of a structurally arbitrary n x n matrix (0-based);
// it is symmetrised internally.  Returns empty string on success.
std::string analyze_pattern(int n, const int64_t* colptr, const int64_t* rowidx, const SymbolicOptions& opt, Symbolic& S);

// Single dense front (dense A, B path).
void analyze_dense(int n, Symbolic& S);

// Arena destination of entry (i_old, j_old); returns -1 if outside the structure.
int64_t entry_dest(const Symbolic& S, int i_old, int j_old);

}  // namespace feasthip
