// Host-only statistic for a left-looking forward sweep over the deep tree levels: how densely the update rows of
// a front D fill the 16-row slabs of the pivot blocks of its ancestors (and of the contribution rows of its ancestor at the cut depth).
//   g++ -O2 -std=c++17 -pthread -Ifeastsolver.jl_amd/csrc -o /tmp/ll_density tools/probes/ll_density.cpp feastsolver.jl_amd/csrc/symbolic.cpp feastsolver.jl_amd/csrc/ordering.cpp
#include "symbolic.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
using namespace feasthip;
int main(int argc, char** argv) {
    const int g = argc > 1 ? atoi(argv[1]) : 46, D0 = argc > 2 ? atoi(argv[2]) : 7;
    const int n = g * g * g;
    std::vector<int64_t> cp(n + 1, 0), ri;
    auto id = [&](int x, int y, int z) { return (x * g + y) * g + z; };
    for (int x = 0; x < g; ++x) for (int y = 0; y < g; ++y) for (int z = 0; z < g; ++z) {
        std::vector<int64_t> rows;
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
            if (abs(dx) + abs(dy) + abs(dz) > 2) continue;       // 19-point stencil
            int a = x + dx, b = y + dy, c = z + dz;
            if (a < 0 || b < 0 || c < 0 || a >= g || b >= g || c >= g) continue;
            rows.push_back(id(a, b, c));
        }
        std::sort(rows.begin(), rows.end());
        ri.insert(ri.end(), rows.begin(), rows.end());
        cp[id(x, y, z) + 1] = (int64_t)ri.size();
    }
    Symbolic S; SymbolicOptions opt;
    std::string e = analyze_pattern(n, cp.data(), ri.data(), opt, S);
    if (!e.empty()) { printf("error %s\n", e.c_str()); return 1; }
    printf("n %d supernodes %d depth %d nnz_l %lld sum_u %lld\n", n, S.nsuper, S.maxdepth, (long long)S.nnz_l, (long long)S.sum_u);
    long long rows_piv = 0, slabs_piv = 0, rows_con = 0, slabs_con = 0, segs = 0, deep_u = 0, deep_fronts = 0, vreads = 0, deep_p = 0;
    for (int D = 0; D < S.nsuper; ++D) {
        if (S.depth[D] < D0) continue;
        ++deep_fronts; deep_u += S.u(D); deep_p += S.p(D);
        int A0 = D;                                   // ancestor at the cut depth
        while (S.depth[A0] > D0) A0 = S.sn_parent[A0];
        std::map<std::pair<int, int>, int> slab;      // (destination front or -1 for A0's contribution rows, slab) -> rows
        const int* st = S.struct_idx.data() + S.struct_ptr[D];
        const int* a0s = S.struct_idx.data() + S.struct_ptr[A0];
        const int ua0 = S.u(A0);
        for (int i = 0; i < S.u(D); ++i) {
            int row = st[i], A = S.sn_of[row];
            if (S.depth[A] >= D0) slab[{A, (row - S.sn_start[A]) / 16}]++;
            else {
                int pos = A0 == D ? i : (int)(std::lower_bound(a0s, a0s + ua0, row) - a0s);
                slab[{-1, pos / 16}]++;
            }
        }
        for (auto& kv : slab) {
            ++segs; vreads += S.p(D);
            if (kv.first.first >= 0) { rows_piv += kv.second; ++slabs_piv; } else { rows_con += kv.second; ++slabs_con; }
        }
    }
    printf("cut depth %d: deep fronts %lld, pivot columns %lld, update rows %lld (2 x that = %lld row transfers of the stacks)\n", D0, deep_fronts, deep_p, deep_u, 2 * deep_u);
    printf("pivot destinations: %lld rows in %lld slabs of 16 (density %.2f); contribution destinations: %lld rows in %lld slabs (density %.2f)\n",
           rows_piv, slabs_piv, rows_piv / (16.0 * slabs_piv), rows_con, slabs_con, slabs_con ? rows_con / (16.0 * slabs_con) : 0.0);
    printf("segments %lld, block-vector rows read as V operand %lld (%.2f per update row)\n", segs, vreads, (double)vreads / deep_u);
}
