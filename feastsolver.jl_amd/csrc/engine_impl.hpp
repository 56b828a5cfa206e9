// Shared by the engine translation units (engine*.hip): includes, the HIP error macro, small helpers.
#pragma once
#include "engine.hpp"

#include <algorithm>
#include <limits>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/feasthip.h"
#include "smalleig.hpp"

namespace feasthip {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

static inline cz to_cz(zc v) { return cz_make(v.real(), v.imag()); }

// Coefficient table of the accumulation kernel: entry [q][unit kk][column] = f(contour node, column j < m0, q); with the
// conjugate fold a unit carries two nodes and the mirror node's coefficients occupy columns [ldw, 2 ldw).
template <class F>
std::vector<zc> Engine::node_table(F f, int nq) const {
    const int lw = solve_width(), ni = ninst();       // adjoint fold: the mirror node's coefficients are the rows of the adjoint instances
    std::vector<zc> e((size_t)nq * ni * lw, zc(0.0));
    for (int q = 0; q < nq; ++q)
        for (int kk = 0; kk < Kloc; ++kk)
            for (int j = 0; j < m0; ++j) {
                e[((size_t)q * ni + kk) * lw + j] = f(local_[kk], j, q);
                if (fold_) e[((size_t)q * ni + kk) * lw + ldw + j] = f(partner_[kk], j, q);
                if (adj_fold_) e[((size_t)q * ni + Kloc + kk) * lw + j] = f(partner_[kk], j, q);
            }
    return e;
}

}  // namespace feasthip
