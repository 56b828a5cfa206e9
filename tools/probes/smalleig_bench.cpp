// host timing of the reduced general eigensolver (csrc/smalleig.cpp), random and nearly diagonal pencils, m = 64 and 128:
//   g++ -O3 -mavx2 -mfma -std=c++17 -pthread -Ifeastsolver.jl_amd/csrc -o tools/probes/smalleig_bench tools/probes/smalleig_bench.cpp feastsolver.jl_amd/csrc/smalleig.cpp feastsolver.jl_amd/csrc/hostpool.cpp
// FEASTHIP_HOST_THREADS=1 runs everything on the calling thread (the result hashes must not change)
#include "smalleig.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
using namespace feasthip;
int main() {
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (int m : {64, 128}) for (int warm = 0; warm < 2; ++warm) {
        std::vector<zc> A((size_t)m*m), B((size_t)m*m), lam(m), V((size_t)m*m);
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) {
                zc e(nd(g), nd(g));
                A[i + (size_t)j*m] = warm ? (i == j ? zc(j + 1.0, 0.3 * j) : 1e-6 * e) : e;
                zc f(nd(g), nd(g));
                B[i + (size_t)j*m] = (i == j ? zc(1.0) : zc(0.0)) + (warm ? 1e-6 : 0.1 / std::sqrt(m)) * f;
            }
            auto t0 = std::chrono::steady_clock::now();
            int info = eig_generalized(m, A.data(), B.data(), lam.data(), V.data());
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (info) printf("info %d\n", info);
            best = std::min(best, ms);
        }
        unsigned long long hsh = 1469598103934665603ull;      // bit pattern of the last result: the threaded paths must reproduce the serial one exactly
        for (auto& v : V) { unsigned long long b[2]; double d[2] = {v.real(), v.imag()}; memcpy(b, d, 16); hsh = (hsh ^ b[0]) * 1099511628211ull; hsh = (hsh ^ b[1]) * 1099511628211ull; }
        for (auto& v : lam) { unsigned long long b[2]; double d[2] = {v.real(), v.imag()}; memcpy(b, d, 16); hsh = (hsh ^ b[0]) * 1099511628211ull; hsh = (hsh ^ b[1]) * 1099511628211ull; }
        printf("m=%d %s: eig_generalized %.3f ms  result hash %016llx\n", m, warm ? "nearly diagonal" : "random", best, hsh);
    }
}
