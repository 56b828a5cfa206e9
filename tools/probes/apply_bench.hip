// Stand-alone timing harness for the block-substitution kernel (k_apply) on synthetic levels:
// F fronts of p pivot columns and u update rows, K contour nodes, ldw block-vector columns.
// Kinds timed: 1 (forward, contribution rows: C_I = -P21 W_J, K loop = p) and 3 (backward, gathered: Y_J -= U12 W[struct], K loop = u).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fcx-limited-range -I../../feastsolver.jl_amd/csrc apply_bench.hip -o apply_bench
#include "kernels_solve.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace feasthip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct Case { const char* name; int F, p, u, K; };

static void fill_random(cz* d, size_t n) {
    size_t m = std::min<size_t>(n, (size_t)4 << 20);
    std::vector<cz> h(m);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < m; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].x = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 1e-3; s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].y = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 1e-3; }
    for (size_t off = 0; off < n; off += m) CK(hipMemcpy(d + off, h.data(), std::min(m, n - off) * 16, hipMemcpyHostToDevice));
}

static void run_case(const Case& c, int ldw, int sl) {
    int F = c.F, p = c.p, u = c.u, K = c.K;
    int64_t n = (int64_t)F * p + u + 64;               // update rows of every front live after all pivot rows
    std::vector<FrontD> fr(F);
    std::vector<int> sidx((size_t)F * u);
    int64_t fs = 0, crows = 0;
    for (int f = 0; f < F; ++f) {
        FrontD& d = fr[f];
        d = FrontD();
        d.c0 = f * p; d.p = p; d.u = u; d.parent = -1;
        d.lp = fs; fs += (int64_t)(p + u) * p;
        d.f12 = fs; fs += (int64_t)p * u;
        d.cb = 0; d.sptr = (int64_t)f * u; d.crow = crows; crows += u; d.child_ptr = 0; d.nchild = 0; d.depth = 0; d.sk_off = -1; d.cmap = -1; d.ebp = -1; d.nb = 128;
        for (int i = 0; i < u; ++i) sidx[(size_t)f * u + i] = F * p + i;
    }
    std::vector<Tile> t1, t3;
    for (int f = 0; f < F; ++f) {
        for (int a = 0; a < (u + 16 * sl - 1) / (16 * sl); ++a) t1.push_back({f, 1 | (sl << 8), a, 0});
        for (int a = 0; a < (p + 16 * sl - 1) / (16 * sl); ++a) t3.push_back({f, 3 | (sl << 8), a, 0});
    }
    int64_t ws = n * ldw, crs = crows * ldw;
    FrontD* dfr; int* dsidx; Tile *dt1, *dt3; cz *fact, *W, *Y, *crow;
    CK(hipMalloc(&dfr, F * sizeof(FrontD))); CK(hipMemcpy(dfr, fr.data(), F * sizeof(FrontD), hipMemcpyHostToDevice));
    CK(hipMalloc(&dsidx, sidx.size() * 4)); CK(hipMemcpy(dsidx, sidx.data(), sidx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dt1, t1.size() * sizeof(Tile))); CK(hipMemcpy(dt1, t1.data(), t1.size() * sizeof(Tile), hipMemcpyHostToDevice));
    CK(hipMalloc(&dt3, t3.size() * sizeof(Tile))); CK(hipMemcpy(dt3, t3.data(), t3.size() * sizeof(Tile), hipMemcpyHostToDevice));
    CK(hipMalloc(&fact, (size_t)K * fs * 16)); fill_random(fact, (size_t)K * fs);      // random operands: the clock under load depends on the data
    CK(hipMalloc(&W, (size_t)K * ws * 16)); fill_random(W, (size_t)K * ws);
    CK(hipMalloc(&Y, (size_t)K * ws * 16)); CK(hipMemset(Y, 0, (size_t)K * ws * 16));
    CK(hipMalloc(&crow, (size_t)K * crs * 16)); CK(hipMemset(crow, 0, (size_t)K * crs * 16));
    DevSym S{}; S.fr = dfr; S.struct_idx = dsidx; S.rel = nullptr; S.child_idx = nullptr; S.perm = nullptr; S.c0_of_row = nullptr; S.cmap = nullptr; S.ebp = nullptr; S.n = (int)n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 1; kind <= 3; kind += 2) {
        auto go = [&]() {
            if (kind == 1) launch_apply(0, S, dt1, (int)t1.size(), fact, fs, W, Y, ws, ldw, crow, crs, K, false, nullptr, 0, nullptr, 0, sl);
            else launch_apply(0, S, dt3, (int)t3.size(), fact, fs, W, Y, ws, ldw, nullptr, 0, K, true, nullptr, 0, nullptr, 0, sl);
        };
        go(); CK(hipDeviceSynchronize());
        int reps = 5;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) go();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        double fb = (double)F * p * u * 16.0 * K;                                 // factor bytes
        double vb = (kind == 1 ? ((double)F * (p + u)) : ((double)F * (u + 2.0 * p))) * ldw * 16.0 * K;   // block-vector rows read + written once
        double fl = (double)F * p * u * 8.0 * ldw * K;
        printf("%-10s kind %d sl %d ldw %3d: tiles %6zu x %2d nodes  %.3f ms  factor %.2f TB/s  (+vec %.2f TB/s)  %.1f TFLOP/s (8/cmadd) = %.0f%% of 3M peak\n", c.name, kind, sl, ldw,
               kind == 1 ? t1.size() : t3.size(), K, ms, fb / ms / 1e9, (fb + vb) / ms / 1e9, fl / ms / 1e9, fl / ms / 1e9 / (77.2 * 8 / 6) * 100);
    }
    (void)hipFree(dfr); (void)hipFree(dsidx); (void)hipFree(dt1); (void)hipFree(dt3); (void)hipFree(fact); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(crow);
}

int main(int argc, char** argv) {
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_apply<false>, 256, 0));
    printf("occupancy k_apply<false>: %d workgroups / CU\n", occ);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_apply<true>, 256, 0));
    printf("occupancy k_apply<true>: %d workgroups / CU\n", occ);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k_apply<false>));
    printf("k_apply<false>: regs %d, static LDS %zu, max threads %d\n", fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
    Case cases[] = {
        {"top", 3, 1024, 2048, 16}, {"top8", 3, 1024, 2048, 8}, {"root-ish", 1, 3072, 3072, 16},
        {"mid", 48, 128, 1024, 16}, {"mid2", 24, 256, 1536, 16}, {"low", 600, 40, 400, 16}, {"low2", 400, 64, 600, 16},
    };
    if (argc > 1) {     // few-node study: one factorised unit of 128 columns, tile heights 64 / 32 / 16
        select_apply_kernel(1);
        Case one[] = { {"top-1unit", 3, 1024, 2048, 1}, {"mid-1unit", 48, 128, 1024, 1}, {"top-2unit", 3, 1024, 2048, 2}, {"top-8unit", 3, 1024, 2048, 8},
                       {"low-8unit", 600, 40, 400, 8}, {"low2-8unit", 400, 64, 600, 8} };
        for (const Case& c : one) for (int sl : {4, 2, 1}) run_case(c, 128, sl);
        return 0;
    }
    for (int variant = 0; variant < 2; ++variant) {
        select_apply_kernel(variant);
        printf("=== %s\n", variant ? "k_apply2 (no LDS, no barriers)" : "k_apply (LDS-staged)");
        for (const Case& c : cases) run_case(c, 64, 4);
        run_case(cases[0], 128, 4);
    }
    return 0;
}
