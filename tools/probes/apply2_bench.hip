// Experiment: barrier-free, LDS-free variant of the block-substitution tile.  Every wave owns 16 output rows x 64 block-vector
// columns and streams BOTH operands straight from global memory into matrix-core operand registers, software pipelined per
// k-step (A: DA k-steps ahead, HBM; V: DV k-steps ahead, L2): no workgroup barrier, no LDS, counted vmcnt waits.
// Compared against k_apply (kind 1: C_I = -P21 W_J) on the synthetic levels of apply_bench.hip.
#include "kernels_solve.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace feasthip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int DA = 8, DV = 4;      // ring sizes (k-steps): A loads are issued DA - 1 steps ahead, V loads DV - 1 steps ahead

// out[r, :] = -sum_k A[r, k] V[k, :]   r in [rbase, rend), k < Kd;  A column-major (lda), V/out row-major (ldw = 64)
__global__ __launch_bounds__(256, 2) void k_apply2p(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, const cz* __restrict__ fact, int64_t fs,
                                                   const cz* __restrict__ W, int64_t ws, cz* crow, int64_t crs, int ntiles) {
    const int item = xcd_item();
    Tile t = tiles[item % ntiles];
    const int k = item / ntiles;
    FrontD f = fr[t.f];
    const int p = f.p, ld = f.p + f.u, ldw = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const int rbase = p + t.b * 64 + wave * 16, rend = ld;
    if (rbase >= rend) return;
    const int Kd = p;
    const cz* A = fact + (int64_t)k * fs + f.lp + min(rbase + li, rend - 1);            // + kk * ld
    const cz* V = W + (int64_t)k * ws + (int64_t)f.c0 * ldw + li;                        // + kk * ldw + ct * 16
    cz* out = crow + (int64_t)k * crs + (f.crow + rbase - p) * (int64_t)ldw;
    d4 t1[4], t2[4], t3[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { t1[q] = (d4){0, 0, 0, 0}; t2[q] = (d4){0, 0, 0, 0}; t3[q] = (d4){0, 0, 0, 0}; }
    cz ar[DA], vr[DV][4];
    const int nks = (Kd + 3) >> 2;
    auto loadA = [&](int s, int slot) { int kk = min(4 * s + lk, Kd - 1); ar[slot] = A[(int64_t)kk * ld]; };
    auto loadV = [&](int s, int slot) {
        int kk = min(4 * s + lk, Kd - 1);
        const cz* vp = V + (int64_t)kk * ldw;
#pragma unroll
        for (int q = 0; q < 4; ++q) vr[slot][q] = vp[q * 16];
    };
    // prologue
#pragma unroll
    for (int s = 0; s < DA - 1; ++s) loadA(min(s, nks - 1), s);
#pragma unroll
    for (int s = 0; s < DV - 1; ++s) loadV(min(s, nks - 1), s);
    for (int s0 = 0; s0 < nks; s0 += DA) {
#pragma unroll
        for (int j = 0; j < DA; ++j) {
            const int s = s0 + j;
            {
                __builtin_amdgcn_sched_barrier(0);
                loadA(min(s + DA - 1, nks - 1), (j + DA - 1) % DA);
                loadV(min(s + DV - 1, nks - 1), (j + DV - 1) % DV);
                __builtin_amdgcn_sched_barrier(0);
                cz a = ar[j];
                const bool keep = 4 * s + lk < Kd;
                a.x = keep ? a.x : 0.0; a.y = keep ? a.y : 0.0;
                const double as = a.x + a.y;
#pragma unroll
                for (int q = 0; q < 4; ++q) t1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, vr[j % DV][q].x, t1[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) t2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, vr[j % DV][q].y, t2[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) t3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, vr[j % DV][q].x + vr[j % DV][q].y, t3[q], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int orow = rbase + lk + 4 * r;
        if (orow < rend) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(int64_t)(lk + 4 * r) * ldw + q * 16 + li] = cz_make(-(t1[q][r] - t2[q][r]), -(t3[q][r] - t1[q][r] - t2[q][r]));
        }
    }
}

struct Case { const char* name; int F, p, u, K; };

static void fill_random(cz* d, size_t n) {
    size_t m = std::min<size_t>(n, (size_t)4 << 20);
    std::vector<cz> h(m);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < m; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].x = (double)(s >> 11) / 9007199254740992.0 - 0.5; s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].y = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
    for (size_t off = 0; off < n; off += m) CK(hipMemcpy(d + off, h.data(), std::min(m, n - off) * 16, hipMemcpyHostToDevice));
}

static void run_case(const Case& c, bool check) {
    int F = c.F, p = c.p, u = c.u, K = c.K, ldw = 64;
    int64_t n = (int64_t)F * p + u + 64;
    std::vector<FrontD> fr(F);
    int64_t fs = 0, crows = 0;
    for (int f = 0; f < F; ++f) {
        FrontD& d = fr[f];
        d = FrontD();
        d.c0 = f * p; d.p = p; d.u = u; d.parent = -1; d.lp = fs; fs += (int64_t)(p + u) * p; d.f12 = fs; fs += (int64_t)p * u;
        d.sptr = 0; d.crow = crows; crows += u; d.nchild = 0; d.sk_off = -1; d.cmap = -1; d.ebp = -1; d.nb = 128;
    }
    std::vector<Tile> t1;
    for (int f = 0; f < F; ++f) for (int a = 0; a < (u + 63) / 64; ++a) t1.push_back({f, 1 | (4 << 8), a, 0});
    int64_t ws = n * ldw, crs = crows * ldw;
    FrontD* dfr; Tile* dt1; cz *fact, *W, *Y, *crow, *crow2;
    CK(hipMalloc(&dfr, F * sizeof(FrontD))); CK(hipMemcpy(dfr, fr.data(), F * sizeof(FrontD), hipMemcpyHostToDevice));
    CK(hipMalloc(&dt1, t1.size() * sizeof(Tile))); CK(hipMemcpy(dt1, t1.data(), t1.size() * sizeof(Tile), hipMemcpyHostToDevice));
    CK(hipMalloc(&fact, (size_t)K * fs * 16)); fill_random(fact, (size_t)K * fs);
    CK(hipMalloc(&W, (size_t)K * ws * 16)); fill_random(W, (size_t)K * ws);
    CK(hipMalloc(&Y, (size_t)K * ws * 16)); CK(hipMemset(Y, 0, (size_t)K * ws * 16));
    CK(hipMalloc(&crow, (size_t)K * crs * 16)); CK(hipMemset(crow, 0, (size_t)K * crs * 16));
    CK(hipMalloc(&crow2, (size_t)K * crs * 16)); CK(hipMemset(crow2, 0, (size_t)K * crs * 16));
    DevSym S{}; S.fr = dfr; S.n = (int)n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        select_apply_kernel(variant == 1);
        auto go = [&]() {
            if (variant < 2) launch_apply(0, S, dt1, (int)t1.size(), fact, fs, W, Y, ws, ldw, crow, crs, K);
            else hipLaunchKernelGGL(k_apply2p, dim3((unsigned)(t1.size() * K)), dim3(256), 0, 0, dfr, dt1, fact, fs, W, ws, crow2, crs, (int)t1.size());
        };
        go(); CK(hipDeviceSynchronize());
        int reps = 10;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) go();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        double fb = (double)F * p * u * 16.0 * K, vb = ((double)F * (p + u)) * ldw * 16.0 * K, fl = (double)F * p * u * 8.0 * ldw * K;
        printf("%-10s %s: tiles %6zu x %2d nodes  %.3f ms  factor %.2f TB/s  (+vec %.2f TB/s)  %.1f TFLOP/s (8/cmadd)\n", c.name, variant == 0 ? "k_apply  (LDS)" : variant == 1 ? "k_apply2 (lib)" : "prototype     ",
               t1.size(), K, ms, fb / ms / 1e9, (fb + vb) / ms / 1e9, fl / ms / 1e9);
    }
    if (check) {
        size_t cnt = (size_t)K * crs;
        std::vector<cz> a(cnt), b(cnt);
        CK(hipMemcpy(a.data(), crow, cnt * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), crow2, cnt * 16, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < cnt; ++i) { md = fmax(md, fabs(a[i].x - b[i].x) + fabs(a[i].y - b[i].y)); mx = fmax(mx, fabs(a[i].x) + fabs(a[i].y)); }
        printf("   check: max |diff| %.3e of max %.3e\n", md, mx);
    }
    (void)hipFree(dfr); (void)hipFree(dt1); (void)hipFree(fact); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(crow); (void)hipFree(crow2);
}

int main() {
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_apply2p, 256, 0));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k_apply2p));
    printf("k_apply2: %d workgroups / CU, regs %d, LDS %zu\n", occ, fa.numRegs, fa.sharedSizeBytes);
    Case cases[] = { {"tiny", 2, 40, 100, 2}, {"top", 3, 1024, 2048, 16}, {"mid", 48, 128, 1024, 16}, {"mid2", 24, 256, 1536, 16}, {"low", 600, 40, 400, 16}, {"low2", 400, 64, 600, 16} };
    run_case(cases[0], true);
    for (int i = 1; i < 6; ++i) run_case(cases[i], i == 2);
    return 0;
}
