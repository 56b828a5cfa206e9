// Supernodal block forward/back substitution for ALL local contour nodes at
// once: Y_k = T(z_k)^-1 R for k in the local shard, m0 right-hand sides.
// GPU side of the reference's `left_divider` seam (src/feast.jl:61-66, 138-143,
// src/nlfeast.jl:41,49; zgetrs / umfpack_zl_solve in the reference) with the
// accumulation of src/feast.jl:68-70 as a deterministic epilogue kernel.
//
// Block vectors are ROW-major n x ldw (one row = ldw contiguous complex128), so a
// supernode's rows are one contiguous slab and update rows gather as whole 1 KiB
// lines.  The factor panels are streamed from HBM exactly once per sweep and go
// straight into the f64 matrix cores; the right-hand-side tile is staged in LDS.
#include "kernels.hpp"

namespace feasthip {

__device__ inline int lower_bound_dev2(const int* a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

__device__ inline void cmfma2(d4& cr, d4& ci, double ar, double ai, double br, double bi) {
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, ci, 0, 0, 0);
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, ci, 0, 0, 0);
}

// W_k[i,:] = src[perm[c0 + rowperm_k[i]], :]   (fill-reducing + pivot permutation folded into one gather)
__global__ void k_gather_rhs(const int* __restrict__ perm, const int* __restrict__ c0_of_row, int n,
                             const cz* __restrict__ src, int ldw, cz* W, int64_t ws, const int* __restrict__ rowperm) {
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    int k = blockIdx.y, lane = threadIdx.x & 63;
    if (i >= n) return;
    int c0 = c0_of_row[i];
    int srow = perm[c0 + rowperm[(int64_t)k * n + i]];
    const cz* s = src + (int64_t)srow * ldw;
    cz* d = W + (int64_t)k * ws + (int64_t)i * ldw;
    for (int j = lane; j < ldw; j += 64) d[j] = s[j];
}

void launch_gather_rhs(hipStream_t st, const DevSym& S, const cz* src, int ldw, cz* W, int64_t ws,
                       const int* rowperm, int K) {
    hipLaunchKernelGGL(k_gather_rhs, dim3((S.n + 3) / 4, K), dim3(256), 0, st, S.perm, S.c0_of_row, S.n, src, ldw, W, ws, rowperm);
}

// Forward sweep, assembly of the children's contribution rows into the parent:
// rows that map into the pivot block go to W (at their pivoted position), the
// rest into the parent's own contribution rows.  A workgroup owns 64 destination
// rows and pulls from the children in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void k_fwd_assemble(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                      const int* __restrict__ rel, const int* __restrict__ child_idx,
                                                      cz* W, int64_t ws, int ldw, cz* crp, int64_t crsp,
                                                      const cz* __restrict__ crc, int64_t crsc,
                                                      const int* __restrict__ rowinv, int n) {
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int p = f.p, r0 = t.a * 64;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int ci = 0; ci < f.nchild; ++ci) {
        FrontD g = fr[child_idx[f.child_ptr + ci]];
        const int* rc = rel + g.sptr;
        int ia = lower_bound_dev2(rc, g.u, r0), ib = lower_bound_dev2(rc, g.u, r0 + 64);
        for (int i = ia + wave; i < ib; i += 4) {
            int dst = rc[i];
            const cz* s = crc + (int64_t)k * crsc + (g.crow + i) * (int64_t)ldw;
            cz* d;
            if (dst < p) d = W + (int64_t)k * ws + (int64_t)(f.c0 + rowinv[(int64_t)k * n + f.c0 + dst]) * ldw;
            else d = crp + (int64_t)k * crsp + (f.crow + dst - p) * (int64_t)ldw;
            for (int j = lane; j < ldw; j += 64) { cz a = d[j], b = s[j]; a.x += b.x; a.y += b.y; d[j] = a; }
        }
        __syncthreads();
    }
}

void launch_fwd_assemble(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* W, int64_t ws, int ldw,
                         cz* crow_parent, int64_t crs_parent, const cz* crow_child, int64_t crs_child,
                         const int* rowinv, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_fwd_assemble, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, S.rel, S.child_idx, W, ws, ldw,
                       crow_parent, crs_parent, crow_child, crs_child, rowinv, S.n);
}

// Triangular solve with one SOLVE_NB x SOLVE_NB diagonal block (forward: unit lower,
// backward: upper), all right-hand sides.  One workgroup per (front, node); lanes
// are right-hand-side columns, the four waves split the rows of each elimination step.
template <int MODE>
__global__ __launch_bounds__(256) void k_diag_solve(const FrontD* __restrict__ fr, const int* __restrict__ list, int k0,
                                                     const cz* __restrict__ fact, int64_t fs, cz* W, int64_t ws, int ldw) {
    __shared__ cz Ts[SOLVE_NB * SOLVE_NB];
    __shared__ cz Ws[SOLVE_NB][LDS_COLS];
    __shared__ cz Dinv[SOLVE_NB];
    FrontD f = fr[list[blockIdx.x]];
    int k = blockIdx.y;
    int p = f.p, ld = f.p + f.u;
    int b = min(SOLVE_NB, p - k0);
    const cz* Lp = fact + (int64_t)k * fs + f.lp;
    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int idx = tid; idx < b * b; idx += 256) {
        int i = idx % b, kk = idx / b;
        Ts[i + kk * SOLVE_NB] = Lp[(k0 + i) + (int64_t)(k0 + kk) * ld];
    }
    __syncthreads();
    if (MODE == 1 && tid < b) Dinv[tid] = cz_inv(Ts[tid + tid * SOLVE_NB]);
    cz* Wb = W + (int64_t)k * ws + (int64_t)(f.c0 + k0) * ldw;
    for (int cg = 0; cg < ldw; cg += LDS_COLS) {
        int col = cg + lane;
        bool cok = col < ldw;
        __syncthreads();
        for (int r = wave; r < b; r += 4) Ws[r][lane] = cok ? Wb[(int64_t)r * ldw + col] : cz_make(0, 0);
        if (MODE == 0) {
            for (int i = 0; i < b; ++i) {
                __syncthreads();
                cz wi = Ws[i][lane];
                for (int r = i + 1 + wave; r < b; r += 4) { cz v = Ws[r][lane]; cz_msub(v, Ts[r + i * SOLVE_NB], wi); Ws[r][lane] = v; }
            }
            __syncthreads();
            for (int r = wave; r < b; r += 4) if (cok) Wb[(int64_t)r * ldw + col] = Ws[r][lane];
        } else {
            for (int i = b - 1; i >= 0; --i) {
                __syncthreads();
                cz xi = cz_mul(Ws[i][lane], Dinv[i]);
                if ((i & 3) == wave && cok) Wb[(int64_t)i * ldw + col] = xi;
                for (int r = wave; r < i; r += 4) { cz v = Ws[r][lane]; cz_msub(v, Ts[r + i * SOLVE_NB], xi); Ws[r][lane] = v; }
            }
        }
    }
}

void launch_diag_solve(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, int mode, const cz* fact,
                       int64_t fs, cz* W, int64_t ws, int ldw, int K) {
    if (nlist <= 0) return;
    if (mode == 0) hipLaunchKernelGGL(k_diag_solve<0>, dim3(nlist, K), dim3(256), 0, st, S.fr, list, k0, fact, fs, W, ws, ldw);
    else hipLaunchKernelGGL(k_diag_solve<1>, dim3(nlist, K), dim3(256), 0, st, S.fr, list, k0, fact, fs, W, ws, ldw);
}

// ---------------------------------------------------------------------------
// THE hot kernel: out[rows] -= A[rows x Kd] * V[Kd x 64 cols], A a column-major
// slab of a factor panel streamed once from HBM, V rows of the block vector
// staged in LDS, f64 MFMA.  Workgroup = 64 output rows x 64 columns, wave w owns
// rows [16w, 16w+16): its A fragment is one 16-byte load per lane per k-step
// (16 consecutive rows = 256 B segments), V fragments are conflict-free
// ds_read_b128, and D's layout (row = (l>>4)+4r, col = l&15) writes 16 consecutive
// columns of one block-vector row per lane group.
//   KIND 0  forward: rows below diagonal block k0 of the L panel; rows < p update W,
//           rows >= p the front's contribution rows.      V = y block (contiguous)
//   KIND 1  backward: rows above block k0 (U11 column block).   V = x block
//   KIND 2  backward: W[J] -= U12 * x[struct]                    V = gathered rows
// ---------------------------------------------------------------------------
constexpr int APPLY_KC = 64;

template <int KIND>
__global__ __launch_bounds__(256) void k_panel_apply(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0,
                                                      const int* __restrict__ struct_idx, const cz* __restrict__ fact,
                                                      int64_t fs, cz* W, int64_t ws, int ldw, cz* crow, int64_t crs) {
    __shared__ cz Vs[APPLY_KC][LDS_COLS];
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y, cg = blockIdx.z * LDS_COLS;
    FrontD f = fr[t.f];
    int p = f.p, u = f.u, ld = p + u;
    const cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* Wk = W + (int64_t)k * ws;
    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    int b = min(SOLVE_NB, p - k0);
    const cz* A;
    int64_t lda;
    int rbase, rend, Kd;
    if (KIND == 0) { rbase = k0 + b + t.a * 64; rend = ld; A = Lp + (int64_t)k0 * ld; lda = ld; Kd = b; }
    else if (KIND == 1) { rbase = t.a * 64; rend = k0; A = Lp + (int64_t)k0 * ld; lda = ld; Kd = b; }
    else { rbase = t.a * 64; rend = p; A = fact + (int64_t)k * fs + f.f12; lda = p; Kd = u; }
    int arow = rbase + wave * 16 + li;      // row of A this lane feeds to the MFMA
    bool aok = arow < rend;
    d4 cr[4], ci[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { cr[nt] = (d4){0, 0, 0, 0}; ci[nt] = (d4){0, 0, 0, 0}; }
    const int* sidx = struct_idx + f.sptr;
    for (int kc = 0; kc < Kd; kc += APPLY_KC) {
        __syncthreads();
        for (int r = wave; r < APPLY_KC; r += 4) {
            int gk = kc + r;
            cz v = cz_make(0, 0);
            if (gk < Kd && cg + lane < ldw) {
                int64_t vrow = (KIND == 2) ? (int64_t)sidx[gk] : (int64_t)(f.c0 + k0 + gk);
                v = Wk[vrow * ldw + cg + lane];
            }
            Vs[r][lane] = v;
        }
        cz av[APPLY_KC / 4];
#pragma unroll
        for (int ks = 0; ks < APPLY_KC / 4; ++ks) {
            int gk = kc + ks * 4 + lk;
            av[ks] = (aok && gk < Kd) ? A[arow + (int64_t)gk * lda] : cz_make(0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < APPLY_KC / 4; ++ks) {
            int kk = ks * 4 + lk;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                cz bv = Vs[kk][nt * 16 + li];
                cmfma2(cr[nt], ci[nt], av[ks].x, av[ks].y, bv.x, bv.y);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int orow = rbase + wave * 16 + lk + 4 * r;
        if (orow >= rend) continue;
        cz* d;
        if (KIND == 0 && orow >= p) d = crow + (int64_t)k * crs + (f.crow + orow - p) * (int64_t)ldw;
        else d = Wk + (int64_t)(f.c0 + orow) * ldw;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = cg + nt * 16 + li;
            if (col < ldw) { cz v = d[col]; v.x -= cr[nt][r]; v.y -= ci[nt][r]; d[col] = v; }
        }
    }
}

void launch_panel_apply(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, int kind, const cz* fact,
                        int64_t fs, cz* W, int64_t ws, int ldw, cz* crow, int64_t crs, int K) {
    if (ntiles <= 0) return;
    dim3 g(ntiles, K, (ldw + LDS_COLS - 1) / LDS_COLS);
    if (kind == 0) hipLaunchKernelGGL(k_panel_apply<0>, g, dim3(256), 0, st, S.fr, tiles, k0, S.struct_idx, fact, fs, W, ws, ldw, crow, crs);
    else if (kind == 1) hipLaunchKernelGGL(k_panel_apply<1>, g, dim3(256), 0, st, S.fr, tiles, k0, S.struct_idx, fact, fs, W, ws, ldw, crow, crs);
    else hipLaunchKernelGGL(k_panel_apply<2>, g, dim3(256), 0, st, S.fr, tiles, k0, S.struct_idx, fact, fs, W, ws, ldw, crow, crs);
}

// Q0[old,:] = sum_k T_k[i,:] .* e0[k,:]  (and Q1 with e1), T_k = mode ? X[old,:] - W_k[i,:] : W_k[i,:].
// Fixed summation order over k: run-to-run reproducible (src/feast.jl:68-70, src/nlfeast.jl:56-59).
__global__ void k_accumulate(const int* __restrict__ perm, int n, const cz* __restrict__ X, const cz* __restrict__ W,
                             int64_t ws, int ldw, int K, const cz* __restrict__ e0, const cz* __restrict__ e1, int mode,
                             cz* Q0, cz* Q1) {
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (i >= n) return;
    int64_t old = perm[i];
    for (int j = lane; j < ldw; j += 64) {
        cz x = mode ? X[old * ldw + j] : cz_make(0, 0);
        cz q0 = cz_make(0, 0), q1 = cz_make(0, 0);
        for (int k = 0; k < K; ++k) {
            cz w = W[(int64_t)k * ws + (int64_t)i * ldw + j];
            cz tv = mode ? cz_sub(x, w) : w;
            cz_madd(q0, tv, e0[(int64_t)k * ldw + j]);
            if (Q1) cz_madd(q1, tv, e1[(int64_t)k * ldw + j]);
        }
        Q0[old * ldw + j] = q0;
        if (Q1) Q1[old * ldw + j] = q1;
    }
}

void launch_accumulate(hipStream_t st, const DevSym& S, const cz* X, const cz* W, int64_t ws, int ldw, int K,
                       const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1) {
    hipLaunchKernelGGL(k_accumulate, dim3((S.n + 3) / 4), dim3(256), 0, st, S.perm, S.n, X, W, ws, ldw, K, e0, e1, mode, Q0, Q1);
}

}  // namespace feasthip
