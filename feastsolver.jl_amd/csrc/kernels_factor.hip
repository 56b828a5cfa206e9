// Batched multifrontal LU on gfx950: all fronts of one elimination-tree depth
// and ALL contour nodes of this GPU are processed by each launch (the pattern
// of T(z_k) is node-independent, only values differ).  This is the GPU side of
// the reference's `factorizer` seam (src/feast.jl:28-38, 104-114,
// src/nlfeast.jl:17-28; dense zgetrf / UMFPACK in the reference).
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace feasthip {

// Development aid (tools/diag_phases.py, tools/diag_phases_insitu.py): wall-clock stamps (100 MHz) of workgroup (0, 0) of the group kernels at
// their phase boundaries.  Compiled in only with -DFEASTHIP_PROF (make PROF=1): a stamp reads its switch from global memory, a round trip
// of 1-2 us at every phase boundary -- 16 per k_diag_grp launch -- that the shipped kernels must not pay.
#ifdef FEASTHIP_PROF
__device__ unsigned long long g_prof[64];
__device__ int g_prof_on;
__device__ int g_prof_k0;      // stamps of the launch with this first column only (-1: every launch, the last one stays)
__device__ __forceinline__ void prof_stamp(int slot, int k0 = -1) {
    if (g_prof_on && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (g_prof_k0 < 0 || k0 < 0 || k0 == g_prof_k0)) g_prof[slot] = wall_clock64();
}
#else
__device__ __forceinline__ void prof_stamp(int, int = -1) {}
#endif

// workgroup barrier that waits for the wave's LDS operations only (__syncthreads() also waits for every global store in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline int lower_bound_dev(const int* a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// complex accumulate D += A*B on the f64 matrix cores; per-lane operands
// a = A[i = l&15][k = l>>4], b = B[k = l>>4][j = l&15]; D: reg r -> row (l>>4)+4r, col l&15.
__device__ inline void cmfma(d4& cr, d4& ci, double ar, double ai, double br, double bi) {
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, ci, 0, 0, 0);
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, ci, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// assembly of T(z_k) = sum_t coef[k][t] * M_t into the fronts
// ---------------------------------------------------------------------------
__global__ void k_scatter(cz* fact, int64_t fs, const int64_t* __restrict__ dest, const cz* __restrict__ val, int64_t nnz,
                          const cz* __restrict__ coef, int coef_stride, int term) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (e >= nnz) return;
    int64_t d = dest[e];
    if (d < 0) return;
    cz c = coef[(int64_t)k * coef_stride + term];
    cz* p = fact + (int64_t)k * fs + d;
    cz v = *p;
    cz_madd(v, c, val[e]);
    *p = v;
}

void launch_scatter(hipStream_t st, cz* fact, int64_t fs, const int64_t* dest, const cz* val, int64_t nnz,
                    const cz* coef, int coef_stride, int term, int K) {
    if (nnz <= 0 || K <= 0) return;
    dim3 g((unsigned)((nnz + 255) / 256), K);
    hipLaunchKernelGGL(k_scatter, g, dim3(256), 0, st, fact, fs, dest, val, nnz, coef, coef_stride, term);
}

__global__ void k_dense_assemble(cz* fact, int64_t fs, int n, const cz* const* mats, int nterms, const cz* __restrict__ coef) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    int64_t nn = (int64_t)n * n;
    if (e >= nn) return;
    int i = (int)(e % n), j = (int)(e / n);
    cz acc = cz_make(0, 0);
    for (int t = 0; t < nterms; ++t) {
        cz c = coef[k * nterms + t];
        const cz* M = mats[t];
        if (M) cz_madd(acc, c, M[e]);
        else if (i == j) acc = cz_add(acc, c);
    }
    fact[(int64_t)k * fs + e] = acc;
}

void launch_dense_assemble(hipStream_t st, cz* fact, int64_t fs, int n, const cz* const* mats, int nterms,
                           const cz* coef, int K) {
    int64_t nn = (int64_t)n * n;
    dim3 g((unsigned)((nn + 255) / 256), K);
    hipLaunchKernelGGL(k_dense_assemble, g, dim3(256), 0, st, fact, fs, n, mats, nterms, coef);
}

// Import of a caller's dense matrix (feasthip_set_dense): the raw host bytes are uploaded as they are and this kernel converts the
// element type (Float64 / ComplexF64 / Int64 / ComplexF32) and the storage order (row-major callers: numpy) into the complex128
// column-major copy the engine keeps, and collects max |a_ij| and "has an imaginary part" on the way; k_dense_symcheck then
// compares the copy with its transpose.  props: [0] bits of max |a| (non-negative doubles order like their bit patterns),
// [1] != 0: not real, [2] != 0: not equal to its transpose, [3] != 0: not equal to its conjugate transpose off the diagonal, [4] != 0: a diagonal entry with an imaginary part.  raw == out is allowed for complex128 column-major input.
// (The first version did all of this on one host core before the upload: 1 s per 4096 x 4096 matrix.)
__global__ __launch_bounds__(256) void k_dense_import(const void* raw, int type, int row_major, int n, cz* out, unsigned long long* props) {
    __shared__ cz tile[32][33];
    const int bi = blockIdx.x * 32, bj = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double mx = 0.0;
    int notreal = 0;
    for (int q = ty; q < 32; q += 8) {
        const int i = row_major ? bi + q : bi + tx, j = row_major ? bj + tx : bj + q;
        cz v = cz_make(0, 0);
        if (i < n && j < n) {
            const int64_t idx = row_major ? (int64_t)i * n + j : i + (int64_t)j * n;
            if (type == 0) v = cz_make(static_cast<const double*>(raw)[idx], 0.0);
            else if (type == 1) v = static_cast<const cz*>(raw)[idx];
            else if (type == 2) v = cz_make((double)static_cast<const long long*>(raw)[idx], 0.0);
            else { const cf w = static_cast<const cf*>(raw)[idx]; v = cz_make(w.x, w.y); }
            mx = fmax(mx, hypot(v.x, v.y));
            notreal |= (v.y != 0.0);
        }
        tile[i - bi][j - bj] = v;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int i = bi + tx, j = bj + q;
        if (i < n && j < n) out[i + (int64_t)j * n] = tile[tx][q];
    }
    for (int off = 32; off > 0; off >>= 1) { mx = fmax(mx, __shfl_xor(mx, off)); notreal |= __shfl_xor(notreal, off); }
    if ((threadIdx.x & 63) == 0) {
        if (mx > 0.0) atomicMax(props, (unsigned long long)__double_as_longlong(mx));
        if (notreal) atomicOr(props + 1, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_dense_symcheck(int n, const cz* __restrict__ A, unsigned long long* props) {
    const int bi = blockIdx.x * 32, bj = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (bi < bj) return;                    // the tile pairs on and below the diagonal cover every (i, j), i > j
    int bad = 0;
    for (int q = ty; q < 32; q += 8) {
        const int i = bi + tx, j = bj + q;
        if (i < n && j < n && i > j) {
            const cz a = A[i + (int64_t)j * n], b = A[j + (int64_t)i * n];
            if (!(a.x == b.x && a.y == b.y)) bad |= 1;
            if (!(a.x == b.x && a.y == -b.y)) bad |= 2;
        }
        if (i < n && i == j && A[i + (int64_t)j * n].y != 0.0) bad |= 4;
    }
    if (__any(bad & 1) && (threadIdx.x & 63) == 0) atomicOr(props + 2, 1ull);
    if (__any(bad & 2) && (threadIdx.x & 63) == 0) atomicOr(props + 3, 1ull);
    if (__any(bad & 4) && (threadIdx.x & 63) == 0) atomicOr(props + 4, 1ull);
}

void launch_dense_import(hipStream_t st, const void* raw, int type, int row_major, int n, cz* out, unsigned long long* props) {
    const dim3 g((n + 31) / 32, (n + 31) / 32);
    hipLaunchKernelGGL(k_dense_import, g, dim3(256), 0, st, raw, type, row_major, n, out, props);
    hipLaunchKernelGGL(k_dense_symcheck, g, dim3(256), 0, st, n, out, props);
}

// ---------------------------------------------------------------------------
// extend-add: parent front += children contribution blocks.  One workgroup owns
// a 64 x 64 tile of the parent front and pulls from every child in a fixed
// order, so the sum is deterministic and needs no atomics.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_extend_add(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                    const int* __restrict__ rel, const int* __restrict__ child_idx,
                                                    const int* __restrict__ ebp, cz* fact, int64_t fs, cz* cbp, int64_t cbsp,
                                                    const cz* __restrict__ cbc, int64_t cbsc, int phase, int sym) {
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int p = f.p, u = f.u, ld = p + u;
    int r0 = t.a * 64, q0 = t.b * 64;
    // symmetric elimination: the U12 panel and the tiles strictly above the diagonal of the pivot block are never read; of a contribution
    // block only the entries on and below its diagonal exist (round 5: the products no longer store the mirror image, every reader
    // takes entry (i, j) with i < j from (j, i))
    if (sym && phase == 0 && (q0 >= p || r0 + 64 <= q0)) return;
    if (sym && phase == 1 && q0 > r0) return;
    // phase 0 (before the front is factorised): targets in the L / U12 panels;
    // phase 1 (after the Schur complement has been WRITTEN, CB = -L21 U12): targets in the contribution block
    if (phase == 0 ? (r0 >= p && q0 >= p) : (r0 + 64 <= p || q0 + 64 <= p)) return;
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* F12 = fact + (int64_t)k * fs + f.f12;
    cz* CB = cbp + (int64_t)k * cbsp + f.cb;
    for (int ci = 0; ci < f.nchild; ++ci) {
        FrontD g = fr[child_idx[f.child_ptr + ci]];
        const int* rc = rel + g.sptr;
        int uc = g.u;
        // the child's rows / columns that land in this tile (host-made break points instead of four binary searches over rel)
        const int* bp = ebp + g.ebp;
        int ia = bp[t.a], ib = bp[t.a + 1], ja = bp[t.b], jb = bp[t.b + 1];
        int nr = ib - ia, nc = jb - ja;
        if (nr > 0 && nc > 0) {
            const cz* src = cbc + (int64_t)k * cbsc + g.cb;
            for (int idx = threadIdx.x; idx < nr * nc; idx += 256) {
                int i = ia + idx % nr, j = ja + idx / nr;
                cz v = sym ? src[max(i, j) + (int64_t)min(i, j) * uc] : src[i + (int64_t)j * uc];
                int r = rc[i], q = rc[j];
                if ((r >= p && q >= p) != (phase == 1)) continue;
                if (sym && phase == 1 && r < q) continue;
                cz* d;
                if (q < p) d = Lp + r + (int64_t)q * ld;
                else if (r < p) d = F12 + r + (int64_t)(q - p) * p;
                else d = CB + (r - p) + (int64_t)(q - p) * u;
                cz o = *d;
                o.x += v.x; o.y += v.y;
                *d = o;
            }
        }
        __syncthreads();
    }
}

void launch_extend_add(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs,
                       cz* cb_parent, int64_t cbs_parent, const cz* cb_child, int64_t cbs_child, int K, int phase, int sym) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_extend_add, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, S.rel, S.child_idx, S.ebp, fact, fs,
                       cb_parent, cbs_parent, cb_child, cbs_child, phase, sym);
}

// ---------------------------------------------------------------------------
// panel factorisation with partial pivoting restricted to the fully summed
// rows [k0+j, p) of the front.  One workgroup per (front, node).
// ---------------------------------------------------------------------------
constexpr int PANEL_NT_SMALL = 256, PANEL_NT_TALL = 1024;      // threads per panel workgroup (tall panels: the row loops are the work)
constexpr int PANEL_SB = 8;      // sub-panel width: columns right of the sub-panel are touched once per 8 pivots

template <int PANEL_NT>
__global__ __launch_bounds__(PANEL_NT) void k_panel(const FrontD* __restrict__ fr, const int* __restrict__ list, int k0,
                                                     cz* fact, int64_t fs, int* piv, int npiv,
                                                     const double* __restrict__ tiny, int* nperturb) {
    __shared__ double s_val[PANEL_NT / 64];
    __shared__ int s_idx[PANEL_NT / 64];
    __shared__ int s_piv;
    __shared__ cz s_urow[FACT_NB];
    __shared__ cz s_u[PANEL_SB][FACT_NB];
    __shared__ cz s_l[PANEL_SB][PANEL_SB];
    FrontD f = fr[list[blockIdx.x]];
    int k = blockIdx.y;
    int p = f.p, ld = f.p + f.u;
    int b = min(FACT_NB, p - k0);
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double tiny_k = tiny[k];
    for (int sb0 = 0; sb0 < b; sb0 += PANEL_SB) {
        int sbe = min(b, sb0 + PANEL_SB);
        for (int j = sb0; j < sbe; ++j) {
            int kj = k0 + j;
            cz* col = Lp + (int64_t)kj * ld;
            double best = -1.0;
            int bi = kj;
            for (int r = k0 + tid; r < p; r += PANEL_NT) {
                if (r < kj) continue;
                double a = cz_abs1(col[r]);
                if (a > best) { best = a; bi = r; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                double ob = __shfl_xor(best, off);
                int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                double bb = s_val[0];
                int ii = s_idx[0];
                for (int w = 1; w < PANEL_NT / 64; ++w)
                    if (s_val[w] > bb || (s_val[w] == bb && s_idx[w] < ii)) { bb = s_val[w]; ii = s_idx[w]; }
                if (bb < 0) ii = kj;
                s_piv = ii;
                piv[(int64_t)k * npiv + f.c0 + kj] = ii;
            }
            __syncthreads();
            int pr = s_piv;
            if (pr != kj && tid < b) {      // interchange across the whole panel
                cz* c2 = Lp + (int64_t)(k0 + tid) * ld;
                cz a = c2[kj], bb = c2[pr];
                c2[kj] = bb; c2[pr] = a;
            }
            __syncthreads();
            cz pv = col[kj];
            double pa = cz_abs1(pv);
            if (!(pa >= tiny_k)) {   // static pivot perturbation (also catches NaN)
                pv = (pa > 0.0) ? cz_make(pv.x * (tiny_k / pa), pv.y * (tiny_k / pa)) : cz_make(tiny_k, 0.0);
                if (tid == 0) { col[kj] = pv; atomicAdd(nperturb + k, 1); }
            }
            cz inv = cz_inv(pv);
            if (tid > j && tid < sbe) s_urow[tid] = Lp[(int64_t)(k0 + tid) * ld + kj];
            __syncthreads();
            for (int r = k0 + tid; r < ld; r += PANEL_NT) {     // scale + rank-1 update inside the sub-panel only
                if (r <= kj) continue;
                cz l = cz_mul(col[r], inv);
                col[r] = l;
                for (int jj = j + 1; jj < sbe; ++jj) {
                    cz* d = Lp + (int64_t)(k0 + jj) * ld + r;
                    cz v = *d;
                    cz_msub(v, l, s_urow[jj]);
                    *d = v;
                }
            }
            __syncthreads();
        }
        int nt = b - sbe;                 // trailing panel columns
        if (nt <= 0) continue;
        int sw = sbe - sb0;
        // (a) U rows of the trailing columns: forward substitution with the unit-lower sw x sw block
        for (int idx = tid; idx < sw * sw; idx += PANEL_NT) { int i = idx % sw, q = idx / sw; s_l[i][q] = Lp[(int64_t)(k0 + sb0 + q) * ld + k0 + sb0 + i]; }
        __syncthreads();
        if (tid < nt) {
            cz* c2 = Lp + (int64_t)(k0 + sbe + tid) * ld + k0 + sb0;
            cz x[PANEL_SB];
#pragma unroll
            for (int i = 0; i < PANEL_SB; ++i) x[i] = i < sw ? c2[i] : cz_make(0, 0);
#pragma unroll
            for (int i = 1; i < PANEL_SB; ++i)
#pragma unroll
                for (int q = 0; q < i; ++q) if (i < sw) cz_msub(x[i], s_l[i][q], x[q]);
#pragma unroll
            for (int i = 0; i < PANEL_SB; ++i) if (i < sw) { c2[i] = x[i]; s_u[i][tid] = x[i]; }
        }
        __syncthreads();
        // (b) rows below the sub-panel: one pass over the trailing columns with a rank-sw update
        for (int r = k0 + sbe + tid; r < ld; r += PANEL_NT) {
            cz l[PANEL_SB];
#pragma unroll
            for (int q = 0; q < PANEL_SB; ++q) l[q] = q < sw ? Lp[(int64_t)(k0 + sb0 + q) * ld + r] : cz_make(0, 0);
            for (int jj = 0; jj < nt; ++jj) {
                cz* d = Lp + (int64_t)(k0 + sbe + jj) * ld + r;
                cz v = *d;
#pragma unroll
                for (int q = 0; q < PANEL_SB; ++q) cz_msub(v, l[q], s_u[q][jj]);
                *d = v;
            }
        }
        __syncthreads();
    }
}

// Register-resident variant for panels of at most PANEL_RPT * PANEL_NT rows.  A thread keeps its rows (k0 + tid + NT q) of the
// current 4-column sub-panel in registers for the four pivot steps, so that a pivot step costs two barriers and no global
// round trip: the pivot search reads registers, the two interchanged rows travel through LDS (the pivot row doubles as the
// broadcast U row of the rank-1 update), and the columns are written back once per sub-panel.  The generic kernel above
// updated the sub-panel in global memory, one dependent load/store chain per column and pivot (12 us per pivot at 3000 rows).
constexpr int PANEL_RPT_MAX = 4;
// Sub-panel width PANEL_SBR (columns a thread keeps in registers across their pivot steps) x rows per thread PANEL_RPT: 4 x 4 entries with
// 1024 threads (128 registers each) and in the narrow workgroups of levels with many fronts (three waves per SIMD); the 512-thread
// workgroup of a dependency chain (one per matrix, a CU to itself) holds 8 x 4: 48 passes over trailing columns of the panel through
// L2 instead of 112 -- C5 32 LUs 21.3 -> 19.9 ms, C2 95.0 -> 93.8.  (16 x 2, 16 x 1 and 32 x 1 for short panels: the compiler keeps
// the column arrays in scratch memory from 16 columns on, 23.1 ms.)
constexpr int PANEL_SBR_OF(int nt) { return nt == 512 ? 8 : 4; }

template <int PANEL_NT, int PANEL_SBR = PANEL_SBR_OF(PANEL_NT), int PANEL_RPT = PANEL_RPT_MAX>
__global__ __launch_bounds__(PANEL_NT) void k_panel_reg(const FrontD* __restrict__ fr, const int* __restrict__ list, int k0,
                                                         cz* fact, int64_t fs, int* piv, int npiv,
                                                         const double* __restrict__ tiny, int* nperturb) {
    constexpr int NW = PANEL_NT / 64;
    __shared__ double s_val[2][NW];
    __shared__ int s_idx[2][NW];
    __shared__ cz s_row[2][2][PANEL_SBR];      // [parity][pivot row / row j][sub-panel column]
    __shared__ cz s_l[PANEL_SBR][PANEL_SBR];
    __shared__ cz s_u[PANEL_SBR][FACT_NB];
    FrontD f = fr[list[blockIdx.x]];
    int k = blockIdx.y;
    int p = f.p, ld = f.p + f.u;
    int b = min(FACT_NB, p - k0);
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrow = ld - k0, npr = p - k0;       // local row i = r - k0; pivots come from i < npr
    const double tiny_k = tiny[k];
    for (int sb0 = 0; sb0 < b; sb0 += PANEL_SBR) {
        const int sbe = min(b, sb0 + PANEL_SBR), sw = sbe - sb0;
        cz v[PANEL_RPT][PANEL_SBR];
#pragma unroll
        for (int q = 0; q < PANEL_RPT; ++q) {
            int i = tid + PANEL_NT * q;
            bool act = i >= sb0 && i < nrow;
#pragma unroll
            for (int c = 0; c < PANEL_SBR; ++c)
                v[q][c] = (act && c < sw) ? Lp[(int64_t)(k0 + sb0 + c) * ld + k0 + i] : cz_make(0, 0);
        }
#pragma unroll
        for (int jc = 0; jc < PANEL_SBR; ++jc) {
            if (jc >= sw) break;
            const int j = sb0 + jc, par = jc & 1;
            double best = -1.0;
            int bi = j;
#pragma unroll
            for (int q = 0; q < PANEL_RPT; ++q) {
                int i = tid + PANEL_NT * q;
                double a = cz_abs1(v[q][jc]);
                if (i >= j && i < npr && a > best) { best = a; bi = i; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                double ob = __shfl_xor(best, off);
                int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) { s_val[par][wave] = best; s_idx[par][wave] = bi; }
            __syncthreads();
            double bb = s_val[par][0];
            int pr = s_idx[par][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                double ov = s_val[par][w];
                int oi = s_idx[par][w];
                if (ov > bb || (ov == bb && oi < pr)) { bb = ov; pr = oi; }
            }
            if (bb < 0) pr = j;
            if (tid == 0) piv[(int64_t)k * npiv + f.c0 + k0 + j] = k0 + pr;
#pragma unroll
            for (int q = 0; q < PANEL_RPT; ++q) {
                int i = tid + PANEL_NT * q;
                if (i == pr) {
#pragma unroll
                    for (int c = 0; c < PANEL_SBR; ++c) s_row[par][0][c] = v[q][c];
                }
                if (i == j) {
#pragma unroll
                    for (int c = 0; c < PANEL_SBR; ++c) s_row[par][1][c] = v[q][c];
                }
            }
            if (pr != j && tid < b && (tid < sb0 || tid >= sbe)) {      // panel columns outside the sub-panel: interchange in memory
                cz* c2 = Lp + (int64_t)(k0 + tid) * ld + k0;
                cz a = c2[j], c = c2[pr];
                c2[j] = c; c2[pr] = a;
            }
            __syncthreads();
            cz u[PANEL_SBR];
#pragma unroll
            for (int c = 0; c < PANEL_SBR; ++c) u[c] = s_row[par][0][c];
            if (pr != j) {
#pragma unroll
                for (int q = 0; q < PANEL_RPT; ++q) {
                    int i = tid + PANEL_NT * q;
                    if (i == j) {
#pragma unroll
                        for (int c = 0; c < PANEL_SBR; ++c) v[q][c] = u[c];
                    }
                    if (i == pr) {
#pragma unroll
                        for (int c = 0; c < PANEL_SBR; ++c) v[q][c] = s_row[par][1][c];
                    }
                }
            }
            cz pv = u[jc];
            double pa = cz_abs1(pv);
            if (!(pa >= tiny_k)) {   // static pivot perturbation (also catches NaN)
                pv = (pa > 0.0) ? cz_make(pv.x * (tiny_k / pa), pv.y * (tiny_k / pa)) : cz_make(tiny_k, 0.0);
#pragma unroll
                for (int q = 0; q < PANEL_RPT; ++q)
                    if (tid + PANEL_NT * q == j) v[q][jc] = pv;
                if (tid == 0) atomicAdd(nperturb + k, 1);
            }
            cz inv = cz_inv(pv);
#pragma unroll
            for (int q = 0; q < PANEL_RPT; ++q) {
                int i = tid + PANEL_NT * q;
                if (i > j && i < nrow) {
                    cz l = cz_mul(v[q][jc], inv);
                    v[q][jc] = l;
#pragma unroll
                    for (int c = jc + 1; c < PANEL_SBR; ++c) cz_msub(v[q][c], l, u[c]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PANEL_RPT; ++q) {
            int i = tid + PANEL_NT * q;
            if (i >= sb0 && i < nrow) {
#pragma unroll
                for (int c = 0; c < PANEL_SBR; ++c)
                    if (c < sw) Lp[(int64_t)(k0 + sb0 + c) * ld + k0 + i] = v[q][c];
            }
        }
        const int nt = b - sbe;                 // trailing panel columns
        if (nt <= 0) break;
        // (a) U rows of the trailing columns: forward substitution with the unit-lower sw x sw block
#pragma unroll
        for (int q = 0; q < PANEL_RPT; ++q) {
            int i = tid + PANEL_NT * q - sb0;
            if (i >= 0 && i < sw) {
#pragma unroll
                for (int c = 0; c < PANEL_SBR; ++c) s_l[i][c] = v[q][c];
            }
        }
        __syncthreads();
        if (tid < nt) {
            cz* c2 = Lp + (int64_t)(k0 + sbe + tid) * ld + k0 + sb0;
            cz x[PANEL_SBR];
#pragma unroll
            for (int i = 0; i < PANEL_SBR; ++i) x[i] = i < sw ? c2[i] : cz_make(0, 0);
#pragma unroll
            for (int i = 1; i < PANEL_SBR; ++i)
#pragma unroll
                for (int q = 0; q < i; ++q) if (i < sw) cz_msub(x[i], s_l[i][q], x[q]);
#pragma unroll
            for (int i = 0; i < PANEL_SBR; ++i) if (i < sw) { c2[i] = x[i]; s_u[i][tid] = x[i]; }
        }
        __syncthreads();
        // (b) rows below the sub-panel: rank-sw update of the trailing columns with the L values still in registers.  Column by
        // column, the thread's four rows in flight together (16 waves x 4 loads per CU cover the L2 latency; wider batches
        // spill at 128 registers)
        {
            bool on[PANEL_RPT];
#pragma unroll
            for (int q = 0; q < PANEL_RPT; ++q) on[q] = tid + PANEL_NT * q >= sbe && tid + PANEL_NT * q < nrow;
            cz* d0 = Lp + (int64_t)(k0 + sbe) * ld + k0 + tid;
            // two trailing columns per step: eight loads per thread in flight (the loop is a chain of L2 round trips: 112 columns per
            // panel; nt is a multiple of 4 except in the last panel of a front)
            int jj = 0;
            for (; PANEL_NT <= 512 && jj + 2 <= nt; jj += 2) {      // (1024 threads have 128 registers each: one column per step)
                cz t[2][PANEL_RPT];
                cz* dj = d0 + (int64_t)jj * ld;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < PANEL_RPT; ++q) t[h][q] = on[q] ? dj[(int64_t)h * ld + PANEL_NT * q] : cz_make(0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < PANEL_SBR; ++c) {
                        const cz uu = s_u[c][jj + h];
#pragma unroll
                        for (int q = 0; q < PANEL_RPT; ++q) cz_msub(t[h][q], v[q][c], uu);
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < PANEL_RPT; ++q)
                        if (on[q]) dj[(int64_t)h * ld + PANEL_NT * q] = t[h][q];
            }
            for (; jj < nt; ++jj) {
                cz uu[PANEL_SBR], t[PANEL_RPT];
                cz* dj = d0 + (int64_t)jj * ld;
#pragma unroll
                for (int c = 0; c < PANEL_SBR; ++c) uu[c] = s_u[c][jj];
#pragma unroll
                for (int q = 0; q < PANEL_RPT; ++q) t[q] = on[q] ? dj[PANEL_NT * q] : cz_make(0, 0);
#pragma unroll
                for (int q = 0; q < PANEL_RPT; ++q)
#pragma unroll
                    for (int c = 0; c < PANEL_SBR; ++c) cz_msub(t[q], v[q][c], uu[c]);
#pragma unroll
                for (int q = 0; q < PANEL_RPT; ++q)
                    if (on[q]) dj[PANEL_NT * q] = t[q];
            }
        }
        // no barrier: the next sub-panel's loads read this thread's own rows, and s_l / s_u are rewritten only after the
        // barriers of the pivot steps
    }
}

// (Round 5 built two more panel kernels with IMPLICIT interchanges -- rows keep their places during the panel and move once at its end; the
// pivot search breaks ties by the LAPACK position --, right-looking (commit c55dc22) and left-looking with the L columns streamed two ahead
// (commit b61a384).  Both give the pivots of k_panel_reg; neither is faster where it matters: 394 us against 490 for 4096-row panels,
// 163 against 111 us for short ones, C2 97.9 against 95.8 ms.  A panel on ONE CU is a chain of memory round trips)
void launch_panel(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, cz* fact, int64_t fs,
                  int* piv, const double* tiny, int* nperturb, int K, bool tall, int max_rows) {
    if (nlist <= 0) return;
    static const bool generic = getenv("FEASTHIP_PANEL_GENERIC") != nullptr;
    // max_rows: tallest panel (ld - k0) of the launch.  The smallest workgroup whose threads hold PANEL_RPT rows each: a
    // 1024-thread workgroup occupies a whole CU, so levels with many mid-sized fronts (81 fronts x 16 nodes of 1000-1500 rows)
    // otherwise run their panels five CU-rounds deep.
    if (!generic && max_rows <= PANEL_RPT_MAX * 1024) {
        constexpr int PANEL_RPT = PANEL_RPT_MAX;
        // few workgroups (every one gets a CU of its own anyway): the widest workgroup has the shortest pivot step
        const int floor_nt = (int64_t)nlist * K <= 256 ? (tall ? (max_rows <= PANEL_RPT * 512 ? 512 : 1024) : 256) : 64;
#define FEASTHIP_PANEL_CASE(NT) \
        if (max_rows <= PANEL_RPT * NT && NT >= floor_nt) { \
            hipLaunchKernelGGL(k_panel_reg<NT>, dim3(nlist, K), dim3(NT), 0, st, S.fr, list, k0, fact, fs, piv, S.n, tiny, nperturb); \
            return; \
        }
        FEASTHIP_PANEL_CASE(64)
        FEASTHIP_PANEL_CASE(128)
        FEASTHIP_PANEL_CASE(256)
        FEASTHIP_PANEL_CASE(512)
        FEASTHIP_PANEL_CASE(1024)
#undef FEASTHIP_PANEL_CASE
    }
    if (tall) hipLaunchKernelGGL(k_panel<PANEL_NT_TALL>, dim3(nlist, K), dim3(PANEL_NT_TALL), 0, st, S.fr, list, k0, fact, fs, piv, S.n, tiny, nperturb);
    else hipLaunchKernelGGL(k_panel<PANEL_NT_SMALL>, dim3(nlist, K), dim3(PANEL_NT_SMALL), 0, st, S.fr, list, k0, fact, fs, piv, S.n, tiny, nperturb);
}

// ---------------------------------------------------------------------------
// Pivot-light elimination for complex SYMMETRIC pencils (T = T^T: real symmetric / complex symmetric coefficient matrices; the
// shifted matrices of FEAST, z B - A with Im z != 0 and B definite, have a definite imaginary part, so every leading principal
// block and every Schur complement is nonsingular and Gaussian elimination needs no row interchanges).  Safety: the magnitude of
// the multipliers |l_ij| and of the pivots is recorded per node (growth[], as the bit patterns of non-negative doubles) and tiny
// pivots are perturbed and counted exactly like in the pivoted kernels; Engine::factorize falls back to the pivoted path when
// either says the elimination was not harmless.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void growth_max(unsigned long long* g, double v) {
    // wave-wide maximum, one atomic per wave (non-negative doubles order like their bit patterns; NaN counts as +inf)
    if (!(v == v)) v = __longlong_as_double(0x7ff0000000000000ll);
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax(g, (unsigned long long)__double_as_longlong(v));
}

// ---------------------------------------------------------------------------
// Pivot-light elimination, one GROUP of up to 128 pivot columns per step (a first version worked per 32-column panel: three dependent
// launches per panel -- diagonal block, rows below, in-group update -- and on the big fronts that chain, not the matrix cores,
// set the pace: 97 ms for the 8 factorisations of C3 instead of 84).  Per group [k0, k0 + gw):
//   k_diag_grp   one workgroup per (front, node) factorises the gw x gw diagonal block LEFT-looking in 32-column blocks: the rows
//                of block c first get their L entries of the earlier blocks and the Schur update of block (c, c) (grp_substitute),
//                then two waves eliminate block (c, c) in registers: wave 0 the block itself (L\U in place), wave 1 the identity
//                with the same multipliers, which yields inv(L_cc) at no extra time; W_c = inv(U_cc) = inv(L_cc)^T D^-1 goes to a
//                scratch slot.
//   k_lscale_grp every wave takes 16 rows below the block and solves X U11 = A for them over the whole group on the matrix cores,
//                X_c = (A_c - sum_{k < 32 c} Z_k L11[32 c + j][k]) W_c with Z = X D (U11 = D L11^T), without any exchange between
//                waves: the D-layout of one product (lane (li, lk) holds columns lk + 4 s of row li) IS the B-operand layout of
//                the next when the k-steps are taken with stride 4.  It writes the L rows and, transposed, the U rows (U = D L^T).
// After that the rows below and right of the group see ONE product of depth gw (k_gemm_nn), nothing else.
// ---------------------------------------------------------------------------
constexpr int GRP_W = 128;
constexpr int WSLOT = 4 * 1024 + GRP_W;      // scratch entries per (node, front): W_c of the four 32-column blocks, then the group's pivots d

// 16 rows (row = rbase + li < rend) of the group [k0, k0 + gw).  upd_only = false: substitution over the column blocks c < ncb.
// upd_only = true: the Schur update of column block ncb of these rows, A -= Z L11^T written back in place, with Z = (L D) of the
// rows re-read from memory -- the L11 operand of that update are the L entries of ALL rows of block ncb, i.e. also what the other
// wave of k_diag_grp has just computed, so the caller separates the two passes by a workgroup barrier.
// Wt: W blocks, element [c][k][j], then the pivots.
struct GrowthLU { double l, u; };       // max |l_ij| and max |u_ij| a substitution wrote

__device__ __forceinline__ GrowthLU grp_substitute(cz* __restrict__ Lp, cz* __restrict__ F12, int ld, int p, int k0, int gw, int ncb, bool upd_only,
                                                   int rbase, int rend, const cz* __restrict__ Wt) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int row = rbase + li;
    const bool rok = row < rend;
    const int row_c = min(row, rend - 1);
    const cz* D0 = Lp + k0 + (int64_t)k0 * ld;          // diagonal block of the group: L11 strictly below, U11 on and above the diagonal
    cz* ucol = (row_c < p) ? Lp + k0 + (int64_t)row_c * ld : F12 + k0 + (int64_t)(row_c - p) * p;
    cz z[8][4];                                          // Z = X D, tile t (16 columns), k-step s: column 16 t + lk + 4 s
    double lmax = 0.0, umax = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c > ncb || (c == ncb && !upd_only) || 32 * c >= gw) break;
        const bool solve = c < ncb;
        if (upd_only && solve) {          // Z of this block from the stored L entries of these rows
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int col = 32 * c + 16 * tl + lk + 4 * sidx;          // (< 32 ncb <= gw)
                    const cz x = Lp[row_c + (int64_t)(k0 + col) * ld];
                    z[2 * c + tl][sidx] = rok ? cz_mul(x, Wt[4096 + col]) : cz_make(0, 0);
                }
            continue;
        }
        cz y[2][4];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int col = 32 * c + 16 * tl + lk + 4 * sidx;
                cz v = Lp[row_c + (int64_t)(k0 + min(col, gw - 1)) * ld];
                const bool keep = rok && col < gw;
                y[tl][sidx].x = keep ? v.x : 0.0; y[tl][sidx].y = keep ? v.y : 0.0;
            }
        if (c > 0) {
            d4 t1[2], t2[2], t3[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) { t1[tl] = (d4){0, 0, 0, 0}; t2[tl] = (d4){0, 0, 0, 0}; t3[tl] = (d4){0, 0, 0, 0}; }
#pragma unroll
            for (int tp = 0; tp < 2 * c; ++tp)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int kk = 16 * tp + lk + 4 * sidx;
                    const cz zb = z[tp][sidx];
                    const double zs = zb.x + zb.y;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        const int jrow = 32 * c + 16 * tl + li;
                        cz a = D0[min(jrow, gw - 1) + (int64_t)kk * ld];
                        if (jrow >= gw) { a.x = 0.0; a.y = 0.0; }
                        t1[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, zb.x, t1[tl], 0, 0, 0);
                        t2[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, zb.y, t2[tl], 0, 0, 0);
                        t3[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, zs, t3[tl], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    y[tl][sidx].x -= t1[tl][sidx] - t2[tl][sidx];
                    y[tl][sidx].y -= t3[tl][sidx] - t1[tl][sidx] - t2[tl][sidx];
                }
        }
        if (!solve) {       // the updated block (ncb, ncb) of these rows goes back in place
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int col = 32 * c + 16 * tl + lk + 4 * sidx;
                    if (rok && col < gw) Lp[row + (int64_t)(k0 + col) * ld] = y[tl][sidx];
                }
            break;
        }
        // X_c = Y_c W_c
        d4 u1[2], u2[2], u3[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) { u1[tl] = (d4){0, 0, 0, 0}; u2[tl] = (d4){0, 0, 0, 0}; u3[tl] = (d4){0, 0, 0, 0}; }
        const cz* Wc = Wt + c * 1024;
#pragma unroll
        for (int ts = 0; ts < 2; ++ts)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int kk = 16 * ts + lk + 4 * sidx;
                const cz yb = y[ts][sidx];
                const double ys = yb.x + yb.y;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const cz a = Wc[kk * 32 + 16 * tl + li];
                    u1[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, yb.x, u1[tl], 0, 0, 0);
                    u2[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, yb.y, u2[tl], 0, 0, 0);
                    u3[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, ys, u3[tl], 0, 0, 0);
                }
            }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int col = 32 * c + 16 * tl + lk + 4 * sidx;
                const bool cok = col < gw;
                const cz x = cz_make(u1[tl][sidx] - u2[tl][sidx], u3[tl][sidx] - u1[tl][sidx] - u2[tl][sidx]);
                cz dk = Wt[4096 + min(col, gw - 1)];        // pivot d_col (the diagonal block itself holds the blocks' inverses by now)
                if (!cok) { dk.x = 0.0; dk.y = 0.0; }
                const cz zz = cz_mul(x, dk);
                z[2 * c + tl][sidx] = zz;
                if (rok && cok) {
                    Lp[row + (int64_t)(k0 + col) * ld] = x;
                    ucol[col] = zz;                       // U[k0 + col][row] = d_col l_(row, col)
                    lmax = fmax(lmax, fabs(x.x) + fabs(x.y));
                    umax = fmax(umax, fabs(zz.x) + fabs(zz.y));
                }
            }
    }
    return GrowthLU{lmax, umax};
}

// reciprocal of a pivot: 1 / (x + i y) = (x - i y) / (x^2 + y^2) with one hardware reciprocal and two Newton steps (the IEEE
// division sequences of Smith's algorithm are three times as long, and this sits on the critical path of every elimination step);
// magnitudes whose square leaves the double range take the careful route
__device__ __forceinline__ cz cz_inv_pivot(cz a) {
    const double den = a.x * a.x + a.y * a.y;
    if (!(den > 1e-280 && den < 1e280)) return cz_inv(a);
    double r = __builtin_amdgcn_rcp(den);
    r = r * (2.0 - den * r);
    r = r * (2.0 - den * r);
    return cz_make(a.x * r, -a.y * r);
}

// Elimination of one b x b (b <= 32) block at D (leading dimension ld) by 512 threads: thread (i, q) = (tid & 31, tid >> 5) keeps
// the entries (i, q) and (i, q + 16) of the block AND of an identity that receives the same row operations (it ends as inv(L)) in
// registers.  A step publishes what the next one needs -- the pivot column, the pivot rows of both matrices, the reciprocal of the
// pivot -- through LDS: one barrier per step, no register indexing, a rolled loop.  (The earlier one-wave version kept whole rows
// in registers and paid 2 us per step in selects, lane reads and double-precision issue slots.)
// On return the block holds ITS INVERSES in place -- inv(L) strictly below the diagonal (unit diagonal implied), inv(U) = inv(L)^T D^-1
// on and above it: the first level of the solve-ready form (k_couple32 couples two such blocks into what k_block_inverse makes of a
// 64 x 64 block of L\U), nothing after this kernel reads L\U of a diagonal 32 x 32 block -- Wc[kk * 32 + j] = inv(U)[kk][j], and
// dout[j] = the pivots d_j.
struct LuShared {
    cz col[2][FACT_NB], row[2][FACT_NB], erow[2][FACT_NB], pinv[2], dinv[FACT_NB];
};

template <int NQ = 16>      // NQ = threads / 32: thread (i, q) keeps the columns q, q + NQ, .. of row i (2 entries with 512 threads, 4 with 256)
__device__ __forceinline__ void lu32_block(cz* __restrict__ D, const int ld, const int b, const double tiny_k, LuShared& sh, cz* __restrict__ Wc,
                                           cz* __restrict__ dout, double& lmax, double& dmax, int& npert) {
    constexpr int NC = FACT_NB / NQ;
    const int i = threadIdx.x & 31, q = threadIdx.x >> 5;
    const bool rok = i < b;
    cz a[NC], e[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        const int cm = q + NQ * m;
        a[m] = (rok && cm < b) ? D[i + (int64_t)cm * ld] : cz_make(i == cm ? 1.0 : 0.0, 0.0);
        e[m] = cz_make(i == cm ? 1.0 : 0.0, 0.0);
    }
    auto publish = [&](int jn, int nbuf) {       // what step jn reads: column jn, row jn of both matrices, the pivot's reciprocal
        const bool mine = (jn % NQ) == q;          // this thread holds column jn (in slot jn / NQ)
#pragma unroll
        for (int m = 0; m < NC; ++m) if (mine && jn / NQ == m) sh.col[nbuf][i] = a[m];
        if (i == jn) {
            if (mine) {
                cz pv = a[0];
#pragma unroll
                for (int m = 1; m < NC; ++m) if (jn / NQ == m) pv = a[m];
                const double pa = fabs(pv.x) + fabs(pv.y);
                if (jn < b) dmax = fmax(dmax, pa);
                if (!(pa >= tiny_k)) {            // static pivot perturbation (also catches NaN), as in k_panel_reg
                    pv = (pa > 0.0) ? cz_make(pv.x * (tiny_k / pa), pv.y * (tiny_k / pa)) : cz_make(tiny_k, 0.0);
#pragma unroll
                    for (int m = 0; m < NC; ++m) if (jn / NQ == m) a[m] = pv;
                    if (jn < b) ++npert;
                }
                const cz pin = cz_inv_pivot(pv);
                sh.pinv[nbuf] = pin;
                sh.dinv[jn] = pin;
            }
#pragma unroll
            for (int m = 0; m < NC; ++m) { sh.row[nbuf][q + NQ * m] = a[m]; sh.erow[nbuf][q + NQ * m] = e[m]; }
        }
    };
    publish(0, 0);
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < FACT_NB; ++j) {
        const int buf = j & 1;
        if (i > j) {
            const cz l = cz_mul(sh.col[buf][i], sh.pinv[buf]);
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                const int cm = q + NQ * m;
                if (cm > j) cz_msub(a[m], l, sh.row[buf][cm]);
                else { cz_msub(e[m], l, sh.erow[buf][cm]); if (cm == j) { a[m] = l; lmax = fmax(lmax, fabs(l.x) + fabs(l.y)); } }
            }
        }
        if (j + 1 < FACT_NB) publish(j + 1, buf ^ 1);
        __syncthreads();
    }
    // thread (i, c) holds inv(L)[i][c] = inv(U)[c][i] d_i; a diagonal thread holds the pivot d_i, a thread with c > i the entry U[i][c]
    const cz dinv = sh.dinv[i];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        const int cm = q + NQ * m;
        if (rok && cm > i && cm < b) dmax = fmax(dmax, fabs(a[m].x) + fabs(a[m].y));
        const cz w = cz_mul(e[m], dinv);
        Wc[cm * 32 + i] = w;
        if (cm == i) dout[i] = a[m];
        if (rok) {
            if (cm < i) D[i + (int64_t)cm * ld] = e[m];            // inv(L)[i][cm]           (cm < i < b)
            if (cm <= i) D[cm + (int64_t)i * ld] = w;              // inv(U)[cm][i]
        }
    }
}

// Round 5: the same elimination with ONE value per (row, column) slot.  At step j a slot holds the entry of the block for the columns
// c > j and the entry of inv(L) for c <= j (the multiplier numerator turns into -l at step c: e_ic was zero until then and e_cc = 1), so a
// thread keeps 2 values instead of 4, the pivot row is published once instead of twice, the update has no branch on the side of the
// diagonal, and the barriers wait for LDS only.  Same multipliers, pivots and perturbations as lu32_block; the products of the inverse
// are formed in the same order.
struct LuShared2 { cz col[2][FACT_NB], row[2][FACT_NB], pinv[2], dinv[FACT_NB]; };
__device__ __forceinline__ void lu32_block2(cz* __restrict__ D, const int ld, const int b, const double tiny_k, LuShared2& sh, cz* __restrict__ Wc,
                                            cz* __restrict__ dout, double& lmax, double& dmax, int& npert) {
    const int i = threadIdx.x & 31, q = threadIdx.x >> 5, c0 = q, c1 = q + 16;      // 512 threads
    prof_stamp(40);
    const bool rok = i < b;
    cz s0 = (rok && c0 < b) ? D[i + (int64_t)c0 * ld] : cz_make(i == c0 ? 1.0 : 0.0, 0.0);
    cz s1 = (rok && c1 < b) ? D[i + (int64_t)c1 * ld] : cz_make(i == c1 ? 1.0 : 0.0, 0.0);
    auto publish = [&](int jn, int nbuf) {
        if (c0 == jn) sh.col[nbuf][i] = s0;
        if (c1 == jn) sh.col[nbuf][i] = s1;
        if (i == jn) {
            if (c0 == jn || c1 == jn) {
                cz pv = (c0 == jn) ? s0 : s1;
                const double pa = fabs(pv.x) + fabs(pv.y);
                if (jn < b) dmax = fmax(dmax, pa);
                if (__builtin_expect(!(pa >= tiny_k), 0)) {            // static pivot perturbation (also catches NaN), as in k_panel_reg
                    pv = (pa > 0.0) ? cz_make(pv.x * (tiny_k / pa), pv.y * (tiny_k / pa)) : cz_make(tiny_k, 0.0);
                    if (c0 == jn) s0 = pv; else s1 = pv;
                    if (jn < b) ++npert;
                }
                const cz pin = cz_inv_pivot(pv);
                sh.pinv[nbuf] = pin;
                sh.dinv[jn] = pin;
            }
            sh.row[nbuf][c0] = s0; sh.row[nbuf][c1] = s1;
        }
    };
    publish(0, 0);
    lds_barrier();
    prof_stamp(41);
#pragma unroll 1
    for (int j = 0; j < FACT_NB; ++j) {
        const int buf = j & 1;
        if (i > j) {
            const cz l = cz_mul(sh.col[buf][i], sh.pinv[buf]);
            const cz r0 = sh.row[buf][c0], r1 = sh.row[buf][c1];
            if (c0 == j) { s0 = cz_make(-l.x, -l.y); lmax = fmax(lmax, fabs(l.x) + fabs(l.y)); } else cz_msub(s0, l, r0);
            if (c1 == j) { s1 = cz_make(-l.x, -l.y); lmax = fmax(lmax, fabs(l.x) + fabs(l.y)); } else cz_msub(s1, l, r1);
        }
        if (j + 1 < FACT_NB) publish(j + 1, buf ^ 1);
        lds_barrier();
    }
    prof_stamp(42);
    // slot (i, c): c < i  inv(L)[i][c];  c == i  the pivot d_i;  c > i  U[i][c]
    const cz dinv = sh.dinv[i];
    auto finish = [&](int c, cz sv) {
        if (rok && c > i && c < b) dmax = fmax(dmax, fabs(sv.x) + fabs(sv.y));
        const cz e = c < i ? sv : cz_make(c == i ? 1.0 : 0.0, 0.0);
        const cz w = cz_mul(e, dinv);
        Wc[c * 32 + i] = w;
        if (c == i) dout[i] = sv;
        if (rok) {
            if (c < i) D[i + (int64_t)c * ld] = e;              // inv(L)[i][c]           (c < i < b)
            if (c <= i) D[c + (int64_t)i * ld] = w;             // inv(U)[c][i]
        }
    };
    finish(c0, s0);
    finish(c1, s1);
    prof_stamp(43);
}

// (the substitution's registers and the elimination's are never live together; as a call the two are allocated separately)
__device__ __attribute__((noinline)) GrowthLU grp_substitute_call(cz* Lp, cz* F12, int ld, int p, int k0, int gw, int ncb, bool upd_only, int rbase, int rend, const cz* Wt) {
    return grp_substitute(Lp, F12, ld, p, k0, gw, ncb, upd_only, rbase, rend, Wt);
}

constexpr int DIAG_NT = 512;

template <int LUV>
__global__ __launch_bounds__(DIAG_NT) void k_diag_grp(const FrontD* __restrict__ fr, const int* __restrict__ list, int k0,
                                                      cz* fact, int64_t fs, int* piv, int npiv, const double* __restrict__ tiny, int* nperturb,
                                                      unsigned long long* growth, cz* wscr, int64_t wstride) {
    __shared__ LuShared sh;
    __shared__ LuShared2 sh2;
    __builtin_amdgcn_s_setprio(3);             // the few waves of the dependency chain go first wherever they share a SIMD
    FrontD f = fr[list[blockIdx.x]];
    const int k = blockIdx.y;
    const int p = f.p, ld = f.p + f.u;
    const int gw = min(GRP_W, p - k0), nb = (gw + FACT_NB - 1) / FACT_NB;
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* F12 = fact + (int64_t)k * fs + f.f12;
    cz* Wt = wscr + ((int64_t)k * wstride + (int64_t)f.wslot) * WSLOT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double tiny_k = tiny[k];
    double lmax = 0.0, dmax = 0.0;
    int npert = 0;
    prof_stamp(0, k0);
    for (int c = 0; c < nb; ++c) {
        const int b = min(FACT_NB, gw - FACT_NB * c);
        if (c > 0) {
            // rows of block c, 16 per wave, two waves: L entries of the blocks before it; barrier (the update multiplies with the L
            // entries of all 32 rows); update of block (c, c)
            if (wave < 2) {
                const GrowthLU g2 = grp_substitute_call(Lp, F12, ld, p, k0, gw, c, false, k0 + FACT_NB * c + 16 * wave, k0 + FACT_NB * c + b, Wt);
                lmax = fmax(lmax, g2.l); dmax = fmax(dmax, g2.u);
            }
            __syncthreads();
            prof_stamp(1 + 4 * c + 1, k0);
            if (wave < 2) grp_substitute_call(Lp, F12, ld, p, k0, gw, c, true, k0 + FACT_NB * c + 16 * wave, k0 + FACT_NB * c + b, Wt);
            __syncthreads();                   // the block's entries (global memory, same workgroup) are final for the elimination below
        }
        prof_stamp(1 + 4 * c + 2, k0);
        if constexpr (LUV == 2) lu32_block2(Lp + (k0 + FACT_NB * c) + (int64_t)(k0 + FACT_NB * c) * ld, ld, b, tiny_k, sh2, Wt + c * 1024, Wt + 4096 + FACT_NB * c, lmax, dmax, npert);
        else lu32_block(Lp + (k0 + FACT_NB * c) + (int64_t)(k0 + FACT_NB * c) * ld, ld, b, tiny_k, sh, Wt + c * 1024, Wt + 4096 + FACT_NB * c, lmax, dmax, npert);
        if (threadIdx.x < b) piv[(int64_t)k * npiv + f.c0 + k0 + FACT_NB * c + threadIdx.x] = k0 + FACT_NB * c + threadIdx.x;
        __syncthreads();                       // W_c, L\U of block c: visible to the next block's substitution
        prof_stamp(1 + 4 * c + 3, k0);
    }
    if (npert > 0) atomicAdd(nperturb + k, npert);
    growth_max(growth + gridDim.y + k, dmax);
    growth_max(growth + k, lmax);
}

// rows below the group's diagonal block, 64 per workgroup (Tile.a): see grp_substitute
__global__ __launch_bounds__(256) void k_lscale_grp(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0,
                                                    cz* fact, int64_t fs, unsigned long long* growth, const cz* __restrict__ wscr, int64_t wstride) {
    Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    FrontD f = fr[t.f];
    const int p = f.p, ld = p + f.u;
    const int gw = min(GRP_W, p - k0), nb = (gw + FACT_NB - 1) / FACT_NB;
    const int r0 = k0 + gw + t.a * 64 + 16 * (threadIdx.x >> 6);
    if (r0 >= ld) return;                      // (no barriers in this kernel: a wave without rows may leave)
    __builtin_amdgcn_s_setprio(2);             // part of the dependency chain of the big fronts: ahead of the trailing products
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* F12 = fact + (int64_t)k * fs + f.f12;
    const cz* Wt = wscr + ((int64_t)k * wstride + (int64_t)f.wslot) * WSLOT;
    const GrowthLU lm = grp_substitute(Lp, F12, ld, p, k0, gw, nb, false, r0, ld, Wt);
    growth_max(growth + k, lm.l);
    growth_max(growth + 2 * gridDim.y + k, lm.u);      // U = D L^T entries as written: what the backward error of the elimination scales with
}

// The same substitution with its shared operands staged in LDS (round 5).  k_lscale_grp reads the group's L11 entries and the W
// blocks straight from global memory inside its MFMA chain: a dozen dependent round trips per wave, each of which costs several
// microseconds next to the trailing product of the other stream -- 170-380 us per launch on the big fronts of C3 for 13 us of
// matrix-core work per wave.  Here a workgroup (4 waves x 16 rows) stages, per 32-column block c, the block row c of L11 (32 x 32 c)
// and W_c once, with all of the stage's global loads (those and the waves' own A rows) issued together before the barrier: ONE
// round trip per block.  Same products in the same order as grp_substitute: bit-identical L and U.
constexpr int LS2_LROWS = GRP_W - FACT_NB;       // k extent of the largest staged L11 block row (96)
__global__ __launch_bounds__(256, 2) void k_lscale_grp2(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0,
                                                        cz* fact, int64_t fs, unsigned long long* growth, const cz* __restrict__ wscr, int64_t wstride) {
    __shared__ cz Ls[LS2_LROWS * FACT_NB];       // [kk][jj]: L11[32 c + jj][kk], kk < 32 c
    __shared__ cz Ws[FACT_NB * FACT_NB];         // [kk][j]:  W_c
    __shared__ cz Ds[GRP_W];                     // pivots d_col (zero beyond gw)
    const Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    const FrontD f = fr[t.f];
    const int p = f.p, ld = p + f.u;
    const int gw = min(GRP_W, p - k0), nb = (gw + FACT_NB - 1) / FACT_NB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int rbase = k0 + gw + t.a * 64 + 16 * wave, rend = ld;
    __builtin_amdgcn_s_setprio(2);
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* F12 = fact + (int64_t)k * fs + f.f12;
    const cz* Wt = wscr + ((int64_t)k * wstride + (int64_t)f.wslot) * WSLOT;
    const cz* D0 = Lp + k0 + (int64_t)k0 * ld;
    const int row = rbase + li;
    const bool rok = row < rend;                 // (a wave without rows still takes part in the barriers)
    const int row_c = max(k0, min(row, rend - 1));
    cz* ucol = (row_c < p) ? Lp + k0 + (int64_t)row_c * ld : F12 + k0 + (int64_t)(row_c - p) * p;
    if (tid < GRP_W) Ds[tid] = tid < gw ? Wt[4096 + tid] : cz_make(0, 0);
    cz z[6][4];
    double lmax = 0.0, umax = 0.0;
    auto stage = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        // every global load of the stage: the wave's A rows of block c, W_c, the block row c of L11
        cz y[2][4], wpre[4], lpre[c > 0 ? 4 * c : 1];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int col = 32 * c + 16 * tl + lk + 4 * sidx;
                y[tl][sidx] = Lp[row_c + (int64_t)(k0 + min(col, gw - 1)) * ld];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) wpre[q] = Wt[c * 1024 + tid + 256 * q];
#pragma unroll
        for (int q = 0; q < 4 * c; ++q) {
            const int idx = tid + 256 * q, kk = idx >> 5, jj = idx & 31;
            lpre[q] = D0[min(32 * c + jj, gw - 1) + (int64_t)kk * ld];
        }
        __syncthreads();                          // the previous stage's reads of Ls / Ws are done
#pragma unroll
        for (int q = 0; q < 4; ++q) Ws[tid + 256 * q] = wpre[q];
#pragma unroll
        for (int q = 0; q < 4 * c; ++q) {
            const int idx = tid + 256 * q, jj = idx & 31;
            Ls[idx] = (32 * c + jj < gw) ? lpre[q] : cz_make(0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int col = 32 * c + 16 * tl + lk + 4 * sidx;
                const bool keep = rok && col < gw;
                y[tl][sidx].x = keep ? y[tl][sidx].x : 0.0; y[tl][sidx].y = keep ? y[tl][sidx].y : 0.0;
            }
        if constexpr (c > 0) {
            d4 t1[2], t2[2], t3[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) { t1[tl] = (d4){0, 0, 0, 0}; t2[tl] = (d4){0, 0, 0, 0}; t3[tl] = (d4){0, 0, 0, 0}; }
#pragma unroll
            for (int tp = 0; tp < 2 * c; ++tp)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int kk = 16 * tp + lk + 4 * sidx;
                    const cz zb = z[tp][sidx];
                    const double zs = zb.x + zb.y;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        const cz a = Ls[kk * 32 + 16 * tl + li];
                        t1[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, zb.x, t1[tl], 0, 0, 0);
                        t2[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, zb.y, t2[tl], 0, 0, 0);
                        t3[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, zs, t3[tl], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    y[tl][sidx].x -= t1[tl][sidx] - t2[tl][sidx];
                    y[tl][sidx].y -= t3[tl][sidx] - t1[tl][sidx] - t2[tl][sidx];
                }
        }
        d4 u1[2], u2[2], u3[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) { u1[tl] = (d4){0, 0, 0, 0}; u2[tl] = (d4){0, 0, 0, 0}; u3[tl] = (d4){0, 0, 0, 0}; }
#pragma unroll
        for (int ts = 0; ts < 2; ++ts)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int kk = 16 * ts + lk + 4 * sidx;
                const cz yb = y[ts][sidx];
                const double ys = yb.x + yb.y;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const cz a = Ws[kk * 32 + 16 * tl + li];
                    u1[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, yb.x, u1[tl], 0, 0, 0);
                    u2[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, yb.y, u2[tl], 0, 0, 0);
                    u3[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, ys, u3[tl], 0, 0, 0);
                }
            }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int col = 32 * c + 16 * tl + lk + 4 * sidx;
                const bool cok = col < gw;
                const cz x = cz_make(u1[tl][sidx] - u2[tl][sidx], u3[tl][sidx] - u1[tl][sidx] - u2[tl][sidx]);
                const cz zz = cz_mul(x, Ds[col]);
                if constexpr (c < 3) z[2 * c + tl][sidx] = zz;
                if (rok && cok) {
                    Lp[row + (int64_t)(k0 + col) * ld] = x;
                    ucol[col] = zz;                       // U[k0 + col][row] = d_col l_(row, col)
                    lmax = fmax(lmax, fabs(x.x) + fabs(x.y));
                    umax = fmax(umax, fabs(zz.x) + fabs(zz.y));
                }
            }
    };
    stage(std::integral_constant<int, 0>{});
    if (nb > 1) stage(std::integral_constant<int, 1>{});
    if (nb > 2) stage(std::integral_constant<int, 2>{});
    if (nb > 3) stage(std::integral_constant<int, 3>{});
    growth_max(growth + k, lmax);
    growth_max(growth + 2 * gridDim.y + k, umax);
}

void launch_diag_grp(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, cz* fact, int64_t fs, int* piv,
                     const double* tiny, int* nperturb, unsigned long long* growth, cz* wscr, int64_t wstride, int K) {
    if (nlist <= 0) return;
    // FEASTHIP_LU32=1: the round-3 elimination of the 32 x 32 blocks (four values per thread, two published rows).  (A 256-thread form of
    // the one-slot elimination and a k_diag_grp with the group's L rows, W blocks and pivots resident in LDS -- substitutions 33 us instead
    // of 57-78 next to the other stream's product, but only 256 threads for the elimination: 53 us per block against 18 -- were built
    // and measured slower in the factorisation, C3 93.5 / 85.4 against 78.9 ms: commits 3f24d11, c55dc22.)
    const char* e3 = getenv("FEASTHIP_LU32");
    if (e3 && atoi(e3) == 1) hipLaunchKernelGGL(k_diag_grp<1>, dim3(nlist, K), dim3(DIAG_NT), 0, st, S.fr, list, k0, fact, fs, piv, S.n, tiny, nperturb, growth, wscr, wstride);
    else hipLaunchKernelGGL(k_diag_grp<2>, dim3(nlist, K), dim3(DIAG_NT), 0, st, S.fr, list, k0, fact, fs, piv, S.n, tiny, nperturb, growth, wscr, wstride);
}

void launch_lscale_grp(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                       unsigned long long* growth, const cz* wscr, int64_t wstride, int K) {
    if (ntiles <= 0) return;
    const char* e2 = getenv("FEASTHIP_LSCALE2");               // 0: the round-3 kernel (operands from global memory); read per launch (tests switch it)
    const bool staged = !(e2 && atoi(e2) == 0);
    if (staged) hipLaunchKernelGGL(k_lscale_grp2, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, k0, fact, fs, growth, wscr, wstride);
    else hipLaunchKernelGGL(k_lscale_grp, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, k0, fact, fs, growth, wscr, wstride);
}

// ---------------------------------------------------------------------------
// Elimination without interchanges for pencils that are NOT symmetric (complex Hermitian A, B: z B - A has a definite
// "imaginary part" exactly like the real symmetric case, but U != D L^T and T(conj z) != conj(T(z))).  It keeps the schedule and
// every other kernel of the pivoted path (U rows by k_trsm_u / k_group_utrsm, products by k_gemm_nn, solve-ready conversion from
// L\U) and replaces its two latency chains: the one-workgroup-per-front panel with its pivot search becomes
//   k_diag32_gen    the 32 x 32 diagonal block of the panel eliminated in LDS (L\U in place, W = inv(U) to the scratch slot,
//                   identity pivots), one workgroup per (front, node), and
//   k_lscale32_gen  the rows below, one per thread over as many workgroups as there are rows: X = A W,
// and the row interchanges disappear.  Growth check as on the symmetric path (max |l_ij|, max |pivot|, max |u_ij| per node, tiny
// pivots perturbed and counted); Engine::factorize falls back to partial pivoting when it fails.
// ---------------------------------------------------------------------------
// Thread (i, g) = (tid & 31, tid >> 5), tid < 256, keeps the entries (i, g + 8 s), s < 4, of the block in registers; a step publishes
// what the next one needs -- the pivot row, the pivot column, the reciprocal of the pivot -- through LDS (two buffers): one barrier
// per step.  A fifth wave builds W = inv(U) beside the elimination, one column per step: column j of U is final when step j starts
// (it is the published pivot column), and W[r][j] = (delta_rj - sum_{q<j} W[r][q] U[q][j]) / U[j][j] needs nothing else.
constexpr int DIAG32_NT = 320;
__global__ __launch_bounds__(DIAG32_NT) void k_diag32_gen(const FrontD* __restrict__ fr, const int* __restrict__ list, int k0, cz* fact, int64_t fs,
                                                          int* piv, int npiv, const double* __restrict__ tiny, int* nperturb, unsigned long long* growth,
                                                          cz* wscr, int64_t wstride, int K) {
    __shared__ cz A[FACT_NB][FACT_NB + 1];
    __shared__ cz W[FACT_NB][FACT_NB + 1];
    __shared__ cz s_row[2][FACT_NB], s_col[2][FACT_NB], s_pinv[2];
    FrontD f = fr[list[blockIdx.x]];
    const int k = blockIdx.y, tid = threadIdx.x, i = tid & (FACT_NB - 1), g = (tid >> 5) & 7;
    const bool elim = tid < 256, winv = tid >= 256 && tid < 256 + FACT_NB;
    const int p = f.p, ld = f.p + f.u;
    const int b = min(FACT_NB, p - k0);
    cz* D = fact + (int64_t)k * fs + f.lp + k0 + (int64_t)k0 * ld;
    cz v[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const int c = g + 8 * s2;
        v[s2] = (elim && i < b && c < b) ? D[i + (int64_t)c * ld] : cz_make(i == c ? 1.0 : 0.0, 0.0);      // (padding: identity)
    }
    const double tiny_k = tiny[k];
    double lmax = 0.0, dmax = 0.0, umax = 0.0;
    // what step j needs: row j (entries of the threads with i == j), column j (threads with g == j & 7, slot j >> 3), 1 / pivot
    auto publish = [&](int j, int buf) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int c = g + 8 * s2;
            if (i == j) {
                cz val = v[s2];
                if (c == j) {
                    const double pa = cz_abs1(val);
                    if (!(pa >= tiny_k)) {       // static pivot perturbation (also catches NaN), as in the pivoted kernels
                        val = (pa > 0.0) ? cz_make(val.x * (tiny_k / pa), val.y * (tiny_k / pa)) : cz_make(tiny_k, 0.0);
                        v[s2] = val;
                        if (j < b) atomicAdd(nperturb + k, 1);
                    }
                    if (j < b) dmax = fmax(dmax, cz_abs1(val));
                    s_pinv[buf] = cz_inv_pivot(val);
                }
                s_row[buf][c] = val;
            }
            if (c == j) s_col[buf][i] = v[s2];
        }
    };
    if (elim) publish(0, 0);
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < FACT_NB; ++j) {
        const int buf = j & 1;
        if (elim) {
            if (i > j) {
                const cz l = cz_mul(s_col[buf][i], s_pinv[buf]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int c = g + 8 * s2;
                    if (c == j) { v[s2] = l; if (i < b && j < b) lmax = fmax(lmax, cz_abs1(l)); }
                    else if (c > j) cz_msub(v[s2], l, s_row[buf][c]);
                }
            }
            if (j + 1 < FACT_NB) publish(j + 1, buf ^ 1);
        } else if (winv) {
            const int r = i;                      // row r of W; its entries left of column j are in W[r][.] already
            if (r <= j) {
                cz acc = cz_make(r == j ? 1.0 : 0.0, 0.0), acc2 = cz_make(0.0, 0.0), acc3 = acc2, acc4 = acc2;
                int q = r;
                for (; q + 3 < j; q += 4) {       // four independent partial sums: the LDS reads of a group are in flight together
                    const cz w0 = W[r][q], w1 = W[r][q + 1], w2 = W[r][q + 2], w3 = W[r][q + 3];
                    const cz u0 = s_col[buf][q], u1 = s_col[buf][q + 1], u2 = s_col[buf][q + 2], u3 = s_col[buf][q + 3];
                    cz_msub(acc, w0, u0); cz_msub(acc2, w1, u1); cz_msub(acc3, w2, u2); cz_msub(acc4, w3, u3);
                }
                for (; q < j; ++q) cz_msub(acc, W[r][q], s_col[buf][q]);
                W[r][j] = cz_mul(cz_add(cz_add(acc, acc2), cz_add(acc3, acc4)), s_pinv[buf]);
                if (j < b) umax = fmax(umax, cz_abs1(s_col[buf][r]));
            } else W[r][j] = cz_make(0.0, 0.0);
        }
        __syncthreads();
    }
    if (elim) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) A[i][g + 8 * s2] = v[s2];
    }
    __syncthreads();
    cz* Wt = wscr + ((int64_t)k * wstride + (int64_t)f.wslot) * WSLOT;
    for (int e = tid; e < FACT_NB * FACT_NB; e += DIAG32_NT) {
        const int r = e & (FACT_NB - 1), c = e / FACT_NB;
        if (r < b && c < b) D[r + (int64_t)c * ld] = A[r][c];
        Wt[r + FACT_NB * c] = W[r][c];
    }
    if (tid < b) piv[(int64_t)k * npiv + f.c0 + k0 + tid] = k0 + tid;
    growth_max(growth + k, lmax);
    growth_max(growth + K + k, dmax);
    growth_max(growth + 2 * K + k, umax);
}

// rows below the diagonal block of the panel: a workgroup takes 64 of them (tile.a counts 256-row slices -- the list is the step's
// column-slice list, the rows below the block are no more than the columns right of it -- and blockIdx.z the quarter of the slice),
// wave q of it the columns 8 q .. 8 q + 7 of the result: X = A W with the row's 32 entries in registers and W broadcast from LDS
__global__ __launch_bounds__(256) void k_lscale32_gen(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0, cz* fact, int64_t fs,
                                                      unsigned long long* growth, const cz* __restrict__ wscr, int64_t wstride) {
    __shared__ cz W[FACT_NB][FACT_NB];       // W[kk][c]
    Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y, tid = threadIdx.x;
    FrontD f = fr[t.f];
    const int p = f.p, ld = p + f.u;
    const int b = min(FACT_NB, p - k0);
    const int r0 = k0 + b + t.a * 256 + blockIdx.z * 64;
    if (r0 >= ld) return;
    const cz* Wt = wscr + ((int64_t)k * wstride + (int64_t)f.wslot) * WSLOT;
    for (int e = tid; e < FACT_NB * FACT_NB; e += 256) W[e & (FACT_NB - 1)][e / FACT_NB] = Wt[e];      // Wt[i + 32 c] = inv(U)[i][c]
    __syncthreads();
    const int r = r0 + (tid & 63), cq = (tid >> 6) * 8;
    double lmax = 0.0;
    if (r < ld) {
        cz* row = fact + (int64_t)k * fs + f.lp + r + (int64_t)k0 * ld;
        cz x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = cz_make(0, 0);
#pragma unroll 1
        for (int q0 = 0; q0 < FACT_NB; q0 += 8) {       // (rolled: unrolled, the compiler hoists all 256 LDS reads and spills)
            cz a8[8];
#pragma unroll
            for (int u2 = 0; u2 < 8; ++u2) a8[u2] = q0 + u2 < b ? row[(int64_t)(q0 + u2) * ld] : cz_make(0, 0);
#pragma unroll
            for (int u2 = 0; u2 < 8; ++u2)
#pragma unroll
                for (int c = 0; c < 8; ++c) cz_madd(x[c], a8[u2], W[q0 + u2][cq + c]);       // (W is upper triangular: zeros below its diagonal)
        }
        // every wave of the workgroup has read the row before any of them writes its part of it
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (cq + c < b) { row[(int64_t)(cq + c) * ld] = x[c]; lmax = fmax(lmax, cz_abs1(x[c])); }
    } else __syncthreads();
    growth_max(growth + k, lmax);
}

void launch_diag32_gen(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, cz* fact, int64_t fs, int* piv,
                       const double* tiny, int* nperturb, unsigned long long* growth, cz* wscr, int64_t wstride, int K) {
    if (nlist <= 0) return;
    hipLaunchKernelGGL(k_diag32_gen, dim3(nlist, K), dim3(DIAG32_NT), 0, st, S.fr, list, k0, fact, fs, piv, S.n, tiny, nperturb, growth, wscr, wstride, K);
}
void launch_lscale32_gen(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                         unsigned long long* growth, const cz* wscr, int64_t wstride, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_lscale32_gen, dim3(ntiles, K, 4), dim3(256), 0, st, S.fr, tiles, k0, fact, fs, growth, wscr, wstride);
}

// ---------------------------------------------------------------------------
// apply the panel's row interchanges to every other column of the front
// ---------------------------------------------------------------------------
// ColSel picks the columns an elimination step touches in one launch, for the look-ahead schedule of Engine::factorize:
// left columns (the L columns of earlier panels) with llo <= index < lhi, pivot-block columns right of the panel whose
// offset from the first such column lies in [clo, chi), and the U12 columns when u12 is set.
// The b <= 256 interchanges of a panel (or of a group of panels) are composed once per workgroup into one permutation of at
// most 2 b rows: the b rows of the panel's diagonal block and the rows below it that a pivot came from.  A wave then moves a
// whole column at a time: every row it needs is loaded first (the diagonal-block rows as one contiguous run), the values change
// places through LDS, and only the rows that changed are written back.  (The first version walked the interchanges one after
// the other in every column, one thread per column: b dependent pairs of loads and stores, 0.74 ms per group of 128 on the
// 4096 x 4096 fronts of C2 whatever the number of columns, a quarter of that factorisation.)
constexpr int SWAP_MAXB = 256, SWAP_SPLIT = 4;
__global__ __launch_bounds__(256) void k_swap(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0, cz* fact, int64_t fs,
                                              const int* __restrict__ piv, int npiv, ColSel cs, int bw) {
    // bw: FACT_NB, or the width of a group of panels whose interchanges are applied in one pass
    __shared__ int s_row[2 * SWAP_MAXB];      // position -> row (positions [0, b): rows k0 .. k0 + b - 1, then the rows below in order of first use)
    __shared__ int s_src[2 * SWAP_MAXB];      // position -> position whose old content it receives
    __shared__ int s_npos;
    __shared__ cz s_val[4][2 * SWAP_MAXB];
    Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    FrontD f = fr[t.f];
    const int p = f.p, u = f.u, ld = p + u;
    const int b = min(min(bw, SWAP_MAXB), p - k0);
    const int nv = (p - b) + u;
    // a 256-column tile is shared by SWAP_SPLIT workgroups (blockIdx.z): the look-ahead steps select 96 .. 128 columns of one tile, and
    // one workgroup moving them column after column was the longest kernel of the panel stream
    const int v0 = t.a * 256 + blockIdx.z * (256 / SWAP_SPLIT), v1 = min(min(v0 + 256 / SWAP_SPLIT, t.a * 256 + 256), nv);
    if (v0 >= nv) return;
    {   // tile without a selected column (the look-ahead steps pick narrow column ranges)
        const bool left = v0 < k0 && v0 < cs.lhi && min(v1, k0) > cs.llo;
        const bool mid = v1 > k0 && v0 < p - b && max(v0, k0) - k0 < cs.chi && min(v1, p - b) - k0 > cs.clo;
        const bool right = cs.u12 && v1 > p - b;
        if (!left && !mid && !right) return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        const int* pv = piv + (int64_t)k * npiv + f.c0 + k0;
        int pvr[SWAP_MAXB / 64];
#pragma unroll
        for (int q = 0; q < SWAP_MAXB / 64; ++q) pvr[q] = (lane + 64 * q < b) ? pv[lane + 64 * q] : -1;
        // position -> source position, entry q in lane q % 64, register q / 64: the b interchanges are b dependent exchanges of two entries,
        // by lane reads / writes with uniform lane numbers (round 5; through LDS by lane 0 an exchange cost 1 us: 128 us per group of 128
        // columns in EVERY workgroup of the launch, which was most of the 160 us these launches took on the panel stream)
        int srcr[2 * SWAP_MAXB / 64];
#pragma unroll
        for (int q = 0; q < 2 * SWAP_MAXB / 64; ++q) srcr[q] = lane + 64 * q;
        int low[SWAP_MAXB / 64];              // rows below the diagonal block that are part of the permutation: entry i in lane i % 64, register i / 64
#pragma unroll
        for (int q = 0; q < SWAP_MAXB / 64; ++q) low[q] = -1;
        int nlow = 0;
        for (int j = 0; j < b; ++j) {
            int pr = 0;
#pragma unroll
            for (int q = 0; q < SWAP_MAXB / 64; ++q) if ((j >> 6) == q) pr = __builtin_amdgcn_readlane(pvr[q], j & 63);
            if (pr == k0 + j) continue;
            int pos;
            if (pr < k0 + b) pos = pr - k0;
            else {
                pos = -1;
#pragma unroll
                for (int q = 0; q < SWAP_MAXB / 64; ++q) {
                    const unsigned long long m = __ballot(low[q] == pr);
                    if (m) pos = b + 64 * q + __builtin_ctzll(m);
                }
                if (pos < 0) {
                    pos = b + nlow;
#pragma unroll
                    for (int q = 0; q < SWAP_MAXB / 64; ++q) if ((nlow >> 6) == q && lane == (nlow & 63)) low[q] = pr;
                    ++nlow;
                }
            }
            int a = 0, c = 0;
#pragma unroll
            for (int q = 0; q < 2 * SWAP_MAXB / 64; ++q) {
                if ((j >> 6) == q) a = __builtin_amdgcn_readlane(srcr[q], j & 63);
                if ((pos >> 6) == q) c = __builtin_amdgcn_readlane(srcr[q], pos & 63);
            }
#pragma unroll
            for (int q = 0; q < 2 * SWAP_MAXB / 64; ++q) {
                if ((j >> 6) == q && lane == (j & 63)) srcr[q] = c;
                if ((pos >> 6) == q && lane == (pos & 63)) srcr[q] = a;
            }
        }
#pragma unroll
        for (int q = 0; q < 2 * SWAP_MAXB / 64; ++q) s_src[lane + 64 * q] = srcr[q];
        for (int q = lane; q < b; q += 64) s_row[q] = k0 + q;
#pragma unroll
        for (int q = 0; q < SWAP_MAXB / 64; ++q) if (lane + 64 * q < nlow) s_row[b + lane + 64 * q] = low[q];
        if (lane == 0) s_npos = b + nlow;
    }
    __syncthreads();
    const int npos = s_npos;
    if (npos == b) {        // no row below the block took part: maybe nothing moved at all
        bool any = false;
        for (int q = lane; q < b; q += 64) any |= s_src[q] != q;
        if (!__any(any)) return;
    }
    // the selected columns of this tile, compacted (their order does not matter)
    __shared__ int s_cols[256];
    __shared__ int s_ncols;
    if (threadIdx.x == 0) s_ncols = 0;
    __syncthreads();
    {
        const int v = v0 + threadIdx.x;
        bool sel = false;
        if (v < v1) {
            if (v < k0) sel = v >= cs.llo && v < cs.lhi;
            else if (v < p - b) sel = v - k0 >= cs.clo && v - k0 < cs.chi;
            else sel = cs.u12 != 0;
        }
        if (sel) s_cols[atomicAdd(&s_ncols, 1)] = v;
    }
    __syncthreads();
    const int ncols = s_ncols;
    cz* const base = fact + (int64_t)k * fs;
    cz* sv = s_val[wave];
    // A wave moves NC = 8 / ppl columns per step (ppl = positions per lane and column, 1 .. 8): eight loads per lane in flight.
    const int ppl = npos <= 64 ? 1 : npos <= 128 ? 2 : npos <= 256 ? 4 : 8, nc = 8 / ppl;
    int rows[8], srcs[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        const int pos = lane + 64 * (sl & (ppl - 1)), cbase = 64 * ppl * (sl / ppl);
        rows[sl] = pos < npos ? s_row[pos] : -1;
        srcs[sl] = cbase + (pos < npos ? s_src[pos] : pos);
    }
    for (int i0 = wave * nc; i0 < ncols; i0 += 4 * nc) {
        cz* colp[8];
        bool live[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int ci = i0 + sl / ppl;
            live[sl] = ci < ncols;
            const int v = s_cols[min(ci, ncols - 1)];
            colp[sl] = v < k0 ? base + f.lp + (int64_t)v * ld : v < p - b ? base + f.lp + (int64_t)(v + b) * ld : base + f.f12 + (int64_t)(v - (p - b)) * p;
        }
        cz val[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) val[sl] = colp[sl][max(rows[sl], k0)];       // (unused positions: a harmless re-read of row k0)
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) sv[64 * sl + lane] = val[sl];
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's LDS writes are done before its lanes read each other's values
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
            if (live[sl] && rows[sl] >= 0 && srcs[sl] != 64 * sl + lane) colp[sl][rows[sl]] = sv[srcs[sl]];
        __builtin_amdgcn_s_waitcnt(0xc07f);      // ... and read before the next columns overwrite them
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_swap(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                 const int* piv, int K, ColSel cs, int bw) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_swap, dim3(ntiles, K, SWAP_SPLIT), dim3(256), 0, st, S.fr, tiles, k0, fact, fs, piv, S.n, cs, bw);
}

// ---------------------------------------------------------------------------
// U row block: columns right of the panel (inside the pivot block and U12)
// get x <- L11^-1 x, one thread per column.
// ---------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void trsm_sweeps(cz (&x)[FACT_NB], const cz* s_l) {
#pragma unroll 1
    for (int kk = KB; kk < KB + 8; ++kk) {
        // x_kk picked with bit masks: a select chain over x[] is turned back into a dynamically indexed load, which sends
        // the whole array to scratch memory
        long long xr = 0, xi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            long long m = -(long long)(kk == KB + j);
            xr |= m & __double_as_longlong(x[KB + j].x);
            xi |= m & __double_as_longlong(x[KB + j].y);
        }
        cz xk = cz_make(__longlong_as_double(xr), __longlong_as_double(xi));
        const cz* lcol = s_l + kk * FACT_NB;
#pragma unroll
        for (int i = KB + 1; i < FACT_NB; ++i) cz_msub(x[i], lcol[i], xk);
    }
}

__global__ __launch_bounds__(TRSM_COLS) void k_trsm_u(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0, cz* fact, int64_t fs,
                                                     ColSel cs, unsigned long long* umax) {
    // One thread per column, the column's FACT_NB rows in registers (each thread reads 512 contiguous bytes), L11 broadcast
    // from LDS; the substitution is fully unrolled.  The former version kept the slab in LDS as well (50 KB per 64-thread
    // workgroup: three waves per CU, 50 us per workgroup, 1.7 ms per launch on the wide levels of C3).
    __shared__ cz s_l[FACT_NB * FACT_NB];
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int p = f.p, u = f.u, ld = p + u;
    int b = min(FACT_NB, p - k0);
    int nin = p - k0 - b, ncol = nin + u;
    {   // column tile without a selected column
        int v0 = t.a * TRSM_COLS, v1 = v0 + TRSM_COLS;
        bool any = (v0 < nin && v0 < cs.chi && min(v1, nin) > cs.clo) || (cs.u12 && v1 > nin);
        if (!any) return;
    }
    const cz* Lp = fact + (int64_t)k * fs + f.lp;
    for (int idx = threadIdx.x; idx < FACT_NB * FACT_NB; idx += TRSM_COLS) {      // rows / columns beyond b: zero (no coupling)
        int i = idx % FACT_NB, kk = idx / FACT_NB;
        s_l[idx] = (i < b && kk < i) ? Lp[(k0 + i) + (int64_t)(k0 + kk) * ld] : cz_make(0, 0);
    }
    __syncthreads();
    int v = t.a * TRSM_COLS + threadIdx.x;
    bool mine = v < ncol && (v < nin ? (v >= cs.clo && v < cs.chi) : (cs.u12 != 0));
    if (!mine) return;
    cz* col = (v < nin) ? fact + (int64_t)k * fs + f.lp + (int64_t)(k0 + b + v) * ld + k0
                        : fact + (int64_t)k * fs + f.f12 + (int64_t)(v - nin) * p + k0;
    cz x[FACT_NB];
#pragma unroll
    for (int i = 0; i < FACT_NB; ++i) x[i] = i < b ? col[i] : cz_make(0, 0);
    // column sweeps x_i -= L_ik x_k (i > k).  k runs in a real loop (the multiplier x_k is selected from the eight registers of
    // its block of rows, s_l holds zeros on and above the diagonal), so the L reads of one sweep are all that is in flight;
    // with the k loop unrolled too the compiler hoists all 496 LDS reads and spills 1600 registers.
    trsm_sweeps<0>(x, s_l);
    trsm_sweeps<8>(x, s_l);
    trsm_sweeps<16>(x, s_l);
    trsm_sweeps<24>(x, s_l);
#pragma unroll
    for (int i = 1; i < FACT_NB; ++i)
        if (i < b) col[i] = x[i];
    if (umax) {      // elimination without interchanges: the size of the U entries is part of its growth check
        double m = 0.0;
#pragma unroll
        for (int i = 0; i < FACT_NB; ++i) if (i < b) m = fmax(m, cz_abs1(x[i]));
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(umax + k, (unsigned long long)__double_as_longlong(m == m ? m : __longlong_as_double(0x7ff0000000000000ll)));
    }
}

void launch_trsm_u(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs, int K, ColSel cs, unsigned long long* umax) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_trsm_u, dim3(ntiles, K), dim3(TRSM_COLS), 0, st, S.fr, tiles, k0, fact, fs, cs, umax);
}

// ---------------------------------------------------------------------------
// U rows of a whole panel group: X <- inv(L11) X for the gw <= GROUP_ROWS rows [k0, k0 + gw) of the columns right of the
// group (pivot-block columns by offset from column k0 + gw, then the U12 columns), L11 the group's unit lower triangle.
// One workgroup per 32-column tile.  Eight consecutive lanes share a column and hold its rows r = (lane & 7) + 8 m
// (m < 16) in registers, so a column's row k reaches the seven other lanes with one shuffle and the workgroup needs no
// barrier inside the 32 elimination steps of a phase; the phase's 32 columns of L11 are staged in LDS (zeros on and above
// the diagonal, so every row below the phase's first is updated unconditionally).  Replaces, per group, four k_trsm_u
// launches and the three 32-row products between them (7 dependent launches, 0.55 ms on the 2200-column rest of C3's
// root front) by one launch.
// ---------------------------------------------------------------------------
constexpr int GROUP_ROWS = 128, GROUP_TILE = 32;

template <int JB>
__device__ __forceinline__ void group_phase(cz (&x)[GROUP_ROWS / 8], const cz (*sL)[GROUP_ROWS + 1], int rsub, int lane) {
#pragma unroll 1
    for (int s = 0; s < 32; ++s) {
        // row k = 32 JB + s lives in register m = 4 JB + (s >> 3) of the lane with rsub == (s & 7); bit masks instead of a
        // select chain keep x[] in registers (see k_trsm_u)
        long long xr = 0, xi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long long msk = -(long long)((s >> 3) == j);
            xr |= msk & __double_as_longlong(x[4 * JB + j].x);
            xi |= msk & __double_as_longlong(x[4 * JB + j].y);
        }
        int src = (lane & ~7) | (s & 7);
        cz xk = cz_make(__shfl(__longlong_as_double(xr), src), __shfl(__longlong_as_double(xi), src));
#pragma unroll
        for (int m = 4 * JB; m < GROUP_ROWS / 8; ++m) cz_msub(x[m], sL[s][rsub + 8 * m], xk);
    }
}

__global__ __launch_bounds__(256) void k_group_utrsm(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0, int gw,
                                                      cz* fact, int64_t fs, ColSel cs, unsigned long long* umax) {
    __shared__ cz sL[32][GROUP_ROWS + 1];
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int p = f.p, u = f.u, ld = p + u;
    int b = min(gw, p - k0);                 // rows of the group this front has
    int nin = p - k0 - b, ncol = nin + u;
    int v0 = t.a * GROUP_TILE, v1 = v0 + GROUP_TILE;
    {   // column tile without a selected column
        bool any = (v0 < nin && v0 < cs.chi && min(v1, nin) > cs.clo) || (cs.u12 && v1 > nin && v0 < ncol);
        if (!any) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, rsub = tid & 7;
    const int v = v0 + (tid >> 3);
    const bool mine = v < ncol && (v < nin ? (v >= cs.clo && v < cs.chi) : (cs.u12 != 0));
    const cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* col = (v < nin) ? fact + (int64_t)k * fs + f.lp + (int64_t)(k0 + b + v) * ld + k0
                        : fact + (int64_t)k * fs + f.f12 + (int64_t)(v - nin) * p + k0;
    cz x[GROUP_ROWS / 8];
#pragma unroll
    for (int m = 0; m < GROUP_ROWS / 8; ++m) x[m] = (mine && rsub + 8 * m < b) ? col[rsub + 8 * m] : cz_make(0, 0);
    auto stage = [&](int jb) {               // columns 32 jb .. 32 jb + 31 of L11, strictly below the diagonal
        __syncthreads();
        for (int idx = tid; idx < 32 * GROUP_ROWS; idx += 256) {
            int r = idx % GROUP_ROWS, s2 = idx / GROUP_ROWS, c = 32 * jb + s2;
            sL[s2][r] = (r > c && r < b) ? Lp[(k0 + r) + (int64_t)(k0 + c) * ld] : cz_make(0, 0);
        }
        __syncthreads();
    };
    stage(0); group_phase<0>(x, sL, rsub, lane);
    if (b > 32) { stage(1); group_phase<1>(x, sL, rsub, lane); }
    if (b > 64) { stage(2); group_phase<2>(x, sL, rsub, lane); }
    if (b > 96) { stage(3); group_phase<3>(x, sL, rsub, lane); }
    if (mine) {
#pragma unroll
        for (int m = 0; m < GROUP_ROWS / 8; ++m)
            if (rsub + 8 * m < b) col[rsub + 8 * m] = x[m];
    }
    if (umax) {
        double mx = 0.0;
#pragma unroll
        for (int m = 0; m < GROUP_ROWS / 8; ++m) if (mine && rsub + 8 * m < b) mx = fmax(mx, cz_abs1(x[m]));
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        if (lane == 0 && mx > 0.0) atomicMax(umax + k, (unsigned long long)__double_as_longlong(mx == mx ? mx : __longlong_as_double(0x7ff0000000000000ll)));
    }
}

void launch_group_utrsm(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, int gw, cz* fact, int64_t fs, int K, ColSel cs, unsigned long long* umax) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_group_utrsm, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, k0, gw, fact, fs, cs, umax);
}

// ---------------------------------------------------------------------------
// grouped complex GEMM  C -= A * B  (all column-major) on v_mfma_f64_16x16x4_f64.
// Workgroup tile 64 (m) x 64 (n); wave w owns rows [16w, 16w+16) and streams its
// A rows straight from global memory (each A element is used once per n tile);
// the K x 64 B tile is staged in LDS as [k][n] so every fragment read is a
// conflict-free ds_read_b128.  The MFMA computes the transposed tile D[n][m] so
// that C is written with 16 consecutive m per lane group (256 B segments).
// ---------------------------------------------------------------------------
constexpr int GEMM_KC = 32;
constexpr int GEMM_CHUNK = 64;     // work items per XCD chunk (one 8 x 8 super-tile)

__global__ __launch_bounds__(256, 2) void k_gemm_nn(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int k0,
                                                  cz* fact, int64_t fs, cz* cb, int64_t cbs, ColSel cs, int bw, int mrows, int ntiles, int K,
                                                  const int* __restrict__ child_idx, const int* __restrict__ cmap,
                                                  const cz* __restrict__ cbc, int64_t cbsc, int mirror) {
    // cs: the columns updated by this launch (pivot-block columns by offset from column k0 + bw; the U12 tiles when cs.u12).
    // bw: width of the eliminated block, FACT_NB or the width of a group of panels that is applied in one product.
    // mrows > 0: only the first mrows rows below the block (the U rows of the next panel of the group).
    __shared__ cz Bs[GEMM_KC][65];
    // Workgroups go to the XCDs round-robin.  XCD x takes the work items (node, tile) in chunks of GEMM_CHUNK consecutive items:
    // the host lists the tiles of a front in 8 x 8 super-tiles, so the tiles in flight on one XCD share their A row panels
    // and B column panels through that XCD's L2 instead of every tile fetching both from memory.
    if (mirror & 4) __builtin_amdgcn_s_setprio(2);      // a product on the dependency chain of the look-ahead schedule
    const int64_t seq = blockIdx.x >> 3;
    const int64_t item = (seq / GEMM_CHUNK) * (8 * GEMM_CHUNK) + (int64_t)(blockIdx.x & 7) * GEMM_CHUNK + seq % GEMM_CHUNK;
    if (item >= (int64_t)ntiles * K) return;
    const int k = (int)(item / ntiles);
    Tile t = tiles[item % ntiles];
    if (t.a == 0 && (t.c * 64 >= cs.chi || t.c * 64 + 64 <= cs.clo)) return;
    if (t.a == 1 && !cs.u12) return;
    // symmetric elimination (mirror bit 1): the trailing matrix stays symmetric and its upper part is never read (U rows are
    // written as D L^T by k_lscale_np), so the U12 updates and the tiles strictly above the diagonal of the pivot block are skipped
    if ((mirror & 2) && (t.a == 1 || (t.a == 0 && t.b < t.c))) return;
    FrontD f = fr[t.f];
    int p = f.p, u = f.u, ld = p + u;
    int b = max(1, min(bw, p - k0));
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    cz* F12 = fact + (int64_t)k * fs + f.f12;
    const cz *A, *B;
    cz* C;
    int M, N, Kd;
    int64_t lda, ldb, ldc;
    if (t.a == 0) {
        A = Lp + (k0 + b) + (int64_t)k0 * ld; lda = ld; M = ld - k0 - b; Kd = b;
        B = Lp + k0 + (int64_t)(k0 + b) * ld; ldb = ld; N = p - k0 - b;
        C = Lp + (k0 + b) + (int64_t)(k0 + b) * ld; ldc = ld;
    } else if (t.a == 1) {
        A = Lp + (k0 + b) + (int64_t)k0 * ld; lda = ld; M = p - k0 - b; Kd = b;
        B = F12 + k0; ldb = p; N = u;
        C = F12 + (k0 + b); ldc = p;
    } else if (t.a == 2) {
        A = Lp + p; lda = ld; M = u; Kd = p;
        B = F12; ldb = p; N = u;
        C = cb + (int64_t)k * cbs + f.cb; ldc = u;
    } else {
        // coupling of two solve-ready super blocks of bw columns (launch_couple_outer): pair index in t.a >> 8
        const int half = bw, kp = (t.a >> 8) * 2 * half, bb = min(half, p - kp - half);
        lda = ldb = ldc = ld;
        if ((t.a & 0xff) == 3) {      // rows below the pair:  P_a -= P_b * Pc  (mirror bit 8, lazy conversion: the update rows are raw L21 and stay out of it)
            A = Lp + (kp + half + bb) + (int64_t)(kp + half) * ld; M = ((mirror & 8) ? p : ld) - (kp + half + bb); Kd = bb;
            B = Lp + (kp + half) + (int64_t)kp * ld; N = half;
            C = Lp + (kp + half + bb) + (int64_t)kp * ld;
        } else {                      // rows above the pair:  Q_b -= Q_a * Qd
            A = Lp + (int64_t)kp * ld; M = kp; Kd = half;
            B = Lp + kp + (int64_t)(kp + half) * ld; N = bb;
            C = Lp + (int64_t)(kp + half) * ld;
        }
    }
    const int kind = t.a & 0xff;
    if (mrows > 0 && kind < 2) M = min(M, min(mrows, p - k0 - b));     // pivot-block rows only
    int nlo = 0;
    if (t.a == 0) { N = min(N, cs.chi); nlo = cs.clo; }
    int m0 = t.b * 64, n0 = t.c * 64;
    if (m0 >= M || n0 >= N) return;
    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    int mrow = m0 + wave * 16 + li;
    bool mok = mrow < M;
    int mrow_c = min(mrow, M - 1);
    // 3M complex product (T1 = br ar, T2 = bi ai, T3 = (br+bi)(ar+ai); re = T1 - T2, im = T3 - T1 - T2): three MFMAs per complex
    // multiply-add instead of four; the third accumulator fits at two waves per SIMD (138 VGPRs + 96 AGPRs)
    d4 t1[4], t2[4], t3[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { t1[nt] = (d4){0, 0, 0, 0}; t2[nt] = (d4){0, 0, 0, 0}; t3[nt] = (d4){0, 0, 0, 0}; }
    // The read-modify-write loads of the epilogue are issued in front of the MFMAs of the last K chunk (the B staging registers
    // are dead by then): at the depths of the panel-group updates (32..128) their latency is otherwise a tenth of the tile.
    // The contribution block (kind 2) is written, not updated: CB = -L21 U12, the children's blocks are added afterwards.
    cz old[4][4];
    const bool rmw = (kind != 2);
    // loads are unconditional (clamped addresses) and masked with selects, so they all stay in flight
    auto chunk = [&](int kc, auto is_last) {
        cz bvals[GEMM_KC * 64 / 256];
#pragma unroll
        for (int q = 0; q < GEMM_KC * 64 / 256; ++q) {
            int idx = tid + 256 * q;
            int kk = idx % GEMM_KC, nn = idx / GEMM_KC;
            bvals[q] = B[min(kc + kk, Kd - 1) + (int64_t)min(n0 + nn, N - 1) * ldb];
        }
        cz av[GEMM_KC / 4];
#pragma unroll
        for (int ks = 0; ks < GEMM_KC / 4; ++ks) av[ks] = A[mrow_c + (int64_t)min(kc + ks * 4 + lk, Kd - 1) * lda];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GEMM_KC * 64 / 256; ++q) {
            int idx = tid + 256 * q;
            int kk = idx % GEMM_KC, nn = idx / GEMM_KC;
            bool keep = (kc + kk < Kd) && (n0 + nn < N) && (n0 + nn >= nlo);
            Bs[kk][nn] = keep ? bvals[q] : cz_make(0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < GEMM_KC / 4; ++ks) {
            bool keep = mok && (kc + ks * 4 + lk < Kd);
            av[ks].x = keep ? av[ks].x : 0.0;
            av[ks].y = keep ? av[ks].y : 0.0;
        }
        __syncthreads();
        if constexpr (decltype(is_last)::value) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[nt][r] = rmw ? C[mrow_c + (int64_t)min(n0 + nt * 16 + lk + 4 * r, N - 1) * ldc] : cz_make(0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < GEMM_KC / 4; ++ks) {
            int kk = ks * 4 + lk;
            cz bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = Bs[kk][nt * 16 + li];
            double ar = av[ks].x, ai = av[ks].y, as = av[ks].x + av[ks].y;
            // D[n][m] += B^T[n][k] A^T[k][m]   (MFMA "A" operand = B tile from LDS, "B" operand = A slab)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t1[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[nt].x, ar, t1[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[nt].y, ai, t2[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t3[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[nt].x + bv[nt].y, as, t3[nt], 0, 0, 0);
        }
    };
    int kc0 = 0;
    for (; kc0 + GEMM_KC < Kd; kc0 += GEMM_KC) chunk(kc0, std::false_type{});
    chunk(kc0, std::true_type{});
    // Contribution block with children (cbc given): the children's contribution entries that land in this tile are pulled in
    // here, in child order, instead of in a second pass (k_extend_add phase 1) that would read and rewrite the whole block:
    // cmap[row][child] is the child's contribution row behind a parent contribution row (as in the forward sweep).
    if (t.a == 2 && cbc != nullptr && f.cmap >= 0 && mok) {
        const int nch = f.nchild;
        for (int c = 0; c < nch; ++c) {
            const FrontD* g = fr + child_idx[f.child_ptr + c];
            const int64_t gcrow = g->crow;
            const int rm = cmap[f.cmap + (int64_t)mrow * nch + c];
            if (rm < 0) continue;
            const int gu = g->u;
            const cz* src = cbc + (int64_t)k * cbsc + g->cb + (rm - gcrow);
            int cn[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) cn[nt][r] = cmap[f.cmap + (int64_t)min(n0 + nt * 16 + lk + 4 * r, N - 1) * nch + c];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (cn[nt][r] >= 0) {
                        // (symmetric contribution blocks hold their lower triangle only: entry (row, col) with row < col is read as (col, row))
                        const int64_t ri = rm - gcrow, ci = cn[nt][r] - gcrow;
                        const cz* cbase = cbc + (int64_t)k * cbsc + g->cb;
                        cz v = ((mirror & 1) && ri < ci) ? cbase[ci + ri * gu] : src[ci * gu];
                        old[nt][r].x += v.x; old[nt][r].y += v.y;
                    }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int gn = n0 + nt * 16 + lk + 4 * r;
            if (mok && gn < N && gn >= nlo) {
                const cz v = cz_make(old[nt][r].x - (t1[nt][r] - t2[nt][r]), old[nt][r].y - (t3[nt][r] - t1[nt][r] - t2[nt][r]));
                C[mrow + (int64_t)gn * ldc] = v;
                // symmetric pencil: the contribution block -L21 U12 (+ children) is symmetric; only the tiles on and below the
                // diagonal are computed and (round 5) nothing is mirrored: the parent reads entry (i, j), i < j, as (j, i)
            }
        }
}

void launch_gemm_nn(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                    cz* cb, int64_t cbs, int K, ColSel cs, int bw, int mrows, const cz* cbc, int64_t cbsc, int mirror) {
    if (ntiles <= 0) return;
    const int64_t items = (int64_t)ntiles * K, per = 8 * GEMM_CHUNK;
    hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((items + per - 1) / per * per)), dim3(256), 0, st, S.fr, tiles, k0, fact, fs, cb, cbs, cs, bw, mrows,
                       ntiles, K, S.child_idx, S.cmap, cbc, cbsc, mirror);
}

// ---------------------------------------------------------------------------
// Solve-ready form.  After a front is factorised, every SOLVE_NB diagonal block of
// its pivot block is replaced by the explicit inverses inv(L_tt) (unit lower, stored
// strictly below the diagonal) and inv(U_tt) (upper incl. diagonal), and every
// off-diagonal block of block column t is scaled:  below the diagonal block
// P = L[:,t] inv(L_tt) (this includes L21), above it PU = U[:,t] inv(U_tt).
// The block substitutions then contain no triangular solves at all:
//   forward  w_r -= P_rt w_t,  y_t = inv(L_tt) w_t      backward  t_r -= PU_rt t_t,  x_t = inv(U_tt) t_t
// (inverse-based blocked TRSM as in MAGMA), i.e. only GEMM-shaped work for the matrix cores.
// ---------------------------------------------------------------------------
// Pivot-light path: k_diag_grp leaves the inverses of the 32 x 32 diagonal blocks in place.  Two of them and the two blocks
// between them (C: L entries below the first block, Dd: U entries right of it) are coupled into the inverses of the 64 x 64
// block -- exactly what k_block_inverse computes from L\U by substitution, here four products of 32 x 32 blocks:
//   inv([L1 0; C L2]) = [I1 0; -I2 C I1, I2],   inv([U1 Dd; 0 U2]) = [J1, -J1 Dd J2; 0, J2].     Tile.a = 64-column block.
__global__ __launch_bounds__(256) void k_couple32(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, cz* fact, int64_t fs) {
    __shared__ cz M0[32][33], M1[32][33], T0[32][33], T1[32][33];
    Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    FrontD f = fr[t.f];
    const int ld = f.p + f.u, k0 = t.a * INV_NB;
    const int b2 = min(32, f.p - k0 - 32);          // size of the second 32-block (the tile exists only if b2 > 0)
    if (b2 <= 0) return;
    cz* D0 = fact + (int64_t)k * fs + f.lp + k0 + (int64_t)k0 * ld;
    auto at = [&](int i, int j) -> cz& { return D0[i + (int64_t)j * ld]; };
    const int tid = threadIdx.x;
    // ---- lower: C (rows 32.., cols 0..31) <- -I2 (C I1);  I1 = unit lower inverse of block 1 (strictly lower stored), I2 of block 2
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        int i = idx % 32, j = idx / 32;
        M0[i][j] = i < b2 ? at(32 + i, j) : cz_make(0, 0);                                   // C
        T0[i][j] = i > j ? at(i, j) : cz_make(i == j ? 1.0 : 0.0, 0.0);                        // I1
        T1[i][j] = (i > j && i < b2) ? at(32 + i, 32 + j) : cz_make(i == j ? 1.0 : 0.0, 0.0);  // I2
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        int i = idx % 32, j = idx / 32;
        cz sacc = cz_make(0, 0);
        for (int q = j; q < 32; ++q) cz_madd(sacc, M0[i][q], T0[q][j]);        // (C I1)[i][j], I1 lower
        M1[i][j] = sacc;
    }
    __syncthreads();
    cz lo[4];
    for (int r = 0; r < 4; ++r) {
        int idx = tid + 256 * r, i = idx % 32, j = idx / 32;
        cz sacc = cz_make(0, 0);
        for (int q = 0; q <= i; ++q) cz_madd(sacc, T1[i][q], M1[q][j]);        // (I2 (C I1))[i][j], I2 lower
        lo[r] = cz_make(-sacc.x, -sacc.y);
    }
    __syncthreads();
    // ---- upper: Dd (rows 0..31, cols 32..) <- -J1 (Dd J2);  J1, J2 = upper inverses incl. diagonal
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        int i = idx % 32, j = idx / 32;
        M0[i][j] = j < b2 ? at(i, 32 + j) : cz_make(0, 0);                                    // Dd
        T0[i][j] = i <= j ? at(i, j) : cz_make(0, 0);                                            // J1
        T1[i][j] = (i <= j && j < b2) ? at(32 + i, 32 + j) : cz_make(0, 0);                      // J2
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        int i = idx % 32, j = idx / 32;
        cz sacc = cz_make(0, 0);
        for (int q = 0; q <= j; ++q) cz_madd(sacc, M0[i][q], T1[q][j]);        // (Dd J2)[i][j], J2 upper
        M1[i][j] = sacc;
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
        int idx = tid + 256 * r, i = idx % 32, j = idx / 32;
        cz sacc = cz_make(0, 0);
        for (int q = i; q < 32; ++q) cz_madd(sacc, T0[i][q], M1[q][j]);        // (J1 (Dd J2))[i][j], J1 upper
        if (j < b2) at(i, 32 + j) = cz_make(-sacc.x, -sacc.y);
        if (i < b2) at(32 + i, j) = lo[r];
    }
}

void launch_couple32(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_couple32, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, fact, fs);
}

__global__ __launch_bounds__(128) void k_block_inverse(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                       cz* fact, int64_t fs) {
    __shared__ cz T[INV_NB][INV_NB + 1];
    __shared__ cz O[INV_NB][INV_NB + 1];
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int ld = f.p + f.u, k0 = t.a * INV_NB;
    int b = min(INV_NB, f.p - k0);
    cz* D = fact + (int64_t)k * fs + f.lp + k0 + (int64_t)k0 * ld;
    for (int idx = threadIdx.x; idx < b * b; idx += 128) { int i = idx % b, j = idx / b; T[i][j] = D[i + (int64_t)j * ld]; }
    __syncthreads();
    int j = threadIdx.x & 63;
    if (j < b) {
        if (threadIdx.x < 64) {          // column j of inv(L): solve L x = e_j, x_i = 0 for i < j, x_j = 1
            for (int i = j + 1; i < b; ++i) {
                cz s = T[i][j];                                  // l_ij * x_j
                for (int q = j + 1; q < i; ++q) cz_madd(s, T[i][q], O[q][j]);
                O[i][j] = cz_make(-s.x, -s.y);
            }
        } else {                          // column j of inv(U): solve U x = e_j, x_i = 0 for i > j
            cz dj = cz_inv(T[j][j]);
            O[j][j] = dj;
            for (int i = j - 1; i >= 0; --i) {
                cz s = cz_make(0, 0);
                for (int q = i + 1; q <= j; ++q) cz_madd(s, T[i][q], O[q][j]);
                cz di = cz_inv(T[i][i]);
                cz v = cz_mul(s, di);
                O[i][j] = cz_make(-v.x, -v.y);
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < b * b; idx += 128) { int i = idx % b, jj = idx / b; D[i + (int64_t)jj * ld] = O[i][jj]; }
}

void launch_block_inverse(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_block_inverse, dim3(ntiles, K), dim3(128), 0, st, S.fr, tiles, fact, fs);
}

// Couple two inverted INV_NB blocks of a SOLVE_NB super block:  inv([L1 0; C L2]) = [I1 0; -I2 C I1  I2] and
// inv([U1 D; 0 U2]) = [J1  -J1 D J2; 0 J2], in place.  One workgroup per (front, super block, node).
__global__ __launch_bounds__(256) void k_couple_inverse(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                        cz* fact, int64_t fs) {
    // one staging buffer (66 KB; two of them kept every other LDS user off the CU): a product is formed in registers (16 entries
    // per thread) and replaces its input in LDS after a barrier
    __shared__ cz M0[INV_NB][INV_NB + 1];
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int ld = f.p + f.u, k0 = t.a * SOLVE_NB;
    int b2 = min(INV_NB, f.p - k0 - INV_NB);        // size of the second block (the tile exists only if b2 > 0)
    cz* D0 = fact + (int64_t)k * fs + f.lp + k0 + (int64_t)k0 * ld;      // super block origin
    auto at = [&](int i, int j) -> cz& { return D0[i + (int64_t)j * ld]; };
    int tid = threadIdx.x;
    constexpr int PER = INV_NB * INV_NB / 256;
    cz acc[PER];
    // ---- lower: C (rows 64.., cols 0..63) <- -I2 (C I1)
    for (int idx = tid; idx < b2 * INV_NB; idx += 256) { int i = idx % b2, j = idx / b2; M0[i][j] = at(INV_NB + i, j); }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        int idx = tid + 256 * r;
        acc[r] = cz_make(0, 0);
        if (idx < b2 * INV_NB) {
            int i = idx % b2, j = idx / b2;
            cz sacc = M0[i][j];                                // I1[j][j] = 1
            for (int q = j + 1; q < INV_NB; ++q) cz_madd(sacc, M0[i][q], at(q, j));
            acc[r] = sacc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) { int idx = tid + 256 * r; if (idx < b2 * INV_NB) M0[idx % b2][idx / b2] = acc[r]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        int idx = tid + 256 * r;
        if (idx < b2 * INV_NB) {
            int i = idx % b2, j = idx / b2;
            cz sacc = M0[i][j];                                // I2[i][i] = 1
            for (int q = 0; q < i; ++q) cz_madd(sacc, at(INV_NB + i, INV_NB + q), M0[q][j]);
            acc[r] = cz_make(-sacc.x, -sacc.y);
        }
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) { int idx = tid + 256 * r; if (idx < b2 * INV_NB) at(INV_NB + idx % b2, idx / b2) = acc[r]; }
    __syncthreads();
    // ---- upper: D (rows 0..63, cols 64..) <- -J1 (D J2)
    for (int idx = tid; idx < INV_NB * b2; idx += 256) { int i = idx % INV_NB, j = idx / INV_NB; M0[i][j] = at(i, INV_NB + j); }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        int idx = tid + 256 * r;
        acc[r] = cz_make(0, 0);
        if (idx < INV_NB * b2) {
            int i = idx % INV_NB, j = idx / INV_NB;
            cz sacc = cz_make(0, 0);
            for (int q = 0; q <= j; ++q) cz_madd(sacc, M0[i][q], at(INV_NB + q, INV_NB + j));
            acc[r] = sacc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) { int idx = tid + 256 * r; if (idx < INV_NB * b2) M0[idx % INV_NB][idx / INV_NB] = acc[r]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        int idx = tid + 256 * r;
        if (idx < INV_NB * b2) {
            int i = idx % INV_NB, j = idx / INV_NB;
            cz sacc = cz_make(0, 0);
            for (int q = i; q < INV_NB; ++q) cz_madd(sacc, at(i, q), M0[q][j]);
            at(i, INV_NB + j) = cz_make(-sacc.x, -sacc.y);
        }
    }
}

void launch_couple_inverse(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_couple_inverse, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, fact, fs);
}

// Rows of super-block column T (tile.a) outside the diagonal super block: 64-row tile (tile.b) of the rows
// above (tile.c = 0: row <- row * inv(U_TT)) or below (tile.c = 1: row <- row * inv(L_TT)).  In place: the
// two 64-column halves are produced in an order that never overwrites an input that is still needed, and every
// wave reads its 16 rows completely before it writes them.  f64 MFMA, transposed-tile layout of k_gemm_nn.
constexpr int SCALE_KC = 32;        // k rows of the inverse block staged per step (33 KB of LDS, like k_gemm_nn)

// The earlier version staged a whole 128 x 64 half of the inverse block (133 KB of LDS: one workgroup per CU, and no other LDS user
// beside it while it ran in the background of the levels above) and kept the wave's 16 x 128 row slab in registers (450 registers:
// one wave per SIMD).  Here the K loop runs in chunks of 32 like k_gemm_nn: the row entries of a chunk are loaded when the chunk is
// multiplied; in place is still safe because a wave writes its rows only after the last chunk of a pass, and the second pass reads
// the half of the row the first pass did not write.
__global__ __launch_bounds__(256, 2) void k_scale_panels(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                         cz* fact, int64_t fs, int pivot_rows_only) {
    __shared__ cz Bs[SCALE_KC][65];      // Bs[k][n] = Inv[kfirst + kc + k][ncol0 + n]
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int ld = f.p + f.u, k0 = t.a * SOLVE_NB;
    int b = min(SOLVE_NB, f.p - k0);
    bool below = t.c != 0;
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    // pivot_rows_only (Engine::lazy_conversion_: factors that serve ONE sweep, the reference's store = false): the update rows L21 stay
    // raw -- three quarters of this kernel's traffic -- and the forward sweep multiplies them with the solved block Y_J in a launch of
    // its own instead of P21 = L21 inv(L11) with W_J
    const int rend = below ? (pivot_rows_only ? f.p : ld) : k0;
    int nhalf = (b > INV_NB) ? 2 : 1;
    for (int pass = 0; pass < nhalf; ++pass) {
        // lower: first the left half (needs both halves of the row), then the right half (needs only itself);
        // upper: first the right half (needs both), then the left half.
        int h = below ? pass : (nhalf - 1 - pass);
        int ncol0 = h * INV_NB, ncols = min(INV_NB, b - ncol0);
        int kfirst = below ? ncol0 : 0;
        int klen = below ? b - ncol0 : ncol0 + ncols;
        for (int sub = 0; sub < SCALE_ROWTILES; ++sub) {
            int rbase = (below ? k0 + b : 0) + (t.b * SCALE_ROWTILES + sub) * 64;
            if (rbase >= rend) break;
            int mrow = rbase + wave * 16 + li;
            bool mok = mrow < rend;
            int mrow_c = min(mrow, rend - 1);
            d4 t1[4], t2[4], t3[4];       // 3M complex product, see k_gemm_nn
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { t1[nt] = (d4){0, 0, 0, 0}; t2[nt] = (d4){0, 0, 0, 0}; t3[nt] = (d4){0, 0, 0, 0}; }
            for (int kc = 0; kc < klen; kc += SCALE_KC) {
                cz av[SCALE_KC / 4];
#pragma unroll
                for (int ks = 0; ks < SCALE_KC / 4; ++ks) {
                    int kk = kc + ks * 4 + lk;
                    cz v = Lp[mrow_c + (int64_t)(k0 + kfirst + min(kk, klen - 1)) * ld];
                    bool keep = mok && kk < klen;
                    av[ks].x = keep ? v.x : 0.0;
                    av[ks].y = keep ? v.y : 0.0;
                }
                cz bvals[SCALE_KC * 64 / 256];
#pragma unroll
                for (int q = 0; q < SCALE_KC * 64 / 256; ++q) {
                    int idx = tid + 256 * q, kk = idx % SCALE_KC, nn = idx / SCALE_KC;
                    int gi = kfirst + kc + kk, gj = ncol0 + nn;              // element (gi, gj) of the super-block inverse
                    cz v = Lp[(k0 + min(gi, b - 1)) + (int64_t)(k0 + min(gj, b - 1)) * ld];
                    const bool in = kc + kk < klen && nn < ncols;
                    const bool nz = below ? gi > gj : gi <= gj;
                    if (!(in && nz)) v = cz_make((in && below && gi == gj) ? 1.0 : 0.0, 0.0);
                    bvals[q] = v;
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < SCALE_KC * 64 / 256; ++q) { int idx = tid + 256 * q; Bs[idx % SCALE_KC][idx / SCALE_KC] = bvals[q]; }
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < SCALE_KC / 4; ++ks) {
                    const int kk = ks * 4 + lk;
                    const double ar = av[ks].x, ai = av[ks].y, as = ar + ai;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const cz bv = Bs[kk][nt * 16 + li];
                        // D[n][m] = sum_k Inv[k][n] A[m][k]
                        t1[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.x, ar, t1[nt], 0, 0, 0);
                        t2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.y, ai, t2[nt], 0, 0, 0);
                        t3[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.x + bv.y, as, t3[nt], 0, 0, 0);
                    }
                }
            }
            if (mok) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int gn = nt * 16 + lk + 4 * r;
                        if (gn < ncols) Lp[mrow + (int64_t)(k0 + ncol0 + gn) * ld] = cz_make(t1[nt][r] - t2[nt][r], t3[nt][r] - t1[nt][r] - t2[nt][r]);
                    }
            }
        }
    }
}

void launch_scale_panels(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K, bool pivot_rows_only) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_scale_panels, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, fact, fs, pivot_rows_only ? 1 : 0);
}

void launch_couple_outer(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int half, cz* fact, int64_t fs, int K, bool pivot_rows_only) {
    launch_gemm_nn(st, S, tiles, ntiles, 0, fact, fs, nullptr, 0, K, ColSel(), half, 0, nullptr, 0, pivot_rows_only ? 8 : 0);
}

// Second half of the coupling of two solve-ready super blocks (kernels.hpp: launch_couple_level): one workgroup owns 16 whole
// columns of the coupling block, so the product can be formed in place: the strip goes to LDS, thread i then owns row i of the
// result, out[i,:] = -sum_k T[i,k] S[k,:], with T the unit lower inverse inv(L_bb) (side 0, rows/columns of block b) or the upper
// inverse inv(U_aa) (side 1, block a), both read from the diagonal super blocks they live in.  m x m x 16 multiply-adds on the
// vector units: a few hundred workgroups of < 20 us per tree level.
__global__ __launch_bounds__(256) void k_couple_level(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int half,
                                                      cz* fact, int64_t fs) {
    extern __shared__ cz couple_lds[];      // half x 17 entries (launch_couple_level)
    cz (*Ss)[17] = reinterpret_cast<cz (*)[17]>(couple_lds);
    Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    FrontD f = fr[t.f];
    const int ld = f.p + f.u, kp = t.a * 2 * half, bb = min(half, f.p - kp - half);
    const bool upper = t.b != 0;
    cz* Lp = fact + (int64_t)k * fs + f.lp;
    // lower: strip of Pc (rows of b, columns of a), triangle = diagonal block b;  upper: strip of Qd (rows of a, columns of b), triangle = block a
    const int m = upper ? half : bb, ncols = upper ? bb : half, c0 = t.c * 16;
    if (c0 >= ncols) return;
    const int nc = min(16, ncols - c0);
    cz* Cb = upper ? Lp + kp + (int64_t)(kp + half + c0) * ld : Lp + (kp + half) + (int64_t)(kp + c0) * ld;
    const cz* Tb = upper ? Lp + kp + (int64_t)kp * ld : Lp + (kp + half) + (int64_t)(kp + half) * ld;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < m * 16; idx += 256) { int i = idx % m, j = idx / m; Ss[i][j] = j < nc ? Cb[i + (int64_t)j * ld] : cz_make(0, 0); }
    __syncthreads();
    // out = -T S on the matrix cores (the vector-unit version read 16 LDS values per multiply-add row and thread: 1.4 ms for the
    // 512-column blocks of the root front, at the very end of the factorisation).  Transposed tile as in k_gemm_nn:
    // D[j][i] = sum_k S[k][j] T[i][k], so the result is stored with 16 consecutive rows per lane group.
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    for (int rt = wave; rt * 16 < m; rt += 4) {
        const int i = rt * 16 + li;
        const bool iok = i < m;
        const int ic = min(i, m - 1);
        const int kbeg = upper ? rt * 16 : 0, kend = upper ? m : min(m, rt * 16 + 16);      // upper incl. diagonal / strictly lower + unit diagonal
        d4 t1 = (d4){0, 0, 0, 0}, t2 = (d4){0, 0, 0, 0}, t3 = (d4){0, 0, 0, 0};
        for (int kk0 = kbeg; kk0 < kend; kk0 += 4) {
            const int kk = kk0 + lk, kkc = min(kk, m - 1);
            const cz tv = Tb[ic + (int64_t)kkc * ld];
            const bool keep = iok && kk < kend && (upper ? kk >= i : kk < i);
            const double br = keep ? tv.x : ((!upper && iok && kk == i) ? 1.0 : 0.0), bi = keep ? tv.y : 0.0;
            cz sv = Ss[kkc][li];
            if (kk >= m) { sv.x = 0.0; sv.y = 0.0; }
            t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sv.x, br, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(sv.y, bi, t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sv.x + sv.y, br + bi, t3, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lk + 4 * r;
            if (iok && j < nc) Cb[i + (int64_t)j * ld] = cz_make(-(t1[r] - t2[r]), -(t3[r] - t1[r] - t2[r]));
        }
    }
}

void launch_couple_level(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int half, cz* fact, int64_t fs, int K) {
    if (ntiles <= 0) return;
    static std::atomic<uint64_t> attr_seen{0};
    if (first_on_this_device(attr_seen)) { (void)hipFuncSetAttribute((const void*)k_couple_level, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 17 * (int)sizeof(cz)); }
    hipLaunchKernelGGL(k_couple_level, dim3(ntiles, K), dim3(256), (size_t)half * 17 * sizeof(cz), st, S.fr, tiles, half, fact, fs);
}

// ---------------------------------------------------------------------------
// pivot sequence -> row permutation of the pivot block (and its inverse), so the
// solves gather the right-hand side once instead of replaying p swaps.
// rowperm[c0+i] = original local row that ended in position i.
// ---------------------------------------------------------------------------
constexpr int ROWPERM_LDS = 12288;

__global__ void k_rowperm(const FrontD* __restrict__ fr, const int* __restrict__ list, const int* __restrict__ piv,
                          int* rowperm, int* rowinv, int n) {
    __shared__ int s_rp[ROWPERM_LDS];
    FrontD f = fr[list[blockIdx.x]];
    int k = blockIdx.y;
    int p = f.p;
    const int* pv = piv + (int64_t)k * n + f.c0;
    int* rp = rowperm + (int64_t)k * n + f.c0;
    int* ri = rowinv + (int64_t)k * n + f.c0;
    if (p <= ROWPERM_LDS / 2) {
        int* s_pv = s_rp + p;
        for (int i = threadIdx.x; i < p; i += blockDim.x) { s_rp[i] = i; s_pv[i] = pv[i]; }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int j = 0; j < p; ++j) { int q = s_pv[j]; if (q != j) { int a = s_rp[j]; s_rp[j] = s_rp[q]; s_rp[q] = a; } }
        __syncthreads();
        for (int i = threadIdx.x; i < p; i += blockDim.x) { int o = s_rp[i]; rp[i] = o; ri[o] = i; }
    } else {
        for (int i = threadIdx.x; i < p; i += blockDim.x) rp[i] = i;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int j = 0; j < p; ++j) { int q = pv[j]; if (q != j) { int a = rp[j]; rp[j] = rp[q]; rp[q] = a; } }
        __syncthreads();
        for (int i = threadIdx.x; i < p; i += blockDim.x) ri[rp[i]] = i;
    }
}

void launch_rowperm(hipStream_t st, const DevSym& S, const int* list, int nlist, const int* piv, int* rowperm,
                    int* rowinv, int K) {
    if (nlist <= 0) return;
    hipLaunchKernelGGL(k_rowperm, dim3(nlist, K), dim3(256), 0, st, S.fr, list, piv, rowperm, rowinv, S.n);
}

}  // namespace feasthip

// not part of include/feasthip.h: development hook (see prof_stamp above; -1 when the library was built without FEASTHIP_PROF).
// on: 0 off, 1 on (every launch), 2 + k0: on for the launches whose first column is k0; out (64 values) receives the stamps.
extern "C" int feasthip_debug_prof(int on, unsigned long long* out) {
#ifdef FEASTHIP_PROF
    const int flag = on > 0 ? 1 : 0, k0sel = on >= 2 ? on - 2 : -1;
    if (on >= 0 && (hipMemcpyToSymbol(HIP_SYMBOL(feasthip::g_prof_on), &flag, sizeof(int)) != hipSuccess ||
                    hipMemcpyToSymbol(HIP_SYMBOL(feasthip::g_prof_k0), &k0sel, sizeof(int)) != hipSuccess)) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(feasthip::g_prof), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return 0;
#else
    (void)on; (void)out;
    return -1;
#endif
}
