// Rayleigh-Ritz block of the FEAST iteration on the GPU (src/feast.jl:41-50,
// 117-127; src/utils.jl:104-116, 151-171): products A*Q / B*Q, the tall-skinny
// projection Q^H (A Q) on the f64 matrix cores, Ritz back-transform, residual
// vectors and column norms.  Block vectors are row-major n x ldw complex128.
#include "kernels.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace feasthip {

__device__ inline void cmfma3(d4& cr, d4& ci, double ar, double ai, double br, double bi) {
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, ci, 0, 0, 0);
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, ci, 0, 0, 0);
}

__global__ void k_permute_vals(int64_t nnz, const int* __restrict__ q, const cz* __restrict__ in, cz* __restrict__ out) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < nnz) out[q[e]] = in[e];
}
void launch_permute_vals(hipStream_t st, int64_t nnz, const int* q, const cz* in, cz* out) {
    hipLaunchKernelGGL(k_permute_vals, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, q, in, out);
}

// CSR SpMM: one wave per matrix row, one lane per block-vector column; the CSR
// (rowptr, col, val) arrays are read coalesced by the wave, the gathered V rows are
// whole contiguous lines.
__global__ void k_spmm_csr(int n, const int64_t* __restrict__ rowptr, const int* __restrict__ col,
                           const cz* __restrict__ val, const cz* __restrict__ V, int ldw, cz* out, cz alpha, int accumulate) {
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (i >= n) return;
    int64_t a = rowptr[i], b = rowptr[i + 1];
    for (int j = lane; j < ldw; j += 64) {
        cz acc = cz_make(0, 0);
        int64_t t = a;
        for (; t + 4 <= b; t += 4) {          // four gathered rows of V in flight (the plain loop is one dependent chain per entry)
            int c0 = col[t], c1 = col[t + 1], c2 = col[t + 2], c3 = col[t + 3];
            cz a0 = val[t], a1 = val[t + 1], a2 = val[t + 2], a3 = val[t + 3];
            cz v0 = V[(int64_t)c0 * ldw + j], v1 = V[(int64_t)c1 * ldw + j], v2 = V[(int64_t)c2 * ldw + j], v3 = V[(int64_t)c3 * ldw + j];
            cz_madd(acc, a0, v0); cz_madd(acc, a1, v1); cz_madd(acc, a2, v2); cz_madd(acc, a3, v3);
        }
        for (; t < b; ++t) cz_madd(acc, val[t], V[(int64_t)col[t] * ldw + j]);
        cz r = cz_mul(alpha, acc);
        if (accumulate) r = cz_add(r, out[(int64_t)i * ldw + j]);
        out[(int64_t)i * ldw + j] = r;
    }
}

void launch_spmm_csr(hipStream_t st, int n, const int64_t* rowptr, const int* col, const cz* val, const cz* V, int ldw,
                     cz* out, cz alpha, int accumulate) {
    hipLaunchKernelGGL(k_spmm_csr, dim3((n + 3) / 4), dim3(256), 0, st, n, rowptr, col, val, V, ldw, out, alpha, accumulate);
}

// Dense product out = alpha * A * V (+ out): same streaming structure as the solve's panel kernel.
// kparts > 1: blockIdx.z takes the z-th slice of the K range and writes its raw partial sums to part + z n ldw (summed, scaled
// and accumulated by k_dense_mv_reduce in a fixed order): a 4096 x 4096 term has only 64 row slabs, a quarter of the CUs.
__global__ __launch_bounds__(256) void k_dense_mv(int n, const cz* __restrict__ A, const cz* __restrict__ V, int ldw,
                                                   cz* out, cz alpha, int accumulate, int kparts, cz* part) {
    __shared__ cz Vs[64][LDS_COLS];
    int rbase = blockIdx.x * 64, cg = blockIdx.y * LDS_COLS;
    const int kchunks = (n + 63) / 64, kper = (kchunks + kparts - 1) / kparts;
    const int kbeg = blockIdx.z * kper * 64, kend = min(n, kbeg + kper * 64);
    if (kparts > 1) { out = part + (int64_t)blockIdx.z * n * ldw; alpha = cz_make(1, 0); accumulate = 0; }
    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    int arow = rbase + wave * 16 + li;
    bool aok = arow < n;
    d4 cr[4], ci[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { cr[nt] = (d4){0, 0, 0, 0}; ci[nt] = (d4){0, 0, 0, 0}; }
    for (int kc = kbeg; kc < kend; kc += 64) {
        __syncthreads();
        for (int r = wave; r < 64; r += 4) {
            int gk = kc + r;
            Vs[r][lane] = (gk < kend && cg + lane < ldw) ? V[(int64_t)gk * ldw + cg + lane] : cz_make(0, 0);
        }
        cz av[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            int gk = kc + ks * 4 + lk;
            av[ks] = (aok && gk < kend) ? A[arow + (int64_t)gk * n] : cz_make(0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                cz bv = Vs[ks * 4 + lk][nt * 16 + li];
                cmfma3(cr[nt], ci[nt], av[ks].x, av[ks].y, bv.x, bv.y);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int orow = rbase + wave * 16 + lk + 4 * r;
        if (orow >= n) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = cg + nt * 16 + li;
            if (col < ldw) {
                cz v = cz_mul(alpha, cz_make(cr[nt][r], ci[nt][r]));
                cz* d = out + (int64_t)orow * ldw + col;
                if (accumulate) v = cz_add(v, *d);
                *d = v;
            }
        }
    }
}

// The same product without LDS and without barriers, as the sweep kernel k_apply2 does it (kernels_solve.hip): every wave owns
// 16 rows x 64 columns and streams both operands into the matrix-core registers through rings (the matrix 8 k-steps ahead, the
// block-vector rows 4), 3M complex products.  (k_dense_mv above, two barriers per 64 columns of A: 244 us per 4096 x 4096 x 64
// product, 35 TFLOP/s; it stays for leading dimensions this kernel does not take.)
constexpr int DMV_DA = 8, DMV_DV = 4;
__global__ __launch_bounds__(256, 2) void k_dense_mv2(int n, const cz* __restrict__ A, const cz* __restrict__ V, int ldw,
                                                      cz* out, cz alpha, int accumulate, int kparts, cz* part) {
    const int rbase = blockIdx.x * 64, cg = blockIdx.y * 64;
    const int kchunks = (n + 63) / 64, kper = (kchunks + kparts - 1) / kparts;
    const int kbeg = blockIdx.z * kper * 64, kend = min(n, kbeg + kper * 64);
    if (kparts > 1) { out = part + (int64_t)blockIdx.z * n * ldw; alpha = cz_make(1, 0); accumulate = 0; }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    if (rbase + wave * 16 >= n) return;                      // (no barriers: a wave without rows may leave)
    const int arow = rbase + wave * 16 + li;
    const bool aok = arow < n;
    const cz* Ap = A + min(arow, n - 1);
    const int Kd = kend - kbeg, nks = (Kd + 3) >> 2;
    int vcol[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) vcol[q] = min(cg + q * 16 + li, ldw - 1);
    cz ar[DMV_DA], vr[DMV_DV][4];
    auto kof = [&](int s2) { return kbeg + min(4 * s2 + lk, Kd - 1); };      // clamped: steps beyond the end re-read the last column (masked)
    auto loadA = [&](int s2, int slot) { ar[slot] = Ap[(int64_t)kof(s2) * n]; };
    auto loadV = [&](int s2, int slot) {
        const cz* vp = V + (int64_t)kof(s2) * ldw;
#pragma unroll
        for (int q = 0; q < 4; ++q) vr[slot][q] = vp[vcol[q]];
    };
    d4 t1[4], t2[4], t3[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { t1[q] = (d4){0, 0, 0, 0}; t2[q] = (d4){0, 0, 0, 0}; t3[q] = (d4){0, 0, 0, 0}; }
    if (Kd > 0) {
#pragma unroll
        for (int s2 = 0; s2 < DMV_DA - 1; ++s2) loadA(s2, s2);
#pragma unroll
        for (int s2 = 0; s2 < DMV_DV - 1; ++s2) loadV(s2, s2);
        auto step = [&](int s2, auto jc) {
            constexpr int j = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
            loadV(s2 + DMV_DV - 1, (j + DMV_DV - 1) % DMV_DV);
            loadA(s2 + DMV_DA - 1, (j + DMV_DA - 1) % DMV_DA);
            __builtin_amdgcn_sched_barrier(0);
            const bool keep = aok && 4 * s2 + lk < Kd;
            const double ax = keep ? ar[j % DMV_DA].x : 0.0, ay = keep ? ar[j % DMV_DA].y : 0.0, as = ax + ay;
#pragma unroll
            for (int q = 0; q < 4; ++q) t1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, vr[j % DMV_DV][q].x, t1[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) t2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, vr[j % DMV_DV][q].y, t2[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) t3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, vr[j % DMV_DV][q].x + vr[j % DMV_DV][q].y, t3[q], 0, 0, 0);
        };
        for (int s0 = 0; s0 < nks; s0 += 8) {      // (a step beyond the end multiplies masked zeros)
            step(s0 + 0, std::integral_constant<int, 0>{}); step(s0 + 1, std::integral_constant<int, 1>{});
            step(s0 + 2, std::integral_constant<int, 2>{}); step(s0 + 3, std::integral_constant<int, 3>{});
            step(s0 + 4, std::integral_constant<int, 4>{}); step(s0 + 5, std::integral_constant<int, 5>{});
            step(s0 + 6, std::integral_constant<int, 6>{}); step(s0 + 7, std::integral_constant<int, 7>{});
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = rbase + wave * 16 + lk + 4 * r;
        if (orow >= n) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = cg + q * 16 + li;
            if (col < ldw) {
                cz v = cz_mul(alpha, cz_make(t1[q][r] - t2[q][r], t3[q][r] - t1[q][r] - t2[q][r]));
                cz* d = out + (int64_t)orow * ldw + col;
                if (accumulate) v = cz_add(v, *d);
                *d = v;
            }
        }
    }
}

__global__ void k_dense_mv_reduce(int64_t total, int kparts, const cz* __restrict__ part, cz* out, cz alpha, int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    cz s = part[i];
    for (int z = 1; z < kparts; ++z) s = cz_add(s, part[i + (int64_t)z * total]);
    s = cz_mul(alpha, s);
    if (accumulate) s = cz_add(s, out[i]);
    out[i] = s;
}

int dense_mv_parts(int n, int ldw) {
    int wgs = ((n + 63) / 64) * ((ldw + LDS_COLS - 1) / LDS_COLS);
    int parts = 1;
    while (parts < 8 && wgs * parts * 2 <= 256 && (n + 63) / 64 >= parts * 8) parts *= 2;     // every slice keeps at least 8 K chunks
    return parts;
}

void launch_dense_mv(hipStream_t st, int n, const cz* A, const cz* V, int ldw, cz* out, cz alpha, int accumulate, cz* part) {
    const int parts = part ? dense_mv_parts(n, ldw) : 1;
    static const bool old_kernel = getenv("FEASTHIP_DENSE_MV_LDS") != nullptr;
    if (LDS_COLS == 64 && !old_kernel)
        hipLaunchKernelGGL(k_dense_mv2, dim3((n + 63) / 64, (ldw + 63) / 64, parts), dim3(256), 0, st, n, A, V, ldw, out, alpha, accumulate, parts, part);
    else
        hipLaunchKernelGGL(k_dense_mv, dim3((n + 63) / 64, (ldw + LDS_COLS - 1) / LDS_COLS, parts), dim3(256), 0, st, n, A, V, ldw, out, alpha, accumulate,
                           parts, part);
    if (parts > 1) {
        const int64_t total = (int64_t)n * ldw;
        hipLaunchKernelGGL(k_dense_mv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, parts, part, out, alpha, accumulate);
    }
}

// Tall-skinny projection G = V^H W on the f64 matrix cores.  Stage 1: a workgroup reduces a contiguous range of rows (about
// n / 512: two workgroups per CU, each with its accumulators in registers for the whole range, so a partial block is written
// once per workgroup and not once per 128 rows); wave jobs are (16-row i tile) x (64-col j group), operands are read straight
// from the row-major blocks (16 consecutive columns of a row per lane group = 256 B segments) through a ring of GRAM_ST
// stages of one k-step (4 rows, five 16-byte loads per lane): GRAM_ST - 1 stages are in flight while one runs on the matrix
// cores.  Stage 2 sums the partials in a fixed order.
// (Round 2: 128 rows per workgroup, one stage in flight, the row scaling multiplied into the operand right behind its load:
// 82 + 17 us for n = 97,336, m = 64, i.e. 31 TFLOP/s.)
constexpr int GRAM_ROWS = 128;      // smallest row range of a workgroup; gram_scratch_elems() is sized for it
constexpr int GRAM_ST = 5;

size_t gram_scratch_elems(int n, int m) {
    size_t chunks = ((size_t)n + GRAM_ROWS - 1) / GRAM_ROWS;
    size_t mp = (size_t)((m + 15) / 16) * 16;
    return chunks * mp * mp;
}

static int gram_rows_per_wg(int n) {
    int r = (n + 511) / 512;
    r = (r + 3) / 4 * 4;
    return r < GRAM_ROWS ? GRAM_ROWS : r;
}

template <bool SCALE>
__global__ __launch_bounds__(256, 2) void k_gram_partial(int n, int mp, int ldw, const cz* __restrict__ V, const cz* __restrict__ W,
                                                          cz* scratch, const cz* __restrict__ rowscale, int rows_per_wg) {
    constexpr int ST = SCALE ? GRAM_ST - 1 : GRAM_ST;      // (the scale factors of a stage cost four registers more)
    const int chunk = blockIdx.x;
    const int r0 = chunk * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int nit = mp / 16, njg = (mp + 63) / 64;
    cz* out = scratch + (size_t)chunk * mp * mp;
    for (int job = wave; job < nit * njg; job += 4) {
        const int it = job / njg, jg = job % njg;
        d4 t1[4], t2[4], t3[4];          // 3M product of conj(a) * b
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { t1[nt] = (d4){0, 0, 0, 0}; t2[nt] = (d4){0, 0, 0, 0}; t3[nt] = (d4){0, 0, 0, 0}; }
        int colb[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) colb[nt] = min(jg * 64 + nt * 16 + li, mp - 1);      // clamped: the extra columns are not stored
        cz a[ST], b[ST][4], sc[SCALE ? ST : 1];
        auto load4 = [&](int r, cz& aa, cz (&bb)[4], cz& ss) {
            const int64_t row = min(r + lk, r1 - 1);
            aa = V[row * ldw + it * 16 + li];
            if constexpr (SCALE) ss = rowscale[row];       // G = V^H diag(d) W: applied when the stage is consumed, not behind the load
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bb[nt] = W[row * ldw + colb[nt]];
        };
        auto mac4 = [&](int r, const cz& aa, const cz (&bb)[4], const cz& ss) {
            const bool ok = r + lk < r1;
            cz av = aa;
            if constexpr (SCALE) av = cz_mul(av, cz_conj(ss));                              // conj(v d~) w with d~ = conj(d)
            const double ar = ok ? av.x : 0.0, ai = ok ? -av.y : 0.0, as = ar + ai;         // conj(V)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t1[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bb[nt].x, t1[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bb[nt].y, t2[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t3[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bb[nt].x + bb[nt].y, t3[nt], 0, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < ST - 1; ++s) load4(r0 + 4 * s, a[s], b[s], sc[SCALE ? s : 0]);
        for (int r = r0; r < r1; r += 4 * ST) {
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                constexpr int NX = ST - 1;
                __builtin_amdgcn_sched_barrier(0);
                load4(r + 4 * (s + NX), a[(s + NX) % ST], b[(s + NX) % ST], sc[SCALE ? (s + NX) % ST : 0]);
                __builtin_amdgcn_sched_barrier(0);
                mac4(r + 4 * s, a[s], b[s], sc[SCALE ? s : 0]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int i = it * 16 + lk + 4 * r, j = jg * 64 + nt * 16 + li;
                if (j < mp) out[i + (size_t)j * mp] = cz_make(t1[nt][r] - t2[nt][r], t3[nt][r] - t1[nt][r] - t2[nt][r]);
            }
    }
}

// 64 consecutive elements per workgroup; the chunks are summed in 4 interleaved groups (one per wave, four
// independent accumulators each) that are combined in a fixed order: deterministic, and the 1 KiB lines of
// a chunk are read by one wave.
__global__ __launch_bounds__(256) void k_gram_reduce(int nchunks, int mp, int m, const cz* __restrict__ scratch, cz* G) {
    __shared__ cz part[4][64];
    int e = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    bool ok = e < m * m;
    int i = ok ? e % m : 0, j = ok ? e / m : 0;
    const cz* p = scratch + i + (size_t)j * mp;
    size_t cs = (size_t)mp * mp;
    cz a0 = cz_make(0, 0), a1 = a0, a2 = a0, a3 = a0;
    int c = g;
    for (; c + 12 < nchunks; c += 16) {
        a0 = cz_add(a0, p[(size_t)c * cs]); a1 = cz_add(a1, p[(size_t)(c + 4) * cs]);
        a2 = cz_add(a2, p[(size_t)(c + 8) * cs]); a3 = cz_add(a3, p[(size_t)(c + 12) * cs]);
    }
    for (; c < nchunks; c += 4) a0 = cz_add(a0, p[(size_t)c * cs]);
    part[g][threadIdx.x & 63] = cz_add(cz_add(a0, a1), cz_add(a2, a3));
    __syncthreads();
    if (g == 0 && ok) {
        int l = threadIdx.x;
        G[i + (size_t)j * m] = cz_add(cz_add(part[0][l], part[1][l]), cz_add(part[2][l], part[3][l]));
    }
}

void launch_gram(hipStream_t st, int n, int m, int ldw, const cz* V, const cz* W, cz* scratch, cz* G, const cz* rowscale) {
    const int rpw = gram_rows_per_wg(n);
    int chunks = (n + rpw - 1) / rpw;
    int mp = ((m + 15) / 16) * 16;    // ldw >= mp by construction
    if (rowscale) hipLaunchKernelGGL(k_gram_partial<true>, dim3(chunks), dim3(256), 0, st, n, mp, ldw, V, W, scratch, rowscale, rpw);
    else hipLaunchKernelGGL(k_gram_partial<false>, dim3(chunks), dim3(256), 0, st, n, mp, ldw, V, W, scratch, rowscale, rpw);
    hipLaunchKernelGGL(k_gram_reduce, dim3((m * m + 63) / 64), dim3(256), 0, st, chunks, mp, m, scratch, G);
}

// out = V * M  (V n x ldw row-major, M m x m column-major) on the f64 matrix cores.  Persistent workgroups: a workgroup
// stages its 64 output columns of M ONCE in LDS (row-major [k][64 cols], all m <= TS_KMAX rows) and then walks over 64-row
// tiles of V (one 16-row slab per wave; V fragment: row = lane & 15, k = lane >> 4, the four lane groups read 64 consecutive
// bytes of a row); the waves never synchronise after the staging, and the second workgroup of the CU computes while this
// one loads.  (One tile per workgroup with M re-staged every time: 86 us for n = 97,336, m = 64.)
constexpr int TS_KMAX = 64;       // rows of M held in LDS (64 KB: two workgroups per CU; 16 k-steps of V fragments per tile in registers, twice); larger m runs in K chunks

template <int KS>     // k-steps per pass (m rounded up to 4 KS, at most TS_KMAX / 4)
__device__ __forceinline__ void ts_tile_loop(int n, int m, int ldw, const cz* __restrict__ V, cz* out, const cz (*Ms)[64], int kc, int klen,
                                             int cg, bool accumulate, int tile0, int tstride, int ntiles) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    // the V fragment of a k-step is reloaded for the NEXT tile as soon as this tile's products with it are issued: the whole next tile
    // is in flight while this one runs on the matrix cores and writes its rows, in the same registers
    cz av[KS];
    {
        const int64_t arow = min(tile0 * 64 + wave * 16 + li, n - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = V[arow * ldw + min(kc + ks * 4 + lk, ldw - 1)];
    }
    for (int tile = tile0; tile < ntiles; tile += tstride) {
        asm volatile("" ::: "memory");     // M is the same for every tile: without this the compiler hoists all of its LDS reads (256 registers)
        const cz* vnext = V + (int64_t)min(min(tile + tstride, ntiles - 1) * 64 + wave * 16 + li, n - 1) * ldw;
        d4 t1[4], t2[4], t3[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { t1[nt] = (d4){0, 0, 0, 0}; t2[nt] = (d4){0, 0, 0, 0}; t3[nt] = (d4){0, 0, 0, 0}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool kok = ks * 4 + lk < klen;                  // padding columns of V never enter the product
            const double ar = kok ? av[ks].x : 0.0, ai = kok ? av[ks].y : 0.0, as = ar + ai;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                cz bv = Ms[ks * 4 + lk][nt * 16 + li];
                t1[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bv.x, t1[nt], 0, 0, 0);
                t2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bv.y, t2[nt], 0, 0, 0);
                t3[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bv.x + bv.y, t3[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            av[ks] = vnext[min(kc + ks * 4 + lk, ldw - 1)];
            __builtin_amdgcn_sched_barrier(0);
        }
        const int r0 = tile * 64 + wave * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int orow = r0 + lk + 4 * r;
            if (orow >= n) continue;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int col = cg + nt * 16 + li;
                if (col < ldw) {
                    cz v = cz_make(t1[nt][r] - t2[nt][r], t3[nt][r] - t1[nt][r] - t2[nt][r]);
                    if (accumulate) { cz o = out[(int64_t)orow * ldw + col]; v.x += o.x; v.y += o.y; }
                    out[(int64_t)orow * ldw + col] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_ts_gemm(int n, int m, int ldw, const cz* __restrict__ V, const cz* __restrict__ M, cz* out) {
    extern __shared__ cz ts_lds[];
    cz (*Ms)[64] = reinterpret_cast<cz (*)[64]>(ts_lds);
    const int tid = threadIdx.x;
    const int cg = blockIdx.y * 64, ntiles = (n + 63) / 64;
    for (int kc = 0; kc < m; kc += TS_KMAX) {
        // the tile loops walk 16, 32 or 64 rows of M: all of them are staged, the ones beyond klen as zeros.  (Until round 4 only
        // (klen + 3) & ~3 rows were: for m <= 12 the loop of 16 multiplied the masked -- zero -- entries of V with whatever LDS held
        // beyond the staged rows, and 0 x NaN is NaN: the intermittent all-NaN subspaces of the 8-column moment blocks.)
        const int klen = min(TS_KMAX, m - kc), krows = klen <= 16 ? 16 : klen <= 32 ? 32 : 64;
        __syncthreads();
        for (int idx = tid; idx < krows * 64; idx += 256) {
            int k = idx >> 6, c = idx & 63;
            Ms[k][c] = (k < klen && cg + c < m) ? M[(kc + k) + (size_t)(cg + c) * m] : cz_make(0, 0);
        }
        __syncthreads();
        const bool acc = kc > 0;
        if (krows <= 16) ts_tile_loop<4>(n, m, ldw, V, out, Ms, kc, klen, cg, acc, blockIdx.x, gridDim.x, ntiles);
        else if (krows <= 32) ts_tile_loop<8>(n, m, ldw, V, out, Ms, kc, klen, cg, acc, blockIdx.x, gridDim.x, ntiles);
        else ts_tile_loop<16>(n, m, ldw, V, out, Ms, kc, klen, cg, acc, blockIdx.x, gridDim.x, ntiles);
    }
}

void launch_ts_gemm(hipStream_t st, int n, int m, int ldw, const cz* V, const cz* M, cz* out) {
    const int ntiles = (n + 63) / 64, ncg = (ldw + 63) / 64;
    const int kmax = std::min(m, TS_KMAX), krows = kmax <= 16 ? 16 : kmax <= 32 ? 32 : 64;      // rows of M a tile loop walks (k_ts_gemm)
    const size_t lds = (size_t)krows * 64 * sizeof(cz);                 // 64 KB for m = 64: two workgroups per CU
    const int per_cu = 2;
    const int wgs = std::max(1, std::min(ntiles, 256 * per_cu / std::max(1, ncg) ));
    static std::atomic<uint64_t> attr_seen{0};
    if (first_on_this_device(attr_seen)) { (void)hipFuncSetAttribute((const void*)k_ts_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, TS_KMAX * 64 * (int)sizeof(cz)); }
    hipLaunchKernelGGL(k_ts_gemm, dim3(wgs, ncg), dim3(256), lds, st, n, m, ldw, V, M, out);
}

// column sums of |V|^2, two deterministic stages
constexpr int CN_ROWS = 128;
size_t colnorm_scratch_elems(int n, int ldw) { return ((size_t)n + CN_ROWS - 1) / CN_ROWS * (size_t)ldw; }

__global__ __launch_bounds__(256) void k_colnorm_partial(int n, int ldw, const cz* __restrict__ V, double* scratch) {
    __shared__ double red[4][64];
    int r0 = blockIdx.x * CN_ROWS, r1 = min(n, r0 + CN_ROWS);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int cg = 0; cg < ldw; cg += 64) {
        int col = cg + lane;
        double s = 0;
        if (col < ldw)
            for (int r = r0 + wave; r < r1; r += 4) { cz v = V[(int64_t)r * ldw + col]; s += v.x * v.x + v.y * v.y; }
        red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && col < ldw) scratch[(size_t)blockIdx.x * ldw + col] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        __syncthreads();
    }
}

// 64 columns per workgroup; wave w sums the chunks c = w, w + 16, ... (four loads in flight), the 16 partial sums are added in a
// fixed order.  (One thread per column walking all 760 chunks of C3 took 181 us.)
__global__ __launch_bounds__(1024) void k_colnorm_reduce(int nchunks, int ldw, const double* __restrict__ scratch, double* nrm2) {
    __shared__ double red[16][64];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int j = blockIdx.x * 64 + lane;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (j < ldw) {
        int c = wave;
        for (; c + 48 < nchunks; c += 64) {
            s0 += scratch[(size_t)c * ldw + j];
            s1 += scratch[(size_t)(c + 16) * ldw + j];
            s2 += scratch[(size_t)(c + 32) * ldw + j];
            s3 += scratch[(size_t)(c + 48) * ldw + j];
        }
        for (; c < nchunks; c += 16) s0 += scratch[(size_t)c * ldw + j];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && j < ldw) {
        double s = 0;
        for (int w = 0; w < 16; ++w) s += red[w][lane];
        nrm2[j] = s;
    }
}

void launch_colnorm2(hipStream_t st, int n, int ldw, const cz* V, double* scratch, double* nrm2) {
    int chunks = (n + CN_ROWS - 1) / CN_ROWS;
    hipLaunchKernelGGL(k_colnorm_partial, dim3(chunks), dim3(256), 0, st, n, ldw, V, scratch);
    hipLaunchKernelGGL(k_colnorm_reduce, dim3((ldw + 63) / 64), dim3(1024), 0, st, chunks, ldw, scratch, nrm2);
}

__global__ void k_scale_cols(int64_t total, int ldw, cz* V, const cz* __restrict__ s) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    V[e] = cz_mul(V[e], s[e % ldw]);
}

void launch_scale_cols(hipStream_t st, int n, int ldw, cz* V, const cz* s) {
    int64_t total = (int64_t)n * ldw;
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, ldw, V, s);
}

__global__ void k_axpby(int64_t total, cz* out, cz alpha, const cz* __restrict__ V, int accumulate) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    cz r = cz_mul(alpha, V[e]);
    if (accumulate) r = cz_add(r, out[e]);
    out[e] = r;
}

void launch_axpby(hipStream_t st, int n, int ldw, cz* out, cz alpha, const cz* V, int accumulate) {
    int64_t total = (int64_t)n * ldw;
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, out, alpha, V, accumulate);
}

__global__ void k_residual_combine(int64_t total, int ldw, const cz* __restrict__ AX, const cz* __restrict__ BX,
                                   const cz* __restrict__ lam, cz* R, const cz* __restrict__ rowscale) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    cz r = AX[e];
    cz bx = BX[e];
    if (rowscale) bx = cz_mul(bx, rowscale[e / ldw]);          // BX is X itself, B = diag(rowscale)
    cz_msub(r, bx, lam[e % ldw]);
    R[e] = r;
}

void launch_residual_combine(hipStream_t st, int n, int ldw, const cz* AX, const cz* BX, const cz* lam, cz* R, const cz* rowscale) {
    int64_t total = (int64_t)n * ldw;
    hipLaunchKernelGGL(k_residual_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, ldw, AX, BX, lam, R, rowscale);
}

__global__ void k_poly_combine(int64_t total, int ldw, const cz* __restrict__ MX, int64_t blk, int nterms,
                               const cz* __restrict__ lam, cz* R) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    cz l = lam[e % ldw];
    cz acc = MX[(int64_t)(nterms - 1) * blk + e];      // Horner in lambda
    for (int t = nterms - 2; t >= 0; --t) acc = cz_add(cz_mul(acc, l), MX[(int64_t)t * blk + e]);
    R[e] = acc;
}

void launch_poly_combine(hipStream_t st, int n, int ldw, const cz* MX, int64_t blk, int nterms, const cz* lam, cz* R) {
    int64_t total = (int64_t)n * ldw;
    hipLaunchKernelGGL(k_poly_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, ldw, MX, blk, nterms, lam, R);
}

// Block-Hankel matrix of the contour moments (src/nlfeast.jl:213-216, 270-273):
// H[(i n + r), j m0 + c] = Q_{i+j+shift}[r, c], i, j < mm; row-major (mm n) x ldh, zero beyond mm m0.
__global__ void k_hankel(int n, int m0, int ldw, int mm, int ldh, int shift, const cz* __restrict__ Qm, int64_t blk, cz* H) {
    int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= (int64_t)mm * n) return;
    int i = (int)(row / n), r = (int)(row % n);
    for (int col = lane; col < ldh; col += 64) {
        int j = col / m0, c = col % m0;
        H[row * ldh + col] = (j < mm) ? Qm[(int64_t)(i + j + shift) * blk + (int64_t)r * ldw + c] : cz_make(0, 0);
    }
}

void launch_hankel(hipStream_t st, int n, int m0, int ldw, int mm, int ldh, int shift, const cz* Qm, cz* H) {
    int64_t rows = (int64_t)mm * n;
    hipLaunchKernelGGL(k_hankel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, n, m0, ldw, mm, ldh, shift, Qm, (int64_t)n * ldw, H);
}

// out[r, j] = in[r, idx[j]] for j < mo (idx on device), zero for the padding columns
__global__ void k_gather_cols(int n, int ldi, const cz* __restrict__ in, int ldo, int mo, const int* __restrict__ idx, cz* out) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * ldo) return;
    int64_t r = e / ldo;
    int j = (int)(e % ldo);
    out[e] = (j < mo) ? in[r * ldi + idx[j]] : cz_make(0, 0);
}

void launch_gather_cols(hipStream_t st, int n, int ldi, const cz* in, int ldo, int mo, const int* idx, cz* out) {
    int64_t total = (int64_t)n * ldo;
    hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n, ldi, in, ldo, mo, idx, out);
}

// R[r, j] = sum_t tab[t, j] MX_t[r, j]   (T(lam_j) x_j for a NEP in split form T(z) = sum_t f_t(z) M_t, tab[t, j] = f_t(lam_j))
__global__ void k_coef_combine(int64_t total, int ldw, const cz* __restrict__ MX, int64_t blk, int nterms,
                               const cz* __restrict__ tab, cz* R) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    int j = (int)(e % ldw);
    cz acc = cz_make(0, 0);
    for (int t = 0; t < nterms; ++t) cz_madd(acc, tab[(size_t)t * ldw + j], MX[(int64_t)t * blk + e]);
    R[e] = acc;
}

void launch_coef_combine(hipStream_t st, int n, int ldw, const cz* MX, int64_t blk, int nterms, const cz* tab, cz* R) {
    int64_t total = (int64_t)n * ldw;
    hipLaunchKernelGGL(k_coef_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, ldw, MX, blk, nterms, tab, R);
}

__global__ void k_cm_to_rm(int n, int m0, int ldw, const cz* __restrict__ cm, int64_t ldcm, cz* rm) {
    __shared__ cz tile[32][33];
    int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: ty 0..7
    for (int jj = ty; jj < 32; jj += 8) {
        int i = i0 + tx, j = j0 + jj;
        tile[jj][tx] = (i < n && j < m0) ? cm[i + (int64_t)j * ldcm] : cz_make(0, 0);
    }
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {
        int i = i0 + ii, j = j0 + tx;
        if (i < n && j < ldw) rm[(int64_t)i * ldw + j] = tile[tx][ii];
    }
}

void launch_cm_to_rm(hipStream_t st, int n, int m0, int ldw, const cz* cm, int64_t ldcm, cz* rm) {
    hipLaunchKernelGGL(k_cm_to_rm, dim3((n + 31) / 32, (ldw + 31) / 32), dim3(256), 0, st, n, m0, ldw, cm, ldcm, rm);
}

__global__ void k_rm_to_cm(int n, int m0, int ldw, const cz* __restrict__ rm, cz* cm, int64_t ldcm) {
    __shared__ cz tile[32][33];
    int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int ii = ty; ii < 32; ii += 8) {
        int i = i0 + ii, j = j0 + tx;
        tile[ii][tx] = (i < n && j < ldw) ? rm[(int64_t)i * ldw + j] : cz_make(0, 0);
    }
    __syncthreads();
    for (int jj = ty; jj < 32; jj += 8) {
        int i = i0 + tx, j = j0 + jj;
        if (i < n && j < m0) cm[i + (int64_t)j * ldcm] = tile[tx][jj];
    }
}

void launch_rm_to_cm(hipStream_t st, int n, int m0, int ldw, const cz* rm, cz* cm, int64_t ldcm) {
    hipLaunchKernelGGL(k_rm_to_cm, dim3((n + 31) / 32, (m0 + 31) / 32), dim3(256), 0, st, n, m0, ldw, rm, cm, ldcm);
}

}  // namespace feasthip
