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
                                                      const int* __restrict__ rowinv, int n, int phase) {
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    int p = f.p, r0 = t.a * 64;
    // phase 0: destinations inside the pivot block (before the front is solved);
    // phase 1: destinations among the contribution rows (after F2a has written them)
    if (phase == 0 ? (r0 >= p) : (r0 + 64 <= p)) return;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int ci = 0; ci < f.nchild; ++ci) {
        FrontD g = fr[child_idx[f.child_ptr + ci]];
        const int* rc = rel + g.sptr;
        int ia = lower_bound_dev2(rc, g.u, r0), ib = lower_bound_dev2(rc, g.u, r0 + 64);
        for (int i = ia + wave; i < ib; i += 4) {
            int dst = rc[i];
            if ((dst < p) != (phase == 0)) continue;
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
                         const int* rowinv, int K, int phase) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_fwd_assemble, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, S.rel, S.child_idx, W, ws, ldw,
                       crow_parent, crs_parent, crow_child, crs_child, rowinv, S.n, phase);
}

// ---------------------------------------------------------------------------
// THE hot kernel.  Every step of the block substitution on the solve-ready factors
// (kernels_factor.hip: k_block_inverse / k_scale_panels) is  out (-)= A * V  with
// A a column-major slab of a factor panel, streamed from HBM exactly once per sweep,
// and V up to 64 columns of block-vector rows staged in LDS.  Workgroup = 64 output
// rows x 64 columns, wave w owns rows [16w, 16w+16): its A fragment is one 16-byte
// load per lane per k-step (16 consecutive rows = 256 B segments) that goes straight
// into v_mfma_f64_16x16x4_f64; V fragments are conflict-free ds_read_b128; D's layout
// (row = (l>>4)+4r, col = l&15) writes 16 consecutive columns of one block-vector row
// per lane group.  The K loop is software pipelined: the next chunk's A and V loads are
// in flight while the current chunk runs on the matrix cores.
// Tile kinds: see kernels.hpp (launch_apply).
// ---------------------------------------------------------------------------
constexpr int APPLY_KC = 32;

template <bool GATHER>
__global__ __launch_bounds__(256, 2) void k_apply(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                               const int* __restrict__ struct_idx, const cz* __restrict__ fact, int64_t fs,
                                               cz* W, cz* Y, int64_t ws, int ldw, cz* crow, int64_t crs) {
    __shared__ cz Vs[APPLY_KC][LDS_COLS];
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y, cg = blockIdx.z * LDS_COLS;
    FrontD f = fr[t.f];
    int kind = t.a;
    int p = f.p, u = f.u, ld = p + u;
    const cz* Lp = fact + (int64_t)k * fs + f.lp;
    int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    // ---- tile geometry ------------------------------------------------------
    const cz* A;
    int64_t lda;
    int rbase, rend, Kd, k0 = 0, mask = 0;     // mask 1: unit lower (F2b), 2: upper incl. diagonal (B3)
    const cz* Vsrc;                            // block vector the V rows come from
    int vrow0 = 0;                             // first V row (contiguous kinds)
    if (kind == 0) { k0 = t.c * SOLVE_NB; int b = min(SOLVE_NB, p - k0); rbase = k0 + b + t.b * 64; rend = p; A = Lp + (int64_t)k0 * ld; lda = ld; Kd = b; Vsrc = W; vrow0 = f.c0 + k0; }
    else if (kind == 1) { rbase = p + t.b * 64; rend = ld; A = Lp; lda = ld; Kd = p; Vsrc = W; vrow0 = f.c0; }
    else if (kind == 2) { k0 = t.b * SOLVE_NB; int b = min(SOLVE_NB, p - k0); rbase = k0; rend = k0 + b; A = Lp + (int64_t)k0 * ld; lda = ld; Kd = b; Vsrc = W; vrow0 = f.c0 + k0; mask = 1; }
    else if (kind == 3) { rbase = t.b * 64; rend = p; A = fact + (int64_t)k * fs + f.f12; lda = p; Kd = u; Vsrc = Y; }
    else if (kind == 4) { k0 = t.c * SOLVE_NB; int b = min(SOLVE_NB, p - k0); rbase = t.b * 64; rend = k0; A = Lp + (int64_t)k0 * ld; lda = ld; Kd = b; Vsrc = Y; vrow0 = f.c0 + k0; }
    else { k0 = t.b * SOLVE_NB; int b = min(SOLVE_NB, p - k0); rbase = k0; rend = k0 + b; A = Lp + (int64_t)k0 * ld; lda = ld; Kd = b; Vsrc = Y; vrow0 = f.c0 + k0; mask = 2; }
    const cz* Vk = Vsrc + (int64_t)k * ws;
    const int* sidx = struct_idx + f.sptr;
    int arow = rbase + wave * 16 + li;         // row of A this lane feeds to the MFMA
    bool aok = arow < rend;
    int adiag = arow - k0;                     // column (relative to k0) of this row's diagonal element
    bool cok = cg + lane < ldw;

    cz av[APPLY_KC / 4], an[APPLY_KC / 4], vn[APPLY_KC / 4];
    // Loads are unconditional (addresses clamped into the slab) and masked with selects afterwards, so
    // the compiler emits straight-line global_load_dwordx4 that all stay in flight together.
    const int arow_c = min(arow, rend - 1);
    const int vcol_c = min(cg + lane, ldw - 1);
    const int lo_keep = (mask == 2) ? adiag : 0;                 // keep gk in [lo_keep, hi_keep)
    const int hi_keep = (mask == 1) ? adiag : 0x7fffffff;
    const int unit_at = (mask == 1 && aok) ? adiag : -1;
    auto load_chunk = [&](int kc) {
        int vidx[APPLY_KC / 4];
#pragma unroll
        for (int q = 0; q < APPLY_KC / 4; ++q) {
            int gk = min(kc + wave + 4 * q, Kd - 1);
            vidx[q] = GATHER ? sidx[gk] : vrow0 + gk;
        }
#pragma unroll
        for (int ks = 0; ks < APPLY_KC / 4; ++ks) {
            int gk = kc + ks * 4 + lk;
            cz v = A[arow_c + (int64_t)min(gk, Kd - 1) * lda];
            bool keep = aok && gk < Kd && gk >= lo_keep && gk < hi_keep;
            v.x = keep ? v.x : (gk == unit_at ? 1.0 : 0.0);
            v.y = keep ? v.y : 0.0;
            an[ks] = v;
        }
#pragma unroll
        for (int q = 0; q < APPLY_KC / 4; ++q) {
            int gk = kc + wave + 4 * q;
            cz v = Vk[(int64_t)vidx[q] * ldw + vcol_c];
            bool keep = gk < Kd && cok;
            v.x = keep ? v.x : 0.0;
            v.y = keep ? v.y : 0.0;
            vn[q] = v;
        }
    };
    d4 cr[4], ci[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { cr[nt] = (d4){0, 0, 0, 0}; ci[nt] = (d4){0, 0, 0, 0}; }
    load_chunk(0);
    for (int kc = 0; kc < Kd; kc += APPLY_KC) {
        __syncthreads();                       // previous chunk's fragment reads are done
#pragma unroll
        for (int q = 0; q < APPLY_KC / 4; ++q) Vs[wave + 4 * q][lane] = vn[q];
#pragma unroll
        for (int ks = 0; ks < APPLY_KC / 4; ++ks) av[ks] = an[ks];
        __syncthreads();
        if (kc + APPLY_KC < Kd) load_chunk(kc + APPLY_KC);
#pragma unroll
        for (int ks = 0; ks < APPLY_KC / 4; ++ks) {
            int kk = ks * 4 + lk;
            cz bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = Vs[kk][nt * 16 + li];
            double ar = av[ks].x, ai = av[ks].y, nai = -av[ks].y;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cr[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bv[nt].x, cr[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ci[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bv[nt].y, ci[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cr[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, bv[nt].y, cr[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ci[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bv[nt].x, ci[nt], 0, 0, 0);
        }
    }
    // ---- epilogue: all 16 read-modify-write loads are issued together (clamped addresses), then stored
    bool overwrite = (kind == 1 || kind == 2 || kind == 5);     // kind 1 writes C_I = -P21 W_J (children are added afterwards)
    double sgn = (kind == 1) ? -1.0 : 1.0;
    cz* Ok = ((kind == 0) ? W : Y) + (int64_t)k * ws;
    cz* dptr[4];
    bool rok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int orow = rbase + wave * 16 + lk + 4 * r;
        rok[r] = orow < rend;
        int oc = min(orow, rend - 1);
        dptr[r] = (kind == 1) ? crow + (int64_t)k * crs + (f.crow + oc - p) * (int64_t)ldw
                              : Ok + (int64_t)(f.c0 + oc) * ldw;
    }
    cz old[4][4];
    if (!overwrite) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) old[r][nt] = dptr[r][min(cg + nt * 16 + li, ldw - 1)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = cg + nt * 16 + li;
            cz v;
            if (overwrite) v = cz_make(sgn * cr[nt][r], sgn * ci[nt][r]);
            else v = cz_make(old[r][nt].x - cr[nt][r], old[r][nt].y - ci[nt][r]);
            if (rok[r] && col < ldw) dptr[r][col] = v;
        }
}

void launch_apply(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const cz* fact, int64_t fs,
                  cz* W, cz* Y, int64_t ws, int ldw, cz* crow, int64_t crs, int K, bool gather) {
    if (ntiles <= 0) return;
    dim3 g(ntiles, K, (ldw + LDS_COLS - 1) / LDS_COLS);
    if (gather) hipLaunchKernelGGL(k_apply<true>, g, dim3(256), 0, st, S.fr, tiles, S.struct_idx, fact, fs, W, Y, ws, ldw, crow, crs);
    else hipLaunchKernelGGL(k_apply<false>, g, dim3(256), 0, st, S.fr, tiles, S.struct_idx, fact, fs, W, Y, ws, ldw, crow, crs);
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
