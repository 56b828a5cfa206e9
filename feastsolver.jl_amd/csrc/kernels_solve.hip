// Supernodal block forward/back substitution for ALL local contour nodes at
// once: Y_k = T(z_k)^-1 R for k in the local shard, m0 right-hand sides.
// GPU side of the reference's `left_divider` seam (src/feast.jl:61-66, 138-143,
// src/nlfeast.jl:41,49; zgetrs / umfpack_zl_solve in the reference) with the
// accumulation of src/feast.jl:68-70 as a deterministic epilogue kernel.
//
// Block vectors are ROW-major n x ldw (one row = ldw contiguous complex128), so a
// supernode's rows are one contiguous slab and update rows gather as whole 1 KiB
// lines.  The factor panels are streamed from HBM exactly once per sweep and go
// straight into the f64 matrix cores; the block-vector rows come the same way, per k-step,
// into the other operand registers (k_apply2).  With the conjugate fold of real pencils a
// unit's block vectors are 2 m0 columns wide: [R, conj(R)].
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>

namespace feasthip {

// The mixed-precision path (feast!'s mixed_prec, src/feast.jl:19-25) runs the same sweeps on complex64 copies of
// the factors and block vectors: the small kernels are templates over the element type T (cz or cf).
template <typename T> __device__ inline T el_make(double x, double y);
template <> __device__ inline cz el_make<cz>(double x, double y) { return cz_make(x, y); }
template <> __device__ inline cf el_make<cf>(double x, double y) { cf r; r.x = (float)x; r.y = (float)y; return r; }

// W_k[i,:] = src[perm[c0 + rowperm_k[i]], :]   (fill-reducing + pivot permutation folded into one gather)
// Conjugate fold (lds > ldw): a real pencil gives T(conj z) = conj(T(z)), so the system of the conjugate node,
// T(conj z)^-1 R = conj(T(z)^-1 conj(R)), rides along as ldw more right-hand-side columns conj(R) of the SAME factor.
// Rows that receive contributions of children (asmflag, by pre-pivot row) are left to k_fwd_assemble, whose first
// contribution writes src + contribution instead of reading the row back: one write here and one read there less per such row
// (two thirds of the rows of a nested-dissection tree).
// A wave owns 64 consecutive rows: the index chain (supernode start, pivot position, assembly flag, fill-reducing permutation: four
// dependent loads) is resolved for all of them at once, one row per lane, and the rows that are left are then copied four at a
// time by the whole wave.  (One row per wave with its own chain of dependent loads: 211 us per contour step on C3.)
template <typename T>
__global__ __launch_bounds__(256) void k_gather_rhs(const int* __restrict__ perm, const int* __restrict__ c0_of_row, int n,
                                                    const cz* __restrict__ src, int ldw, int lds, T* W, int64_t ws, const int* __restrict__ rowperm,
                                                    const int* __restrict__ asmflag, const int* __restrict__ depth_of_row, int only_depth) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + wave) * 64, k = blockIdx.y;
    if (i0 >= n) return;
    const int i = i0 + lane;
    int srow = -1;
    if (i < n && (only_depth < 0 || depth_of_row[i] == only_depth)) {      // (one tree level: the early forward sweep gathers level by level)
        const int pre = c0_of_row[i] + rowperm[(int64_t)k * n + i];
        if (!(asmflag && asmflag[pre])) srow = perm[pre];
    }
    unsigned long long todo = __ballot(srow >= 0);
    T* const Wk = W + (int64_t)k * ws + (int64_t)i0 * lds;
    while (todo) {
        int l[4];
        const cz* s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            l[u] = todo ? __builtin_ctzll(todo) : -1;
            if (todo) todo &= todo - 1;
            s[u] = l[u] >= 0 ? src + (int64_t)__builtin_amdgcn_readlane(srow, l[u]) * ldw : s[0];      // (a spare slot re-reads the first row)
        }
        for (int j = lane; j < lds; j += 64) {
            const int js = j < ldw ? j : j - ldw;
            cz v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = s[u][js];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (l[u] >= 0) Wk[(int64_t)l[u] * lds + j] = el_make<T>(v[u].x, j < ldw ? v[u].y : -v[u].y);
        }
    }
}

void launch_gather_rhs(hipStream_t st, const DevSym& S, const cz* src, int ldw, int lds, cz* W, int64_t ws,
                       const int* rowperm, int K, int only_depth) {
    hipLaunchKernelGGL(k_gather_rhs<cz>, dim3((S.n + 255) / 256, K), dim3(256), 0, st, S.perm, S.c0_of_row, S.n, src, ldw, lds, W, ws, rowperm, S.asmflag, S.depth_of_row, only_depth);
}
void launch_gather_rhs(hipStream_t st, const DevSym& S, const cz* src, int ldw, int lds, cf* W, int64_t ws,
                       const int* rowperm, int K, int only_depth) {
    hipLaunchKernelGGL(k_gather_rhs<cf>, dim3((S.n + 255) / 256, K), dim3(256), 0, st, S.perm, S.c0_of_row, S.n, src, ldw, lds, W, ws, rowperm, S.asmflag, S.depth_of_row, only_depth);
}

// Forward sweep, assembly of the children's contribution rows into the pivot rows of the parent (the rest of a child's rows is pulled
// into the parent's own contribution rows by the epilogue of the kind-1 tiles).  Destination-major: a wave owns 16 pivot rows and adds
// to each of them the children's rows that land on it, in child order (deterministic, no atomics), through the host-built pull map
// DevSym::pmap -- one read and one write of the destination row whatever the number of children, no index chain per child, no
// workgroup barrier.  (Rounds 1-3 went child by child through front record -> break points -> rel -> row: the same sums in the same
// order, 40 us per launch with one node on the GPU where this one takes 25.)
template <typename T>
__global__ __launch_bounds__(256) void k_fwd_assemble(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                       const int* __restrict__ pmap, T* W, int64_t ws, int ldw,
                                                       const T* __restrict__ crc, int64_t crsc, const int* __restrict__ rowinv, int n,
                                                       const int* __restrict__ perm, const cz* __restrict__ src, int ldsrc) {
    const Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    const FrontD f = fr[t.f];
    const int p = f.p, nc = f.nchild, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rw = t.a * 64 + wave * 16;            // first row of this wave
    if (rw >= p) return;
    T* const Wk = W + (int64_t)k * ws;
    const T* const Ck = crc + (int64_t)k * crsc;
    const int* const pm = pmap + f.pmap;
    // lane l < 16: where row rw + l of the pivot block lives in W (after this node's row interchanges), and the row of the right-hand
    // side it starts from when the gather is fused (rows that receive contributions are not written by k_gather_rhs)
    int drow = 0, frow = 0;
    if (lane < 16 && rw + lane < p) {
        drow = f.c0 + rowinv[(int64_t)k * n + f.c0 + rw + lane];
        if (src) frow = perm[f.c0 + rw + lane];
    }
    const int lu = lane >> 4, ls = lane & 15;       // lane = (row of the group of four, child slot)
    for (int q = 0; q < 16 && rw + q < p; q += 4) {
        unsigned started = 0;                       // rows of this group that already hold a sum (more than 16 children)
        for (int s0 = 0; s0 < nc; s0 += 16) {
            const int ns = min(16, nc - s0);
            const int mv = (ls < ns && rw + q + lu < p) ? pm[(int64_t)(rw + q + lu) * nc + s0 + ls] : -1;
            const unsigned long long hasm = __ballot(mv >= 0);
            if (hasm == 0ull) continue;
            T* d[4];
            const cz* fp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool has = ((hasm >> (16 * u)) & 0xffffull) != 0ull;
                const int dr = __builtin_amdgcn_readlane(drow, q + u), fr2 = __builtin_amdgcn_readlane(frow, q + u);
                d[u] = has ? Wk + (int64_t)dr * ldw : nullptr;
                fp[u] = (has && src && !((started >> u) & 1u)) ? src + (int64_t)fr2 * ldsrc : nullptr;
                if (has) started |= 1u << u;
            }
            for (int j = lane; j < ldw; j += 64) {
                const int js = j < ldsrc ? j : j - ldsrc;          // columns ldsrc .. of a folded block vector: conj(src)
                T a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (fp[u]) { const cz v = fp[u][js]; a[u] = el_make<T>(v.x, j < ldsrc ? v.y : -v.y); }
                    else if (d[u]) a[u] = d[u][j];
                }
                for (int s = 0; s < ns; s += 2) {
                    T b[4][2];
                    int ix[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ix[u][h] = (s + h < ns) ? __builtin_amdgcn_readlane(mv, 16 * u + min(s + h, 15)) : -1;
                            if (ix[u][h] >= 0) b[u][h] = Ck[(int64_t)ix[u][h] * ldw + j];
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            if (ix[u][h] >= 0) { a[u].x += b[u][h].x; a[u].y += b[u][h].y; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (d[u]) d[u][j] = a[u];
            }
        }
    }
}

void launch_fwd_assemble(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* W, int64_t ws, int ldw,
                         const cz* crow_child, int64_t crs_child, const int* rowinv, int K, const cz* src, int ldsrc) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_fwd_assemble<cz>, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, S.pmap, W, ws, ldw, crow_child, crs_child, rowinv, S.n,
                       S.perm, src, ldsrc);
}
void launch_fwd_assemble(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cf* W, int64_t ws, int ldw,
                         const cf* crow_child, int64_t crs_child, const int* rowinv, int K, const cz* src, int ldsrc) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_fwd_assemble<cf>, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, S.pmap, W, ws, ldw, crow_child, crs_child, rowinv, S.n,
                       S.perm, src, ldsrc);
}

// ---------------------------------------------------------------------------
// The hot kernel: every step of the block substitution on the solve-ready factors (kernels_factor.hip: k_couple32 /
// k_block_inverse, k_scale_panels) is  out (-)= A * V  with A a column-major slab of a factor panel, streamed from HBM exactly
// once per sweep, and V up to 64 columns of block-vector rows.  Tile kinds: see kernels.hpp (launch_apply).
// (Round 1 staged the V rows in LDS by LDS-DMA with one workgroup barrier per 32-column chunk -- k_apply, and k_apply32 for the
// complex64 sweeps; round 2's k_apply2 below replaced both, bit-identically, and they were removed in round 3.)
// ---------------------------------------------------------------------------
template <typename T>
struct ApplyGeomT {
    const T* A;          // factor slab, column-major
    int64_t lda;
    int rbase, rend, Kd, k0, mask;          // mask 1: unit lower (F2b), 2: upper incl. diagonal (B3), 3: lower incl. diagonal, 4: unit upper (adjoint sweeps)
    int trans;                              // 1: the operand is the conjugate transpose of the slab: element (row r, k) = conj(A[k + r lda]) (adjoint sweeps)
    const T* Vk;         // block vector the V rows come from (this node)
    const int* sidx;     // struct_idx of the front (gather kinds)
    int vrow0;           // first V row (contiguous kinds)
    T* out;              // output rows base: row r of the tile is out + r * ldw   (r relative to orow0)
    int orow_shift;      // output row index = orow + orow_shift
    int mode;            // 0: out -= acc, 1: out = acc, 2: out = -acc, 3: out = pulled child rows - acc
    const int* pull;     // mode 3: [output row][child slot] -> row of the children's contribution arena, or -1
    int npull;           // child slots per row
    const T* pullbase;   // children's contribution arena of this node
    // kind 6 (B1 of a front whose pivot block fits one row tile, fused with its B3): after  y' = y - U12 x_I  the tile goes
    // through LDS and is multiplied by the upper inverse D (p x p, leading dimension dld) in the same workgroup;  x = D y'  is
    // written to xout (row + orow_shift).  Kernels without the fusion treat the tile as kind 3 and run the B3 tiles separately.
    int fuse;
    const T* D;
    int dld, dp;
    T* xout;
};
typedef ApplyGeomT<cz> ApplyGeom;

// tile -> geometry (kernels.hpp: launch_apply lists the kinds); returns the number of rows of the tile
template <typename T>
__device__ __forceinline__ int decode_tile(ApplyGeomT<T>& g, const FrontD& f, const Tile& t, int k, const int* struct_idx,
                                           const T* fact, int64_t fs, T* W, T* Y, int64_t ws, int ldw, T* crow, int64_t crs,
                                           T* skbuf, int64_t sks, const int* cmap, const T* crow_child, int64_t crs_child) {
    int kind = t.a & 0xff;
    int th = ((t.a >> 8) & 0xff) ? 16 * ((t.a >> 8) & 0xff) : 64;   // tile height in rows (16, 32 or 64), chosen per launch by the host
    int nsplit = t.a >> 16;                            // kinds 1 and 3: number of K parts (0 = not split), part index in t.c
    int p = f.p, u = f.u, ld = p + u;
    const T* Lp = fact + (int64_t)k * fs + f.lp;
    T* Wk = W + (int64_t)k * ws;
    T* Yk = Y + (int64_t)k * ws;
    g.k0 = 0; g.mask = 0; g.trans = 0; g.vrow0 = 0; g.lda = ld; g.mode = 0; g.pull = nullptr; g.npull = 0; g.pullbase = nullptr;
    g.fuse = 0; g.D = nullptr; g.dld = 0; g.dp = 0; g.xout = nullptr;
    g.sidx = struct_idx + f.sptr;
    g.out = Yk; g.orow_shift = f.c0;
    const int NB = f.nb;         // diagonal super-block size of this front's solve-ready form (128, 256 or 512)
    if (kind == 0) { g.k0 = t.c * NB; int b = min(NB, p - g.k0); g.rbase = g.k0 + b + t.b * th; g.rend = p; g.A = Lp + (int64_t)g.k0 * ld; g.Kd = b; g.Vk = Wk; g.vrow0 = f.c0 + g.k0; g.out = Wk; }
    else if (kind == 1) {
        g.rbase = p + t.b * th; g.rend = ld; g.A = Lp; g.Kd = p; g.Vk = Wk; g.vrow0 = f.c0; g.out = crow + (int64_t)k * crs + f.crow * (int64_t)ldw; g.orow_shift = -p; g.mode = 2;
        if (f.cmap >= 0) { g.mode = 3; g.pull = cmap + f.cmap; g.npull = f.nchild; g.pullbase = crow_child + (int64_t)k * crs_child; }
        if (nsplit > 0) {   // split-K part t.c of the long products at the top of the tree when few nodes are left on this GPU: -P21[:, part] W[part]
                            // into the scratch slab (part, row); k_splitk_reduce_fwd adds the slabs and the children's rows
            int Kp = (((p + nsplit - 1) / nsplit) + 31) & ~31;
            int kbeg = t.c * Kp;
            g.A += (int64_t)kbeg * ld; g.vrow0 += kbeg; g.Kd = min(Kp, p - kbeg);
            g.out = skbuf + (int64_t)k * sks + (f.skf_off + (int64_t)t.c * u) * ldw; g.mode = 2; g.pull = nullptr; g.npull = 0;
        }
    }
    else if (kind == 2) {   // unit lower inverse: row tile c only needs the columns left of its last row
        g.k0 = t.b * NB; int b = min(NB, p - g.k0); g.rbase = g.k0 + t.c * th; g.rend = g.k0 + b; g.A = Lp + (int64_t)g.k0 * ld; g.Kd = min(b, t.c * th + th); g.Vk = Wk; g.vrow0 = f.c0 + g.k0; g.mask = 1; g.mode = 1;
    }
    else if (kind == 3 || kind == 6) {
        g.rbase = t.b * th; g.rend = p; g.A = fact + (int64_t)k * fs + f.f12; g.lda = p; g.Kd = u; g.Vk = Wk;
        if (kind == 6) { g.fuse = 1; g.D = Lp; g.dld = ld; g.dp = p; g.xout = Wk; }
        if (nsplit > 0) {   // split-K part t.c: partial product into the scratch slab (part, row), reduced afterwards
            int Kp = (((u + nsplit - 1) / nsplit) + 31) & ~31;
            int kbeg = t.c * Kp;
            g.A += (int64_t)kbeg * p; g.sidx += kbeg; g.Kd = min(Kp, u - kbeg);
            g.out = skbuf + (int64_t)k * sks + (f.sk_off + (int64_t)t.c * p) * ldw; g.orow_shift = 0; g.mode = 2;
        }
    }
    else if (kind == 4) { g.k0 = t.c * NB; int b = min(NB, p - g.k0); g.rbase = t.b * th; g.rend = g.k0; g.A = Lp + (int64_t)g.k0 * ld; g.Kd = b; g.Vk = Yk; g.vrow0 = f.c0 + g.k0; }
    else if (kind == 5) {   // upper inverse: row tile c only needs the columns from its first row on
        int kb = t.b * NB, b = min(NB, p - kb), ks = t.c * th;
        g.rbase = kb + ks; g.rend = kb + b; g.k0 = kb + ks; g.A = Lp + (int64_t)g.k0 * ld; g.Kd = b - ks; g.Vk = Yk; g.vrow0 = f.c0 + g.k0; g.mask = 2; g.mode = 1; g.out = Wk;
    }
    // ---- adjoint sweeps: T(z)^H x = r over the SAME solve-ready factors (kernels.hpp, kinds 8 .. 13).  Every operand is the conjugate
    //      transpose of a slab the primal sweeps read: element (output row r, k) = conj(slab[k + r lda]).
    else if (kind == 8) {   // forward: z_b = inv(U_bb)^H w_b, lower incl. diagonal: row tile c needs the columns up to its last row
        g.k0 = t.b * NB; int b = min(NB, p - g.k0); g.rbase = g.k0 + t.c * th; g.rend = g.k0 + b; g.A = Lp + g.k0; g.trans = 1;
        g.Kd = min(b, t.c * th + th); g.Vk = Wk; g.vrow0 = f.c0 + g.k0; g.mask = 3; g.mode = 1;
    }
    else if (kind == 9) {   // forward step c >= 1, rows of block c: z_c -= (PU_[0, k0), c)^H z_[0, k0)
        int kc0 = t.c * NB, b = min(NB, p - kc0); g.rbase = kc0 + t.b * th; g.rend = kc0 + b; g.A = Lp; g.trans = 1; g.Kd = kc0; g.Vk = Yk; g.vrow0 = f.c0;
    }
    else if (kind == 10) {  // forward, contribution rows: C_I = sum_children C_child[I] - U12^H z_J   (K = p)
        g.rbase = p + t.b * th; g.rend = ld; g.A = fact + (int64_t)k * fs + f.f12 - (int64_t)p * p; g.lda = p; g.trans = 1; g.Kd = p; g.Vk = Yk; g.vrow0 = f.c0;
        g.out = crow + (int64_t)k * crs + f.crow * (int64_t)ldw; g.orow_shift = -p; g.mode = 2;
        if (f.cmap >= 0) { g.mode = 3; g.pull = cmap + f.cmap; g.npull = f.nchild; g.pullbase = crow_child + (int64_t)k * crs_child; }
    }
    else if (kind == 11) {  // t_b = inv(L_bb)^H z_b, unit upper: row tile c needs the columns from its first row on
        int kb = t.b * NB, b = min(NB, p - kb), ks = t.c * th;
        g.rbase = kb + ks; g.rend = kb + b; g.k0 = kb + ks; g.A = Lp + g.k0; g.trans = 1; g.Kd = b - ks; g.Vk = Yk; g.vrow0 = f.c0 + g.k0; g.mask = 4; g.mode = 1; g.out = Wk;
    }
    else if (kind == 12) {  // backward, pivot rows: W_J -= P21^H W[struct]   (K = u, gathered)
        // (the gathered rows are the ancestors' solution at its PRE-pivot positions: the Y argument of the launch -- the unpermuted copy
        // of k_adj_unpermute, or W itself when the factorisation made no interchanges)
        g.rbase = t.b * th; g.rend = p; g.A = Lp + p; g.trans = 1; g.Kd = u; g.Vk = Yk; g.out = Wk;
        if (nsplit > 0) {
            int Kp = (((u + nsplit - 1) / nsplit) + 31) & ~31;
            int kbeg = t.c * Kp;
            g.A += kbeg; g.sidx += kbeg; g.Kd = min(Kp, u - kbeg);
            g.out = skbuf + (int64_t)k * sks + (f.sk_off + (int64_t)t.c * p) * ldw; g.orow_shift = 0; g.mode = 2;
        }
    }
    else {                  // kind 13, backward step c, rows of block c: W_c -= (P_[k0 + b, p), c)^H W_[k0 + b, p)
        int kc0 = t.c * NB, b = min(NB, p - kc0); g.rbase = kc0 + t.b * th; g.rend = kc0 + b; g.A = Lp + (kc0 + b); g.trans = 1; g.Kd = p - kc0 - b;
        g.Vk = Wk; g.vrow0 = f.c0 + kc0 + b; g.out = Wk;
    }
    return min(th, g.rend - g.rbase);
}

template <typename T>
static void launch_apply2(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const T* fact, int64_t fs,
                          T* W, T* Y, int64_t ws, int ldw, T* crow, int64_t crs, int K, bool gather, T* skbuf, int64_t sks,
                          const T* crow_child, int64_t crs_child, int max_slabs, const XcdParts* xpp);

void launch_apply(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const cz* fact, int64_t fs,
                  cz* W, cz* Y, int64_t ws, int ldw, cz* crow, int64_t crs, int K, bool gather, cz* skbuf, int64_t sks,
                  const cz* crow_child, int64_t crs_child, int max_slabs, const XcdParts* xp) {
    if (ntiles <= 0) return;
    launch_apply2<cz>(st, S, tiles, ntiles, fact, fs, W, Y, ws, ldw, crow, crs, K, gather, skbuf, sks, crow_child, crs_child, max_slabs, xp);
}

// ---------------------------------------------------------------------------
// k_apply2 -- no LDS staging, no workgroup barriers.
// Every wave owns 16 output rows x 16*NS block-vector columns and streams BOTH operands straight from global memory into the
// matrix-core operand registers: the factor slab (HBM) A2_DA - 1 k-steps ahead, the block-vector rows (they are shared by
// the waves and tiles of a front, so they come from L1 / L2) A2_DV - 1 k-steps ahead -- a 16 B load per lane IS the MFMA
// fragment layout for both (A: 16 consecutive rows of one factor column, V: 16 consecutive columns of one block-vector row).
// The loop body is straight-line code (8 k-steps, uniform exits every second step), so the compiler's s_waitcnt vmcnt(N) waits
// for exactly the loads a k-step consumes while 15-20 younger ones stay in flight; waves never wait for each other.
// Measured on random operands (tools/probes/apply2_bench.hip): +21 % on matrix-core-bound tiles (K = 1024: 60.7 -> 73.8 TFLOP/s),
// +12 % on short-K tiles against the LDS-staged kernel of round 1; bit-identical results (same summation order).
// ---------------------------------------------------------------------------
// Ring depths and occupancy (tools/probes/apply2_variants.sh rebuilds this file with other values and times the bench): until round 5 the
// factor ring held 8 k-steps everywhere and the 32-row instances ran 3 workgroups per CU.  The counters of round 5 showed the small-front
// levels latency-bound, not HBM-bound (3.5 TB/s, matrix cores 46 % busy): 4 k-steps in flight leave the contiguous 32-row instance at 128
// registers = 4 workgroups per CU (the gathered one would spill: it stays at 3), and the shallower ring costs the long-K tiles nothing:
// C3 contour step 15.07 -> 14.67 ms of k_apply2 (58.7 -> 60.3 iterations/s on the same box).
#ifndef FEASTHIP_A2_DA4
#define FEASTHIP_A2_DA4 4
#endif
#ifndef FEASTHIP_A2_DA2
#define FEASTHIP_A2_DA2 4
#endif
#ifndef FEASTHIP_A2_OCCN
#define FEASTHIP_A2_OCCN 4
#endif
#ifndef FEASTHIP_A2_OCCG
#define FEASTHIP_A2_OCCG 3
#endif
#ifndef FEASTHIP_A2_DV4
#define FEASTHIP_A2_DV4 4
#endif
#ifndef FEASTHIP_A2_DV2
#define FEASTHIP_A2_DV2 4
#endif
constexpr int A2_DI = 2;

// element type of a sweep: complex128 on the f64 matrix cores, or complex64 (feast!'s mixed_prec) on the f32 ones
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T> struct A2;
template <> struct A2<cz> {
    typedef double real;
    typedef d4 acc;
    static __device__ __forceinline__ acc mfma(real a, real b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lk, int r) { return lk + 4 * r; }     // D layout: register r of lane l holds row (l >> 4) + 4 r
};
template <> struct A2<cf> {
    typedef float real;
    typedef f4 acc;
    static __device__ __forceinline__ acc mfma(real a, real b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lk, int r) { return 4 * lk + r; }     // row 4 (l >> 4) + r
};

template <typename T, bool GATHER, int NS>
__device__ __forceinline__ void apply2_body(const ApplyGeomT<T>& g, int cg, int ldw, T* Ys = nullptr) {
    typedef typename A2<T>::real real;
    typedef typename A2<T>::acc acc;
    constexpr int A2_DA = NS >= 4 ? FEASTHIP_A2_DA4 : FEASTHIP_A2_DA2;      // factor ring: k-steps in flight (must divide 8)
    constexpr int A2_DV = NS >= 4 ? FEASTHIP_A2_DV4 : FEASTHIP_A2_DV2;      // block-vector ring
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wave = tid >> 6;
    const int slab = wave & (NS - 1), ct0 = (wave / NS) * NS;        // column tiles [ct0, ct0 + NS)
    const bool fused = GATHER && g.fuse != 0;                       // kind 6: the tile ends with a barrier, so every wave stays (its rows
                                                                    // and columns are clamped and masked like any ragged tile's)
    if (!fused && (g.rbase + slab * 16 >= g.rend || cg + ct0 * 16 >= ldw)) return;   // nothing to do for this wave (no barriers: it may leave)
    const int arow = g.rbase + slab * 16 + li;
    const bool aok = arow < g.rend;
    const bool aok_all = g.rbase + slab * 16 + 16 <= g.rend;        // wave-uniform: the whole slab has rows
    const int adiag = arow - g.k0;
    const int Kd = g.Kd, nks = (Kd + 3) >> 2;
    const int lo_keep = (g.mask == 2) ? adiag : (g.mask == 4) ? adiag + 1 : 0;      // keep gk in [lo_keep, hi_keep)
    const int hi_keep = (g.mask == 1) ? adiag : (g.mask == 3) ? adiag + 1 : 0x7fffffff;
    const int unit_at = ((g.mask == 1 || g.mask == 4) && aok) ? adiag : -1;
    // adjoint sweeps (g.trans): the operand element of (row, k) sits at A[k + row lda] and enters conjugated; the lanes of a row
    // group then read 64 contiguous bytes of a column per k-step, consecutive k-steps the next 64
    const int lda = g.trans ? 1 : (int)g.lda;       // address step per k.  A panel's leading dimension fits 31 bits: one v_mad_i64_i32 per address
    const T* Ap = g.A + (int64_t)min(arow, g.rend - 1) * (g.trans ? g.lda : 1);
    const real cjs = g.trans ? (real)-1 : (real)1;
    // Column of lane li in column tile q.  complex128: 16 consecutive columns per tile, one 16-byte load per lane and tile.
    // complex64: two tiles share a 32-column span, lane li owning the columns 2 li (even tile) and 2 li + 1 (odd tile), so that one
    // 16-byte load (store) per lane serves both -- the complex64 sweeps are bound by the number of vector memory instructions,
    // not by bytes or matrix-core time.  (ldw is a multiple of 16, so a pair never straddles the end of a row.)
    constexpr bool PAIR = sizeof(T) == 8 && NS >= 2;
    int vcol[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q)
        vcol[q] = PAIR ? min(cg + (ct0 + (q & ~1)) * 16 + 2 * li, ldw - 2) + (q & 1) : min(cg + (ct0 + q) * 16 + li, ldw - 1);
    auto ld_cols = [&](T (&dst)[NS], const T* rowp, bool imm) {      // the lane's NS block-vector entries of one row (imm: rowp points at vcol[0])
        if constexpr (PAIR) {
#pragma unroll
            for (int q = 0; q < NS; q += 2) {
                const float4 w = *reinterpret_cast<const float4*>(imm ? rowp + q * 16 : rowp + vcol[q]);
                dst[q].x = w.x; dst[q].y = w.y; dst[q + 1].x = w.z; dst[q + 1].y = w.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) dst[q] = imm ? rowp[q * 16] : rowp[vcol[q]];
        }
    };
    T ar[A2_DA], vr[A2_DV][NS];
    int ir[8];
    auto kof = [&](int s) { return min(4 * s + lk, Kd - 1); };        // clamped: steps beyond the end re-read the last column (masked)
#ifndef FEASTHIP_A2_ABLATE
#define FEASTHIP_A2_ABLATE 0      // timing experiments (tools/probes/ablate2_bench.sh), bit mask: 1 no factor loads, 2 no block-vector loads, 4 no MFMA
#endif
    auto loadA = [&](int s, int slot) {
#if FEASTHIP_A2_ABLATE & 1
        ar[slot] = cz_make(1.0 + s, lane);
#else
        ar[slot] = Ap[(int64_t)kof(s) * lda];
#endif
    };
    auto loadI = [&](int s, int slot) { if (GATHER) ir[slot] = g.sidx[kof(s)]; };
    auto loadV = [&](int s, int slot, int islot) {
        const int vrow = GATHER ? ir[islot] : g.vrow0 + kof(s);
        const T* vp = g.Vk + (int64_t)vrow * ldw;
#if FEASTHIP_A2_ABLATE & 2
#pragma unroll
        for (int q = 0; q < NS; ++q) { vr[slot][q].x = 0.5 + q; vr[slot][q].y = (double)(size_t)vp * 1e-30; }
#else
        ld_cols(vr[slot], vp, false);
#endif
    };
    acc t1[NS], t2[NS], t3[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) { t1[q] = (acc){0, 0, 0, 0}; t2[q] = (acc){0, 0, 0, 0}; t3[q] = (acc){0, 0, 0, 0}; }
    // prologue: fill the rings
#pragma unroll
    for (int s = 0; s < A2_DV - 1 + A2_DI; ++s) loadI(s, s % 8);
#pragma unroll
    for (int s = 0; s < A2_DA - 1; ++s) loadA(s, s);
#pragma unroll
    for (int s = 0; s < A2_DV - 1; ++s) loadV(s, s, s % 8);
    auto step = [&](int s, auto jc) {
        constexpr int j = decltype(jc)::value;
        __builtin_amdgcn_sched_barrier(0);
        // order matters: vmcnt counts in order, so the wait for a block-vector row covers every load issued before it.  With the
        // factor load LAST in its step, the row loads of step s + 3 only wait for factor loads of earlier steps.
        loadI(s + A2_DV - 1 + A2_DI, (j + A2_DV - 1 + A2_DI) % 8);
        loadV(s + A2_DV - 1, (j + A2_DV - 1) % A2_DV, (j + A2_DV - 1) % 8);
        loadA(s + A2_DA - 1, (j + A2_DA - 1) % A2_DA);
        __builtin_amdgcn_sched_barrier(0);
        const int gk = 4 * s + lk;
        T a = ar[j % A2_DA];
        const bool keep = aok && gk < Kd && gk >= lo_keep && gk < hi_keep;
        a.x = keep ? a.x : (gk == unit_at ? (real)1 : (real)0);
        a.y = keep ? a.y : (real)0;
        const real as = a.x + cjs * a.y;       // conj(a) v by the 3M scheme: t2 accumulates +a.y v.y, t3 (a.x - a.y)(v.x + v.y); the epilogue takes the signs
        // 3M complex product, see apply_body
#if FEASTHIP_A2_ABLATE & 4
#pragma unroll
        for (int q = 0; q < NS; ++q) { t1[q][0] += a.x * vr[j % A2_DV][q].x; t2[q][0] += a.y * vr[j % A2_DV][q].y; t3[q][0] += as; }
        return;
#endif
#pragma unroll
        for (int q = 0; q < NS; ++q) t1[q] = A2<T>::mfma(a.x, vr[j % A2_DV][q].x, t1[q]);
#pragma unroll
        for (int q = 0; q < NS; ++q) t2[q] = A2<T>::mfma(a.y, vr[j % A2_DV][q].y, t2[q]);
#pragma unroll
        for (int q = 0; q < NS; ++q) t3[q] = A2<T>::mfma(as, vr[j % A2_DV][q].x + vr[j % A2_DV][q].y, t3[q]);
    };
    // Bulk of a long K loop: every k-step it touches (the one it computes and the ones it prefetches) is complete and unmasked, so the
    // addresses advance by running pointers, the column tiles sit at immediate offsets of one row pointer and nothing is clamped or
    // masked: about half the vector instructions of the generic step (which matters for the waves that own few columns).
    int s0 = 0;
    const bool fast_ok = g.mask == 0 && aok_all && cg + (ct0 + NS) * 16 <= ldw;      // (then no column is clamped: the pairs sit 32 entries apart)
    if (fast_ok) {
        const int nfull = Kd >> 2;                                   // complete k-steps
        const T* pAi = Ap + (int64_t)(4 * (A2_DA - 1) + lk) * lda;  // next factor load to issue: step DA - 1
        const int64_t stepA = 4 * (int64_t)lda;
        const T* pVi = g.Vk + (int64_t)(g.vrow0 + 4 * (A2_DV - 1) + lk) * ldw + vcol[0];    // contiguous kinds: step DV - 1
        const int64_t stepV = 4 * (int64_t)ldw;
        const int* pIi = g.sidx + 4 * (A2_DV - 1 + A2_DI) + lk;       // gathered kinds: next index to load
        const T* vbase = g.Vk + vcol[0];
        auto fstep = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
            if (GATHER) { ir[(j + A2_DV - 1 + A2_DI) % 8] = *pIi; pIi += 4; }
            {
                const T* vp = GATHER ? vbase + (int64_t)ir[(j + A2_DV - 1) % 8] * ldw : pVi;
#if FEASTHIP_A2_ABLATE & 2
#pragma unroll
                for (int q = 0; q < NS; ++q) { vr[(j + A2_DV - 1) % A2_DV][q].x = 0.5 + q; vr[(j + A2_DV - 1) % A2_DV][q].y = (double)(size_t)vp * 1e-30; }
#else
                ld_cols(vr[(j + A2_DV - 1) % A2_DV], vp, true);
#endif
                pVi += stepV;
            }
#if FEASTHIP_A2_ABLATE & 1
            ar[(j + A2_DA - 1) % A2_DA] = cz_make(1.0 + j, lane); pAi += stepA;
#else
            ar[(j + A2_DA - 1) % A2_DA] = *pAi; pAi += stepA;
#endif
            __builtin_amdgcn_sched_barrier(0);
            const T a = ar[j % A2_DA];
            const real as = a.x + cjs * a.y;
#if FEASTHIP_A2_ABLATE & 4
#pragma unroll
            for (int q = 0; q < NS; ++q) { t1[q][0] += a.x * vr[j % A2_DV][q].x; t2[q][0] += a.y * vr[j % A2_DV][q].y; t3[q][0] += as; }
            return;
#endif
#pragma unroll
            for (int q = 0; q < NS; ++q) t1[q] = A2<T>::mfma(a.x, vr[j % A2_DV][q].x, t1[q]);
#pragma unroll
            for (int q = 0; q < NS; ++q) t2[q] = A2<T>::mfma(a.y, vr[j % A2_DV][q].y, t2[q]);
#pragma unroll
            for (int q = 0; q < NS; ++q) t3[q] = A2<T>::mfma(as, vr[j % A2_DV][q].x + vr[j % A2_DV][q].y, t3[q]);
        };
        // a pass computes steps s0 .. s0 + 7 and prefetches up to step s0 + 7 + DA - 1 (factor) / + DV - 1 + DI (indices, gathered kinds): all
        // of them must be < nfull -- with the 8-deep factor ring of rounds 2-4 the factor bound implied the index bound; the 4-deep ring of
        // round 5 does not (a first build read struct_idx two k-steps past a short gather: memory access fault in the seam test)
        constexpr int AHEAD = (GATHER && A2_DV - 1 + A2_DI > A2_DA - 1) ? A2_DV - 1 + A2_DI : (A2_DV - 1 > A2_DA - 1 ? A2_DV - 1 : A2_DA - 1);
        for (; s0 + 8 + AHEAD <= nfull; s0 += 8) {
            fstep(std::integral_constant<int, 0>{}); fstep(std::integral_constant<int, 1>{}); fstep(std::integral_constant<int, 2>{}); fstep(std::integral_constant<int, 3>{});
            fstep(std::integral_constant<int, 4>{}); fstep(std::integral_constant<int, 5>{}); fstep(std::integral_constant<int, 6>{}); fstep(std::integral_constant<int, 7>{});
        }
    }
#define FEASTHIP_STEP2(S0, J) step((S0) + (J), std::integral_constant<int, (J)>{}); step((S0) + (J) + 1, std::integral_constant<int, (J) + 1>{});
    for (; s0 + 8 <= nks; s0 += 8) { FEASTHIP_STEP2(s0, 0) FEASTHIP_STEP2(s0, 2) FEASTHIP_STEP2(s0, 4) FEASTHIP_STEP2(s0, 6) }
    const int rem = nks - s0;       // 0 .. 7 k-steps left: pairs of steps (a step beyond the end multiplies masked zeros)
    if (rem > 0) {
        FEASTHIP_STEP2(s0, 0)
        if (rem > 2) {
            FEASTHIP_STEP2(s0, 2)
            if (rem > 4) {
                FEASTHIP_STEP2(s0, 4)
                if (rem > 6) { FEASTHIP_STEP2(s0, 6) }
            }
        }
    }
#undef FEASTHIP_STEP2
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: output rows of this lane (D layout: row (l >> 4) + 4 r, columns (ct0 + q) * 16 + li)
    T* dptr[4];
    bool rok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int orow = g.rbase + slab * 16 + A2<T>::drow(lk, r);
        rok[r] = orow < g.rend;
        dptr[r] = g.out + (int64_t)(min(orow, g.rend - 1) + g.orow_shift) * ldw;
    }
    T old[4][NS];
    if (g.mode == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ld_cols(old[r], dptr[r], false);
    } else if (g.mode == 3) {
        // extend-add of the forward sweep fused into the producer (fixed child order: deterministic)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < NS; ++q) { old[r][q].x = 0; old[r][q].y = 0; }
        for (int c = 0; c < g.npull; ++c) {
            int src[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int orow = g.rbase + slab * 16 + A2<T>::drow(lk, r);
                src[r] = rok[r] ? g.pull[(int64_t)(orow + g.orow_shift) * g.npull + c] : -1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (src[r] >= 0) {
                    T pv[NS];
                    ld_cols(pv, g.pullbase + (int64_t)src[r] * ldw, false);
#pragma unroll
                    for (int q = 0; q < NS; ++q) { old[r][q].x += pv[q].x; old[r][q].y += pv[q].y; }
                }
        }
    }
    const real sgn = (g.mode == 2) ? (real)-1 : (real)1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T o[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const real cr = t1[q][r] - cjs * t2[q][r], ci = t3[q][r] - t1[q][r] - cjs * t2[q][r];
            if (g.mode == 1 || g.mode == 2) { o[q].x = sgn * cr; o[q].y = sgn * ci; }
            else { o[q].x = old[r][q].x - cr; o[q].y = old[r][q].y - ci; }
        }
        if constexpr (GATHER) {
            if (fused) {     // y' goes to LDS (rows beyond the pivot block as zeros: the second product runs over whole k-steps)
                T* yr = Ys + (slab * 16 + A2<T>::drow(lk, r)) * LDS_COLS;
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const int lc = PAIR ? (ct0 + (q & ~1)) * 16 + 2 * li + (q & 1) : (ct0 + q) * 16 + li;
                    T v = o[q];
                    if (!rok[r]) { v.x = 0; v.y = 0; }
                    yr[lc] = v;
                }
                continue;
            }
        }
        if constexpr (PAIR) {
#pragma unroll
            for (int q = 0; q < NS; q += 2) {
                const int col = cg + (ct0 + q) * 16 + 2 * li;
                if (rok[r] && col < ldw) *reinterpret_cast<float4*>(dptr[r] + col) = make_float4(o[q].x, o[q].y, o[q + 1].x, o[q + 1].y);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int col = cg + (ct0 + q) * 16 + li;
                if (rok[r] && col < ldw) dptr[r][col] = o[q];
            }
        }
    }
    if constexpr (GATHER) {
        if (!fused) return;
        // ---- x = D y' : the upper inverse of the pivot block times the tile just computed (the B3 step of this front), same
        // k order and 3M products as the kind-5 tile it replaces
        __syncthreads();
        const int p2 = g.dp, nk2 = (p2 + 3) >> 2;
        const int r2 = slab * 16 + li;
        const T* Dp = g.D + min(r2, p2 - 1);
        acc u1[NS], u2[NS], u3[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) { u1[q] = (acc){0, 0, 0, 0}; u2[q] = (acc){0, 0, 0, 0}; u3[q] = (acc){0, 0, 0, 0}; }
        constexpr int MAXK2 = 4 * NS;                    // k-steps of a pivot block of at most 16 NS rows
        T dreg[MAXK2];
#pragma unroll
        for (int s2 = 0; s2 < MAXK2; ++s2) dreg[s2] = Dp[(int64_t)min(4 * s2 + lk, p2 - 1) * g.dld];
#pragma unroll
        for (int s2 = 0; s2 < MAXK2; ++s2) {
            if (s2 < nk2) {
                const int gk = 4 * s2 + lk;
                const bool keep = r2 < p2 && gk < p2 && gk >= r2;
                T a = dreg[s2];
                a.x = keep ? a.x : (real)0;
                a.y = keep ? a.y : (real)0;
                const real as = a.x + a.y;
                const T* yk = Ys + gk * LDS_COLS;
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const int lc = PAIR ? (ct0 + (q & ~1)) * 16 + 2 * li + (q & 1) : (ct0 + q) * 16 + li;
                    const T v = yk[lc];
                    u1[q] = A2<T>::mfma(a.x, v.x, u1[q]);
                    u2[q] = A2<T>::mfma(a.y, v.y, u2[q]);
                    u3[q] = A2<T>::mfma(as, v.x + v.y, u3[q]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int xr = slab * 16 + A2<T>::drow(lk, r);
            if (xr >= p2) continue;
            T* xrow = g.xout + (int64_t)(xr + g.orow_shift) * ldw;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int col = cg + (PAIR ? (ct0 + (q & ~1)) * 16 + 2 * li + (q & 1) : (ct0 + q) * 16 + li);
                T v;
                v.x = u1[q][r] - u2[q][r]; v.y = u3[q][r] - u1[q][r] - u2[q][r];
                if (col < ldw) xrow[col] = v;
            }
        }
    }
}

template <typename T, bool GATHER, int NSMAX>
__global__ __launch_bounds__(256, NSMAX == 4 ? 2 : (GATHER ? FEASTHIP_A2_OCCG : FEASTHIP_A2_OCCN)) void k_apply2(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles,
                                                const int* __restrict__ struct_idx, const T* __restrict__ fact, int64_t fs,
                                                T* W, T* Y, int64_t ws, int ldw, T* crow, int64_t crs, T* skbuf, int64_t sks,
                                                XcdParts xp, const int* __restrict__ xperm, int lanes, int ncg, const int* __restrict__ cmap, const T* __restrict__ crow_child, int64_t crs_child) {
    // Workgroup b runs on XCD b % 8.  The XCDs form `lanes` groups of 8 / lanes: group l takes the nodes l, l + lanes, ... whole, and its
    // XCDs the 8 / lanes parts of every launch list (dev_common.hpp: XcdParts, `lanes` adjacent parts merged), node-major, the column
    // groups of a tile next to each other (the second one finds the factor slab in the XCD's L2).  lanes = gcd(nodes, 8): with 8 nodes
    // every XCD sweeps whole nodes (no block-vector row is shared between L2s), with one node every XCD takes an eighth of each list.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, P = 8 / lanes, lane = xcd / P, q = xcd - lane * P;
    const int lo = xp.off[q * lanes], cnt = xp.off[(q + 1) * lanes] - lo, per = xp.cmax * ncg;
    const int ki = slot / per, rem = slot - ki * per, j = rem / ncg, cg = (rem - j * ncg) * LDS_COLS, k = lane + ki * lanes;
    if (j >= cnt) return;
    Tile t = tiles[xp.perm >= 0 ? xperm[xp.perm + lo + j] : lo + j];
    FrontD f = fr[t.f];
    ApplyGeomT<T> g;
    int trows = decode_tile<T>(g, f, t, k, struct_idx, fact, fs, W, Y, ws, ldw, crow, crs, skbuf, sks, cmap, crow_child, crs_child);
    __shared__ T Ys[GATHER ? 16 * NSMAX * LDS_COLS : 1];     // the y' tile of a fused B1 + B3 tile (kind 6)
    if (trows <= 16) apply2_body<T, GATHER, 1>(g, cg, ldw, Ys);
    else if (NSMAX == 2 || trows <= 32) apply2_body<T, GATHER, 2>(g, cg, ldw, Ys);
    else if (NSMAX == 4) apply2_body<T, GATHER, 4>(g, cg, ldw, Ys);
}

template <typename T>
static void launch_apply2(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const T* fact, int64_t fs,
                          T* W, T* Y, int64_t ws, int ldw, T* crow, int64_t crs, int K, bool gather, T* skbuf, int64_t sks,
                          const T* crow_child, int64_t crs_child, int max_slabs, const XcdParts* xpp) {
    const int ncg = (ldw + LDS_COLS - 1) / LDS_COLS;
    XcdParts xp;
    if (xpp && xpp->off[8] == ntiles) xp = *xpp;
    else { for (int x = 0; x <= 8; ++x) xp.off[x] = (int)((int64_t)ntiles * x / 8); xp.perm = -1; }     // no plan: equal counts
    int lanes = 8;
    while (K % lanes) lanes >>= 1;          // gcd(K, 8)
    if (lanes == 8) xp.perm = -1;           // whole nodes per XCD: the list's own order
    xp.cmax = 0;
    for (int q = 0; q < 8 / lanes; ++q) xp.cmax = std::max(xp.cmax, xp.off[(q + 1) * lanes] - xp.off[q * lanes]);
    dim3 g(8 * xp.cmax * (K / lanes) * ncg);
#define FEASTHIP_A2(G, N) hipLaunchKernelGGL((k_apply2<T, G, N>), g, dim3(256), 0, st, S.fr, tiles, S.struct_idx, fact, fs, W, Y, ws, ldw, crow, crs, skbuf, sks, xp, S.xperm, lanes, ncg, S.cmap, crow_child, crs_child)
    if (max_slabs <= 2) { if (gather) FEASTHIP_A2(true, 2); else FEASTHIP_A2(false, 2); }
    else { if (gather) FEASTHIP_A2(true, 4); else FEASTHIP_A2(false, 4); }
#undef FEASTHIP_A2
}

// ---------------------------------------------------------------------------
// Mixed precision (feast!'s mixed_prec, src/feast.jl:19-25): the complex64 instance of k_apply2 (v_mfma_f32_16x16x4_f32, paired 16-byte lanes)
void launch_apply(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const cf* fact, int64_t fs,
                  cf* W, cf* Y, int64_t ws, int ldw, cf* crow, int64_t crs, int K, bool gather, cf* skbuf, int64_t sks,
                  const cf* crow_child, int64_t crs_child, int max_slabs, const XcdParts* xp) {
    if (ntiles <= 0) return;
    launch_apply2<cf>(st, S, tiles, ntiles, fact, fs, W, Y, ws, ldw, crow, crs, K, gather, skbuf, sks, crow_child, crs_child, max_slabs, xp);
}

__global__ void k_to_f32(int64_t total, const cz* __restrict__ in, cf* out) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    cz v = in[e];
    cf r; r.x = (float)v.x; r.y = (float)v.y;
    out[e] = r;
}

// The same for the fronts of one tree level (list of front indices): a front's L panel ((p + u) x p) and its U12 block (p x u).  blockIdx.y
// walks the front's entries in strides of gridDim.y * 256, blockIdx.z the units.
__global__ __launch_bounds__(256) void k_to_f32_fronts(const FrontD* __restrict__ fr, const int* __restrict__ list, const cz* __restrict__ in, cf* out, int64_t fs) {
    const FrontD f = fr[list[blockIdx.x]];
    const int64_t nl = (int64_t)(f.p + f.u) * f.p, nu = (int64_t)f.p * f.u;
    const cz* src = in + (int64_t)blockIdx.z * fs;
    cf* dst = out + (int64_t)blockIdx.z * fs;
    for (int64_t e = blockIdx.y * 256 + threadIdx.x; e < nl + nu; e += (int64_t)gridDim.y * 256) {
        const int64_t a = e < nl ? f.lp + e : f.f12 + (e - nl);
        const cz v = src[a];
        cf r; r.x = (float)v.x; r.y = (float)v.y;
        dst[a] = r;
    }
}

void launch_to_f32_fronts(hipStream_t st, const DevSym& S, const int* list, int nlist, int chunks, const cz* in, cf* out, int64_t fs, int K) {
    if (nlist <= 0) return;
    hipLaunchKernelGGL(k_to_f32_fronts, dim3(nlist, std::max(1, std::min(chunks, 65535)), K), dim3(256), 0, st, S.fr, list, in, out, fs);
}

void launch_to_f32(hipStream_t st, int64_t total, const cz* in, cf* out) {
    hipLaunchKernelGGL(k_to_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, in, out);
}

// Y[c0 + r, :] += sum_part scratch[part][r, :]   (parts in increasing order: deterministic)
// A thread owns one (row, column) element: the row's value and all its partials (at most SPLITK_MAX, the planner's cap) are
// loaded together -- with few nodes per GPU this kernel is a handful of workgroups and its time is its longest dependent chain.
constexpr int SPLITK_MAX = 8;
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int nsplit, T* Y, int64_t ws,
                                                       int ldw, const T* __restrict__ skbuf, int64_t sks) {
    Tile t = tiles[blockIdx.x];
    int k = blockIdx.y;
    FrontD f = fr[t.f];
    const int Kp = (((f.u + nsplit - 1) / nsplit) + 31) & ~31;
    const int nparts = (f.u + Kp - 1) / Kp;
    const int r0 = t.a * 64, nr = min(f.p, r0 + 64) - r0;
    const int total = nr * ldw;
    T* y0 = Y + (int64_t)k * ws + (int64_t)(f.c0 + r0) * ldw;                       // the tile's rows are contiguous
    const T* p0 = skbuf + (int64_t)k * sks + (f.sk_off + (int64_t)r0) * ldw;        // part q of row r0 + i: p0 + (q p + i) ldw
    const int64_t pstride = (int64_t)f.p * ldw;
    for (int e0 = threadIdx.x; e0 < total; e0 += 4 * 256) {
        T v[4], a[4][SPLITK_MAX];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + u * 256, total - 1);
            v[u] = y0[e];
#pragma unroll
            for (int q = 0; q < SPLITK_MAX; ++q) a[u][q] = p0[(int64_t)min(q, nparts - 1) * pstride + e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < SPLITK_MAX; ++q)
                if (q < nparts) { v[u].x += a[u][q].x; v[u].y += a[u][q].y; }
            if (e0 + u * 256 < total) y0[e0 + u * 256] = v[u];
        }
    }
}

// Forward counterpart: C_I[r, :] = sum_children C_child[...] + sum_part scratch[part][r, :]  (the slabs hold -P21[:, part] W[part];
// children in slot order, parts in increasing order: the same additions in the same order as the unsplit tile makes them).
// Tile.a = 64-row slice of the front's contribution rows.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce_fwd(const FrontD* __restrict__ fr, const Tile* __restrict__ tiles, int nsplit, T* crow, int64_t crs,
                                                           int ldw, const T* __restrict__ skbuf, int64_t sks, const int* __restrict__ cmap,
                                                           const T* __restrict__ crow_child, int64_t crs_child) {
    Tile t = tiles[blockIdx.x];
    const int k = blockIdx.y;
    FrontD f = fr[t.f];
    const int u = f.u, p = f.p;
    const int Kp = (((p + nsplit - 1) / nsplit) + 31) & ~31;
    const int nparts = (p + Kp - 1) / Kp;
    const int r0 = t.a * 64, nr = min(u, r0 + 64) - r0;
    T* out = crow + (int64_t)k * crs + (f.crow + (int64_t)r0) * ldw;
    const T* p0 = skbuf + (int64_t)k * sks + (f.skf_off + (int64_t)r0) * ldw;
    const int64_t pstride = (int64_t)u * ldw;
    const T* cbase = crow_child + (int64_t)k * crs_child;
    const int nch = f.cmap >= 0 ? f.nchild : 0;
    for (int e = threadIdx.x; e < nr * ldw; e += 256) {
        const int i = e / ldw, j = e - i * ldw;
        T v; v.x = 0; v.y = 0;
        for (int c = 0; c < nch; ++c) {
            const int src = cmap[f.cmap + (int64_t)(r0 + i) * nch + c];
            if (src >= 0) { const T w = cbase[(int64_t)src * ldw + j]; v.x += w.x; v.y += w.y; }
        }
        T a[SPLITK_MAX];
#pragma unroll
        for (int q = 0; q < SPLITK_MAX; ++q) a[q] = p0[(int64_t)min(q, nparts - 1) * pstride + e];
        // (out = children - acc_0 - acc_1 ...: the slabs hold -acc_q)
#pragma unroll
        for (int q = 0; q < SPLITK_MAX; ++q)
            if (q < nparts) { v.x += a[q].x; v.y += a[q].y; }
        out[e] = v;
    }
}

void launch_splitk_reduce_fwd(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cz* crow, int64_t crs, int ldw,
                              const cz* skbuf, int64_t sks, const cz* crow_child, int64_t crs_child, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_splitk_reduce_fwd<cz>, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, nsplit, crow, crs, ldw, skbuf, sks, S.cmap, crow_child, crs_child);
}
void launch_splitk_reduce_fwd(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cf* crow, int64_t crs, int ldw,
                              const cf* skbuf, int64_t sks, const cf* crow_child, int64_t crs_child, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_splitk_reduce_fwd<cf>, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, nsplit, crow, crs, ldw, skbuf, sks, S.cmap, crow_child, crs_child);
}

void launch_splitk_reduce(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cz* Y, int64_t ws, int ldw,
                          const cz* skbuf, int64_t sks, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_splitk_reduce<cz>, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, nsplit, Y, ws, ldw, skbuf, sks);
}
void launch_splitk_reduce(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cf* Y, int64_t ws, int ldw,
                          const cf* skbuf, int64_t sks, int K) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_splitk_reduce<cf>, dim3(ntiles, K), dim3(256), 0, st, S.fr, tiles, nsplit, Y, ws, ldw, skbuf, sks);
}

// Q0[old,:] = sum_k T_k[i,:] .* e0[k,:]  (and Q1 with e1), T_k = mode ? X[old,:] - W_k[i,:] : W_k[i,:].
// Fixed summation order over k: run-to-run reproducible (src/feast.jl:68-70, src/nlfeast.jl:56-59).
// Conjugate fold (lds = 2 ldw): unit k carries two contour nodes; columns [ldw, 2 ldw) of W_k hold T(z_k)^-1 conj(R),
// whose conjugate is the solution of the node conj(z_k); its coefficients are columns [ldw, 2 ldw) of e0 / e1.
// add: Q0 (Q1) += instead of = (a second launch over the adjoint instances of an adjoint-folded step)
template <typename T>
__global__ void k_accumulate(const int* __restrict__ iperm, int row0, int nrows, const cz* __restrict__ X, const T* __restrict__ W,
                             int64_t ws, int ldw, int lds, int K, const cz* __restrict__ e0, const cz* __restrict__ e1, int mode,
                             cz* Q0, cz* Q1, int add) {
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (r >= nrows) return;
    int64_t old = row0 + r;            // rows [row0, row0 + nrows) of Q in ORIGINAL order (a contiguous chunk for the all-reduce)
    const int i = iperm[old];
    const bool fold = lds > ldw;
    for (int j = lane; j < ldw; j += 64) {
        cz x = mode ? X[old * ldw + j] : cz_make(0, 0);
        cz q0 = add ? Q0[old * ldw + j] : cz_make(0, 0), q1 = (add && Q1) ? Q1[old * ldw + j] : cz_make(0, 0);
        for (int k = 0; k < K; ++k) {
            T wt = W[(int64_t)k * ws + (int64_t)i * lds + j];
            cz w = cz_make(wt.x, wt.y);                 // accumulation is always fp64 (src/feast.jl:68-70)
            cz tv = mode ? cz_sub(x, w) : w;
            cz_madd(q0, tv, e0[(int64_t)k * lds + j]);
            if (Q1) cz_madd(q1, tv, e1[(int64_t)k * lds + j]);
            if (fold) {
                T wc = W[(int64_t)k * ws + (int64_t)i * lds + ldw + j];
                cz w2 = cz_make(wc.x, -(double)wc.y);
                cz tv2 = mode ? cz_sub(x, w2) : w2;
                cz_madd(q0, tv2, e0[(int64_t)k * lds + ldw + j]);
                if (Q1) cz_madd(q1, tv2, e1[(int64_t)k * lds + ldw + j]);
            }
        }
        Q0[old * ldw + j] = q0;
        if (Q1) Q1[old * ldw + j] = q1;
    }
}

void launch_accumulate(hipStream_t st, const DevSym& S, const cz* X, const cz* W, int64_t ws, int ldw, int lds, int K,
                       const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1, int row0, int nrows, int add) {
    if (nrows < 0) nrows = S.n - row0;
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_accumulate<cz>, dim3((nrows + 3) / 4), dim3(256), 0, st, S.iperm, row0, nrows, X, W, ws, ldw, lds, K, e0, e1, mode, Q0, Q1, add);
}

// Adjoint sweeps on factors with row interchanges: T = P^T L U gives T^H = U^H L^H P, so the interchanges of a front act on its part
// of the SOLUTION, x_J = P_J^T w_J.  One workgroup per (front, unit) writes the finished pivot rows of W into X at their pre-pivot
// positions (row c0 + rowperm[i] <- row c0 + i); the children's gathers and the accumulation then read X.
template <typename T>
__global__ __launch_bounds__(256) void k_adj_unpermute(const FrontD* __restrict__ fr, const int* __restrict__ list, const int* __restrict__ rowperm, int n,
                                                       const T* __restrict__ W, T* X, int64_t ws, int ld) {
    const FrontD f = fr[list[blockIdx.x]];
    const int k = blockIdx.y;
    const int* rp = rowperm + (int64_t)k * n + f.c0;
    const T* Wk = W + (int64_t)k * ws + (int64_t)f.c0 * ld;
    T* Xk = X + (int64_t)k * ws + (int64_t)f.c0 * ld;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < f.p; i += 4) {
        const int o = rp[i];
        for (int j = lane; j < ld; j += 64) Xk[(int64_t)o * ld + j] = Wk[(int64_t)i * ld + j];
    }
}

void launch_adj_unpermute(hipStream_t st, const DevSym& S, const int* list, int nlist, const int* rowperm, const cz* W, cz* X, int64_t ws, int ld, int K) {
    if (nlist <= 0) return;
    hipLaunchKernelGGL(k_adj_unpermute<cz>, dim3(nlist, K), dim3(256), 0, st, S.fr, list, rowperm, S.n, W, X, ws, ld);
}
void launch_accumulate(hipStream_t st, const DevSym& S, const cz* X, const cf* W, int64_t ws, int ldw, int lds, int K,
                       const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1, int row0, int nrows) {
    if (nrows < 0) nrows = S.n - row0;
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_accumulate<cf>, dim3((nrows + 3) / 4), dim3(256), 0, st, S.iperm, row0, nrows, X, W, ws, ldw, lds, K, e0, e1, mode, Q0, Q1, 0);
}

}  // namespace feasthip
