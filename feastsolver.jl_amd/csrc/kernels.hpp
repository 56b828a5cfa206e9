// Host-callable launchers of the gfx950 kernels (definitions in kernels_*.hip).
#pragma once
#include "dev_common.hpp"

namespace feasthip {

constexpr int FACT_NB = 32;     // panel width of the blocked front factorisation
constexpr int SOLVE_NB = 128;   // base diagonal (super)block of the inverse-based block substitution; fronts with many pivot columns
                                // couple these further into 256- / 512-column super blocks (FrontD::nb, launch_couple_level)
constexpr int INV_NB = 64;      // size of the directly inverted triangular blocks (two are coupled into a super block)
constexpr int TRSM_COLS = 64;       // columns per workgroup of the U-row triangular solves
constexpr int SCALE_ROWTILES = 4;  // 64-row tiles per workgroup of the panel-scaling pass (they share one staged inverse block)
constexpr int LDS_COLS = 64;    // block-vector columns processed per workgroup pass

struct DevSym {          // device copies of the symbolic structure
    const FrontD* fr;
    const int* struct_idx;
    const int* rel;
    const int* child_idx;
    const int* perm;     // new -> old
    const int* iperm;    // old -> new
    const int* c0_of_row;  // first column of the supernode owning (new) row i
    const int* depth_of_row;  // tree depth of that supernode (the right-hand-side gather of ONE tree level: early forward sweep)
    const int* cmap;       // per front with children: [contribution row][child slot] -> row in the children's arena, or -1
    const int* ebp;        // extend-add break points, see FrontD::ebp
    const int* pmap;       // per front with children: [pivot row][child slot] -> row in the children's arena, or -1 (k_fwd_assemble)
    const int* xperm;      // dealt tile orders of the block-substitution launches (dev_common.hpp: XcdParts::perm)
    // fused right-hand-side gather (nullptr: off): rows of the block vectors that receive contributions of children are not
    // written by k_gather_rhs but by k_fwd_assemble (W = R + contributions instead of W += contributions)
    const int* asmflag;    // [n] by pre-pivot (new) row: 1 = some child contributes to this pivot row
    int n;
};

// ---- numeric factorisation ------------------------------------------------
void launch_scatter(hipStream_t st, cz* fact, int64_t fs, const int64_t* dest, const cz* val, int64_t nnz,
                    const cz* coef, int coef_stride, int term, int K);
// caller's dense matrix (raw host element type / storage order, already on the device) -> complex128 column-major copy + properties
// props[3]: bits of max |a_ij|, not-real flag, not-symmetric flag (zeroed by the caller)
void launch_dense_import(hipStream_t st, const void* raw, int type, int row_major, int n, cz* out, unsigned long long* props);
void launch_dense_assemble(hipStream_t st, cz* fact, int64_t fs, int n, const cz* const* mats, int nterms,
                           const cz* coef, int K);
void launch_extend_add(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs,
                       cz* cb_parent, int64_t cbs_parent, const cz* cb_child, int64_t cbs_child, int K, int phase, int sym = 0);
void launch_panel(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, cz* fact, int64_t fs,
                  int* piv, const double* tiny, int* nperturb, int K, bool tall = false, int max_rows = 1 << 30);
// Pivot-light elimination for symmetric pencils (no row interchanges), per GROUP of up to 128 pivot columns: k_diag_grp factorises the group's diagonal block in one
// workgroup per (front, node) and leaves W_c = inv(U_cc) of its 32-column blocks in wscr[(k * wstride + FrontD::wslot * 4 + c) * 1024];
// k_lscale_grp (Tile {front, 64-row slice of the rows below the block}) solves X U11 = A for the rows below on the matrix cores
// and writes L and U = D L^T.  See kernels_factor.hip.
void launch_diag_grp(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, cz* fact, int64_t fs, int* piv,
                     const double* tiny, int* nperturb, unsigned long long* growth, cz* wscr, int64_t wstride, int K);
void launch_lscale_grp(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                       unsigned long long* growth, const cz* wscr, int64_t wstride, int K);
// Column selection of one launch of an elimination step (look-ahead schedule): left columns [llo, lhi), pivot-block
// columns right of the panel with offset in [clo, chi), U12 columns iff u12.  The default selects everything.
struct ColSel { int llo = 0, lhi = 1 << 30, clo = 0, chi = 1 << 30, u12 = 1; };
inline ColSel col_right(int clo, int chi) { ColSel c; c.llo = 0; c.lhi = 0; c.clo = clo; c.chi = chi; c.u12 = 0; return c; }
// elimination without interchanges of a NOT symmetric panel (complex Hermitian pencils): diagonal 32 x 32 block, then the rows below it
// (tile list: the step's 256-column slices, read as 256-row slices); growth: [K] max |l|, [K] max |pivot|, [K] max |u|
void launch_diag32_gen(hipStream_t st, const DevSym& S, const int* list, int nlist, int k0, cz* fact, int64_t fs, int* piv,
                       const double* tiny, int* nperturb, unsigned long long* growth, cz* wscr, int64_t wstride, int K);
void launch_lscale32_gen(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                         unsigned long long* growth, const cz* wscr, int64_t wstride, int K);
void launch_swap(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                 const int* piv, int K, ColSel cs = ColSel(), int bw = FACT_NB);
void launch_trsm_u(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs, int K, ColSel cs = ColSel(),
                   unsigned long long* umax = nullptr);      // umax (per node, optional): max |u_ij| written, for the growth check of the eliminations without interchanges
// U rows of a group of panels (gw <= 128 rows) for 32-column tiles of the columns right of the group
void launch_group_utrsm(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, int gw, cz* fact, int64_t fs, int K, ColSel cs,
                        unsigned long long* umax = nullptr);
void launch_gemm_nn(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int k0, cz* fact, int64_t fs,
                    cz* cb, int64_t cbs, int K, ColSel cs = ColSel(), int bw = FACT_NB, int mrows = 0,
                    const cz* cb_child = nullptr, int64_t cbs_child = 0,     // cb_child: fuse the children's extend-add into the contribution block (kind 2)
                    int mirror = 0);     // symmetric elimination.  bit 0: symmetric contribution blocks -- the list holds the kind-2 tiles with b >= c only, nothing is mirrored (round 5), and the
                                         //        children's entries pulled in the epilogue are read from their lower triangles;
                                         // bit 1: kind-1 tiles and kind-0 tiles strictly above the diagonal are skipped (U = D L^T is written by k_lscale_grp);
                                         // bit 2: raised wave priority (products on the dependency chain of the look-ahead schedule)
// solve-ready form: explicit inverses of the SOLVE_NB diagonal blocks, off-diagonal blocks pre-scaled
void launch_block_inverse(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K);
// pivot-light path: the 32 x 32 diagonal blocks already hold their inverses (k_diag_grp); couple pairs into 64 x 64 block inverses
// (same tiles as launch_block_inverse, which this replaces)
void launch_couple32(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K);
void launch_couple_inverse(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K);
void launch_scale_panels(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* fact, int64_t fs, int K, bool pivot_rows_only = false);
// Coupling of two solve-ready super blocks of `half` columns (a = [k0, k0+half), b = [k0+half, ...)) into one of 2*half, k0 = pair * 2*half:
//   launch_couple_outer (Tile {front, 3 | pair << 8, m tile, n tile}: rows below the pair, P_a -= P_b Pc;
//                              {front, 4 | pair << 8, m tile, n tile}: rows above the pair, Q_b -= Q_a Qd)       -- k_gemm_nn
//   launch_couple_level (Tile {front, pair, side (0 lower / 1 upper), 16-column strip}): Pc <- -inv(L_bb) Pc, Qd <- -inv(U_aa) Qd, in place
// (inv([L_a 0; C L_b]) = [I_a 0; -I_b C I_a, I_b], and the mirrored identity for the upper factor); run outer first, it reads Pc / Qd.
void launch_couple_outer(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int half, cz* fact, int64_t fs, int K, bool pivot_rows_only = false);
void launch_couple_level(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int half, cz* fact, int64_t fs, int K);
void launch_rowperm(hipStream_t st, const DevSym& S, const int* list, int nlist, const int* piv, int* rowperm,
                    int* rowinv, int K);

// ---- block triangular solves ------------------------------------------------
// lds = leading dimension of W: ldw, or 2 ldw for the conjugate fold (columns ldw.. receive conj(src))
// only_depth >= 0: the rows of the fronts of that tree depth only
void launch_gather_rhs(hipStream_t st, const DevSym& S, const cz* src, int ldw, int lds, cz* W, int64_t ws,
                       const int* rowperm, int K, int only_depth = -1);
// (src, ldsrc: the right-hand side the sweep started from, for the rows k_gather_rhs left to the first contribution)
void launch_fwd_assemble(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cz* W, int64_t ws, int ldw,
                         const cz* crow_child, int64_t crs_child, const int* rowinv, int K, const cz* src = nullptr, int ldsrc = 0);
// Block substitution on the solve-ready factors: every tile is out (-)= A * V on the f64 matrix cores.
// Tile.a = kind, Tile.b = row tile (or diagonal block), Tile.c = step block index:
//   0 F1  forward, rows of the pivot block below diagonal block c:   W_r -= P_rc W_c
//   1 F2a forward, contribution rows:                                  C_I  = sum_children C_child[I] - P21 W_J   (K = p; the
//                                                                      children's rows are pulled in the epilogue through DevSym::cmap)
//   2 F2b forward, diagonal super block b, row tile c:                 Y_b  = inv(L_bb) W_b
//   3 B1  backward, pivot rows:                                        Y_J -= U12 W[struct]    (K = u, gathered; final x lives in W)
//   4 B2  backward, rows above diagonal block c:                       Y_r -= PU_rc Y_c
//   5 B3  backward, diagonal super block b, row tile c:                W_b  = inv(U_bb) Y_b
// Adjoint sweeps, T(z)^H x = r over the SAME solve-ready factors (round 4; T(conj z) = T(z)^H for Hermitian pencils, and the left systems
// (A - z B)^H of dual_gen_feast!, src/feast.jl:185-197, 241).  U^-1 = F E_0 .. E_{S-1} and L^-1 = F' E'_{S-2} .. E'_0 in terms of the block
// operations above; the adjoint sweeps apply the conjugate transposes of the same blocks in the reverse order, every operand read as
// conj(slab)^T (ApplyGeom::trans):
//   8  AF-D forward, diagonal super block b, row tile c:               Y_b  = inv(U_bb)^H W_b            (first step of a front)
//   9  AF-E forward step c >= 1, rows of block c:                      Y_c -= (PU_[0,c) c)^H Y_[0,c)     (K = c NB, growing)
//   10 AF-C forward, contribution rows:                                C_I  = sum_children C_child[I] - U12^H Y_J   (K = p)
//   11 AB-D diagonal super block b, row tile c:                        W_b  = inv(L_bb)^H Y_b            (same launch as kind 10)
//   12 AB-G backward, pivot rows:                                      W_J -= P21^H W[struct]            (K = u, gathered; split like kind 3)
//   13 AB-E backward step c, rows of block c:                          W_c -= (P_(c,S) c)^H W_(c,S)      (descending c; final x in W)
// max_slabs: an upper bound of the tile heights of this launch in 16-row slabs (4, 2 or 1).  Launches whose tiles all have
// at most 32 rows run a k_apply2 instance that needs fewer registers and keeps three workgroups per CU instead of two.
void launch_apply(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const cz* fact, int64_t fs,
                  cz* W, cz* Y, int64_t ws, int ldw, cz* crow, int64_t crs, int K, bool gather = false,
                  cz* skbuf = nullptr, int64_t sks = 0, const cz* crow_child = nullptr, int64_t crs_child = 0, int max_slabs = 4,
                  const XcdParts* xp = nullptr);
// Tile kind 6 = the B1 tile of a front whose pivot block fits one row tile; k_apply2 multiplies it by the front's upper inverse in
// the same workgroup (so the front has no separate B3 tile).
// complex64 twins for the mixed-precision solve path (same tiles, same plans)
void launch_apply(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, const cf* fact, int64_t fs,
                  cf* W, cf* Y, int64_t ws, int ldw, cf* crow, int64_t crs, int K, bool gather = false,
                  cf* skbuf = nullptr, int64_t sks = 0, const cf* crow_child = nullptr, int64_t crs_child = 0, int max_slabs = 4,
                  const XcdParts* xp = nullptr);
void launch_gather_rhs(hipStream_t st, const DevSym& S, const cz* src, int ldw, int lds, cf* W, int64_t ws,
                       const int* rowperm, int K, int only_depth = -1);
void launch_fwd_assemble(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, cf* W, int64_t ws, int ldw,
                         const cf* crow_child, int64_t crs_child, const int* rowinv, int K, const cz* src = nullptr, int ldsrc = 0);
void launch_splitk_reduce(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cf* Y, int64_t ws, int ldw,
                          const cf* skbuf, int64_t sks, int K);
void launch_accumulate(hipStream_t st, const DevSym& S, const cz* X, const cf* W, int64_t ws, int ldw, int lds, int K,
                       const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1, int row0 = 0, int nrows = -1);
void launch_to_f32(hipStream_t st, int64_t total, const cz* in, cf* out);
// the fronts of one tree level (front indices `list`), `chunks` workgroups of 256 entries in flight per front and unit
void launch_to_f32_fronts(hipStream_t st, const DevSym& S, const int* list, int nlist, int chunks, const cz* in, cf* out, int64_t fs, int K);
// Kind-3 tiles may be split along K (Tile.a bits 16.. = number of parts, Tile.c = part): each part writes
// -U12[:, part] * x[part] into its own slab of `skbuf`; launch_splitk_reduce then adds the slabs to Y in a fixed order.
void launch_splitk_reduce(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cz* Y, int64_t ws, int ldw,
                          const cz* skbuf, int64_t sks, int K);
// Kind-1 tiles may be split the same way (K = p; few nodes per GPU leave too few, too long tiles at the top of the tree): part
// Tile.c writes -P21[:, part] W[part] into its slab; launch_splitk_reduce_fwd (Tile {front, 64-row slice of the contribution rows})
// adds the children's rows and the slabs in a fixed order.
void launch_splitk_reduce_fwd(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cz* crow, int64_t crs, int ldw,
                              const cz* skbuf, int64_t sks, const cz* crow_child, int64_t crs_child, int K);
void launch_splitk_reduce_fwd(hipStream_t st, const DevSym& S, const Tile* tiles, int ntiles, int nsplit, cf* crow, int64_t crs, int ldw,
                              const cf* skbuf, int64_t sks, const cf* crow_child, int64_t crs_child, int K);
// (gather = true for launches made only of kind-3 tiles: V rows are gathered through struct_idx)
// Q0 (and Q1) = sum_k T_k .* e0[k,:] (e1[k,:]),  T_k = mode ? X - W_k : W_k, rows scattered to original order;
// lds = leading dimension of W and of the coefficient tables (2 ldw: conjugate fold, see k_accumulate)
// row0 / nrows: the rows of Q (original order) this launch produces; nrows < 0 = all from row0
// add: += instead of =
void launch_accumulate(hipStream_t st, const DevSym& S, const cz* X, const cz* W, int64_t ws, int ldw, int lds, int K,
                       const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1, int row0 = 0, int nrows = -1, int add = 0);
// adjoint sweeps on factors with row interchanges: X[c0 + rowperm[i]] = W[c0 + i] for the pivot rows of the listed fronts (see kernels_solve.hip)
void launch_adj_unpermute(hipStream_t st, const DevSym& S, const int* list, int nlist, const int* rowperm, const cz* W, cz* X, int64_t ws, int ld, int K);

// ---- Rayleigh-Ritz block ------------------------------------------------------
// out[q[e]] = in[e]: the values of a term in CSR order from its CSC values (Engine::analyze)
void launch_permute_vals(hipStream_t st, int64_t nnz, const int* q, const cz* in, cz* out);
void launch_spmm_csr(hipStream_t st, int n, const int64_t* rowptr, const int* col, const cz* val, const cz* V, int ldw,
                     cz* out, cz alpha, int accumulate);
// out(row-major n x ldw) (+)= alpha * A(col-major n x n) * V
// part: scratch of dense_mv_parts(n, ldw) * n * ldw elements for the split-K partial sums (nullptr: no split)
int dense_mv_parts(int n, int ldw);
void launch_dense_mv(hipStream_t st, int n, const cz* A, const cz* V, int ldw, cz* out, cz alpha, int accumulate, cz* part = nullptr);
// G(m x m, column-major, ld m) = V^H W  (row-major n x ldw blocks); scratch >= gram_scratch_elems()
size_t gram_scratch_elems(int n, int m);
// rowscale (n entries, optional): G = V^H diag(rowscale) W -- the projection with a diagonal B without forming B W
void launch_gram(hipStream_t st, int n, int m, int ldw, const cz* V, const cz* W, cz* scratch, cz* G, const cz* rowscale = nullptr);
// out = V * M  (M m x m column-major on device)
void launch_ts_gemm(hipStream_t st, int n, int m, int ldw, const cz* V, const cz* M, cz* out);
// nrm2[j] = sum_i |V[i,j]|^2 ; scratch >= colnorm_scratch_elems()
size_t colnorm_scratch_elems(int n, int ldw);
void launch_colnorm2(hipStream_t st, int n, int ldw, const cz* V, double* scratch, double* nrm2);
// kernels_small.hip: the m0 x m0 Hermitian(-definite) reduced eigenproblem in one workgroup (m <= 64), and the residual kernel that
// normalises the Ritz vectors on the way (update_R!, src/utils.jl:111-116)
void launch_rr_small_herm(hipStream_t st, int m, const cz* Ga, const cz* Gb, cz* Xq, cz* lam_out, int ldw, double* status, cz* tscratch, double cond_max);
void launch_residual_combine_scaled(hipStream_t st, int n, int ldw, const cz* AX, const cz* BX, const cz* lam, cz* X, cz* R, const cz* rowscale,
                                    const double* nrm2);
void launch_scale_cols(hipStream_t st, int n, int ldw, cz* V, const cz* s);
// out = alpha*V (+ out)
void launch_axpby(hipStream_t st, int n, int ldw, cz* out, cz alpha, const cz* V, int accumulate);
// R = AX - BX .* lam[j]   (rowscale: BX = diag(rowscale) * the block passed as BX)
void launch_residual_combine(hipStream_t st, int n, int ldw, const cz* AX, const cz* BX, const cz* lam, cz* R, const cz* rowscale = nullptr);
// R = sum_t MX[t] .* lam[j]^t   (polynomial NEP residual; MX blocks contiguous with stride blk)
void launch_poly_combine(hipStream_t st, int n, int ldw, const cz* MX, int64_t blk, int nterms, const cz* lam, cz* R);
// block-Hankel matrix of contour moments: H (mm n x ldh, row-major), block (i, j) = Q_{i+j+shift}
void launch_hankel(hipStream_t st, int n, int m0, int ldw, int mm, int ldh, int shift, const cz* Qm, cz* H);
// out[:, j] = in[:, idx[j]] (j < mo), padding columns zeroed; row-major blocks with leading dimensions ldi / ldo
void launch_gather_cols(hipStream_t st, int n, int ldi, const cz* in, int ldo, int mo, const int* idx, cz* out);
// R = sum_t MX[t] .* tab[t, j]   (split-form NEP residual; tab is nterms x ldw on the device)
void launch_coef_combine(hipStream_t st, int n, int ldw, const cz* MX, int64_t blk, int nterms, const cz* tab, cz* R);
// column-major (ld n) <-> row-major (ld ldw) conversion of n x m0 blocks, zero padding columns m0..ldw
void launch_cm_to_rm(hipStream_t st, int n, int m0, int ldw, const cz* cm, int64_t ldcm, cz* rm);
void launch_rm_to_cm(hipStream_t st, int n, int m0, int ldw, const cz* rm, cz* cm, int64_t ldcm);

}  // namespace feasthip
