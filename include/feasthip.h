/* feasthip.h -- C ABI of libfeasthip.so: the MI355X (gfx950) FEAST inner loop.
 *
 * Drop-in boundary for the hot path of spacedome/FEASTSolver.jl (reference @
 * 2024-12-18).  File:line citations below are relative to the reference tree.
 * All entry points are extern "C", take plain pointers and sizes, never throw
 * and never abort; they return an int status:
 *     0   success
 *    <0   FEASTHIP_E* (argument / dimension / HIP failure); text via
 *         feasthip_last_error().  Dimension errors correspond to the
 *         reference's error("Incorrect dimensions of A, must be square") /
 *         error("Incorrect dimensions of X, must match A") (src/feast.jl:13-17).
 *    >0   number of pivots that had to be statically perturbed during the
 *         numeric factorisation (the LAPACK "info > 0 => singular" convention,
 *         maps to Julia's SingularException).
 * The caller owns every host array; pointers need to stay valid only for the
 * duration of the call (the library copies to the device and keeps no host
 * pointer) -- the same contract as the reference's one ccall,
 * src/lapack.jl:79-89.  Complex numbers are interleaved (re, im) doubles, i.e.
 * Julia ComplexF64 / numpy complex128.  Dense matrices and block vectors are
 * column-major with leading dimension n (Julia Matrix layout).
 */
#ifndef FEASTHIP_H
#define FEASTHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct feasthip_handle_s* feasthip_handle_t;
typedef struct feasthip_fact_s* feasthip_fact_t;

enum {
    FEASTHIP_OK = 0,
    FEASTHIP_EARG = -1,      /* bad argument */
    FEASTHIP_EDIM_A = -2,    /* "Incorrect dimensions of A, must be square"  src/feast.jl:13-14 */
    FEASTHIP_EDIM_X = -3,    /* "Incorrect dimensions of X, must match A"    src/feast.jl:15-16 */
    FEASTHIP_ESTATE = -4,    /* call order (e.g. solve before factorize) */
    FEASTHIP_EHIP = -5,      /* HIP runtime failure */
    FEASTHIP_ENOGPU = -6,    /* no gfx950 device: the product path has NO CPU fallback */
    FEASTHIP_ENUM = -7,      /* numerical breakdown (Cholesky of a Gram matrix, small eigensolver) */
    FEASTHIP_ECOMM = -8      /* RCCL failure (librccl.so.1 not found, communicator or collective error) */
};

/* element type tags for matrix values */
enum { FEASTHIP_F64 = 0, FEASTHIP_C128 = 1, FEASTHIP_I64 = 2, FEASTHIP_C64 = 3 /* dense matrices, lu_dense_c64 / lu_csc: ComplexF32 (the shifted matrix of the mixed_prec path) */ };
/* feasthip_set_dense only: OR into the type tag of a matrix stored row-major (numpy's default order; Julia passes column-major, no flag).
 * The library converts element type and storage order on the device. */
#define FEASTHIP_ROW_MAJOR 0x100
/* contour region kinds for the inside test (src/contour.jl:88-100) */
enum { FEASTHIP_REGION_DISC = 0, FEASTHIP_REGION_RECT = 1 };

/* ---- lifecycle --------------------------------------------------------- */
int feasthip_version(void);
/* number of visible gfx950 devices (0 when none); does not create a context */
int feasthip_device_count(void);
int feasthip_create(feasthip_handle_t* h, int device_id);
int feasthip_destroy(feasthip_handle_t h);
/* Device blocks of 64 MiB and more that a destroyed (or re-analysed) handle releases stay in a process-wide cache for the next
 * handle of this process.  The cache never holds more than ONE handle's footprint (a handle that is destroyed first returns what
 * earlier handles left on its device to the driver) and at most FEASTHIP_POOL_GB (default 48, 0 = no cache); this call returns
 * all of it to the driver.  The counterpart of the reference's finalize! for whole solver runs (src/utils.jl:173): memory is the
 * library's, the caller decides when it goes.
 * A destroyed handle (or factor object) also leaves its streams, events and 8 MiB of pinned landing slots for the next handle of
 * the same device (creating them costs 10-17 ms, which matters at the factorizer seam: one factor object per contour node and
 * iteration); at most 64 such sets are kept, this call destroys them as well, FEASTHIP_REUSE_STREAMS=0 disables the reuse. */
int feasthip_trim_memory(void);
const char* feasthip_last_error(feasthip_handle_t h);   /* h may be NULL: last create() error */
/* run all work of this handle on an existing hipStream_t (e.g. the caller's) */
int feasthip_set_stream(feasthip_handle_t h, void* hip_stream);

/* ---- problem definition -------------------------------------------------
 * The pencil / NEP is a matrix polynomial T(z) = sum_j z^j M_j, j = 0..nterms-1.
 *   feast!    (src/feast.jl:10-80):    T(z) = A - z I       -> M_0 = A, M_1 = -I
 *   gen_feast!(src/feast.jl:89-156):   T(z) = A - z B       -> M_0 = A, M_1 = -B
 *   nlfeast!  (src/nlfeast.jl:2-84):   T(z) closure; polynomial fast path
 *                                       (test/butterfly.jl:60-62).
 * A NULL matrix pointer means the identity.  `scale[j]` (2 doubles, re/im)
 * multiplies M_j (pass -1 for the B term of a linear pencil).
 */
int feasthip_set_dense(feasthip_handle_t h, int64_t n, int nterms, const void* const* mats,
                       const int* types, const double* scale);
/* SparseMatrixCSC fields (colptr, rowval, nzval) passed zero-copy; index_base 1 for Julia. */
int feasthip_set_csc(feasthip_handle_t h, int64_t n, int nterms, const int64_t* const* colptr,
                     const int64_t* const* rowval, const void* const* nzval, const int* types,
                     const double* scale, int index_base);

/* contour.nodes / contour.weights (src/contour.jl:3-22): K complex pairs each.
 * region parameters: DISC -> {c_re, c_im, r, 0}; RECT -> {bl_re, bl_im, tr_re, tr_im}. */
int feasthip_set_contour(feasthip_handle_t h, int K, const double* nodes, const double* weights,
                         int region_kind, const double* region);
/* Multi-GPU sharding: this handle owns nodes k = first, first+stride, ... */
int feasthip_set_node_shard(feasthip_handle_t h, int first, int stride);
/* The one collective of the path inside the library (round 4): one process per GPU, every process a handle with its node shard;
 * rank 0 draws an id (128 bytes, RCCL's ncclUniqueId) and hands it to the others by whatever channel the host program has (Julia:
 * Distributed / MPI.jl; Python: torch.distributed, a file, a socket), every rank then joins with feasthip_comm_init.  From then on
 * feasthip_contour_step (and _begin / _end; feasthip_feast_run with it) sums every row range of the moment Q = sum_k w_k ...
 * (src/feast.jl:57-72: the node loop :59-71 is what is sharded) over the ranks -- ncclAllReduce(sum, float64) on the library's own
 * communication stream behind the event of that range's accumulation, no host wait in between -- and returns with the summed moment
 * in place on every rank, so that the replicated Rayleigh-Ritz step needs no broadcast.  RCCL is loaded with dlopen("librccl.so.1") at
 * the first of these calls (FEASTHIP_ECOMM if it cannot be).  world = 1 is allowed (a one-rank communicator: the sum is the identity).
 * NOTE: executed with one rank only so far; no multi-GPU node was available to any round of this build. */
int feasthip_comm_unique_id(void* id128);
int feasthip_comm_init(feasthip_handle_t h, const void* id128, int rank, int world);
int feasthip_comm_destroy(feasthip_handle_t h);

/* ---- factorizer seam: facts[k] = factorizer(T(z_k))  (src/feast.jl:28-38, 104-114;
 *      src/nlfeast.jl:17-28).  analyze = ordering + symbolic (once per pattern),
 *      factorize = numeric LU of every local node, batched over nodes on the GPU. */
int feasthip_analyze(feasthip_handle_t h);
int feasthip_factorize(feasthip_handle_t h);

/* ---- iteration state -----------------------------------------------------
 * X is n x m0 ComplexF64 column-major, in/out (the `!` convention). */
int feasthip_set_x(feasthip_handle_t h, const void* X, int64_t n, int m0);
int feasthip_get_x(feasthip_handle_t h, void* X);
int feasthip_get_block(feasthip_handle_t h, int which /*0 X,1 R,2 Q,3 Q1*/, void* out);

/* Rayleigh-Ritz half of one iteration (src/feast.jl:41-50 / 117-127):
 * Q <- orth(Q); Aq = Q'AQ (, Bq = Q'BQ); reduced eigenproblem; X = Q Xq;
 * update_R! (src/utils.jl:111-116); residuals! (:166-171).
 * Outputs: lambda (m0 complex, sorted by (re,im)), res (m0), inside (m0 bytes). */
int feasthip_rr_step(feasthip_handle_t h, double* lambda, double* res, uint8_t* inside);
/* The reduced eigenproblem of that step alone (src/feast.jl:45 `eigen!(Aq)`, :122 `eigen!(Aq, Bq)`) as the device solves it for Hermitian
 * A, Hermitian positive definite B (NULL: identity), 1 <= m <= 64, column-major m x m complex128: lambda (m doubles, ascending), X (m x m,
 * X' B X = I, A X = B X diag(lambda)), status (8 doubles or NULL: flags {1 not Hermitian, 2 B not positive definite, 4 ill-conditioned,
 * 8 no convergence}, Jacobi sweeps, ||R^-1||_F / sqrt(m) of the column-scaled Cholesky factor of B, Hermitian defect, min/max diag(R)).
 * One 1024-thread workgroup, everything in LDS (csrc/kernels_small.hip). */
int feasthip_small_eig_herm(feasthip_handle_t h, int m, const void* A, const void* B, double* lambda, void* X, double* status);

/* on != 0: this handle runs only the Rayleigh-Ritz half (feasthip_set_x / feasthip_rr_step / feasthip_get_block): no ordering,
 * no symbolic analysis, no factor arenas; factorize / contour_step return FEASTHIP_ESTATE.  For callers that supply their own
 * factorizer / left_divider (src/feast.jl:5,11,84,90) and run the node loop (:59-71) themselves.  Call before feasthip_set_x. */
int feasthip_set_rr_only(feasthip_handle_t h, int on);

/* Contour half (src/feast.jl:57-72 / 134-148): Q = sum_k (X - T(z_k)^-1 R) diag(w_k/(z_k-lambda))
 * over the local nodes.  refactor != 0 reproduces store=false (src/utils.jl:175-179). */
int feasthip_contour_step(feasthip_handle_t h, int refactor);
/* The same step for the multi-GPU driver, without the final host synchronisation: _begin enqueues the sweeps and the
 * accumulation, the latter in `nchunks` contiguous row ranges of Q (rows of the n x ldq moment, original order);
 * _moment_chunk_wait(c) blocks until range c is final and returns it, so that its all-reduce can run while the later ranges
 * are still being accumulated; _end waits for the whole step (and fills stats.solve_ms).  feasthip_contour_step = begin(1 chunk) + end. */
int feasthip_contour_step_begin(feasthip_handle_t h, int refactor, int nchunks);
int feasthip_moment_chunk_wait(feasthip_handle_t h, int chunk, int64_t* row0, int64_t* rows);
int feasthip_contour_step_end(feasthip_handle_t h);

/* NLFEAST halves (src/nlfeast.jl:32-61 and :64-67, src/utils.jl:69-77,104-109,151-157). */
int feasthip_nl_contour_step(feasthip_handle_t h, int first_iteration);
int feasthip_nl_rr_step(feasthip_handle_t h, double* lambda, double* res, uint8_t* inside);
int feasthip_nl_orthonormalize_x(feasthip_handle_t h);    /* src/nlfeast.jl:12-13 */
int feasthip_nl_finalize_x(feasthip_handle_t h);          /* normalize!(X), src/nlfeast.jl:82 */

/* Device pointer of the local partial moment(s) so the host can sum them across
 * ranks (RCCL all-reduce through torch.distributed).  Layout: row-major
 * n x ldq complex128, `nblocks` consecutive blocks (1 for FEAST, 2 for NLFEAST). */
int feasthip_q_device(feasthip_handle_t h, void** dptr, int64_t* n, int64_t* ldq, int* nblocks);
/* Alternatively the caller supplies the moment buffer (e.g. a torch tensor's data_ptr, so the
 * collective runs on memory its framework owns): >= 2 * n * ldq * 16 bytes with
 * ldq = 16*ceil(m0/16).  Call before feasthip_set_x; NULL restores the internal buffer. */
int feasthip_set_q_buffer(feasthip_handle_t h, void* dptr, size_t bytes);

/* ---- whole loops (same semantics as the reference drivers) ----------------
 * feasthip_feast_run   = feast!/gen_feast! `for nit=0:iter` loop, src/feast.jl:40-73.
 * On return: lambda/res/inside for all m0 Ritz pairs (caller filters with
 * inside, src/feast.jl:77-79), X overwritten, *nit = iterations used. */
int feasthip_feast_run(feasthip_handle_t h, void* X, int64_t n, int m0, int iter, double eps,
                       int store, double* lambda, double* res, uint8_t* inside, int* nit);
int feasthip_nlfeast_run(feasthip_handle_t h, void* X, int64_t n, int m0, int iter, double eps,
                         double spurious, double* lambda, double* res, uint8_t* inside, int* nit);
/* mixed_prec of feast! (src/feast.jl:10,19-25): mixed != 0 runs the shifted block solves on complex64 copies of the
 * factors and block vectors (factorisation itself stays complex128, the copy is made once per factorisation), the
 * accumulation Q += (X - temp) diag(w/(z-lambda)) stays complex128 (src/feast.jl:68-70).  Call before feasthip_set_x /
 * feasthip_factorize; default 0. */
int feasthip_set_precision(feasthip_handle_t h, int mixed);
/* Conjugate fold (default on).  When every coefficient matrix and scale is real, T(conj z) = conj(T(z)); if in addition every
 * contour node z_k has its mirror image conj(z_k) on the contour (circles and rectangles centred on the real axis), the pair
 * shares ONE factorisation: T(conj z_k)^-1 R = conj(T(z_k)^-1 conj(R)) is obtained as m0 more right-hand-side columns of the
 * sweep over the factors of T(z_k).  The moment Q = sum_k w_k ... (src/feast.jl:57-72) is the same sum over all K nodes with
 * each node's own weight; only the number of factorisations (src/feast.jl:28-38) and the factor bytes streamed per iteration
 * halve.  Applies to feast!/gen_feast!/nlfeast!/nlfeast_moments! on real matrices; never to per-node coefficient overrides
 * (feasthip_set_coef_functions, feasthip_set_node_coefficients, dual_gen_feast).  Node shards (feasthip_set_node_shard) then
 * deal out pairs instead of nodes.  on = 0 factorises every node separately, exactly like the reference. */
int feasthip_set_conjugate_fold(feasthip_handle_t h, int on);
/* Lazy solve-ready conversion, for factorisations that serve ONE contour step (the reference's default store = false,
 * src/feast.jl:5,64-65,84,141-142: every iteration factorises every node again).  The conversion then leaves the update rows L21 of
 * every front raw (three quarters of its traffic) and the forward sweep takes one more launch per tree level.  feasthip_feast_run
 * switches it on by itself for store = 0 and restores the handle's setting afterwards; drivers of the phase calls
 * (feasthip_contour_step(h, refactor = 1)) set it here.  It takes effect at the next factorisation.  Ignored (full conversion) where
 * the adjoint sweeps read the factors: adjoint fold, dual_gen_feast; feasthip_ldiv_adjoint on lazily converted factors returns
 * FEASTHIP_ESTATE.  Results agree with the full conversion to rounding. */
int feasthip_set_lazy_conversion(feasthip_handle_t h, int on);

/* ---- NEPs that are not polynomials (SURVEY 8b "generic-T NLFEAST"; the reference takes any closure z -> matrix,
 * src/nlfeast.jl:2,21, src/utils.jl:107; test/gun2.jl:30-37, test/hadeler.jl:27-59).
 * (1) split form T(z) = sum_t f_t(z) M_t: the M_t are the terms of feasthip_set_dense/_csc, the scalar functions are
 *     supplied as ONE callback evaluated on the host whenever coefficients are needed (nodes at factorisation,
 *     Ritz values at every residual): z = count complex points (re,im), f = count x nterms complex, point-major.
 *     Everything else (feasthip_nlfeast_run, feasthip_nlfeast_moments_run, the phase calls) is unchanged.
 *     fn = NULL restores the polynomial default f_t(z) = z^t. */
typedef void (*feasthip_coef_fn)(const double* z, int count, double* f, void* user);
int feasthip_set_coef_functions(feasthip_handle_t h, feasthip_coef_fn fn, void* user);
/* Call order: feasthip_set_coef_functions, feasthip_set_node_coefficients, feasthip_set_conjugate_fold, feasthip_set_contour and
 * feasthip_set_node_shard may change whether the conjugate fold applies (per-node overrides and complex data switch it off) and
 * with it the number of factorisations this handle holds.  When they do, the analysis and the iteration state are invalidated:
 * feasthip_factorize re-analyses by itself, but feasthip_set_x has to be called again before the next feasthip_rr_step /
 * feasthip_contour_step (which otherwise return FEASTHIP_ESTATE).  Install overrides BEFORE feasthip_set_x. */
/* (2) arbitrary closure: hand over T(z_k) for every node as terms 0..K-1 together with the K x nterms table of node
 *     coefficients (row k = coefficients of T(z_k); identity for one matrix per node), run the loop from the host with
 *     feasthip_nl_contour_step / feasthip_nl_beyn_step, form R[:,i] = T(lambda_i) x_i on the host (the reference's
 *     update_R!, src/utils.jl:104-109) and return it with feasthip_nl_set_r.  coef = NULL removes the table. */
int feasthip_set_node_coefficients(feasthip_handle_t h, const double* coef, int K, int nterms);
/* Beyn SVD step only (src/utils.jl:69-77): Ritz values + column-normalised X on the device, no residual. */
int feasthip_nl_beyn_step(feasthip_handle_t h, double* lambda, uint8_t* inside);
/* R (n x m0 column-major complex128) for the next contour step. */
int feasthip_nl_set_r(feasthip_handle_t h, const void* R);

/* nlfeast_moments!(T, X, nodes, iter; c, r, eps, moments, store=true, spurious)  src/nlfeast.jl:173-318
 * (SURVEY 8f rank 1): 2*moments contour moments from the same shifted block solves, block-Hankel Beyn step.
 * M = moments*m0 candidates are returned sorted by relative residual (update_R_moments!, src/utils.jl:118-134):
 * lambda (M complex128 as re/im pairs), res (M), inside (M, may be NULL), Y (n x M column-major complex128, may be
 * NULL); X (n x m0, in/out) receives the m0 best candidates, column-normalised. */
int feasthip_nlfeast_moments_run(feasthip_handle_t h, void* X, int64_t n, int m0, int moments, int iter, double eps,
                                 double spurious, double* lambda, double* res, uint8_t* inside, void* Y, int* nit);

/* The same variant as phase calls, for a T that is an opaque closure (the reference's gun / fiber scripts,
 * test/gun.jl:36, test/fiber.jl:33): terms = T(z_k) per node + feasthip_set_node_coefficients, then per iteration
 *   feasthip_nlm_contour_step(first) -> feasthip_nlm_candidates (lambda[M], normalised candidates Y n x M)
 *   -> host: R[:,i] = T(lambda_i) y_i, res_i = |R_i| / |T(lambda_i)|_F  (update_R_moments!, src/utils.jl:118-123)
 *   -> feasthip_nlm_finish(R, res_in, ...) : sort by residual, X / R[:,1:m0] for the next step; outputs sorted.
 * R = res_in = NULL lets the library form the residuals itself (polynomial / split-form terms).  feasthip_nlm_get_y returns
 * the sorted candidates, feasthip_get_x the m0 best (normalise with feasthip_nl_finalize_x at the end). */
int feasthip_nlm_begin(feasthip_handle_t h, const void* X, int64_t n, int m0, int moments);
int feasthip_nlm_contour_step(feasthip_handle_t h, int first);
int feasthip_nlm_candidates(feasthip_handle_t h, double* lambda, void* Y);
int feasthip_nlm_finish(feasthip_handle_t h, const void* R, const double* res_in, double* lambda, double* res, uint8_t* inside);
int feasthip_nlm_get_y(feasthip_handle_t h, void* Y);

/* ---- dual_gen_feast!(Xr, Xl, A, B, contour; iter, eps, store)  src/feast.jl:158-257: right AND left
 * eigenvectors.  Set-up: terms {A, B, A', B'} with scales {1, -1, 1, -1} (feasthip_set_dense/_csc; A', B' serve the left residual
 * A' y - conj(lam) B' y, :214), the contour's K nodes and the K x 4 node coefficient table {(1, z_k, 0, 0)}: K factorisations of
 * A - z_k B; the adjoint systems (A - z_k B)' y = r that the reference factorises separately (lfacts, :185-197, used at :241) are
 * solved by the adjoint sweeps over the SAME factors (round 4).  The layout of rounds 2-3 is still accepted: a contour of 2K entries
 * {z; conj(z)} / {w; conj(w)} with the 2K x 4 table {(1, z_k, 0, 0)} U {(0, 0, 1, conj(z_k))}, i.e. the adjoint matrices factorised
 * as K more nodes (twice the factor memory).  Xr, Xl n x m0 in/out; lambda, resr, resl, inside have m0 entries (resl may be NULL);
 * the caller filters with `inside` (:256). */
int feasthip_dual_feast_run(feasthip_handle_t h, void* Xr, void* Xl, int64_t n, int m0, int iter, double eps, int store,
                            double* lambda, double* resr, double* resl, uint8_t* inside, int* nit);

/* ---- adjacent caller of the same seam: contour_estimate_eig, src/stochastic.jl:2-33.
 * est = real(sum_k w_k tr(X' (z_k B - A)^-1 X)) / m0 over the LOCAL nodes of this handle, X n x m0 random block
 * supplied by the caller (randn(ComplexF64, N, samples) in the reference). */
int feasthip_estimate_count(feasthip_handle_t h, const void* X, int64_t n, int m0, double* est);

/* ---- fine-grained seam: factorizer / left_divider / finalize!  (src/utils.jl:173-179).
 * F = factorizer(C): C is copied, never aliased (src/feast.jl:29-36).
 * left_divider(Y, F, R): Y = C^-1 R, R untouched (src/feast.jl:62).  Thread-safe
 * across distinct factor objects (src/nlfeast.jl:19-26, 36-55). */
int feasthip_lu_dense(int device_id, const void* C, int64_t n, feasthip_fact_t* F);
/* factorizer(ZmA) with ZmA::Matrix{ComplexF32} (mixed_prec, src/feast.jl:19-36): C is complex64 column-major; the
 * factor object then solves in complex64 (feasthip_ldiv still takes/returns complex128 buffers, the caller's
 * ComplexF32 `temp` is a cast of Y).  feasthip_lu_csc does the same for type == FEASTHIP_C64. */
int feasthip_lu_dense_c64(int device_id, const void* C, int64_t n, feasthip_fact_t* F);
int feasthip_lu_csc(int device_id, int64_t n, const int64_t* colptr, const int64_t* rowval,
                    const void* nzval, int type, int index_base, feasthip_fact_t* F);
int feasthip_ldiv(feasthip_fact_t F, void* Y, int64_t ldy, const void* R, int64_t ldr, int m0);
/* Y = C^-H R over the SAME factor object (Julia: `ldiv!(Y, F', R)`): the adjoint sweeps U^H, L^H of round 4.  The reference factorises
 * the adjoint matrices a second time (`lfacts`, src/feast.jl:185-197, used at :241); a caller of this seam need not.  complex128 only. */
int feasthip_ldiv_adjoint(feasthip_fact_t F, void* Y, int64_t ldy, const void* R, int64_t ldr, int m0);
int feasthip_fact_free(feasthip_fact_t F);
/* How F was factorised: *symmetric = 1 if C equals its transpose and the pivot-light elimination passed its growth check
 * (*growth = max(1, max |l_ij|) * max(1, max |u_ij| / ||C||) as it saw them), 2 if C is Hermitian off its diagonal and the elimination without interchanges (L, U apart) passed
 * its own, 0 for partial pivoting; *perturbed = statically perturbed pivots.  Any pointer may be NULL. */
int feasthip_fact_info(feasthip_fact_t F, int* symmetric, double* growth, int64_t* perturbed);

/* ---- statistics ------------------------------------------------------------ */
typedef struct {
    int64_t n, nsuper, nnz_l, nnz_u, fact_elems, max_front, max_p, depth;
    double flops_factor;        /* real flops of one numeric factorisation        */
    double analyze_seconds;     /* host ordering + symbolic + plan upload (without the arena allocation below) */
    double factor_ms;           /* last feasthip_factorize, all local nodes (HIP events) */
    double solve_ms;            /* last contour step: block solves + accumulation   */
    double rr_ms;               /* last Rayleigh-Ritz step                          */
    double solve_bytes;         /* algorithmic bytes of one contour step */
    double solve_flops;         /* real flops of one contour step                   */
    int64_t perturbed_pivots;
    int32_t local_nodes, m0, ldw, dense;     /* local_nodes: contour nodes this handle integrates over */
    double alloc_seconds;       /* hipMalloc of the factor / contribution arenas at the end of feasthip_analyze (the driver may
                                 * have to scrub tens of GB freed by a previous process: 0.03 .. 2 s for the same request) */
    int32_t factor_units;       /* factorisations held: local_nodes, or local_nodes / 2 with the conjugate fold */
    int32_t folded;             /* 1 if the conjugate fold is active (feasthip_set_conjugate_fold); 2: the adjoint fold (Hermitian, not real, coefficient
                                 * matrices: T(conj z) = T(z)^H, the mirror node of a unit is solved by the adjoint sweeps over the unit's factors) */
    double solve_bytes_streamed;/* bytes the sweeps of one contour step move (a folded pair reads its factor once) */
    int32_t symmetric;          /* 1 if the last factorisation took the pivot-light path: every coefficient matrix equals its transpose
                                 * (complex symmetric T(z)), no row interchanges, growth-checked; 2: every coefficient matrix equals its
                                 * conjugate transpose off the diagonal (Hermitian pencils, and their assembled shifted matrices z D - A):
                                 * no row interchanges either, L and U kept apart, growth-checked (FEASTHIP_NOPIV_GEN=0 switches it off);
                                 * 0: partial pivoting inside the fronts */
    int32_t rr_device_sweeps;   /* last feasthip_rr_step: Jacobi sweeps of the reduced eigenproblem when it ran on the device (Hermitian(-definite)
                                 * problem, m0 <= 64, well-conditioned filtered block; FEASTHIP_RR_DEVICE=0 switches it off); 0: host solver */
    double growth;              /* max(1, max |l_ij|) * max(1, max |u_ij| / ||T||) of that factorisation (paths 1 and 2; beyond 256 the handle refactorises with pivoting) */
    double proj_ms;             /* last feasthip_rr_step: the projection Aq = Q'(AQ) (src/feast.jl:43,118), 8 n m0^2 flops on the f64 matrix cores */
    int32_t early_sweep_levels; /* last feasthip_contour_step(h, 1) (store = false, src/feast.jl:64-65): tree levels whose forward sweep was enqueued under
                                 * the factorisation, behind their own level's conversion (0: none -- FEASTHIP_EARLY_SWEEP=0, dense matrices, dual_gen_feast);
                                 * factor_ms then contains those launches and solve_ms only the root's forward level and the backward sweep */
    int32_t reserved0;
} feasthip_stats_t;
int feasthip_get_stats(feasthip_handle_t h, feasthip_stats_t* out);
/* One instrumented contour sweep: HIP-event time per kernel class.  out = 8 rows x
 * {ms, algorithmic bytes, flops, launches}: 0 gather, 1 assemble, 2 diag solve,
 * 3 panel apply (the dominant kernel), 4 accumulate. */
int feasthip_profile_solve(feasthip_handle_t h, double* out);
/* per-launch rows of the last feasthip_profile_solve: {class, ms, bytes, flops, tiles, tree depth (-1: not a level launch)};
 * returns the row count */
int feasthip_profile_detail(feasthip_handle_t h, double* out, int cap);
/* Elimination-tree levels of the analysed pattern, root = depth 0: per depth 5 doubles {fronts, sum of pivot columns p, sum of
 * update rows u, factor entries sum(p*p + 2*u*p) = that level's share of nnz(L)+nnz(U), widest pivot block}; returns the number of levels.  With it the
 * SURVEY 8(d) bytes of a contour step, (nnz(L)+nnz(U))*20 + 2(n+1)*4 + 4*n*m0*16 per node, split by tree level. */
int feasthip_level_table(feasthip_handle_t h, double* out, int cap);

/* ---- MatrixMarket coordinate files: the on-disk format of the reference's data sets (the data/ .mtx files,
 * written by test/gen_butterfly.jl:6-41, read with MatrixMarket.jl in test/butterfly.jl:57-59, test/polynomial.jl:5-11).
 * Host only.  feasthip_mm_info -> sizes (nnz after expanding symmetric/hermitian/skew storage and merging duplicates;
 * field 0 real, 1 complex, 2 integer, 3 pattern); feasthip_mm_read_csc fills caller-allocated CSC arrays (colptr cols+1,
 * rowval/nzval nnz; 0-based int64 indices, sorted rows, values always complex128) -- the layout feasthip_set_csc takes with
 * index_base = 0.  feasthip_mm_write_csc writes `coordinate <field> general` like the reference's writer. */
const char* feasthip_mm_last_error(void);
int feasthip_mm_info(const char* path, int64_t* rows, int64_t* cols, int64_t* nnz, int* field);
int feasthip_mm_read_csc(const char* path, int64_t* colptr, int64_t* rowval, void* nzval, int64_t nnz_capacity);
int feasthip_mm_write_csc(const char* path, int64_t rows, int64_t cols, const int64_t* colptr, const int64_t* rowval,
                          const void* nzval, int type, int index_base);

/* ---- host-only helpers (no GPU needed; used by the CPU test-suite) ---------- */
/* symbolic analysis of a pattern: stats[0..7] = nsuper, nnz_l, nnz_u, fact_elems,
 * max_front, max_p, depth, flops(as int64).  perm_out (n ints, new->old) optional;
 * fronts_out (4 ints per supernode, at most 4 n: pivot columns, update rows, depth, parent) optional. */
int feasthip_symbolic_probe(int64_t n, const int64_t* colptr, const int64_t* rowval, int index_base,
                            int ordering, int64_t* stats, int32_t* perm_out,
                            int32_t* sn_of_out, int32_t* fronts_out);
/* structural check: returns 0 if every (i,j) of the pattern and of its LU fill (given as a
 * dense n x n 0/1 byte mask, column-major, in ORIGINAL ordering after the caller applied
 * perm) lies inside the predicted supernodal structure. */
int feasthip_symbolic_contains(int64_t n, const int64_t* colptr, const int64_t* rowval, int index_base,
                               int ordering, const int32_t* ii, const int32_t* jj, int64_t count);
/* node sharding of feasthip_set_node_shard / the conjugate fold, without a device: for K contour nodes (complex pairs) and the shard
 * (first, stride), local[i] (count entries, caller-allocated K ints) are the nodes whose matrices this shard factorises and
 * partner[i] the mirror node that rides along (-1 without the fold); *folded = 1 if pairs were dealt out.  fold_allowed = the
 * matrices are real and there are no per-node coefficient overrides. */
int feasthip_shard_plan(int K, const double* nodes, int fold_allowed, int first, int stride, int32_t* local, int32_t* partner,
                        int* count, int* folded);
/* small dense complex eigensolver used for the reduced problem (replaces Julia's
 * eigen!(Aq) / eigen!(Aq,Bq), src/feast.jl:45,122): m x m column-major; B may be NULL.
 * Outputs sorted by (re, im), unit 2-norm vectors. */
int feasthip_small_eig(int m, const double* A, const double* B, double* lambda, double* V);

#ifdef __cplusplus
}
#endif
#endif /* FEASTHIP_H */
