// Host orchestration of the FEAST inner loop on one MI355X: problem storage,
// launch plans built from the symbolic analysis, numeric factorisation of all
// local contour nodes, block solves + moment accumulation, Rayleigh-Ritz step.
#pragma once
#include "../../include/feasthip.h"
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "symbolic.hpp"

namespace feasthip {

typedef std::complex<double> zc;

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t cap_bytes = 0;              // size of the underlying block (a recycled block may be larger than n elements)
    int dev = 0;                       // device the block was allocated on (a block is released to the cache / the driver of THAT device)
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), cap_bytes(o.cap_bytes), dev(o.dev) { o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; cap_bytes = o.cap_bytes; dev = o.dev; o.p = nullptr; o.n = 0; o.cap_bytes = 0; } return *this; }
    ~DevBuf() { release(); }
    void release();
    bool alloc(size_t count);          // false on failure
    bool upload(const T* h, size_t count, hipStream_t st);
};

void trim_device_memory();     // frees the process-wide cache of released device blocks (engine.hip)
void pool_drop_device(int dev); // ... of one device

struct Launch {        // a slice of the tile / list buffers
    int64_t off = 0;
    int count = 0;
    double bytes = 0, flops = 0;   // algorithmic bytes / real flops of this launch for ONE contour node
    double vrows = 0;              // block-vector rows read or written by the in-block steps (one node)
    int sl = 4;                    // tile height of the launch in 16-row slabs (launch_apply: max_slabs)
    XcdParts xp = {{0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, -1};     // launch_apply launches: the tiles of each XCD (dev_common.hpp; filled by build_plans)
};

struct FactorStep { int k0; Launch panel, swap, trsm, gemm, gemm0, gemm_in, grp;
                    Launch lsg; };     // pivot-light path, first step of a 128-column group: 64-row slices of the rows below the group's diagonal block (k_lscale_grp)
struct FactorLevel { Launch extend, extend1; std::vector<FactorStep> steps; Launch cbgemm;
                     Launch cbgemm_sym;     // symmetric pencils: the tiles on and below the diagonal of the contribution block only (nothing above it is stored or read)
                     Launch fronts; Launch binv, couple, scale;
                     Launch outer[3], inner[3]; };      // coupling of 128 -> 256 ([0]), 256 -> 512 ([1]) and 512 -> 1024 ([2]) super blocks (launch_couple_outer / _level)
struct SolveLevel { Launch assemble; std::vector<Launch> f1; Launch f2; Launch b1; std::vector<Launch> b2; Launch b3; Launch b1red; int b1split = 0;
                    Launch f2red; int f2split = 0;      // forward L21 products split along K (few nodes per GPU): number of parts, reduction tiles
                    Launch f2y, f2c;      // the tiles of f2 apart: diagonal solves (kind 2) / contribution rows (kind 1) -- the two launches of a sweep over lazily converted factors
                    Launch b3s;       // B3 tiles of the fronts whose B1 tile is kind 6 (launched only when the apply kernel does not fuse them)
                    double asm_rows = 0, pull_rows = 0;      // children's contribution rows landing in pivot blocks / in contribution rows of this level
                    // adjoint sweeps (kernels.hpp, tile kinds 8 .. 13): T(z)^H x = r over the same factors
                    Launch afd; std::vector<Launch> afe; Launch af2; Launch ab1; std::vector<Launch> abe; };

struct SparseTerm {
    std::vector<int64_t> colptr;
    std::vector<int> rowval;
    std::vector<zc> val;
    bool identity = false;
    zc scale = 1.0;
    double maxabs = 0;
    // device
    DevBuf<int64_t> d_rowptr;   // CSR (transpose of the CSC input) for SpMM
    DevBuf<int> d_col;
    DevBuf<cz> d_val;
    DevBuf<int64_t> d_dest;     // arena destination of each CSC entry
    DevBuf<cz> d_cscval;
    DevBuf<int> d_csrq;         // position of every CSC entry in the CSR arrays (d_val = d_cscval permuted, k_permute_vals)
    DevBuf<cz> d_dense;         // dense column-major copy (dense path)
    bool diagonal = false;      // sparse term with entries on the diagonal only (a lumped mass matrix, Diagonal(...) in the reference's tests)
    DevBuf<cz> d_diag;          // ... its diagonal (n entries, zeros where no entry is stored)
};

struct ProfileRow { double ms = 0, bytes = 0, flops = 0; int launches = 0; };

// Which contour nodes a shard (first, stride) integrates over, and which of them are factorised.  fold_allowed: real coefficient
// matrices and no per-node overrides.  When every node has its mirror image conj(z) among the nodes (and there are at least
// `stride` pairs) the units dealt out are conjugate pairs: local[i] is the member in the upper half plane (the one that is
// factorised), partner[i] its mirror node; otherwise single nodes first, first + stride, ... with partner[i] = -1.
// Host-only (also behind feasthip_shard_plan for the CPU test-suite); returns whether the conjugate fold is active.
bool plan_node_shard(const std::vector<zc>& nodes, int first, int stride, bool fold_allowed, std::vector<int>& local, std::vector<int>& partner);

class Engine {
public:
    explicit Engine(int device);
    ~Engine();
    std::string err;
    int status = 0;

    int set_stream(void* s);
    int set_dense(int64_t n, int nterms, const void* const* mats, const int* types, const double* scale);
    int set_csc(int64_t n, int nterms, const int64_t* const* colptr, const int64_t* const* rowval,
                const void* const* nzval, const int* types, const double* scale, int index_base);
    int set_contour(int K, const double* nodes, const double* weights, int kind, const double* region);
    int set_shard(int first, int stride);
    int set_q_buffer(void* dptr, size_t bytes);
    int set_precision(int mixed);
    bool mixed() const { return mixed_; }
    // Conjugate fold: a pencil / polynomial with REAL coefficient matrices satisfies T(conj z) = conj(T(z)), so a contour node
    // and its mirror image share ONE factorisation; the mirror system rides along as m0 more right-hand-side columns conj(R)
    // (T(conj z)^-1 R = conj(T(z)^-1 conj(R))).  On by default whenever every node of the contour has its mirror node.
    int set_fold(int on);
    int set_lazy_conversion(int on);
    bool folded() const { return fold_; }
    int solve_width() const { return fold_ ? 2 * ldw : ldw; }    // leading dimension of the per-unit solve blocks
    int set_coef_functions(feasthip_coef_fn fn, void* user);
    int set_node_coefficients(const double* coef, int K, int nterms);
    int nl_set_r(const void* R);
    int analyze();
    int analyze_rr_only();
    int set_rr_only(int on) { if ((on != 0) != rr_only_) { rr_only_ = on != 0; analyzed_ = factored_ = have_x_ = false; } return 0; }
    int factorize();
    int set_x(const void* X, int64_t n, int m0);
    int get_block(int which, void* out);
    int rr_step(double* lambda, double* res, uint8_t* inside);
    // the Hermitian(-definite) Rayleigh-Ritz step without a host round trip (engine_rr.hip); *done = false: the caller takes the host path
    int rr_step_device(double* lambda, double* res, uint8_t* inside, bool* done);
    int small_eig_herm(int m, const void* A, const void* B, double* lambda, void* X, double* status);   // the device solver alone (tests)
    int rr_device_sweeps = 0;             // Jacobi sweeps of the last device-side reduced eigenproblem (0: the host solved it)
    int contour_step(int refactor);
    int contour_step_begin(int refactor, int nchunks);
    int moment_chunk_wait(int c, int64_t* row0, int64_t* rows);
    int comm_allreduce_rows(int row0, int rows);
    int contour_step_end();
    // RCCL inside the library (round 4): with a communicator on the handle every range of the moment is all-reduced (sum) over the
    // ranks on the library's own communication stream, behind the event of that range's accumulation -- stream-ordered, no host wait --
    // and contour_step / contour_step_end return with the SUMMED moment in place.  The functions are resolved from librccl.so.1 at the
    // first use (dlopen: the library carries no link-time dependency on RCCL, and shares the copy a host process already loaded).
    static int comm_unique_id(void* id128, std::string& err);
    int comm_init(const void* id128, int rank, int world);
    int comm_destroy();
    bool has_comm() const { return comm_ != nullptr; }
    double allreduce_ms = 0;              // last contour step: first all-reduce enqueued .. last one finished (HIP events on the communication stream)
    int nl_contour_step(int first);
    int nl_rr_step(double* lambda, double* res, uint8_t* inside, int external_residual = 0);
    int nl_orthonormalize_x();
    int nl_finalize_x();
    int q_device(void** p, int64_t* n, int64_t* ldq, int* nblocks);
    int feast_run(void* X, int64_t n, int m0, int iter, double eps, int store, double* lambda, double* res,
                  uint8_t* inside, int* nit);
    int nlfeast_run(void* X, int64_t n, int m0, int iter, double eps, double spurious, double* lambda, double* res,
                    uint8_t* inside, int* nit);
    int nlm_begin(const void* X, int64_t n, int m0, int moments);
    int nlm_step_contour(int first);
    int nlm_step_candidates(double* lambda, void* Y);
    int nlm_step_finish(const void* R, const double* res_in, double* lambda, double* res, uint8_t* inside);
    int nlm_get_y(void* Y);
    // dual_gen_feast! (src/feast.jl:165-257): terms {A, B, A', B'}, 2K nodes (K right + K adjoint systems)
    int dual_run(void* Xr, void* Xl, int64_t n, int m0, int iter, double eps, int store, double* lambda, double* resr,
                 double* resl, uint8_t* inside, int* nit);
    // nlfeast_moments! (src/nlfeast.jl:173-318): lambda/res/inside have moments*m0 entries, Y is n x moments*m0
    int nlfeast_moments_run(void* X, int64_t n, int m0, int moments, int iter, double eps, double spurious, double* lambda,
                            double* res, uint8_t* inside, void* Y, int* nit);
    // fine-grained seam: Y = T^-1 R for the single factorised matrix of this engine (node 0)
    int ldiv(void* Y, int64_t ldy, const void* R, int64_t ldr, int m0, int adjoint = 0);
    int estimate_count(const void* X, int64_t n, int m0, double* est);
    int profile_solve(ProfileRow* rows /*[8]*/);
    std::vector<double> prof_detail;   // per launch: class, ms, bytes, flops, tiles, tree depth (filled by profile_solve)

    // stats
    double analyze_seconds = 0, alloc_seconds = 0, factor_ms = 0, solve_ms = 0, rr_ms = 0, proj_ms = 0;
    int early_levels_ = 0;                 // tree levels of the last factorisation's early forward sweep (feasthip_stats_t::early_sweep_levels)
    int64_t perturbed = 0;
    Symbolic S;
    int n = 0, nterms = 0, K = 0, Kloc = 0, m0 = 0, ldw = 0;
    bool dense = false;
    bool nopiv_gen_used = false;          // ... the elimination without interchanges for Hermitian (not symmetric) pencils
    bool nopiv_used = false;              // the last factorisation ran the pivot-light path (symmetric pencil) and passed its growth check
    double growth = 0;                    // max |l_ij| it saw (0 on the pivoted path, where |l_ij| <= 1 by construction)
    double solve_bytes() const;           // algorithmic bytes of one contour step (per contour NODE)
    double solve_bytes_streamed() const;  // what the sweeps actually stream (folded pairs read their factor once)
    double solve_flops() const;
    int nodes_local() const { return Kloc * ((fold_ || adj_fold_) ? 2 : 1); }
    // Adjoint fold (round 4): a pencil / polynomial with HERMITIAN coefficient matrices (and real scales) satisfies T(conj z) = T(z)^H, so
    // a contour node and its mirror image share one factorisation here too; the mirror system is solved by the adjoint sweeps (U^H, L^H)
    // over the same factors, with the right-hand side as it is.  Chosen where the conjugate fold does not apply (complex matrices).
    bool adj_folded() const { return adj_fold_; }
    // solve instances: units, plus one adjoint instance per unit under the adjoint fold and in dual_gen_feast (block vectors W / Y,
    // contribution arenas and coefficient tables are sized and indexed by instance: [0, Kloc) primal, [Kloc, 2 Kloc) adjoint)
    int ninst() const { return Kloc * ((adj_fold_ || dual_adj_) ? 2 : 1); }

private:
    int device_;
    hipStream_t stream_ = nullptr;
    bool streams_warm_ = false;
    bool fuse_b3_ = true;                 // plan kind-6 tiles (FEASTHIP_FUSE_B3=0: never)           // analyze() touches every stream once (queue creation) beside the host-side analysis
    bool own_stream_ = false;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr, ev_p0_ = nullptr, ev_p1_ = nullptr;    // ev_p*: around the projection Aq = Q'(AQ) of rr_step
    std::vector<SparseTerm> terms_;
    std::vector<zc> nodes_, weights_;      // all K
    std::vector<int> local_;               // indices of local nodes (conjugate fold: the member of each local pair that is factorised)
    std::vector<int> partner_;             // conjugate fold: index of the mirror node of local_[kk], else -1
    bool fold_ = false, fold_enabled_ = true, terms_real_ = false;
    bool adj_fold_ = false, adj_fold_enabled_ = true;   // FEASTHIP_ADJ_FOLD=0
    bool terms_herm_exact_ = false;        // every coefficient matrix equals its conjugate transpose INCLUDING a real diagonal
    bool dual_adj_ = false;                // dual_gen_feast with K factorisations: the left systems are adjoint sweeps
    bool analyzed_once_ = false;           // analyze() ran on this handle before (its device buffers may be in use by queued work)
    bool terms_hermitian_ = false;         // every coefficient matrix equals its conjugate transpose (elimination without interchanges, L and U apart)
    bool nopiv_gen_enabled_ = true, nopiv_gen_failed_ = false;      // FEASTHIP_NOPIV_GEN=0; set when the growth check rejected it for this handle's matrices
    bool terms_symmetric_ = false;         // every coefficient matrix equals its transpose: T(z) = T(z)^T for all z (pivot-light factorisation)
    bool nopiv_enabled_ = true, nopiv_failed_ = false;   // FEASTHIP_NOPIV=0 / a factorisation whose growth check failed: pivoted path from then on
    double nopiv_max_growth_ = 256.0;      // largest max(1, max |l_ij|) * max(1, max |u_ij| / ||T||) an elimination without interchanges is accepted with (FEASTHIP_NOPIV_GROWTH)
    // ... and with complex64 sweeps over the factors (mixed_prec): the error of a sweep grows with the same product, and at 2^-24 a growth
    // of 14-37 already put the moment of a dense symmetric 1200 x 1200 pencil off by 1.5e-2 -- the iteration stagnated (tools/fuzz_feast.py,
    // seed 9002 case 103; with partial pivoting 8e-4 and 3 iterations).  FEASTHIP_NOPIV_GROWTH32.
    double nopiv_max_growth32_ = 4.0;
    double nopiv_growth_bound() const { return mixed_ ? std::min(nopiv_max_growth_, nopiv_max_growth32_) : nopiv_max_growth_; }
    bool use_nopiv() const { return terms_symmetric_ && nopiv_enabled_ && !nopiv_failed_; }
    bool use_nopiv_gen() const { return terms_hermitian_ && !terms_symmetric_ && nopiv_enabled_ && nopiv_gen_enabled_ && !nopiv_gen_failed_; }
    void rebuild_shard();                  // local_ / partner_ / fold_ / Kloc from contour, shard, terms and overrides
    template <class F> std::vector<zc> node_table(F f, int nq = 1) const;   // [q][unit][solve_width] table of f(node, column, q) for the accumulation kernel
    int shard_first_ = 0, shard_stride_ = 1;
    int region_kind_ = 0;
    double region_[4] = {0, 0, 1, 0};
    bool analyzed_ = false, factored_ = false, have_x_ = false;
    std::vector<zc> lambda_;               // current Ritz values (m0)
    std::vector<std::vector<double>> frob_;  // <M_s, M_t>_F table for the NLFEAST residual norm

    // device symbolic
    DevBuf<FrontD> d_fr_;
    DevBuf<int> d_struct_, d_rel_, d_child_, d_perm_, d_iperm_, d_c0row_, d_drow_, d_cmap_, d_ebp_;
    DevBuf<int> d_asmflag_;     // fused right-hand-side gather (kernels.hpp: DevSym::asmflag; FEASTHIP_FUSE_GATHER=0 turns it off)
    DevBuf<Tile> d_tiles_;
    DevBuf<int> d_lists_;
    DevBuf<int> d_xperm_;       // XcdParts::perm
    DevBuf<int> d_pmap_;        // DevSym::pmap
    std::vector<FactorLevel> fplan_;
    std::vector<SolveLevel> splan_;
    DevSym dsym_;
    // numeric
    DevBuf<cz> d_fact_, d_cb_[2], d_coef_;
    DevBuf<const cz*> d_matptr_;
    DevBuf<int> d_piv_, d_rowperm_, d_rowinv_, d_nperturb_;
    DevBuf<double> d_tiny_;
    DevBuf<cz> d_wscr_;                   // pivot-light group kernels: inv(U_cc) of the 32-column blocks of the current group, per unit and front of the level
    int64_t wstride_ = 0;                 // ... scratch slots per unit (one per front of the widest level; kernels_factor.hip: WSLOT entries each)
    DevBuf<int64_t> d_growth_;            // per unit: max |l_ij| of the pivot-light path (bit pattern of a double)
    // iteration state (row-major n x ldw)
    DevBuf<cz> d_X_, d_R_, d_Q_, d_Q1_, d_T1_, d_T2_, d_MX_, d_W_, d_Y_, d_crow_[2], d_sk_;
    int64_t sk_rows_ = 0;                   // rows of split-K scratch per node
    std::vector<int64_t> sk_off_, skf_off_;  // per front (or -1): backward / forward split-K scratch rows
    DevBuf<cz> d_e0_, d_e1_, d_small_, d_lam_, d_gram_scratch_, d_io_;
    DevBuf<double> d_nrm_, d_nrm_scratch_;
    DevBuf<cz> d_rrscr_;                 // device-side reduced eigenproblem: T = R^-1 between the congruence and the back-transform (64 x 64)
    DevBuf<double> d_rrstat_;            // ... its status words (16: 8 public + phase timings), then the squared column norms of X = Q Xq (ldw)
    bool rr_device_ = true;              // FEASTHIP_RR_DEVICE=0: reduced eigenproblem always on the host
    int rr_dev_state_ = 0;               // -1: the projected matrices of this handle's pencil are not Hermitian (host path from then on)
    double rr_cond_pass2_ = 1e3;         // host path, generalized problem: the second CholeskyQR pass is dropped below this value of the same quantity
    double rr_cond_max_ = 8.0;           // largest ||R^-1||_F / sqrt(m) of the column-scaled basis the device path takes without orthonormalising first

    // pinned staging slots for the small host <-> device hand-offs of the Rayleigh-Ritz step (Gram matrices, reduced eigenvectors,
    // coefficient tables): pageable memory costs an extra runtime staging copy and 10-20 us per transfer
    static constexpr int PIN_SLOTS = 8;
    static constexpr size_t PIN_SLOT_BYTES = (size_t)1 << 20;
    char* pin_ = nullptr;
    std::string res_sig_;                // stream configuration (environment switches) this handle was created under
    bool take_resources();               // streams / events / pinned slots of an earlier handle of this device (engine.hip: HandleResources)
    bool give_resources();
    void* pin_slot(int slot, size_t bytes) { return (pin_ && bytes <= PIN_SLOT_BYTES) ? (void*)(pin_ + (size_t)slot * PIN_SLOT_BYTES) : nullptr; }
    hipError_t d2h_async(int slot, void* host_dst, const void* dev, size_t bytes, void** landed);      // *landed: where the bytes arrive (copy out after the sync if != host_dst)
    hipError_t h2d_async(int slot, void* dev, const void* host_src, size_t bytes);
    cz* q_ext_ = nullptr;
    size_t q_ext_bytes_ = 0;
    cz* Qp() const { return q_ext_ ? q_ext_ : d_Q_.p; }
    int fail(int code, const std::string& msg) { err = msg; status = code; return code; }
    int hip_fail(hipError_t e, const char* what);
    bool build_plans();
    int alloc_iteration_buffers();
    void apply_term(int t, const cz* V, cz* out, zc alpha, bool accumulate);
    // a tall row-major block (rows x ld, m meaningful columns) and the scratch its reductions use
    struct Tall { int rows, m, ld; cz* small; cz* lam; cz* gram; double* nrm; double* nrm_scratch; };
    Tall tall_default() { return Tall{n, m0, ldw, d_small_.p, d_lam_.p, d_gram_scratch_.p, d_nrm_.p, d_nrm_scratch_.p}; }
    int orthonormalize(cz* Q, cz* tmp, std::vector<zc>* Rtotal) { return orthonormalize(tall_default(), Q, tmp, Rtotal); }
    int orthonormalize(const Tall& t, cz* Q, cz* tmp, std::vector<zc>* Rtotal, int min_passes = 2);
    int orthonormalize_rr(const Tall& t, cz* Q, cz* tmp, std::vector<zc>& Rtot, std::vector<uint8_t>& keep);
    int col_norms(const cz* V, std::vector<double>& out) { return col_norms(tall_default(), V, out); }
    int col_norms(const Tall& t, const cz* V, std::vector<double>& out);
    void apply_term_ld(int t, const cz* V, cz* out, zc alpha, bool accumulate, int ld);
    double poly_frob_norm(zc lam);
    void eval_coefs(const zc* pts, int count, std::vector<zc>& out);
    int residual_block(const cz* X, cz* MX, cz* R, const std::vector<zc>& lam, int mcols, int ld, cz* d_coeftab);
    feasthip_coef_fn coef_fn_ = nullptr;
    void* coef_user_ = nullptr;
    std::vector<zc> node_coef_;              // optional K x nterms override of f_t(z_k) (per-node matrices)
    DevBuf<cz> d_coeftab_;           // ||T(lam)||_F from the <M_s, M_t>_F table
    int build_frob_table();
    // nlfeast_moments state
    int mm_ = 0, ldM_ = 0;
    DevBuf<cz> d_Qm_, d_H0_, d_H1_, d_Ht_, d_Ym_, d_Rm_, d_MXm_, d_em_, d_smallM_, d_lamM_, d_gramM_;
    DevBuf<double> d_nrmM_, d_nrmMs_;
    DevBuf<int> d_colidx_;
    DevBuf<cz> d_Xl_, d_Rl_;
    int dual_contour_step();
    int dual_rr_step(double* lambda, double* resr, double* resl, uint8_t* inside);
    int nlm_alloc(int moments);
    int nlm_contour_step(int first);
    int nlm_rr_step(std::vector<zc>& lam, std::vector<double>& res);
    int nlm_candidates(std::vector<zc>& lam);
    int nlm_finish(std::vector<zc>& lam, std::vector<double>& res, const void* ext_R, const double* ext_res);
    int nlm_rank_ = 0;
    std::vector<zc> nlm_lam_;
    int solve_all(const cz* rhs);            // W_k = T(z_k)^-1 rhs for all local nodes
    void inside_test(const zc* lam, uint8_t* inside) const;
    int solve_all_impl(const cz* rhs, ProfileRow* prof);
    int solve_range(hipStream_t st, int kb, int kc, const cz* rhs, ProfileRow* prof, int part = 0, int only_level = -1);
    // Adjoint sweeps T(z_k)^H x = rhs over the factors of the units kb .. kb + kc - 1; the block vectors of the sweep are the instance
    // slots slot0 .. slot0 + kc - 1 of W / Y / the contribution arenas.  The solution: adj_solution() (W, or Y when the fronts interchanged rows)
    int solve_range_adj(hipStream_t st, int kb, int kc, int slot0, const cz* rhs, int ldrhs, int part = 0, int only_level = -1);      // part / only_level: as solve_range
    const cz* adj_solution() const { return (nopiv_used || nopiv_gen_used) ? d_W_.p : d_Y_.p; }
    DevBuf<int> d_rowid_;                 // identity "row interchanges" (relative index within the supernode) for the adjoint sweeps' gather / assembly
    template <typename T>
    int solve_range_t(hipStream_t st, int kb, int kc, const cz* rhs, ProfileRow* prof, T* fact, T* Wall, T* Yall, T* crow0, T* crow1, T* skall, int part, int only_level);
    // mixed precision (feast!'s mixed_prec, src/feast.jl:19-25): complex64 copies of factors and block vectors
    bool mixed_ = false;
    DevBuf<cf> d_fact32_, d_W32_, d_Y32_, d_crow32_[2], d_sk32_;
    int stream_groups() const;
    void accumulate_solutions(int K_, const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1, int row0 = 0, int nrows = -1);
    bool streams_forced_ = false;
    DevBuf<cz> d_mvpart_;                // split-K partial sums of the dense term products (k_dense_mv)
    hipStream_t la_stream_ = nullptr;    // high-priority stream of the look-ahead panels
    hipStream_t fin_stream_ = nullptr;   // low-priority stream of the solve-ready conversion
    hipEvent_t ev_fin_ = nullptr, ev_fin_fork_ = nullptr;
    hipStream_t rest_stream_ = nullptr;  // trailing updates of the look-ahead levels: a CU mask keeps reserve_cus_ CUs per XCD free for la_stream_
    int reserve_cus_ = 4;      // (round 5: 4 per XCD = 32 CUs; C2 96.1 -> 94.6 ms, C5 21.9 -> 21.3, C3 unchanged; 8 and more cost the products more than the chain gains)
    std::vector<hipEvent_t> ev_la_;      // look-ahead events: [0,1] panel done (alternating), [2,3] rest update done, [4] fork, [5] join, [6] rest stream join
    bool look_ahead_ = true, finalize_async_ = true, fused_utrsm_ = true, fuse_extend_ = true;
    int fuse_min_p_ = 1;              // smallest pivot-block width of a level whose Schur product pulls the children's contributions itself (round 3: every level; 128 before, A/B 84.5 -> 82.7 ms)
    int group_panels_ = 4;            // panels per deferred trailing update in the look-ahead factorisation
    // Lazy conversion (feasthip_set_lazy_conversion; feast_run switches it on for store = false): factors that serve one sweep keep their
    // update rows L21 raw (k_scale_panels touches the pivot rows only) and the forward sweep computes C_I = .. - L21 Y_J after Y_J in a
    // second launch per level.  lazy_factors_: what the factors in memory are.  Not with adjoint sweeps or the forward split-K.
    bool lazy_conversion_ = false, lazy_factors_ = false;
    // Early forward sweep (store = false: a contour step that factorises first, src/feast.jl:64-65).  The forward sweep of a tree level
    // needs that level's factors and row interchanges and the levels' below, nothing above: contour_step_begin hands the right-hand side to
    // factorize() (early_rhs_), which enqueues the right-hand-side gather and the forward sweep of every level below the root on the
    // conversion stream, right behind that level's conversion -- they run in the gaps of the levels above.  early_done_: the sweep that
    // follows gathers the root's rows and takes up there (solve_range part 2; in node groups on their own streams where the late sweep would
    // run so).  A rejected elimination without interchanges starts over with pivoting, early sweep included.  The same launches as the late
    // sweep but for the gather (level by level instead of one pass: a copy), in the same order: bitwise the late sweep's result
    // (test_early_forward_sweep_is_bitwise_the_late_one).  The adjoint sweeps of the adjoint fold go the same way (their forward sweep, with U^H, climbs the tree too); not dual_gen_feast's.
    bool early_sweep_enabled_ = true;      // FEASTHIP_EARLY_SWEEP=0
    const cz* early_rhs_ = nullptr;
    bool early_done_ = false;
    bool early_idrows_ = false;            // the factorisation under way runs without interchanges: its early launches read the identity row tables
    bool fwd_splitk_ = false;         // FEASTHIP_FWD_SPLITK=1: split the long forward L21 products along K when few nodes are left on this GPU.
                                      // Measured at one unit per GPU (C3): the split launches gain 10-40 us each, their reductions cost 50-80 us:
                                      // contour step 4.31 instead of 4.11 ms -- off by default (profiles/r03_per_rank_load_single_gpu.txt)
    bool xcd_balance_ = true;         // FEASTHIP_XCD_BALANCE=0: equal tile counts per XCD in every block-substitution launch (build_plans)
    int slab64_min_wgs_ = 1024;       // launches with at least this many 64-row workgroups (over the local nodes) use 64-row tiles, smaller ones 32-row tiles (FEASTHIP_SLAB64_MIN)
    int short_k_max_p_ = 96;          // levels whose fronts have at most this many pivot columns use 32-row substitution tiles
    int nb256_min_p_ = 129, nb512_min_p_ = 1 << 30;   // fronts with at least this many pivot columns get 256- / 512-column solve super blocks.
                                      // Round 5: no 512-column blocks by default (they were taken from 1024 pivot columns on).  Coupling 256 -> 512 costs the
                                      // factorisation 2.1 ms on C3 and 5 ms on C2 (it ends the conversion: nothing hides it) and saves a contour step 0.03 ms
                                      // (C3) / 0.14 ms (C2): break-even after 70 / 36 iterations of a solver that needs 3 - 10 (FEASTHIP_NB512_MINP=1024 restores it)
                                      // (256 from 129 on, 193 until round 4: a level whose widest front has 132 pivot columns no longer pays an in-block
                                      // step forward and one backward for it: 65 -> 61 launches per sweep pair on C3, contour step -0.2 %, one unit per GPU -1.5 %)
    int nb1024_min_p_ = 1 << 30;      // ... 1024-column ones: built and tested in round 4 (FEASTHIP_NB1024_MINP=2048: the root of C3, p = 3174, gets 4 instead of 7
                                      // blocks, 59 instead of 65 launches per sweep pair) and OFF by default: contour step 16.12 -> 16.09 ms at 16 nodes for + 4.5 ms
                                      // of factorisation (one more coupling level): the root's chain is throughput-bound there, not latency-bound
    int solve_nb(int s) const { int p = S.p(s); return p >= nb1024_min_p_ ? 1024 : p >= nb512_min_p_ ? 512 : p >= nb256_min_p_ ? 256 : 128; }
    std::vector<hipStream_t> aux_streams_;
    std::vector<hipStream_t> up_streams_;  // analyze(): one upload stream per term (helper threads)
    std::vector<hipEvent_t> ev_join_;
    std::vector<hipEvent_t> ev_chunk_;   // contour_step_begin: one per row range of the moment
    void* comm_ = nullptr;               // ncclComm_t
    int comm_rank_ = 0, comm_world_ = 1;
    hipStream_t comm_stream_ = nullptr;
    hipEvent_t ev_comm0_ = nullptr, ev_comm1_ = nullptr;
    std::vector<int> chunk_rows_;
    bool step_open_ = false;
    bool rr_only_ = false;               // feasthip_set_rr_only: only the Rayleigh-Ritz half runs on this handle
    hipEvent_t ev_fork_ = nullptr;
};

}  // namespace feasthip
