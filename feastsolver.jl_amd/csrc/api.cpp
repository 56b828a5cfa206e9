// extern "C" surface of libfeasthip.so (include/feasthip.h).  Nothing here
// throws across the ABI: every entry point catches and maps to a status code.
#include <cstring>
#include <chrono>
#include <complex>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/feasthip.h"
#include "engine.hpp"
#include "smalleig.hpp"

using namespace feasthip;

struct feasthip_handle_s { Engine* e; };
struct feasthip_fact_s { Engine* e; std::mutex mu; };

static std::string g_create_error;
static std::mutex g_mu;

#define GUARD(h) if (!(h) || !(h)->e) return FEASTHIP_EARG; try {
#define ENDGUARD(h) } catch (const std::bad_alloc&) { (h)->e->err = "host out of memory"; return FEASTHIP_EARG; } \
                      catch (const std::exception& ex) { (h)->e->err = ex.what(); return FEASTHIP_EARG; } \
                      catch (...) { (h)->e->err = "unknown C++ exception"; return FEASTHIP_EARG; }

extern "C" {

int feasthip_version(void) { return 100; }

int feasthip_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < cnt; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strstr(p.gcnArchName, "gfx950")) ++ok;
    }
    return ok;
}

int feasthip_create(feasthip_handle_t* h, int device_id) {
    if (!h) return FEASTHIP_EARG;
    *h = nullptr;
    try {
        Engine* e = new Engine(device_id);
        if (e->status) {
            std::lock_guard<std::mutex> lk(g_mu);
            g_create_error = e->err;
            int st = e->status;
            delete e;
            return st;
        }
        feasthip_handle_s* hs = new feasthip_handle_s;
        hs->e = e;
        *h = hs;
        return 0;
    } catch (...) { std::lock_guard<std::mutex> lk(g_mu); g_create_error = "exception in feasthip_create"; return FEASTHIP_EARG; }
}

int feasthip_destroy(feasthip_handle_t h) {
    if (!h) return 0;
    try { delete h->e; delete h; } catch (...) {}
    return 0;
}

int feasthip_trim_memory(void) {
    try { feasthip::trim_device_memory(); } catch (...) {}
    return 0;
}

const char* feasthip_last_error(feasthip_handle_t h) {
    if (!h || !h->e) return g_create_error.c_str();
    return h->e->err.c_str();
}

int feasthip_set_stream(feasthip_handle_t h, void* s) { GUARD(h) return h->e->set_stream(s); ENDGUARD(h) }

int feasthip_set_dense(feasthip_handle_t h, int64_t n, int nterms, const void* const* mats, const int* types, const double* scale) {
    GUARD(h) return h->e->set_dense(n, nterms, mats, types, scale); ENDGUARD(h)
}

int feasthip_set_csc(feasthip_handle_t h, int64_t n, int nterms, const int64_t* const* colptr, const int64_t* const* rowval,
                     const void* const* nzval, const int* types, const double* scale, int index_base) {
    GUARD(h) return h->e->set_csc(n, nterms, colptr, rowval, nzval, types, scale, index_base); ENDGUARD(h)
}

int feasthip_set_contour(feasthip_handle_t h, int K, const double* nodes, const double* weights, int kind, const double* region) {
    GUARD(h) return h->e->set_contour(K, nodes, weights, kind, region); ENDGUARD(h)
}

int feasthip_set_node_shard(feasthip_handle_t h, int first, int stride) { GUARD(h) return h->e->set_shard(first, stride); ENDGUARD(h) }
int feasthip_set_q_buffer(feasthip_handle_t h, void* dptr, size_t bytes) { GUARD(h) return h->e->set_q_buffer(dptr, bytes); ENDGUARD(h) }
int feasthip_analyze(feasthip_handle_t h) { GUARD(h) return h->e->analyze(); ENDGUARD(h) }
int feasthip_factorize(feasthip_handle_t h) { GUARD(h) return h->e->factorize(); ENDGUARD(h) }
int feasthip_set_x(feasthip_handle_t h, const void* X, int64_t n, int m0) { GUARD(h) return h->e->set_x(X, n, m0); ENDGUARD(h) }
int feasthip_get_x(feasthip_handle_t h, void* X) { GUARD(h) return h->e->get_block(0, X); ENDGUARD(h) }
int feasthip_get_block(feasthip_handle_t h, int which, void* out) { GUARD(h) return h->e->get_block(which, out); ENDGUARD(h) }
int feasthip_rr_step(feasthip_handle_t h, double* lambda, double* res, uint8_t* inside) { GUARD(h) return h->e->rr_step(lambda, res, inside); ENDGUARD(h) }
int feasthip_small_eig_herm(feasthip_handle_t h, int m, const void* A, const void* B, double* lambda, void* X, double* status) {
    GUARD(h) return h->e->small_eig_herm(m, A, B, lambda, X, status); ENDGUARD(h)
}
int feasthip_set_rr_only(feasthip_handle_t h, int on) { GUARD(h) return h->e->set_rr_only(on); ENDGUARD(h) }
int feasthip_contour_step(feasthip_handle_t h, int refactor) { GUARD(h) return h->e->contour_step(refactor); ENDGUARD(h) }
int feasthip_contour_step_begin(feasthip_handle_t h, int refactor, int nchunks) { GUARD(h) return h->e->contour_step_begin(refactor, nchunks); ENDGUARD(h) }
int feasthip_moment_chunk_wait(feasthip_handle_t h, int chunk, int64_t* row0, int64_t* rows) { GUARD(h) return h->e->moment_chunk_wait(chunk, row0, rows); ENDGUARD(h) }
int feasthip_contour_step_end(feasthip_handle_t h) { GUARD(h) return h->e->contour_step_end(); ENDGUARD(h) }
int feasthip_nl_contour_step(feasthip_handle_t h, int first) { GUARD(h) return h->e->nl_contour_step(first); ENDGUARD(h) }
int feasthip_nl_rr_step(feasthip_handle_t h, double* lambda, double* res, uint8_t* inside) { GUARD(h) return h->e->nl_rr_step(lambda, res, inside); ENDGUARD(h) }
int feasthip_nl_orthonormalize_x(feasthip_handle_t h) { GUARD(h) return h->e->nl_orthonormalize_x(); ENDGUARD(h) }
int feasthip_nl_finalize_x(feasthip_handle_t h) { GUARD(h) return h->e->nl_finalize_x(); ENDGUARD(h) }
int feasthip_q_device(feasthip_handle_t h, void** p, int64_t* n, int64_t* ldq, int* nb) { GUARD(h) return h->e->q_device(p, n, ldq, nb); ENDGUARD(h) }

int feasthip_feast_run(feasthip_handle_t h, void* X, int64_t n, int m0, int iter, double eps, int store, double* lambda,
                       double* res, uint8_t* inside, int* nit) {
    GUARD(h) return h->e->feast_run(X, n, m0, iter, eps, store, lambda, res, inside, nit); ENDGUARD(h)
}

int feasthip_nlfeast_run(feasthip_handle_t h, void* X, int64_t n, int m0, int iter, double eps, double spurious,
                         double* lambda, double* res, uint8_t* inside, int* nit) {
    GUARD(h) return h->e->nlfeast_run(X, n, m0, iter, eps, spurious, lambda, res, inside, nit); ENDGUARD(h)
}

int feasthip_set_precision(feasthip_handle_t h, int mixed) {
    GUARD(h) return h->e->set_precision(mixed); ENDGUARD(h)
}


int feasthip_set_conjugate_fold(feasthip_handle_t h, int on) {
    GUARD(h) return h->e->set_fold(on); ENDGUARD(h)
}

int feasthip_set_lazy_conversion(feasthip_handle_t h, int on) {
    GUARD(h) return h->e->set_lazy_conversion(on); ENDGUARD(h)
}

int feasthip_set_coef_functions(feasthip_handle_t h, feasthip_coef_fn fn, void* user) {
    GUARD(h) return h->e->set_coef_functions(fn, user); ENDGUARD(h)
}

int feasthip_set_node_coefficients(feasthip_handle_t h, const double* coef, int K, int nterms) {
    GUARD(h) return h->e->set_node_coefficients(coef, K, nterms); ENDGUARD(h)
}

int feasthip_nl_beyn_step(feasthip_handle_t h, double* lambda, uint8_t* inside) {
    GUARD(h) return h->e->nl_rr_step(lambda, nullptr, inside, 1); ENDGUARD(h)
}

int feasthip_nl_set_r(feasthip_handle_t h, const void* R) {
    GUARD(h) return h->e->nl_set_r(R); ENDGUARD(h)
}

int feasthip_nlfeast_moments_run(feasthip_handle_t h, void* X, int64_t n, int m0, int moments, int iter, double eps,
                                 double spurious, double* lambda, double* res, uint8_t* inside, void* Y, int* nit) {
    GUARD(h) return h->e->nlfeast_moments_run(X, n, m0, moments, iter, eps, spurious, lambda, res, inside, Y, nit); ENDGUARD(h)
}

int feasthip_nlm_begin(feasthip_handle_t h, const void* X, int64_t n, int m0, int moments) {
    GUARD(h) return h->e->nlm_begin(X, n, m0, moments); ENDGUARD(h)
}
int feasthip_nlm_contour_step(feasthip_handle_t h, int first) {
    GUARD(h) return h->e->nlm_step_contour(first); ENDGUARD(h)
}
int feasthip_nlm_candidates(feasthip_handle_t h, double* lambda, void* Y) {
    GUARD(h) if (!lambda) return FEASTHIP_EARG; return h->e->nlm_step_candidates(lambda, Y); ENDGUARD(h)
}
int feasthip_nlm_finish(feasthip_handle_t h, const void* R, const double* res_in, double* lambda, double* res, uint8_t* inside) {
    GUARD(h) if (!lambda || !res) return FEASTHIP_EARG; return h->e->nlm_step_finish(R, res_in, lambda, res, inside); ENDGUARD(h)
}
int feasthip_nlm_get_y(feasthip_handle_t h, void* Y) {
    GUARD(h) if (!Y) return FEASTHIP_EARG; return h->e->nlm_get_y(Y); ENDGUARD(h)
}

int feasthip_dual_feast_run(feasthip_handle_t h, void* Xr, void* Xl, int64_t n, int m0, int iter, double eps, int store,
                            double* lambda, double* resr, double* resl, uint8_t* inside, int* nit) {
    GUARD(h) return h->e->dual_run(Xr, Xl, n, m0, iter, eps, store, lambda, resr, resl, inside, nit); ENDGUARD(h)
}

int feasthip_estimate_count(feasthip_handle_t h, const void* X, int64_t n, int m0, double* est) {
    GUARD(h) return h->e->estimate_count(X, n, m0, est); ENDGUARD(h)
}

int feasthip_comm_unique_id(void* id128) {
    if (!id128) return FEASTHIP_EARG;
    try {
        std::string e;
        int r = Engine::comm_unique_id(id128, e);
        if (r) { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e; }
        return r;
    } catch (...) { return FEASTHIP_EARG; }
}
int feasthip_comm_init(feasthip_handle_t h, const void* id128, int rank, int world) { GUARD(h) return h->e->comm_init(id128, rank, world); ENDGUARD(h) }
int feasthip_comm_destroy(feasthip_handle_t h) { GUARD(h) return h->e->comm_destroy(); ENDGUARD(h) }

int feasthip_get_stats(feasthip_handle_t h, feasthip_stats_t* o) {
    GUARD(h)
    if (!o) return FEASTHIP_EARG;
    Engine* e = h->e;
    std::memset(o, 0, sizeof(*o));
    o->n = e->n; o->nsuper = e->S.nsuper; o->nnz_l = e->S.nnz_l; o->nnz_u = e->S.nnz_u; o->fact_elems = e->S.fact_size;
    o->max_front = e->S.max_front; o->max_p = e->S.max_p; o->depth = e->S.maxdepth + 1;
    o->flops_factor = e->S.flops_factor; o->analyze_seconds = e->analyze_seconds; o->factor_ms = e->factor_ms;
    o->solve_ms = e->solve_ms; o->rr_ms = e->rr_ms; o->alloc_seconds = e->alloc_seconds;
    o->solve_bytes = e->ldw ? e->solve_bytes() : 0; o->solve_flops = e->ldw ? e->solve_flops() : 0;
    o->perturbed_pivots = e->perturbed; o->local_nodes = e->nodes_local(); o->m0 = e->m0; o->ldw = e->ldw; o->dense = e->dense ? 1 : 0;
    o->factor_units = e->Kloc; o->folded = e->folded() ? 1 : e->adj_folded() ? 2 : 0; o->solve_bytes_streamed = e->ldw ? e->solve_bytes_streamed() : 0;
    o->proj_ms = e->proj_ms;
    o->early_sweep_levels = e->early_levels_;
    o->reserved0 = 0;
    o->symmetric = e->nopiv_used ? 1 : e->nopiv_gen_used ? 2 : 0; o->growth = e->growth;
    o->rr_device_sweeps = e->rr_device_sweeps;
    return 0;
    ENDGUARD(h)
}

int feasthip_profile_solve(feasthip_handle_t h, double* out /* 8 rows x 4: ms, bytes, flops, launches */) {
    GUARD(h)
    if (!out) return FEASTHIP_EARG;
    ProfileRow rows[8];
    int r = h->e->profile_solve(rows);
    if (r) return r;
    for (int i = 0; i < 8; ++i) { out[4 * i] = rows[i].ms; out[4 * i + 1] = rows[i].bytes; out[4 * i + 2] = rows[i].flops; out[4 * i + 3] = rows[i].launches; }
    return 0;
    ENDGUARD(h)
}

int feasthip_profile_detail(feasthip_handle_t h, double* out, int cap) {
    GUARD(h)
    int n = (int)(h->e->prof_detail.size() / 6);
    if (out) for (int i = 0; i < n && i < cap; ++i) for (int j = 0; j < 6; ++j) out[6 * i + j] = h->e->prof_detail[6 * i + j];
    return n;
    ENDGUARD(h)
}

int feasthip_level_table(feasthip_handle_t h, double* out, int cap) {
    GUARD(h)
    const Symbolic& S = h->e->S;
    int nl = (int)S.levels.size();
    if (out) for (int d = 0; d < nl && d < cap; ++d) {
        double fronts = 0, sp = 0, su = 0, nz = 0, pmax = 0;
        for (int s : S.levels[d]) { double p = S.p(s), u = S.u(s); fronts += 1; sp += p; su += u; nz += p * p + 2.0 * u * p; pmax = std::max(pmax, p); }
        out[5 * d] = fronts; out[5 * d + 1] = sp; out[5 * d + 2] = su; out[5 * d + 3] = nz; out[5 * d + 4] = pmax;
    }
    return nl;
    ENDGUARD(h)
}

// ---- fine-grained seam -----------------------------------------------------
static int make_fact(Engine* e, feasthip_fact_t* F) {
    double node[2] = {0.0, 0.0}, w[2] = {1.0, 0.0}, reg[4] = {0, 0, 1, 0};
    int r = e->set_contour(1, node, w, FEASTHIP_REGION_DISC, reg);
    if (r == 0) r = e->analyze();
    if (r == 0) r = e->factorize();
    if (r < 0) { { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return r; }
    feasthip_fact_s* f = new feasthip_fact_s;
    f->e = e;
    *F = f;
    return r;
}

int feasthip_lu_dense(int device_id, const void* C, int64_t n, feasthip_fact_t* F) {
    if (!C || !F || n <= 0) return FEASTHIP_EARG;
    *F = nullptr;
    try {
        auto t0 = std::chrono::steady_clock::now();
        Engine* e = new Engine(device_id);
        if (e->status) { int st = e->status; { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return st; }
        const void* mats[1] = {C};
        int types[1] = {FEASTHIP_C128};
        auto t1 = std::chrono::steady_clock::now();
        int r = e->set_dense(n, 1, mats, types, nullptr);
        if (r) { { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return r; }
        auto t2 = std::chrono::steady_clock::now();
        r = make_fact(e, F);
        if (getenv("FEASTHIP_SEAM_TRACE")) {
            auto t3 = std::chrono::steady_clock::now();
            fprintf(stderr, "[feasthip lu_dense n=%lld] handle %.3f ms, matrix hand-over %.3f ms, analysis + factorisation %.3f ms (analysis %.3f, factorisation kernels %.3f)\n", (long long)n,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                    std::chrono::duration<double, std::milli>(t3 - t2).count(), r >= 0 ? e->analyze_seconds * 1e3 : 0.0, r >= 0 ? e->factor_ms : 0.0);
        }
        return r;
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_lu_dense_c64(int device_id, const void* C, int64_t n, feasthip_fact_t* F) {
    if (!C || !F || n <= 0) return FEASTHIP_EARG;
    *F = nullptr;
    try {
        Engine* e = new Engine(device_id);
        if (e->status) { int st = e->status; { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return st; }
        const void* mats[1] = {C};
        int types[1] = {FEASTHIP_C64};       // widened on the device
        int r = e->set_dense(n, 1, mats, types, nullptr);
        if (r == 0) r = e->set_precision(1);
        if (r) { { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return r; }
        return make_fact(e, F);
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_lu_csc(int device_id, int64_t n, const int64_t* colptr, const int64_t* rowval, const void* nzval, int type,
                    int index_base, feasthip_fact_t* F) {
    if (!colptr || !rowval || !nzval || !F || n <= 0) return FEASTHIP_EARG;
    *F = nullptr;
    try {
        Engine* e = new Engine(device_id);
        if (e->status) { int st = e->status; { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return st; }
        std::vector<std::complex<double>> wide;
        if (type == FEASTHIP_C64) {          // ComplexF32 shifted matrix: widen the values, solve in complex64
            int64_t nnz = colptr[n] - index_base;
            if (nnz < 0) { delete e; return FEASTHIP_EARG; }
            const std::complex<float>* src = static_cast<const std::complex<float>*>(nzval);
            wide.resize((size_t)nnz);
            for (int64_t i = 0; i < nnz; ++i) wide[i] = std::complex<double>(src[i].real(), src[i].imag());
            nzval = wide.data();
            type = FEASTHIP_C128;
            e->set_precision(1);
        }
        const int64_t* cps[1] = {colptr};
        const int64_t* rvs[1] = {rowval};
        const void* vs[1] = {nzval};
        int types[1] = {type};
        int r = e->set_csc(n, 1, cps, rvs, vs, types, nullptr, index_base);
        if (r) { { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e->err; } delete e; return r; }
        return make_fact(e, F);
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_ldiv(feasthip_fact_t F, void* Y, int64_t ldy, const void* R, int64_t ldr, int m0) {
    if (!F || !F->e) return FEASTHIP_EARG;
    try {
        std::lock_guard<std::mutex> lk(F->mu);
        return F->e->ldiv(Y, ldy, R, ldr, m0);
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_ldiv_adjoint(feasthip_fact_t F, void* Y, int64_t ldy, const void* R, int64_t ldr, int m0) {
    if (!F || !F->e) return FEASTHIP_EARG;
    try {
        std::lock_guard<std::mutex> lk(F->mu);
        return F->e->ldiv(Y, ldy, R, ldr, m0, 1);
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_fact_info(feasthip_fact_t F, int* symmetric, double* growth, int64_t* perturbed) {
    if (!F || !F->e) return FEASTHIP_EARG;
    if (symmetric) *symmetric = F->e->nopiv_used ? 1 : F->e->nopiv_gen_used ? 2 : 0;
    if (growth) *growth = F->e->growth;
    if (perturbed) *perturbed = F->e->perturbed;
    return 0;
}

int feasthip_fact_free(feasthip_fact_t F) {
    if (!F) return 0;
    try { delete F->e; delete F; } catch (...) {}
    return 0;
}

// ---- host-only helpers -----------------------------------------------------
static int build_pattern(int64_t n, const int64_t* colptr, const int64_t* rowval, int base, std::vector<int64_t>& cp, std::vector<int64_t>& ri) {
    if (n <= 0 || !colptr || !rowval) return FEASTHIP_EARG;
    cp.resize(n + 1);
    for (int64_t j = 0; j <= n; ++j) cp[j] = colptr[j] - base;
    if (cp[0] != 0) return FEASTHIP_EARG;
    ri.resize(cp[n]);
    for (int64_t e = 0; e < cp[n]; ++e) ri[e] = rowval[e] - base;
    return 0;
}

int feasthip_symbolic_probe(int64_t n, const int64_t* colptr, const int64_t* rowval, int index_base, int ordering,
                            int64_t* stats, int32_t* perm_out, int32_t* sn_of_out, int32_t* fronts_out) {
    try {
        std::vector<int64_t> cp, ri;
        int r = build_pattern(n, colptr, rowval, index_base, cp, ri);
        if (r) return r;
        Symbolic S;
        SymbolicOptions opt;
        opt.ordering = ordering;
        std::string e = analyze_pattern((int)n, cp.data(), ri.data(), opt, S);
        if (!e.empty()) { std::lock_guard<std::mutex> lk(g_mu); g_create_error = e; return FEASTHIP_EARG; }
        if (stats) {
            stats[0] = S.nsuper; stats[1] = S.nnz_l; stats[2] = S.nnz_u; stats[3] = S.fact_size; stats[4] = S.max_front;
            stats[5] = S.max_p; stats[6] = S.maxdepth + 1; stats[7] = (int64_t)S.flops_factor;
        }
        if (perm_out) for (int64_t i = 0; i < n; ++i) perm_out[i] = S.perm[i];
        if (sn_of_out) for (int64_t i = 0; i < n; ++i) sn_of_out[i] = S.sn_of[i];
        if (fronts_out) for (int s = 0; s < S.nsuper; ++s) { fronts_out[4 * s] = S.p(s); fronts_out[4 * s + 1] = S.u(s); fronts_out[4 * s + 2] = S.depth[s]; fronts_out[4 * s + 3] = S.sn_parent[s]; }
        return 0;
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_symbolic_contains(int64_t n, const int64_t* colptr, const int64_t* rowval, int index_base, int ordering,
                               const int32_t* ii, const int32_t* jj, int64_t count) {
    try {
        std::vector<int64_t> cp, ri;
        int r = build_pattern(n, colptr, rowval, index_base, cp, ri);
        if (r) return r;
        Symbolic S;
        SymbolicOptions opt;
        opt.ordering = ordering;
        std::string e = analyze_pattern((int)n, cp.data(), ri.data(), opt, S);
        if (!e.empty()) return FEASTHIP_EARG;
        int64_t bad = 0;
        for (int64_t t = 0; t < count; ++t) {
            if (ii[t] < 0 || ii[t] >= n || jj[t] < 0 || jj[t] >= n) return FEASTHIP_EARG;
            if (entry_dest(S, ii[t], jj[t]) < 0) ++bad;
        }
        return bad > 0x7fffffff ? 0x7fffffff : (int)bad;
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_shard_plan(int K, const double* nodes, int fold_allowed, int first, int stride, int32_t* local, int32_t* partner,
                        int* count, int* folded) {
    if (K <= 0 || !nodes || first < 0 || stride < 1 || !local || !partner || !count || !folded) return FEASTHIP_EARG;
    try {
        std::vector<zc> z(K);
        for (int k = 0; k < K; ++k) z[k] = zc(nodes[2 * k], nodes[2 * k + 1]);
        std::vector<int> loc, par;
        *folded = plan_node_shard(z, first, stride, fold_allowed != 0, loc, par) ? 1 : 0;
        *count = (int)loc.size();
        for (size_t i = 0; i < loc.size(); ++i) { local[i] = loc[i]; partner[i] = par[i]; }
        return 0;
    } catch (...) { return FEASTHIP_EARG; }
}

int feasthip_small_eig(int m, const double* A, const double* B, double* lambda, double* V) {
    if (m <= 0 || !A || !lambda || !V) return FEASTHIP_EARG;
    try {
        std::vector<zc> a((size_t)m * m), b, lam(m), v((size_t)m * m);
        std::memcpy(a.data(), A, sizeof(zc) * (size_t)m * m);
        int info = 1;
        if (B) { b.resize((size_t)m * m); std::memcpy(b.data(), B, sizeof(zc) * (size_t)m * m); }
        bool herm = hermitian_defect(m, a.data()) <= 1e-13 && (!B || hermitian_defect(m, b.data()) <= 1e-13);
        if (herm) {   // same dispatch as Engine::rr_step
            std::vector<zc> a2(a), b2(b);
            info = eig_hermitian(m, a2.data(), B ? b2.data() : nullptr, lam.data(), v.data());
        }
        if (info) {
            if (B) info = eig_generalized(m, a.data(), b.data(), lam.data(), v.data());
            else info = eig_general(m, a.data(), lam.data(), v.data());
        }
        std::memcpy(lambda, lam.data(), sizeof(zc) * m);
        std::memcpy(V, v.data(), sizeof(zc) * (size_t)m * m);
        return info ? FEASTHIP_ENUM : 0;
    } catch (...) { return FEASTHIP_EARG; }
}

}  // extern "C"
