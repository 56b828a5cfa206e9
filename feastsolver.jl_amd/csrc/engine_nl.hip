// Nonlinear paths: nlfeast (RII form), block-Hankel moments (Beyn), two-sided gen_feast (src/nlfeast.jl:2-84, 120-260; src/feast.jl:164-260).
#include "engine_impl.hpp"

namespace feasthip {

int Engine::nl_orthonormalize_x() {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "nl_orthonormalize_x: call set_x first");
    HIPCHK(hipSetDevice(device_));
    return orthonormalize(d_X_.p, d_T1_.p, nullptr);       // src/nlfeast.jl:12-13
}

int Engine::nl_contour_step(int first) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "nl_contour_step: call set_x first");
    HIPCHK(hipSetDevice(device_));
    if (!factored_) { int r = factorize(); if (r < 0) return r; }
    auto ef = [&](int k, int j, int q) { zc e = first ? weights_[k] : weights_[k] / (nodes_[k] - lambda_[j]); return q ? e * nodes_[k] : e; };   // src/nlfeast.jl:42 / :47, :58
    std::vector<zc> e0 = node_table([&](int k, int j, int) { return ef(k, j, 0); }), e1 = node_table([&](int k, int j, int) { return ef(k, j, 1); });
    HIPCHK(hipMemcpyAsync(d_e0_.p, e0.data(), e0.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(d_e1_.p, e1.data(), e1.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int r = solve_all(first ? d_X_.p : d_R_.p);               // :41 / :49
    if (r) return r;
    accumulate_solutions(Kloc, d_e0_.p, d_e1_.p, first ? 0 : 1, Qp(), Qp() + (size_t)n * ldw);
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    solve_ms = ms;
    return 0;
}

static double frob_inner_sparse(const SparseTerm& A, const SparseTerm& B, int n, zc& out) {
    zc acc = 0;
    for (int j = 0; j < n; ++j) {
        int64_t a = A.colptr[j], ae = A.colptr[j + 1], b = B.colptr[j], be = B.colptr[j + 1];
        while (a < ae && b < be) {
            if (A.rowval[a] < B.rowval[b]) ++a;
            else if (A.rowval[a] > B.rowval[b]) ++b;
            else { acc += std::conj(A.val[a]) * B.val[b]; ++a; ++b; }
        }
    }
    out = acc;
    return 0;
}

int Engine::build_frob_table() {
    if (!frob_.empty()) return 0;
    {
        frob_.assign(nterms, std::vector<double>(2 * nterms, 0.0));
        for (int s = 0; s < nterms; ++s)
            for (int t = 0; t < nterms; ++t) {
                zc ip = 0;
                if (dense) {
                    // dense inner products: download once (n <= a few thousand on this path)
                    std::vector<zc> a, b;
                    auto fetch = [&](int q, std::vector<zc>& o) -> int {
                        o.assign((size_t)n * n, zc(0.0));
                        if (terms_[q].identity) { for (int i = 0; i < n; ++i) o[i + (size_t)i * n] = 1.0; return 0; }
                        if (hipMemcpy(o.data(), terms_[q].d_dense.p, (size_t)n * n * sizeof(zc), hipMemcpyDeviceToHost) != hipSuccess) return 1;
                        return 0;
                    };
                    if (fetch(s, a) || fetch(t, b)) return fail(FEASTHIP_EHIP, "nl_rr_step: download failed");
                    for (size_t e = 0; e < (size_t)n * n; ++e) ip += std::conj(a[e]) * b[e];
                } else {
                    frob_inner_sparse(terms_[s], terms_[t], n, ip);
                }
                ip *= std::conj(terms_[s].scale) * terms_[t].scale;
                frob_[s][2 * t] = ip.real(); frob_[s][2 * t + 1] = ip.imag();
            }
    }
    return 0;
}

double Engine::poly_frob_norm(zc lam) {
    std::vector<zc> f;
    eval_coefs(&lam, 1, f);
    zc acc = 0;
    for (int s = 0; s < nterms; ++s)
        for (int t = 0; t < nterms; ++t) acc += std::conj(f[s]) * f[t] * zc(frob_[s][2 * t], frob_[s][2 * t + 1]);
    return std::sqrt(std::max(acc.real(), 0.0));
}

// R[:, j] = T(lam_j) X[:, j] = sum_t f_t(lam_j) (scale_t M_t X)[:, j]  (src/utils.jl:107) for a block with leading dimension ld
int Engine::residual_block(const cz* X, cz* MX, cz* R, const std::vector<zc>& lam, int mcols, int ld, cz* d_coeftab) {
    int64_t blk = (int64_t)n * ld;
    for (int t = 0; t < nterms; ++t) apply_term_ld(t, X, MX + (size_t)t * blk, 1.0, false, ld);
    std::vector<zc> f, tab((size_t)nterms * ld, zc(0.0));
    eval_coefs(lam.data(), mcols, f);
    for (int j = 0; j < mcols; ++j) for (int t = 0; t < nterms; ++t) tab[(size_t)t * ld + j] = f[(size_t)j * nterms + t];
    HIPCHK(hipMemcpyAsync(d_coeftab, tab.data(), tab.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_coef_combine(stream_, n, ld, MX, blk, nterms, d_coeftab, R);
    HIPCHK(hipStreamSynchronize(stream_));      // tab is a stack-owned host buffer
    return 0;
}

int Engine::nl_rr_step(double* lambda, double* res, uint8_t* inside, int external_residual) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "nl_rr_step: call set_x first");
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int m = m0;
    cz* Q0 = Qp();
    cz* Q1 = Qp() + (size_t)n * ldw;
    // beyn_svd_step!  src/utils.jl:69-77: thin SVD of Q0 through the orthonormal basis Qo (Q0 = Qo Rr)
    std::vector<zc> Rr;
    int r = orthonormalize(Q0, d_T1_.p, &Rr);
    if (r) return r;
    std::vector<zc> Ur((size_t)m * m), Vr((size_t)m * m), G1((size_t)m * m), A((size_t)m * m), T((size_t)m * m), lam(m), vec((size_t)m * m);
    std::vector<double> sv(m);
    jacobi_svd(m, Rr.data(), Ur.data(), sv.data(), Vr.data());          // Rr = Ur S Vr^H  => U = Qo Ur
    cz* dG = d_small_.p;
    cz* dM = d_small_.p + (size_t)2 * ldw * ldw;
    launch_gram(stream_, n, m, ldw, Q0, Q1, d_gram_scratch_.p, dG);     // Qo^H Q1
    HIPCHK(hipMemcpyAsync(G1.data(), dG, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    // A = Ur^H G1 V S^-1      (mul!(A,S.U',Q1); mul!(B,A,S.V); mul!(A,B,Diagonal(1 ./ S.S)))
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { zc s = 0; for (int k = 0; k < m; ++k) s += std::conj(Ur[k + (size_t)i * m]) * G1[k + (size_t)j * m]; T[i + (size_t)j * m] = s; }
    matmul(m, T.data(), Vr.data(), A.data());
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)j * m] *= 1.0 / sv[j];
    int info = eig_general(m, A.data(), lam.data(), vec.data());
    if (info) return fail(FEASTHIP_ENUM, "nl_rr_step: reduced eigenproblem failed to converge");
    matmul(m, Ur.data(), vec.data(), T.data());                           // X = U vecs = Qo (Ur vecs)
    HIPCHK(hipMemcpyAsync(dM, T.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_ts_gemm(stream_, n, m, ldw, Q0, dM, d_X_.p);
    // update_R!(X,R,lam,T)  src/utils.jl:104-109: normalise, R[:,i] = T(lam_i) x_i = sum_t lam_i^t M_t x_i
    std::vector<double> nr;
    r = col_norms(d_X_.p, nr);
    if (r) return r;
    std::vector<zc> sc(ldw, zc(0.0)), lamp(ldw, zc(0.0));
    for (int j = 0; j < m; ++j) { sc[j] = nr[j] > 0 ? 1.0 / nr[j] : 1.0; lamp[j] = lam[j]; }
    HIPCHK(hipMemcpyAsync(d_lam_.p, sc.data(), ldw * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_scale_cols(stream_, n, ldw, d_X_.p, d_lam_.p);
    HIPCHK(hipStreamSynchronize(stream_));
    if (external_residual) {     // generic closure T: the caller forms R[:,i] = T(lam_i) x_i itself and hands it back (feasthip_nl_set_r)
        lambda_ = lam;
        for (int j = 0; j < m; ++j) { lambda[2 * j] = lam[j].real(); lambda[2 * j + 1] = lam[j].imag(); if (res) res[j] = 0.0; }
        if (inside) inside_test(lam.data(), inside);
        return 0;
    }
    if (!d_coeftab_.alloc((size_t)nterms * ldw)) return fail(FEASTHIP_EHIP, "out of device memory");
    r = residual_block(d_X_.p, d_MX_.p, d_R_.p, lam, m, ldw, d_coeftab_.p);
    if (r) return r;
    r = col_norms(d_R_.p, nr);
    if (r) return r;
    // residuals(R,lam,T): ||R_i|| / ||T(lam_i)||_F   src/utils.jl:151-157, Frobenius norm from the
    // table <M_s, M_t>_F (computed once per problem)
    r = build_frob_table();
    if (r) return r;
    for (int j = 0; j < m; ++j) {
        res[j] = nr[j] / poly_frob_norm(lam[j]);
        lambda[2 * j] = lam[j].real(); lambda[2 * j + 1] = lam[j].imag();
    }
    lambda_ = lam;
    inside_test(lam.data(), inside);
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    rr_ms = ms;
    return 0;
}

int Engine::nl_set_r(const void* R) {
    if (!have_x_ || !R) return fail(FEASTHIP_ESTATE, "nl_set_r: call set_x first");
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipMemcpyAsync(d_io_.p, R, (size_t)n * m0 * sizeof(cz), hipMemcpyHostToDevice, stream_));
    launch_cm_to_rm(stream_, n, m0, ldw, d_io_.p, n, d_R_.p);
    HIPCHK(hipStreamSynchronize(stream_));
    return 0;
}

int Engine::nl_finalize_x() {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "nl_finalize_x: call set_x first");
    HIPCHK(hipSetDevice(device_));
    std::vector<double> nr;
    int r = col_norms(d_X_.p, nr);
    if (r) return r;
    std::vector<zc> sc(ldw, zc(0.0));
    for (int j = 0; j < m0; ++j) sc[j] = nr[j] > 0 ? 1.0 / nr[j] : 1.0;
    HIPCHK(hipMemcpyAsync(d_lam_.p, sc.data(), ldw * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_scale_cols(stream_, n, ldw, d_X_.p, d_lam_.p);
    HIPCHK(hipStreamSynchronize(stream_));
    return 0;
}

// ---------------------------------------------------------------------------
// nlfeast_moments!  (src/nlfeast.jl:173-318): 2*moments contour moments, block-Hankel Beyn step
// ---------------------------------------------------------------------------
int Engine::nlm_alloc(int moments) {
    mm_ = moments;
    int M = mm_ * m0;
    ldM_ = ((M + 15) / 16) * 16;
    size_t blk = (size_t)n * ldw, hrows = (size_t)mm_ * n, tall = hrows * ldM_, wide = (size_t)n * ldM_;
    bool ok = d_Qm_.alloc(2 * (size_t)mm_ * blk) && d_H0_.alloc(tall) && d_H1_.alloc(tall) && d_Ht_.alloc(tall) && d_Ym_.alloc(wide) &&
              d_Rm_.alloc(wide) && d_MXm_.alloc(wide * std::max(2, nterms)) && d_em_.alloc(2 * (size_t)mm_ * ninst() * solve_width()) &&
              d_smallM_.alloc((size_t)8 * ldM_ * ldM_) && d_lamM_.alloc(ldM_) && d_gramM_.alloc(gram_scratch_elems((int)hrows, ldM_)) &&
              d_nrmM_.alloc(ldM_) && d_nrmMs_.alloc(colnorm_scratch_elems((int)hrows, ldM_)) && d_colidx_.alloc(ldM_);
    if (!ok) return fail(FEASTHIP_EHIP, "out of device memory for the moment buffers");
    return 0;
}

// Q_j = sum_k w_k z_k^j T(z_k)^-1 X (first pass, :193-212) or sum_k w_k z_k^j / (z_k - lam) .* (X - T(z_k)^-1 R) (:246-268)
int Engine::nlm_contour_step(int first) {
    HIPCHK(hipSetDevice(device_));
    const int L = 2 * mm_;
    std::vector<zc> e = node_table([&](int k, int j, int q) {
        zc v = first ? weights_[k] : weights_[k] / (nodes_[k] - lambda_[j]);
        for (int t = 0; t < q; ++t) v *= nodes_[k];
        return v; }, L);
    HIPCHK(hipMemcpyAsync(d_em_.p, e.data(), e.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int r = solve_all(first ? d_X_.p : d_R_.p);
    if (r) return r;
    size_t blk = (size_t)n * ldw, eb = (size_t)ninst() * solve_width();
    for (int q = 0; q < L; q += 2)      // two moments per pass over the K solution blocks
        accumulate_solutions(Kloc, d_em_.p + q * eb, d_em_.p + (q + 1) * eb, first ? 0 : 1, d_Qm_.p + q * blk, d_Qm_.p + (q + 1) * blk);
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    solve_ms = ms;
    return 0;
}

// Hankel matrices, thin SVD of the first through its orthonormal basis, reduced eigenproblem, candidate
// vectors Y = U[1:N,:] vecs, update_R_moments! (src/utils.jl:118-134).  On return lam/res (moments*m0) are
// sorted by residual, d_Ym_/d_Rm_ hold the sorted candidates, d_X_/d_R_ the m0 best.
extern thread_local std::string g_ortho_diag;
// Part 1: candidates.  Hankel matrices, rank-revealing basis, reduced eigenproblem; on return d_Ym_ holds the
// column-normalised candidate vectors, lam their eigenvalues (NaN beyond the numerical rank nlm_rank_).
int Engine::nlm_candidates(std::vector<zc>& lam) {
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    const int M = mm_ * m0, ld = ldM_, hrows = mm_ * n;
    Tall th{hrows, M, ld, d_smallM_.p, d_lamM_.p, d_gramM_.p, d_nrmM_.p, d_nrmMs_.p};
    Tall tn{n, M, ld, d_smallM_.p, d_lamM_.p, d_gramM_.p, d_nrmM_.p, d_nrmMs_.p};
    launch_hankel(stream_, n, m0, ldw, mm_, ld, 0, d_Qm_.p, d_H0_.p);       // :213-216
    launch_hankel(stream_, n, m0, ldw, mm_, ld, 1, d_Qm_.p, d_H1_.p);
    std::vector<zc> Rr;
    std::vector<uint8_t> keepc;
    int r = orthonormalize_rr(th, d_H0_.p, d_Ht_.p, Rr, keepc);               // svd!(Q0) :218 through Q0 = Qo Rr, rank revealing
    if (r) return r;
    int rk = 0;
    for (int j = 0; j < M; ++j) rk += keepc[j] ? 1 : 0;
    if (rk == 0) {
        // diagnostics of a failure that must not happen with a regular T(z_k): what the moments, the subspace and the residual block hold
        auto stats = [&](const cz* dptr, size_t count, const char* name, std::string& out) {
            std::vector<zc> hbuf(count);
            if (hipMemcpy(hbuf.data(), dptr, count * sizeof(zc), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
            size_t bad = 0; double mx = 0;
            for (const zc& v : hbuf) { if (!std::isfinite(v.real()) || !std::isfinite(v.imag())) ++bad; else mx = std::max(mx, std::abs(v)); }
            char line[160];
            snprintf(line, sizeof line, " %s: max |.| %.3e, %zu non-finite of %zu;", name, mx, bad, count);
            out += line;
        };
        std::string d;
        stats(d_Qm_.p, (size_t)2 * mm_ * n * ldw, "moments", d);
        stats(d_X_.p, (size_t)n * ldw, "X", d);
        stats(d_R_.p, (size_t)n * ldw, "R", d);
        stats(d_em_.p, (size_t)2 * mm_ * ninst() * solve_width(), "weights", d);
        stats(d_W_.p, (size_t)ninst() * n * solve_width(), "solutions", d);
        stats(d_H0_.p, (size_t)hrows * ld, "basis", d);
        return fail(FEASTHIP_ENUM, "nlfeast_moments: the moment matrix is numerically zero (" + d + " orthonormalisation:" + g_ortho_diag + ")");
    }
    std::vector<zc> Ur((size_t)M * M), Vr((size_t)M * M), G1((size_t)M * M), T((size_t)M * M), vec((size_t)M * M, zc(0.0));
    std::vector<double> sv(M);
    jacobi_svd(M, Rr.data(), Ur.data(), sv.data(), Vr.data());               // singular values descending: the first rk are the numerical range
    cz* dG = d_smallM_.p;
    cz* dM = d_smallM_.p + (size_t)2 * ld * ld;
    launch_gram(stream_, hrows, M, ld, d_H0_.p, d_H1_.p, d_gramM_.p, dG);   // Qo^H Q1
    HIPCHK(hipMemcpyAsync(G1.data(), dG, (size_t)M * M * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    // A = U^H Q1 V S^-1 restricted to the rk significant singular triplets (:219-221); the reference divides by the
    // rounding-level singular values as well, which only adds candidates made of noise (they sort last by residual)
    std::vector<zc> Ar((size_t)rk * rk), T1((size_t)M * rk, zc(0.0)), lr(rk), vr((size_t)rk * rk);
    for (int j = 0; j < rk; ++j) for (int i = 0; i < M; ++i) { zc s = 0; for (int k = 0; k < M; ++k) s += G1[i + (size_t)k * M] * Vr[k + (size_t)j * M]; T1[i + (size_t)j * M] = s / sv[j]; }
    for (int j = 0; j < rk; ++j) for (int i = 0; i < rk; ++i) { zc s = 0; for (int k = 0; k < M; ++k) s += std::conj(Ur[k + (size_t)i * M]) * T1[k + (size_t)j * M]; Ar[i + (size_t)j * rk] = s; }
    int info = eig_general(rk, Ar.data(), lr.data(), vr.data());            // :222
    if (info) return fail(FEASTHIP_ENUM, "nlfeast_moments: reduced eigenproblem failed to converge");
    lam.assign(M, zc(std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()));
    for (int j = 0; j < rk; ++j) { lam[j] = lr[j]; for (int i = 0; i < rk; ++i) vec[i + (size_t)j * M] = vr[i + (size_t)j * rk]; }
    matmul(M, Ur.data(), vec.data(), T.data());                             // Y = U[1:N,:] vecs = Qo[1:N,:] (Ur vecs)  :223
    HIPCHK(hipMemcpyAsync(dM, T.data(), (size_t)M * M * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_ts_gemm(stream_, n, M, ld, d_H0_.p, dM, d_Ym_.p);
    // update_R_moments!: normalise, R[:,i] = T(lam_i) y_i, relative residuals, sort by residual
    std::vector<double> nr;
    r = col_norms(tn, d_Ym_.p, nr);
    if (r) return r;
    std::vector<zc> sc(ld, zc(0.0)), lamp(ld, zc(0.0));
    for (int j = 0; j < M; ++j) { sc[j] = (nr[j] > 0 && std::isfinite(nr[j])) ? 1.0 / nr[j] : 1.0; lamp[j] = j < rk ? lam[j] : zc(0.0); }
    HIPCHK(hipMemcpyAsync(d_lamM_.p, sc.data(), ld * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_scale_cols(stream_, n, ld, d_Ym_.p, d_lamM_.p);
    HIPCHK(hipStreamSynchronize(stream_));
    nlm_rank_ = rk;
    return 0;
}

// Part 2: residuals (on the device for polynomial / split-form T, or handed in by the caller for an opaque closure:
// ext_R n x M column-major, ext_res M), then update_R_moments!'s sort and the selection of the m0 best.
int Engine::nlm_finish(std::vector<zc>& lam, std::vector<double>& res, const void* ext_R, const double* ext_res) {
    const int M = mm_ * m0, ld = ldM_, rk = nlm_rank_;
    Tall tn{n, M, ld, d_smallM_.p, d_lamM_.p, d_gramM_.p, d_nrmM_.p, d_nrmMs_.p};
    std::vector<double> nr;
    int r = 0;
    int64_t wide = (int64_t)n * ld;
    std::vector<double> rs(M);
    if (ext_R) {
        DevBuf<cz> cm;
        if (!cm.alloc((size_t)n * M)) return fail(FEASTHIP_EHIP, "out of device memory");
        HIPCHK(hipMemcpyAsync(cm.p, ext_R, (size_t)n * M * sizeof(cz), hipMemcpyHostToDevice, stream_));
        launch_cm_to_rm(stream_, n, M, ld, cm.p, n, d_Rm_.p);
        HIPCHK(hipStreamSynchronize(stream_));
        for (int j = 0; j < M; ++j) rs[j] = (j < rk && std::isfinite(ext_res[j])) ? ext_res[j] : std::numeric_limits<double>::infinity();
    } else {
        if (!d_coeftab_.alloc((size_t)nterms * ld)) return fail(FEASTHIP_EHIP, "out of device memory");
        std::vector<zc> lamk(lam.begin(), lam.begin() + rk);
        lamk.resize(M, zc(0.0));
        r = residual_block(d_Ym_.p, d_MXm_.p, d_Rm_.p, lamk, M, ld, d_coeftab_.p);
        if (r) return r;
        r = col_norms(tn, d_Rm_.p, nr);
        if (r) return r;
        r = build_frob_table();
        if (r) return r;
        for (int j = 0; j < M; ++j) {      // candidates beyond the numerical rank do not exist: residual +inf, zero vector
            rs[j] = j < rk ? nr[j] / poly_frob_norm(lam[j]) : std::numeric_limits<double>::infinity();
            if (!std::isfinite(rs[j])) rs[j] = std::numeric_limits<double>::infinity();
        }
    }
    std::vector<int> p(ld, 0);
    for (int j = 0; j < M; ++j) p[j] = j;
    std::stable_sort(p.begin(), p.begin() + M, [&](int a, int b) { return rs[a] < rs[b]; });     // sortperm(res)  utils.jl:125
    std::vector<zc> lam2(M);
    res.assign(M, 0.0);
    for (int j = 0; j < M; ++j) { lam2[j] = lam[p[j]]; res[j] = rs[p[j]]; }
    lam.swap(lam2);
    HIPCHK(hipMemcpyAsync(d_colidx_.p, p.data(), ld * sizeof(int), hipMemcpyHostToDevice, stream_));
    cz* tmp = d_MXm_.p;
    launch_gather_cols(stream_, n, ld, d_Ym_.p, ld, M, d_colidx_.p, tmp);
    HIPCHK(hipMemcpyAsync(d_Ym_.p, tmp, (size_t)wide * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    launch_gather_cols(stream_, n, ld, d_Rm_.p, ld, M, d_colidx_.p, tmp);
    HIPCHK(hipMemcpyAsync(d_Rm_.p, tmp, (size_t)wide * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    for (int j = 0; j < ld; ++j) p[j] = j < M ? j : 0;                       // X .= Y[:,1:m0], R[:,1:m0]  :230
    HIPCHK(hipMemcpyAsync(d_colidx_.p, p.data(), ld * sizeof(int), hipMemcpyHostToDevice, stream_));
    launch_gather_cols(stream_, n, ld, d_Ym_.p, ldw, m0, d_colidx_.p, d_X_.p);
    launch_gather_cols(stream_, n, ld, d_Rm_.p, ldw, m0, d_colidx_.p, d_R_.p);
    lambda_.assign(lam.begin(), lam.begin() + m0);
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    rr_ms = ms;
    return 0;
}

int Engine::nlm_rr_step(std::vector<zc>& lam, std::vector<double>& res) {
    int r = nlm_candidates(lam);
    if (r) return r;
    return nlm_finish(lam, res, nullptr, nullptr);
}

// Phase calls of the moments variant for a caller-side loop (opaque closure T: the residual block of the candidates is
// formed on the host, like the reference's update_R_moments!, and handed back).
int Engine::nlm_begin(const void* X, int64_t n_, int m0_, int moments) {
    if (moments < 1 || moments > 16) return fail(FEASTHIP_EARG, "nlfeast_moments: moments must be in 1..16");
    if (shard_stride_ != 1) return fail(FEASTHIP_ESTATE, "nlfeast_moments: node shards are not supported on this path");
    int r = set_x(X, n_, m0_);
    if (r) return r;
    r = nlm_alloc(moments);
    if (r) return r;
    return factorize();
}

int Engine::nlm_step_contour(int first) {
    if (!have_x_ || mm_ <= 0) return fail(FEASTHIP_ESTATE, "nlfeast_moments: call feasthip_nlm_begin first");
    return nlm_contour_step(first);
}

int Engine::nlm_step_candidates(double* lambda, void* Y) {
    if (!have_x_ || mm_ <= 0) return fail(FEASTHIP_ESTATE, "nlfeast_moments: call feasthip_nlm_begin first");
    int r = nlm_candidates(nlm_lam_);
    if (r) return r;
    const int M = mm_ * m0;
    for (int j = 0; j < M; ++j) { lambda[2 * j] = nlm_lam_[j].real(); lambda[2 * j + 1] = nlm_lam_[j].imag(); }
    return Y ? nlm_get_y(Y) : 0;
}

int Engine::nlm_get_y(void* Y) {
    const int M = mm_ * m0;
    DevBuf<cz> cm;
    if (!cm.alloc((size_t)n * M)) return fail(FEASTHIP_EHIP, "nlfeast_moments: out of device memory");
    launch_rm_to_cm(stream_, n, M, ldM_, d_Ym_.p, cm.p, n);
    HIPCHK(hipMemcpyAsync(Y, cm.p, (size_t)n * M * sizeof(cz), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return 0;
}

int Engine::nlm_step_finish(const void* R, const double* res_in, double* lambda, double* res, uint8_t* inside) {
    if (!have_x_ || mm_ <= 0 || nlm_lam_.empty()) return fail(FEASTHIP_ESTATE, "nlfeast_moments: call feasthip_nlm_candidates first");
    if ((R == nullptr) != (res_in == nullptr)) return fail(FEASTHIP_EARG, "nlfeast_moments: R and res go together");
    std::vector<double> rs;
    int r = nlm_finish(nlm_lam_, rs, R, res_in);
    if (r) return r;
    const int M = mm_ * m0;
    for (int j = 0; j < M; ++j) {
        lambda[2 * j] = nlm_lam_[j].real(); lambda[2 * j + 1] = nlm_lam_[j].imag(); res[j] = rs[j];
        if (inside) inside[j] = region_kind_ == FEASTHIP_REGION_RECT
                                    ? ((region_[0] < nlm_lam_[j].real() && nlm_lam_[j].real() < region_[2] && region_[1] < nlm_lam_[j].imag() && nlm_lam_[j].imag() < region_[3]) ? 1 : 0)
                                    : ((std::abs(nlm_lam_[j] - zc(region_[0], region_[1])) <= region_[2]) ? 1 : 0);
    }
    return 0;
}

int Engine::nlfeast_moments_run(void* X, int64_t n_, int m0_, int moments, int iter, double eps, double spurious, double* lambda,
                                double* res, uint8_t* inside, void* Y, int* nit_out) {
    if (moments < 1 || moments > 16) return fail(FEASTHIP_EARG, "nlfeast_moments: moments must be in 1..16");
    if ((int64_t)moments * m0_ > (int64_t)moments * n_) return fail(FEASTHIP_EARG, "nlfeast_moments: m0 must not exceed n");
    if (shard_stride_ != 1) return fail(FEASTHIP_ESTATE, "nlfeast_moments: node shards are not supported on this path");
    int r = set_x(X, n_, m0_);
    if (r) return r;
    r = nlm_alloc(moments);
    if (r) return r;
    int warn = factorize();                                                         // :187-201 (store=true)
    if (warn < 0) return warn;
    const int M = moments * m0;
    std::vector<zc> lam;
    std::vector<double> rs;
    std::vector<uint8_t> ins(m0);
    int nit = 0;
    for (nit = 0; nit <= iter; ++nit) {
        r = nlm_contour_step(nit == 0);
        if (r < 0) return r;
        r = nlm_rr_step(lam, rs);
        if (r) return r;
        inside_test(lam.data(), ins.data());
        if (getenv("FEASTHIP_RR_TRACE")) {
            int ni = 0; double mxi = 0;
            for (int j = 0; j < m0; ++j) if (ins[j]) { ++ni; mxi = std::max(mxi, rs[j]); }
            fprintf(stderr, "[feasthip] nlfeast_moments nit=%d inside(first m0)=%d max_res_inside=%.3e res[0..3]=%.2e %.2e %.2e %.2e\n", nit, ni, mxi,
                    rs[0], rs[std::min(1, M - 1)], rs[std::min(2, M - 1)], rs[std::min(3, M - 1)]);
        }
        if (nit == 0) continue;                                                     // the reference tests inside `for nit = 1:iter` only
        bool any = false, anys = false;
        double mx = 0, mxs = 0;
        for (int j = 0; j < m0; ++j) if (ins[j]) { any = true; mx = std::max(mx, rs[j]); if (rs[j] < spurious) { anys = true; mxs = std::max(mxs, rs[j]); } }
        if (any && mx < eps) break;                                                 // :296-299
        if (nit > 1 && anys && mxs < eps) break;                                    // :300-302
    }
    if (nit_out) *nit_out = std::min(nit, iter);
    r = nl_finalize_x();                                                            // normalize!(X) :311
    if (r) return r;
    r = get_block(0, X);
    if (r) return r;
    for (int j = 0; j < M; ++j) {
        lambda[2 * j] = lam[j].real(); lambda[2 * j + 1] = lam[j].imag(); res[j] = rs[j];
        if (inside) inside[j] = region_kind_ == FEASTHIP_REGION_RECT
                                    ? ((region_[0] < lam[j].real() && lam[j].real() < region_[2] && region_[1] < lam[j].imag() && lam[j].imag() < region_[3]) ? 1 : 0)
                                    : ((std::abs(lam[j] - zc(region_[0], region_[1])) <= region_[2]) ? 1 : 0);
    }
    if (Y) {      // all candidates, column-major n x M
        DevBuf<cz> cm;
        if (!cm.alloc((size_t)n * M)) return fail(FEASTHIP_EHIP, "nlfeast_moments: out of device memory");
        launch_rm_to_cm(stream_, n, M, ldM_, d_Ym_.p, cm.p, n);
        HIPCHK(hipMemcpyAsync(Y, cm.p, (size_t)n * M * sizeof(cz), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
    }
    return warn;
}

// ---------------------------------------------------------------------------
// dual_gen_feast!  (src/feast.jl:165-257): right and left eigenvectors of A x = lam B x.
// The reference factorises (A - z_k B)' separately (lfacts, :185-197); here the adjoint systems are simply K more
// "nodes" of the same batched factorisation: terms {A, B, A', B'}, node k < K has coefficients (1, -z_k, 0, 0), node
// K + k has (0, 0, 1, -conj(z_k)) (feasthip_set_node_coefficients), so no transposed-solve kernels are needed and the
// two halves of the sweep run concurrently.  Fixes of the reference's slips (fix, do not reproduce): the
// bi-orthonormalisation scales both bases with S^-1/2 (the reference writes Diagonal(1.0/S.S), :200-201), the left
// residual uses conj(lam) (A' y = conj(lam) B' y), and left Ritz vectors are paired with the right ones by eigenvalue.
// ---------------------------------------------------------------------------
int Engine::dual_contour_step() {
    if (dual_adj_) {
        // round 4: K factorisations of A - z_k B; the left systems (A - z_k B)^H y = Rl (the reference's lfacts, src/feast.jl:185-197, 241)
        // are adjoint sweeps over the same factors, instances K .. 2 K - 1, beside the primal sweeps on a second stream
        const size_t blk = (size_t)n * ldw;
        std::vector<zc> e((size_t)2 * Kloc * ldw, zc(0.0));
        for (int kk = 0; kk < Kloc; ++kk) {
            const zc z = nodes_[local_[kk]], w = weights_[local_[kk]];
            for (int j = 0; j < m0; ++j) {
                e[(size_t)kk * ldw + j] = w / (z - lambda_[j]);                                            // :226
                e[(size_t)(Kloc + kk) * ldw + j] = std::conj(w) / (std::conj(z) - std::conj(lambda_[j]));     // :236
            }
        }
        HIPCHK(hipMemcpyAsync(d_e0_.p, e.data(), e.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipEventRecord(ev0_, stream_));
        hipStream_t sa = aux_streams_.size() >= 2 ? aux_streams_[0] : stream_, sb = aux_streams_.size() >= 2 ? aux_streams_[1] : stream_;
        if (sa != stream_) {
            HIPCHK(hipEventRecord(ev_fork_, stream_));
            HIPCHK(hipStreamWaitEvent(sa, ev_fork_, 0));
            HIPCHK(hipStreamWaitEvent(sb, ev_fork_, 0));
        }
        int r = solve_range(sa, 0, Kloc, d_R_.p, nullptr);
        if (r == 0) r = solve_range_adj(sb, 0, Kloc, Kloc, d_Rl_.p, ldw);
        if (r) return r;
        if (sa != stream_) {
            HIPCHK(hipEventRecord(ev_join_[0], sa)); HIPCHK(hipStreamWaitEvent(stream_, ev_join_[0], 0));
            HIPCHK(hipEventRecord(ev_join_[1], sb)); HIPCHK(hipStreamWaitEvent(stream_, ev_join_[1], 0));
        }
        launch_accumulate(stream_, dsym_, d_X_.p, d_W_.p, (int64_t)blk, ldw, ldw, Kloc, d_e0_.p, nullptr, 1, Qp(), nullptr);                      // :232-234
        launch_accumulate(stream_, dsym_, d_Xl_.p, adj_solution() + (size_t)Kloc * blk, (int64_t)blk, ldw, ldw, Kloc, d_e0_.p + (size_t)Kloc * ldw, nullptr, 1,
                          Qp() + blk, nullptr);                                                                                          // :242-244
        HIPCHK(hipEventRecord(ev1_, stream_));
        HIPCHK(hipEventSynchronize(ev1_));
        HIPCHK(hipGetLastError());
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
        solve_ms = ms;
        return 0;
    }
    const int Kh = Kloc / 2;
    size_t blk = (size_t)n * ldw;
    std::vector<zc> e((size_t)Kloc * ldw, zc(0.0));
    for (int kk = 0; kk < Kloc; ++kk) {
        zc z = nodes_[local_[kk]], w = weights_[local_[kk]];
        for (int j = 0; j < m0; ++j) e[(size_t)kk * ldw + j] = w / (z - (kk < Kh ? lambda_[j] : std::conj(lambda_[j])));      // :226,236
    }
    HIPCHK(hipMemcpyAsync(d_e0_.p, e.data(), e.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    if (aux_streams_.size() >= 2) {      // right systems on one stream, adjoint systems on the other
        HIPCHK(hipEventRecord(ev_fork_, stream_));
        for (int g = 0; g < 2; ++g) {
            HIPCHK(hipStreamWaitEvent(aux_streams_[g], ev_fork_, 0));
            int r = solve_range(aux_streams_[g], g * Kh, Kh, g == 0 ? d_R_.p : d_Rl_.p, nullptr);
            if (r) return r;
            HIPCHK(hipEventRecord(ev_join_[g], aux_streams_[g]));
            HIPCHK(hipStreamWaitEvent(stream_, ev_join_[g], 0));
        }
    } else {
        int r = solve_range(stream_, 0, Kh, d_R_.p, nullptr);
        if (r == 0) r = solve_range(stream_, Kh, Kh, d_Rl_.p, nullptr);
        if (r) return r;
    }
    launch_accumulate(stream_, dsym_, d_X_.p, d_W_.p, (int64_t)blk, ldw, ldw, Kh, d_e0_.p, nullptr, 1, Qp(), nullptr);                          // :232-234
    launch_accumulate(stream_, dsym_, d_Xl_.p, d_W_.p + (size_t)Kh * blk, (int64_t)blk, ldw, ldw, Kh, d_e0_.p + (size_t)Kh * ldw, nullptr, 1,
                      Qp() + blk, nullptr);                                                                                             // :242-244
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    solve_ms = ms;
    return 0;
}

int Engine::dual_rr_step(double* lambda, double* resr, double* resl, uint8_t* inside) {
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    const int m = m0;
    size_t blk = (size_t)n * ldw;
    cz* Qr = Qp();
    cz* Ql = Qp() + blk;
    cz* dG = d_small_.p;
    cz* dM = d_small_.p + (size_t)2 * ldw * ldw;
    std::vector<zc> G((size_t)m * m), U((size_t)m * m), V((size_t)m * m), Aq((size_t)m * m), Bq((size_t)m * m), Aqh((size_t)m * m), Bqh((size_t)m * m);
    std::vector<double> sv(m);
    // S = svd(Ql' B Qr); Qr <- Qr V S^-1/2, Ql <- Ql U S^-1/2   (:199-201)
    apply_term(1, Qr, d_T1_.p, -1.0, false);                                       // B Qr  (term 1 carries scale -1)
    launch_gram(stream_, n, m, ldw, Ql, d_T1_.p, d_gram_scratch_.p, dG);
    HIPCHK(hipMemcpyAsync(G.data(), dG, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    jacobi_svd(m, G.data(), U.data(), sv.data(), V.data());
    double smax = sv[0];
    if (!(smax > 0.0) || !std::isfinite(smax)) return fail(FEASTHIP_ENUM, "dual_gen_feast: Ql' B Qr is zero or not finite");
    for (int j = 0; j < m; ++j) {
        double sc = 1.0 / std::sqrt(std::max(sv[j], 1e-300));
        for (int i = 0; i < m; ++i) { U[i + (size_t)j * m] *= sc; V[i + (size_t)j * m] *= sc; }
    }
    HIPCHK(hipMemcpyAsync(dM, V.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_ts_gemm(stream_, n, m, ldw, Qr, dM, d_T2_.p);
    HIPCHK(hipMemcpyAsync(Qr, d_T2_.p, blk * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipMemcpyAsync(dM, U.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_ts_gemm(stream_, n, m, ldw, Ql, dM, d_T2_.p);
    HIPCHK(hipMemcpyAsync(Ql, d_T2_.p, blk * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    // Aq = Ql' A Qr, Bq = Ql' B Qr   (:202-205)
    apply_term(0, Qr, d_R_.p, 1.0, false);
    launch_gram(stream_, n, m, ldw, Ql, d_R_.p, d_gram_scratch_.p, dG);
    HIPCHK(hipMemcpyAsync(Aq.data(), dG, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    apply_term(1, Qr, d_T1_.p, -1.0, false);
    launch_gram(stream_, n, m, ldw, Ql, d_T1_.p, d_gram_scratch_.p, dG);
    HIPCHK(hipMemcpyAsync(Bq.data(), dG, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { Aqh[i + (size_t)j * m] = std::conj(Aq[j + (size_t)i * m]); Bqh[i + (size_t)j * m] = std::conj(Bq[j + (size_t)i * m]); }
    std::vector<zc> lam(m), Xq((size_t)m * m), mu(m), Yq((size_t)m * m), Ym((size_t)m * m);
    if (eig_generalized(m, Aq.data(), Bq.data(), lam.data(), Xq.data())) return fail(FEASTHIP_ENUM, "dual_gen_feast: reduced eigenproblem failed");      // :206
    if (eig_generalized(m, Aqh.data(), Bqh.data(), mu.data(), Yq.data())) return fail(FEASTHIP_ENUM, "dual_gen_feast: adjoint reduced eigenproblem failed");   // :210
    std::vector<char> used(m, 0);
    for (int i = 0; i < m; ++i) {        // pair the left vector whose eigenvalue is conj(lam_i)
        int best = -1; double bd = 0;
        for (int j = 0; j < m; ++j) if (!used[j]) { double d = std::abs(mu[j] - std::conj(lam[i])); if (best < 0 || d < bd) { best = j; bd = d; } }
        used[best] = 1;
        for (int q = 0; q < m; ++q) Ym[q + (size_t)i * m] = Yq[q + (size_t)best * m];
    }
    auto ritz = [&](const std::vector<zc>& Mq, cz* Qb, cz* Xb) -> int {      // X = Q Mq, columns normalised (update_R!)
        HIPCHK(hipMemcpyAsync(dM, Mq.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
        launch_ts_gemm(stream_, n, m, ldw, Qb, dM, Xb);
        std::vector<double> nr;
        int r = col_norms(Xb, nr);
        if (r) return r;
        std::vector<zc> sc(ldw, zc(0.0));
        for (int j = 0; j < m; ++j) sc[j] = nr[j] > 0 ? 1.0 / nr[j] : 1.0;
        HIPCHK(hipMemcpyAsync(d_lam_.p, sc.data(), ldw * sizeof(zc), hipMemcpyHostToDevice, stream_));
        launch_scale_cols(stream_, n, ldw, Xb, d_lam_.p);
        HIPCHK(hipStreamSynchronize(stream_));
        return 0;
    };
    int r = ritz(Xq, Qr, d_X_.p);                                                   // :208-209
    if (r == 0) r = ritz(Ym, Ql, d_Xl_.p);                                          // :210-212
    if (r) return r;
    // Rr = A Xr - B Xr lam ; Rl = A' Xl - B' Xl conj(lam)   (:213-214)
    std::vector<zc> lp(ldw, zc(0.0)), lc(ldw, zc(0.0));
    for (int j = 0; j < m; ++j) { lp[j] = lam[j]; lc[j] = std::conj(lam[j]); }
    std::vector<double> nr, nl;
    HIPCHK(hipMemcpyAsync(d_lam_.p, lp.data(), ldw * sizeof(zc), hipMemcpyHostToDevice, stream_));
    apply_term(0, d_X_.p, d_T1_.p, 1.0, false);
    apply_term(1, d_X_.p, d_T2_.p, -1.0, false);
    launch_residual_combine(stream_, n, ldw, d_T1_.p, d_T2_.p, d_lam_.p, d_R_.p);
    r = col_norms(d_R_.p, nr);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(d_lam_.p, lc.data(), ldw * sizeof(zc), hipMemcpyHostToDevice, stream_));
    apply_term(2, d_Xl_.p, d_T1_.p, 1.0, false);
    apply_term(3, d_Xl_.p, d_T2_.p, -1.0, false);
    launch_residual_combine(stream_, n, ldw, d_T1_.p, d_T2_.p, d_lam_.p, d_Rl_.p);
    r = col_norms(d_Rl_.p, nl);
    if (r) return r;
    lambda_ = lam;
    for (int j = 0; j < m; ++j) { lambda[2 * j] = lam[j].real(); lambda[2 * j + 1] = lam[j].imag(); resr[j] = nr[j]; if (resl) resl[j] = nl[j]; }
    inside_test(lam.data(), inside);
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    rr_ms = ms;
    return 0;
}

int Engine::dual_run(void* Xr, void* Xl, int64_t n_, int m0_, int iter, double eps, int store, double* lambda, double* resr,
                     double* resl, uint8_t* inside, int* nit_out) {
    if (nterms != 4 || node_coef_.empty() || shard_stride_ != 1 || mixed_)
        return fail(FEASTHIP_ESTATE, "dual_gen_feast: needs the terms {A, B, A', B'} with a node coefficient table, one GPU, fp64");
    if (!Xl) return fail(FEASTHIP_EARG, "dual_gen_feast: null Xl");
    // Two layouts of the node table.  Round 4: K nodes, every row (f_A, f_B, 0, 0): K factorisations, the left systems are adjoint sweeps
    // over them.  Rounds 2-3: 2 K nodes, rows K .. 2 K - 1 = (0, 0, f_A', f_B'): the adjoint matrices factorised as K more nodes.
    bool no_adjoint_terms = true;
    for (int k = 0; k < K; ++k) if (node_coef_[(size_t)k * 4 + 2] != zc(0.0) || node_coef_[(size_t)k * 4 + 3] != zc(0.0)) no_adjoint_terms = false;
    if (!no_adjoint_terms && (K & 1)) return fail(FEASTHIP_ESTATE, "dual_gen_feast: the doubled layout needs 2K nodes");
    if (no_adjoint_terms != dual_adj_) { dual_adj_ = no_adjoint_terms; have_x_ = false; }
    int r = set_x(Xr, n_, m0_);
    if (r) return r;
    size_t blk = (size_t)n * ldw;
    if (!d_Xl_.alloc(blk) || !d_Rl_.alloc(blk)) return fail(FEASTHIP_EHIP, "out of device memory");
    HIPCHK(hipMemcpyAsync(d_io_.p, Xl, (size_t)n * m0 * sizeof(cz), hipMemcpyHostToDevice, stream_));
    launch_cm_to_rm(stream_, n, m0, ldw, d_io_.p, n, d_Xl_.p);
    HIPCHK(hipMemcpyAsync(Qp() + blk, d_Xl_.p, blk * sizeof(cz), hipMemcpyDeviceToDevice, stream_));      // Ql = copy(Xl)  :176
    HIPCHK(hipMemsetAsync(d_Rl_.p, 0, blk * sizeof(cz), stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    int warn = 0;
    if (store) { r = factorize(); if (r < 0) return r; warn = r; }
    int nit = 0;
    for (nit = 0; nit <= iter; ++nit) {
        r = dual_rr_step(lambda, resr, resl, inside);
        if (r) return r;
        bool any = false;
        double mx = 0;
        for (int j = 0; j < m0; ++j) if (inside[j]) { any = true; mx = std::max(mx, resr[j]); }
        if (any && mx < eps) break;                                                 // :218-221
        if (nit < iter) {                                                           // :222
            if (!store || !factored_) { r = factorize(); if (r < 0) return r; if (r > 0) warn = r; }
            r = dual_contour_step();
            if (r) return r;
        }
    }
    if (nit_out) *nit_out = std::min(nit, iter);
    r = get_block(0, Xr);
    if (r) return r;
    launch_rm_to_cm(stream_, n, m0, ldw, d_Xl_.p, d_io_.p, n);
    HIPCHK(hipMemcpyAsync(Xl, d_io_.p, (size_t)n * m0 * sizeof(cz), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return warn;
}

// ---------------------------------------------------------------------------
// whole loops
// ---------------------------------------------------------------------------
int Engine::nlfeast_run(void* X, int64_t n_, int m0_, int iter, double eps, double spurious, double* lambda, double* res,
                        uint8_t* inside, int* nit_out) {
    int r = set_x(X, n_, m0_);
    if (r) return r;
    r = nl_orthonormalize_x();
    if (r) return r;
    int warn = factorize();                                                         // src/nlfeast.jl:17-28
    if (warn < 0) return warn;
    int nit = 0;
    for (nit = 0; nit <= iter; ++nit) {
        r = nl_contour_step(nit == 0);
        if (r < 0) return r;
        r = nl_rr_step(lambda, res, inside);
        if (r) return r;
        bool any = false, anys = false;
        double mx = 0, mxs = 0;
        for (int j = 0; j < m0; ++j) if (inside[j]) { any = true; mx = std::max(mx, res[j]); if (res[j] < spurious) { anys = true; mxs = std::max(mxs, res[j]); } }
        if (any && mx < eps) break;                                                 // :73-76
        if (nit > 1 && anys && mxs < eps) break;                                    // :77-79
    }
    if (nit_out) *nit_out = std::min(nit, iter);
    r = nl_finalize_x();
    if (r) return r;
    r = get_block(0, X);
    if (r) return r;
    return warn;
}

}  // namespace feasthip
