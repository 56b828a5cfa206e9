// Term products, orthonormalisation and the Rayleigh-Ritz step (src/feast.jl:33-57, 108-131; src/utils.jl:69-77).
#include "engine_impl.hpp"

namespace feasthip {

void Engine::apply_term(int t, const cz* V, cz* out, zc alpha, bool accumulate) { apply_term_ld(t, V, out, alpha, accumulate, ldw); }

void Engine::apply_term_ld(int t, const cz* V, cz* out, zc alpha, bool accumulate, int ld) {
    SparseTerm& T = terms_[t];
    zc a = alpha * T.scale;
    if (dense) {
        if (T.identity) launch_axpby(stream_, n, ld, out, to_cz(a), V, accumulate ? 1 : 0);
        else {
            const int parts = dense_mv_parts(n, ld);
            if (parts > 1 && d_mvpart_.n < (size_t)parts * n * ld && !d_mvpart_.alloc((size_t)parts * n * ld)) { (void)hipGetLastError(); }
            launch_dense_mv(stream_, n, T.d_dense.p, V, ld, out, to_cz(a), accumulate ? 1 : 0,
                            parts > 1 && d_mvpart_.n >= (size_t)parts * n * ld ? d_mvpart_.p : nullptr);
        }
    } else {
        launch_spmm_csr(stream_, n, T.d_rowptr.p, T.d_col.p, T.d_val.p, V, ld, out, to_cz(a), accumulate ? 1 : 0);
    }
}

int Engine::col_norms(const Tall& t, const cz* V, std::vector<double>& out) {
    launch_colnorm2(stream_, t.rows, t.ld, V, t.nrm_scratch, t.nrm);
    out.resize(t.ld);
    void* land = nullptr;
    HIPCHK(d2h_async(4, out.data(), t.nrm, t.ld * sizeof(double), &land));
    HIPCHK(hipStreamSynchronize(stream_));
    if (land != (void*)out.data()) std::memcpy(out.data(), land, t.ld * sizeof(double));
    for (double& v : out) v = std::sqrt(v);
    return 0;
}

// Thin orthonormal basis of span(Q) (replaces `Q .= Matrix(qr(Q).Q)`, src/feast.jl:41,117).
// Column-scaled CholeskyQR2: columns are normalised first (the FEAST filter scales column j by
// rho(lambda_j), which is the dominant source of ill-conditioning), then two Cholesky passes on
// the f64-MFMA Gram matrix.  Rtotal (optional) receives R with Q_in = Q_out * R.
int Engine::orthonormalize(const Tall& t, cz* Q, cz* tmp, std::vector<zc>* Rtotal, int min_passes) {
    const int m = t.m, rows = t.rows, ld = t.ld;
    std::vector<zc> G((size_t)m * m), Rn((size_t)m * m), Rt, Tm((size_t)m * m);
    if (Rtotal) { Rt.assign((size_t)m * m, zc(0.0)); for (int j = 0; j < m; ++j) Rt[j + (size_t)j * m] = 1.0; }
    cz* dG = t.small;
    cz* dM = t.small + (size_t)2 * ld * ld;
    cz* src = Q;
    cz* dst = tmp;
    // two clean passes; a pass that needed a shift (numerically rank-deficient block) is followed by one more.
    // The column scaling is folded into the first pass on the host: the Gram matrix of the scaled block is
    // D G D with D = diag(G)^-1/2 (each G_ij carries a relative error eps in units of |q_i||q_j| either way),
    // which saves a norm pass, a scaling pass and a host round trip over the n x m block.
    // min_passes = 1 (generalized Rayleigh-Ritz, where Bq = Q'BQ is formed explicitly and any well-conditioned basis of span(Q)
    // gives the same Ritz pairs): the second pass is dropped when the scaled block was well conditioned.  Round 3 decided that on
    // the ratio of the extreme diagonal entries of the first Cholesky factor, which bounds cond(R) only from below (Kahan-type
    // factors); now on ||R||_F ||R^-1||_F / m = ||R^-1||_F / sqrt(m) <= rr_cond_pass2_ (1e3) from the inverse that is formed anyway
    // (||R||_F^2 = trace of the scaled Gram matrix = m; the quantity is 1 for an orthonormal block and bounds cond_2(R) / sqrt(m)
    // from above; 1.1 .. 2 in the steady state of a FEAST run, 1e2 .. 1e4 on the first filtered block of a random start).  The
    // basis after one pass is then orthonormal to eps * 1e6 m or better, and the reduced pencil (Q'AQ, Q'BQ) is formed with it.
    int need = 2;
    bool first_clean = false;
    for (int pass = 0; pass < 6 && need > 0; ++pass) {
        launch_gram(stream_, rows, m, ld, src, src, t.gram, dG);
        void* land = nullptr;
        HIPCHK(d2h_async(0, G.data(), dG, (size_t)m * m * sizeof(zc), &land));
        HIPCHK(hipStreamSynchronize(stream_));
        if (land != (void*)G.data()) std::memcpy(G.data(), land, (size_t)m * m * sizeof(zc));
        for (int j = 0; j < m; ++j) for (int i = 0; i < j; ++i) { zc a = 0.5 * (G[i + (size_t)j * m] + std::conj(G[j + (size_t)i * m])); G[i + (size_t)j * m] = a; G[j + (size_t)i * m] = std::conj(a); }
        std::vector<double> dsc(m, 1.0);
        if (pass == 0) {
            for (int j = 0; j < m; ++j) {
                double g = G[j + (size_t)j * m].real();
                if (!(g > 0.0) || !std::isfinite(g)) return fail(FEASTHIP_ENUM, "orthonormalize: zero or non-finite column in the subspace");
                dsc[j] = std::sqrt(g);
            }
            for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) G[i + (size_t)j * m] /= dsc[i] * dsc[j];
        }
        std::vector<zc> Gc(G);
        int info = cholesky_upper(m, Gc.data());
        if (info) {   // shifted CholeskyQR for a numerically rank-deficient block
            double tr = 0;
            for (int j = 0; j < m; ++j) tr += G[j + (size_t)j * m].real();
            double shift = 1e-13 * tr + 1e-300;
            for (int tries = 0; tries < 6 && info; ++tries) {
                Gc = G;
                for (int j = 0; j < m; ++j) Gc[j + (size_t)j * m] += shift;
                info = cholesky_upper(m, Gc.data());
                shift *= 100.0;
            }
            if (info) return fail(FEASTHIP_ENUM, "orthonormalize: Gram matrix not positive definite");
            need = 1;
        } else {
            --need;
            if (pass == 0 && min_passes <= 1) first_clean = true;
        }
        if (Rtotal) {   // Q_src = Q_dst (R D^-1)  ->  Rt <- (R D^-1) Rt   (only the Beyn step needs the accumulated factor)
            std::vector<zc> Rd(Gc);
            for (int j = 0; j < m; ++j) for (int i = 0; i <= j; ++i) Rd[i + (size_t)j * m] *= dsc[j];
            matmul(m, Rd.data(), Rt.data(), Rn.data());
            Rt.swap(Rn);
        }
        trtri_upper(m, Gc.data());
        if (first_clean) {
            double fro = 0;
            for (int j = 0; j < m; ++j) for (int i = 0; i <= j; ++i) fro += std::norm(Gc[i + (size_t)j * m]);
            const double kappa = std::sqrt(fro / m);
            if (getenv("FEASTHIP_RR_TRACE")) fprintf(stderr, "[feasthip] CholeskyQR pass 1: ||R^-1||_F / sqrt(m) = %.3g\n", kappa);
            if (kappa <= rr_cond_pass2_) need = 0;
            first_clean = false;
        }
        for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) Tm[i + (size_t)j * m] = Gc[i + (size_t)j * m] / dsc[i];     // T = D^-1 R^-1
        HIPCHK(h2d_async(1, dM, Tm.data(), (size_t)m * m * sizeof(zc)));
        launch_ts_gemm(stream_, rows, m, ld, src, dM, dst);
        std::swap(src, dst);
    }
    if (src != Q) HIPCHK(hipMemcpyAsync(Q, src, (size_t)rows * ld * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    if (Rtotal) *Rtotal = Rt;
    return 0;
}

// Rank-revealing variant for numerically rank-deficient blocks (the block-Hankel matrix of nlfeast_moments! has
// rank = #eigenvalues inside as soon as the iteration converges).  Iterated pivoted block Gram-Schmidt built from
// the same two device operations as above (MFMA Gram, tall-skinny GEMM): per pass a pivoted partial Cholesky of the
// Gram matrix on the host accepts the columns that kept at least 1e-8 of their squared norm (the rest would be
// decided on cancellation noise), orthonormalises them and projects them out of the others explicitly; columns
// whose explicitly formed remainder is below 1e-13 of their original norm are dropped (zeroed).  On return
// Q_in = Q_out * Rtot, kept columns of Q_out are orthonormal, dropped ones (keep[j] = 0) are zero.
thread_local std::string g_ortho_diag;          // what this thread's last orthonormalize_rr saw, pass by pass (error messages)
int Engine::orthonormalize_rr(const Tall& t, cz* Q, cz* tmp, std::vector<zc>& Rtot, std::vector<uint8_t>& keep) {
    std::string& ortho_diag_ = g_ortho_diag;
    const int m = t.m, rows = t.rows, ld = t.ld;
    std::vector<double> nr;
    int r = col_norms(t, Q, nr);
    if (r) return r;
    std::vector<zc> sc(ld, zc(0.0));
    Rtot.assign((size_t)m * m, zc(0.0));
    keep.assign(m, 1);
    ortho_diag_.clear();
    {
        char line[200]; double mn = 1e300, mx = 0;
        for (int j = 0; j < m; ++j) { mn = std::min(mn, nr[j]); mx = std::max(mx, nr[j]); }
        snprintf(line, sizeof line, " column norms %.3e .. %.3e;", mn, mx); ortho_diag_ += line;
    }
    for (int j = 0; j < m; ++j) {
        if (!std::isfinite(nr[j])) return fail(FEASTHIP_ENUM, "orthonormalize: non-finite column in the subspace");
        if (nr[j] > 0.0) { sc[j] = 1.0 / nr[j]; Rtot[j + (size_t)j * m] = nr[j]; } else keep[j] = 0;
    }
    HIPCHK(hipMemcpyAsync(t.lam, sc.data(), ld * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_scale_cols(stream_, rows, ld, Q, t.lam);
    HIPCHK(hipStreamSynchronize(stream_));
    const double tol_keep = 1e-8, tol_drop2 = 1e-26;       // squared-norm thresholds (columns are scaled to unit norm)
    std::vector<zc> G((size_t)m * m), W((size_t)m * m), T((size_t)m * m), Ti((size_t)m * m), Rn((size_t)m * m), R11, R11i;
    cz* dG = t.small;
    cz* dM = t.small + (size_t)2 * ld * ld;
    for (int pass = 0; pass < 10; ++pass) {
        launch_gram(stream_, rows, m, ld, Q, Q, t.gram, dG);
        HIPCHK(hipMemcpyAsync(G.data(), dG, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (const zc& v : G) if (!std::isfinite(v.real()) || !std::isfinite(v.imag())) return fail(FEASTHIP_ENUM, "orthonormalize: non-finite Gram matrix of the subspace (" + ortho_diag_ + ")");
        for (int j = 0; j < m; ++j) for (int i = 0; i < j; ++i) { zc a = 0.5 * (G[i + (size_t)j * m] + std::conj(G[j + (size_t)i * m])); G[i + (size_t)j * m] = a; G[j + (size_t)i * m] = std::conj(a); }
        W = G;
        std::vector<double> d0(m);
        std::vector<int> state(m, 0), acc;                   // 0 undecided, 1 accepted, 2 dropped
        double defect = 0;
        for (int j = 0; j < m; ++j) {
            d0[j] = G[j + (size_t)j * m].real();
            if (!keep[j] || !(d0[j] >= tol_drop2)) { state[j] = 2; keep[j] = 0; }
        }
        for (int j = 0; j < m; ++j) if (state[j] == 0) for (int i = 0; i < m; ++i) if (state[i] == 0) defect = std::max(defect, std::abs(G[i + (size_t)j * m] - (i == j ? 1.0 : 0.0)));
        std::vector<std::vector<zc>> rowsR;                  // row t of [R11 R12] over all columns
        for (;;) {
            int bj = -1; double bd = 0;
            for (int j = 0; j < m; ++j) if (state[j] == 0) { double d = W[j + (size_t)j * m].real(); if (d >= tol_keep * d0[j] && d > bd) { bd = d; bj = j; } }
            if (bj < 0) break;
            double piv = std::sqrt(bd);
            std::vector<zc> row(m, zc(0.0));
            for (int i = 0; i < m; ++i) if (state[i] == 0) row[i] = W[bj + (size_t)i * m] / piv;
            row[bj] = piv;
            state[bj] = 1;
            for (int b = 0; b < m; ++b) if (state[b] == 0) for (int a = 0; a < m; ++a) if (state[a] == 0) W[a + (size_t)b * m] -= std::conj(row[a]) * row[b];
            acc.push_back(bj);
            rowsR.push_back(row);
        }
        int ra = (int)acc.size(), ndef = 0;
        for (int j = 0; j < m; ++j) if (state[j] == 0) ++ndef;   // deferred: projected now, decided on the next pass
        {
            char line[200]; double mn = 1e300, mx = -1e300; int nonfinite = 0;
            for (int j = 0; j < m; ++j) { if (!std::isfinite(d0[j])) ++nonfinite; else { mn = std::min(mn, d0[j]); mx = std::max(mx, d0[j]); } }
            snprintf(line, sizeof line, " pass %d: diag(G) %.3e .. %.3e (%d non-finite), %d accepted, %d deferred, defect %.2e;", pass, mn, mx, nonfinite, ra, ndef, defect);
            ortho_diag_ += line;
        }
        // R11 (ra x ra, upper triangular in acceptance order) and its inverse
        R11.assign((size_t)ra * ra, zc(0.0));
        for (int tt = 0; tt < ra; ++tt) for (int s2 = tt; s2 < ra; ++s2) R11[tt + (size_t)s2 * ra] = rowsR[tt][acc[s2]];
        R11i = R11;
        if (ra > 0) trtri_upper(ra, R11i.data());
        std::fill(T.begin(), T.end(), zc(0.0));
        std::fill(Ti.begin(), Ti.end(), zc(0.0));
        for (int s2 = 0; s2 < ra; ++s2) for (int tt = 0; tt <= s2; ++tt) T[acc[tt] + (size_t)acc[s2] * m] = R11i[tt + (size_t)s2 * ra];
        for (int tt = 0; tt < ra; ++tt) for (int i = 0; i < m; ++i) if (state[i] != 2) Ti[acc[tt] + (size_t)i * m] = rowsR[tt][i];
        for (int b = 0; b < m; ++b) if (state[b] == 0) {
            T[b + (size_t)b * m] = 1.0; Ti[b + (size_t)b * m] = 1.0;
            for (int tt = 0; tt < ra; ++tt) { zc v = 0; for (int s2 = tt; s2 < ra; ++s2) v += R11i[tt + (size_t)s2 * ra] * rowsR[s2][b]; T[acc[tt] + (size_t)b * m] = -v; }
        }
        matmul(m, Ti.data(), Rtot.data(), Rn.data());        // Q_orig = Q_cur Rtot = (Q_cur T)(T^-1 Rtot)
        Rtot.swap(Rn);
        HIPCHK(hipMemcpyAsync(dM, T.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
        launch_ts_gemm(stream_, rows, m, ld, Q, dM, tmp);
        HIPCHK(hipMemcpyAsync(Q, tmp, (size_t)rows * ld * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        // done once a pass started from an orthonormal kept set (Gram = I to 1e-10) and deferred nothing
        if (ndef == 0 && defect < 1e-10) break;
    }
    return 0;
}

void Engine::inside_test(const zc* lam, uint8_t* inside) const {
    for (int j = 0; j < m0; ++j) {
        if (region_kind_ == FEASTHIP_REGION_RECT)
            inside[j] = (region_[0] < lam[j].real() && lam[j].real() < region_[2] && region_[1] < lam[j].imag() && lam[j].imag() < region_[3]) ? 1 : 0;
        else
            inside[j] = (std::abs(lam[j] - zc(region_[0], region_[1])) <= region_[2]) ? 1 : 0;
    }
}

// The Rayleigh-Ritz step (src/feast.jl:41-50, 117-127) for Hermitian(-definite) pencils without a host round trip: the reference
// orthonormalises Q, projects and calls LAPACK; any basis of span(Q) gives the same Ritz pairs, so the projection is taken on the RAW
// filtered block (Ga = Q'AQ, Gb = Q'BQ, three launches) and the whole m0 x m0 problem -- column scaling, Cholesky of Gb, congruence,
// Jacobi eigensolver, back-transform, ordering -- runs in one workgroup on the device (kernels_small.hip); X = Q Xq, the residual block
// and the column norms follow on the same stream and the host synchronises ONCE, for lambda, the residual norms and the solver's status
// words.  The status decides whether the result stands: inputs Hermitian to 1e-13 (Julia's eigen! takes the Hermitian path for exactly
// Hermitian input; here "up to rounding", as on the host path), Gb positive definite, the column-scaled basis well conditioned
// (||R^-1||_F / sqrt(m) <= rr_cond_max_: no orthonormalisation was needed), Jacobi converged.  Otherwise *done = false, Q is untouched
// and rr_step continues on the host path (CholeskyQR2 + smalleig.cpp), exactly as in rounds 1-3.
int Engine::rr_step_device(double* lambda, double* res, uint8_t* inside, bool* done) {
    *done = false;
    const int m = m0;
    const bool b_is_identity = terms_[1].identity && std::abs(terms_[1].scale + 1.0) == 0.0;
    const bool b_diag = !b_is_identity && !dense && terms_[1].diagonal && terms_[1].scale == zc(-1.0);
    cz* dA = d_small_.p;
    cz* dB = d_small_.p + (size_t)ldw * ldw;
    cz* dM = d_small_.p + (size_t)2 * ldw * ldw;
    double* dstat = d_rrstat_.p;
    double* dnx = d_rrstat_.p + 16;
    HIPCHK(hipEventRecord(ev0_, stream_));
    apply_term(0, Qp(), d_R_.p, 1.0, false);                                                  // R = A Q          :42
    HIPCHK(hipEventRecord(ev_p0_, stream_));
    launch_gram(stream_, n, m, ldw, Qp(), d_R_.p, d_gram_scratch_.p, dA);                     // Ga = Q' R        :43
    HIPCHK(hipEventRecord(ev_p1_, stream_));
    if (b_diag) launch_gram(stream_, n, m, ldw, Qp(), Qp(), d_gram_scratch_.p, dB, terms_[1].d_diag.p);
    else if (b_is_identity) launch_gram(stream_, n, m, ldw, Qp(), Qp(), d_gram_scratch_.p, dB);
    else {
        apply_term(1, Qp(), d_T1_.p, -1.0, false);                                            // B Q              :120
        launch_gram(stream_, n, m, ldw, Qp(), d_T1_.p, d_gram_scratch_.p, dB);                // Gb               :121
    }
    // status words start as NaN: a launch the runtime rejected (or a kernel that never wrote them) must not read as "flags 0, done"
    HIPCHK(hipMemsetAsync(dstat, 0xFF, 16 * sizeof(double), stream_));
    launch_rr_small_herm(stream_, m, dA, dB, dM, d_lam_.p, ldw, dstat, d_rrscr_.p, rr_cond_max_);   // :45, :122
    HIPCHK(hipGetLastError());
    launch_ts_gemm(stream_, n, m, ldw, Qp(), dM, d_X_.p);                                     // X = Q Xq         :48
    const cz* AX;
    const cz* BX = nullptr;
    if (dense) {
        launch_ts_gemm(stream_, n, m, ldw, d_R_.p, dM, d_T2_.p); AX = d_T2_.p;               // A X = (A Q) Xq
        if (!b_is_identity) { launch_ts_gemm(stream_, n, m, ldw, d_T1_.p, dM, d_MX_.p); BX = d_MX_.p; }
    } else {
        launch_ts_gemm(stream_, n, m, ldw, d_R_.p, dM, d_T1_.p); AX = d_T1_.p;
        if (!b_is_identity && !b_diag) { apply_term(1, d_X_.p, d_T2_.p, -1.0, false); BX = d_T2_.p; }
    }
    launch_colnorm2(stream_, n, ldw, d_X_.p, d_nrm_scratch_.p, dnx);
    // update_R! (src/utils.jl:111-116): columns of X to unit 2-norm, R[:, i] = (A - lam_i B) x_i
    launch_residual_combine_scaled(stream_, n, ldw, AX, b_is_identity ? d_X_.p : BX, d_lam_.p, d_X_.p, d_R_.p, b_diag ? terms_[1].d_diag.p : nullptr, dnx);
    launch_colnorm2(stream_, n, ldw, d_R_.p, d_nrm_scratch_.p, d_nrm_.p);                   // residuals! :166-171
    std::vector<double> nr(ldw), st(16);
    std::vector<zc> lam(ldw);
    void *l0 = nullptr, *l1 = nullptr, *l2 = nullptr;
    HIPCHK(d2h_async(4, nr.data(), d_nrm_.p, ldw * sizeof(double), &l0));
    HIPCHK(d2h_async(7, lam.data(), d_lam_.p, ldw * sizeof(zc), &l1));      // (slot 5 carries the coefficient table of the contour step)
    HIPCHK(d2h_async(6, st.data(), dstat, 16 * sizeof(double), &l2));
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    if (l0 != (void*)nr.data()) std::memcpy(nr.data(), l0, ldw * sizeof(double));
    if (l1 != (void*)lam.data()) std::memcpy(lam.data(), l1, ldw * sizeof(zc));
    if (l2 != (void*)st.data()) std::memcpy(st.data(), l2, 16 * sizeof(double));
    const int flags = std::isfinite(st[0]) ? (int)st[0] : -2;      // -2: the kernel did not report (sentinel still there)
    if (getenv("FEASTHIP_RR_TRACE"))
        fprintf(stderr, "[feasthip] reduced eigenproblem on the device (m=%d): flags %d, %d sweeps, ||R^-1||_F/sqrt(m) %.3g, Hermitian defect %.2e, diag(R) min/max %.3g; kernel %.0f (load %.0f, Cholesky %.0f, inverse %.0f, congruence %.0f) + %.0f + %.0f us\n",
                m, flags, (int)st[1], st[2], st[3], st[4], st[5], st[8], st[9], st[10], st[11], st[6], st[7]);
    if (flags != 0) {
        if (flags & 1) rr_dev_state_ = -1;          // not a Hermitian problem: never again on this handle
        rr_device_sweeps = 0;
        return 0;
    }
    bool finite = true;
    for (int j = 0; j < m; ++j) if (!std::isfinite(nr[j]) || !std::isfinite(lam[j].real())) finite = false;
    if (!finite) return 0;
    rr_device_sweeps = (int)st[1];
    lam.resize(m);
    lambda_ = lam;
    for (int j = 0; j < m; ++j) { lambda[2 * j] = lam[j].real(); lambda[2 * j + 1] = lam[j].imag(); res[j] = std::sqrt(nr[j]); }
    inside_test(lam.data(), inside);
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    rr_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, ev_p0_, ev_p1_));
    proj_ms = ms;
    *done = true;
    return 0;
}

// the device solver alone: lambda ascending, X with X' B X = I (B = NULL: identity), status[8] as in kernels_small.hip
int Engine::small_eig_herm(int m, const void* A, const void* B, double* lambda, void* X, double* status) {
    if (m < 1 || m > 64 || !A || !lambda || !X) return fail(FEASTHIP_EARG, "small_eig_herm: 1 <= m <= 64");
    HIPCHK(hipSetDevice(device_));
    DevBuf<cz> dA, dB, dX, dl, dscr;
    DevBuf<double> dst;
    const int ld = ((m + 15) / 16) * 16;
    if (!dA.alloc((size_t)m * m) || !dB.alloc((size_t)m * m) || !dX.alloc((size_t)m * m) || !dl.alloc(ld) || !dscr.alloc(64 * 64) || !dst.alloc(16))
        return fail(FEASTHIP_EHIP, "small_eig_herm: device allocation failed");
    std::vector<zc> eye;
    if (!B) { eye.assign((size_t)m * m, zc(0.0)); for (int j = 0; j < m; ++j) eye[j + (size_t)j * m] = 1.0; B = eye.data(); }
    HIPCHK(hipMemcpyAsync(dA.p, A, (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(dB.p, B, (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
    launch_rr_small_herm(stream_, m, dA.p, dB.p, dX.p, dl.p, ld, dst.p, dscr.p, 1e300);
    std::vector<zc> lam(ld);
    double st[8];
    HIPCHK(hipMemcpyAsync(X, dX.p, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(lam.data(), dl.p, ld * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(st, dst.p, sizeof(st), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    for (int j = 0; j < m; ++j) lambda[j] = lam[j].real();
    if (status) std::memcpy(status, st, sizeof(st));
    return 0;
}

int Engine::rr_step(double* lambda, double* res, uint8_t* inside) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "rr_step: call set_x first");
    if (nterms != 2) return fail(FEASTHIP_ESTATE, "rr_step: linear pencil (2 terms) required");
    HIPCHK(hipSetDevice(device_));
    if (rr_device_ && rr_dev_state_ >= 0 && m0 <= 64) {
        bool done = false;
        int rd = rr_step_device(lambda, res, inside, &done);
        if (rd) return rd;
        if (done) return 0;
    }
    rr_device_sweeps = 0;
    HIPCHK(hipEventRecord(ev0_, stream_));
    int m = m0;
    // B = -scale_1 * M_1
    const bool b_is_identity = terms_[1].identity && std::abs(terms_[1].scale + 1.0) == 0.0;
    int r = orthonormalize(tall_default(), Qp(), d_T1_.p, nullptr, b_is_identity ? 2 : 1);      // src/feast.jl:41, 117
    if (r) return r;
    cz* dA = d_small_.p;
    cz* dB = d_small_.p + (size_t)ldw * ldw;
    cz* dM = d_small_.p + (size_t)2 * ldw * ldw;
    std::vector<zc> Aq((size_t)m * m), Bq((size_t)m * m), lam(m), Xq((size_t)m * m);
    apply_term(0, Qp(), d_R_.p, 1.0, false);                            // R = A Q          :42
    HIPCHK(hipEventRecord(ev_p0_, stream_));
    launch_gram(stream_, n, m, ldw, Qp(), d_R_.p, d_gram_scratch_.p, dA);   // Aq = Q' R   :43
    HIPCHK(hipEventRecord(ev_p1_, stream_));
    void *landA = nullptr, *landB = nullptr;
    // B = -scale_1 M_1 with M_1 diagonal (lumped mass matrix; Diagonal(...) in test/runtests.jl:21): Bq = Q' diag(b) Q straight from
    // the Gram kernel and B X = diag(b) X inside the residual kernel -- two SpMM passes over n x m0 blocks less per iteration
    const bool b_diag = !b_is_identity && !dense && terms_[1].diagonal && terms_[1].scale == zc(-1.0);
    if (b_diag) {
        launch_gram(stream_, n, m, ldw, Qp(), Qp(), d_gram_scratch_.p, dB, terms_[1].d_diag.p);      // Bq    :121
        HIPCHK(d2h_async(2, Bq.data(), dB, (size_t)m * m * sizeof(zc), &landB));
    } else if (!b_is_identity) {
        apply_term(1, Qp(), d_T1_.p, -1.0, false);                      // R = B Q          :120
        launch_gram(stream_, n, m, ldw, Qp(), d_T1_.p, d_gram_scratch_.p, dB);   // Bq    :121
        HIPCHK(d2h_async(2, Bq.data(), dB, (size_t)m * m * sizeof(zc), &landB));
    }
    HIPCHK(d2h_async(0, Aq.data(), dA, (size_t)m * m * sizeof(zc), &landA));
    HIPCHK(hipStreamSynchronize(stream_));
    if (landA != (void*)Aq.data()) std::memcpy(Aq.data(), landA, (size_t)m * m * sizeof(zc));
    if (landB && landB != (void*)Bq.data()) std::memcpy(Bq.data(), landB, (size_t)m * m * sizeof(zc));
    // Julia's eigen! takes the Hermitian(-definite) LAPACK path for exactly Hermitian input; here the projected
    // matrices are Hermitian up to rounding whenever A, B are, and the same path is taken below 1e-13.
    auto th0 = std::chrono::steady_clock::now();
    int info = 1;
    bool herm = hermitian_defect(m, Aq.data()) <= 1e-13 && (b_is_identity || hermitian_defect(m, Bq.data()) <= 1e-13);
    if (herm) {
        std::vector<zc> a2(Aq), b2(Bq);
        info = eig_hermitian(m, a2.data(), b_is_identity ? nullptr : b2.data(), lam.data(), Xq.data());
    }
    if (info) info = b_is_identity ? eig_general(m, Aq.data(), lam.data(), Xq.data())     // :45
                                   : eig_generalized(m, Aq.data(), Bq.data(), lam.data(), Xq.data());   // :122
    if (info) return fail(FEASTHIP_ENUM, "rr_step: reduced eigenproblem failed to converge");
    if (getenv("FEASTHIP_RR_TRACE"))
        fprintf(stderr, "[feasthip] reduced eigenproblem (m=%d, %s): %.3f ms host\n", m, herm ? "hermitian" : "general",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count());
    // update_R! normalises the columns of X = Q Xq (src/utils.jl:113); Q is orthonormal, so ||X_j|| = ||Xq_j||
    // and the normalisation is applied to the m x m Xq on the host (no extra pass over the n x m block)
    for (int j = 0; j < m; ++j) {
        double nq = 0;
        for (int i = 0; i < m; ++i) nq += std::norm(Xq[i + (size_t)j * m]);
        nq = std::sqrt(nq);
        if (nq > 0) for (int i = 0; i < m; ++i) Xq[i + (size_t)j * m] /= nq;
    }
    std::vector<zc> lamp(ldw, zc(0.0));
    for (int j = 0; j < m; ++j) lamp[j] = lam[j];
    HIPCHK(h2d_async(1, dM, Xq.data(), (size_t)m * m * sizeof(zc)));
    HIPCHK(h2d_async(3, d_lam_.p, lamp.data(), ldw * sizeof(zc)));
    launch_ts_gemm(stream_, n, m, ldw, Qp(), dM, d_X_.p);              // X = Q Xq         :48
    std::vector<double> nr;
    const cz* BX = d_X_.p;
    if (dense) {
        // A X = (A Q) Xq, B X = (B Q) Xq: reuse the products of the projection instead of two more n x n passes
        launch_ts_gemm(stream_, n, m, ldw, d_R_.p, dM, d_T2_.p);                       // A X
        if (!b_is_identity) { launch_ts_gemm(stream_, n, m, ldw, d_T1_.p, dM, d_MX_.p); BX = d_MX_.p; }
        launch_residual_combine(stream_, n, ldw, d_T2_.p, BX, d_lam_.p, d_R_.p);
    } else {
        // A X = (A Q) Xq: one tall-skinny GEMM on the product the projection already formed instead of a second
        // SpMM with A (the gather of 19 block-vector rows per row costs twice as much); B X is a cheap SpMM
        launch_ts_gemm(stream_, n, m, ldw, d_R_.p, dM, d_T1_.p);        // A X
        if (!b_is_identity && !b_diag) { apply_term(1, d_X_.p, d_T2_.p, -1.0, false); BX = d_T2_.p; }
        launch_residual_combine(stream_, n, ldw, d_T1_.p, BX, d_lam_.p, d_R_.p, b_diag ? terms_[1].d_diag.p : nullptr);
    }
    r = col_norms(d_R_.p, nr);                                           // residuals! :166-171
    if (r) return r;
    lambda_ = lam;
    for (int j = 0; j < m; ++j) { lambda[2 * j] = lam[j].real(); lambda[2 * j + 1] = lam[j].imag(); res[j] = nr[j]; }
    inside_test(lam.data(), inside);
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    rr_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, ev_p0_, ev_p1_));
    proj_ms = ms;
    return 0;
}

// ---------------------------------------------------------------------------

}  // namespace feasthip
