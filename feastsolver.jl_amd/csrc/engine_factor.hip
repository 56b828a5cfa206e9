// Numeric factorisation of the shifted matrices T(z_k) of all local contour nodes (src/feast.jl:59-62, src/nlfeast.jl:33-45).
#include "engine_impl.hpp"

namespace feasthip {

void Engine::eval_coefs(const zc* pts, int count, std::vector<zc>& out) {
    out.assign((size_t)count * nterms, zc(0.0));
    if (coef_fn_) {
        coef_fn_(reinterpret_cast<const double*>(pts), count, reinterpret_cast<double*>(out.data()), coef_user_);
        return;
    }
    for (int i = 0; i < count; ++i) { zc zp = 1.0; for (int t = 0; t < nterms; ++t) { out[(size_t)i * nterms + t] = zp; zp *= pts[i]; } }
}

int Engine::set_coef_functions(feasthip_coef_fn fn, void* user) {
    coef_fn_ = fn; coef_user_ = user;
    factored_ = false;
    rebuild_shard();
    return 0;
}

int Engine::set_node_coefficients(const double* coef, int K_, int nterms_) {
    if (!coef) { node_coef_.clear(); factored_ = false; rebuild_shard(); return 0; }
    if (K_ != K || nterms_ != nterms) return fail(FEASTHIP_EARG, "set_node_coefficients: table must be K x nterms (set the contour and the terms first)");
    node_coef_.assign(reinterpret_cast<const zc*>(coef), reinterpret_cast<const zc*>(coef) + (size_t)K * nterms);
    factored_ = false;
    rebuild_shard();
    return 0;
}

int Engine::factorize() {
    if (rr_only_) return fail(FEASTHIP_ESTATE, "factorize: this handle runs the Rayleigh-Ritz half only (feasthip_set_rr_only)");
    if (!analyzed_) { int r = analyze(); if (r) return r; }
    HIPCHK(hipSetDevice(device_));
    // per-node coefficients scale_t * f_t(z_k) (f_t = z^t unless overridden) and pivot thresholds
    std::vector<zc> coef((size_t)Kloc * nterms), fz, zloc(Kloc);
    std::vector<double> tiny(Kloc);
    for (int kk = 0; kk < Kloc; ++kk) zloc[kk] = nodes_[local_[kk]];
    if (node_coef_.empty()) eval_coefs(zloc.data(), Kloc, fz);
    for (int kk = 0; kk < Kloc; ++kk) {
        double nrm = 0;
        for (int t = 0; t < nterms; ++t) {
            zc f = node_coef_.empty() ? fz[(size_t)kk * nterms + t] : node_coef_[(size_t)local_[kk] * nterms + t];
            coef[(size_t)kk * nterms + t] = terms_[t].scale * f;
            nrm += std::abs(coef[(size_t)kk * nterms + t]) * terms_[t].maxabs;
        }
        tiny[kk] = std::max(nrm * 2.2e-19, 1e-300);
    }
    HIPCHK(hipMemcpyAsync(d_coef_.p, coef.data(), coef.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(d_tiny_.p, tiny.data(), tiny.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemsetAsync(d_nperturb_.p, 0, Kloc * sizeof(int), stream_));
    HIPCHK(hipMemsetAsync(d_growth_.p, 0, 3 * Kloc * sizeof(int64_t), stream_));
    const int64_t fs = S.fact_size;
    const Tile* tl = d_tiles_.p;
    const int* ls = d_lists_.p;
    HIPCHK(hipStreamSynchronize(stream_));   // coef/tiny are stack-owned host buffers
    // Symmetric pencil (every coefficient matrix equals its transpose): elimination without row interchanges by groups of 128
    // columns (k_diag_grp / k_lscale_grp, U = D L^T), growth-checked afterwards; the trailing matrix and the contribution blocks
    // stay symmetric, so only their lower tile triangle is computed.
    const bool nopiv = use_nopiv();
    // Hermitian pencil (every coefficient matrix equals its conjugate transpose, but not its transpose): no interchanges either,
    // with L and U kept apart -- the pivoted schedule with its panel kernel replaced by k_diag32_gen / k_lscale32_gen and its
    // interchange passes dropped (kernels_factor.hip)
    const bool gen = !nopiv && use_nopiv_gen();
    if (nopiv || gen) {
        const size_t need = (size_t)Kloc * wstride_ * (4 * 1024 + 128);
        if (d_wscr_.n < need && !d_wscr_.alloc(need)) {
            // no room for the scratch of the elimination without interchanges: this handle keeps partial pivoting
            (void)hipGetLastError();
            if (nopiv) nopiv_failed_ = true; else nopiv_gen_failed_ = true;
            return factorize();
        }
    }
    const int sym_bits = nopiv ? 2 : 0;        // launch_gemm_nn: skip the U12 updates and the strictly upper tiles of the pivot block
    unsigned long long* growth_dev = reinterpret_cast<unsigned long long*>(d_growth_.p);
    unsigned long long* umax_dev = gen ? growth_dev + 2 * (size_t)Kloc : nullptr;
    auto panel = [&](hipStream_t s, FactorStep& st, bool tall, int level_ld) {
        if (gen) {
            launch_diag32_gen(s, dsym_, ls + st.panel.off, st.panel.count, st.k0, d_fact_.p, fs, d_piv_.p, d_tiny_.p, d_nperturb_.p, growth_dev, d_wscr_.p, wstride_, Kloc);
            launch_lscale32_gen(s, dsym_, tl + st.swap.off, st.swap.count, st.k0, d_fact_.p, fs, growth_dev, d_wscr_.p, wstride_, Kloc);
        } else
            launch_panel(s, dsym_, ls + st.panel.off, st.panel.count, st.k0, d_fact_.p, fs, d_piv_.p, d_tiny_.p, d_nperturb_.p, Kloc, tall, level_ld - st.k0);
    };
    HIPCHK(hipEventRecord(ev0_, stream_));
    if (dense) {
        std::vector<const cz*> ptrs(nterms);
        for (int t = 0; t < nterms; ++t) ptrs[t] = terms_[t].identity ? nullptr : terms_[t].d_dense.p;
        if (!d_matptr_.upload(ptrs.data(), nterms, stream_)) return fail(FEASTHIP_EHIP, "factorize: upload failed");
        launch_dense_assemble(stream_, d_fact_.p, fs, n, d_matptr_.p, nterms, d_coef_.p, Kloc);
    } else {
        HIPCHK(hipMemsetAsync(d_fact_.p, 0, (size_t)Kloc * fs * sizeof(cz), stream_));
        for (int t = 0; t < nterms; ++t)
            launch_scatter(stream_, d_fact_.p, fs, terms_[t].d_dest.p, terms_[t].d_cscval.p, terms_[t].colptr[n], d_coef_.p, nterms, t, Kloc);
    }
    bool fin_used = false;
    // (Round 5 built the conversion block column by block column NEXT to the factorisation of the fronts without update rows -- dense
    // matrices, the root --, with the left columns' interchanges on the side stream: commit c55dc22.  Slower: the conversion kernels are
    // latency chains per block column (0.45 + 0.6 + 0.45 ms for a 4096-row column) that only overlap each other when all columns of a
    // level go in one launch -- 41-54 ms of side-stream time on C2 instead of 13.7 -- and the coupling of the super blocks still waits
    // for the last column: C2 100 -> 103 ms, C3 79 -> 91 ms.)
    const bool skip_conversion = getenv("FEASTHIP_PROBE_SKIP_CONVERSION") != nullptr;
    // lazy conversion (engine.hpp): only where the primal sweeps alone read these factors, and no level splits its forward products along K
    bool lazy = lazy_conversion_ && !adj_fold_ && !dual_adj_ && !fwd_splitk_;
    for (const SolveLevel& V : splan_) if (V.f2split) lazy = false;
    // early forward sweep (engine.hpp): the caller's contour step follows at once and its right-hand sides exist already
    bool early = early_rhs_ && early_sweep_enabled_ && !dense && !dual_adj_ && !skip_conversion && S.maxdepth >= 2 &&
                 finalize_async_ && fin_stream_ && ev_fin_ && ev_fin_fork_ && have_x_;
    for (const SolveLevel& V : splan_) if (V.f2split) early = false;
    early_done_ = false;
    early_levels_ = 0;
    early_idrows_ = nopiv || gen;
    // complex64 sweeps (mixed_prec): the complex64 copy of a level's solve-ready fronts is made right behind the level's conversion on the
    // conversion stream (sparse matrices; one pass over the whole arena at the end otherwise: 18.6 GB, 3.5 ms on C3)
    if (mixed_ && !d_fact32_.alloc((size_t)Kloc * fs)) return fail(FEASTHIP_EHIP, "out of device memory for the complex64 factors");
    const bool f32_by_level = mixed_ && !dense && !skip_conversion && finalize_async_ && fin_stream_ && ev_fin_ && ev_fin_fork_;
    if (!f32_by_level && mixed_) early = false;
    // FEASTHIP_FACTOR_TRACE: main-stream time stamps after the extend-add, the panel steps and the Schur product of every level
    const bool ftrace = getenv("FEASTHIP_FACTOR_TRACE") != nullptr;
    std::vector<hipEvent_t> tev;
    auto stamp = [&]() { if (ftrace) { hipEvent_t e4; (void)hipEventCreate(&e4); (void)hipEventRecord(e4, stream_); tev.push_back(e4); } };
    stamp();
    for (int d = S.maxdepth; d >= 0; --d) {
        FactorLevel& F = fplan_[d];
        int par = d & 1;
        cz* cbp = d_cb_[par].p;
        int64_t cbs = std::max<int64_t>(1, S.cb_size[par]);
        launch_extend_add(stream_, dsym_, tl + F.extend.off, F.extend.count, d_fact_.p, fs, cbp, cbs, d_cb_[par ^ 1].p,
                          std::max<int64_t>(1, S.cb_size[par ^ 1]), Kloc, 0, nopiv ? 1 : 0);
        int level_ld = 0;
        for (int s2 : S.levels[d]) level_ld = std::max(level_ld, S.p(s2) + S.u(s2));
        const bool tall = level_ld >= 1024;      // wide panel workgroups for tall fronts (launch_panel picks the width)
        int level_p = 0;
        for (int s2 : S.levels[d]) level_p = std::max(level_p, S.p(s2));
        static const int la_min_p = getenv("FEASTHIP_LA_MINP") ? atoi(getenv("FEASTHIP_LA_MINP")) : 256;
        const bool lookahead = look_ahead_ && level_p >= la_min_p && la_stream_ && F.steps.size() >= 4;
        stamp();
        if (!lookahead && nopiv) {      // whole 128-column groups per step: diagonal block, rows below, one product
            for (size_t j = 0; j < F.steps.size(); j += 4) {
                FactorStep& st = F.steps[j];
                launch_diag_grp(stream_, dsym_, ls + st.panel.off, st.panel.count, st.k0, d_fact_.p, fs, d_piv_.p, d_tiny_.p, d_nperturb_.p, growth_dev, d_wscr_.p, wstride_, Kloc);
                launch_lscale_grp(stream_, dsym_, tl + st.lsg.off, st.lsg.count, st.k0, d_fact_.p, fs, growth_dev, d_wscr_.p, wstride_, Kloc);
                launch_gemm_nn(stream_, dsym_, tl + st.gemm.off, st.gemm.count, st.k0, d_fact_.p, fs, cbp, cbs, Kloc, ColSel(), 128, 0, nullptr, 0, sym_bits);
            }
        } else if (!lookahead) {
            for (FactorStep& st : F.steps) {
                panel(stream_, st, tall, level_ld);
                if (!gen) launch_swap(stream_, dsym_, tl + st.swap.off, st.swap.count, st.k0, d_fact_.p, fs, d_piv_.p, Kloc);
                launch_trsm_u(stream_, dsym_, tl + st.trsm.off, st.trsm.count, st.k0, d_fact_.p, fs, Kloc, ColSel(), umax_dev);
                launch_gemm_nn(stream_, dsym_, tl + st.gemm.off, st.gemm.count, st.k0, d_fact_.p, fs, cbp, cbs, Kloc);
            }
        } else {
            // LU with look-ahead for the big fronts at the top of the tree (few fronts x nodes: the panel factorisation runs on a
            // handful of workgroups and would otherwise leave the GPU idle between two trailing updates).  The panels are taken
            // in groups of group_panels_ (columns [K0_g, K0_g+1)).  The high-priority stream P factors the panels of group g,
            // updating only the group's own columns in between, and then applies the whole group, as one product of depth
            // K0_g+1 - K0_g, to the columns of group g+1.  Stream R (CU-masked so that the panel workgroups of P always find an
            // empty CU, see the constructor) applies group g to everything else (the left columns' row interchanges, the
            // pivot-block columns from K0_g+2 on and U12) while P is busy with group g+1.
            // Deferring a group's update is legal because all its row interchanges are applied first, to L and to C alike; the
            // U rows then come from one k_group_utrsm launch (or, for groups wider than 128 columns, from a blocked forward
            // substitution: a 32-row product with the panels before it, then the triangular solve with the panel's own L11) and
            // the rest is one product.  Against per-panel updates the product reads and writes C once per group instead of once
            // per panel, which is what bounds it.
            // Pivot-light path: the same pipeline with groups of 128 columns; P runs k_diag_grp + k_lscale_grp instead of the panels
            // (no interchanges, no U-row solves: U = D L^T is written with the L columns), the products skip the upper triangle.
            hipStream_t P = la_stream_, R = rest_stream_ ? rest_stream_ : stream_;
            HIPCHK(hipEventRecord(ev_la_[4], stream_));
            HIPCHK(hipStreamWaitEvent(P, ev_la_[4], 0));
            if (R != stream_) HIPCHK(hipStreamWaitEvent(R, ev_la_[4], 0));
            const int ns = (int)F.steps.size(), G = nopiv ? 4 : group_panels_, ng = (ns + G - 1) / G;
            const int INF = 1 << 30;
            auto K0 = [&](int g) { return g * G < ns ? F.steps[(size_t)g * G].k0 : INF; };       // first column of group g
            // all of group g (panels already factored) applied to the columns [c_lo, c_hi) of the pivot block, and to the left
            // columns [0, K0_g) and U12 when `others`
            auto apply_group = [&](hipStream_t s, int g, int c_lo, int c_hi, bool others) {
                const int j0 = g * G, j1 = std::min(ns, j0 + G), k0g = F.steps[j0].k0;
                auto sel = [&](int origin) {          // column offsets relative to the first column right of `origin`
                    ColSel c;
                    c.llo = 0; c.lhi = others ? k0g : 0;
                    c.clo = std::max(0, c_lo - origin); c.chi = c_hi >= INF ? INF : std::max(0, c_hi - origin);
                    c.u12 = others ? 1 : 0;
                    return c;
                };
                FactorStep& s0 = F.steps[j0];
                const int gw = F.steps[j1 - 1].k0 + FACT_NB - k0g;        // group width
                if (!nopiv && !gen) launch_swap(s, dsym_, tl + s0.swap.off, s0.swap.count, k0g, d_fact_.p, fs, d_piv_.p, Kloc, sel(k0g + gw), gw);
                if (nopiv) {
                    // the group's U rows are already there (D L^T, written with the L columns)
                } else if (s0.grp.count > 0 && fused_utrsm_) {       // U rows of the whole group in one launch
                    launch_group_utrsm(s, dsym_, tl + s0.grp.off, s0.grp.count, k0g, gw, d_fact_.p, fs, Kloc, sel(k0g + gw), umax_dev);
                } else {
                    for (int j = j0; j < j1; ++j) {
                        FactorStep& sj = F.steps[j];
                        const int bw = sj.k0 - k0g;
                        if (bw > 0) {
                            ColSel c = sel(sj.k0); c.llo = c.lhi = 0;
                            launch_gemm_nn(s, dsym_, tl + s0.gemm0.off, s0.gemm0.count, k0g, d_fact_.p, fs, cbp, cbs, Kloc, c, bw, FACT_NB);
                        }
                        ColSel c = sel(sj.k0 + FACT_NB);
                        launch_trsm_u(s, dsym_, tl + sj.trsm.off, sj.trsm.count, sj.k0, d_fact_.p, fs, Kloc, c, umax_dev);
                    }
                }
                // the next group's columns are the first two tile columns right of this group: the short list does
                const bool narrow = !others && c_hi - (k0g + gw) <= 128 && c_lo == k0g + gw;
                const Launch& gl = narrow ? s0.gemm_in : s0.gemm;
                launch_gemm_nn(s, dsym_, tl + gl.off, gl.count, k0g, d_fact_.p, fs, cbp, cbs, Kloc, sel(k0g + gw), gw, 0, nullptr, 0, sym_bits | (others ? 0 : 4));
            };
            for (int g = 0; g < ng; ++g) {
                const int j0 = g * G, j1 = std::min(ns, j0 + G), k0g = F.steps[j0].k0;
                const int kend = K0(g + 1);                     // first column after the group (INF for the last group)
                if (g > 0) {
                    if (g > 1) HIPCHK(hipStreamWaitEvent(P, ev_la_[2 + (g & 1)], 0));      // rest of group g-2 reaches these columns
                    apply_group(P, g - 1, k0g, kend, false);
                }
                if (nopiv) {
                    FactorStep& s0 = F.steps[j0];
                    launch_diag_grp(P, dsym_, ls + s0.panel.off, s0.panel.count, k0g, d_fact_.p, fs, d_piv_.p, d_tiny_.p, d_nperturb_.p, growth_dev, d_wscr_.p, wstride_, Kloc);
                    launch_lscale_grp(P, dsym_, tl + s0.lsg.off, s0.lsg.count, k0g, d_fact_.p, fs, growth_dev, d_wscr_.p, wstride_, Kloc);
                }
                for (int j = j0; j < j1 && !nopiv; ++j) {
                    FactorStep& sj = F.steps[j];
                    panel(P, sj, tall, level_ld);
                    // inside the group: interchanges on the group's earlier L columns and its remaining columns, U rows and update
                    // of the remaining columns
                    ColSel c;
                    c.llo = k0g; c.lhi = sj.k0; c.clo = 0; c.chi = kend >= INF ? INF : std::max(0, kend - sj.k0 - FACT_NB); c.u12 = 0;
                    if (!gen) launch_swap(P, dsym_, tl + sj.swap.off, sj.swap.count, sj.k0, d_fact_.p, fs, d_piv_.p, Kloc, c);
                    if (c.chi > 0) {
                        c.llo = c.lhi = 0;
                        launch_trsm_u(P, dsym_, tl + sj.trsm.off, sj.trsm.count, sj.k0, d_fact_.p, fs, Kloc, c, umax_dev);
                        {
                            // columns left in the group fit the first two tile columns unless the groups are wider than 160
                            const Launch& gl = (G - 1) * FACT_NB <= 128 ? sj.gemm_in : sj.gemm;
                            launch_gemm_nn(P, dsym_, tl + gl.off, gl.count, sj.k0, d_fact_.p, fs, cbp, cbs, Kloc, c);
                        }
                    }
                }
                HIPCHK(hipEventRecord(ev_la_[g & 1], P));
                HIPCHK(hipStreamWaitEvent(R, ev_la_[g & 1], 0));
                apply_group(R, g, K0(g + 2), INF, true);
                HIPCHK(hipEventRecord(ev_la_[2 + (g & 1)], R));
            }
            HIPCHK(hipEventRecord(ev_la_[5], P));
            HIPCHK(hipStreamWaitEvent(stream_, ev_la_[5], 0));
            if (R != stream_) { HIPCHK(hipEventRecord(ev_la_[6], R)); HIPCHK(hipStreamWaitEvent(stream_, ev_la_[6], 0)); }
        }
        stamp();
        // The children's contributions to the contribution block are pulled in by the product's epilogue where the product is long
        // enough to carry the extra dependent loads (pivot blocks of 128 columns and more: extend-add 39 -> 17 ms for +9 ms of
        // products on C3); the levels of small fronts keep the separate second extend-add pass.
        const bool fuse = fuse_extend_ && level_p >= fuse_min_p_;
        const Launch& cbl = nopiv ? F.cbgemm_sym : F.cbgemm;
        launch_gemm_nn(stream_, dsym_, tl + cbl.off, cbl.count, 0, d_fact_.p, fs, cbp, cbs, Kloc, ColSel(), FACT_NB, 0,
                       fuse ? d_cb_[par ^ 1].p : nullptr, std::max<int64_t>(1, S.cb_size[par ^ 1]), nopiv ? 1 : 0);
        stamp();
        // The solve-ready conversion of this level (row permutations, block inverses, panel scaling) only touches this
        // level's own panels, which nothing above reads (parents take the contribution blocks): it runs on a side stream
        // next to the factorisation of the levels above.  It must follow the Schur-complement product, which reads L21 / U12.
        hipStream_t fin = (finalize_async_ && fin_stream_ && ev_fin_ && ev_fin_fork_) ? fin_stream_ : stream_;
        if (skip_conversion) {      // timing probe: how much of the solve-ready conversion a factorisation exposes (the factors are left unusable)
            if (!fuse)
                launch_extend_add(stream_, dsym_, tl + F.extend1.off, F.extend1.count, d_fact_.p, fs, cbp, cbs, d_cb_[par ^ 1].p,
                                  std::max<int64_t>(1, S.cb_size[par ^ 1]), Kloc, 1, nopiv ? 1 : 0);
            continue;
        }
        if (fin != stream_) { HIPCHK(hipEventRecord(ev_fin_fork_, stream_)); HIPCHK(hipStreamWaitEvent(fin, ev_fin_fork_, 0)); fin_used = true; }
        // (without interchanges the row tables are the identity tables the handle holds anyway, d_rowid_: 2.9 ms of C3's conversion stream, 0.2 of them exposed at the root)
        if (!nopiv && !gen) launch_rowperm(fin, dsym_, ls + F.fronts.off, F.fronts.count, d_piv_.p, d_rowperm_.p, d_rowinv_.p, Kloc);
        if (nopiv) launch_couple32(fin, dsym_, tl + F.binv.off, F.binv.count, d_fact_.p, fs, Kloc);      // the 32 x 32 inverses are in place already
        else launch_block_inverse(fin, dsym_, tl + F.binv.off, F.binv.count, d_fact_.p, fs, Kloc);
        launch_couple_inverse(fin, dsym_, tl + F.couple.off, F.couple.count, d_fact_.p, fs, Kloc);
        launch_scale_panels(fin, dsym_, tl + F.scale.off, F.scale.count, d_fact_.p, fs, Kloc, lazy);
        for (int lvl = 0; lvl < 3; ++lvl) {       // 256-, 512- and 1024-column super blocks of the fronts with many pivot columns
            launch_couple_outer(fin, dsym_, tl + F.outer[lvl].off, F.outer[lvl].count, SOLVE_NB << lvl, d_fact_.p, fs, Kloc, lazy);
            launch_couple_level(fin, dsym_, tl + F.inner[lvl].off, F.inner[lvl].count, SOLVE_NB << lvl, d_fact_.p, fs, Kloc);
        }
        // ... and, in that stream's order, the forward sweep of this level (the levels below went the same way): it runs at the
        // conversion's low priority in the gaps of the levels above.  Measured on C3, store=false iterations per second: 11.45 without,
        // 11.87 so; 11.55 with all early levels enqueued at the end on a stream of their own behind per-level events (normal priority:
        // the factorisation slows by what the sweep takes), 11.72 on a low-priority stream of their own enqueued level by level (the
        // chain kernels' reserved CUs get occupied: root chain + 3 ms), 11.67 on a stream with rest_stream_'s CU mask.  (The conversions of
        // the three top levels on a second low-priority stream, so that the root's does not queue behind the early sweeps: store=false
        // 11.76 -> 11.5, a factorisation with the full conversion 75.8 -> 84.7 ms; with GPU_MAX_HW_QUEUES=8 11.8 -> 11.9 and 75.8 -> 80.6.)
        if (f32_by_level) {
            int level_ld2 = 0;
            for (int s2 : S.levels[d]) level_ld2 = std::max(level_ld2, S.p(s2) + S.u(s2));
            launch_to_f32_fronts(fin, dsym_, ls + F.fronts.off, F.fronts.count, (int)std::min<int64_t>(1024, ((int64_t)level_ld2 * level_ld2 + 4095) / 4096), d_fact_.p, d_fact32_.p, fs, Kloc);
        }
        if (early && d >= 1) {
            lazy_factors_ = lazy;
            int r = solve_range(fin, 0, Kloc, early_rhs_, nullptr, 1, d);
            if (r == 0 && adj_fold_) r = solve_range_adj(fin, 0, Kloc, Kloc, early_rhs_, ldw, 1, d);      // the mirror nodes' systems T(z_k)^H y = rhs
            if (r) return r;
        }
        if (!fuse)
            launch_extend_add(stream_, dsym_, tl + F.extend1.off, F.extend1.count, d_fact_.p, fs, cbp, cbs, d_cb_[par ^ 1].p,
                              std::max<int64_t>(1, S.cb_size[par ^ 1]), Kloc, 1, nopiv ? 1 : 0);
    }
    if (early) { early_done_ = true; early_levels_ = S.maxdepth; }      // (the join of the conversion stream below covers the early launches)
    if (fin_used) { HIPCHK(hipEventRecord(ev_fin_, fin_stream_)); HIPCHK(hipStreamWaitEvent(stream_, ev_fin_, 0)); }
    if (ftrace) {
        stamp();
        HIPCHK(hipStreamSynchronize(stream_));
        auto ms = [&](size_t a, size_t b) { float t = 0; (void)hipEventElapsedTime(&t, tev[a], tev[b]); return t; };
        size_t q = 0;
        for (int d = S.maxdepth; d >= 0; --d, q += 3) {
            int mp = 0, mu = 0;
            double fl_schur = 0, fl_steps = 0, by_ext = 0;
            // extend-add into this level (both phases): every child contribution entry is read once and added into the parent
            // (read-modify-write): 48 bytes per entry
            for (int s2 : S.levels[d])
                for (int64_t c = S.child_ptr[s2]; c < S.child_ptr[s2 + 1]; ++c) { double uc = S.u(S.child_idx[c]); by_ext += 48.0 * uc * uc * Kloc; }
            for (int s2 : S.levels[d]) {
                double p2 = S.p(s2), u2 = S.u(s2);
                mp = std::max(mp, S.p(s2)); mu = std::max(mu, S.u(s2));
                fl_schur += 8.0 * u2 * u2 * p2 * Kloc;
                fl_steps += 8.0 * (p2 * p2 * p2 / 3.0 + p2 * p2 * u2) * Kloc;      // LU of the pivot block, L21 and U12 (8 flops per complex multiply-add)
            }
            fprintf(stderr, "[feasthip factor] depth %2d  fronts %5zu  max p %5d  max u %5d  steps %3zu | extend-add %7.2f (%4.1f TB/s)  panel steps %7.2f (%5.1f TF)  schur product %7.2f ms (%5.1f TF)\n",
                    d, S.levels[d].size(), mp, mu, fplan_[d].steps.size(), ms(q, q + 1), by_ext / ms(q, q + 1) * 1e-9, ms(q + 1, q + 2), fl_steps / ms(q + 1, q + 2) * 1e-9, ms(q + 2, q + 3), fl_schur / ms(q + 2, q + 3) * 1e-9);
        }
        for (hipEvent_t e4 : tev) (void)hipEventDestroy(e4);
    }
    if (mixed_ && !f32_by_level) launch_to_f32(stream_, (int64_t)Kloc * fs, d_fact_.p, d_fact32_.p);      // complex64 copy of the solve-ready factors for the mixed-precision sweeps
    HIPCHK(hipEventRecord(ev1_, stream_));
    std::vector<int> np(Kloc);
    std::vector<int64_t> gr((size_t)3 * Kloc, 0);
    HIPCHK(hipMemcpyAsync(np.data(), d_nperturb_.p, Kloc * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (nopiv || gen) HIPCHK(hipMemcpyAsync(gr.data(), d_growth_.p, 3 * Kloc * sizeof(int64_t), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    perturbed = 0;
    for (int v : np) perturbed += v;
    nopiv_used = nopiv;
    nopiv_gen_used = gen;
    growth = 0;
    // What an elimination without interchanges is accepted on (both paths): the backward error of T = L U is bounded entry by entry
    // by k eps (|L| |U|)_ij <= k^2 eps max |l| max |u|, so the PRODUCT of the two maxima, relative to ||T(z_k)|| (tiny[kk] =
    // 2.2e-19 ||T(z_k)||), is what has to stay small -- round 3 tested each factor on its own against 1024, which admitted a backward
    // error near 1e-7.  growth = max over the units of max(1, max |l_ij|) * max(1, max |u_ij| / ||T||); accepted up to
    // FEASTHIP_NOPIV_GROWTH (256: a backward error of a few 1e-14 times the usual modest function of the front size, inside the 1e-11
    // the seam promises) and only without perturbed pivots.  max |u| is measured where the U entries are written (the diagonal-block
    // kernels' in-block rows and pivots, the transposed U = D L^T rows of k_lscale_grp, k_trsm_u / k_group_utrsm on the Hermitian path).
    auto growth_of = [&](int kk) {
        double l, d, u2;
        std::memcpy(&l, &gr[kk], sizeof(l)); std::memcpy(&d, &gr[(size_t)Kloc + kk], sizeof(d)); std::memcpy(&u2, &gr[(size_t)2 * Kloc + kk], sizeof(u2));
        const double urel = std::max(d, u2) * 2.2e-19 / tiny[kk];
        return std::max(1.0, l) * std::max(1.0, urel) * ((l == l && urel == urel) ? 1.0 : std::numeric_limits<double>::infinity());
    };
    if (gen) {
        for (int kk = 0; kk < Kloc; ++kk) growth = std::max(growth, growth_of(kk));
        if (!(growth <= nopiv_growth_bound()) || perturbed > 0) {
            if (getenv("FEASTHIP_FACTOR_TRACE")) fprintf(stderr, "[feasthip factor] elimination without interchanges rejected (growth %.3g, %lld perturbed pivots): refactorising with partial pivoting\n", growth, (long long)perturbed);
            nopiv_gen_failed_ = true;
            const double t_first = ms;
            int r = factorize();
            factor_ms += t_first;
            return r;
        }
    }
    if (nopiv) {
        for (int kk = 0; kk < Kloc; ++kk) growth = std::max(growth, growth_of(kk));
        // the elimination without interchanges was not harmless (a multiplier beyond the bound, or a pivot that had to be
        // perturbed): this matrix gets partial pivoting, now and on every later factorisation of this handle
        if (!(growth <= nopiv_growth_bound()) || perturbed > 0) {
            if (getenv("FEASTHIP_FACTOR_TRACE")) fprintf(stderr, "[feasthip factor] pivot-light path rejected (max |l| %.3g, %lld perturbed pivots): refactorising with partial pivoting\n", growth, (long long)perturbed);
            nopiv_failed_ = true;
            early_done_ = false; early_levels_ = 0;      // (the early forward launches have finished with the stream synchronisation above)
            const double t_first = ms;
            int r = factorize();
            factor_ms += t_first;
            return r;
        }
    }
    factor_ms = ms;
    lazy_factors_ = lazy;
    factored_ = !skip_conversion;
    return perturbed > 0 ? (int)std::min<int64_t>(perturbed, 0x7fffffff) : 0;
}

// ---------------------------------------------------------------------------

}  // namespace feasthip
