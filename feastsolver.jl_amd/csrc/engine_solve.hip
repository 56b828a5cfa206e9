// Block substitution over all local nodes, moment accumulation, contour step (src/feast.jl:59-71), stochastic count estimate.
#include "engine_impl.hpp"

namespace feasthip {

int Engine::solve_all(const cz* rhs) { return solve_all_impl(rhs, nullptr); }

// part 0: the whole sweep pair.  part 1: right-hand-side gather and forward sweep of the one tree level `only_level` >= 1 (early forward
// sweep, engine.hpp).  part 2: what is left after all of those -- the root's gather and forward level, and the backward sweep.
int Engine::solve_range(hipStream_t st, int kb, int kc, const cz* rhs, ProfileRow* prof, int part, int only_level) {
    if (mixed_) return solve_range_t<cf>(st, kb, kc, rhs, prof, d_fact32_.p, d_W32_.p, d_Y32_.p, d_crow32_[0].p, d_crow32_[1].p, d_sk32_.p, part, only_level);
    return solve_range_t<cz>(st, kb, kc, rhs, prof, d_fact_.p, d_W_.p, d_Y_.p, d_crow_[0].p, d_crow_[1].p, d_sk_.p, part, only_level);
}

template <typename T>
int Engine::solve_range_t(hipStream_t st, int kb, int kc, const cz* rhs, ProfileRow* prof, T* fact, T* Wall, T* Yall, T* crow0, T* crow1, T* skall, int part, int only_level) {
    const int ldw = solve_width();       // from here on the block vectors are the per-unit solve blocks (2 x Engine::ldw columns with the conjugate fold)
    const int ldrhs = this->ldw;
    int64_t fs = S.fact_size, ws = (int64_t)n * ldw;
    T* crowb[2] = {crow0, crow1};
    T* Fp = fact + (int64_t)kb * fs;
    T* Wp = Wall + (int64_t)kb * ws;
    T* Yp = Yall + (int64_t)kb * ws;
    // factors without interchanges take the identity tables, k_rowperm fills the others (part 1 runs inside a factorisation: its kind, not the last one's)
    const bool idrows = part == 1 ? early_idrows_ : (nopiv_used || nopiv_gen_used);
    const int* rperm = (idrows ? d_rowid_.p : d_rowperm_.p) + (int64_t)kb * n;
    const int* rinv = (idrows ? d_rowid_.p : d_rowinv_.p) + (int64_t)kb * n;
    const Tile* tl = d_tiles_.p;
    hipEvent_t pa = nullptr, pb = nullptr;
    if (prof) { HIPCHK(hipEventCreate(&pa)); HIPCHK(hipEventCreate(&pb)); }
    if (prof) prof_detail.clear();
    int cur_tiles = 0, cur_depth = -1;
    auto begin = [&]() { if (prof) (void)hipEventRecord(pa, st); };
    auto end = [&](int row, double bytes, double flops) {
        if (!prof) return;
        (void)hipEventRecord(pb, st);
        (void)hipEventSynchronize(pb);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, pa, pb);
        prof[row].ms += ms; prof[row].bytes += bytes; prof[row].flops += flops; prof[row].launches++;
        prof_detail.insert(prof_detail.end(), {(double)row, (double)ms, bytes, flops, (double)cur_tiles, (double)cur_depth});
    };
    double rowb = (double)ldw * sizeof(T);
    {   // right-hand sides into the fronts' pivot rows: all of them, or the rows of one tree level (part 1: level only_level, part 2: the root's)
        begin();
        launch_gather_rhs(st, dsym_, rhs, ldrhs, ldw, Wp, ws, rperm, kc, part == 0 ? -1 : part == 1 ? only_level : 0);
        end(0, 2.0 * n * rowb * kc, 0);
    }
    // Launch::bytes counts factor bytes, Launch::flops complex multiply-adds per block-vector column (one node)
    const double npu = fold_ ? 2.0 : 1.0;      // contour nodes per unit: algorithmic bytes count the factor once per NODE
    auto stat_bytes = [&](const Launch& L, double vec_rows) { return (L.bytes * (sizeof(T) / 16.0) * npu + vec_rows * rowb) * kc; };
    auto stat_flops = [&](const Launch& L) { return L.flops * 8.0 * ldw * kc; };
    for (int d = (part == 2 ? 0 : S.maxdepth); d >= (part == 1 ? 1 : 0); --d) {          // forward sweep, leaves to root
        SolveLevel& V = splan_[d];
        cur_depth = d;
        if (part == 1 && d != only_level) continue;          // (one level: in stream order behind that level's conversion)
        int par = d & 1;
        int64_t crs = std::max<int64_t>(1, S.crow_size[par]) * ldw, crsc = std::max<int64_t>(1, S.crow_size[par ^ 1]) * ldw;
        begin();
        launch_fwd_assemble(st, dsym_, tl + V.assemble.off, V.assemble.count, Wp, ws, ldw, (crowb[par ^ 1] + (int64_t)kb * crsc), crsc, rinv, kc,
                            dsym_.asmflag ? rhs : nullptr, ldrhs);
        if (V.assemble.count) end(1, 3.0 * V.asm_rows * rowb * kc, 0);
        double lvl_rows = 0;
        for (int s : S.levels[d]) lvl_rows += S.p(s) + S.u(s);
        for (Launch& L : V.f1) {
            begin();
            launch_apply(st, dsym_, tl + L.off, L.count, Fp, fs, Wp, Yp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs, kc, false, nullptr, 0, nullptr, 0, L.sl, &L.xp);
            if (L.count) { cur_tiles = L.count; end(3, stat_bytes(L, L.vrows), stat_flops(L)); }
        }
        begin();
        // contribution rows of the children that pass through to this level's contribution rows are pulled in the
        // epilogue of the F2a tiles (kind 1), so there is no second assembly pass
        const int64_t sksf = std::max<int64_t>(1, sk_rows_) * ldw;
        if (lazy_factors_) {
            // lazily converted factors: L21 is raw, so C_I = .. - L21 Y_J needs the solved block first -- the diagonal solves, then
            // the kind-1 tiles with Y in the place of W (they read their block-vector operand through the W pointer only)
            launch_apply(st, dsym_, tl + V.f2y.off, V.f2y.count, Fp, fs, Wp, Yp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs, kc, false,
                         nullptr, 0, (crowb[par ^ 1] + (int64_t)kb * crsc), crsc, V.f2y.sl, &V.f2y.xp);
            launch_apply(st, dsym_, tl + V.f2c.off, V.f2c.count, Fp, fs, Yp, Yp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs, kc, false,
                         nullptr, 0, (crowb[par ^ 1] + (int64_t)kb * crsc), crsc, V.f2c.sl, &V.f2c.xp);
        } else
        launch_apply(st, dsym_, tl + V.f2.off, V.f2.count, Fp, fs, Wp, Yp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs, kc, false,
                     V.f2split ? skall + (int64_t)kb * sksf : nullptr, V.f2split ? sksf : 0, (crowb[par ^ 1] + (int64_t)kb * crsc), crsc, V.f2.sl, &V.f2.xp);
        if (V.f2.count) { cur_tiles = V.f2.count; end(3, stat_bytes(V.f2, 2.0 * lvl_rows + V.pull_rows), stat_flops(V.f2)); }
        if (V.f2split) {
            begin();
            launch_splitk_reduce_fwd(st, dsym_, tl + V.f2red.off, V.f2red.count, V.f2split, (crowb[par] + (int64_t)kb * crs), crs, ldw,
                                     skall + (int64_t)kb * sksf, sksf, (crowb[par ^ 1] + (int64_t)kb * crsc), crsc, kc);
            double urows = 0;
            for (int s : S.levels[d]) urows += S.u(s);
            cur_tiles = V.f2red.count; end(1, (V.f2split + 1.0) * urows * rowb * kc, 0);
        }
    }
    for (int d = 0; d <= S.maxdepth && part != 1; ++d) {          // backward sweep, root to leaves
        SolveLevel& V = splan_[d];
        cur_depth = d;
        double lvl_p = 0, lvl_u = 0;
        for (int s : S.levels[d]) { lvl_p += S.p(s); lvl_u += S.u(s); }
        begin();
        int64_t sks = std::max<int64_t>(1, sk_rows_) * ldw;
        launch_apply(st, dsym_, tl + V.b1.off, V.b1.count, Fp, fs, Wp, Yp, ws, ldw, nullptr, 0, kc, true, skall + (int64_t)kb * sks, sks, nullptr, 0, V.b1.sl, &V.b1.xp);
        if (V.b1split) launch_splitk_reduce(st, dsym_, tl + V.b1red.off, V.b1red.count, V.b1split, Yp, ws, ldw, skall + (int64_t)kb * sks, sks, kc);
        if (V.b1.count) {
            Launch b1s = V.b1;          // the kind-6 tiles carry their fronts' B3 products
            double rows = lvl_u + 2.0 * lvl_p;          // algorithmic row transfers stay what the step needs, fused or not
            b1s.bytes += V.b3s.bytes; b1s.flops += V.b3s.flops; rows += V.b3s.vrows;
            cur_tiles = -V.b1.count; end(3, stat_bytes(b1s, rows), stat_flops(b1s));
        }
        for (int c = (int)V.b2.size() - 1; c >= 1; --c) {
            Launch& L = V.b2[c];
            begin();
            launch_apply(st, dsym_, tl + L.off, L.count, Fp, fs, Wp, Yp, ws, ldw, nullptr, 0, kc, false, nullptr, 0, nullptr, 0, L.sl, &L.xp);
            if (L.count) { cur_tiles = L.count; end(3, stat_bytes(L, L.vrows), stat_flops(L)); }
        }
        begin();
        launch_apply(st, dsym_, tl + V.b3.off, V.b3.count, Fp, fs, Wp, Yp, ws, ldw, nullptr, 0, kc, false, nullptr, 0, nullptr, 0, V.b3.sl, &V.b3.xp);
        if (V.b3.count) { cur_tiles = V.b3.count; end(3, stat_bytes(V.b3, V.b3.vrows), stat_flops(V.b3)); }
    }
    if (prof) { (void)hipEventDestroy(pa); (void)hipEventDestroy(pb); }
    HIPCHK(hipGetLastError());
    return 0;
}

// Adjoint sweeps: T(z_k)^H x = rhs for the units kb .. kb + kc - 1 over their solve-ready factors (kernels.hpp, tile kinds 8 .. 13).
// Forward with U^H (a lower triangular factor), backward with L^H; the fronts' row interchanges act on the solution, so the gather and
// the assembly run with identity interchanges (d_rowid_) and k_accumulate undoes them at the end.  Per level:
//   forward   assemble -> { Y_b = inv(U_bb)^H W_b } -> steps c = 1 .. S - 1 { Y_c -= (PU_[0,c) c)^H Y_[0,c) } -> { C_I = .. - U12^H Y_J ; W_b = inv(L_bb)^H Y_b }
//   backward  { W_J -= P21^H W[struct] } (+ split-K reduction) -> steps c = S - 2 .. 0 { W_c -= (P_(c,S) c)^H W_(c,S) }
int Engine::solve_range_adj(hipStream_t st, int kb, int kc, int slot0, const cz* rhs, int ldrhs, int part, int only_level) {
    if (mixed_) return fail(FEASTHIP_ESTATE, "adjoint sweeps run in complex128 only");
    if (lazy_factors_) return fail(FEASTHIP_ESTATE, "adjoint sweeps need fully converted factors (feasthip_set_lazy_conversion(h, 0) and factorise again)");
    if (fwd_splitk_) return fail(FEASTHIP_ESTATE, "adjoint sweeps: FEASTHIP_FWD_SPLITK is not supported");
    const int lw = solve_width();
    const int64_t fs = S.fact_size, ws = (int64_t)n * lw;
    cz* crowb[2] = {d_crow_[0].p, d_crow_[1].p};
    const cz* Fp = d_fact_.p + (int64_t)kb * fs;
    cz* Wp = d_W_.p + (int64_t)slot0 * ws;
    cz* Yp = d_Y_.p + (int64_t)slot0 * ws;
    const int* rid = d_rowid_.p;                 // (every unit's copy is the same identity table)
    const Tile* tl = d_tiles_.p;
    launch_gather_rhs(st, dsym_, rhs, ldrhs, lw, Wp, ws, rid, kc, part == 0 ? -1 : part == 1 ? only_level : 0);
    for (int d = (part == 2 ? 0 : S.maxdepth); d >= (part == 1 ? 1 : 0); --d) {          // forward sweep with U^H, leaves to root
        if (part == 1 && d != only_level) continue;
        SolveLevel& V = splan_[d];
        const int par = d & 1;
        const int64_t crs = std::max<int64_t>(1, S.crow_size[par]) * lw, crsc = std::max<int64_t>(1, S.crow_size[par ^ 1]) * lw;
        cz* crp = crowb[par] + (int64_t)slot0 * crs;
        cz* crc = crowb[par ^ 1] + (int64_t)slot0 * crsc;
        launch_fwd_assemble(st, dsym_, tl + V.assemble.off, V.assemble.count, Wp, ws, lw, crc, crsc, rid, kc, dsym_.asmflag ? rhs : nullptr, ldrhs);
        launch_apply(st, dsym_, tl + V.afd.off, V.afd.count, Fp, fs, Wp, Yp, ws, lw, crp, crs, kc, false, nullptr, 0, nullptr, 0, V.afd.sl, &V.afd.xp);
        for (size_t c = 1; c < V.afe.size(); ++c) {
            const Launch& L = V.afe[c];
            launch_apply(st, dsym_, tl + L.off, L.count, Fp, fs, Wp, Yp, ws, lw, crp, crs, kc, false, nullptr, 0, nullptr, 0, L.sl, &L.xp);
        }
        launch_apply(st, dsym_, tl + V.af2.off, V.af2.count, Fp, fs, Wp, Yp, ws, lw, crp, crs, kc, false, nullptr, 0, crc, crsc, V.af2.sl, &V.af2.xp);
    }
    const int64_t sks = std::max<int64_t>(1, sk_rows_) * lw;
    cz* skp = d_sk_.p + (int64_t)slot0 * sks;
    // With row interchanges in the fronts (T = P^T L U, T^H = U^H L^H P) a front's finished rows are x_J = P_J^T w_J: they are copied to
    // their pre-pivot positions in Y (free since the forward sweep ended), which is what the children gather from and what the caller reads.
    const bool unperm = !(nopiv_used || nopiv_gen_used);
    cz* Xs = unperm ? Yp : Wp;
    for (int d = 0; d <= S.maxdepth && part != 1; ++d) {          // backward sweep with L^H, root to leaves
        SolveLevel& V = splan_[d];
        launch_apply(st, dsym_, tl + V.ab1.off, V.ab1.count, Fp, fs, Wp, Xs, ws, lw, nullptr, 0, kc, true, skp, sks, nullptr, 0, V.ab1.sl, &V.ab1.xp);
        if (V.b1split) launch_splitk_reduce(st, dsym_, tl + V.b1red.off, V.b1red.count, V.b1split, Wp, ws, lw, skp, sks, kc);
        for (int c = (int)V.abe.size() - 2; c >= 0; --c) {
            const Launch& L = V.abe[c];
            launch_apply(st, dsym_, tl + L.off, L.count, Fp, fs, Wp, Yp, ws, lw, nullptr, 0, kc, false, nullptr, 0, nullptr, 0, L.sl, &L.xp);
        }
        if (unperm) launch_adj_unpermute(st, dsym_, d_lists_.p + fplan_[d].fronts.off, fplan_[d].fronts.count, d_rowperm_.p + (int64_t)kb * n, Wp, Yp, ws, lw, kc);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// Q0 (Q1) = sum_k T_k .* e0[k,:] (e1[k,:]) over the solution blocks of the last sweep (fp64 accumulation in either precision)
void Engine::accumulate_solutions(int K_, const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1, int row0, int nrows) {
    const int lw = solve_width();
    if (mixed_) launch_accumulate(stream_, dsym_, d_X_.p, d_W32_.p, (int64_t)n * lw, ldw, lw, K_, e0, e1, mode, Q0, Q1, row0, nrows);
    else launch_accumulate(stream_, dsym_, d_X_.p, d_W_.p, (int64_t)n * lw, ldw, lw, K_, e0, e1, mode, Q0, Q1, row0, nrows);
    if (adj_fold_ && K_ == Kloc)       // the mirror nodes: solutions of the adjoint instances, their rows of the coefficient tables, added
        launch_accumulate(stream_, dsym_, d_X_.p, adj_solution() + (int64_t)Kloc * n * lw, (int64_t)n * lw, ldw, lw, Kloc, e0 + (size_t)Kloc * lw,
                          e1 ? e1 + (size_t)Kloc * lw : nullptr, mode, Q0, Q1, row0, nrows, 1);
}

// Node groups that run on their own streams.  Two groups overlap the launch tails and the latency-bound small
// launches of one group with the large launches of the other when there are many nodes; with few local nodes
// (multi-GPU shards) or with the 128-column units of the conjugate fold one fuller launch per step is faster
// (measured: 16 unfolded nodes 2 groups; <= 8 nodes, or folded units, 1 group).
int Engine::stream_groups() const {
    int ns = std::min<int>((int)aux_streams_.size(), Kloc);
    if (!streams_forced_ && (nodes_local() < 12 || fold_)) ns = 1;      // folded units are already 128 columns wide: one node group (measured 17.9 vs 18.2 ms)
    return std::max(1, ns);
}

int Engine::solve_all_impl(const cz* rhs, ProfileRow* prof) {
    if (!factored_) return fail(FEASTHIP_ESTATE, "solve: factorize first");
    const int part = early_done_ ? 2 : 0;      // 2: factorize() has enqueued the gather and the forward levels below the root already (engine.hpp)
    early_done_ = false;
    int ns = stream_groups();
    if (adj_fold_) {
        // every unit is solved twice over its one factorisation: T(z_k) y = rhs by the primal sweeps, T(conj z_k) y = T(z_k)^H y = rhs
        // by the adjoint ones (instances Kloc .. 2 Kloc - 1), on two streams when the handle has them
        hipStream_t sa = (!prof && aux_streams_.size() >= 2) ? aux_streams_[0] : stream_, sb = (!prof && aux_streams_.size() >= 2) ? aux_streams_[1] : stream_;
        if (sa != stream_) {
            HIPCHK(hipEventRecord(ev_fork_, stream_));
            HIPCHK(hipStreamWaitEvent(sa, ev_fork_, 0));
            HIPCHK(hipStreamWaitEvent(sb, ev_fork_, 0));
        }
        int r = solve_range(sa, 0, Kloc, rhs, prof, part);
        if (r == 0) r = solve_range_adj(sb, 0, Kloc, Kloc, rhs, ldw, part);
        if (r) return r;
        if (sa != stream_) {
            HIPCHK(hipEventRecord(ev_join_[0], sa)); HIPCHK(hipStreamWaitEvent(stream_, ev_join_[0], 0));
            HIPCHK(hipEventRecord(ev_join_[1], sb)); HIPCHK(hipStreamWaitEvent(stream_, ev_join_[1], 0));
        }
        return 0;
    }
    if (prof || ns <= 1) return solve_range(stream_, 0, Kloc, rhs, prof, part);
    // The nodes are independent: split them over a few streams so that the tails and the latency-bound small
    // launches of one group overlap with the large launches of another.
    HIPCHK(hipEventRecord(ev_fork_, stream_));
    int per = (Kloc + ns - 1) / ns;
    for (int g = 0; g < ns; ++g) {
        int kb = g * per, kc = std::min(per, Kloc - kb);
        if (kc <= 0) break;
        HIPCHK(hipStreamWaitEvent(aux_streams_[g], ev_fork_, 0));
        int r = solve_range(aux_streams_[g], kb, kc, rhs, nullptr, part);
        if (r) return r;
        HIPCHK(hipEventRecord(ev_join_[g], aux_streams_[g]));
        HIPCHK(hipStreamWaitEvent(stream_, ev_join_[g], 0));
    }
    return 0;
}

double Engine::solve_bytes() const {
    // SURVEY 8d, per contour NODE: factor entries streamed once per node and sweep pair + block-vector traffic
    double rowb = (double)ldw * 16.0;
    // contribution rows: written once by their front; then either pulled by the parent's F2a epilogue (1 read) or
    // assembled into the parent's pivot block (read + read-modify-write)
    double asm_rows = 0, pull_rows = 0;
    for (const SolveLevel& V : splan_) { asm_rows += V.asm_rows; pull_rows += V.pull_rows; }
    double es = mixed_ ? 0.5 : 1.0;     // complex64 factors and block vectors in the mixed-precision sweeps
    double per_node = (double)S.fact_size * 16.0 * es + (double)S.sum_u * 4.0 + (4.0 * n + (double)S.sum_u + pull_rows + 3.0 * asm_rows) * rowb * es;
    return per_node * nodes_local() + 3.0 * n * rowb;
}

double Engine::solve_bytes_streamed() const {
    // what the sweeps move: with the conjugate fold the factor (and the index arrays) of a pair is read once for both nodes
    double es = mixed_ ? 0.5 : 1.0;
    return solve_bytes() - (fold_ ? ((double)S.fact_size * 16.0 * es + (double)S.sum_u * 4.0) * Kloc : 0.0);
}

double Engine::solve_flops() const {
    return ((double)(S.nnz_l + S.nnz_u) * 8.0 * ldw) * nodes_local() + 16.0 * n * (double)ldw * nodes_local();
}

int Engine::contour_step(int refactor) {
    int r = contour_step_begin(refactor, 1);
    if (r < 0) return r;
    int r2 = contour_step_end();
    return r2 ? r2 : r;
}

// The contour step without its final host synchronisation (multi-GPU driver): everything is enqueued, the moment is produced in
// `nchunks` contiguous row ranges of Q (original row order) with one event each, so that the caller can hand a finished range to
// the all-reduce while the later ranges are still being accumulated (moment_chunk_wait), then contour_step_end().
int Engine::contour_step_begin(int refactor, int nchunks) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "contour_step: call set_x and rr_step first");
    if (step_open_) return fail(FEASTHIP_ESTATE, "contour_step_begin: the previous step was not ended");
    HIPCHK(hipSetDevice(device_));
    int warn = 0;
    if (refactor || !factored_) {
        early_rhs_ = d_R_.p;      // the right-hand sides exist already: the forward sweep starts under the factorisation (engine.hpp)
        int r = factorize();
        early_rhs_ = nullptr;
        if (r < 0) { early_done_ = false; return r; }
        warn = r;
    }
    std::vector<zc> e0 = node_table([&](int k, int j, int) { return weights_[k] / (nodes_[k] - lambda_[j]); });      // src/feast.jl:60,69
    HIPCHK(h2d_async(5, d_e0_.p, e0.data(), e0.size() * sizeof(zc)));
    // the table sits in pinned slot 5, which nothing else uses and which is rewritten only after the synchronisations of the next
    // Rayleigh-Ritz step: no host wait here (one wake-up of the host thread less per iteration); without a slot the copy reads e0 itself
    if (!pin_slot(5, e0.size() * sizeof(zc))) HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int r = solve_all(d_R_.p);                                                                // :61-66
    if (r) return r;
    if (comm_ && comm_world_ > 1 && nchunks < 4) nchunks = 4;      // ranges to overlap the all-reduce of one with the accumulation of the next
    nchunks = std::max(1, std::min(nchunks, std::min(64, n)));
    while ((int)ev_chunk_.size() < nchunks) { hipEvent_t e = nullptr; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_chunk_.push_back(e); }
    chunk_rows_.assign(nchunks + 1, 0);
    for (int c = 0; c <= nchunks; ++c) chunk_rows_[c] = (int)((int64_t)n * c / nchunks);
    for (int c = 0; c < nchunks; ++c) {
        accumulate_solutions(Kloc, d_e0_.p, nullptr, 1, Qp(), nullptr, chunk_rows_[c], chunk_rows_[c + 1] - chunk_rows_[c]);     // :68-70
        if (nchunks > 1 || comm_) HIPCHK(hipEventRecord(ev_chunk_[c], stream_));
        if (comm_) {      // sum over the ranks, in place, behind this range's accumulation (src/feast.jl:59-71: the node loop is what is sharded)
            HIPCHK(hipStreamWaitEvent(comm_stream_, ev_chunk_[c], 0));
            if (c == 0) HIPCHK(hipEventRecord(ev_comm0_, comm_stream_));
            int r2 = comm_allreduce_rows(chunk_rows_[c], chunk_rows_[c + 1] - chunk_rows_[c]);
            if (r2) return r2;
        }
    }
    if (comm_) {
        HIPCHK(hipEventRecord(ev_comm1_, comm_stream_));
        HIPCHK(hipStreamWaitEvent(stream_, ev_comm1_, 0));
    }
    HIPCHK(hipEventRecord(ev1_, stream_));
    step_open_ = true;
    return warn;
}

int Engine::moment_chunk_wait(int c, int64_t* row0, int64_t* rows) {
    if (!step_open_) return fail(FEASTHIP_ESTATE, "moment_chunk_wait: no open contour step");
    const int nchunks = (int)chunk_rows_.size() - 1;
    if (c < 0 || c >= nchunks) return fail(FEASTHIP_EARG, "moment_chunk_wait: chunk out of range");
    HIPCHK(hipSetDevice(device_));
    if (nchunks > 1) HIPCHK(hipEventSynchronize(ev_chunk_[c])); else HIPCHK(hipEventSynchronize(ev1_));
    if (row0) *row0 = chunk_rows_[c];
    if (rows) *rows = chunk_rows_[c + 1] - chunk_rows_[c];
    return 0;
}

int Engine::contour_step_end() {
    if (!step_open_) return fail(FEASTHIP_ESTATE, "contour_step_end: no open contour step");
    step_open_ = false;
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipEventSynchronize(ev1_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    solve_ms = ms;
    if (comm_) { HIPCHK(hipEventElapsedTime(&ms, ev_comm0_, ev_comm1_)); allreduce_ms = ms; }
    return 0;
}

// contour_estimate_eig (src/stochastic.jl:2-33): est = real( sum_k w_k tr(X^H (z_k B - A)^-1 X) ) / samples.
// The engine factorises T(z) = A - z B, so (z_k B - A)^-1 X = -Y_k; the node sum is the usual accumulation
// kernel with weights -w_k and the trace comes from the MFMA Gram kernel.  Local nodes only (partial sum).
int Engine::estimate_count(const void* X, int64_t n_, int m0_, double* est) {
    if (!est) return fail(FEASTHIP_EARG, "estimate_count: null output");
    int r = set_x(X, n_, m0_);
    if (r) return r;
    if (!factored_) { r = factorize(); if (r < 0) return r; }
    HIPCHK(hipSetDevice(device_));
    std::vector<zc> e0 = node_table([&](int k, int, int) { return -weights_[k]; });
    HIPCHK(hipMemcpyAsync(d_e0_.p, e0.data(), e0.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    r = solve_all(d_X_.p);
    if (r) return r;
    accumulate_solutions(Kloc, d_e0_.p, nullptr, 0, d_T1_.p, nullptr);
    cz* dG = d_small_.p;
    launch_gram(stream_, n, m0, ldw, d_X_.p, d_T1_.p, d_gram_scratch_.p, dG);
    std::vector<zc> G((size_t)m0 * m0);
    HIPCHK(hipMemcpyAsync(G.data(), dG, G.size() * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    zc tr = 0;
    for (int j = 0; j < m0; ++j) tr += G[j + (size_t)j * m0];
    *est = tr.real() / (double)m0;
    return 0;
}

int Engine::profile_solve(ProfileRow* rows) {
    if (!have_x_ || !factored_) return fail(FEASTHIP_ESTATE, "profile_solve: needs a factorised problem and iteration state");
    HIPCHK(hipSetDevice(device_));
    for (int i = 0; i < 8; ++i) rows[i] = ProfileRow();
    int r = solve_all_impl(d_R_.p, rows);
    if (r) return r;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, stream_));
    accumulate_solutions(Kloc, d_e0_.p, nullptr, 1, d_T2_.p, nullptr);
    HIPCHK(hipEventRecord(b, stream_));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    rows[4].ms = ms; rows[4].launches = 1; rows[4].bytes = ((double)nodes_local() + 2.0) * n * ldw * 16.0;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 0;
}

// ---------------------------------------------------------------------------
// NLFEAST  (src/nlfeast.jl:2-84)
// ---------------------------------------------------------------------------

}  // namespace feasthip
