// See engine.hpp.  Reference semantics: src/feast.jl:10-80, 89-156;
// src/nlfeast.jl:2-84; src/utils.jl:69-77, 104-116, 151-179.
#include "engine.hpp"

#include <algorithm>
#include <limits>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/feasthip.h"
#include "smalleig.hpp"

namespace feasthip {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

template <class T> void DevBuf<T>::release() { if (p) { (void)hipFree((void*)p); p = nullptr; n = 0; } }
template <class T> bool DevBuf<T>::alloc(size_t count) {
    if (count <= n && p) return true;
    release();
    if (count == 0) count = 1;
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return false; }
    p = (T*)q;
    n = count;
    return true;
}
template <class T> bool DevBuf<T>::upload(const T* h, size_t count, hipStream_t st) {
    if (!alloc(count)) return false;
    if (count == 0) return true;
    if (hipMemcpyAsync((void*)p, h, count * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess) return false;
    return hipStreamSynchronize(st) == hipSuccess;
}
template struct DevBuf<cz>;
template struct DevBuf<int>;
template struct DevBuf<int64_t>;
template struct DevBuf<double>;
template struct DevBuf<FrontD>;
template struct DevBuf<Tile>;
template struct DevBuf<const cz*>;

static inline cz to_cz(zc v) { return cz_make(v.real(), v.imag()); }

Engine::Engine(int device) : device_(device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { (void)hipGetLastError(); fail(FEASTHIP_ENOGPU, "no HIP device visible: feasthip has no CPU fallback"); return; }
    if (device < 0 || device >= cnt) { fail(FEASTHIP_EARG, "device id out of range"); return; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(FEASTHIP_EHIP, "hipGetDeviceProperties failed"); return; }
    if (std::strstr(prop.gcnArchName, "gfx950") == nullptr) { fail(FEASTHIP_ENOGPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only"); return; }
    if (hipSetDevice(device) != hipSuccess) { fail(FEASTHIP_EHIP, "hipSetDevice failed"); return; }
    if (hipStreamCreate(&stream_) != hipSuccess) { fail(FEASTHIP_EHIP, "hipStreamCreate failed"); return; }
    own_stream_ = true;
    (void)hipEventCreate(&ev0_);
    (void)hipEventCreate(&ev1_);
    if (const char* v = getenv("FEASTHIP_LOOKAHEAD")) look_ahead_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_GROUP")) group_panels_ = std::max(1, std::min(8, atoi(v)));
    if (const char* v = getenv("FEASTHIP_FIN_ASYNC")) finalize_async_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_MIXED")) mixed_ = atoi(v) != 0;      // tuning / test knob; the API is feasthip_set_precision
    int nst = 2;
    if (const char* v = getenv("FEASTHIP_STREAMS")) { nst = std::max(1, std::min(8, atoi(v))); streams_forced_ = true; }
    if (nst > 1) {
        (void)hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming);
        for (int i = 0; i < nst; ++i) {
            hipStream_t s2 = nullptr; hipEvent_t e2 = nullptr;
            if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) break;
            (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
            aux_streams_.push_back(s2); ev_join_.push_back(e2);
        }
    }
    {   // look-ahead stream of the factorisation: its small panel kernels must be dispatched ahead of the trailing updates
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        if (hipStreamCreateWithPriority(&la_stream_, hipStreamNonBlocking, phi) != hipSuccess) la_stream_ = nullptr;
        for (int i = 0; i < 7; ++i) { hipEvent_t e3 = nullptr; (void)hipEventCreateWithFlags(&e3, hipEventDisableTiming); ev_la_.push_back(e3); }
        // A panel workgroup (1024 threads, 126 registers) needs an empty CU: behind a trailing update that fills every CU it
        // would only start when that update drains.  The trailing updates of the look-ahead levels therefore run on a stream
        // whose CU mask leaves reserve_cus_ CUs of every XCD alone (MI355X: 8 XCDs x 32 CUs).  The cleared bits are 32x + 8m + x
        // (x = XCD, m < reserve_cus_): two free CUs per XCD whether the mask bits enumerate the CUs XCD by XCD or round-robin.
        if (const char* v = getenv("FEASTHIP_RESERVE_CUS")) reserve_cus_ = std::max(0, std::min(4, atoi(v)));
        hipDeviceProp_t prop;
        if (la_stream_ && reserve_cus_ > 0 && hipGetDeviceProperties(&prop, device_) == hipSuccess && prop.multiProcessorCount == 256) {
            uint32_t mask[8];
            for (int x = 0; x < 8; ++x) {
                mask[x] = 0xffffffffu;
                for (int m = 0; m < reserve_cus_; ++m) mask[x] &= ~(1u << (8 * m + x));
            }
            if (hipExtStreamCreateWithCUMask(&rest_stream_, 8, mask) != hipSuccess) { rest_stream_ = nullptr; (void)hipGetLastError(); }
        }
    }
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (hipStream_t s2 : aux_streams_) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
    for (hipEvent_t e2 : ev_join_) (void)hipEventDestroy(e2);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    for (hipEvent_t e3 : ev_la_) if (e3) (void)hipEventDestroy(e3);
    if (la_stream_) { (void)hipStreamSynchronize(la_stream_); (void)hipStreamDestroy(la_stream_); }
    if (rest_stream_) { (void)hipStreamSynchronize(rest_stream_); (void)hipStreamDestroy(rest_stream_); }
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

int Engine::hip_fail(hipError_t e, const char* what) {
    (void)hipGetLastError();
    return fail(FEASTHIP_EHIP, std::string("HIP error '") + hipGetErrorString(e) + "' in " + what);
}

int Engine::set_stream(void* s) {
    if (own_stream_ && stream_) { (void)hipStreamSynchronize(stream_); (void)hipStreamDestroy(stream_); }
    stream_ = (hipStream_t)s;
    own_stream_ = false;
    return 0;
}

static zc read_val(const void* p, int type, int64_t idx) {
    switch (type) {
        case FEASTHIP_F64: return zc(((const double*)p)[idx], 0.0);
        case FEASTHIP_C128: return zc(((const double*)p)[2 * idx], ((const double*)p)[2 * idx + 1]);
        default: return zc((double)((const int64_t*)p)[idx], 0.0);
    }
}

int Engine::set_dense(int64_t n_, int nterms_, const void* const* mats, const int* types, const double* scale) {
    if (n_ <= 0 || n_ > 65536 || nterms_ < 1 || nterms_ > 16 || !mats) return fail(FEASTHIP_EARG, "set_dense: bad n or nterms");
    HIPCHK(hipSetDevice(device_));
    n = (int)n_; nterms = nterms_; dense = true;
    analyzed_ = factored_ = have_x_ = false;
    terms_.clear();
    terms_.resize(nterms);
    std::vector<zc> tmp;
    for (int t = 0; t < nterms; ++t) {
        SparseTerm& T = terms_[t];
        T.scale = scale ? zc(scale[2 * t], scale[2 * t + 1]) : zc(1.0);
        if (!mats[t]) { T.identity = true; T.maxabs = 1.0; continue; }
        int ty = types ? types[t] : FEASTHIP_C128;
        if (ty < 0 || ty > 2) return fail(FEASTHIP_EARG, "set_dense: bad element type");
        size_t nn = (size_t)n * n;
        tmp.resize(nn);
        double mx = 0;
        for (size_t e = 0; e < nn; ++e) { tmp[e] = read_val(mats[t], ty, (int64_t)e); mx = std::max(mx, std::abs(tmp[e])); }
        T.maxabs = mx;
        if (!T.d_dense.upload((const cz*)tmp.data(), nn, stream_)) return fail(FEASTHIP_EHIP, "set_dense: device allocation/upload failed");
    }
    return 0;
}

int Engine::set_csc(int64_t n_, int nterms_, const int64_t* const* colptr, const int64_t* const* rowval,
                    const void* const* nzval, const int* types, const double* scale, int base) {
    if (n_ <= 0 || n_ > 0x7fffffff / 2 || nterms_ < 1 || nterms_ > 16 || !colptr || !rowval || !nzval)
        return fail(FEASTHIP_EARG, "set_csc: bad n or nterms");
    HIPCHK(hipSetDevice(device_));
    n = (int)n_; nterms = nterms_; dense = false;
    analyzed_ = factored_ = have_x_ = false;
    terms_.clear();
    terms_.resize(nterms);
    for (int t = 0; t < nterms; ++t) {
        SparseTerm& T = terms_[t];
        T.scale = scale ? zc(scale[2 * t], scale[2 * t + 1]) : zc(1.0);
        T.colptr.resize(n + 1);
        if (!colptr[t]) {   // identity, stored as an explicit diagonal
            T.identity = true;
            T.rowval.resize(n);
            T.val.assign(n, zc(1.0));
            for (int j = 0; j <= n; ++j) T.colptr[j] = j;
            for (int j = 0; j < n; ++j) T.rowval[j] = j;
            T.maxabs = 1.0;
            continue;
        }
        int ty = types ? types[t] : FEASTHIP_C128;
        if (ty < 0 || ty > 2) return fail(FEASTHIP_EARG, "set_csc: bad element type");
        for (int j = 0; j <= n; ++j) T.colptr[j] = colptr[t][j] - base;
        int64_t nnz = T.colptr[n];
        if (T.colptr[0] != 0 || nnz < 0) return fail(FEASTHIP_EARG, "set_csc: colptr[0] != index_base");
        T.rowval.resize(nnz);
        T.val.resize(nnz);
        double mx = 0;
        for (int j = 0; j < n; ++j) {
            if (T.colptr[j + 1] < T.colptr[j]) return fail(FEASTHIP_EARG, "set_csc: colptr not monotone");
            for (int64_t e = T.colptr[j]; e < T.colptr[j + 1]; ++e) {
                int64_t r = rowval[t][e] - base;
                if (r < 0 || r >= n) return fail(FEASTHIP_EDIM_A, "Incorrect dimensions of A, must be square");
                T.rowval[e] = (int)r;
                T.val[e] = read_val(nzval[t], ty, e);
                mx = std::max(mx, std::abs(T.val[e]));
            }
        }
        T.maxabs = mx;
    }
    return 0;
}

int Engine::set_q_buffer(void* dptr, size_t bytes) {
    q_ext_ = (cz*)dptr;
    q_ext_bytes_ = dptr ? bytes : 0;
    have_x_ = false;
    return 0;
}

int Engine::set_contour(int K_, const double* nodes, const double* weights, int kind, const double* region) {
    if (K_ <= 0 || !nodes || !weights) return fail(FEASTHIP_EARG, "set_contour: bad arguments");
    K = K_;
    nodes_.resize(K); weights_.resize(K);
    for (int k = 0; k < K; ++k) { nodes_[k] = zc(nodes[2 * k], nodes[2 * k + 1]); weights_[k] = zc(weights[2 * k], weights[2 * k + 1]); }
    region_kind_ = kind;
    if (region) std::memcpy(region_, region, sizeof(region_));
    factored_ = false;
    return set_shard(shard_first_, shard_stride_);
}

int Engine::set_shard(int first, int stride) {
    if (first < 0 || stride < 1) return fail(FEASTHIP_EARG, "set_node_shard: bad shard");
    shard_first_ = first; shard_stride_ = stride;
    local_.clear();
    for (int k = first; k < K; k += stride) local_.push_back(k);
    if ((int)local_.size() != Kloc) { factored_ = false; analyzed_ = false; have_x_ = false; }
    factored_ = false;
    Kloc = (int)local_.size();
    return 0;
}

// ---------------------------------------------------------------------------
bool Engine::build_plans() {
    std::vector<Tile> tiles;
    std::vector<int> lists;
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    int nd = S.maxdepth + 1;
    fplan_.assign(nd, FactorLevel());
    splan_.assign(nd, SolveLevel());
    sk_off_.assign(S.nsuper, -1);
    sk_rows_ = 0;
    std::vector<int> nchild(S.nsuper);
    for (int s = 0; s < S.nsuper; ++s) nchild[s] = (int)(S.child_ptr[s + 1] - S.child_ptr[s]);
    // product tiles of one front in 8 x 8 super-tiles (64 consecutive tiles share 8 row panels and 8 column panels; k_gemm_nn
    // hands such runs to one XCD)
    auto push_supertiles = [&](int s, int kind, int na, int nc) {
        for (int A = 0; A < na; A += 8)
            for (int C = 0; C < nc; C += 8)
                for (int a = A; a < std::min(na, A + 8); ++a)
                    for (int c = C; c < std::min(nc, C + 8); ++c) tiles.push_back({s, kind, a, c});
    };
    for (int d = 0; d < nd; ++d) {
        const std::vector<int>& lv = S.levels[d];
        FactorLevel& F = fplan_[d];
        SolveLevel& V = splan_[d];
        int pmax = 0;
        for (int s : lv) pmax = std::max(pmax, S.p(s));
        // extend-add / solve assembly
        F.extend.off = (int64_t)tiles.size();
        // phase 0 touches the L / U12 panels (tiles with a row or a column inside the pivot block), phase 1 the contribution block
        for (int s : lv) if (nchild[s] > 0) { int nt = cdiv(S.p(s) + S.u(s), 64), p = S.p(s); for (int a = 0; a < nt; ++a) for (int b = 0; b < nt; ++b) if (a * 64 < p || b * 64 < p) tiles.push_back({s, a, b, 0}); }
        F.extend.count = (int)(tiles.size() - F.extend.off);
        F.extend1.off = (int64_t)tiles.size();
        for (int s : lv) if (nchild[s] > 0 && S.u(s) > 0) { int nt = cdiv(S.p(s) + S.u(s), 64), p = S.p(s); for (int a = 0; a < nt; ++a) for (int b = 0; b < nt; ++b) if (a * 64 + 64 > p && b * 64 + 64 > p) tiles.push_back({s, a, b, 0}); }
        F.extend1.count = (int)(tiles.size() - F.extend1.off);
        V.assemble.off = (int64_t)tiles.size();
        for (int s : lv) if (nchild[s] > 0) { int nt = cdiv(S.p(s), 64); for (int a = 0; a < nt; ++a) tiles.push_back({s, a, 0, 0}); }      // pivot-block rows only
        V.assemble.count = (int)(tiles.size() - V.assemble.off);
        for (int s : lv)
            for (int64_t c = S.child_ptr[s]; c < S.child_ptr[s + 1]; ++c) {
                int gch = S.child_idx[c];
                for (int i = 0; i < S.u(gch); ++i) { if (S.rel[S.struct_ptr[gch] + i] < S.p(s)) V.asm_rows += 1; else V.pull_rows += 1; }
            }
        // factor steps
        for (int k0 = 0; k0 < pmax; k0 += FACT_NB) {
            FactorStep st;
            st.k0 = k0;
            st.panel.off = (int64_t)lists.size();
            for (int s : lv) if (S.p(s) > k0) lists.push_back(s);
            st.panel.count = (int)(lists.size() - st.panel.off);
            st.swap.off = (int64_t)tiles.size();
            for (int s : lv) if (S.p(s) > k0) { int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0); int nt = cdiv(p - b + u, 256); for (int a = 0; a < nt; ++a) tiles.push_back({s, a, 0, 0}); }
            st.swap.count = (int)(tiles.size() - st.swap.off);
            st.trsm.off = (int64_t)tiles.size();
            for (int s : lv) if (S.p(s) > k0) { int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0); int nc = (p - k0 - b) + u; int nt = cdiv(nc, TRSM_COLS); for (int a = 0; a < nt; ++a) tiles.push_back({s, a, 0, 0}); }
            st.trsm.count = (int)(tiles.size() - st.trsm.off);
            st.gemm.off = (int64_t)tiles.size();
            for (int s : lv) if (S.p(s) > k0) {
                int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0), ld = p + u;
                int M0 = ld - k0 - b, N0 = p - k0 - b;
                if (M0 > 0 && N0 > 0) push_supertiles(s, 0, cdiv(M0, 64), cdiv(N0, 64));
                int M1 = p - k0 - b, N1 = u;
                if (M1 > 0 && N1 > 0) push_supertiles(s, 1, cdiv(M1, 64), cdiv(N1, 64));
            }
            st.gemm.count = (int)(tiles.size() - st.gemm.off);
            st.gemm0.off = (int64_t)tiles.size();        // first row of product tiles only: the 32-row products of the grouped U-row solves
            for (int s : lv) if (S.p(s) > k0) {
                int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0);
                int N0 = p - k0 - b;
                for (int c = 0; c < cdiv(N0, 64); ++c) tiles.push_back({s, 0, 0, c});
                if (N0 > 0) for (int c = 0; c < cdiv(u, 64); ++c) tiles.push_back({s, 1, 0, c});
            }
            st.gemm0.count = (int)(tiles.size() - st.gemm0.off);
            F.steps.push_back(st);
        }
        F.cbgemm.off = (int64_t)tiles.size();
        for (int s : lv) if (S.u(s) > 0) { int nt = cdiv(S.u(s), 64); push_supertiles(s, 2, nt, nt); }
        F.cbgemm.count = (int)(tiles.size() - F.cbgemm.off);
        F.fronts.off = (int64_t)lists.size();
        for (int s : lv) lists.push_back(s);
        F.fronts.count = (int)lv.size();
        // solve-ready finalisation (block inverses, coupling into super blocks, panel scaling)
        F.binv.off = (int64_t)tiles.size();
        for (int s : lv) for (int t = 0; t < cdiv(S.p(s), INV_NB); ++t) tiles.push_back({s, t, 0, 0});
        F.binv.count = (int)(tiles.size() - F.binv.off);
        F.couple.off = (int64_t)tiles.size();
        for (int s : lv) for (int t = 0; t < cdiv(S.p(s), SOLVE_NB); ++t) if (S.p(s) - t * SOLVE_NB > INV_NB) tiles.push_back({s, t, 0, 0});
        F.couple.count = (int)(tiles.size() - F.couple.off);
        F.scale.off = (int64_t)tiles.size();
        for (int s : lv) for (int t = 0; t < cdiv(S.p(s), SOLVE_NB); ++t) {
            int p = S.p(s), ld = p + S.u(s), k0 = t * SOLVE_NB, b = std::min(SOLVE_NB, p - k0);
            for (int a = 0; a < cdiv(k0, 64 * SCALE_ROWTILES); ++a) tiles.push_back({s, t, a, 0});
            for (int a = 0; a < cdiv(ld - k0 - b, 64 * SCALE_ROWTILES); ++a) tiles.push_back({s, t, a, 1});
        }
        F.scale.count = (int)(tiles.size() - F.scale.off);
        // block substitution launches (kernels.hpp: launch_apply).  Tile height (64/32/16 rows) is chosen per
        // launch so that even the top of the tree (few fronts, long K) spreads over the whole chip.
        auto pick_slabs = [&](auto&& rows_of) {     // rows_of(callback(rows)) enumerates the row extents of the launch
            int64_t wgs = 0;
            rows_of([&](int rows) { wgs += cdiv(rows, 64); });
            if (wgs * Kloc < 128) return 1;      // latency-bound launch (few nodes left on this GPU): 16-row tiles, four waves share the columns
            return (wgs * Kloc >= 1024) ? 4 : 2;
        };
        double rowb = 1.0;   // bytes are scaled by 16*ldw at run time: count block-vector ROWS here
        (void)rowb;
        int nsteps = cdiv(pmax, SOLVE_NB);
        for (int c = 0; c + 1 < nsteps; ++c) {
            Launch L;
            L.off = (int64_t)tiles.size();
            int sl = pick_slabs([&](auto&& cb) { for (int s : lv) if (S.p(s) > c * SOLVE_NB + SOLVE_NB) cb(S.p(s) - c * SOLVE_NB - SOLVE_NB); });
            for (int s : lv) {
                int p = S.p(s), k0 = c * SOLVE_NB;
                if (p <= k0 + SOLVE_NB) continue;
                int rows = p - k0 - SOLVE_NB;
                for (int a = 0; a < cdiv(rows, 16 * sl); ++a) tiles.push_back({s, 0 | (sl << 8), a, c});
                L.bytes += (double)rows * SOLVE_NB * 16.0; L.flops += (double)rows * SOLVE_NB;
            }
            L.count = (int)(tiles.size() - L.off);
            V.f1.push_back(L);
        }
        V.f2.off = (int64_t)tiles.size();
        int sl2 = pick_slabs([&](auto&& cb) { for (int s : lv) if (S.u(s) > 0) cb(S.u(s)); });
        for (int s : lv) {
            int p = S.p(s), u = S.u(s);
            for (int a = 0; a < cdiv(u, 16 * sl2); ++a) tiles.push_back({s, 1 | (sl2 << 8), a, 0});
            for (int t = 0; t < cdiv(p, SOLVE_NB); ++t)
                for (int rt = 0; rt < cdiv(std::min(SOLVE_NB, p - t * SOLVE_NB), 64); ++rt) tiles.push_back({s, 2, t, rt});
            V.f2.bytes += ((double)u * p + 0.5 * p * std::min(p, SOLVE_NB)) * 16.0; V.f2.flops += (double)u * p + 0.5 * p * std::min(p, SOLVE_NB);
        }
        V.f2.count = (int)(tiles.size() - V.f2.off);
        // longest tiles first (LPT): the K loop of a tile is p (kind 1) or at most SOLVE_NB (kind 2); fronts stay together
        std::stable_sort(tiles.begin() + V.f2.off, tiles.end(), [&](const Tile& a, const Tile& b) {
            int ka = (a.a & 0xff) == 1 ? S.p(a.f) : std::min(SOLVE_NB, S.p(a.f)), kb2 = (b.a & 0xff) == 1 ? S.p(b.f) : std::min(SOLVE_NB, S.p(b.f));
            return ka > kb2;
        });
        V.b1.off = (int64_t)tiles.size();
        int sl3 = pick_slabs([&](auto&& cb) { for (int s : lv) if (S.u(s) > 0) cb(S.p(s)); });
        {   // few, long tiles (top of the tree): split the K = u loop over several workgroups
            int64_t wgs = 0; int umax = 0;
            for (int s : lv) if (S.u(s) > 0) { wgs += cdiv(S.p(s), 64); umax = std::max(umax, S.u(s)); }
            int ns = 0;
            int keff = std::max(1, Kloc / stream_groups());     // nodes per launch
            if (wgs > 0 && wgs * keff < 1024 && umax >= 512) ns = (int)std::min<int64_t>(8, std::min<int64_t>(umax / 256, cdiv(2048, (int)(wgs * keff))));
            V.b1split = ns >= 2 ? ns : 0;
            if (V.b1split) sl3 = 4;
        }
        int64_t sk_level = 0;
        for (int s : lv) if (S.u(s) > 0) {
            int p = S.p(s), u = S.u(s);
            if (V.b1split) {
                int Kp = ((cdiv(u, V.b1split)) + 31) & ~31, nparts = cdiv(u, Kp);
                sk_off_[s] = sk_level; sk_level += (int64_t)nparts * p;
                for (int q = 0; q < nparts; ++q) for (int a = 0; a < cdiv(p, 64); ++a) tiles.push_back({s, 3 | (sl3 << 8) | (V.b1split << 16), a, q});
            } else {
                for (int a = 0; a < cdiv(p, 16 * sl3); ++a) tiles.push_back({s, 3 | (sl3 << 8), a, 0});
            }
            V.b1.bytes += (double)u * p * 16.0; V.b1.flops += (double)u * p;
        }
        V.b1.count = (int)(tiles.size() - V.b1.off);
        std::stable_sort(tiles.begin() + V.b1.off, tiles.end(), [&](const Tile& a, const Tile& b) { return S.u(a.f) > S.u(b.f); });   // K = u
        sk_rows_ = std::max(sk_rows_, sk_level);
        V.b1red.off = (int64_t)tiles.size();
        if (V.b1split) for (int s : lv) if (S.u(s) > 0) for (int a = 0; a < cdiv(S.p(s), 64); ++a) tiles.push_back({s, a, 0, 0});
        V.b1red.count = (int)(tiles.size() - V.b1red.off);
        V.b2.assign(nsteps, Launch());
        for (int c = 1; c < nsteps; ++c) {
            Launch& L = V.b2[c];
            L.off = (int64_t)tiles.size();
            int sl = pick_slabs([&](auto&& cb) { for (int s : lv) if (S.p(s) > c * SOLVE_NB) cb(c * SOLVE_NB); });
            for (int s : lv) {
                int p = S.p(s), k0 = c * SOLVE_NB;
                if (p <= k0) continue;
                int b = std::min(SOLVE_NB, p - k0);
                for (int a = 0; a < cdiv(k0, 16 * sl); ++a) tiles.push_back({s, 4 | (sl << 8), a, c});
                L.bytes += (double)k0 * b * 16.0; L.flops += (double)k0 * b;
            }
            L.count = (int)(tiles.size() - L.off);
        }
        V.b3.off = (int64_t)tiles.size();
        for (int s : lv) {
            int p = S.p(s);
            for (int t = 0; t < cdiv(p, SOLVE_NB); ++t)
                for (int rt = 0; rt < cdiv(std::min(SOLVE_NB, p - t * SOLVE_NB), 64); ++rt) tiles.push_back({s, 5, t, rt});
            V.b3.bytes += 0.5 * p * std::min(p, SOLVE_NB) * 16.0; V.b3.flops += 0.5 * p * std::min(p, SOLVE_NB);
        }
        V.b3.count = (int)(tiles.size() - V.b3.off);
    }
    if (tiles.empty()) tiles.push_back({0, 0, 0, 0});
    if (lists.empty()) lists.push_back(0);
    return d_tiles_.upload(tiles.data(), tiles.size(), stream_) && d_lists_.upload(lists.data(), lists.size(), stream_);
}

static std::mutex g_sym_mu;
static std::vector<std::pair<uint64_t, Symbolic>> g_sym_cache;      // most recent symbolic analyses (see Engine::analyze)

int Engine::analyze() {
    if (n <= 0 || terms_.empty()) return fail(FEASTHIP_ESTATE, "analyze: no matrices set");
    if (Kloc <= 0) return fail(FEASTHIP_ESTATE, "analyze: no contour nodes on this shard");
    HIPCHK(hipSetDevice(device_));
    auto t0 = std::chrono::steady_clock::now();
    if (dense) {
        analyze_dense(n, S);
    } else {
        std::vector<int64_t> cp(n + 1, 0), ri;
        for (int j = 0; j < n; ++j) {
            int64_t c = 1;
            for (auto& T : terms_) c += T.colptr[j + 1] - T.colptr[j];
            cp[j + 1] = cp[j] + c;
        }
        ri.resize(cp[n]);
        for (int j = 0; j < n; ++j) {
            int64_t o = cp[j];
            ri[o++] = j;
            for (auto& T : terms_) for (int64_t e = T.colptr[j]; e < T.colptr[j + 1]; ++e) ri[o++] = T.rowval[e];
        }
        SymbolicOptions opt;
        if (const char* v = getenv("FEASTHIP_RELAX_SMALL")) opt.relax_small = atoi(v);      // tuning knobs
        if (const char* v = getenv("FEASTHIP_RELAX_MID")) opt.relax_mid = atoi(v);
        if (const char* v = getenv("FEASTHIP_ZMID")) opt.zmid = atof(v);
        if (const char* v = getenv("FEASTHIP_ZBIG")) opt.zbig = atof(v);
        if (const char* v = getenv("FEASTHIP_LEAF")) opt.leaf_size = atoi(v);
        const bool trace = getenv("FEASTHIP_ANALYZE_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[feasthip analyze] pattern union                %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        // Ordering + symbolic analysis depend only on the pattern: FEAST is typically run several times on the same
        // matrices (other contours / spectrum slices, README.md:12 of the reference), so the result is kept in a small
        // process-wide cache keyed by a hash of the pattern and the options.
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
        mix((uint64_t)n); mix((uint64_t)cp[n]);
        for (int j = 0; j <= n; ++j) mix((uint64_t)cp[j]);
        for (int64_t t = 0; t < cp[n]; ++t) mix((uint64_t)ri[t]);
        mix((uint64_t)opt.leaf_size); mix((uint64_t)opt.relax_small); mix((uint64_t)opt.relax_mid); mix((uint64_t)(opt.zmid * 1e6)); mix((uint64_t)(opt.zbig * 1e6)); mix((uint64_t)opt.ordering);
        bool hit = false;
        {
            std::lock_guard<std::mutex> lk(g_sym_mu);
            for (auto& c : g_sym_cache) if (c.first == key && c.second.n == n) { S = c.second; hit = true; break; }
        }
        if (!hit) {
            std::string e = analyze_pattern(n, cp.data(), ri.data(), opt, S);
            if (!e.empty()) return fail(FEASTHIP_EARG, "analyze: " + e);
            std::lock_guard<std::mutex> lk(g_sym_mu);
            if (g_sym_cache.size() >= 4) g_sym_cache.erase(g_sym_cache.begin());
            g_sym_cache.emplace_back(key, S);
        }
        if (trace) fprintf(stderr, "[feasthip analyze] analyze_pattern total        %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        // arena destinations + CSR copies for SpMM: host work per term on its own thread (the destination map itself
        // fans out over the columns), uploads afterwards from this thread
        struct TermHost { std::vector<int64_t> dest, rp; std::vector<int> col; std::vector<zc> v; char bad = 0; };
        std::vector<TermHost> th(terms_.size());
        {
            std::vector<std::thread> tpool;
            for (size_t ti = 0; ti < terms_.size(); ++ti)
                tpool.emplace_back([&, ti]() {
                    SparseTerm& T = terms_[ti];
                    TermHost& H = th[ti];
                    int64_t nnz = T.colptr[n];
                    H.dest.resize(nnz);
                    // arena destination of every entry (a binary search in the supernode's row structure): independent per column
                    int nth = (int)std::min<int64_t>(std::max(1, host_cpu_budget() / 2), std::max<int64_t>(1, nnz / 100000));
                    std::vector<std::thread> pool;
                    std::vector<char> bad(nth, 0);
                    for (int tix = 0; tix < nth; ++tix)
                        pool.emplace_back([&, tix]() {
                            int j0 = (int)((int64_t)n * tix / nth), j1 = (int)((int64_t)n * (tix + 1) / nth);
                            for (int j = j0; j < j1; ++j)
                                for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) {
                                    H.dest[e2] = entry_dest(S, T.rowval[e2], j);
                                    if (H.dest[e2] < 0) bad[tix] = 1;
                                }
                        });
                    H.rp.assign(n + 1, 0);
                    for (int64_t e2 = 0; e2 < nnz; ++e2) H.rp[T.rowval[e2] + 1]++;
                    for (int i = 0; i < n; ++i) H.rp[i + 1] += H.rp[i];
                    H.col.resize(nnz);
                    H.v.resize(nnz);
                    std::vector<int64_t> pos(H.rp.begin(), H.rp.end() - 1);
                    for (int j = 0; j < n; ++j)
                        for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) { int64_t q = pos[T.rowval[e2]]++; H.col[q] = j; H.v[q] = T.val[e2]; }
                    for (auto& t2 : pool) t2.join();
                    for (char b2 : bad) if (b2) H.bad = 1;
                });
            for (auto& t2 : tpool) t2.join();
        }
        for (size_t ti = 0; ti < terms_.size(); ++ti) {
            SparseTerm& T = terms_[ti];
            TermHost& H = th[ti];
            if (H.bad) return fail(FEASTHIP_EARG, "analyze: internal, entry outside the symbolic structure");
            int64_t nnz = T.colptr[n];
            bool ok = T.d_dest.upload(H.dest.data(), nnz, stream_) && T.d_cscval.upload((const cz*)T.val.data(), nnz, stream_) &&
                      T.d_rowptr.upload(H.rp.data(), n + 1, stream_) && T.d_col.upload(H.col.data(), nnz, stream_) &&
                      T.d_val.upload((const cz*)H.v.data(), nnz, stream_);
            if (!ok) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
        }
    }
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze] + arena destinations, CSR    %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    // device symbolic
    std::vector<FrontD> fr(S.nsuper);
    for (int s = 0; s < S.nsuper; ++s) {
        FrontD& f = fr[s];
        f.c0 = S.sn_start[s]; f.p = S.p(s); f.u = S.u(s); f.parent = S.sn_parent[s];
        f.lp = S.lp_off[s]; f.f12 = S.f12_off[s]; f.cb = S.cb_off[s];
        f.sptr = S.struct_ptr[s]; f.crow = S.crow_off[s];
        f.child_ptr = S.child_ptr[s]; f.nchild = (int)(S.child_ptr[s + 1] - S.child_ptr[s]); f.depth = S.depth[s];
        f.sk_off = -1;
        f.cmap = -1;
    }
    // pull map of the forward sweep: for every contribution row of a front and every child slot, the row of the child's
    // contribution block that is added to it (rel >= p), as an absolute row of the children's arena; -1 if none
    std::vector<int> cmap;
    for (int s = 0; s < S.nsuper; ++s) {
        int nch = fr[s].nchild, u = fr[s].u, p = fr[s].p;
        if (nch == 0 || u == 0) continue;
        fr[s].cmap = (int64_t)cmap.size();
        size_t base = cmap.size();
        cmap.resize(base + (size_t)u * nch, -1);
        for (int c = 0; c < nch; ++c) {
            int gch = S.child_idx[S.child_ptr[s] + c];
            int ug = S.u(gch);
            for (int i = 0; i < ug; ++i) {
                int dst = S.rel[S.struct_ptr[gch] + i];
                if (dst >= p) cmap[base + (size_t)(dst - p) * nch + c] = (int)(S.crow_off[gch] + i);
            }
        }
    }
    if (cmap.empty()) cmap.push_back(-1);
    std::vector<int> c0row(n);
    for (int i = 0; i < n; ++i) c0row[i] = S.sn_start[S.sn_of[i]];
    std::vector<int> one(1, 0);
    bool ok = d_fr_.upload(fr.data(), fr.size(), stream_) &&
              d_struct_.upload(S.struct_idx.empty() ? one.data() : S.struct_idx.data(), std::max<size_t>(1, S.struct_idx.size()), stream_) &&
              d_rel_.upload(S.rel.empty() ? one.data() : S.rel.data(), std::max<size_t>(1, S.rel.size()), stream_) &&
              d_child_.upload(S.child_idx.empty() ? one.data() : S.child_idx.data(), std::max<size_t>(1, S.child_idx.size()), stream_) &&
              d_perm_.upload(S.perm.data(), n, stream_) && d_c0row_.upload(c0row.data(), n, stream_) &&
              d_cmap_.upload(cmap.data(), cmap.size(), stream_);
    if (!ok || !build_plans()) return fail(FEASTHIP_EHIP, "analyze: device upload of the symbolic structure failed");
    for (int s = 0; s < S.nsuper; ++s) fr[s].sk_off = sk_off_[s];
    if (!d_fr_.upload(fr.data(), fr.size(), stream_)) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
    dsym_.fr = d_fr_.p; dsym_.struct_idx = d_struct_.p; dsym_.rel = d_rel_.p; dsym_.child_idx = d_child_.p;
    dsym_.perm = d_perm_.p; dsym_.c0_of_row = d_c0row_.p; dsym_.cmap = d_cmap_.p; dsym_.n = n;
    // numeric arenas
    analyze_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto t_alloc = std::chrono::steady_clock::now();
    ok = d_fact_.alloc((size_t)Kloc * S.fact_size) && d_cb_[0].alloc((size_t)Kloc * std::max<int64_t>(1, S.cb_size[0])) &&
         d_cb_[1].alloc((size_t)Kloc * std::max<int64_t>(1, S.cb_size[1])) && d_piv_.alloc((size_t)Kloc * n) &&
         d_rowperm_.alloc((size_t)Kloc * n) && d_rowinv_.alloc((size_t)Kloc * n) && d_nperturb_.alloc(Kloc) &&
         d_tiny_.alloc(Kloc) && d_coef_.alloc((size_t)Kloc * nterms);
    if (!ok) return fail(FEASTHIP_EHIP, "analyze: out of device memory for the factor arenas");
    alloc_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_alloc).count();
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze] + plans, uploads             %.1f ms (cumulative)\n[feasthip analyze] arena allocation (hipMalloc)  %.1f ms\n", analyze_seconds * 1e3, alloc_seconds * 1e3);
    analyzed_ = true;
    factored_ = false;
    return 0;
}

// f_t(z) of T(z) = sum_t f_t(z) scale_t M_t at the given points: powers z^t by default (polynomial NEP, linear pencil),
// or the caller's functions (split form, feasthip_set_coef_functions).  out is point-major: out[i*nterms + t].
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
    return 0;
}

int Engine::set_node_coefficients(const double* coef, int K_, int nterms_) {
    if (!coef) { node_coef_.clear(); factored_ = false; return 0; }
    if (K_ != K || nterms_ != nterms) return fail(FEASTHIP_EARG, "set_node_coefficients: table must be K x nterms (set the contour and the terms first)");
    node_coef_.assign(reinterpret_cast<const zc*>(coef), reinterpret_cast<const zc*>(coef) + (size_t)K * nterms);
    factored_ = false;
    return 0;
}

int Engine::factorize() {
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
    HIPCHK(hipStreamSynchronize(stream_));   // coef/tiny are stack-owned host buffers
    HIPCHK(hipEventRecord(ev0_, stream_));
    int64_t fs = S.fact_size;
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
    const Tile* tl = d_tiles_.p;
    const int* ls = d_lists_.p;
    bool fin_used = false;
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
                          std::max<int64_t>(1, S.cb_size[par ^ 1]), Kloc, 0);
        int level_ld = 0;
        for (int s2 : S.levels[d]) level_ld = std::max(level_ld, S.p(s2) + S.u(s2));
        const bool tall = level_ld >= 1024;      // 1024-thread panel workgroups for tall fronts (430 -> 396 ms for the 16 factorisations of C3)
        int level_p = 0;
        for (int s2 : S.levels[d]) level_p = std::max(level_p, S.p(s2));
        const bool lookahead = look_ahead_ && level_p >= 256 && la_stream_ && F.steps.size() >= 4;
        stamp();
        if (!lookahead) {
            for (FactorStep& st : F.steps) {
                launch_panel(stream_, dsym_, ls + st.panel.off, st.panel.count, st.k0, d_fact_.p, fs, d_piv_.p, d_tiny_.p, d_nperturb_.p, Kloc, tall, level_ld - st.k0);
                launch_swap(stream_, dsym_, tl + st.swap.off, st.swap.count, st.k0, d_fact_.p, fs, d_piv_.p, Kloc);
                launch_trsm_u(stream_, dsym_, tl + st.trsm.off, st.trsm.count, st.k0, d_fact_.p, fs, Kloc);
                launch_gemm_nn(stream_, dsym_, tl + st.gemm.off, st.gemm.count, st.k0, d_fact_.p, fs, cbp, cbs, Kloc);
            }
        } else {
            // LU with look-ahead for the big fronts at the top of the tree (few fronts x nodes: the panel factorisation runs on a
            // handful of workgroups and would otherwise leave the GPU idle between two trailing updates).  The panels are taken
            // in groups of group_panels_ (columns [K0_g, K0_g+1)).  The high-priority stream P factors the panels of group g,
            // updating only the group's own columns in between, and then applies the whole group, as one product of depth
            // K0_g+1 - K0_g, to the columns of group g+1 (W below).  The main stream applies group g to everything else (the
            // left columns' row interchanges, the pivot-block columns from K0_g+2 on and U12) while P is busy with group g+1.
            // Deferring a group's update is legal because all its row interchanges are applied first, to L and to C alike; the
            // U rows then come from a blocked forward substitution (a 32-row product with the panels before it, then the
            // triangular solve with the panel's own L11) and the rest is one product.  Against per-panel updates the product
            // reads and writes C once per group instead of once per panel, which is what bounds it.
            hipStream_t P = la_stream_, R = rest_stream_ ? rest_stream_ : stream_;
            HIPCHK(hipEventRecord(ev_la_[4], stream_));
            HIPCHK(hipStreamWaitEvent(P, ev_la_[4], 0));
            if (R != stream_) HIPCHK(hipStreamWaitEvent(R, ev_la_[4], 0));
            const int ns = (int)F.steps.size(), G = group_panels_, ng = (ns + G - 1) / G;
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
                launch_swap(s, dsym_, tl + s0.swap.off, s0.swap.count, k0g, d_fact_.p, fs, d_piv_.p, Kloc, sel(k0g + gw), gw);
                for (int j = j0; j < j1; ++j) {
                    FactorStep& sj = F.steps[j];
                    const int bw = sj.k0 - k0g;
                    if (bw > 0) {
                        ColSel c = sel(sj.k0); c.llo = c.lhi = 0;
                        launch_gemm_nn(s, dsym_, tl + s0.gemm0.off, s0.gemm0.count, k0g, d_fact_.p, fs, cbp, cbs, Kloc, c, bw, FACT_NB);
                    }
                    ColSel c = sel(sj.k0 + FACT_NB);
                    launch_trsm_u(s, dsym_, tl + sj.trsm.off, sj.trsm.count, sj.k0, d_fact_.p, fs, Kloc, c);
                }
                launch_gemm_nn(s, dsym_, tl + s0.gemm.off, s0.gemm.count, k0g, d_fact_.p, fs, cbp, cbs, Kloc, sel(k0g + gw), gw, 0);
            };
            for (int g = 0; g < ng; ++g) {
                const int j0 = g * G, j1 = std::min(ns, j0 + G), k0g = F.steps[j0].k0;
                const int kend = K0(g + 1);                     // first column after the group (INF for the last group)
                if (g > 0) {
                    if (g > 1) HIPCHK(hipStreamWaitEvent(P, ev_la_[2 + (g & 1)], 0));      // rest of group g-2 reaches these columns
                    apply_group(P, g - 1, k0g, kend, false);
                }
                for (int j = j0; j < j1; ++j) {
                    FactorStep& sj = F.steps[j];
                    launch_panel(P, dsym_, ls + sj.panel.off, sj.panel.count, sj.k0, d_fact_.p, fs, d_piv_.p, d_tiny_.p, d_nperturb_.p, Kloc, tall,
                                 level_ld - sj.k0);
                    // inside the group: interchanges on the group's earlier L columns and its remaining columns, U rows and update
                    // of the remaining columns
                    ColSel c;
                    c.llo = k0g; c.lhi = sj.k0; c.clo = 0; c.chi = kend >= INF ? INF : std::max(0, kend - sj.k0 - FACT_NB); c.u12 = 0;
                    launch_swap(P, dsym_, tl + sj.swap.off, sj.swap.count, sj.k0, d_fact_.p, fs, d_piv_.p, Kloc, c);
                    if (c.chi > 0) {
                        c.llo = c.lhi = 0;
                        launch_trsm_u(P, dsym_, tl + sj.trsm.off, sj.trsm.count, sj.k0, d_fact_.p, fs, Kloc, c);
                        launch_gemm_nn(P, dsym_, tl + sj.gemm.off, sj.gemm.count, sj.k0, d_fact_.p, fs, cbp, cbs, Kloc, c);
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
        launch_gemm_nn(stream_, dsym_, tl + F.cbgemm.off, F.cbgemm.count, 0, d_fact_.p, fs, cbp, cbs, Kloc);
        stamp();
        // The solve-ready conversion of this level (row permutations, block inverses, panel scaling) only touches this
        // level's own panels, which nothing above reads (parents take the contribution blocks): it runs on a side stream
        // next to the factorisation of the levels above.  It must follow the Schur-complement product, which reads L21 / U12.
        hipStream_t fin = (finalize_async_ && !aux_streams_.empty()) ? aux_streams_[0] : stream_;
        if (fin != stream_) { HIPCHK(hipEventRecord(ev_fork_, stream_)); HIPCHK(hipStreamWaitEvent(fin, ev_fork_, 0)); fin_used = true; }
        launch_rowperm(fin, dsym_, ls + F.fronts.off, F.fronts.count, d_piv_.p, d_rowperm_.p, d_rowinv_.p, Kloc);
        launch_block_inverse(fin, dsym_, tl + F.binv.off, F.binv.count, d_fact_.p, fs, Kloc);
        launch_couple_inverse(fin, dsym_, tl + F.couple.off, F.couple.count, d_fact_.p, fs, Kloc);
        launch_scale_panels(fin, dsym_, tl + F.scale.off, F.scale.count, d_fact_.p, fs, Kloc);
        launch_extend_add(stream_, dsym_, tl + F.extend1.off, F.extend1.count, d_fact_.p, fs, cbp, cbs, d_cb_[par ^ 1].p,
                          std::max<int64_t>(1, S.cb_size[par ^ 1]), Kloc, 1);
    }
    if (fin_used) { HIPCHK(hipEventRecord(ev_join_[0], aux_streams_[0])); HIPCHK(hipStreamWaitEvent(stream_, ev_join_[0], 0)); }
    if (ftrace) {
        stamp();
        HIPCHK(hipStreamSynchronize(stream_));
        auto ms = [&](size_t a, size_t b) { float t = 0; (void)hipEventElapsedTime(&t, tev[a], tev[b]); return t; };
        size_t q = 0;
        for (int d = S.maxdepth; d >= 0; --d, q += 3) {
            int mp = 0, mu = 0;
            double fl_schur = 0, fl_steps = 0;
            for (int s2 : S.levels[d]) {
                double p2 = S.p(s2), u2 = S.u(s2);
                mp = std::max(mp, S.p(s2)); mu = std::max(mu, S.u(s2));
                fl_schur += 8.0 * u2 * u2 * p2 * Kloc;
                fl_steps += 8.0 * (p2 * p2 * p2 / 3.0 + p2 * p2 * u2) * Kloc;      // LU of the pivot block, L21 and U12 (8 flops per complex multiply-add)
            }
            fprintf(stderr, "[feasthip factor] depth %2d  fronts %5zu  max p %5d  max u %5d  steps %3zu | extend-add %7.2f  panel steps %7.2f (%5.1f TF)  schur product %7.2f ms (%5.1f TF)\n",
                    d, S.levels[d].size(), mp, mu, fplan_[d].steps.size(), ms(q, q + 1), ms(q + 1, q + 2), fl_steps / ms(q + 1, q + 2) * 1e-9, ms(q + 2, q + 3), fl_schur / ms(q + 2, q + 3) * 1e-9);
        }
        for (hipEvent_t e4 : tev) (void)hipEventDestroy(e4);
    }
    if (mixed_) {      // complex64 copy of the solve-ready factors for the mixed-precision sweeps
        if (!d_fact32_.alloc((size_t)Kloc * fs)) return fail(FEASTHIP_EHIP, "out of device memory for the complex64 factors");
        launch_to_f32(stream_, (int64_t)Kloc * fs, d_fact_.p, d_fact32_.p);
    }
    HIPCHK(hipEventRecord(ev1_, stream_));
    std::vector<int> np(Kloc);
    HIPCHK(hipMemcpyAsync(np.data(), d_nperturb_.p, Kloc * sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    factor_ms = ms;
    perturbed = 0;
    for (int v : np) perturbed += v;
    factored_ = true;
    return perturbed > 0 ? (int)std::min<int64_t>(perturbed, 0x7fffffff) : 0;
}

// ---------------------------------------------------------------------------
int Engine::alloc_iteration_buffers() {
    size_t blk = (size_t)n * ldw;
    bool ok = d_X_.alloc(blk) && d_R_.alloc(blk) && (q_ext_ ? (q_ext_bytes_ >= 2 * blk * sizeof(cz)) : d_Q_.alloc(2 * blk)) && d_T1_.alloc(blk) && d_T2_.alloc(blk) &&
              d_MX_.alloc(blk * std::max(2, nterms)) && d_W_.alloc((size_t)Kloc * blk) && d_Y_.alloc((size_t)Kloc * blk) && d_sk_.alloc((size_t)Kloc * std::max<int64_t>(1, sk_rows_) * ldw) &&
              d_crow_[0].alloc((size_t)Kloc * std::max<int64_t>(1, S.crow_size[0]) * ldw) &&
              d_crow_[1].alloc((size_t)Kloc * std::max<int64_t>(1, S.crow_size[1]) * ldw) &&
              d_e0_.alloc((size_t)Kloc * ldw) && d_e1_.alloc((size_t)Kloc * ldw) && d_small_.alloc((size_t)8 * ldw * ldw) &&
              d_lam_.alloc(ldw) && d_gram_scratch_.alloc(gram_scratch_elems(n, ldw)) && d_io_.alloc((size_t)n * std::max(1, m0)) &&
              d_nrm_.alloc(ldw) && d_nrm_scratch_.alloc(colnorm_scratch_elems(n, ldw));
    if (ok && mixed_)
        ok = d_W32_.alloc((size_t)Kloc * blk) && d_Y32_.alloc((size_t)Kloc * blk) && d_sk32_.alloc((size_t)Kloc * std::max<int64_t>(1, sk_rows_) * ldw) &&
             d_crow32_[0].alloc((size_t)Kloc * std::max<int64_t>(1, S.crow_size[0]) * ldw) &&
             d_crow32_[1].alloc((size_t)Kloc * std::max<int64_t>(1, S.crow_size[1]) * ldw);
    if (!ok) return fail(FEASTHIP_EHIP, "out of device memory for the iteration buffers");
    return 0;
}

int Engine::set_precision(int mixed) {
    if ((mixed != 0) != mixed_) { mixed_ = mixed != 0; factored_ = false; have_x_ = false; }
    return 0;
}

int Engine::set_x(const void* X, int64_t n_, int m0_) {
    if (!X || m0_ <= 0) return fail(FEASTHIP_EARG, "set_x: bad arguments");
    if (n_ != n) return fail(FEASTHIP_EDIM_X, "Incorrect dimensions of X, must match A");
    if (m0_ > n) return fail(FEASTHIP_EARG, "set_x: m0 must not exceed n");
    if (!analyzed_) { int r = analyze(); if (r) return r; }
    HIPCHK(hipSetDevice(device_));
    m0 = m0_;
    ldw = ((m0 + 15) / 16) * 16;
    int r = alloc_iteration_buffers();
    if (r) return r;
    HIPCHK(hipMemcpyAsync(d_io_.p, X, (size_t)n * m0 * sizeof(cz), hipMemcpyHostToDevice, stream_));
    launch_cm_to_rm(stream_, n, m0, ldw, d_io_.p, n, d_X_.p);
    HIPCHK(hipMemcpyAsync(Qp(), d_X_.p, (size_t)n * ldw * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    HIPCHK(hipMemsetAsync(d_R_.p, 0, (size_t)n * ldw * sizeof(cz), stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    lambda_.assign(m0, zc(0.0));
    have_x_ = true;
    return 0;
}

int Engine::get_block(int which, void* out) {
    if (!have_x_ || !out) return fail(FEASTHIP_ESTATE, "get_block: no iteration state");
    HIPCHK(hipSetDevice(device_));
    const cz* src = which == 0 ? d_X_.p : which == 1 ? d_R_.p : which == 2 ? Qp() : Qp() + (size_t)n * ldw;
    launch_rm_to_cm(stream_, n, m0, ldw, src, d_io_.p, n);
    HIPCHK(hipMemcpyAsync(out, d_io_.p, (size_t)n * m0 * sizeof(cz), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return 0;
}

int Engine::q_device(void** p, int64_t* n_, int64_t* ldq, int* nblocks) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "q_device: call set_x first");
    if (p) *p = (void*)Qp();
    if (n_) *n_ = n;
    if (ldq) *ldq = ldw;
    if (nblocks) *nblocks = 2;
    return 0;
}

void Engine::apply_term(int t, const cz* V, cz* out, zc alpha, bool accumulate) { apply_term_ld(t, V, out, alpha, accumulate, ldw); }

void Engine::apply_term_ld(int t, const cz* V, cz* out, zc alpha, bool accumulate, int ld) {
    SparseTerm& T = terms_[t];
    zc a = alpha * T.scale;
    if (dense) {
        if (T.identity) launch_axpby(stream_, n, ld, out, to_cz(a), V, accumulate ? 1 : 0);
        else launch_dense_mv(stream_, n, T.d_dense.p, V, ld, out, to_cz(a), accumulate ? 1 : 0);
    } else {
        launch_spmm_csr(stream_, n, T.d_rowptr.p, T.d_col.p, T.d_val.p, V, ld, out, to_cz(a), accumulate ? 1 : 0);
    }
}

int Engine::col_norms(const Tall& t, const cz* V, std::vector<double>& out) {
    launch_colnorm2(stream_, t.rows, t.ld, V, t.nrm_scratch, t.nrm);
    out.resize(t.ld);
    HIPCHK(hipMemcpyAsync(out.data(), t.nrm, t.ld * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    for (double& v : out) v = std::sqrt(v);
    return 0;
}

// Thin orthonormal basis of span(Q) (replaces `Q .= Matrix(qr(Q).Q)`, src/feast.jl:41,117).
// Column-scaled CholeskyQR2: columns are normalised first (the FEAST filter scales column j by
// rho(lambda_j), which is the dominant source of ill-conditioning), then two Cholesky passes on
// the f64-MFMA Gram matrix.  Rtotal (optional) receives R with Q_in = Q_out * R.
int Engine::orthonormalize(const Tall& t, cz* Q, cz* tmp, std::vector<zc>* Rtotal) {
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
    int need = 2;
    for (int pass = 0; pass < 6 && need > 0; ++pass) {
        launch_gram(stream_, rows, m, ld, src, src, t.gram, dG);
        HIPCHK(hipMemcpyAsync(G.data(), dG, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
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
        }
        if (Rtotal) {   // Q_src = Q_dst (R D^-1)  ->  Rt <- (R D^-1) Rt   (only the Beyn step needs the accumulated factor)
            std::vector<zc> Rd(Gc);
            for (int j = 0; j < m; ++j) for (int i = 0; i <= j; ++i) Rd[i + (size_t)j * m] *= dsc[j];
            matmul(m, Rd.data(), Rt.data(), Rn.data());
            Rt.swap(Rn);
        }
        trtri_upper(m, Gc.data());
        for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) Tm[i + (size_t)j * m] = Gc[i + (size_t)j * m] / dsc[i];     // T = D^-1 R^-1
        HIPCHK(hipMemcpyAsync(dM, Tm.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
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
int Engine::orthonormalize_rr(const Tall& t, cz* Q, cz* tmp, std::vector<zc>& Rtot, std::vector<uint8_t>& keep) {
    const int m = t.m, rows = t.rows, ld = t.ld;
    std::vector<double> nr;
    int r = col_norms(t, Q, nr);
    if (r) return r;
    std::vector<zc> sc(ld, zc(0.0));
    Rtot.assign((size_t)m * m, zc(0.0));
    keep.assign(m, 1);
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

int Engine::rr_step(double* lambda, double* res, uint8_t* inside) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "rr_step: call set_x first");
    if (nterms != 2) return fail(FEASTHIP_ESTATE, "rr_step: linear pencil (2 terms) required");
    HIPCHK(hipSetDevice(device_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int m = m0;
    int r = orthonormalize(Qp(), d_T1_.p, nullptr);                     // src/feast.jl:41
    if (r) return r;
    cz* dA = d_small_.p;
    cz* dB = d_small_.p + (size_t)ldw * ldw;
    cz* dM = d_small_.p + (size_t)2 * ldw * ldw;
    std::vector<zc> Aq((size_t)m * m), Bq((size_t)m * m), lam(m), Xq((size_t)m * m);
    apply_term(0, Qp(), d_R_.p, 1.0, false);                            // R = A Q          :42
    launch_gram(stream_, n, m, ldw, Qp(), d_R_.p, d_gram_scratch_.p, dA);   // Aq = Q' R   :43
    // B = -scale_1 * M_1
    bool b_is_identity = terms_[1].identity && std::abs(terms_[1].scale + 1.0) == 0.0;
    if (!b_is_identity) {
        apply_term(1, Qp(), d_T1_.p, -1.0, false);                      // R = B Q          :120
        launch_gram(stream_, n, m, ldw, Qp(), d_T1_.p, d_gram_scratch_.p, dB);   // Bq    :121
        HIPCHK(hipMemcpyAsync(Bq.data(), dB, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    }
    HIPCHK(hipMemcpyAsync(Aq.data(), dA, (size_t)m * m * sizeof(zc), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
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
    HIPCHK(hipMemcpyAsync(dM, Xq.data(), (size_t)m * m * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(d_lam_.p, lamp.data(), ldw * sizeof(zc), hipMemcpyHostToDevice, stream_));
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
        if (!b_is_identity) { apply_term(1, d_X_.p, d_T2_.p, -1.0, false); BX = d_T2_.p; }
        launch_residual_combine(stream_, n, ldw, d_T1_.p, BX, d_lam_.p, d_R_.p);
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
    return 0;
}

// ---------------------------------------------------------------------------
int Engine::solve_all(const cz* rhs) { return solve_all_impl(rhs, nullptr); }

int Engine::solve_range(hipStream_t st, int kb, int kc, const cz* rhs, ProfileRow* prof) {
    if (mixed_) return solve_range_t<cf>(st, kb, kc, rhs, prof, d_fact32_.p, d_W32_.p, d_Y32_.p, d_crow32_[0].p, d_crow32_[1].p, d_sk32_.p);
    return solve_range_t<cz>(st, kb, kc, rhs, prof, d_fact_.p, d_W_.p, d_Y_.p, d_crow_[0].p, d_crow_[1].p, d_sk_.p);
}

template <typename T>
int Engine::solve_range_t(hipStream_t st, int kb, int kc, const cz* rhs, ProfileRow* prof, T* fact, T* Wall, T* Yall, T* crow0, T* crow1, T* skall) {
    int64_t fs = S.fact_size, ws = (int64_t)n * ldw;
    T* crowb[2] = {crow0, crow1};
    T* Fp = fact + (int64_t)kb * fs;
    T* Wp = Wall + (int64_t)kb * ws;
    T* Yp = Yall + (int64_t)kb * ws;
    const int* rperm = d_rowperm_.p + (int64_t)kb * n;
    const int* rinv = d_rowinv_.p + (int64_t)kb * n;
    const Tile* tl = d_tiles_.p;
    hipEvent_t pa = nullptr, pb = nullptr;
    if (prof) { HIPCHK(hipEventCreate(&pa)); HIPCHK(hipEventCreate(&pb)); }
    if (prof) prof_detail.clear();
    int cur_tiles = 0;
    auto begin = [&]() { if (prof) (void)hipEventRecord(pa, st); };
    auto end = [&](int row, double bytes, double flops) {
        if (!prof) return;
        (void)hipEventRecord(pb, st);
        (void)hipEventSynchronize(pb);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, pa, pb);
        prof[row].ms += ms; prof[row].bytes += bytes; prof[row].flops += flops; prof[row].launches++;
        prof_detail.insert(prof_detail.end(), {(double)row, (double)ms, bytes, flops, (double)cur_tiles});
    };
    double rowb = (double)ldw * sizeof(T);
    begin();
    launch_gather_rhs(st, dsym_, rhs, ldw, Wp, ws, rperm, kc);
    end(0, 2.0 * n * rowb * kc, 0);
    // Launch::bytes counts factor bytes, Launch::flops complex multiply-adds per block-vector column (one node)
    auto stat_bytes = [&](const Launch& L, double vec_rows) { return (L.bytes * (sizeof(T) / 16.0) + vec_rows * rowb) * kc; };
    auto stat_flops = [&](const Launch& L) { return L.flops * 8.0 * ldw * kc; };
    for (int d = S.maxdepth; d >= 0; --d) {          // forward sweep, leaves to root
        SolveLevel& V = splan_[d];
        int par = d & 1;
        int64_t crs = std::max<int64_t>(1, S.crow_size[par]) * ldw, crsc = std::max<int64_t>(1, S.crow_size[par ^ 1]) * ldw;
        begin();
        launch_fwd_assemble(st, dsym_, tl + V.assemble.off, V.assemble.count, Wp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs,
                            (crowb[par ^ 1] + (int64_t)kb * crsc), crsc, rinv, kc, 0);
        if (V.assemble.count) end(1, 3.0 * V.asm_rows * rowb * kc, 0);
        double lvl_rows = 0;
        for (int s : S.levels[d]) lvl_rows += S.p(s) + S.u(s);
        for (Launch& L : V.f1) {
            begin();
            launch_apply(st, dsym_, tl + L.off, L.count, Fp, fs, Wp, Yp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs, kc);
            if (L.count) { cur_tiles = L.count; end(3, stat_bytes(L, 2.0 * L.flops / SOLVE_NB), stat_flops(L)); }
        }
        begin();
        // contribution rows of the children that pass through to this level's contribution rows are pulled in the
        // epilogue of the F2a tiles (kind 1), so there is no second assembly pass
        launch_apply(st, dsym_, tl + V.f2.off, V.f2.count, Fp, fs, Wp, Yp, ws, ldw, (crowb[par] + (int64_t)kb * crs), crs, kc, false,
                     nullptr, 0, (crowb[par ^ 1] + (int64_t)kb * crsc), crsc);
        if (V.f2.count) { cur_tiles = V.f2.count; end(3, stat_bytes(V.f2, 2.0 * lvl_rows + V.pull_rows), stat_flops(V.f2)); }
    }
    for (int d = 0; d <= S.maxdepth; ++d) {          // backward sweep, root to leaves
        SolveLevel& V = splan_[d];
        double lvl_p = 0, lvl_u = 0;
        for (int s : S.levels[d]) { lvl_p += S.p(s); lvl_u += S.u(s); }
        begin();
        int64_t sks = std::max<int64_t>(1, sk_rows_) * ldw;
        launch_apply(st, dsym_, tl + V.b1.off, V.b1.count, Fp, fs, Wp, Yp, ws, ldw, nullptr, 0, kc, true, skall + (int64_t)kb * sks, sks);
        if (V.b1split) launch_splitk_reduce(st, dsym_, tl + V.b1red.off, V.b1red.count, V.b1split, Yp, ws, ldw, skall + (int64_t)kb * sks, sks, kc);
        if (V.b1.count) { cur_tiles = -V.b1.count; end(3, stat_bytes(V.b1, lvl_u + 2.0 * lvl_p), stat_flops(V.b1)); }
        for (int c = (int)V.b2.size() - 1; c >= 1; --c) {
            Launch& L = V.b2[c];
            begin();
            launch_apply(st, dsym_, tl + L.off, L.count, Fp, fs, Wp, Yp, ws, ldw, nullptr, 0, kc);
            if (L.count) { cur_tiles = L.count; end(3, stat_bytes(L, 2.0 * L.flops / SOLVE_NB), stat_flops(L)); }
        }
        begin();
        launch_apply(st, dsym_, tl + V.b3.off, V.b3.count, Fp, fs, Wp, Yp, ws, ldw, nullptr, 0, kc);
        if (V.b3.count) { cur_tiles = V.b3.count; end(3, stat_bytes(V.b3, 2.0 * lvl_p), stat_flops(V.b3)); }
    }
    if (prof) { (void)hipEventDestroy(pa); (void)hipEventDestroy(pb); }
    HIPCHK(hipGetLastError());
    return 0;
}

// Node groups that run on their own streams.  Two groups overlap the launch tails and the latency-bound small
// launches of one group with the large launches of the other when there are many nodes; with few local nodes
// (multi-GPU shards) one fuller launch per step is faster (measured: 16 nodes 2 groups, <= 8 nodes 1 group).
// Q0 (Q1) = sum_k T_k .* e0[k,:] (e1[k,:]) over the solution blocks of the last sweep (fp64 accumulation in either precision)
void Engine::accumulate_solutions(int K_, const cz* e0, const cz* e1, int mode, cz* Q0, cz* Q1) {
    if (mixed_) launch_accumulate(stream_, dsym_, d_X_.p, d_W32_.p, (int64_t)n * ldw, ldw, K_, e0, e1, mode, Q0, Q1);
    else launch_accumulate(stream_, dsym_, d_X_.p, d_W_.p, (int64_t)n * ldw, ldw, K_, e0, e1, mode, Q0, Q1);
}

int Engine::stream_groups() const {
    int ns = std::min<int>((int)aux_streams_.size(), Kloc);
    if (!streams_forced_ && Kloc < 12) ns = 1;
    return std::max(1, ns);
}

int Engine::solve_all_impl(const cz* rhs, ProfileRow* prof) {
    if (!factored_) return fail(FEASTHIP_ESTATE, "solve: factorize first");
    int ns = stream_groups();
    if (prof || ns <= 1) return solve_range(stream_, 0, Kloc, rhs, prof);
    // The nodes are independent: split them over a few streams so that the tails and the latency-bound small
    // launches of one group overlap with the large launches of another.
    HIPCHK(hipEventRecord(ev_fork_, stream_));
    int per = (Kloc + ns - 1) / ns;
    for (int g = 0; g < ns; ++g) {
        int kb = g * per, kc = std::min(per, Kloc - kb);
        if (kc <= 0) break;
        HIPCHK(hipStreamWaitEvent(aux_streams_[g], ev_fork_, 0));
        int r = solve_range(aux_streams_[g], kb, kc, rhs, nullptr);
        if (r) return r;
        HIPCHK(hipEventRecord(ev_join_[g], aux_streams_[g]));
        HIPCHK(hipStreamWaitEvent(stream_, ev_join_[g], 0));
    }
    return 0;
}

double Engine::solve_bytes() const {
    // factor entries streamed once per node and sweep pair + block-vector traffic
    double rowb = (double)ldw * 16.0;
    // contribution rows: written once by their front; then either pulled by the parent's F2a epilogue (1 read) or
    // assembled into the parent's pivot block (read + read-modify-write)
    double asm_rows = 0, pull_rows = 0;
    for (const SolveLevel& V : splan_) { asm_rows += V.asm_rows; pull_rows += V.pull_rows; }
    double es = mixed_ ? 0.5 : 1.0;     // complex64 factors and block vectors in the mixed-precision sweeps
    double per_node = (double)S.fact_size * 16.0 * es + (double)S.sum_u * 4.0 + (4.0 * n + (double)S.sum_u + pull_rows + 3.0 * asm_rows) * rowb * es;
    return per_node * Kloc + 3.0 * n * rowb;
}

double Engine::solve_flops() const {
    return ((double)(S.nnz_l + S.nnz_u) * 8.0 * ldw) * Kloc + 16.0 * n * (double)ldw * Kloc;
}

int Engine::contour_step(int refactor) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "contour_step: call set_x and rr_step first");
    HIPCHK(hipSetDevice(device_));
    if (refactor || !factored_) { int r = factorize(); if (r < 0) return r; }
    std::vector<zc> e0((size_t)Kloc * ldw, zc(0.0));
    for (int kk = 0; kk < Kloc; ++kk) {
        zc z = nodes_[local_[kk]], w = weights_[local_[kk]];
        for (int j = 0; j < m0; ++j) e0[(size_t)kk * ldw + j] = w / (z - lambda_[j]);      // src/feast.jl:60,69
    }
    HIPCHK(hipMemcpyAsync(d_e0_.p, e0.data(), e0.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int r = solve_all(d_R_.p);                                                                // :61-66
    if (r) return r;
    accumulate_solutions(Kloc, d_e0_.p, nullptr, 1, Qp(), nullptr);                                // :68-70
    HIPCHK(hipEventRecord(ev1_, stream_));
    HIPCHK(hipEventSynchronize(ev1_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
    solve_ms = ms;
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
    std::vector<zc> e0((size_t)Kloc * ldw, zc(0.0));
    for (int kk = 0; kk < Kloc; ++kk) for (int j = 0; j < m0; ++j) e0[(size_t)kk * ldw + j] = -weights_[local_[kk]];
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
    rows[4].ms = ms; rows[4].launches = 1; rows[4].bytes = ((double)Kloc + 2.0) * n * ldw * 16.0;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 0;
}

// ---------------------------------------------------------------------------
// NLFEAST  (src/nlfeast.jl:2-84)
// ---------------------------------------------------------------------------
int Engine::nl_orthonormalize_x() {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "nl_orthonormalize_x: call set_x first");
    HIPCHK(hipSetDevice(device_));
    return orthonormalize(d_X_.p, d_T1_.p, nullptr);       // src/nlfeast.jl:12-13
}

int Engine::nl_contour_step(int first) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "nl_contour_step: call set_x first");
    HIPCHK(hipSetDevice(device_));
    if (!factored_) { int r = factorize(); if (r < 0) return r; }
    std::vector<zc> e0((size_t)Kloc * ldw, zc(0.0)), e1((size_t)Kloc * ldw, zc(0.0));
    for (int kk = 0; kk < Kloc; ++kk) {
        zc z = nodes_[local_[kk]], w = weights_[local_[kk]];
        for (int j = 0; j < m0; ++j) {
            zc e = first ? w : w / (z - lambda_[j]);           // src/nlfeast.jl:42 / :47
            e0[(size_t)kk * ldw + j] = e;
            e1[(size_t)kk * ldw + j] = e * z;                  // :58
        }
    }
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
              d_Rm_.alloc(wide) && d_MXm_.alloc(wide * std::max(2, nterms)) && d_em_.alloc(2 * (size_t)mm_ * Kloc * ldw) &&
              d_smallM_.alloc((size_t)8 * ldM_ * ldM_) && d_lamM_.alloc(ldM_) && d_gramM_.alloc(gram_scratch_elems((int)hrows, ldM_)) &&
              d_nrmM_.alloc(ldM_) && d_nrmMs_.alloc(colnorm_scratch_elems((int)hrows, ldM_)) && d_colidx_.alloc(ldM_);
    if (!ok) return fail(FEASTHIP_EHIP, "out of device memory for the moment buffers");
    return 0;
}

// Q_j = sum_k w_k z_k^j T(z_k)^-1 X (first pass, :193-212) or sum_k w_k z_k^j / (z_k - lam) .* (X - T(z_k)^-1 R) (:246-268)
int Engine::nlm_contour_step(int first) {
    HIPCHK(hipSetDevice(device_));
    const int L = 2 * mm_;
    std::vector<zc> e((size_t)L * Kloc * ldw, zc(0.0));
    for (int kk = 0; kk < Kloc; ++kk) {
        zc z = nodes_[local_[kk]], w = weights_[local_[kk]];
        for (int j = 0; j < m0; ++j) {
            zc v = first ? w : w / (z - lambda_[j]);
            for (int q = 0; q < L; ++q) { e[((size_t)q * Kloc + kk) * ldw + j] = v; v *= z; }
        }
    }
    HIPCHK(hipMemcpyAsync(d_em_.p, e.data(), e.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipEventRecord(ev0_, stream_));
    int r = solve_all(first ? d_X_.p : d_R_.p);
    if (r) return r;
    size_t blk = (size_t)n * ldw, eb = (size_t)Kloc * ldw;
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
    if (rk == 0) return fail(FEASTHIP_ENUM, "nlfeast_moments: the moment matrix is numerically zero");
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
    launch_accumulate(stream_, dsym_, d_X_.p, d_W_.p, (int64_t)blk, ldw, Kh, d_e0_.p, nullptr, 1, Qp(), nullptr);                          // :232-234
    launch_accumulate(stream_, dsym_, d_Xl_.p, d_W_.p + (size_t)Kh * blk, (int64_t)blk, ldw, Kh, d_e0_.p + (size_t)Kh * ldw, nullptr, 1,
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
    if (nterms != 4 || (K & 1) || node_coef_.empty() || shard_stride_ != 1 || mixed_)
        return fail(FEASTHIP_ESTATE, "dual_gen_feast: needs the terms {A, B, A', B'}, 2K nodes with a node coefficient table, one GPU, fp64");
    if (!Xl) return fail(FEASTHIP_EARG, "dual_gen_feast: null Xl");
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
int Engine::feast_run(void* X, int64_t n_, int m0_, int iter, double eps, int store, double* lambda, double* res,
                      uint8_t* inside, int* nit_out) {
    int r = set_x(X, n_, m0_);
    if (r) return r;
    int warn = 0;
    if (store) { r = factorize(); if (r < 0) return r; warn = r; }                 // src/feast.jl:28-38
    int nit = 0;
    for (nit = 0; nit <= iter; ++nit) {                                             // :40
        r = rr_step(lambda, res, inside);
        if (r) return r;
        bool any = false;
        double mx = 0;
        for (int j = 0; j < m0; ++j) if (inside[j]) { any = true; mx = std::max(mx, res[j]); }
        if (any && mx < eps) break;                                                 // :53-56
        if (nit < iter) { r = contour_step(store ? 0 : 1); if (r < 0) return r; if (r > 0) warn = r; }   // :57-72
    }
    if (nit_out) *nit_out = std::min(nit, iter);
    r = get_block(0, X);
    if (r) return r;
    return warn;
}

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

int Engine::ldiv(void* Y, int64_t ldy, const void* R, int64_t ldr, int m0_) {
    if (!Y || !R || m0_ <= 0 || ldy < n || ldr < n) return fail(FEASTHIP_EARG, "ldiv: bad arguments");
    if (!factored_) return fail(FEASTHIP_ESTATE, "ldiv: not factorised");
    HIPCHK(hipSetDevice(device_));
    if (m0_ != m0 || !have_x_) {
        m0 = m0_;
        ldw = ((m0 + 15) / 16) * 16;
        int r = alloc_iteration_buffers();
        if (r) return r;
        have_x_ = true;
        lambda_.assign(m0, zc(0.0));
    }
    // stage R column by column when ldr != n
    if (ldr == n) HIPCHK(hipMemcpyAsync(d_io_.p, R, (size_t)n * m0 * sizeof(cz), hipMemcpyHostToDevice, stream_));
    else HIPCHK(hipMemcpy2DAsync(d_io_.p, (size_t)n * sizeof(cz), R, (size_t)ldr * sizeof(cz), (size_t)n * sizeof(cz), m0, hipMemcpyHostToDevice, stream_));
    launch_cm_to_rm(stream_, n, m0, ldw, d_io_.p, n, d_R_.p);
    int r = solve_all(d_R_.p);
    if (r) return r;
    std::vector<zc> e0((size_t)Kloc * ldw, zc(0.0));
    for (int j = 0; j < ldw; ++j) e0[j] = 1.0;        // node 0 only
    HIPCHK(hipMemcpyAsync(d_e0_.p, e0.data(), e0.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    accumulate_solutions(1, d_e0_.p, nullptr, 0, d_T1_.p, nullptr);
    launch_rm_to_cm(stream_, n, m0, ldw, d_T1_.p, d_io_.p, n);
    if (ldy == n) HIPCHK(hipMemcpyAsync(Y, d_io_.p, (size_t)n * m0 * sizeof(cz), hipMemcpyDeviceToHost, stream_));
    else HIPCHK(hipMemcpy2DAsync(Y, (size_t)ldy * sizeof(cz), d_io_.p, (size_t)n * sizeof(cz), (size_t)n * sizeof(cz), m0, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace feasthip
