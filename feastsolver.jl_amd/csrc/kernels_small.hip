// The m0 x m0 reduced Hermitian(-definite) eigenproblem of the Rayleigh-Ritz step on the device (src/feast.jl:45, 122:
// `eigen!(Aq)` / `eigen!(Aq, Bq)`), m0 <= 64, ONE workgroup of 1024 threads, everything in LDS.
//
// Rounds 1-3 solved it on one host core (csrc/smalleig.cpp: 0.55 ms for m0 = 64, two host synchronisations around it while the
// GPU idles).  Here:  Ga = Q^H A Q, Gb = Q^H B Q (raw Gram matrices of the UNorthonormalised filtered block Q) ->
//   D = diag(Gb)^-1/2 (column scaling), Cholesky D Gb D = R^H R, T = R^-1, C = T^H (D Ga D) T          (f64 matrix cores)
//   C = V diag(lam) V^H by cyclic Jacobi, 32 independent rotations per step (round-robin pairing), one barrier per step
//   Xq = D T V, columns ordered by ascending lam: B-orthonormal Ritz vectors in the basis Q.
// In the steady state of a FEAST run Q = rho(B^-1 A) X with X the previous Ritz vectors, so Ga, Gb and with them C are nearly
// diagonal and the Jacobi iteration starts warm: 2-3 sweeps instead of 8-9.
// The kernel reports (status[]) the Hermitian defect of its inputs, the conditioning of the scaled basis (||R||_F ||R^-1||_F),
// failed pivots and the number of sweeps; the host falls back to the orthonormalising path (engine_rr.hip) on any of them.
#include "kernels.hpp"

namespace feasthip {

constexpr int SM_N = 64;           // padded order
constexpr int SM_LD = 65;          // LDS row stride in complex entries
constexpr int SM_NT = 1024;
constexpr int SM_MAX_SWEEPS = 24;

struct SmStage { double alpha, gamma; cz beta; };      // what a rotation is computed from: C[p][p], C[q][q], C[p][q]

struct SmRot { double c, sx, sy, on; };                 // J = [[c, s], [-conj(s), c]], on = 0: identity

struct SmShared {
    SmRot rot[2][SM_N / 2];
    double dinv[SM_N];
    double lam[SM_N];
    double red[16];
    double thr2, scale, pinv[2];
    double rdinv[SM_N];              // 1 / R[i][i]
    int rank[SM_N];
    int flags;
};

__device__ __forceinline__ double sm_block_max(double v, SmShared& sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh.red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) r = fmax(r, sh.red[w]);
    return r;
}

__device__ __forceinline__ double sm_block_sum(double v, SmShared& sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) r += sh.red[w];
    return r;
}

// value of lane (lane rotated right by n within its row of 16): DPP row_ror:n = 0x120 + n
template <int CTRL>
__device__ __forceinline__ double sm_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// One 16 x 16 tile (ti, tj) of a 64 x 64 x 64 complex product per wave on the f64 matrix cores, 3M: fa(i, k) and fb(k, j) return
// the operands.  Lane (li, lk) ends with the entries (16 ti + lk + 4 r, 16 tj + li), r < 4, in out[r].
template <class FA, class FB>
__device__ __forceinline__ void sm_mm_tile(int ti, int tj, FA fa, FB fb, cz (&out)[4]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    d4 t1 = (d4){0, 0, 0, 0}, t2 = (d4){0, 0, 0, 0}, t3 = (d4){0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < SM_N / 4; ++ks) {
        const cz a = fa(16 * ti + li, 4 * ks + lk);
        const cz b = fb(4 * ks + lk, 16 * tj + li);
        t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, t2, 0, 0, 0);
        t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, b.x + b.y, t3, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = cz_make(t1[r] - t2[r], t3[r] - t1[r] - t2[r]);
}

// 1 / sqrt(x) and 1 / x from the hardware estimates plus two Newton steps (the IEEE square-root and division sequences are 30-40
// instructions each, and one f64 instruction of a wave occupies its SIMD for four cycles)
__device__ __forceinline__ double sm_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ double sm_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

// rotation J = [[c, su], [-conj(su), c]] that diagonalises [[alpha, beta], [conj(beta), gamma]]; identity below the threshold.
// d = (gamma - alpha) / 2, r = sqrt(d^2 + |beta|^2), t = sign(d) |beta| / (|d| + r), c = 1 / sqrt(1 + t^2), su = c t beta / |beta|
__device__ __forceinline__ void sm_rotation(const SmStage& s, double thr2, double& c, cz& su, double& ab) {
    const double ab2 = s.beta.x * s.beta.x + s.beta.y * s.beta.y;
    c = 1.0; su = cz_make(0.0, 0.0); ab = 0.0;
    if (!(ab2 > thr2) || !(ab2 < 1e280)) return;
    const double d = 0.5 * (s.gamma - s.alpha);
    const double q = d * d + ab2;
    if (!(q < 1e280)) return;
    const double rinv = sm_rsqrt(q), rr = q * rinv;
    ab = ab2 * sm_rsqrt(ab2);
    const double g = copysign(1.0, d) * sm_rcp(fabs(d) + rr);         // t / |beta|
    const double t = ab * g;
    c = sm_rsqrt(1.0 + t * t);
    su = cz_make(c * g * s.beta.x, c * g * s.beta.y);
}

// status[0] flags (bit 0 inputs not Hermitian to 1e-13, 1 Gb not positive definite / non-finite, 2 basis ill-conditioned, 3 no convergence)
// status[1] sweeps, [2] ||R^-1||_F / sqrt(m) of the scaled basis, [3] Hermitian defect, [4] min / max of diag(R), [5..7] microseconds spent before,
// in and after the Jacobi iteration
__global__ __launch_bounds__(SM_NT) void k_rr_small_herm(int m, const cz* __restrict__ Ga, const cz* __restrict__ Gb, cz* Xq, cz* lam_out, int ldw,
                                                         double* status, cz* tscratch, double cond_max) {
    extern __shared__ unsigned char sm_raw[];
    cz* P = reinterpret_cast<cz*>(sm_raw);                         // Gb -> R -> T -> V
    cz* Cm = P + SM_N * SM_LD;                                     // Ga -> W -> C
    SmShared& sh = *reinterpret_cast<SmShared*>(Cm + SM_N * SM_LD);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = tid & 63, cq = tid >> 6;                         // entry layout of the elementwise phases: (r, cq + 16 e)
    const int mp = m + (m & 1), npair = mp >> 1;
    if (tid == 0) sh.flags = 0;
    const unsigned long long tick0 = wall_clock64();           // 100 MHz

    // ---- load, symmetrise, Hermitian defect, column scaling -------------------------------------------------------------
    // (column-major inputs read along their columns, the mirror entries through LDS: the transposed global reads of the first version
    // were 64 separate lines per load)
    cz ga[4], gb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq + 16 * e;
        const bool in = r < m && c < m;
        ga[e] = in ? Ga[r + (size_t)c * m] : cz_make(0.0, 0.0);
        gb[e] = in ? Gb[r + (size_t)c * m] : cz_make(r == c ? 1.0 : 0.0, 0.0);
        Cm[r * SM_LD + c] = ga[e];
        P[r * SM_LD + c] = gb[e];
    }
    __syncthreads();
    double def = 0.0, big = 0.0, defb = 0.0, bigb = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq + 16 * e;
        const cz a1 = ga[e], a2 = Cm[c * SM_LD + r], b1 = gb[e], b2 = P[c * SM_LD + r];
        ga[e] = cz_make(0.5 * (a1.x + a2.x), 0.5 * (a1.y - a2.y));
        gb[e] = cz_make(0.5 * (b1.x + b2.x), 0.5 * (b1.y - b2.y));
        // relative to the size of each matrix: tracked separately through two maxima
        def = fmax(def, fmax(fabs(a1.x - a2.x), fabs(a1.y + a2.y)));
        big = fmax(big, fmax(fabs(a1.x), fabs(a1.y)));
        defb = fmax(defb, fmax(fabs(b1.x - b2.x), fabs(b1.y + b2.y)));
        bigb = fmax(bigb, fmax(fabs(b1.x), fabs(b1.y)));
    }
    {
        const double da = sm_block_max(def, sh), ba = sm_block_max(big, sh), db = sm_block_max(defb, sh), bb = sm_block_max(bigb, sh);
        const double hd = fmax(ba > 0.0 ? da / ba : 0.0, bb > 0.0 ? db / bb : 0.0);
        if (tid == 0) { status[3] = hd; if (!(hd <= 1e-13)) sh.flags |= 1; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (cq + 16 * e == r) {
            const double d = gb[e].x;
            const bool ok = d > 0.0 && d < 1e300;
            sh.dinv[r] = ok ? 1.0 / sqrt(d) : 1.0;
            if (!ok && r < m) atomicOr(&sh.flags, 2);
        }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq + 16 * e;
        const double s = sh.dinv[r] * sh.dinv[c];
        P[r * SM_LD + c] = cz_make(gb[e].x * s, gb[e].y * s);
        Cm[r * SM_LD + c] = cz_make(ga[e].x * s, ga[e].y * s);
    }
    __syncthreads();

    const unsigned long long tickA = wall_clock64();
    // ---- Cholesky of the scaled Gb, upper factor, right-looking: row j of the Schur complement is final after step j - 1 ------
    // (the reciprocal of the next pivot is published by the thread that finishes it: one division per step, not 1024)
    if (tid == 0) { const double p0 = P[0].x; sh.pinv[0] = p0 > 0.0 ? 1.0 / p0 : 1.0; if (!(p0 > 0.0)) sh.flags |= 2; }
    __syncthreads();
    {
        // wave w owns rows 4 w .. 4 w + 3 (16 lanes per row, columns cl + 16 e): a wave whose rows are all finished skips the step
        const int crow = 4 * wave + (lane >> 4), cl = lane & 15;
        for (int j = 0; j < m; ++j) {
            const double pinv = sh.pinv[j & 1];                 // (two slots: step j + 1 writes the other one)
            if (4 * wave + 3 > j && crow > j && crow < m) {
                const cz lrj = P[j * SM_LD + crow];             // S[j][row]
                const cz f = cz_make(lrj.x * pinv, -lrj.y * pinv);          // conj(S[j][row]) / S[j][j]
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cl + 16 * e;
                    if (c >= crow) {
                        cz v = P[crow * SM_LD + c]; cz_msub(v, f, P[j * SM_LD + c]); P[crow * SM_LD + c] = v;
                        if (crow == j + 1 && c == crow) { sh.pinv[(j + 1) & 1] = v.x > 0.0 ? 1.0 / v.x : 1.0; if (!(v.x > 0.0)) atomicOr(&sh.flags, 2); }
                    }
                }
            }
            __syncthreads();
        }
    }
    const unsigned long long tickB = wall_clock64();
    double dmin = 1e300, dmax = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq + 16 * e;
        const double piv = P[r * SM_LD + r].x;
        const double s = piv > 0.0 ? 1.0 / sqrt(piv) : 1.0;
        cz v = P[r * SM_LD + c];
        if (c == r) {
            sh.rdinv[r] = s;
            if (r < m) { dmin = fmin(dmin, sqrt(fmax(piv, 0.0))); dmax = fmax(dmax, sqrt(fmax(piv, 0.0))); }
        }
        gb[e] = (c >= r) ? cz_make(v.x * s, v.y * s) : cz_make(0.0, 0.0);          // R[r][c]
    }
    dmin = -sm_block_max(-dmin, sh);
    dmax = sm_block_max(dmax, sh);
#pragma unroll
    for (int e = 0; e < 4; ++e) P[r * SM_LD + cq + 16 * e] = gb[e];
    __syncthreads();

    // ---- T = R^-1: 16 lanes per column c, lane l holds T[l + 16 e][c]; back substitution over the rows, no barriers -------------
    {
        const int c = tid >> 4, l = tid & 15;
        cz t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = cz_make((l + 16 * e == c && c >= m) ? 1.0 : 0.0, 0.0);       // identity in the padding
        for (int i = m - 1; i >= 0; --i) {
            if (4 * wave + 3 < i) continue;                      // all four columns of this wave lie left of row i: T[i][c] = 0
            cz part = cz_make(0.0, 0.0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = l + 16 * e;
                if (k > i && k <= c) cz_madd(part, P[i * SM_LD + k], t[e]);
            }
            // sum over the 16 lanes of the column: rotations within the row of 16 (DPP, no LDS round trip)
            part.x += sm_dpp<0x128>(part.x); part.y += sm_dpp<0x128>(part.y);
            part.x += sm_dpp<0x124>(part.x); part.y += sm_dpp<0x124>(part.y);
            part.x += sm_dpp<0x122>(part.x); part.y += sm_dpp<0x122>(part.y);
            part.x += sm_dpp<0x121>(part.x); part.y += sm_dpp<0x121>(part.y);
            const double rii = sh.rdinv[i];                      // 1 / R[i][i], real positive diagonal
            cz ti = cz_make(0.0, 0.0);
            if (i == c) ti = cz_make(rii, 0.0);
            else if (i < c) ti = cz_make(-part.x * rii, -part.y * rii);
            if ((i & 15) == l) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if ((i >> 4) == e) t[e] = ti;
            }
        }
        double tf = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c < m && l + 16 * e < m) tf += t[e].x * t[e].x + t[e].y * t[e].y;
        tf = sm_block_sum(tf, sh);                 // (barriers: every read of R is done)
#pragma unroll
        for (int e = 0; e < 4; ++e) P[(l + 16 * e) * SM_LD + c] = t[e];
        const double cond = sqrt(tf / (double)m);                 // ||R||_F ||R^-1||_F / m with ||R||_F^2 = trace of the scaled Gb = m: 1 for an orthonormal block
        if (tid == 0) {
            status[2] = cond; status[4] = dmax > 0.0 ? dmin / dmax : 0.0;
            if (!(cond <= cond_max)) sh.flags |= 4;
        }
    }
    __syncthreads();

    const unsigned long long tickC = wall_clock64();
    // ---- C = T^H (D Ga D) T on the matrix cores; T goes to global scratch for the back-transform ---------------------------------
    const int ti = wave >> 2, tj = wave & 3, li = lane & 15, lk = lane >> 4;
    {
        cz w[4];
        sm_mm_tile(ti, tj, [&](int i, int k) { return Cm[i * SM_LD + k]; }, [&](int k, int j) { return P[k * SM_LD + j]; }, w);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) Cm[(16 * ti + lk + 4 * q) * SM_LD + 16 * tj + li] = w[q];
        __syncthreads();
        sm_mm_tile(ti, tj, [&](int i, int k) { return cz_conj(P[k * SM_LD + i]); }, [&](int k, int j) { return Cm[k * SM_LD + j]; }, w);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) Cm[(16 * ti + lk + 4 * q) * SM_LD + 16 * tj + li] = w[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cq + 16 * e;
            tscratch[r * SM_N + c] = P[r * SM_LD + c];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cq + 16 * e;
            P[r * SM_LD + c] = cz_make(r == c ? 1.0 : 0.0, 0.0);       // V = I
        }
    }
    // scale of C for the rotation threshold
    {
        double mx = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const cz v = Cm[r * SM_LD + cq + 16 * e]; mx = fmax(mx, fabs(v.x) + fabs(v.y)); }
        mx = sm_block_max(mx, sh);
        if (tid == 0) { sh.scale = mx; sh.thr2 = (1e-16 * mx) * (1e-16 * mx); }
    }
    __syncthreads();

    // ---- cyclic Jacobi, round-robin pairing: at step s position k holds index arr(k) = k ? 1 + (k - 1 + s) mod (mp - 1) : 0, pair a =
    //      (arr(a), arr(mp - 1 - a)).  A step has two phases with a barrier behind each.
    //      1: thread t < nblk owns the 2 x 2 block rows pair a x columns pair b, a <= b, of the Hermitian C for this step (in place, the
    //         mirror entries written with it: half the products of the full update) and applies the rotations rot[s & 1].
    //      2: the lanes of wave 0 compute the rotations of step s + 1 from the updated C (one wave, once per pair: the square roots and
    //         divisions of 1024 redundant copies were most of the first version's 5 us per step) WHILE the other waves take V <- V J
    //         of step s, one pair of columns of V^T per wave (lane = row; a wave whose pair did not rotate skips).
    //      One f64 vector instruction of a 64-lane wave occupies its SIMD for four cycles and the 16 waves share the four SIMDs of the
    //      one CU this runs on: a step is bound by the ~67k multiply-adds of the two updates.
    const unsigned long long tick1 = wall_clock64();
    const int period = mp - 1;
    const int nblk = npair * (npair + 1) / 2;
    int ba = 0, bb = 0;                              // this thread's block (a <= b), fixed for the whole iteration
    {
        int t = tid;
        while (ba < npair && t >= npair - ba) { t -= npair - ba; ++ba; }
        bb = ba + t;
    }
    const bool cact = tid < nblk;
    auto arr_at = [&](int k, int s) { int v = k - 1 + s; v = v >= period ? v - period : v; return k == 0 ? 0 : 1 + v; };
    auto rotations = [&](int s, int buf, double& maxb) {       // lanes of wave 0: rotation of pair `tid` at step s from the current C
        const int p = arr_at(tid, s), q = arr_at(mp - 1 - tid, s);
        SmStage st;
        st.alpha = Cm[p * SM_LD + p].x; st.gamma = Cm[q * SM_LD + q].x; st.beta = Cm[p * SM_LD + q];
        double c, ab; cz su;
        sm_rotation(st, sh.thr2, c, su, ab);
        maxb = fmax(maxb, ab);
        sh.rot[buf][tid] = SmRot{c, su.x, su.y, ab > 0.0 ? 1.0 : 0.0};
    };
    int sweeps = 0;
    bool converged = (mp < 2);
    const double stop = 1e-9 * sh.scale;
    double maxb = 0.0;
    if (tid < npair) rotations(0, 0, maxb);
    __syncthreads();
    int gs = 0;                                      // global step counter: rot[gs & 1] holds this step's rotations
    for (; sweeps < SM_MAX_SWEEPS && !converged; ++sweeps) {
        for (int s = 0; s < period; ++s, ++gs) {
            const SmRot* rot = sh.rot[gs & 1];
            // phase 1
            if (cact) {
                const SmRot ra = rot[ba], rb = rot[bb];
                if (ra.on != 0.0 || rb.on != 0.0) {
                    const int pa = arr_at(ba, s), qa = arr_at(mp - 1 - ba, s), pb = arr_at(bb, s), qb = arr_at(mp - 1 - bb, s);
                    cz n00 = Cm[pa * SM_LD + pb], n01 = Cm[pa * SM_LD + qb], n10 = Cm[qa * SM_LD + pb], n11 = Cm[qa * SM_LD + qb];
                    const double ca = ra.c, cb = rb.c;
                    const cz sa = cz_make(ra.sx, ra.sy), sb = cz_make(rb.sx, rb.sy);
                    // M = Ja^H B,  Ja^H = [[ca, -sa], [conj(sa), ca]]
                    cz m00 = cz_make(ca * n00.x, ca * n00.y); cz_msub(m00, sa, n10);
                    cz m01 = cz_make(ca * n01.x, ca * n01.y); cz_msub(m01, sa, n11);
                    cz m10 = cz_make(ca * n10.x, ca * n10.y); cz_madd(m10, cz_conj(sa), n00);
                    cz m11 = cz_make(ca * n11.x, ca * n11.y); cz_madd(m11, cz_conj(sa), n01);
                    // B' = M Jb,  Jb = [[cb, sb], [-conj(sb), cb]]
                    n00 = cz_make(cb * m00.x, cb * m00.y); cz_msub(n00, m01, cz_conj(sb));
                    n01 = cz_make(cb * m01.x, cb * m01.y); cz_madd(n01, m00, sb);
                    n10 = cz_make(cb * m10.x, cb * m10.y); cz_msub(n10, m11, cz_conj(sb));
                    n11 = cz_make(cb * m11.x, cb * m11.y); cz_madd(n11, m10, sb);
                    if (ba == bb) { n01 = cz_make(0.0, 0.0); n10 = n01; n00.y = 0.0; n11.y = 0.0; }
                    Cm[pa * SM_LD + pb] = n00; Cm[pa * SM_LD + qb] = n01; Cm[qa * SM_LD + pb] = n10; Cm[qa * SM_LD + qb] = n11;
                    if (ba != bb) {
                        Cm[pb * SM_LD + pa] = cz_conj(n00); Cm[qb * SM_LD + pa] = cz_conj(n01);
                        Cm[pb * SM_LD + qa] = cz_conj(n10); Cm[qb * SM_LD + qa] = cz_conj(n11);
                    }
                }
            }
            __syncthreads();
            // phase 2
            if (wave == 0) {
                if (tid < npair) rotations(s + 1 == period ? 0 : s + 1, (gs + 1) & 1, maxb);
            } else {
                for (int b = wave - 1; b < npair; b += 15) {
                    const SmRot rb = rot[b];
                    if (rb.on == 0.0) continue;
                    const int pb = arr_at(b, s), qb = arr_at(mp - 1 - b, s);
                    const cz sb = cz_make(rb.sx, rb.sy);
                    const cz v0 = P[pb * SM_LD + lane], v1 = P[qb * SM_LD + lane];          // V^T: row = column of V
                    cz w0 = cz_make(rb.c * v0.x, rb.c * v0.y); cz_msub(w0, v1, cz_conj(sb));
                    cz w1 = cz_make(rb.c * v1.x, rb.c * v1.y); cz_madd(w1, v0, sb);
                    P[pb * SM_LD + lane] = w0; P[qb * SM_LD + lane] = w1;
                }
            }
            __syncthreads();
        }
        // the rotations counted in maxb: those of steps 1 .. period of this sweep (the first of the next sweep among them, the first
        // of this one was counted by the previous sweep): a full cycle either way
        maxb = sm_block_max(maxb, sh);
        converged = maxb <= stop;          // quadratic convergence: what is left after this sweep is of order maxb^2 / gap
        maxb = 0.0;
    }
    const unsigned long long tick2 = wall_clock64();
    if (tid == 0) {
        if (!converged) sh.flags |= 8;
        status[1] = (double)sweeps;
    }

    // ---- eigenvalues in ascending order, Xq = D T V with the columns in that order ------------------------------------------------
    if (tid < SM_N) sh.lam[tid] = tid < m ? Cm[tid * SM_LD + tid].x : 1e308;
    __syncthreads();
    if (tid < m) {
        const double lj = sh.lam[tid];
        int rk = 0;
        for (int i = 0; i < m; ++i) { const double lv = sh.lam[i]; rk += (lv < lj || (lv == lj && i < tid)) ? 1 : 0; }
        sh.rank[tid] = rk;
        lam_out[rk] = cz_make(lj, 0.0);
    }
    if (tid >= m && tid < ldw) lam_out[tid] = cz_make(0.0, 0.0);
    __syncthreads();
    {
        cz y[4];
        sm_mm_tile(ti, tj, [&](int i, int k) { return tscratch[i * SM_N + k]; }, [&](int k, int j) { return P[j * SM_LD + k]; }, y);      // P holds V^T
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * ti + lk + 4 * q, j = 16 * tj + li;
            if (i < m && j < m) { const double d = sh.dinv[i]; Xq[i + (size_t)sh.rank[j] * m] = cz_make(y[q].x * d, y[q].y * d); }
        }
    }
    if (tid == 0) {
        status[0] = (double)sh.flags;
        status[8] = 0.01 * (double)(tickA - tick0); status[9] = 0.01 * (double)(tickB - tickA); status[10] = 0.01 * (double)(tickC - tickB);
        status[11] = 0.01 * (double)(tick1 - tickC);      // load / Cholesky / inverse / congruence
        status[5] = 0.01 * (double)(tick1 - tick0); status[6] = 0.01 * (double)(tick2 - tick1); status[7] = 0.01 * (double)(wall_clock64() - tick2);     // us
    }
}

size_t rr_small_lds_bytes() { return (size_t)2 * SM_N * SM_LD * sizeof(cz) + sizeof(SmShared); }

// Xq: m x m column-major; lam_out: ldw entries (zero beyond m); status: 8 doubles; tscratch: 64 x 64 entries
void launch_rr_small_herm(hipStream_t st, int m, const cz* Ga, const cz* Gb, cz* Xq, cz* lam_out, int ldw, double* status, cz* tscratch, double cond_max) {
    static std::atomic<uint64_t> attr_seen{0};
    if (first_on_this_device(attr_seen)) { (void)hipFuncSetAttribute((const void*)k_rr_small_herm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rr_small_lds_bytes()); }
    hipLaunchKernelGGL(k_rr_small_herm, dim3(1), dim3(SM_NT), rr_small_lds_bytes(), st, m, Ga, Gb, Xq, lam_out, ldw, status, tscratch, cond_max);
}

// X <- X diag(s), R = (AX - BX diag(lam)) diag(s) with s_j = 1 / sqrt(nrm2[j]) (update_R! normalises the columns of X, src/utils.jl:111-116);
// BX = X diag(rowscale) rows when rowscale is given (diagonal B), else a separate block
__global__ void k_residual_combine_scaled(int64_t total, int ldw, const cz* __restrict__ AX, const cz* BX, const cz* __restrict__ lam, cz* X, cz* R,
                                          const cz* __restrict__ rowscale, const double* __restrict__ nrm2) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e % ldw);
    const double n2 = nrm2[j];
    const double s = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
    cz r = AX[e];
    const cz x = X[e];
    cz bx = rowscale ? cz_mul(x, rowscale[e / ldw]) : BX[e];
    cz_msub(r, bx, lam[j]);
    R[e] = cz_make(r.x * s, r.y * s);
    X[e] = cz_make(x.x * s, x.y * s);
}

void launch_residual_combine_scaled(hipStream_t st, int n, int ldw, const cz* AX, const cz* BX, const cz* lam, cz* X, cz* R, const cz* rowscale,
                                    const double* nrm2) {
    int64_t total = (int64_t)n * ldw;
    hipLaunchKernelGGL(k_residual_combine_scaled, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, ldw, AX, BX, lam, X, R, rowscale, nrm2);
}

}  // namespace feasthip
