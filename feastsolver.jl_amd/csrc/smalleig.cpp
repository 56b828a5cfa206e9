#include "smalleig.hpp"
#include "hostpool.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace feasthip {
namespace {
inline double cabs1(zc z) { return std::fabs(z.real()) + std::fabs(z.imag()); }
const double EPS = 2.220446049250313e-16;

// sqrt(a^2 + b^2) without the overflow/underflow ceremony of libm's hypot (EISPACK pythag): one divide, one sqrt
inline double pythag(double a, double b) {
    a = std::fabs(a); b = std::fabs(b);
    if (a < b) std::swap(a, b);
    if (a == 0.0) return 0.0;
    double r = b / a;
    return a * std::sqrt(1.0 + r * r);
}

// |z| for the rotation parameters of the QR iteration: libm's hypot (what std::abs calls) guards against overflow and underflow at
// 15-20 ns a call, three calls per rotation on a chain of 140 ns; the plain formula is safe unless a component is beyond 1e+-140
inline double fast_abs(zc z) {
    const double a = std::fabs(z.real()), b = std::fabs(z.imag()), mx = a > b ? a : b;
    if (mx > 1e-140 && mx < 1e140) return std::sqrt(a * a + b * b);
    return pythag(a, b);
}

// Split storage (separate real and imaginary planes, column-major) keeps every inner loop a stride-1 real loop
// the compiler vectorises; these routines sit on the critical path of every FEAST iteration (GPU idle meanwhile).
struct Planes {
    int m;
    std::vector<double> re, im;
    explicit Planes(int m_) : m(m_), re((size_t)m_ * m_, 0.0), im((size_t)m_ * m_, 0.0) {}
    double* r(int j) { return re.data() + (size_t)j * m; }
    double* i(int j) { return im.data() + (size_t)j * m; }
    void load(const zc* A) { for (size_t q = 0; q < (size_t)m * m; ++q) { re[q] = A[q].real(); im[q] = A[q].imag(); } }
    void store(zc* A) const { for (size_t q = 0; q < (size_t)m * m; ++q) A[q] = zc(re[q], im[q]); }
    zc at(int i2, int j) const { return zc(re[i2 + (size_t)j * m], im[i2 + (size_t)j * m]); }
    void set(int i2, int j, zc v) { re[i2 + (size_t)j * m] = v.real(); im[i2 + (size_t)j * m] = v.imag(); }
};

// y[lo, hi) -= a * x[lo, hi)
inline void caxpy_sub(int lo, int hi, double ar, double ai, const double* __restrict xr, const double* __restrict xi,
                      double* __restrict yr, double* __restrict yi) {
    for (int i = lo; i < hi; ++i) {
        yr[i] -= ar * xr[i] - ai * xi[i];
        yi[i] -= ar * xi[i] + ai * xr[i];
    }
}

// Householder reduction to upper Hessenberg form, H = Z^H A Z.  Both one-sided updates run over columns (stride 1):
// the right-hand one as w = H v (a sum of columns) followed by H[:, j] -= 2 w conj(v_j).
void hessenberg(Planes& H, Planes& Z) {
    const int m = H.m;
    std::fill(Z.re.begin(), Z.re.end(), 0.0);
    std::fill(Z.im.begin(), Z.im.end(), 0.0);
    for (int i = 0; i < m; ++i) Z.re[i + (size_t)i * m] = 1.0;
    std::vector<double> vr(m), vi(m), wr(m), wi(m), zr(m), zi(m);
    for (int k = 0; k + 2 < m; ++k) {
        double* hk_r = H.r(k);
        double* hk_i = H.i(k);
        double nrm = 0;
        for (int i = k + 1; i < m; ++i) nrm += hk_r[i] * hk_r[i] + hk_i[i] * hk_i[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        zc x0(hk_r[k + 1], hk_i[k + 1]);
        double ax0 = std::abs(x0);
        zc ph = (ax0 == 0.0) ? zc(1.0) : x0 / ax0;
        zc alpha = -ph * nrm;
        double vn = 0;
        for (int i = k + 1; i < m; ++i) { vr[i] = hk_r[i]; vi[i] = hk_i[i]; }
        vr[k + 1] -= alpha.real(); vi[k + 1] -= alpha.imag();
        for (int i = k + 1; i < m; ++i) vn += vr[i] * vr[i] + vi[i] * vi[i];
        vn = std::sqrt(vn);
        if (vn == 0.0) continue;
        const double inv = 1.0 / vn;
        for (int i = k + 1; i < m; ++i) { vr[i] *= inv; vi[i] *= inv; }
        // H <- (I - 2 v v^H) H
        for (int j = k; j < m; ++j) {
            double* cr = H.r(j);
            double* ci = H.i(j);
            double sr = 0, si = 0;
            for (int i = k + 1; i < m; ++i) { sr += vr[i] * cr[i] + vi[i] * ci[i]; si += vr[i] * ci[i] - vi[i] * cr[i]; }
            caxpy_sub(k + 1, m, 2.0 * sr, 2.0 * si, vr.data(), vi.data(), cr, ci);
        }
        // H <- H (I - 2 v v^H),  Z <- Z (I - 2 v v^H)
        std::fill(wr.begin(), wr.end(), 0.0); std::fill(wi.begin(), wi.end(), 0.0);
        std::fill(zr.begin(), zr.end(), 0.0); std::fill(zi.begin(), zi.end(), 0.0);
        for (int j = k + 1; j < m; ++j) {
            caxpy_sub(0, m, -vr[j], -vi[j], H.r(j), H.i(j), wr.data(), wi.data());      // w += H[:, j] v_j
            caxpy_sub(0, m, -vr[j], -vi[j], Z.r(j), Z.i(j), zr.data(), zi.data());
        }
        for (int j = k + 1; j < m; ++j) {
            caxpy_sub(0, m, 2.0 * vr[j], -2.0 * vi[j], wr.data(), wi.data(), H.r(j), H.i(j));    // H[:, j] -= 2 w conj(v_j)
            caxpy_sub(0, m, 2.0 * vr[j], -2.0 * vi[j], zr.data(), zi.data(), Z.r(j), Z.i(j));
        }
        hk_r[k + 1] = alpha.real(); hk_i[k + 1] = alpha.imag();
        for (int i = k + 2; i < m; ++i) { hk_r[i] = 0.0; hk_i[i] = 0.0; }
    }
}

// [p; q] <- [c p + conj(s) q; -s p + c q] element-wise over [lo, hi): a Givens rotation applied to two columns
inline void rot_cols(int lo, int hi, double c, double sr, double si, double* __restrict pr, double* __restrict pi,
                     double* __restrict qr, double* __restrict qi) {
    for (int i = lo; i < hi; ++i) {
        double ar = pr[i], ai = pi[i], br = qr[i], bi = qi[i];
        pr[i] = c * ar + (sr * br + si * bi);
        pi[i] = c * ai + (sr * bi - si * br);
        qr[i] = c * br - (sr * ar - si * ai);
        qi[i] = c * bi - (sr * ai + si * ar);
    }
}

// (Round 5 handed the rotations of the Schur vectors -- Z <- Z G, which the iteration never reads -- to helper threads of the host pool
// through a chunked log, every helper owning blocks of 16 rows of Z; bit-identical results.  In the stand-alone probe
// (tools/probes/smalleig_bench.cpp, calls back to back) m = 128 went from 6.6 to 5.7 ms; inside a FEAST run, where the pool sleeps
// 10 ms between two Rayleigh-Ritz steps, the helpers arrive late and slow (cold cores): the caller finished 0.7 - 1 ms of their
// rotations itself and C5's step went from 6.2 to 6.6 ms.  Removed; the eigenvector and B^-1 A column loops below do use the pool.)
// Single-shift complex QR on an upper Hessenberg matrix; H becomes the Schur form T, Z the Schur vectors.
// H is kept TRANSPOSED (Ht(j, i) = H(i, j)) so that the row rotations, the longer of the two updates of H, run over
// contiguous memory like the rotations of Z; only the short column part (rows 0..k+2) is strided.
constexpr int QR_GRP = 8;
int hqr_schur(Planes& H, Planes& Z) {
    const int m = H.m;
    Planes T(m);
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { T.re[j + (size_t)i * m] = H.re[i + (size_t)j * m]; T.im[j + (size_t)i * m] = H.im[i + (size_t)j * m]; }
    auto h = [&](int i, int j) { return zc(T.re[j + (size_t)i * m], T.im[j + (size_t)i * m]); };
    auto hset = [&](int i, int j, zc v) { T.re[j + (size_t)i * m] = v.real(); T.im[j + (size_t)i * m] = v.imag(); };
    int ihi = m - 1, its = 0, total = 0;
    double hn = 0;
    for (size_t q = 0; q < (size_t)m * m; ++q) hn = std::max(hn, std::fabs(T.re[q]) + std::fabs(T.im[q]));
    if (hn == 0.0) return 0;
    int info = 0;
    while (ihi >= 0) {
        int l = ihi;
        for (; l > 0; --l) {
            double sub = cabs1(h(l, l - 1));
            double dd = cabs1(h(l - 1, l - 1)) + cabs1(h(l, l));
            if (dd == 0.0) dd = hn;
            if (sub <= EPS * dd) break;
        }
        if (l > 0) hset(l, l - 1, 0.0);
        if (l == ihi) { --ihi; its = 0; continue; }
        if (++total > 60 * m + 200) { info = 2; break; }
        zc s;
        if (its == 10 || its == 20) {
            s = zc(std::fabs(h(ihi, ihi - 1).real()) + (ihi >= 2 ? std::fabs(h(ihi - 1, ihi - 2).real()) : 0.0), 0.0) + h(ihi, ihi);
        } else {   // Wilkinson shift: eigenvalue of the trailing 2x2 closer to H[ihi][ihi]
            zc a = h(ihi - 1, ihi - 1), b = h(ihi - 1, ihi);
            zc c = h(ihi, ihi - 1), d = h(ihi, ihi);
            zc tr2 = 0.5 * (a + d), det = a * d - b * c;
            zc disc = std::sqrt(tr2 * tr2 - det);
            zc e1 = tr2 + disc, e2 = tr2 - disc;
            s = (std::abs(e1 - d) < std::abs(e2 - d)) ? e1 : e2;
        }
        ++its;
        zc x = h(l, l) - s, y = h(l + 1, l);
        // The rotations of a sweep are generated in groups of QR_GRP.  Each one is applied at once to its two rows (contiguous in T) and to
        // the few entries of its two columns the next rotations read (rows k0 - 1 .. k + 2); the rows above the group take the whole group
        // in one visit -- QR_GRP + 1 neighbouring entries of a T column loaded once, rotated QR_GRP times in registers, stored once --
        // instead of one strided visit per rotation (the strided part was 45 % of the iteration for m = 128).  The same rotations reach every entry
        // in the same order; QR iteration 9.8 -> 7.7 ms for m = 128 random, 5.8 -> 4.7 nearly diagonal, m = 64 unchanged (build host).
        const int grp = (ihi - l >= 80) ? QR_GRP : 1;       // short windows (and m = 64 altogether: measured 7 % slower grouped) keep one visit per rotation
        for (int k0 = l; k0 < ihi; k0 += grp) {
            const int ng = std::min(grp, ihi - k0);
            double gc[QR_GRP], gsr[QR_GRP], gsi[QR_GRP];
            const int near0 = grp == 1 ? 0 : std::max(0, k0 - 1);          // first row handled rotation by rotation
            for (int q = 0; q < ng; ++q) {
                const int k = k0 + q;
                // Givens G = [c sn; -conj(sn) c] with G [x;y] = [r;0]
                double ax = fast_abs(x), ay = fast_abs(y), c;
                zc sn;
                if (ay == 0.0) { c = 1.0; sn = 0.0; }
                else if (ax == 0.0) { c = 0.0; sn = std::conj(y) / ay; }
                else { double nr = std::sqrt(ax * ax + ay * ay); c = ax / nr; sn = (x / ax) * std::conj(y) / nr; }
                gc[q] = c; gsr[q] = sn.real(); gsi[q] = sn.imag();
                int j0 = std::max(l, k - 1);
                // rows k, k+1 of H = columns k, k+1 of T:  [h0; h1] <- [c h0 + sn h1; -conj(sn) h0 + c h1]
                rot_cols(j0, m, c, sn.real(), -sn.imag(), T.r(k), T.i(k), T.r(k + 1), T.i(k + 1));
                if (k > l) hset(k + 1, k - 1, 0.0);
                int i1 = std::min(k + 2, ihi);
                for (int i = near0; i <= i1; ++i) {         // columns k, k+1 of H, rows near the diagonal (strided in T)
                    zc h0 = h(i, k), h1 = h(i, k + 1);
                    hset(i, k, c * h0 + std::conj(sn) * h1);
                    hset(i, k + 1, -sn * h0 + c * h1);
                }
                rot_cols(0, m, c, sn.real(), sn.imag(), Z.r(k), Z.i(k), Z.r(k + 1), Z.i(k + 1));
                if (k + 1 < ihi) { x = h(k + 1, k); y = h(k + 2, k); }
            }
            // rows 0 .. near0 - 1: columns k0 .. k0 + ng of H = entries k0 .. k0 + ng of column i of T; four rows per pass (independent
            // chains of dependent rotations: the loads and the arithmetic of one hide the latencies of the others)
            {
                int i = 0;
                for (; i + 4 <= near0; i += 4) {
                    double* tr[4]; double* ti[4]; double ar[4], ai[4];
                    for (int u = 0; u < 4; ++u) { tr[u] = T.re.data() + (size_t)(i + u) * m + k0; ti[u] = T.im.data() + (size_t)(i + u) * m + k0; ar[u] = tr[u][0]; ai[u] = ti[u][0]; }
                    for (int q = 0; q < ng; ++q) {
                        const double c = gc[q], sr = gsr[q], si = gsi[q];
                        for (int u = 0; u < 4; ++u) {
                            const double br = tr[u][q + 1], bi = ti[u][q + 1];
                            tr[u][q] = c * ar[u] + (sr * br + si * bi);
                            ti[u][q] = c * ai[u] + (sr * bi - si * br);
                            const double nr = c * br - (sr * ar[u] - si * ai[u]), ni = c * bi - (sr * ai[u] + si * ar[u]);
                            ar[u] = nr; ai[u] = ni;
                        }
                    }
                    for (int u = 0; u < 4; ++u) { tr[u][ng] = ar[u]; ti[u][ng] = ai[u]; }
                }
                for (; i < near0; ++i) {
                    double* tr = T.re.data() + (size_t)i * m + k0;
                    double* ti = T.im.data() + (size_t)i * m + k0;
                    double ar = tr[0], ai = ti[0];
                    for (int q = 0; q < ng; ++q) {
                        const double c = gc[q], sr = gsr[q], si = gsi[q], br = tr[q + 1], bi = ti[q + 1];
                        tr[q] = c * ar + (sr * br + si * bi);
                        ti[q] = c * ai + (sr * bi - si * br);
                        const double nr = c * br - (sr * ar - si * ai), ni = c * bi - (sr * ai + si * ar);
                        ar = nr; ai = ni;
                    }
                    tr[ng] = ar; ti[ng] = ai;
                }
            }
        }
    }
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { H.re[i + (size_t)j * m] = T.re[j + (size_t)i * m]; H.im[i + (size_t)j * m] = T.im[j + (size_t)i * m]; }
    return info;
}
}  // namespace

void matmul(int m, const zc* A, const zc* B, zc* C) {
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i < m; ++i) C[i + (size_t)j * m] = 0.0;
        for (int k = 0; k < m; ++k) {
            zc b = B[k + (size_t)j * m];
            for (int i = 0; i < m; ++i) C[i + (size_t)j * m] += A[i + (size_t)k * m] * b;
        }
    }
}

int eig_general(int m, zc* A, zc* lam, zc* V) {
    if (m <= 0) return 0;
    Planes T(m), Z(m);
    T.load(A);
    static const bool trace = getenv("FEASTHIP_EIG_TRACE") != nullptr;
    auto tp0 = std::chrono::steady_clock::now();
    hessenberg(T, Z);
    auto tp1 = std::chrono::steady_clock::now();
    int info = hqr_schur(T, Z);
    auto tp2 = std::chrono::steady_clock::now();
    if (info) return info;
    double tn = 0;
    for (int j = 0; j < m; ++j) for (int i = 0; i <= j; ++i) tn = std::max(tn, std::fabs(T.r(j)[i]) + std::fabs(T.i(j)[i]));
    double small = std::max(EPS * tn, 1e-300);
    std::vector<zc> vals(m), vecs((size_t)m * m);
    // the eigenvectors are independent of each other: interleaved column sets on the host pool (column k costs ~ k^2 / 2 + k m); the
    // caller takes part, so a pool that is slow to wake up only means the caller does more of them
    const int nvp = (m >= 48 && pool_size() >= 2) ? std::min(8, pool_size() + 1) : 1;
    run_parts(nvp, [&](int part) {
    std::vector<double> xr(m), xi(m), wr(m), wi(m);
    for (int k = part; k < m; k += nvp) {
        // eigenvector of the triangular T for t = T_kk: back substitution in column (axpy) form, x_k = 1
        zc t = T.at(k, k);
        vals[k] = t;
        for (int j = 0; j < k; ++j) { xr[j] = -T.r(k)[j]; xi[j] = -T.i(k)[j]; }     // right-hand side: -T[0:k, k]
        xr[k] = 1.0; xi[k] = 0.0;
        double scale_guard = 1.0;
        for (int i = k - 1; i >= 0; --i) {
            zc d = T.at(i, i) - t;
            if (cabs1(d) < small) d = small;
            zc x = zc(xr[i], xi[i]) / d;
            if (cabs1(x) > 1e100) {          // rescale to avoid overflow
                for (int j = 0; j <= k; ++j) { xr[j] *= 1e-100; xi[j] *= 1e-100; }
                x *= 1e-100; scale_guard *= 1e-100;
            }
            xr[i] = x.real(); xi[i] = x.imag();
            caxpy_sub(0, i, x.real(), x.imag(), T.r(i), T.i(i), xr.data(), xi.data());     // x[0:i] -= T[0:i, i] x_i
        }
        (void)scale_guard;
        std::fill(wr.begin(), wr.end(), 0.0); std::fill(wi.begin(), wi.end(), 0.0);
        for (int j = 0; j <= k; ++j) caxpy_sub(0, m, -xr[j], -xi[j], Z.r(j), Z.i(j), wr.data(), wi.data());      // w = Z[:, 0:k+1] x
        double nr = 0;
        for (int i = 0; i < m; ++i) nr += wr[i] * wr[i] + wi[i] * wi[i];
        nr = 1.0 / std::sqrt(nr);
        for (int i = 0; i < m; ++i) vecs[i + (size_t)k * m] = zc(wr[i] * nr, wi[i] * nr);
    }
    });
    if (trace) {
        auto msf = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[feasthip eig] m=%d: Hessenberg %.3f, QR iteration %.3f, eigenvectors %.3f ms\n", m, msf(tp0, tp1), msf(tp1, tp2), msf(tp2, std::chrono::steady_clock::now()));
    }
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (vals[a].real() != vals[b].real()) return vals[a].real() < vals[b].real();
        return vals[a].imag() < vals[b].imag();
    });
    for (int k = 0; k < m; ++k) {
        lam[k] = vals[order[k]];
        for (int i = 0; i < m; ++i) V[i + (size_t)k * m] = vecs[i + (size_t)order[k] * m];
    }
    return 0;
}

int eig_generalized(int m, zc* A, zc* B, zc* lam, zc* V) {
    // C = B^-1 A: LU of B with partial pivoting (right-looking, column updates), then L and U solves column by column
    // of A in axpy form -- every inner loop is stride 1 on the split planes.
    int info = 0;
    Planes Bp(m), Ap(m);
    Bp.load(B);
    Ap.load(A);
    std::vector<int> piv(m);
    for (int k = 0; k < m; ++k) {
        double* bk_r = Bp.r(k);
        double* bk_i = Bp.i(k);
        int pr = k;
        double best = std::fabs(bk_r[k]) + std::fabs(bk_i[k]);
        for (int i = k + 1; i < m; ++i) { double a = std::fabs(bk_r[i]) + std::fabs(bk_i[i]); if (a > best) { best = a; pr = i; } }
        piv[k] = pr;
        if (best == 0.0) { info = 1; bk_r[pr] = 1e-300; }
        if (pr != k)
            for (int j = 0; j < m; ++j) { std::swap(Bp.r(j)[k], Bp.r(j)[pr]); std::swap(Bp.i(j)[k], Bp.i(j)[pr]); }
        zc inv = 1.0 / zc(bk_r[k], bk_i[k]);
        for (int i = k + 1; i < m; ++i) { zc l = zc(bk_r[i], bk_i[i]) * inv; bk_r[i] = l.real(); bk_i[i] = l.imag(); }
        for (int j = k + 1; j < m; ++j) caxpy_sub(k + 1, m, Bp.r(j)[k], Bp.i(j)[k], bk_r, bk_i, Bp.r(j), Bp.i(j));
    }
    const int nsp = (m >= 48 && pool_size() >= 2) ? std::min(8, pool_size() + 1) : 1;      // the columns of A are independent right-hand sides
    run_parts(nsp, [&](int part) {
    for (int j = (int)((long long)m * part / nsp); j < (int)((long long)m * (part + 1) / nsp); ++j) {
        double* xr = Ap.r(j);
        double* xi = Ap.i(j);
        for (int k = 0; k < m; ++k) if (piv[k] != k) { std::swap(xr[k], xr[piv[k]]); std::swap(xi[k], xi[piv[k]]); }
        for (int k = 0; k < m; ++k) caxpy_sub(k + 1, m, xr[k], xi[k], Bp.r(k), Bp.i(k), xr, xi);          // L y = P a
        for (int k = m - 1; k >= 0; --k) {                                                                 // U x = y
            zc x = zc(xr[k], xi[k]) / zc(Bp.r(k)[k], Bp.i(k)[k]);
            xr[k] = x.real(); xi[k] = x.imag();
            caxpy_sub(0, k, x.real(), x.imag(), Bp.r(k), Bp.i(k), xr, xi);
        }
    }
    });
    Ap.store(A);
    int e = eig_general(m, A, lam, V);
    return e ? e : info;
}

double hermitian_defect(int m, const zc* M) {
    double mx = 0, df = 0;
    for (int j = 0; j < m; ++j)
        for (int i = 0; i <= j; ++i) {
            mx = std::max(mx, std::abs(M[i + (size_t)j * m]));
            df = std::max(df, std::abs(M[i + (size_t)j * m] - std::conj(M[j + (size_t)i * m])));
        }
    return mx > 0 ? df / mx : 0.0;
}

namespace {
// M <- R^-H M, column by column (R upper triangular with real positive diagonal): forward substitution with R^H
void solve_rh_cols(int m, Planes& R, Planes& M) {
    for (int j = 0; j < m; ++j) {
        double* yr = M.r(j);
        double* yi = M.i(j);
        for (int i = 0; i < m; ++i) {
            const double* rr = R.r(i);
            const double* ri = R.i(i);
            double sr = 0, si = 0;
            for (int k = 0; k < i; ++k) {       // conj(R_ki) * y_k
                sr += rr[k] * yr[k] + ri[k] * yi[k];
                si += rr[k] * yi[k] - ri[k] * yr[k];
            }
            double d = rr[i];                    // diagonal of a Cholesky factor is real
            yr[i] = (yr[i] - sr) / d;
            yi[i] = (yi[i] - si) / d;
        }
    }
}

void conj_transpose(int m, Planes& M) {
    for (int j = 0; j < m; ++j) {
        M.im[j + (size_t)j * m] = -M.im[j + (size_t)j * m];
        for (int i = 0; i < j; ++i) {
            double ar = M.re[i + (size_t)j * m], ai = M.im[i + (size_t)j * m], br = M.re[j + (size_t)i * m], bi = M.im[j + (size_t)i * m];
            M.re[i + (size_t)j * m] = br; M.im[i + (size_t)j * m] = -bi;
            M.re[j + (size_t)i * m] = ar; M.im[j + (size_t)i * m] = -ai;
        }
    }
}
}  // namespace

int eig_hermitian(int m, zc* A, zc* B, zc* lam, zc* V) {
    if (m <= 0) return 0;
    Planes Ap(m), Rp(m), Qp(m);
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { Ap.re[i + (size_t)j * m] = A[i + (size_t)j * m].real(); Ap.im[i + (size_t)j * m] = A[i + (size_t)j * m].imag(); }
    if (B) {   // C = R^-H A R^-1 with B = R^H R
        std::vector<zc> R(B, B + (size_t)m * m);
        for (int j = 0; j < m; ++j) for (int i = 0; i < j; ++i) { zc a = 0.5 * (R[i + (size_t)j * m] + std::conj(R[j + (size_t)i * m])); R[i + (size_t)j * m] = a; }
        if (cholesky_upper(m, R.data())) return 1;
        for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { Rp.re[i + (size_t)j * m] = R[i + (size_t)j * m].real(); Rp.im[i + (size_t)j * m] = R[i + (size_t)j * m].imag(); }
        solve_rh_cols(m, Rp, Ap);          // T = R^-H A
        conj_transpose(m, Ap);             // T^H = A R^-1 ... (A Hermitian): C^H = R^-H T^H
        solve_rh_cols(m, Rp, Ap);
        conj_transpose(m, Ap);
    }
    // symmetrise
    for (int j = 0; j < m; ++j) {
        Ap.im[j + (size_t)j * m] = 0.0;
        for (int i = 0; i < j; ++i) {
            double ar = 0.5 * (Ap.re[i + (size_t)j * m] + Ap.re[j + (size_t)i * m]), ai = 0.5 * (Ap.im[i + (size_t)j * m] - Ap.im[j + (size_t)i * m]);
            Ap.re[i + (size_t)j * m] = ar; Ap.im[i + (size_t)j * m] = ai;
            Ap.re[j + (size_t)i * m] = ar; Ap.im[j + (size_t)i * m] = -ai;
        }
    }
    // Householder tridiagonalisation, Q accumulated
    for (int i = 0; i < m; ++i) Qp.re[i + (size_t)i * m] = 1.0;
    std::vector<double> vr(m), vi(m), wr(m), wi(m);
    for (int k = 0; k + 2 < m; ++k) {
        double* ckr = Ap.r(k);
        double* cki = Ap.i(k);
        double nrm = 0;
        for (int i = k + 1; i < m; ++i) nrm += ckr[i] * ckr[i] + cki[i] * cki[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        double x0r = ckr[k + 1], x0i = cki[k + 1], x0a = pythag(x0r, x0i);
        double phr = x0a == 0.0 ? 1.0 : x0r / x0a, phi = x0a == 0.0 ? 0.0 : x0i / x0a;
        double alr = -phr * nrm, ali = -phi * nrm;
        double vn = 0;
        for (int i = k + 1; i < m; ++i) { vr[i] = ckr[i]; vi[i] = cki[i]; }
        vr[k + 1] -= alr; vi[k + 1] -= ali;
        for (int i = k + 1; i < m; ++i) vn += vr[i] * vr[i] + vi[i] * vi[i];
        vn = std::sqrt(vn);
        if (vn == 0.0) continue;
        double ivn = 1.0 / vn;
        for (int i = k + 1; i < m; ++i) { vr[i] *= ivn; vi[i] *= ivn; }
        // w = A v on the trailing block (A Hermitian: row i = conj(column i)), K = v^H w, w <- 2 (w - K v)
        double Kr = 0, Ki = 0;
        for (int i = k + 1; i < m; ++i) {
            const double* cr = Ap.r(i);
            const double* ci = Ap.i(i);
            double sr = 0, si = 0;
            for (int j = k + 1; j < m; ++j) {   // conj(A_ji) * v_j
                sr += cr[j] * vr[j] + ci[j] * vi[j];
                si += cr[j] * vi[j] - ci[j] * vr[j];
            }
            wr[i] = sr; wi[i] = si;
            Kr += vr[i] * sr + vi[i] * si;      // conj(v_i) * w_i
            Ki += vr[i] * si - vi[i] * sr;
        }
        for (int i = k + 1; i < m; ++i) {
            double tr = wr[i] - (Kr * vr[i] - Ki * vi[i]), ti = wi[i] - (Kr * vi[i] + Ki * vr[i]);
            wr[i] = 2.0 * tr; wi[i] = 2.0 * ti;
        }
        // A -= v w^H + w v^H
        for (int j = k + 1; j < m; ++j) {
            double* cr = Ap.r(j);
            double* ci = Ap.i(j);
            double wjr = wr[j], wji = -wi[j], vjr = vr[j], vji = -vi[j];     // conj(w_j), conj(v_j)
            for (int i = k + 1; i < m; ++i) {
                cr[i] -= vr[i] * wjr - vi[i] * wji + wr[i] * vjr - wi[i] * vji;
                ci[i] -= vr[i] * wji + vi[i] * wjr + wr[i] * vji + wi[i] * vjr;
            }
        }
        ckr[k + 1] = alr; cki[k + 1] = ali;
        Ap.re[k + (size_t)(k + 1) * m] = alr; Ap.im[k + (size_t)(k + 1) * m] = -ali;
        for (int i = k + 2; i < m; ++i) { ckr[i] = 0.0; cki[i] = 0.0; Ap.re[k + (size_t)i * m] = 0.0; Ap.im[k + (size_t)i * m] = 0.0; }
        // Q <- Q H = Q - 2 (Q v) v^H
        std::fill(wr.begin(), wr.end(), 0.0);
        std::fill(wi.begin(), wi.end(), 0.0);
        for (int j = k + 1; j < m; ++j) {
            const double* qr = Qp.r(j);
            const double* qi = Qp.i(j);
            double vjr = vr[j], vji = vi[j];
            for (int i = 0; i < m; ++i) { wr[i] += qr[i] * vjr - qi[i] * vji; wi[i] += qr[i] * vji + qi[i] * vjr; }
        }
        for (int j = k + 1; j < m; ++j) {
            double* qr = Qp.r(j);
            double* qi = Qp.i(j);
            double cr2 = 2.0 * vr[j], ci2 = -2.0 * vi[j];       // 2 conj(v_j)
            for (int i = 0; i < m; ++i) { qr[i] -= wr[i] * cr2 - wi[i] * ci2; qi[i] -= wr[i] * ci2 + wi[i] * cr2; }
        }
    }
    // real tridiagonal: T = D^H A D, D = diag(phases)
    std::vector<double> d(m), e(m, 0.0), phr(m, 1.0), phi(m, 0.0);
    for (int i = 0; i < m; ++i) d[i] = Ap.re[i + (size_t)i * m];
    for (int i = 0; i + 1 < m; ++i) {
        double cr = Ap.re[i + 1 + (size_t)i * m], ci = Ap.im[i + 1 + (size_t)i * m], a = pythag(cr, ci);
        e[i + 1] = a;                                    // tql2 convention: e[i] couples i-1 and i
        if (a > 0) { phr[i + 1] = (phr[i] * cr - phi[i] * ci) / a; phi[i + 1] = (phr[i] * ci + phi[i] * cr) / a; }
        else { phr[i + 1] = phr[i]; phi[i + 1] = phi[i]; }
    }
    // implicit QL with accumulation (EISPACK tql2)
    std::vector<double> Z((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) Z[i + (size_t)i * m] = 1.0;
    for (int i = 1; i < m; ++i) e[i - 1] = e[i];
    e[m - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    for (int l = 0; l < m; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int mm = l;
        while (mm < m) { if (std::fabs(e[mm]) <= EPS * tst1) break; ++mm; }
        if (mm > l) {
            int iter = 0;
            do {
                if (++iter > 60) return 1;
                double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = pythag(p, 1.0);
                d[l] = e[l] / (p + (p < 0 ? -r : r));
                d[l + 1] = e[l] * (p + (p < 0 ? -r : r));
                double dl1 = d[l + 1], h = g - d[l];
                for (int i = l + 2; i < m; ++i) d[i] -= h;
                f += h;
                p = d[mm];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = mm - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i]; h = c * p;
                    r = pythag(p, e[i]);
                    e[i + 1] = s * r; s = e[i] / r; c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    double* z1 = Z.data() + (size_t)(i + 1) * m;
                    double* z0 = Z.data() + (size_t)i * m;
                    for (int kk = 0; kk < m; ++kk) {
                        double hh = z1[kk];
                        z1[kk] = s * z0[kk] + c * hh;
                        z0[kk] = c * z0[kk] - s * hh;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > EPS * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    // vectors: X = Q D Z, then back-transform with R^-1, normalise, sort ascending
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
    std::vector<double> xr(m), xi(m);
    for (int kk = 0; kk < m; ++kk) {
        int j = order[kk];
        const double* zj = Z.data() + (size_t)j * m;
        std::fill(xr.begin(), xr.end(), 0.0);
        std::fill(xi.begin(), xi.end(), 0.0);
        for (int q = 0; q < m; ++q) {
            double yr = phr[q] * zj[q], yi = phi[q] * zj[q];
            const double* qr = Qp.r(q);
            const double* qi = Qp.i(q);
            for (int i = 0; i < m; ++i) { xr[i] += qr[i] * yr - qi[i] * yi; xi[i] += qr[i] * yi + qi[i] * yr; }
        }
        if (B) {    // x <- R^-1 x, column-oriented back substitution
            for (int q = m - 1; q >= 0; --q) {
                const double* rr = Rp.r(q);
                const double* ri = Rp.i(q);
                double dq = rr[q];
                xr[q] /= dq; xi[q] /= dq;
                double tr = xr[q], ti = xi[q];
                for (int i = 0; i < q; ++i) { xr[i] -= rr[i] * tr - ri[i] * ti; xi[i] -= rr[i] * ti + ri[i] * tr; }
            }
        }
        double nr = 0;
        for (int i = 0; i < m; ++i) nr += xr[i] * xr[i] + xi[i] * xi[i];
        nr = 1.0 / std::sqrt(nr);
        lam[kk] = d[j];
        for (int i = 0; i < m; ++i) V[i + (size_t)kk * m] = zc(xr[i] * nr, xi[i] * nr);
    }
    return 0;
}

int cholesky_upper(int m, zc* G) {
    for (int j = 0; j < m; ++j) {
        double d = G[j + (size_t)j * m].real();
        for (int k = 0; k < j; ++k) d -= std::norm(G[k + (size_t)j * m]);
        if (!(d > 0.0)) return j + 1;
        d = std::sqrt(d);
        G[j + (size_t)j * m] = d;
        for (int i = j + 1; i < m; ++i) {
            zc s = G[j + (size_t)i * m];
            for (int k = 0; k < j; ++k) s -= std::conj(G[k + (size_t)j * m]) * G[k + (size_t)i * m];
            G[j + (size_t)i * m] = s / d;
        }
        for (int i = j + 1; i < m; ++i) G[i + (size_t)j * m] = 0.0;
    }
    return 0;
}

void trtri_upper(int m, zc* R) {
    for (int j = 0; j < m; ++j) {
        R[j + (size_t)j * m] = 1.0 / R[j + (size_t)j * m];
        for (int i = 0; i < j; ++i) {
            zc s = 0;
            for (int k = i; k < j; ++k) s += R[i + (size_t)k * m] * R[k + (size_t)j * m];
            R[i + (size_t)j * m] = -s * R[j + (size_t)j * m];
        }
    }
}

int jacobi_svd(int m, zc* M, zc* U, double* s, zc* V) {
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) V[i + (size_t)j * m] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < m - 1; ++p)
            for (int q = p + 1; q < m; ++q) {
                double a = 0, b = 0;
                zc g = 0;
                for (int i = 0; i < m; ++i) {
                    a += std::norm(M[i + (size_t)p * m]);
                    b += std::norm(M[i + (size_t)q * m]);
                    g += std::conj(M[i + (size_t)p * m]) * M[i + (size_t)q * m];
                }
                double ag = std::abs(g);
                if (ag <= 1e-15 * std::sqrt(a * b) || ag == 0.0) continue;
                off = std::max(off, ag / std::sqrt(a * b));
                zc ph = g / ag;
                double zeta = (b - a) / (2.0 * ag);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < m; ++i) {
                    zc mp = M[i + (size_t)p * m], mq = M[i + (size_t)q * m];
                    M[i + (size_t)p * m] = c * mp - sn * std::conj(ph) * mq;
                    M[i + (size_t)q * m] = sn * ph * mp + c * mq;
                    zc vp = V[i + (size_t)p * m], vq = V[i + (size_t)q * m];
                    V[i + (size_t)p * m] = c * vp - sn * std::conj(ph) * vq;
                    V[i + (size_t)q * m] = sn * ph * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    std::vector<int> order(m);
    std::vector<double> sv(m);
    for (int j = 0; j < m; ++j) { double a = 0; for (int i = 0; i < m; ++i) a += std::norm(M[i + (size_t)j * m]); sv[j] = std::sqrt(a); }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sv[a] > sv[b]; });
    std::vector<zc> Vt((size_t)m * m);
    for (int k = 0; k < m; ++k) {
        int j = order[k];
        s[k] = sv[j];
        for (int i = 0; i < m; ++i) {
            U[i + (size_t)k * m] = sv[j] > 0 ? M[i + (size_t)j * m] / sv[j] : zc(i == k ? 1.0 : 0.0);
            Vt[i + (size_t)k * m] = V[i + (size_t)j * m];
        }
    }
    std::copy(Vt.begin(), Vt.end(), V);
    return 0;
}

}  // namespace feasthip
