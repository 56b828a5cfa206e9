// Small dense complex kernels for the m0 x m0 reduced problems (host side).
// They replace Julia's eigen!(Aq), eigen!(Aq,Bq) (src/feast.jl:45,122 ->
// LAPACK zgeevx/zggev), svd!(Q0) on the R factor (src/utils.jl:70) and the
// Cholesky used by the tall-skinny orthonormalisation.  O(m0^3), m0 <= 256.
#pragma once
#include <complex>
#include <vector>

namespace feasthip {
typedef std::complex<double> zc;

// Eigen-decomposition of a general complex matrix (column-major m x m, destroyed).
// lam sorted by (re, im); V column-major unit-2-norm right eigenvectors.  0 on success.
int eig_general(int m, zc* A, zc* lam, zc* V);
// Generalised problem A x = lam B x via B^-1 A (B destroyed).  0 on success, 1 if B singular.
int eig_generalized(int m, zc* A, zc* B, zc* lam, zc* V);
// Hermitian(-definite) fast path: A Hermitian, B Hermitian positive definite (or NULL).  Householder
// tridiagonalisation + implicit QL.  Same outputs as eig_generalized (lam real, sorted ascending, unit
// 2-norm vectors).  Returns 0, or 1 if B is not positive definite / QL did not converge.
int eig_hermitian(int m, zc* A, zc* B, zc* lam, zc* V);
// max |M - M^H| / max |M|
double hermitian_defect(int m, const zc* M);
// Upper Cholesky factor: G = R^H R, R overwrites the upper triangle of G (lower zeroed).
// Returns 0 or the 1-based index of the first non-positive pivot.
int cholesky_upper(int m, zc* G);
// In-place inverse of an upper-triangular matrix.
void trtri_upper(int m, zc* R);
// One-sided Jacobi SVD of a square matrix M = U diag(s) V^H (column-major; M destroyed),
// singular values sorted descending.
int jacobi_svd(int m, zc* M, zc* U, double* s, zc* V);
// C = A * B (all m x m column-major)
void matmul(int m, const zc* A, const zc* B, zc* C);
}  // namespace feasthip
