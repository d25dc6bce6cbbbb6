/*
 * gp_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C, fp64, single-threaded restatement of the arithmetic that the
 * reference (bbbales2/gp) executes on the GP marginal-likelihood hot path.
 * Nothing under gp_amd/ (the product) may import, link or call this file:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / reported baseline.
 *
 * Pinning status: the reference ships no golden vectors and R / Stan / Rcpp
 * cannot run in the build container, so the pieces that restate R and Stan
 * code are pinned by (a) closed-form known-answer tests, (b) LAPACK (scipy)
 * and mpmath cross-checks and (c) golden vectors emitted by importing the
 * reference's own gp_derivs.py (tests/golden/make_golden.py).  Pieces whose
 * only executable reference is R/Stan are "parity unpinned by the reference"
 * and say so next to the function.
 *
 * All matrices are column-major (R's native storage), leading dimension
 * explicit.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define A_(M, ld, i, j) ((M)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

/* ------------------------------------------------------------------ */
/* Squared-exponential kernels, matrix API  (R/kernels.R)             */
/* ------------------------------------------------------------------ */

/* QQard(X,Y,phi): phi1^2 * exp(-(1/2) * sum(((x-y)/phi2)^2))   R/kernels.R:19
 * with mat_to_obs_list / obs_list_outer (R/kernels.R:2-17) giving the
 * n x m "all row pairs" matrix.  phi2 is recycled by R: length 1 (isotropic)
 * or length D (ARD).  D = 1 reproduces QQ(x,y,phi) of R/kernels.R:22-24 up to
 * the last ulp (QQ divides by 2*phi2^2 instead; see orc_QQ). */
void orc_QQard(const double *X, int n, int ldx, const double *Y, int m, int ldy,
               int D, double alpha, const double *ell, int n_ell,
               double *K, int ldk)
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                double l = ell[n_ell == 1 ? 0 : d];
                double r = (A_(X, ldx, i, d) - A_(Y, ldy, j, d)) / l;
                s += r * r;
            }
            A_(K, ldk, i, j) = alpha * alpha * exp(-(1.0 / 2.0) * s);
        }
}

/* QQ(x,y,phi) = phi1^2*exp(-((x - y)^2/(2 * phi2^2)))           R/kernels.R:22-24 */
void orc_QQ(const double *x, int n, const double *y, int m, double alpha, double l,
            double *K, int ldk)
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            double r = x[i] - y[j];
            A_(K, ldk, i, j) = alpha * alpha * exp(-(r * r / (2 * l * l)));
        }
}

/* QR(x,y,phi) = phi1^2*(exp(..)*(x - y))/phi2^2                  R/kernels.R:26-28 */
void orc_QR(const double *x, int n, const double *y, int m, double alpha, double l,
            double *K, int ldk)
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            double r = x[i] - y[j];
            A_(K, ldk, i, j) = alpha * alpha * (exp(-(r * r / (2 * l * l))) * r) / (l * l);
        }
}

/* RR(x,y,phi)                                                    R/kernels.R:30-32
 * As written in the reference, operator precedence applies phi1^2 to the
 * first term only: phi1^2*e/phi2^2 - (e*(x-y)^2)/phi2^4.
 * compat != 0 reproduces that; compat == 0 is the mathematically intended
 * alpha^2 * e * (1/l^2 - r^2/l^4) (== derivative_kernels.R:51-53 times a^2 as
 * used at pendulum_fit.R:240).  Identical when alpha == 1. */
void orc_RR(const double *x, int n, const double *y, int m, double alpha, double l,
            int compat, double *K, int ldk)
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            double r = x[i] - y[j];
            double e = exp(-(r * r / (2 * l * l)));
            double a2 = alpha * alpha;
            if (compat)
                A_(K, ldk, i, j) = a2 * e / (l * l) - (e * r * r) / (l * l * l * l);
            else
                A_(K, ldk, i, j) = a2 * e / (l * l) - (a2 * e * r * r) / (l * l * l * l);
        }
}

/* ------------------------------------------------------------------ */
/* Elementwise derivative kernels  (derivative_kernels.R:39-73)       */
/* kind: 0 QQ 1 QR 2 RQ 3 RR 4 QT 5 TQ 6 RT 7 TR 8 TT                 */
/* ------------------------------------------------------------------ */
double orc_deriv_elem(int kind, double tj, double tk, double l)
{
    double r, e, l2 = l * l;
    switch (kind) {
    case 2: /* RQ(tj,tk) = QR(tk,tj)   :47-49 */
        return orc_deriv_elem(1, tk, tj, l);
    case 5: /* TQ(tj,tk) = QT(tk,tj)   :59-61 */
        return orc_deriv_elem(4, tk, tj, l);
    case 7: /* TR(tj,tk) = RT(tk,tj)   :67-69 */
        return orc_deriv_elem(6, tk, tj, l);
    default:
        break;
    }
    r = tj - tk;
    e = exp(-(r * r / (2 * l2)));
    switch (kind) {
    case 0: /* :39-41 */ return e;
    case 1: /* :43-45 */ return (e * r) / l2;
    case 3: /* :51-53 */ return e / l2 - (e * r * r) / (l2 * l2);
    case 4: /* :55-57 */ return -(e / l2) + (e * r * r) / (l2 * l2);
    case 6: /* :63-65 */ return (3 * e * r) / (l2 * l2) - (e * r * r * r) / (l2 * l2 * l2);
    case 8: /* :71-73 */
        return (3 * e) / (l2 * l2) - (6 * e * r * r) / (l2 * l2 * l2) +
               (e * r * r * r * r) / (l2 * l2 * l2 * l2);
    }
    return NAN;
}

void orc_deriv_vec(int kind, const double *tj, const double *tk, long len, double l, double *out)
{
    for (long i = 0; i < len; ++i) out[i] = orc_deriv_elem(kind, tj[i], tk[i], l);
}

/* a^2 * outer(x, y, FUN = kern(tj,tk,l))            pendulum_fit.R:237-240 */
void orc_deriv_cov(int kind, const double *x, int n, const double *y, int m,
                   double alpha, double l, double *K, int ldk)
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i)
            A_(K, ldk, i, j) = alpha * alpha * orc_deriv_elem(kind, x[i], y[j], l);
}

/* Joint [values; derivatives] covariance              R/ode_gp_library.R:29-30
 * K = rbind(cbind(UU + sigma^2 I, UD), cbind(t(UD), DD)) + jitter*I_{2N},
 * UU/UD/DD == QQ/QR/RR of R/kernels.R (the reference renamed them). */
void orc_joint_cov(const double *t, int n, double alpha, double l, double sigma,
                   double jitter, int compat, double *K, int ldk)
{
    double *B = (double *)malloc(sizeof(double) * (size_t)n * n);
    orc_QQ(t, n, t, n, alpha, l, B, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
            A_(K, ldk, i, j) = A_(B, n, i, j) + (i == j ? sigma * sigma : 0.0);
    orc_QR(t, n, t, n, alpha, l, B, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            A_(K, ldk, i, n + j) = A_(B, n, i, j); /* UD      */
            A_(K, ldk, n + j, i) = A_(B, n, i, j); /* t(UD)   */
        }
    orc_RR(t, n, t, n, alpha, l, compat, B, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) A_(K, ldk, n + i, n + j) = A_(B, n, i, j);
    for (int i = 0; i < 2 * n; ++i) A_(K, ldk, i, i) += jitter;
    free(B);
}

/* ------------------------------------------------------------------ */
/* Stan cov_exp_quad                       models/fit_hyperparameters.stan:19 */
/* Stan Math semantics (third-party, unpinned version; algorithm restated):   */
/* off-diagonals alpha^2 * exp(-0.5/rho^2 * ||xi-xj||^2) computed once per    */
/* pair and mirrored, diagonal exactly alpha^2.         */
/* ------------------------------------------------------------------ */
void orc_cov_exp_quad(const double *X, int n, int ldx, int D, double alpha, double rho,
                      double *K, int ldk)
{
    double a2 = alpha * alpha, nh = -0.5 / (rho * rho);
    for (int j = 0; j < n; ++j) {
        A_(K, ldk, j, j) = a2;
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                double r = A_(X, ldx, i, d) - A_(X, ldx, j, d);
                s += r * r;
            }
            double v = a2 * exp(s * nh);
            A_(K, ldk, i, j) = v;
            A_(K, ldk, j, i) = v;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Dense factorisations                                               */
/* ------------------------------------------------------------------ */

/* cholesky_decompose(Sigma)  models/fit_hyperparameters.stan:25, exact_gp.stan:23
 * Lower L with L L^T = A, in place; strict upper triangle zeroed (Stan returns
 * a full matrix with zeros above).  Unblocked column (left-looking dot
 * product) form == Eigen LLT's unblocked kernel.  Returns 0, or k>0 when the
 * leading minor of order k is not positive definite (base-R chol error /
 * Stan domain_error). */
int orc_cholesky(double *A, int n, int lda)
{
    for (int j = 0; j < n; ++j) {
        double d = A_(A, lda, j, j);
        for (int k = 0; k < j; ++k) d -= A_(A, lda, j, k) * A_(A, lda, j, k);
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        A_(A, lda, j, j) = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A_(A, lda, i, j);
            for (int k = 0; k < j; ++k) s -= A_(A, lda, i, k) * A_(A, lda, j, k);
            A_(A, lda, i, j) = s / d;
        }
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) A_(A, lda, i, j) = 0.0;
    return 0;
}

/* Cache-blocked right-looking variant of the same factorisation; used for the
 * timed CPU baseline at sizes where the unblocked loop is cache-starved.
 * Same arithmetic per element up to summation order. */
int orc_cholesky_blocked(double *A, int n, int lda, int nb)
{
    if (nb < 8) nb = 64;
    for (int k = 0; k < n; k += nb) {
        int kb = (n - k < nb) ? n - k : nb;
        /* diagonal block: unblocked */
        for (int j = k; j < k + kb; ++j) {
            double d = A_(A, lda, j, j);
            for (int p = k; p < j; ++p) d -= A_(A, lda, j, p) * A_(A, lda, j, p);
            if (!(d > 0.0)) return j + 1;
            d = sqrt(d);
            A_(A, lda, j, j) = d;
            for (int i = j + 1; i < k + kb; ++i) {
                double s = A_(A, lda, i, j);
                for (int p = k; p < j; ++p) s -= A_(A, lda, i, p) * A_(A, lda, j, p);
                A_(A, lda, i, j) = s / d;
            }
        }
        /* panel: A21 <- A21 * L11^-T */
        for (int j = k; j < k + kb; ++j) {
            double d = A_(A, lda, j, j);
            for (int p = k; p < j; ++p) {
                double ljp = A_(A, lda, j, p);
                double *cj = &A_(A, lda, 0, j), *cp = &A_(A, lda, 0, p);
                for (int i = k + kb; i < n; ++i) cj[i] -= cp[i] * ljp;
            }
            double *cj = &A_(A, lda, 0, j);
            for (int i = k + kb; i < n; ++i) cj[i] /= d;
        }
        /* trailing: A22 <- A22 - L21 L21^T (lower only) */
        for (int j = k + kb; j < n; ++j)
            for (int p = k; p < k + kb; ++p) {
                double ljp = A_(A, lda, j, p);
                double *cj = &A_(A, lda, 0, j), *cp = &A_(A, lda, 0, p);
                for (int i = j; i < n; ++i) cj[i] -= cp[i] * ljp;
            }
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) A_(A, lda, i, j) = 0.0;
    return 0;
}

/* mdivide_left_tri_low(L, b): forward substitution, in place.
 * Used by multi_normal_cholesky  (models/fit_hyperparameters.stan:31). */
void orc_trsv_lower(const double *L, int n, int ldl, double *b)
{
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A_(L, ldl, i, k) * b[k];
        b[i] = s / A_(L, ldl, i, i);
    }
}

/* back substitution with L^T, in place */
void orc_trsv_lower_t(const double *L, int n, int ldl, double *b)
{
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A_(L, ldl, k, i) * b[k];
        b[i] = s / A_(L, ldl, i, i);
    }
}

/* f = L * z     models/exact_gp.stan:25 */
void orc_trmv_lower(const double *L, int n, int ldl, const double *z, double *f)
{
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s += A_(L, ldl, i, k) * z[k];
        f[i] = s;
    }
}

/* base-R solve(A, B) == LAPACK dgesv: LU with partial pivoting, then two
 * triangular solves per right-hand side  (R/ode_gp.R:10-11,28-29;
 * pendulum_fit.R:244,250).  A is overwritten by its LU factors, B by X.
 * Returns 0 or k>0 for an exactly singular pivot. */
int orc_lu_solve(double *A, int n, int lda, double *B, int nrhs, int ldb)
{
    int *piv = (int *)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double mx = fabs(A_(A, lda, k, k));
        for (int i = k + 1; i < n; ++i)
            if (fabs(A_(A, lda, i, k)) > mx) { mx = fabs(A_(A, lda, i, k)); p = i; }
        piv[k] = p;
        if (mx == 0.0) { free(piv); return k + 1; }
        if (p != k)
            for (int j = 0; j < n; ++j) {
                double t = A_(A, lda, k, j); A_(A, lda, k, j) = A_(A, lda, p, j); A_(A, lda, p, j) = t;
            }
        double inv = 1.0 / A_(A, lda, k, k);
        for (int i = k + 1; i < n; ++i) A_(A, lda, i, k) *= inv;
        for (int j = k + 1; j < n; ++j) {
            double akj = A_(A, lda, k, j);
            double *cj = &A_(A, lda, 0, j), *ck = &A_(A, lda, 0, k);
            for (int i = k + 1; i < n; ++i) cj[i] -= ck[i] * akj;
        }
    }
    for (int c = 0; c < nrhs; ++c) {
        double *b = &A_(B, ldb, 0, c);
        for (int k = 0; k < n; ++k)
            if (piv[k] != k) { double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
        for (int i = 0; i < n; ++i) {
            double s = b[i];
            for (int k = 0; k < i; ++k) s -= A_(A, lda, i, k) * b[k];
            b[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < n; ++k) s -= A_(A, lda, i, k) * b[k];
            b[i] = s / A_(A, lda, i, i);
        }
    }
    free(piv);
    return 0;
}

/* C(n x m) = A(n x k) %*% B(k x m) */
static void matmul(const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                   int n, int k, int m)
{
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i < n; ++i) A_(C, ldc, i, j) = 0.0;
        for (int p = 0; p < k; ++p) {
            double b = A_(B, ldb, p, j);
            const double *ap = &A_(A, lda, 0, p);
            double *cj = &A_(C, ldc, 0, j);
            for (int i = 0; i < n; ++i) cj[i] += ap[i] * b;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Marginal likelihood        models/fit_hyperparameters.stan:18-32   */
/* ------------------------------------------------------------------ */

/* One evaluation of the model block with double inputs:
 *   Sigma = cov_exp_quad(x, alpha, rho); Sigma[k,k] += sigma^2 (+ jitter)  :19-24
 *   L = cholesky_decompose(Sigma)                                          :25
 *   y ~ multi_normal_cholesky(0, L)                                        :31
 * out[0] = log marginal likelihood = -1/2 z'z - sum log L_ii - N/2 log(2 pi)
 * out[1] = sum_i log L_ii      out[2] = z'z  with z = L^-1 y
 * Returns Cholesky info (0 ok).  X is n x D column-major. */
int orc_logml(const double *X, int n, int ldx, int D, const double *y, double alpha,
              double rho, double sigma, double jitter, double *out)
{
    double *K = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    orc_cov_exp_quad(X, n, ldx, D, alpha, rho, K, n);
    for (int i = 0; i < n; ++i) A_(K, n, i, i) += sigma * sigma + jitter;
    int info = (n > 512) ? orc_cholesky_blocked(K, n, n, 64) : orc_cholesky(K, n, n);
    if (info) { free(K); free(z); out[0] = out[1] = out[2] = NAN; return info; }
    memcpy(z, y, sizeof(double) * (size_t)n);
    orc_trsv_lower(K, n, n, z);
    double ld = 0.0, q = 0.0;
    for (int i = 0; i < n; ++i) { ld += log(A_(K, n, i, i)); q += z[i] * z[i]; }
    out[1] = ld;
    out[2] = q;
    out[0] = -0.5 * q - ld - 0.5 * n * log(2.0 * M_PI);
    free(K); free(z);
    return 0;
}

/* Gradient of the log marginal likelihood with respect to (alpha, rho, sigma) -- what Stan's
 * reverse-mode autodiff hands NUTS for models/fit_hyperparameters.stan:18-32:
 *   d logml / d theta = 1/2 tr((a a' - K^-1) dK/dtheta),  a = K^-1 y
 *   dK/dalpha = 2 Kse / alpha,  dK/drho = Kse .* d2 / rho^3,  dK/dsigma = 2 sigma I
 * (Kse = K without the diagonal term, d2 = squared distances).  Dense O(n^3) restatement:
 * K^-1 column by column through the Cholesky factor.  grad[0..2] = d/dalpha, d/drho, d/dsigma.
 * The reference has no test for it: pinned by finite differences of orc_logml (tests). */
int orc_logml_grad(const double *X, int n, int ldx, int D, const double *y, double alpha,
                   double rho, double sigma, double jitter, double *out, double *grad)
{
    size_t nn = (size_t)n * n;
    double *Kse = (double *)malloc(sizeof(double) * nn);
    double *L = (double *)malloc(sizeof(double) * nn);
    double *Ki = (double *)malloc(sizeof(double) * nn);
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    orc_cov_exp_quad(X, n, ldx, D, alpha, rho, Kse, n);
    memcpy(L, Kse, sizeof(double) * nn);
    for (int i = 0; i < n; ++i) A_(L, n, i, i) += sigma * sigma + jitter;
    int info = orc_cholesky(L, n, n);
    if (info) { free(Kse); free(L); free(Ki); free(a); out[0] = out[1] = out[2] = NAN; return info; }
    memcpy(a, y, sizeof(double) * (size_t)n);
    orc_trsv_lower(L, n, n, a);
    double ld = 0.0, q = 0.0;
    for (int i = 0; i < n; ++i) { ld += log(A_(L, n, i, i)); q += a[i] * a[i]; }
    out[1] = ld; out[2] = q;
    out[0] = -0.5 * q - ld - 0.5 * n * log(2.0 * M_PI);
    orc_trsv_lower_t(L, n, n, a);                       /* a = K^-1 y */
    for (int j = 0; j < n; ++j) {                       /* K^-1 e_j */
        double *col = Ki + (size_t)j * n;
        for (int i = 0; i < n; ++i) col[i] = (i == j) ? 1.0 : 0.0;
        orc_trsv_lower(L, n, n, col);
        orc_trsv_lower_t(L, n, n, col);
    }
    double ga = 0.0, gr = 0.0, gs = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double g = 0.5 * (a[i] * a[j] - A_(Ki, n, i, j));
            double d2 = 0.0;
            for (int d = 0; d < D; ++d) {
                double r = X[(size_t)i + (size_t)d * ldx] - X[(size_t)j + (size_t)d * ldx];
                d2 += r * r;
            }
            ga += g * A_(Kse, n, i, j);
            gr += g * A_(Kse, n, i, j) * d2;
            if (i == j) gs += g;
        }
    grad[0] = 2.0 * ga / alpha;
    grad[1] = gr / (rho * rho * rho);
    grad[2] = 2.0 * sigma * gs;
    free(Kse); free(L); free(Ki); free(a);
    return 0;
}

/* Stan's lp__ for fit_hyperparameters.stan (derived from
 * Stan semantics, cannot be executed here -> parity unpinned by the reference):
 *   -sum log L_ii - 1/2 z'z                      (y ~ multi_normal_cholesky, :31,
 *                                                  constant -N/2 log 2pi dropped by ~)
 *   + 3 log rho - 4 rho                          (rho ~ gamma(4,4), :27)
 *   - alpha^2/2 - sigma^2/2                      (half-normal(0,1), :28-29)
 *   + log rho + log alpha + log sigma            (Jacobian of <lower=0>, :13-15) */
double orc_stan_lp(double sum_log_diag, double quad, double alpha, double rho, double sigma)
{
    return -sum_log_diag - 0.5 * quad + (3.0 * log(rho) - 4.0 * rho) - 0.5 * alpha * alpha -
           0.5 * sigma * sigma + (log(rho) + log(alpha) + log(sigma));
}

/* ------------------------------------------------------------------ */
/* rbf_cov_chol(x1, l)                          covariance.cpp:9-47   */
/* Forward-mode AD through the factorisation: every scalar is a dual  */
/* (value, tangent) with dl = 1 seeded at :13.                        */
/* ------------------------------------------------------------------ */
int orc_rbf_cov_chol(const double *x, int n, double l, double *L, int ldl, double *dL, int lddl)
{
    /* Sigma(i,j) = exp(-(xi-xj)^2 / (2 l l)),  d/dl = Sigma * (xi-xj)^2 / l^3   :17-21 */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double r2 = (x[i] - x[j]) * (x[i] - x[j]);
            double v = exp(-r2 / (2 * l * l));
            A_(L, ldl, i, j) = v;
            A_(dL, lddl, i, j) = v * r2 / (l * l * l);
        }
    for (int i = 0; i < n; ++i) A_(L, ldl, i, i) += 1e-10; /* :23-25 */
    /* cholesky_decompose on duals (:29): same recurrences, product rule. */
    for (int j = 0; j < n; ++j) {
        double d = A_(L, ldl, j, j), dd = A_(dL, lddl, j, j);
        for (int k = 0; k < j; ++k) {
            d -= A_(L, ldl, j, k) * A_(L, ldl, j, k);
            dd -= 2.0 * A_(L, ldl, j, k) * A_(dL, lddl, j, k);
        }
        if (!(d > 0.0)) return j + 1;
        double s = sqrt(d), ds = dd / (2.0 * s);
        A_(L, ldl, j, j) = s;
        A_(dL, lddl, j, j) = ds;
        for (int i = j + 1; i < n; ++i) {
            double v = A_(L, ldl, i, j), dv = A_(dL, lddl, i, j);
            for (int k = 0; k < j; ++k) {
                v -= A_(L, ldl, i, k) * A_(L, ldl, j, k);
                dv -= A_(dL, lddl, i, k) * A_(L, ldl, j, k) + A_(L, ldl, i, k) * A_(dL, lddl, j, k);
            }
            double q = v / s;
            A_(L, ldl, i, j) = q;
            A_(dL, lddl, i, j) = (dv - q * ds) / s;
        }
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) { A_(L, ldl, i, j) = 0.0; A_(dL, lddl, i, j) = 0.0; }
    return 0;
}

/* approx_L(l, lp, Ls, dLdls): piecewise cubic Hermite interpolation of the
 * Cholesky factor over the length-scale            covariance.cpp:49-96
 * Ls / dLs are P stacked n x n column-major matrices.  (covariance.cpp:53
 * reads Ls[1] for sizing; n is passed explicitly here.) */
void orc_approx_L(double l, const double *lp, int P, const double *Ls, const double *dLs,
                  int n, double *out, int ldo)
{
    int lidx = 0;
    for (; lidx < P - 1; lidx++)
        if (lp[lidx + 1] >= l) break;
    if (lidx > P - 2) lidx = P - 2;
    double x1 = lp[lidx], x2 = lp[lidx + 1];
    double t = (l - x1) / (x2 - x1);
    const double *L1 = Ls + (size_t)lidx * n * n, *L2 = Ls + (size_t)(lidx + 1) * n * n;
    const double *D1 = dLs + (size_t)lidx * n * n, *D2 = dLs + (size_t)(lidx + 1) * n * n;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            if (j <= i) {
                double y1 = A_(L1, n, i, j), y2 = A_(L2, n, i, j);
                double k1 = A_(D1, n, i, j), k2 = A_(D2, n, i, j);
                double a = k1 * (x2 - x1) - (y2 - y1);
                double b = -k2 * (x2 - x1) + (y2 - y1);
                A_(out, ldo, i, j) = (1 - t) * y1 + t * y2 + t * (1 - t) * (a * (1 - t) + b * t);
            } else
                A_(out, ldo, i, j) = 0.0;
        }
}

/* approx_Lz(l, lp, Ls, dLdls, z) = v z with v the same Hermite blend formed as full matrices
 * (the upper triangles of the factors are zero)        models/cubic_interpolated_gp.hpp:38-73
 * (value only: what the `double` overload of build_output, :34-36, leaves; orc_approx_Lz_grad below
 * restates the `var` overload's tangent) */
void orc_approx_Lz(double l, const double *lp, int P, const double *Ls, const double *dLs,
                   int n, const double *z, double *f)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)n * n);
    orc_approx_L(l, lp, P, Ls, dLs, n, v, n);
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += A_(v, n, i, j) * z[j];
        f[i] = acc;
    }
    free(v);
}

/* approx_Lz under Stan's reverse mode: the `var` overload of build_output
 * (models/cubic_interpolated_gp.hpp:6-32) attaches to output i the partial dfdl(i) = (dvdl z)(i) with
 * respect to l (precomp_v_vari(0.0, l.vi_, dfdl(i))); the value stays v z (:72).  Only the `double`
 * overload (:34-36) returns zeros.  dvdl as written at :67, dtdl = 1 / (x2 - x1) at :53; the
 * matrices are formed in full (the factors' upper triangles are zero), as :59-67 do. */
void orc_approx_Lz_grad(double l, const double *lp, int P, const double *Ls, const double *dLs,
                        int n, const double *z, double *f, double *dfdl)
{
    int lidx = 0;
    for (; lidx < P - 1; lidx++)
        if (lp[lidx + 1] >= l) break;
    if (lidx > P - 2) lidx = P - 2;
    double x1 = lp[lidx], x2 = lp[lidx + 1];
    double t = (l - x1) / (x2 - x1);
    double dtdl = 1 / (x2 - x1);
    const double *Y1 = Ls + (size_t)lidx * n * n, *Y2 = Ls + (size_t)(lidx + 1) * n * n;
    const double *K1 = dLs + (size_t)lidx * n * n, *K2 = dLs + (size_t)(lidx + 1) * n * n;
    for (int i = 0; i < n; ++i) {
        double accv = 0.0, accd = 0.0;
        for (int j = 0; j < n; ++j) {
            double y1 = A_(Y1, n, i, j), y2 = A_(Y2, n, i, j);
            double k1 = A_(K1, n, i, j), k2 = A_(K2, n, i, j);
            double a = k1 * (x2 - x1) - (y2 - y1);
            double b = -k2 * (x2 - x1) + (y2 - y1);
            double v = (1 - t) * y1 + t * y2 + t * (1 - t) * (a * (1 - t) + b * t);
            double dvdl = (b * (2 - 3 * t) * t + a * (1 + t * (-4 + 3 * t)) - y1 + y2) * dtdl;
            accv += v * z[j];
            accd += dvdl * z[j];
        }
        f[i] = accv;
        dfdl[i] = accd;
    }
}

/* ------------------------------------------------------------------ */
/* GP posterior of the state and of its time derivative               */
/* ------------------------------------------------------------------ */

/* Generic form shared by p_Xn (R/ode_gp.R:1-14), p_dotXn (R/ode_gp.R:19-32)
 * and sample_derivs' build_mu/build_cov (pendulum_fit.R:242-251):
 *   mn = Ks %*% solve(K + s2 I, y)
 *   Kn = Kss - Ks %*% solve(K + s2 I, t(Ks)) + jitter I
 * with base-R solve() == LU (dgesv), factorised twice in the reference
 * (same factors; done once here).  K: n x n, Ks: m x n, Kss: m x m. */
int orc_gp_condition(const double *K, int n, const double *Ks, int m, const double *Kss,
                     const double *y, double s2, double jitter, double *mn, double *Kn)
{
    size_t nn = (size_t)n * n;
    double *A = (double *)malloc(sizeof(double) * nn);
    double *B = (double *)malloc(sizeof(double) * (size_t)n * (m + 1));
    memcpy(A, K, sizeof(double) * nn);
    for (int i = 0; i < n; ++i) A_(A, n, i, i) += s2;
    /* B = [ t(Ks) | y ] */
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) A_(B, n, i, j) = A_(Ks, m, j, i);
    memcpy(B + (size_t)n * m, y, sizeof(double) * (size_t)n);
    int info = orc_lu_solve(A, n, n, B, m + 1, n);
    if (!info) {
        matmul(Ks, m, B + (size_t)n * m, n, mn, m, m, n, 1);
        double *T = (double *)malloc(sizeof(double) * (size_t)m * m);
        matmul(Ks, m, B, n, T, m, m, n, m);
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i)
                A_(Kn, m, i, j) = A_(Kss, m, i, j) - A_(T, m, i, j) + (i == j ? jitter : 0.0);
        free(T);
    }
    free(A); free(B);
    return info;
}

/* p_Xn(tn, Xn, phi_n, sigma_n)                          R/ode_gp.R:1-14 */
int orc_p_Xn(const double *tn, const double *Xn, int n, double alpha, double l, double sigma,
             double *mn, double *Kn)
{
    double *K = (double *)malloc(sizeof(double) * (size_t)n * n);
    orc_QQ(tn, n, tn, n, alpha, l, K, n);
    int info = orc_gp_condition(K, n, K, n, K, Xn, sigma * sigma, 0.0, mn, Kn);
    free(K);
    return info;
}

/* p_dotXn(tn, Xn, phi_n, sigma_n)                       R/ode_gp.R:19-32
 * mn = RQ (QQ + s^2 I)^-1 Xn ; Kn = RR - RQ (QQ + s^2 I)^-1 QR ; RQ = t(QR). */
int orc_p_dotXn(const double *tn, const double *Xn, int n, double alpha, double l, double sigma,
                int compat, double *mn, double *Kn)
{
    size_t nn = (size_t)n * n;
    double *QQm = (double *)malloc(sizeof(double) * nn);
    double *QRm = (double *)malloc(sizeof(double) * nn);
    double *RQm = (double *)malloc(sizeof(double) * nn);
    double *RRm = (double *)malloc(sizeof(double) * nn);
    orc_QQ(tn, n, tn, n, alpha, l, QQm, n);
    orc_QR(tn, n, tn, n, alpha, l, QRm, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) A_(RQm, n, i, j) = A_(QRm, n, j, i);
    orc_RR(tn, n, tn, n, alpha, l, compat, RRm, n);
    int info = orc_gp_condition(QQm, n, RQm, n, RRm, Xn, sigma * sigma, 0.0, mn, Kn);
    free(QQm); free(QRm); free(RQm); free(RRm);
    return info;
}

/* Joint-covariance form                          R/ode_gp_library.R:23-33
 * K2n = [[UU+s^2 I, UD],[t(UD), DD]] + 1e-6 I; condMVN conditions block 2 on
 * block 1 (condMVNorm::condMVN: condMean = S21 S11^-1 x, condVar = S22 -
 * S21 S11^-1 S12 with solve() on the given block).  Third-party, unpinned. */
int orc_p_dotXn_joint(const double *tn, const double *Xn, int n, double alpha, double l,
                      double sigma, double jitter, int compat, double *condMean, double *condVar)
{
    int n2 = 2 * n;
    double *K = (double *)malloc(sizeof(double) * (size_t)n2 * n2);
    orc_joint_cov(tn, n, alpha, l, sigma, jitter, compat, K, n2);
    double *S11 = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *S21 = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *S22 = (double *)malloc(sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            A_(S11, n, i, j) = A_(K, n2, i, j);
            A_(S21, n, i, j) = A_(K, n2, n + i, j);
            A_(S22, n, i, j) = A_(K, n2, n + i, n + j);
        }
    int info = orc_gp_condition(S11, n, S21, n, S22, Xn, 0.0, 0.0, condMean, condVar);
    free(K); free(S11); free(S21); free(S22);
    return info;
}

/* sample_derivs(params=(l,a,sy), ynoise, ti): moments only  pendulum_fit.R:227-255
 * K = a^2 QQ, KsK = a^2 RQ, KsKs = a^2 RR (derivative_kernels.R elementwise
 * functions through outer()), cov jitter 1e-8 (:250).  The single
 * MASS::mvrnorm draw (:253) is stochastic and unseeded in the reference. */
int orc_sample_derivs_moments(const double *ti, const double *ynoise, int n, double l, double a,
                              double sy, double jitter, double *mu, double *cov)
{
    size_t nn = (size_t)n * n;
    double *K = (double *)malloc(sizeof(double) * nn);
    double *KsK = (double *)malloc(sizeof(double) * nn);
    double *KsKs = (double *)malloc(sizeof(double) * nn);
    orc_deriv_cov(0, ti, n, ti, n, a, l, K, n);
    orc_deriv_cov(2, ti, n, ti, n, a, l, KsK, n);
    orc_deriv_cov(3, ti, n, ti, n, a, l, KsKs, n);
    int info = orc_gp_condition(K, n, KsK, n, KsKs, ynoise, sy * sy, jitter, mu, cov);
    free(K); free(KsK); free(KsKs);
    return info;
}

/* Log marginal likelihood of stacked observations [y; y'] under the joint
 * covariance of R/ode_gp_library.R:29-30 (BASELINE config c5).  out as orc_logml. */
int orc_joint_logml(const double *t, int n, const double *yy, double alpha, double l,
                    double sigma, double jitter, double *out)
{
    int n2 = 2 * n;
    double *K = (double *)malloc(sizeof(double) * (size_t)n2 * n2);
    double *z = (double *)malloc(sizeof(double) * (size_t)n2);
    orc_joint_cov(t, n, alpha, l, sigma, jitter, 0, K, n2);
    int info = (n2 > 512) ? orc_cholesky_blocked(K, n2, n2, 64) : orc_cholesky(K, n2, n2);
    if (info) { free(K); free(z); out[0] = out[1] = out[2] = NAN; return info; }
    memcpy(z, yy, sizeof(double) * (size_t)n2);
    orc_trsv_lower(K, n2, n2, z);
    double ld = 0.0, q = 0.0;
    for (int i = 0; i < n2; ++i) { ld += log(A_(K, n2, i, i)); q += z[i] * z[i]; }
    out[1] = ld; out[2] = q;
    out[0] = -0.5 * q - ld - 0.5 * n2 * log(2.0 * M_PI);
    free(K); free(z);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Deterministic synthetic inputs: counter-based  */
/* SplitMix64 so CPU and GPU sides can regenerate identical bits.     */
/* ------------------------------------------------------------------ */
static unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static double u01(unsigned long long seed, unsigned long long idx)
{
    return (double)(splitmix64(seed * 0x100000001B3ULL + idx) >> 11) * (1.0 / 9007199254740992.0);
}
/* X: n x D column-major U[0,1); y = sin(2 pi sum_d X) + 0.1 eps (Box-Muller) */
void orc_synth(int n, int D, unsigned long long seed, double *X, double *y)
{
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < n; ++i) X[(size_t)i + (size_t)d * n] = u01(seed, (unsigned long long)i + (unsigned long long)n * d);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += X[(size_t)i + (size_t)d * n];
        double u1 = u01(seed + 1, 2ULL * i), u2 = u01(seed + 1, 2ULL * i + 1);
        if (u1 < 1e-300) u1 = 1e-300;
        double eps = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
        y[i] = sin(2.0 * M_PI * s) + 0.1 * eps;
    }
}
