/*
 * gpmi_shim.c -- R `.Call()` shim over the libgpmi C ABI (include/gpmi.h).
 *
 * Reference-side binding a maintainer adds to bbbales2/gp:
 *     R CMD SHLIB -o gpmi_shim.so gpmi_shim.c -I<repo>/include -L<repo>/gp_amd/csrc -lgpmi
 *     dyn.load("gpmi_shim.so"); source("r/gpmi.R")
 * R is absent from the build image (no Rinternals.h), so this file is NOT compiled by
 * __graft_entry__.build(); it is a thin translation layer whose every code path is one
 * C-ABI call that tests/ exercise through ctypes.  tests/test_abi.py runs `gcc -fsyntax-only` over it
 * against tests/r_api/ (declarations of the R API it uses: a syntax / type check, nothing more).
 * Rules it follows:
 *   - every vector's type and length is checked against n, m, D, G, B BEFORE the ABI call (need());
 *   - arguments of .Call are shared, never modified (copy-on-modify semantics);
 *   - outputs are allocated with Rf_allocMatrix / Rf_allocVector under PROTECT;
 *   - the ABI returns a status; all C resources are released BEFORE Rf_error (a longjmp);
 *   - one lazily created context per process.  HIP does not survive fork() and cannot be
 *     re-initialised in the child: parallel::mclapply children of an R session that already
 *     used the GPU get GPMI_EFORK from every call (gpmi_create included) and the shim reports
 *     it as an R error -- use mc.cores = 1, PSOCK workers, or the grid entry point.
 */
#include <string.h>
#include <R.h>
#include <Rinternals.h>
#include <unistd.h>

#include "gpmi.h"

static gpmi_ctx *g_ctx = NULL;

static gpmi_ctx *ctx(void)
{
    /* in a forked child the parent's handle answers GPMI_EFORK by itself; it is not replaced */
    if (!g_ctx) {
        int rc = gpmi_create(&g_ctx, 0);
        if (rc) Rf_error("libgpmi: %s", gpmi_last_error());
    }
    return g_ctx;
}

/* argument validation BEFORE any pointer reaches the C ABI: a length-1 R vector where the ABI reads G doubles is
 * an out-of-bounds heap read, not an error message (nothing is allocated yet, so Rf_error may longjmp freely) */
static void need(int ok, const char *what)
{
    if (!ok) Rf_error("libgpmi shim: %s", what);
}
static int is_real(SEXP x) { return TYPEOF(x) == REALSXP; }

static void check(int rc)
{
    if (rc > 0) Rf_error("the leading minor of order %d is not positive definite", rc); /* base-R chol() wording */
    if (rc < 0) Rf_error("libgpmi: %s", gpmi_last_error());
}

/* QQard(X, Y, phi): R/kernels.R:11-19.  X n x D, Y m x D, alpha scalar, ell length 1 or D */
SEXP gpmi_R_se_cov(SEXP X, SEXP Y, SEXP alpha, SEXP ell)
{
    int n = Rf_nrows(X), D = Rf_ncols(X), m = Rf_nrows(Y);
    need(is_real(X) && is_real(Y) && is_real(ell), "X, Y and the length-scales must be double");
    need(Rf_ncols(Y) == D, "X and Y must have the same number of columns");
    need(Rf_length(ell) == 1 || Rf_length(ell) == D, "length-scale must have length 1 or ncol(X)");
    SEXP K = PROTECT(Rf_allocMatrix(REALSXP, n, m));
    int rc = gpmi_se_cov(ctx(), REAL(X), n, n, REAL(Y), m, m, D, Rf_asReal(alpha), REAL(ell), Rf_length(ell),
                         0.0, GPMI_FULL, REAL(K), n > 0 ? n : 1);
    UNPROTECT(1);
    check(rc);
    return K;
}

/* QQ / QR / RR (x, y, phi) of R/kernels.R:22-32 and a^2 * outer(x, y, kern) of pendulum_fit.R:237-240 */
SEXP gpmi_R_deriv_cov(SEXP kind, SEXP x, SEXP y, SEXP alpha, SEXP l, SEXP flags)
{
    int n = Rf_length(x), m = Rf_length(y);
    need(is_real(x) && is_real(y), "x and y must be double");
    SEXP K = PROTECT(Rf_allocMatrix(REALSXP, n, m));
    int rc = gpmi_deriv_cov(ctx(), Rf_asInteger(kind), REAL(x), n, REAL(y), m, Rf_asReal(alpha), Rf_asReal(l),
                            Rf_asInteger(flags), REAL(K), n > 0 ? n : 1);
    UNPROTECT(1);
    check(rc);
    return K;
}

/* QQ..TT(tj, tk, l) of derivative_kernels.R:39-73 (vectors already recycled to equal length by the R wrapper) */
SEXP gpmi_R_deriv_elem(SEXP kind, SEXP tj, SEXP tk, SEXP l)
{
    R_xlen_t len = Rf_xlength(tj);
    need(is_real(tj) && is_real(tk) && Rf_xlength(tk) == len, "tj and tk must be double vectors of equal length (recycle in R)");
    SEXP out = PROTECT(Rf_allocVector(REALSXP, len));
    int rc = gpmi_deriv_elem(ctx(), Rf_asInteger(kind), REAL(tj), REAL(tk), (size_t)len, Rf_asReal(l), REAL(out));
    UNPROTECT(1);
    check(rc);
    return out;
}

/* list(L=, dLdl=) = rbf_cov_chol(x1, l_): covariance.cpp:9-47 */
SEXP gpmi_R_rbf_cov_chol(SEXP x, SEXP l)
{
    int n = Rf_length(x);
    need(is_real(x), "x must be double");
    SEXP L = PROTECT(Rf_allocMatrix(REALSXP, n, n));
    SEXP dL = PROTECT(Rf_allocMatrix(REALSXP, n, n));
    int rc = gpmi_rbf_cov_chol(ctx(), REAL(x), n, Rf_asReal(l), REAL(L), n, REAL(dL), n);
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2)), names = PROTECT(Rf_allocVector(STRSXP, 2));
    SET_VECTOR_ELT(out, 0, L); SET_VECTOR_ELT(out, 1, dL);
    SET_STRING_ELT(names, 0, Rf_mkChar("L")); SET_STRING_ELT(names, 1, Rf_mkChar("dLdl"));
    Rf_setAttrib(out, R_NamesSymbol, names);
    UNPROTECT(4);
    check(rc);
    return out;
}

/* interp_build(x, lp): the device-resident table of L(lp[p]), dL/dl(lp[p]) that
 * test_interpolate.R:9-19 builds with P calls of rbf_cov_chol */
SEXP gpmi_R_interp_build(SEXP x, SEXP lp)
{
    need(is_real(x) && is_real(lp), "x and lp must be double");
    check(gpmi_interp_build(ctx(), REAL(x), Rf_length(x), REAL(lp), Rf_length(lp)));
    return R_NilValue;
}

/* approx_L(l, lp, Ls, dLdls): covariance.cpp:49-96.  Ls, dLdls are R lists of n x n matrices; only
 * the two table entries around l travel to the device. */
static int interval(double l, const double *lp, int P)
{
    int k = 0;
    for (; k < P - 1; ++k)
        if (lp[k + 1] >= l) break;
    return k > P - 2 ? P - 2 : k;
}
static int load_pair(double l, SEXP lp, SEXP Ls, SEXP dLdls, int *n_out)
{
    int P = Rf_length(lp);
    need(is_real(lp) && P >= 2, "lp must be a double vector of at least two length-scales");
    need(TYPEOF(Ls) == VECSXP && TYPEOF(dLdls) == VECSXP && Rf_length(Ls) == P && Rf_length(dLdls) == P,
         "Ls and dLdls must be lists with one matrix per entry of lp");
    int k = interval(l, REAL(lp), P), n = Rf_nrows(VECTOR_ELT(Ls, k));
    for (int q = k; q <= k + 1; ++q) {   /* the two table entries that travel: n x n doubles each */
        SEXP a = VECTOR_ELT(Ls, q), b = VECTOR_ELT(dLdls, q);
        need(is_real(a) && is_real(b) && Rf_nrows(a) == n && Rf_ncols(a) == n && Rf_nrows(b) == n && Rf_ncols(b) == n,
             "the entries of Ls and dLdls must be square double matrices of one order");
    }
    double *buf = (double *)R_alloc((size_t)4 * n * n, sizeof(double));  /* freed by R, also on error */
    size_t m = (size_t)n * n;
    memcpy(buf, REAL(VECTOR_ELT(Ls, k)), m * sizeof(double));
    memcpy(buf + m, REAL(VECTOR_ELT(Ls, k + 1)), m * sizeof(double));
    memcpy(buf + 2 * m, REAL(VECTOR_ELT(dLdls, k)), m * sizeof(double));
    memcpy(buf + 3 * m, REAL(VECTOR_ELT(dLdls, k + 1)), m * sizeof(double));
    double lp2[2] = {REAL(lp)[k], REAL(lp)[k + 1]};
    *n_out = n;
    return gpmi_interp_load(ctx(), lp2, 2, buf, buf + 2 * m, n, n);
}
SEXP gpmi_R_approx_L(SEXP l, SEXP lp, SEXP Ls, SEXP dLdls)
{
    int n = 0;
    check(load_pair(Rf_asReal(l), lp, Ls, dLdls, &n));
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, n));
    int rc = gpmi_approx_L(ctx(), Rf_asReal(l), REAL(out), n);
    UNPROTECT(1);
    check(rc);
    return out;
}

/* approx_Lz(l, lp, Ls, dLdls, z): models/cubic_interpolated_gp.hpp:38-73; with Ls = NULL the table of
 * gpmi_R_interp_build is used (one call per leapfrog step, nothing but z crosses PCIe) */
SEXP gpmi_R_approx_Lz(SEXP l, SEXP lp, SEXP Ls, SEXP dLdls, SEXP z)
{
    int n = Rf_length(z), nz = n;
    need(is_real(z), "z must be double");
    if (!Rf_isNull(Ls)) check(load_pair(Rf_asReal(l), lp, Ls, dLdls, &n));
    need(n == nz, "z must have the order of the table's matrices");   /* (a device-resident table checks z in gpmi.R) */
    SEXP f = PROTECT(Rf_allocVector(REALSXP, n));
    int rc = gpmi_approx_Lz(ctx(), Rf_asReal(l), REAL(z), REAL(f));
    UNPROTECT(1);
    check(rc);
    return f;
}

/* list(f = approx_L(l) z, dfdl = (dv/dl) z): value and the reverse-mode partial of the Stan external function
 * (`var` overload of build_output, models/cubic_interpolated_gp.hpp:6-32, dvdl :67) */
SEXP gpmi_R_approx_Lz_grad(SEXP l, SEXP lp, SEXP Ls, SEXP dLdls, SEXP z)
{
    int n = Rf_length(z), nz = n;
    need(is_real(z), "z must be double");
    if (!Rf_isNull(Ls)) check(load_pair(Rf_asReal(l), lp, Ls, dLdls, &n));
    need(n == nz, "z must have the order of the table's matrices");
    SEXP f = PROTECT(Rf_allocVector(REALSXP, n)), g = PROTECT(Rf_allocVector(REALSXP, n));
    int rc = gpmi_approx_Lz_grad(ctx(), Rf_asReal(l), REAL(z), REAL(f), REAL(g));
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2)), names = PROTECT(Rf_allocVector(STRSXP, 2));
    SET_VECTOR_ELT(out, 0, f); SET_VECTOR_ELT(out, 1, g);
    SET_STRING_ELT(names, 0, Rf_mkChar("f")); SET_STRING_ELT(names, 1, Rf_mkChar("dfdl"));
    Rf_setAttrib(out, R_NamesSymbol, names);
    UNPROTECT(4);
    check(rc);
    return out;
}

/* c(logml, sum log L_ii, z'z): one evaluation of models/fit_hyperparameters.stan:18-32 */
SEXP gpmi_R_logml(SEXP X, SEXP y, SEXP alpha, SEXP ell, SEXP sigma, SEXP jitter)
{
    int n = Rf_nrows(X), D = Rf_ncols(X);
    need(is_real(X) && is_real(y) && is_real(ell), "X, y and the length-scales must be double");
    need(Rf_length(y) == n, "length(y) must equal nrow(X)");
    need(Rf_length(ell) == 1 || Rf_length(ell) == D, "length-scale must have length 1 or ncol(X)");
    SEXP out = PROTECT(Rf_allocVector(REALSXP, 3));
    int rc = gpmi_logml(ctx(), REAL(X), n, n, D, REAL(y), Rf_asReal(alpha), REAL(ell), Rf_length(ell),
                        Rf_asReal(sigma), Rf_asReal(jitter), REAL(out));
    UNPROTECT(1);
    check(rc);
    return out;
}

/* f = cholesky_decompose(cov_exp_quad(X, alpha, ell) + jitter I) z: the transform of models/exact_gp.stan:17-25 */
SEXP gpmi_R_exact_gp_f(SEXP X, SEXP alpha, SEXP ell, SEXP jitter, SEXP z)
{
    int n = Rf_nrows(X), D = Rf_ncols(X);
    need(is_real(X) && is_real(z) && is_real(ell), "X, z and the length-scales must be double");
    need(Rf_length(z) == n, "length(z) must equal nrow(X)");
    need(Rf_length(ell) == 1 || Rf_length(ell) == D, "length-scale must have length 1 or ncol(X)");
    SEXP out = PROTECT(Rf_allocVector(REALSXP, n));
    int rc = gpmi_exact_gp_f(ctx(), REAL(X), n, n, D, Rf_asReal(alpha), REAL(ell), Rf_length(ell), Rf_asReal(jitter), REAL(z),
                             REAL(out));
    UNPROTECT(1);
    check(rc);
    return out;
}

/* list(value = c(logml, sum log L_ii, z'z), grad = c(d/dalpha, d/dell..., d/dsigma)): what Stan's
 * autodiff computes per leapfrog step for models/fit_hyperparameters.stan:18-32 */
SEXP gpmi_R_logml_grad(SEXP X, SEXP y, SEXP alpha, SEXP ell, SEXP sigma, SEXP jitter)
{
    int n = Rf_nrows(X), D = Rf_ncols(X), ne = Rf_length(ell);
    need(is_real(X) && is_real(y) && is_real(ell), "X, y and the length-scales must be double");
    need(Rf_length(y) == n, "length(y) must equal nrow(X)");
    need(ne == 1 || ne == D, "length-scale must have length 1 or ncol(X)");
    SEXP val = PROTECT(Rf_allocVector(REALSXP, 3)), g = PROTECT(Rf_allocVector(REALSXP, 2 + ne));
    int rc = gpmi_logml_grad(ctx(), REAL(X), n, n, D, REAL(y), Rf_asReal(alpha), REAL(ell), ne,
                             Rf_asReal(sigma), Rf_asReal(jitter), REAL(val), REAL(g));
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2)), names = PROTECT(Rf_allocVector(STRSXP, 2));
    SET_VECTOR_ELT(out, 0, val); SET_VECTOR_ELT(out, 1, g);
    SET_STRING_ELT(names, 0, Rf_mkChar("value")); SET_STRING_ELT(names, 1, Rf_mkChar("grad"));
    Rf_setAttrib(out, R_NamesSymbol, names);
    UNPROTECT(4);
    check(rc);
    return out;
}

/* value + gradient at G points (one per chain): list(value = 3 x G, grad = 3 x G (d/dalpha, d/drho, d/dsigma), info) */
SEXP gpmi_R_logml_grad_grid(SEXP X, SEXP y, SEXP alpha, SEXP rho, SEXP sigma, SEXP jitter)
{
    int n = Rf_nrows(X), D = Rf_ncols(X), G = Rf_length(rho);
    need(is_real(X) && is_real(y) && is_real(alpha) && is_real(rho) && is_real(sigma), "X, y, alpha, rho, sigma must be double");
    need(Rf_length(y) == n, "length(y) must equal nrow(X)");
    need(Rf_length(alpha) == G && Rf_length(sigma) == G, "alpha, rho and sigma must have one entry per grid point (recycle in R)");
    SEXP val = PROTECT(Rf_allocMatrix(REALSXP, 3, G)), g = PROTECT(Rf_allocMatrix(REALSXP, 3, G)), info = PROTECT(Rf_allocVector(INTSXP, G));
    int rc = gpmi_logml_grad_grid(ctx(), REAL(X), n, n, D, REAL(y), REAL(alpha), REAL(rho), REAL(sigma), G, Rf_asReal(jitter),
                                  REAL(val), REAL(g), INTEGER(info));
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 3));
    SET_VECTOR_ELT(out, 0, val); SET_VECTOR_ELT(out, 1, g); SET_VECTOR_ELT(out, 2, info);
    UNPROTECT(4);
    check(rc);
    return out;
}

/* G x 3 matrix + info for a hyper-parameter grid (non-PD points NaN, grid continues) */
SEXP gpmi_R_logml_grid(SEXP X, SEXP y, SEXP alpha, SEXP rho, SEXP sigma, SEXP jitter)
{
    int n = Rf_nrows(X), D = Rf_ncols(X), G = Rf_length(rho);
    need(is_real(X) && is_real(y) && is_real(alpha) && is_real(rho) && is_real(sigma), "X, y, alpha, rho, sigma must be double");
    need(Rf_length(y) == n, "length(y) must equal nrow(X)");
    need(Rf_length(alpha) == G && Rf_length(sigma) == G, "alpha, rho and sigma must have one entry per grid point (recycle in R)");
    SEXP res = PROTECT(Rf_allocMatrix(REALSXP, 3, G)), info = PROTECT(Rf_allocVector(INTSXP, G));
    int rc = gpmi_logml_grid(ctx(), REAL(X), n, n, D, REAL(y), REAL(alpha), REAL(rho), REAL(sigma), G,
                             Rf_asReal(jitter), REAL(res), INTEGER(info));
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2));
    SET_VECTOR_ELT(out, 0, res); SET_VECTOR_ELT(out, 1, info);
    UNPROTECT(3);
    check(rc);
    return out;
}

/* the same grid with a length-scale per dimension and point: ell is a D x G matrix (column g = point g), QQard's vector
 * phi[[2]] (R/kernels.R:11-19) */
SEXP gpmi_R_logml_grid_ard(SEXP X, SEXP y, SEXP alpha, SEXP ell, SEXP sigma, SEXP jitter)
{
    int n = Rf_nrows(X), D = Rf_ncols(X), G = Rf_ncols(ell);
    need(is_real(X) && is_real(y) && is_real(alpha) && is_real(ell) && is_real(sigma), "X, y, alpha, ell, sigma must be double");
    need(Rf_length(y) == n, "length(y) must equal nrow(X)");
    need(Rf_nrows(ell) == D, "ell must be ncol(X) x (grid points)");
    need(Rf_length(alpha) == G && Rf_length(sigma) == G, "alpha and sigma must have one entry per grid point (recycle in R)");
    SEXP res = PROTECT(Rf_allocMatrix(REALSXP, 3, G)), info = PROTECT(Rf_allocVector(INTSXP, G));
    int rc = gpmi_logml_grid_ard(ctx(), REAL(X), n, n, D, REAL(y), REAL(alpha), REAL(ell), REAL(sigma), G, Rf_asReal(jitter),
                                 REAL(res), INTEGER(info));
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2));
    SET_VECTOR_ELT(out, 0, res); SET_VECTOR_ELT(out, 1, info);
    UNPROTECT(3);
    check(rc);
    return out;
}

/* list(mn, Kn): p_Xn / p_dotXn (R/ode_gp.R:1-32, R/ode_gp_library.R:4-33), sample_derivs moments
 * (pendulum_fit.R:242-251) */
SEXP gpmi_R_gp_condition(SEXP t, SEXP ts, SEXP y, SEXP alpha, SEXP l, SEXP s2, SEXP jitter, SEXP kinds, SEXP flags)
{
    int n = Rf_length(t), m = Rf_length(ts);
    need(is_real(t) && is_real(ts) && is_real(y), "t, ts and y must be double");
    need(Rf_length(y) == n, "length(y) must equal length(t)");
    need(TYPEOF(kinds) == INTSXP && Rf_length(kinds) == 3, "kinds must be three integers (K, Ks, Kss)");
    int *k = INTEGER(kinds);
    SEXP mn = PROTECT(Rf_allocMatrix(REALSXP, m, 1)), Kn = PROTECT(Rf_allocMatrix(REALSXP, m, m));
    int rc = gpmi_gp_condition(ctx(), REAL(t), n, REAL(ts), m, REAL(y), Rf_asReal(alpha), Rf_asReal(l), Rf_asReal(s2),
                               Rf_asReal(jitter), k[0], k[1], k[2], Rf_asInteger(flags), REAL(mn), REAL(Kn), m);
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2));
    SET_VECTOR_ELT(out, 0, mn); SET_VECTOR_ELT(out, 1, Kn);
    UNPROTECT(3);
    check(rc);
    return out;
}

/* sample_derivs(params, ynoise, ti): pendulum_fit.R:227-255, one draw mu + chol(cov) z fused on the device; z = rnorm(m)
 * comes from R's own RNG (the reference's MASS::mvrnorm stream cannot be reproduced: eigen-decomposition draw) */
SEXP gpmi_R_sample_derivs(SEXP t, SEXP ts, SEXP y, SEXP params, SEXP jitter, SEXP z)
{
    int n = Rf_length(t), m = Rf_length(ts);
    need(is_real(t) && is_real(ts) && is_real(y) && is_real(params) && is_real(z), "t, ts, y, params and z must be double");
    need(Rf_length(y) == n, "length(ynoise) must equal length(ti)");
    need(Rf_length(z) == m, "length(z) must equal the number of prediction times");
    need(Rf_length(params) >= 3, "params must hold (l, a, sy)");
    double *p = REAL(params);
    SEXP draw = PROTECT(Rf_allocVector(REALSXP, m));
    int rc = gpmi_sample_derivs(ctx(), REAL(t), n, REAL(ts), m, REAL(y), p[0], p[1], p[2], Rf_asReal(jitter), REAL(z),
                                REAL(draw), NULL);
    UNPROTECT(1);
    check(rc);
    return draw;
}

/* the loop mclapply(s_list[1:100], sample_derivs_both_states, mc.cores = 2) of pendulum_fit.R:261-268 as ONE call:
 * params 3 x B, Y n x B, Z m x B -> draws m x B (independent conditionings on the GPU's lanes) */
SEXP gpmi_R_sample_derivs_batch(SEXP t, SEXP ts, SEXP Y, SEXP params, SEXP jitter, SEXP Z)
{
    int n = Rf_length(t), m = Rf_length(ts), B = Rf_ncols(Y);
    need(is_real(t) && is_real(ts) && is_real(Y) && is_real(params) && is_real(Z), "t, ts, Y, params and Z must be double");
    need(Rf_nrows(Y) == n, "nrow(Y) must equal length(ti)");
    need(Rf_nrows(Z) == m && Rf_ncols(Z) == B, "Z must be (prediction times) x (draws)");
    need(Rf_nrows(params) == 3 && Rf_ncols(params) == B, "params must be 3 x (draws): rows l, a, sy");
    SEXP draws = PROTECT(Rf_allocMatrix(REALSXP, m, B));
    int *info = (int *)R_alloc(B > 0 ? B : 1, sizeof(int));
    int rc = gpmi_sample_derivs_batch(ctx(), REAL(t), n, REAL(ts), m, REAL(Y), n, REAL(params), B, Rf_asReal(jitter), REAL(Z),
                                      m, REAL(draws), m, NULL, 0, info);
    UNPROTECT(1);
    check(rc);
    for (int b = 0; b < B; ++b) check(info[b]);
    return draws;
}

/* t(chol(A)) replacement: lower factor, upper zeroed (Stan convention); base-R chol() is t() of this */
SEXP gpmi_R_potrf(SEXP A)
{
    int n = Rf_nrows(A);
    need(is_real(A) && Rf_ncols(A) == n, "A must be a square double matrix");
    SEXP L = PROTECT(Rf_duplicate(A)); /* never modify a .Call argument */
    int rc = gpmi_potrf(ctx(), REAL(L), n, n > 0 ? n : 1);
    UNPROTECT(1);
    check(rc);
    return L;
}

/* create_p_dotXnS(Xn_list, mn, Kn, theta): R/ode_gp_library.R:43-93.  The sampler lives behind an
 * external pointer whose finalizer releases the device memory; step/commit are the two halves of
 * one closure call (the draw in between is R's own rnorm, so R's RNG stream is the reference's). */
static void seq_finalizer(SEXP ptr)
{
    gpmi_seq *q = (gpmi_seq *)R_ExternalPtrAddr(ptr);
    if (q) gpmi_seq_destroy(q);
    R_ClearExternalPtr(ptr);
}

SEXP gpmi_R_seq_create(SEXP X, SEXP mn, SEXP Kn, SEXP alpha, SEXP ell, SEXP jitter, SEXP max_steps)
{
    int n = Rf_nrows(X), D = Rf_ncols(X);
    need(is_real(X) && is_real(mn) && is_real(Kn) && is_real(ell), "X, mn, Kn and the length-scales must be double");
    need(Rf_length(mn) == n && Rf_nrows(Kn) == n && Rf_ncols(Kn) == n, "mn must have nrow(X) entries and Kn be nrow(X) x nrow(X)");
    need(Rf_length(ell) == 1 || Rf_length(ell) == D, "length-scale must have length 1 or ncol(X)");
    gpmi_seq *q = NULL;
    int rc = gpmi_seq_create(ctx(), &q, REAL(X), n, n, D, REAL(mn), REAL(Kn), n, Rf_asReal(alpha), REAL(ell),
                             Rf_length(ell), Rf_asReal(jitter), Rf_asInteger(max_steps));
    check(rc); /* nothing to release on failure: gpmi_seq_create frees what it allocated */
    SEXP ptr = PROTECT(R_MakeExternalPtr(q, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(ptr, seq_finalizer, TRUE);
    UNPROTECT(1);
    return ptr;
}

SEXP gpmi_R_seq_step(SEXP ptr, SEXP xs, SEXP D)
{
    gpmi_seq *q = (gpmi_seq *)R_ExternalPtrAddr(ptr);
    need(q != NULL, "the sampler has been released");
    need(is_real(xs) && Rf_length(xs) == Rf_asInteger(D), "xs must be a double vector with one entry per input dimension");
    SEXP out = PROTECT(Rf_allocVector(REALSXP, 2)); /* condMean, condVar */
    int rc = gpmi_seq_step(q, REAL(xs), REAL(out));
    UNPROTECT(1);
    check(rc);
    return out;
}

SEXP gpmi_R_seq_commit(SEXP ptr, SEXP dot_xs)
{
    need(R_ExternalPtrAddr(ptr) != NULL, "the sampler has been released");
    check(gpmi_seq_commit((gpmi_seq *)R_ExternalPtrAddr(ptr), Rf_asReal(dot_xs)));
    return R_NilValue;
}
