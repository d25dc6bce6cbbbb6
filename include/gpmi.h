/*
 * gpmi.h -- C ABI of libgpmi: the MI355X (gfx950) implementation of the exact-GP
 * marginal-likelihood hot path of bbbales2/gp.
 *
 * This is the drop-in boundary: what an R `.Call()` shim (r/gpmi_shim.c), an
 * Rcpp replacement of covariance.cpp, or any FFI binds.  Plain pointers and
 * sizes only; no C++ / torch types.  Reference interfaces replaced are cited
 * per entry point (paths relative to the reference repository root).
 *
 * Conventions
 *  - All matrices column-major doubles (R's native layout) with explicit
 *    leading dimension.  X is n x D column-major, i.e. each input dimension is
 *    one contiguous stream.
 *  - Every function returns int: 0 ok; k > 0 = "leading minor of order k is not
 *    positive definite" (LAPACK dpotrf style; mirrors base-R chol()'s error and
 *    Stan's cholesky_decompose domain_error); < 0 = GPMI_E* (text through
 *    gpmi_last_error()).  Nothing throws or aborts across this boundary.
 *  - Host-buffer entry points (no suffix) take ordinary host memory, stage it
 *    through HBM and block until the result is back: this is what `.Call`
 *    binds.  `_dev` entry points take device pointers (e.g. torch tensors'
 *    data_ptr()), enqueue on the context's stream and do NOT synchronise.
 *  - A gpmi_ctx owns its device workspace and stream; it is bound to one GPU
 *    and to the creating process.  HIP state does not survive fork() and cannot
 *    be re-initialised in the child: once the library has touched the GPU in a
 *    process, EVERY call from a forked child of it -- e.g. parallel::mclapply
 *    at pendulum_fit.R:268 --, gpmi_create and gpmi_device_count included,
 *    returns GPMI_EFORK.  Workers must be fresh processes (or fork before the
 *    first gpmi call); prefer the grid API to fork-per-draw.
 *  - There is no CPU fallback anywhere in the library: without a HIP device
 *    gpmi_create fails with GPMI_ENODEV.
 */
#ifndef GPMI_H
#define GPMI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMI_VERSION 302

/* the ABI: the ONLY symbols libgpmi.so exports (it is built with -fvisibility=hidden) */
#define GPMI_API __attribute__((visibility("default")))

typedef struct gpmi_ctx gpmi_ctx;
typedef struct gpmi_seq gpmi_seq; /* sequential conditional sampler (gpmi_seq_*) */

enum {
    GPMI_OK = 0,
    GPMI_EARG = -1,   /* bad argument                              */
    GPMI_EHIP = -2,   /* HIP runtime error                         */
    GPMI_ENOMEM = -3, /* device allocation failed                  */
    GPMI_ENODEV = -4, /* no usable gfx950 device                   */
    GPMI_EFORK = -5   /* called from a forked child of a GPU process */
};

/* derivative-kernel selector: derivative_kernels.R:39-73 (Q value, R first,
 * T second derivative of the process; row argument first) */
enum {
    GPMI_QQ = 0, GPMI_QR = 1, GPMI_RQ = 2, GPMI_RR = 3, GPMI_QT = 4,
    GPMI_TQ = 5, GPMI_RT = 6, GPMI_TR = 7, GPMI_TT = 8
};

/* flags for gpmi_se_cov* */
enum {
    GPMI_FULL = 0,        /* write all n x m entries                              */
    GPMI_LOWER = 1,       /* square case only: write i >= j, leave the rest alone */
    GPMI_COMPAT_RR = 2    /* reproduce R/kernels.R:31's amplitude precedence bug  */
};

/* ---- library / context ------------------------------------------------ */
GPMI_API int gpmi_version(void);
GPMI_API const char *gpmi_last_error(void);
GPMI_API int gpmi_device_count(int *count);
GPMI_API int gpmi_create(gpmi_ctx **ctx, int device);
GPMI_API int gpmi_destroy(gpmi_ctx *ctx);
/* run on a caller-owned hipStream_t; NULL (handle 0) is HIP's legacy null stream -- what
 * torch's default stream is -- so `_dev` calls are ordered with the caller's other work there */
GPMI_API int gpmi_set_stream(gpmi_ctx *ctx, void *hip_stream);
/* back to the context's own (non-blocking) stream */
GPMI_API int gpmi_reset_stream(gpmi_ctx *ctx);
GPMI_API int gpmi_sync(gpmi_ctx *ctx);
/* pre-size the factorisation workspace for matrices of order <= n_max */
GPMI_API int gpmi_reserve(gpmi_ctx *ctx, int n_max);
/* algorithm switches of THIS context (never process-global): "nb_outer" (outer panel width, multiple
 * of 128, 0 = auto), "grid_lanes", "fuse_diag", "ksplit", "block_recursive", "stagger", "se_nt", "nb_adapt",
 * "nb_thr1024", "nb_thr512", "nb_thr256", "small_n", "small_n1", "small_m" (one-workgroup kernels for small
 * problems), "small_ng1", "small_ng" (value + gradient by one workgroup: one evaluation up to n <= small_ng1, several at once
 * up to n <= small_ng <= 256), "grad_aug_n", "grad_aug_ng" (value + gradient through ONE augmented partial factorisation up
 * to this n: one evaluation / several at once), "small_gc" (gpmi_gp_condition in one launch up to n + m + 1 <= small_gc rows), "small_sd", "small_sdb" (gpmi_sample_derivs[_batch]: one workgroup per draw up to
 * n + m + 1 <= small_sd rows for at least small_sdb ((n + m + 1) / 400)^2 draws), "small_n2", "small_g2" (grids of at least small_g2 (n / 1024)^2 + 2 points run one workgroup per point up
 * to n <= small_n2 <= 1024), "calibrate", "timing", "kernel_timing"; unknown names return GPMI_EARG.  Switches of variants that
 * were measured and rejected ("lookahead", "syrk_order", "diag_waves", "gemm_variant", ...) exist in the probe
 * build only. */
GPMI_API int gpmi_set_option(gpmi_ctx *ctx, const char *name, int value);

/* ---- covariance builders ---------------------------------------------- */

/* K[i,j] = alpha^2 exp(-1/2 sum_d ((X[i,d]-Y[j,d])/ell[d])^2) (+ diag_add on
 * i == j when X and Y are the same n points).  n_ell = 1 (isotropic) or D (ARD).
 * Replaces QQard(X,Y,phi) R/kernels.R:11-19, QQ(x,y,phi) R/kernels.R:22-24
 * (D = 1), Stan cov_exp_quad models/fit_hyperparameters.stan:19 and the
 * diagonal update :21-24 / models/exact_gp.stan:20-22 (diag_add).
 * Y == NULL means Y = X (symmetric; enables GPMI_LOWER). */
GPMI_API int gpmi_se_cov(gpmi_ctx *ctx, const double *X, int n, int ldx, const double *Y, int m, int ldy,
                int D, double alpha, const double *ell, int n_ell, double diag_add, int flags,
                double *K, int ldk);
GPMI_API int gpmi_se_cov_dev(gpmi_ctx *ctx, const double *dX, int n, int ldx, const double *dY, int m,
                    int ldy, int D, double alpha, const double *ell, int n_ell, double diag_add,
                    int flags, double *dK, int ldk);

/* K[i,j] = alpha^2 * kind(x[i], y[j], l), 1-D inputs.  Replaces the matrix API
 * QQ/QR/RR(x,y,phi) of R/kernels.R:22-32 (GPMI_COMPAT_RR for :31 as written)
 * and a^2 * outer(ti, ti, FUN = kern) of pendulum_fit.R:237-240. */
GPMI_API int gpmi_deriv_cov(gpmi_ctx *ctx, int kind, const double *x, int n, const double *y, int m,
                   double alpha, double l, int flags, double *K, int ldk);
GPMI_API int gpmi_deriv_cov_dev(gpmi_ctx *ctx, int kind, const double *dx, int n, const double *dy, int m,
                       double alpha, double l, int flags, double *dK, int ldk);

/* out[i] = kind(tj[i], tk[i], l): the vectorised elementwise functions
 * QQ..TT(tj,tk,l) of derivative_kernels.R:39-73 (unit amplitude). */
GPMI_API int gpmi_deriv_elem(gpmi_ctx *ctx, int kind, const double *tj, const double *tk, size_t len,
                    double l, double *out);

/* Joint [values; derivatives] covariance of order 2n,
 * [[QQ + sigma^2 I, QR], [RQ, RR]] + jitter I   (R/ode_gp_library.R:29-30). */
GPMI_API int gpmi_joint_cov(gpmi_ctx *ctx, const double *t, int n, double alpha, double l, double sigma,
                   double jitter, int flags, double *K, int ldk);

/* ---- dense factorisation ---------------------------------------------- */

/* In-place lower Cholesky A = L L^T (strict upper triangle zeroed on return,
 * like Stan's cholesky_decompose: models/fit_hyperparameters.stan:25,
 * models/exact_gp.stan:23; base-R chol() returns t(L)). */
GPMI_API int gpmi_potrf(gpmi_ctx *ctx, double *A, int n, int lda);
GPMI_API int gpmi_potrf_dev(gpmi_ctx *ctx, double *dA, int n, int lda, int *d_info);

/* f = L z (models/exact_gp.stan:25) and z = L^-1 b (mdivide_left_tri_low inside
 * multi_normal_cholesky, models/fit_hyperparameters.stan:31). */
GPMI_API int gpmi_trmv_lower(gpmi_ctx *ctx, const double *L, int n, int ldl, const double *z, double *f);
GPMI_API int gpmi_trsv_lower(gpmi_ctx *ctx, const double *L, int n, int ldl, const double *b, double *z);
/* The latent exact GP's transform in ONE call, models/exact_gp.stan:17-25 (test_interpolate.R:31-36 runs it at N = 100):
 * f = cholesky_decompose(cov_exp_quad(X, alpha, ell) + jitter I) z -- covariance, factor and product stay on the
 * device, only z goes in and f comes out (n <= 256: one launch of one workgroup).  Returns 0, or the order of the first
 * non-positive leading minor (f is NaN then). */
GPMI_API int gpmi_exact_gp_f(gpmi_ctx *ctx, const double *X, int n, int ldx, int D, double alpha, const double *ell, int n_ell,
                    double jitter, const double *z, double *f);

/* ---- marginal likelihood ---------------------------------------------- */

/* One evaluation of models/fit_hyperparameters.stan:18-32 with double inputs:
 * Sigma = cov_exp_quad(X, alpha, rho) + (sigma^2 + jitter) I; L = chol(Sigma);
 * out[0] = -1/2 z'z - sum log L_ii - n/2 log(2 pi), out[1] = sum log L_ii,
 * out[2] = z'z, z = L^-1 y.  ell/n_ell as in gpmi_se_cov (rho == ell[0]). */
GPMI_API int gpmi_logml(gpmi_ctx *ctx, const double *X, int n, int ldx, int D, const double *y,
               double alpha, const double *ell, int n_ell, double sigma, double jitter,
               double *out3);
/* device-resident X, y; d_out3 (3 doubles) and d_info (1 int) in device memory */
GPMI_API int gpmi_logml_dev(gpmi_ctx *ctx, const double *dX, int n, int ldx, int D, const double *dy,
                   double alpha, const double *ell, int n_ell, double sigma, double jitter,
                   double *d_out3, int *d_info);

/* Log marginal likelihood AND its gradient with respect to (alpha, ell[0..n_ell), sigma):
 * grad[0] = d/dalpha, grad[1 .. n_ell] = d/dell, grad[1 + n_ell] = d/dsigma.  This is what Stan's
 * autodiff computes per leapfrog step for models/fit_hyperparameters.stan:18-32 (the reference has
 * no function of its own for it); 1/2 tr((a a' - K^-1) dK/dtheta) with K^-1 formed on the device.
 * Any D the covariance builder takes (<= 64; QQard, R/kernels.R:11-19).  Same status codes as gpmi_logml
 * (k > 0: not positive definite, grad = NaN). */
GPMI_API int gpmi_logml_grad(gpmi_ctx *ctx, const double *X, int n, int ldx, int D, const double *y,
                    double alpha, const double *ell, int n_ell, double sigma, double jitter,
                    double *out3, double *grad);

/* Value AND gradient at G independent points (alpha[g], rho[g], sigma[g]), concurrently on the context's lanes: what
 * rstan's default four chains (pendulum_fit.R:140: chains = 4, cores = 4) ask for per leapfrog step.  out3: 3 G;
 * grad: 3 G, (d/dalpha, d/drho, d/dsigma) per point; info: G (non-PD: NaN, the grid continues).  D <= 64. */
GPMI_API int gpmi_logml_grad_grid(gpmi_ctx *ctx, const double *X, int n, int ldx, int D, const double *y,
                                  const double *alpha, const double *rho, const double *sigma, int G, double jitter,
                                  double *out3, double *grad, int *info);

/* G independent hyper-parameter points (alpha[g], rho[g], sigma[g]) on the same
 * data: out3[3*g..], info[g].  Non-PD points get NaN and info[g] = k and the
 * grid continues.  Replaces the stan()-fit + arg-max of R/tests.R:13-27 when a
 * grid search is acceptable; sharding over GPUs is done by the host layer
 * (one context per device / process). */
GPMI_API int gpmi_logml_grid(gpmi_ctx *ctx, const double *X, int n, int ldx, int D, const double *y,
                    const double *alpha, const double *rho, const double *sigma, int G,
                    double jitter, double *out3, int *info);
GPMI_API int gpmi_logml_grid_dev(gpmi_ctx *ctx, const double *dX, int n, int ldx, int D, const double *dy,
                        const double *alpha, const double *rho, const double *sigma, int G,
                        double jitter, double *d_out3, int *d_info);

/* The same grid with ONE LENGTH-SCALE PER DIMENSION and point (ARD): ell is G x D, point-major (ell[g * D + d]).
 * QQard(X, Y, phi) takes a vector phi[[2]] (R/kernels.R:11-19); a grid search or optimiser over ARD
 * length-scales evaluates exactly this.  Everything else as gpmi_logml_grid. */
GPMI_API int gpmi_logml_grid_ard(gpmi_ctx *ctx, const double *X, int n, int ldx, int D, const double *y,
                        const double *alpha, const double *ell, const double *sigma, int G,
                        double jitter, double *out3, int *info);
GPMI_API int gpmi_logml_grid_ard_dev(gpmi_ctx *ctx, const double *dX, int n, int ldx, int D, const double *dy,
                            const double *alpha, const double *ell, const double *sigma, int G,
                            double jitter, double *d_out3, int *d_info);

/* Log marginal likelihood of stacked observations yy = [y; y'] (length 2n)
 * under gpmi_joint_cov's matrix (BASELINE config c5). */
GPMI_API int gpmi_joint_logml(gpmi_ctx *ctx, const double *t, int n, const double *yy, double alpha,
                     double l, double sigma, double jitter, double *out3);
GPMI_API int gpmi_joint_logml_dev(gpmi_ctx *ctx, const double *dt, int n, const double *dyy, double alpha,
                         double l, double sigma, double jitter, double *d_out3, int *d_info);
/* G independent (alpha[g], l[g], sigma[g]) points of the joint model on the same device-resident data, concurrently
 * on the context's lanes: d_out3[3 g ..], d_info[g] (non-PD points get NaN and the grid continues). */
GPMI_API int gpmi_joint_logml_grid_dev(gpmi_ctx *ctx, const double *dt, int n, const double *dyy, const double *alpha,
                                       const double *l, const double *sigma, int G, double jitter, double *d_out3,
                                       int *d_info);

/* ---- Rcpp export ------------------------------------------------------ */

/* rbf_cov_chol(x1, l): Sigma_ij = exp(-(xi-xj)^2/(2 l^2)) + 1e-10 I, L = chol,
 * dLdl = dL/dl (forward-mode tangent).  covariance.cpp:9-47.  L, dLdl n x n. */
GPMI_API int gpmi_rbf_cov_chol(gpmi_ctx *ctx, const double *x, int n, double l, double *L, int ldl,
                      double *dLdl, int lddl);

/* ---- Cholesky-factor interpolation over the length-scale ---------------
 * Table of P factors L(lp[p]) and tangents dL/dl(lp[p]) kept in HBM (2 P n^2 doubles), then
 * piecewise cubic Hermite blends at any l.  Interval rule of the reference: the first p with
 * lp[p+1] >= l (clamped to the last interval, where the reference reads past the end).
 *   gpmi_interp_build : the table test_interpolate.R:9-19 builds with P calls of rbf_cov_chol
 *   gpmi_interp_load  : a caller-supplied table (P stacked n x n column-major matrices, leading
 *                       dimension ld, matrix p at Ls + p*ld*n) -- the `Ls`, `dLdls` data of
 *                       models/cubic_interpolated_gp.stan:11-12
 *   gpmi_approx_L     : covariance.cpp:49-96 (lower triangle blended, zeros above)
 *   gpmi_approx_Lz    : models/cubic_interpolated_gp.hpp:38-73, f = approx_L(l) z, fused (the
 *                       blended matrix is never stored: 4 n^2/2 doubles read per call)
 *   gpmi_approx_Lz_grad : the same value plus dfdl = (dv/dl) z, the partial Stan's reverse mode
 *                       attaches to every output of approx_Lz (`var` overload of build_output,
 *                       cubic_interpolated_gp.hpp:6-32, dvdl :67) -- what makes the external
 *                       function differentiable in l (cubic_interpolated_gp.stan:1-3); same pass,
 *                       same traffic.  (d f / d z is the blend itself: gpmi_approx_L.)
 * gpmi_interp_build factors the P table entries concurrently on the context's grid lanes.      */
GPMI_API int gpmi_interp_build(gpmi_ctx *ctx, const double *x, int n, const double *lp, int P);
GPMI_API int gpmi_interp_load(gpmi_ctx *ctx, const double *lp, int P, const double *Ls, const double *dLdls,
                     int n, int ld);
GPMI_API int gpmi_approx_L(gpmi_ctx *ctx, double l, double *out, int ldo);
GPMI_API int gpmi_approx_Lz(gpmi_ctx *ctx, double l, const double *z, double *f);
GPMI_API int gpmi_approx_Lz_dev(gpmi_ctx *ctx, double l, const double *dz, double *df);
GPMI_API int gpmi_approx_Lz_grad(gpmi_ctx *ctx, double l, const double *z, double *f, double *dfdl);
GPMI_API int gpmi_approx_Lz_grad_dev(gpmi_ctx *ctx, double l, const double *dz, double *df, double *ddfdl);
GPMI_API int gpmi_interp_free(gpmi_ctx *ctx);

/* ---- GP posterior (value / derivative) -------------------------------- */

/* mn = Ks (K + s2 I)^-1 y,  Kn = Kss - Ks (K + s2 I)^-1 Ks^T + jitter I with
 * K = alpha^2 kindK(t,t), Ks = alpha^2 kindS(ts,t), Kss = alpha^2 kindSS(ts,ts)
 * through ONE Cholesky of K + s2 I (the reference does two LU solves).
 *  p_Xn      R/ode_gp.R:1-14     : (QQ, QQ, QQ), ts = t
 *  p_dotXn   R/ode_gp.R:19-32    : (QQ, RQ, RR), ts = t
 *  sample_derivs moments pendulum_fit.R:242-251 : (QQ, RQ, RR), jitter 1e-8
 *  lorenz.Rmd:80-107 variant: separate prediction times ts.
 * mn: m doubles; Kn: m x m (symmetric, both triangles written). */
GPMI_API int gpmi_gp_condition(gpmi_ctx *ctx, const double *t, int n, const double *ts, int m,
                      const double *y, double alpha, double l, double s2, double jitter,
                      int kindK, int kindS, int kindSS, int flags, double *mn, double *Kn, int ldkn);

/* One draw of the derivative process given noisy values: sample_derivs(params = (l, a, sy), ynoise, ti),
 * pendulum_fit.R:227-255 (separate prediction times ts: lorenz.Rmd:80-107, jitter 1e-6 there, 1e-8 here):
 * draw = mu + chol(cov) z with the (QQ, RQ, RR) moments of gpmi_gp_condition, fused on the device -- the
 * m x m covariance is factored in place in the workspace and never crosses PCIe.  The reference draws
 * with MASS::mvrnorm and R's unseeded RNG, so the standard-normal variate z (m doubles) is the caller's;
 * mu (nullable) receives the mean.  Status k in 1..n: K + sy^2 I is not positive definite at order k;
 * n + k: the posterior covariance is not (at order k). */
GPMI_API int gpmi_sample_derivs(gpmi_ctx *ctx, const double *t, int n, const double *ts, int m, const double *y,
                                double l, double a, double sy, double jitter, const double *z, double *draw, double *mu);
/* B independent draws -- the loop mclapply(s_list[1:100], sample_derivs_both_states, mc.cores = 2) of
 * pendulum_fit.R:261-268 (one posterior draw of (l, a, sy) and one noisy series per call) -- run as
 * concurrent conditionings on the context's lanes.  params: 3 x B (l, a, sy per draw); Y: n x B, Z: m x B,
 * draws: m x B, mus (nullable): m x B, column-major with the given leading dimensions; info[b] as the
 * status of gpmi_sample_derivs (a failed draw does not stop the others). */
GPMI_API int gpmi_sample_derivs_batch(gpmi_ctx *ctx, const double *t, int n, const double *ts, int m, const double *Y,
                                      int ldy, const double *params, int B, double jitter, const double *Z, int ldz,
                                      double *draws, int ldd, double *mus, int ldmu, int *info);

/* ---- sequential conditional sampler ------------------------------------ */

/* create_p_dotXnS(Xn_list, mn, Kn, theta), R/ode_gp_library.R:43-93 (R/tests.R:78-91): a stateful
 * sampler of the derivative at new states xs, one at a time, each conditioned on the draws
 * already made.  X = do.call(cbind, Xn_list) (n x D), theta = (alpha, ell) of QQard; mn, Kn the
 * derivative posterior at the data (p_dotXn); jitter = 1e-6 in the reference (:55 and :77).
 * With K~ = K_XX + jitter I the joint law of the star points is N(m, K),
 *   m = K_XsX K~^-1 mn,  K = K_XsXs - K_XsX (K~^-1 - K~^-1 Kn K~^-1) K_XXs + jitter I   (:74-77);
 * gpmi_seq_step returns out2 = (condMean, condVar) of the NEW point given the committed draws
 * (what condMVN returns at :80-81; the closure's `mu`, `sigma`), gpmi_seq_commit appends the
 * value drawn for it (the reference draws rnorm(1, condMean, condVar) at :83 with R's RNG, so
 * the draw itself belongs to the caller).  A step that is not committed is discarded by the next
 * step.  The O(n^3) work (one Cholesky, two n-row triangular solves) happens once in
 * gpmi_seq_create; a step reads 12 n^2 B (factor + one n x n matrix).  max_steps <= 2048.
 * Status k > 0: K~ (create) or the star covariance (step, k = its order) is not positive definite. */
GPMI_API int gpmi_seq_create(gpmi_ctx *ctx, gpmi_seq **out, const double *X, int n, int ldx, int D,
                    const double *mn, const double *Kn, int ldkn, double alpha, const double *ell,
                    int n_ell, double jitter, int max_steps);
GPMI_API int gpmi_seq_step(gpmi_seq *seq, const double *xs /* D */, double *out2);
GPMI_API int gpmi_seq_commit(gpmi_seq *seq, double dot_xs);
GPMI_API int gpmi_seq_count(const gpmi_seq *seq); /* committed draws */
GPMI_API int gpmi_seq_destroy(gpmi_seq *seq);

/* ---- diagnostics (tests / bench) -------------------------------------- */

/* Elapsed ms of the most recent evaluation's stages, measured with HIP events
 * on the context's stream when timing is enabled via gpmi_set_option("timing",1):
 * ms[0] covariance build, ms[1] Cholesky, ms[2] finalize; n_launch[0..2]. */
GPMI_API int gpmi_last_timing(gpmi_ctx *ctx, double *ms3);

/* Per-kernel HIP-event timing, enabled with gpmi_set_option("kernel_timing", 1): event
 * pairs bracket every covariance-build launch (category 0; work = bytes written), every
 * trailing-update SYRK launch (1; work = algorithmic flops) on the context's stream.
 * out9[3*cat + 0..2] = launches, total ms, total work since the last reset. */
GPMI_API int gpmi_kernel_timing(gpmi_ctx *ctx, int reset, double *out9);
/* The same with out12[9..11] = launches, total ms, total flops of the trailing-update launches of more than one
 * round of tiles -- the throughput-bound subset of category 1 (a single-round launch carries the next diagonal
 * block's factorisation and is bounded by that latency chain). */
GPMI_API int gpmi_kernel_timing_ex(gpmi_ctx *ctx, int reset, double *out12);

/* ---- probes: tools/ only ------------------------------------------------
 * Built into libgpmi_probes.so (-DGPMI_PROBES: `python -m gp_amd._build --probes`), never into the
 * shipped libgpmi.so: micro-benchmarks, the A/B kernel variants and their option names
 * (gemm_variant, rect_auto, debug_topology). */
#ifdef GPMI_PROBES
/* Stand-alone trailing-update (SYRK, lower) launch on synthetic data, C(m x m) -= P P^T with
 * P m x k: average ms per launch over `reps` back-to-back launches (HIP events). */
GPMI_API int gpmi_probe_syrk(gpmi_ctx *ctx, int m, int k, int reps, double *ms);

/* Shader-clock probe of the SYRK kernel: out3 = summed shader cycles, summed 100 MHz ticks and number of
 * workgroups of every trailing-update launch since the last reset. */
GPMI_API int gpmi_probe_clock(gpmi_ctx *ctx, int reset, double *out3);

/* Fused in-block launches since the last call, block 0: cycles in the sub-tile product, in the wait for the
 * other two sub-tiles, in the diagonal-block body; number of launches; sum of their K. */
GPMI_API int gpmi_probe_fused(gpmi_ctx *ctx, double *out5);
/* Diagonal-block bodies (128-pivot factorisations) since the last call: shader cycles in the block loads, the 8-step
 * loop and the stores of tile wave 0, inside factor16 and waiting for the next diagonal tile (factor wave); bodies. */
GPMI_API int gpmi_probe_body(gpmi_ctx *ctx, double *out6);
/* One-workgroup small-N kernels since the last call (block 0 of every launch): shader cycles in the covariance
 * build, the diagonal blocks, the rows below, number of launches, cycles in the trailing tiles, in the finalize. */
GPMI_API int gpmi_probe_small(gpmi_ctx *ctx, double *out6);

/* MFMA f64 fragment-layout probe: D = A(16x4) * B(4x16) on one wave with the
 * library's fragment conventions; out256 row-major D[i][j].  Host buffers. */
GPMI_API int gpmi_probe_mfma(gpmi_ctx *ctx, const double *A64, const double *B64, double *out256);
/* Back-to-back v_mfma_f64_16x16x4_f64 issue-rate microbenchmark: achieved TFLOP/s over
 * the whole chip and (nullable) the shader clock held meanwhile, from
 * d(s_memtime)/d(s_memrealtime). */
GPMI_API int gpmi_probe_mfma_peak(gpmi_ctx *ctx, int iters, double *tflops, double *clock_mhz);
#endif /* GPMI_PROBES */

#ifdef __cplusplus
}
#endif
#endif /* GPMI_H */
