# gpmi.R -- R wrappers with the reference's function names and signatures over the
# .Call shim (r/gpmi_shim.c).  Two environments, exactly as the reference keeps two files
# whose names collide (SURVEY section 9 Q2): source ONE of
#   gpmi_kernels_env            <-> R/kernels.R            QQ, QR, RR, QQard (matrix API, phi list)
#   gpmi_derivative_kernels_env <-> derivative_kernels.R   QQ..TT(tj, tk, l) (elementwise API)
# R is not present in the build image; these wrappers are exercised through their Python
# mirror (gp_amd/*.py), which makes the same C-ABI calls.

.gpmi_kinds <- c(QQ = 0L, QR = 1L, RQ = 2L, RR = 3L, QT = 4L, TQ = 5L, RT = 6L, TR = 7L, TT = 8L)
GPMI_COMPAT_RR <- 2L   # reproduce R/kernels.R:31 as written (alpha^2 on the first term only)

gpmi_kernels_env <- local({
  compat_kernels_R_RR <- FALSE
  QQ <- function(x, y, phi) .Call("gpmi_R_deriv_cov", 0L, as.double(x), as.double(y), phi[[1]], phi[[2]], 0L)
  QR <- function(x, y, phi) .Call("gpmi_R_deriv_cov", 1L, as.double(x), as.double(y), phi[[1]], phi[[2]], 0L)
  RR <- function(x, y, phi) .Call("gpmi_R_deriv_cov", 3L, as.double(x), as.double(y), phi[[1]], phi[[2]],
                                  if (compat_kernels_R_RR) GPMI_COMPAT_RR else 0L)
  QQard <- function(X, Y, phi) {
    X <- as.matrix(X); Y <- as.matrix(Y); storage.mode(X) <- "double"; storage.mode(Y) <- "double"
    .Call("gpmi_R_se_cov", X, Y, phi[[1]], as.double(phi[[2]]))
  }
  # R/ode_gp.R:1-32 (mn/Kn) and R/ode_gp_library.R:4-33 (condMean/condVar): both name pairs
  .cond <- function(tn, Xn, phi_n, sigma_n, kinds, joint) {
    jit <- if (joint) 1e-6 else 0
    r <- .Call("gpmi_R_gp_condition", as.double(tn), as.double(tn), as.double(Xn), phi_n[[1]], phi_n[[2]],
               sigma_n^2 + jit, jit, kinds, if (compat_kernels_R_RR) GPMI_COMPAT_RR else 0L)
    list(mn = r[[1]], Kn = r[[2]], condMean = as.numeric(r[[1]]), condVar = r[[2]])
  }
  p_Xn <- function(tn, Xn, phi_n, sigma_n, joint = FALSE) .cond(tn, Xn, phi_n, sigma_n, c(0L, 0L, 0L), joint)
  p_dotXn <- function(tn, Xn, phi_n, sigma_n, joint = FALSE) .cond(tn, Xn, phi_n, sigma_n, c(0L, 2L, 3L), joint)
  environment()
})

gpmi_derivative_kernels_env <- local({
  mk <- function(k) function(tj, tk, l) {
    n <- max(length(tj), length(tk))
    .Call("gpmi_R_deriv_elem", k, rep_len(as.double(tj), n), rep_len(as.double(tk), n), l)
  }
  QQ <- mk(0L); QR <- mk(1L); RQ <- mk(2L); RR <- mk(3L); QT <- mk(4L)
  TQ <- mk(5L); RT <- mk(6L); TR <- mk(7L); TT <- mk(8L)
  # matrix-level fast path for a^2 * outer(ti, ti, FUN = kern_fixed_l(kern, l)), pendulum_fit.R:237-240
  se_deriv_cov <- function(t1, t2, l, alpha, block) .Call("gpmi_R_deriv_cov", .gpmi_kinds[[block]], as.double(t1),
                                                          as.double(t2), alpha, l, 0L)
  # sample_derivs(params, ynoise, ti): pendulum_fit.R:227-255, params = c(l, a, sy).  One draw mu + chol(cov) z, fused
  # on the GPU (the N x N covariance never leaves the device); z from R's RNG
  sample_derivs <- function(params, ynoise, ti, tis = ti, jitter = 1e-8)
    .Call("gpmi_R_sample_derivs", as.double(ti), as.double(tis), as.double(ynoise), as.double(params[1:3]), jitter,
          rnorm(length(tis)))
  # the mclapply(s_list[1:100], sample_derivs_both_states, mc.cores = 2) loop of pendulum_fit.R:261-268 as ONE call:
  # params_list: list of c(l, a, sy); ynoise_list: list of series -> matrix of draws, one column each
  sample_derivs_many <- function(params_list, ynoise_list, ti, tis = ti, jitter = 1e-8) {
    B <- length(params_list)
    .Call("gpmi_R_sample_derivs_batch", as.double(ti), as.double(tis), do.call(cbind, lapply(ynoise_list, as.double)),
          do.call(cbind, lapply(params_list, function(p) as.double(p[1:3]))), jitter, matrix(rnorm(length(tis) * B), ncol = B))
  }
  # moments only (mu, cov), as in round 1
  sample_derivs_moments <- function(params, ynoise, ti, tis = ti, jitter = 1e-8)
    .Call("gpmi_R_gp_condition", as.double(ti), as.double(tis), as.double(ynoise), params[2], params[1],
          params[3]^2, jitter, c(0L, 2L, 3L), 0L)
  environment()
})

# covariance.cpp:9-47
rbf_cov_chol <- function(x1, l_) .Call("gpmi_R_rbf_cov_chol", as.double(x1), l_)

# covariance.cpp:49-96 and models/cubic_interpolated_gp.hpp:38-73 (test_interpolate.R:9-19 builds the table)
approx_L <- function(l, lp, Ls, dLdls) .Call("gpmi_R_approx_L", l, as.double(lp), Ls, dLdls)
approx_Lz <- function(l, lp, Ls, dLdls, z) .Call("gpmi_R_approx_Lz", l, as.double(lp), Ls, dLdls, as.double(z))
approx_Lz_grad <- function(l, lp, Ls, dLdls, z) .Call("gpmi_R_approx_Lz_grad", l, as.double(lp), Ls, dLdls, as.double(z))
.gpmi_interp_n <- NULL   # order of the device-resident table
gp_interp_build <- function(x, lp) {
  .Call("gpmi_R_interp_build", as.double(x), as.double(lp))
  .gpmi_interp_n <<- length(x)
  invisible(NULL)
}
gp_interp_Lz <- function(l, z) {
  stopifnot(!is.null(.gpmi_interp_n), length(z) == .gpmi_interp_n)
  .Call("gpmi_R_approx_Lz", l, numeric(0), NULL, NULL, as.double(z))
}

# models/exact_gp.stan:17-25: f = cholesky_decompose(cov_exp_quad(x, alpha, rho) + 1e-10 I) * z, fused on the device
gp_exact_f <- function(X, alpha, rho, z, jitter = 1e-10)
  .Call("gpmi_R_exact_gp_f", as.matrix(X), alpha, as.double(rho), jitter, as.double(z))

# models/fit_hyperparameters.stan:18-32 as plain functions
gp_log_marginal <- function(X, y, alpha, rho, sigma, jitter = 0)
  .Call("gpmi_R_logml", as.matrix(X), as.double(y), alpha, as.double(rho), sigma, jitter)[1]

# value and gradient w.r.t. (alpha, rho, sigma): feeds optim(..., method = "L-BFGS-B") in place of a Stan fit
gp_log_marginal_grad <- function(X, y, alpha, rho, sigma, jitter = 0)
  .Call("gpmi_R_logml_grad", as.matrix(X), as.double(y), alpha, as.double(rho), sigma, jitter)

# the same for several chains' (alpha, rho, sigma) at once (rstan: chains = 4), concurrently on the GPU
gp_log_marginal_grad_chains <- function(X, y, alpha, rho, sigma, jitter = 0) {
  G <- max(length(alpha), length(rho), length(sigma))   # R recycling: gp_log_marginal_grad_chains(X, y, 1, rho_vec, 0.1)
  .Call("gpmi_R_logml_grad_grid", as.matrix(X), as.double(y), rep_len(as.double(alpha), G), rep_len(as.double(rho), G),
        rep_len(as.double(sigma), G), jitter)
}

gp_log_marginal_grid <- function(X, y, alpha, rho_vec, sigma_vec, jitter = 0) {
  g <- expand.grid(rho = rho_vec, sigma = sigma_vec)
  r <- .Call("gpmi_R_logml_grid", as.matrix(X), as.double(y), rep_len(as.double(alpha), nrow(g)), as.double(g$rho),
             as.double(g$sigma), jitter)
  matrix(r[[1]][1, ], nrow = length(rho_vec), ncol = length(sigma_vec))
}

# ARD points: ell_mat is D x G (column g = the length-scale vector of point g, QQard's phi[[2]]); returns logml per point
gp_log_marginal_points_ard <- function(X, y, alpha, ell_mat, sigma, jitter = 0) {
  ell_mat <- as.matrix(ell_mat); storage.mode(ell_mat) <- "double"
  G <- ncol(ell_mat)
  r <- .Call("gpmi_R_logml_grid_ard", as.matrix(X), as.double(y), rep_len(as.double(alpha), G), ell_mat,
             rep_len(as.double(sigma), G), jitter)
  r[[1]][1, ]
}

# arg-max over the grid: mirror of get_ml_from_stan_samples, R/tests.R:21-27
get_ml_from_grid <- function(values, alpha, rho_vec, sigma_vec) {
  idx <- which(values == max(values, na.rm = TRUE), arr.ind = TRUE)[1, ]
  list(alpha = alpha, rho = rho_vec[idx[1]], sigma = sigma_vec[idx[2]])
}

# create_p_dotXnS(Xn_list, mn, Kn, theta): R/ode_gp_library.R:43-93, same closure, same return
# list(mu, sigma, dot_xs); the O(N^3) algebra of :55-57 runs once on the GPU, a call reads the
# Cholesky factor and one N x N matrix instead of re-solving for every star point so far.
# rnorm(1, condMean, condVar) is kept exactly as the reference writes it (:83).
create_p_dotXnS <- function(Xn_list, mn, Kn, theta, max_steps = 256L) {
  X <- do.call(cbind, Xn_list)
  h <- .Call("gpmi_R_seq_create", as.matrix(X), as.double(mn), as.matrix(Kn), theta[[1]], as.double(theta[[2]]),
             1e-6, as.integer(max_steps))
  p_dotXnS <- function(xs_vec) {
    r <- .Call("gpmi_R_seq_step", h, as.double(xs_vec), ncol(X))
    dot_xs <- rnorm(1, r[1], r[2])
    .Call("gpmi_R_seq_commit", h, dot_xs)
    list(mu = r[1], sigma = r[2], dot_xs = dot_xs)
  }
}
