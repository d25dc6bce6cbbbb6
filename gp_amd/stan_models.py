"""Double-precision evaluations of the reference's Stan model blocks on the GPU.

models/fit_hyperparameters.stan:18-32 (== stan/fit_hyperparameters.stan):
    Sigma = cov_exp_quad(t, alpha, rho) + sigma^2 I;  L = cholesky_decompose(Sigma)
    y ~ multi_normal_cholesky(0, L)
models/exact_gp.stan:16-26:  f = cholesky_decompose(cov_exp_quad(x, 1, l) + 1e-10 I) * z
"""
import math

import numpy as np

from ._lib import NotPositiveDefinite, default_context


def gp_log_marginal(X, y, alpha, rho, sigma, jitter=0.0, ctx=None):
    """log p(y | X, alpha, rho, sigma) = -1/2 z'z - sum log L_ii - N/2 log(2 pi)."""
    return (ctx or default_context()).logml(X, y, alpha, rho, sigma, jitter)[0]


def stan_lp(sum_log_diag, quad, alpha, rho, sigma):
    """lp__ of fit_hyperparameters.stan: likelihood without the
    -N/2 log 2pi constant (`~` drops it, :31), gamma(4,4) / half-normal(0,1) priors without
    constants (:27-29), plus the log-Jacobian of the <lower=0> transforms (:13-15)."""
    return (-sum_log_diag - 0.5 * quad + (3.0 * math.log(rho) - 4.0 * rho) - 0.5 * alpha * alpha
            - 0.5 * sigma * sigma + (math.log(rho) + math.log(alpha) + math.log(sigma)))


def fit_hyperparameters_log_prob(t, y, rho, alpha, sigma, ctx=None):
    """Stan's lp__ for one (rho, alpha, sigma); -inf when Sigma is not positive definite
    (Stan rejects the proposal on cholesky_decompose's domain_error)."""
    try:
        _, sld, q = (ctx or default_context()).logml(t, y, alpha, rho, sigma, 0.0)
    except NotPositiveDefinite:
        return -math.inf
    return stan_lp(sld, q, alpha, rho, sigma)


def fit_hyperparameters_log_prob_grad(t, y, rho, alpha, sigma, ctx=None):
    """(lp__, d lp__/d(rho, alpha, sigma)) on the constrained scale -- the value/gradient pair Stan's
    autodiff produces per leapfrog step for fit_hyperparameters.stan; feeds an optimiser or HMC outside
    Stan.  Prior and Jacobian terms of stan_lp() differentiate to 4/rho - 4, 1/alpha - alpha,
    1/sigma - sigma."""
    try:
        out, g = (ctx or default_context()).logml_grad(t, y, alpha, [rho], sigma, 0.0)
    except NotPositiveDefinite:
        return -math.inf, np.full(3, math.nan)
    lp = stan_lp(out[1], out[2], alpha, rho, sigma)
    return lp, np.array([g[1] + 4.0 / rho - 4.0, g[0] + 1.0 / alpha - alpha, g[2] + 1.0 / sigma - sigma])


def fit_hyperparameters_log_prob_grad_chains(t, y, rho, alpha, sigma, ctx=None):
    """The same for several chains at once -- rstan runs `chains = 4` (pendulum_fit.R:140), each asking for one
    value + gradient per leapfrog step: (lp (C,), grad (C, 3) in (rho, alpha, sigma) order), evaluated
    concurrently on the GPU's lanes; rejected (non-PD) proposals get -inf / NaN."""
    rho = np.atleast_1d(np.asarray(rho, float)); alpha = np.atleast_1d(np.asarray(alpha, float))
    sigma = np.atleast_1d(np.asarray(sigma, float))
    out, g, info = (ctx or default_context()).logml_grad_grid(t, y, alpha, rho, sigma, 0.0)
    lp = np.array([stan_lp(o[1], o[2], a, r, s) if i == 0 else -math.inf for o, a, r, s, i in zip(out, alpha, rho, sigma, info)])
    grad = np.column_stack([g[:, 1] + 4.0 / rho - 4.0, g[:, 0] + 1.0 / alpha - alpha, g[:, 2] + 1.0 / sigma - sigma])
    grad[info != 0] = math.nan
    return lp, grad


def gp_log_marginal_grid(X, y, alpha, rho_vec, sigma_vec, jitter=0.0, lp=False, ctx=None):
    """|rho| x |sigma| matrix of log marginal likelihoods (lp=True: Stan lp__ instead);
    non-PD points are NaN (-inf for lp) and the grid continues."""
    rho_vec = np.atleast_1d(np.asarray(rho_vec, float)); sigma_vec = np.atleast_1d(np.asarray(sigma_vec, float))
    R, S = np.meshgrid(rho_vec, sigma_vec, indexing="ij")
    out, info = (ctx or default_context()).logml_grid(X, y, np.full(R.size, float(alpha)), R.ravel(), S.ravel(), jitter)
    if lp:
        vals = np.array([stan_lp(o[1], o[2], float(alpha), r, s) if i == 0 else -math.inf
                         for o, r, s, i in zip(out, R.ravel(), S.ravel(), info)])
    else:
        vals = out[:, 0]
    return vals.reshape(R.shape)


def get_ml_from_grid(values, alpha, rho_vec, sigma_vec):
    """arg-max over the grid -> list(alpha=, rho=, sigma=): mirror of
    get_ml_from_stan_samples (R/tests.R:21-27) with grid points in place of posterior draws."""
    v = np.where(np.isfinite(values), values, -np.inf)
    i, j = np.unravel_index(int(np.argmax(v)), v.shape)
    return {"alpha": float(alpha), "rho": float(np.atleast_1d(rho_vec)[i]), "sigma": float(np.atleast_1d(sigma_vec)[j])}


def exact_gp_f(x, l, z, ctx=None):
    """f = L z with L = chol(cov_exp_quad(x, 1, l) + 1e-10 I) -- models/exact_gp.stan:17-25."""
    c = ctx or default_context()
    x = np.asarray(x, float).reshape(len(z), -1)
    return c.exact_gp_f(x, 1.0, [l], z, 1e-10)   # covariance, factor and product stay on the device (gpmi_exact_gp_f)
