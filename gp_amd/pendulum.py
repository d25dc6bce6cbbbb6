"""Derivative imputation: mirror of sample_derivs, pendulum_fit.R:227-255 (and the
separate-prediction-times variant of lorenz.Rmd:80-107)."""
import numpy as np

from ._lib import default_context


def sample_derivs_moments(params, ynoise, ti, tis=None, jitter=1e-8, ctx=None):
    """(mu, cov) with K = a^2 QQ, KsK = a^2 RQ, KsKs = a^2 RR (derivative_kernels.R through
    outer(), pendulum_fit.R:237-240); mu = KsK (K + sy^2 I)^-1 y (:242-245);
    cov = KsKs - KsK (K + sy^2 I)^-1 t(KsK) + jitter I (:247-251; 1e-6 in lorenz.Rmd:101-102)."""
    l, a, sy = (float(p) for p in params[:3])
    ti = np.asarray(ti, float)
    tis = ti if tis is None else np.asarray(tis, float)
    return (ctx or default_context()).gp_condition(ti, tis, ynoise, a, l, sy * sy, jitter, "QQ", "RQ", "RR")


def sample_derivs(params, ynoise, ti, tis=None, jitter=1e-8, z=None, rng=None, ctx=None):
    """One draw of the derivative process.  The reference draws with MASS::mvrnorm
    (eigen-decomposition, R's unseeded RNG, :253); here the draw is mu + L z with
    L = chol(cov) on the GPU and z standard normal (given, or from `rng`)."""
    c = ctx or default_context()
    mu, cov = sample_derivs_moments(params, ynoise, ti, tis, jitter, c)
    if z is None:
        z = (rng or np.random.default_rng()).standard_normal(mu.size)
    return mu + c.trmv_lower(c.potrf(cov), z)
