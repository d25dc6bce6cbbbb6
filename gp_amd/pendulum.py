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
    L = chol(cov) and z standard normal (given, or from `rng`), fused on the GPU
    (gpmi_sample_derivs): the covariance is factored in place and never leaves the device."""
    c = ctx or default_context()
    l, a, sy = (float(p) for p in params[:3])
    ti = np.asarray(ti, float)
    tis = ti if tis is None else np.asarray(tis, float)
    if z is None:
        z = (rng or np.random.default_rng()).standard_normal(tis.size)
    return c.sample_derivs(ti, tis, ynoise, l, a, sy, jitter, z)[0]


def sample_derivs_many(params_list, ynoise_list, ti, tis=None, jitter=1e-8, Z=None, rng=None, ctx=None):
    """mclapply(s_list[1:100], sample_derivs_both_states, mc.cores = 2) of pendulum_fit.R:261-268: one
    draw per (params, ynoise) pair -- a posterior draw of (l, a, sy) and a noisy series each --, here as
    ONE call whose conditionings run concurrently on the GPU's lanes (gpmi_sample_derivs_batch).
    Returns the draws as the columns of an (m, B) array."""
    c = ctx or default_context()
    ti = np.asarray(ti, float)
    tis = ti if tis is None else np.asarray(tis, float)
    P = np.asarray([[float(p[0]), float(p[1]), float(p[2])] for p in params_list])
    Y = np.column_stack([np.asarray(y, float).ravel() for y in ynoise_list])
    if Z is None:
        Z = (rng or np.random.default_rng()).standard_normal((tis.size, P.shape[0]))
    draws, _, info = c.sample_derivs_batch(ti, tis, Y, P, jitter, Z)
    if np.any(info):
        from ._lib import NotPositiveDefinite
        raise NotPositiveDefinite(int(info[np.nonzero(info)[0][0]]))
    return draws
