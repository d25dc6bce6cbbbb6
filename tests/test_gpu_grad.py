"""GPU: gradient of the log marginal likelihood against the oracle's dense
restatement (itself pinned by finite differences in tests/test_oracle.py) and, for ARD length-scales,
against central differences of the GPU log marginal likelihood."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,D", [(7, 1), (50, 1), (129, 2), (300, 3), (700, 2), (200, 9), (333, 16), (150, 33), (70, 64)])
def test_grad_vs_oracle(ctx, orc, n, D):
    rng = np.random.default_rng(100 + n)
    X = rng.random((n, D)); y = np.sin(3 * X.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    a, r, s = 1.2, 0.4 * math.sqrt(D), 0.15    # (D > 8: the LDS-staged contraction, gpmi_api.hip k_grad_partial_big)
    out, g = ctx.logml_grad(X, y, a, [r], s)
    want_out, want_g, info = orc.logml_grad(X, y, a, r, s)
    assert info == 0
    assert abs(out[0] - want_out[0]) <= 1e-9 * abs(want_out[0])
    assert out[0] == ctx.logml(X, y, a, [r], s)[0]  # same factorisation as the plain entry point
    np.testing.assert_allclose(g, want_g, rtol=1e-8, atol=1e-8 * np.abs(want_g).max())


def test_grad_ard_vs_central_differences(ctx):
    rng = np.random.default_rng(7)
    n = 400
    X = rng.random((n, 3)); y = np.cos(2 * X[:, 0]) * X[:, 1] + 0.05 * rng.standard_normal(n)
    a, ell, s = 0.9, np.array([0.3, 0.6, 1.1]), 0.2
    _, g = ctx.logml_grad(X, y, a, ell, s)
    assert g.shape == (5,)
    h = 1e-5
    f = lambda a_, e_, s_: ctx.logml(X, y, a_, e_, s_)[0]
    fd = [(f(a + h, ell, s) - f(a - h, ell, s)) / (2 * h)]
    for d in range(3):
        e = np.zeros(3); e[d] = h
        fd.append((f(a, ell + e, s) - f(a, ell - e, s)) / (2 * h))
    fd.append((f(a, ell, s + h) - f(a, ell, s - h)) / (2 * h))
    np.testing.assert_allclose(g, fd, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("D", [9, 20])
def test_grad_ard_many_dimensions_vs_central_differences(ctx, D):
    rng = np.random.default_rng(D)
    n = 300
    X = rng.random((n, D)); y = np.cos(2 * X[:, 0]) * X[:, 1] + X[:, D - 1] + 0.05 * rng.standard_normal(n)
    a, ell, s = 0.9, 0.8 + 1.5 * rng.random(D), 0.2
    _, g = ctx.logml_grad(X, y, a, ell, s)
    assert g.shape == (D + 2,)
    h = 1e-5
    f = lambda a_, e_, s_: ctx.logml(X, y, a_, e_, s_)[0]
    fd = [(f(a + h, ell, s) - f(a - h, ell, s)) / (2 * h)]
    for d in range(D):
        e = np.zeros(D); e[d] = h
        fd.append((f(a, ell + e, s) - f(a, ell - e, s)) / (2 * h))
    fd.append((f(a, ell, s + h) - f(a, ell, s - h)) / (2 * h))
    np.testing.assert_allclose(g, fd, rtol=5e-6, atol=2e-6)


def test_grad_not_positive_definite_and_bad_arguments(ctx):
    import gp_amd
    X = np.zeros((20, 1)); y = np.ones(20)  # identical points, no noise: singular
    with pytest.raises(gp_amd.NotPositiveDefinite):
        ctx.logml_grad(X, y, 1.0, [0.5], 0.0)
    with pytest.raises(gp_amd.GpmiError):
        ctx.logml_grad(np.zeros((5, 65)), np.zeros(5), 1.0, [0.5], 0.1)  # D > 64
    out, g = ctx.logml_grad(np.linspace(0, 1, 30), np.ones(30), 1.0, [0.3], 0.1)  # healthy afterwards
    assert math.isfinite(out[0]) and np.all(np.isfinite(g))


def test_stan_lp_gradient_mirror(ctx):
    from gp_amd.stan_models import fit_hyperparameters_log_prob, fit_hyperparameters_log_prob_grad
    t = np.linspace(-2, 2, 21); y = np.exp(t)  # the R/tests.R:5 grid
    rho, alpha, sigma = 0.9, 1.1, 0.07
    lp, g = fit_hyperparameters_log_prob_grad(t, y, rho, alpha, sigma, ctx=ctx)
    assert lp == pytest.approx(fit_hyperparameters_log_prob(t, y, rho, alpha, sigma, ctx=ctx), rel=1e-13)
    h = 1e-6
    f = lambda r, a, s: fit_hyperparameters_log_prob(t, y, r, a, s, ctx=ctx)
    fd = [(f(rho + h, alpha, sigma) - f(rho - h, alpha, sigma)) / (2 * h),
          (f(rho, alpha + h, sigma) - f(rho, alpha - h, sigma)) / (2 * h),
          (f(rho, alpha, sigma + h) - f(rho, alpha, sigma - h)) / (2 * h)]
    np.testing.assert_allclose(g, fd, rtol=1e-5, atol=1e-5)


def test_grad_fullsize_n8192_property(ctx, orc):
    # size-independent check at a BASELINE size: the directional derivative along a random direction in
    # (alpha, rho, sigma) equals the central difference of the log marginal likelihood
    X, y = orc.synth(8192, 3)
    a, r, s = 1.0, 0.3, 0.1
    out, g = ctx.logml_grad(X, y, a, [r], s)
    d = np.array([0.3, -0.5, 0.8]); d /= np.linalg.norm(d)
    h = 1e-5
    f = lambda q: ctx.logml(X, y, a + q * d[0], [r + q * d[1]], s + q * d[2])[0]
    fd = (f(h) - f(-h)) / (2 * h)
    assert g @ d == pytest.approx(fd, rel=5e-6)


def test_grad_grid_on_lanes_equals_single_calls(ctx, orc):
    """gpmi_logml_grad_grid (value + gradient of several points at once: what rstan's four chains ask for per
    leapfrog step) equals gpmi_logml_grad point by point, bit for bit, for G not a multiple of the lane count, with
    a non-PD point in the middle; and the Stan-lp wrapper adds the prior / Jacobian terms."""
    from gp_amd import stan_models as sm
    X, y = orc.synth(900, 3)
    a = np.array([1.0, 1.1, 0.9, 1.0, 1.2]); r = np.array([0.3, 0.35, 0.25, 50.0, 0.3]); s = np.array([0.1, 0.12, 0.2, 1e-9, 0.15])
    out, g, info = ctx.logml_grad_grid(X, y, a, r, s)
    assert info[3] > 0 and np.all(np.isnan(g[3])) and np.all(np.delete(info, 3) == 0)
    for k in (0, 1, 2, 4):
        o1, g1 = ctx.logml_grad(X, y, a[k], [r[k]], s[k])
        assert np.array_equal(out[k], o1) and np.array_equal(g[k], g1)
    wo, wg, winfo = orc.logml_grad(X, y, a[1], r[1], s[1])
    np.testing.assert_allclose(g[1], wg, rtol=1e-8, atol=1e-8 * np.abs(wg).max())
    lp, grad = sm.fit_hyperparameters_log_prob_grad_chains(X, y, r, a, s, ctx=ctx)
    assert lp[3] == -math.inf and np.all(np.isnan(grad[3]))
    lp1, grad1 = sm.fit_hyperparameters_log_prob_grad(X, y, r[2], a[2], s[2], ctx=ctx)
    assert lp[2] == lp1 and np.array_equal(grad[2], grad1)
