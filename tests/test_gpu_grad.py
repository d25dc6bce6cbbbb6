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
    plain = ctx.logml(X, y, a, [r], s)[0]
    if n <= 128:
        assert out[0] == plain      # the same one-workgroup factorisation as the plain entry point
    else:                           # mid sizes: the augmented (2n + 1)-row factorisation picks other block widths: to rounding
        assert abs(out[0] - plain) <= 1e-12 * abs(plain)
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


@pytest.mark.parametrize("D", [1, 3, 8])
def test_small_value_and_gradient_by_one_workgroup(ctx, orc, D):
    """n <= 256 (the sizes the reference's Stan fits run at: R/tests.R:5 N = 21, pendulum_fit.R 79 .. 199): value and
    gradient in ONE launch of one workgroup (k_logml_grad_small) against the oracle, against the launch chain
    (small_ng1 = 0) and, for ARD, against the oracle on inputs scaled per dimension."""
    worst = worst_chain = 0.0
    ctx.set_option("small_ng1", 256)   # default 128: a SINGLE evaluation beyond one panel is faster through the launch chain
    for n in (1, 2, 5, 16, 17, 21, 64, 79, 127, 128, 129, 160, 199, 255, 256):
        rng = np.random.default_rng(10 * n + D)
        X = rng.random((n, D)) * (1.0 + n / 60.0); y = np.sin(3 * X.sum(axis=1)) + 0.1 * rng.standard_normal(n)
        a, r, s = 1.2, 0.4 * math.sqrt(D), 0.15
        out, g = ctx.logml_grad(X, y, a, [r], s)
        want_out, want_g, info = orc.logml_grad(X, y, a, r, s)
        assert info == 0 and abs(out[0] - want_out[0]) <= 1e-9 * abs(want_out[0]), (n, out, want_out)
        np.testing.assert_allclose(g, want_g, rtol=1e-8, atol=1e-8 * np.abs(want_g).max(), err_msg="n=%d" % n)
        worst = max(worst, np.max(np.abs(g - want_g)) / np.abs(want_g).max())
        if n in (21, 128, 199, 256):
            ctx.set_option("small_ng1", 0)
            try:
                out_c, g_c = ctx.logml_grad(X, y, a, [r], s)
            finally:
                ctx.set_option("small_ng1", 256)
            assert abs(out[0] - out_c[0]) <= 1e-12 * abs(out_c[0])
            np.testing.assert_allclose(g, g_c, rtol=1e-9, atol=1e-9 * np.abs(g_c).max())
            worst_chain = max(worst_chain, np.max(np.abs(g - g_c)) / np.abs(g_c).max())
    print("one-workgroup gradient, D=%d: worst vs oracle %.1e, vs the launch chain %.1e" % (D, worst, worst_chain))
    ctx.set_option("small_ng1", 128)
    if D > 1:   # ARD: d/d ell_d against central differences of the value
        n = 150
        rng = np.random.default_rng(D)
        X = rng.random((n, D)); y = np.cos(2 * X[:, 0]) + 0.05 * rng.standard_normal(n)
        ell = 0.5 + rng.random(D)
        _, g = ctx.logml_grad(X, y, 0.9, ell, 0.2)
        h = 1e-5
        f = lambda a_, e_, s_: ctx.logml(X, y, a_, e_, s_)[0]
        fd = [(f(0.9 + h, ell, 0.2) - f(0.9 - h, ell, 0.2)) / (2 * h)]
        for d in range(D):
            e = np.zeros(D); e[d] = h
            fd.append((f(0.9, ell + e, 0.2) - f(0.9, ell - e, 0.2)) / (2 * h))
        fd.append((f(0.9, ell, 0.2 + h) - f(0.9, ell, 0.2 - h)) / (2 * h))
        np.testing.assert_allclose(g, fd, rtol=5e-6, atol=2e-6)


def test_small_gradient_chains_in_one_launch(ctx, orc):
    """gpmi_logml_grad_grid at n <= 256: the chains are workgroups of ONE launch (k_logml_grad_small_batch); each equals
    the single call bit for bit, a rejected (non-PD) proposal gives NaN + info and does not disturb the others; more
    points than one launch carries (128) are cut into launches."""
    t = np.linspace(-2, 2, 21); y = np.exp(t)          # the R/tests.R:5 grid
    a = np.array([1.0, 1.1, 0.9, 1.0]); r = np.array([0.9, 1.0, 50.0, 0.8]); s = np.array([0.07, 0.1, 1e-9, 0.2])
    out, g, info = ctx.logml_grad_grid(t, y, a, r, s)
    assert info[2] > 0 and np.all(np.isnan(g[2])) and np.all(np.isnan(out[2])) and np.all(np.delete(info, 2) == 0)
    for k in (0, 1, 3):
        o1, g1 = ctx.logml_grad(t, y, a[k], [r[k]], s[k])
        assert np.array_equal(out[k], o1) and np.array_equal(g[k], g1)
        wo, wg, _ = orc.logml_grad(t.reshape(-1, 1), y, a[k], r[k], s[k])
        np.testing.assert_allclose(g[k], wg, rtol=1e-8, atol=1e-8 * np.abs(wg).max())
    X, yy = orc.synth(199, 1)
    G = 150
    rng = np.random.default_rng(4)
    a = 0.8 + 0.4 * rng.random(G); r = 0.2 + 0.3 * rng.random(G); s = 0.05 + 0.2 * rng.random(G)
    out, g, info = ctx.logml_grad_grid(X, yy, a, r, s)
    assert np.all(info == 0)
    ctx.set_option("small_ng1", 256)   # the single call through the same kernel: bit for bit
    try:
        for k in (0, 127, 128, 149):
            o1, g1 = ctx.logml_grad(X, yy, a[k], [r[k]], s[k])
            assert np.array_equal(out[k], o1) and np.array_equal(g[k], g1)
    finally:
        ctx.set_option("small_ng1", 128)
    o1, g1 = ctx.logml_grad(X, yy, a[5], [r[5]], s[5])   # default: the launch chain at n = 199 -- same numbers to rounding
    assert abs(out[5, 0] - o1[0]) <= 1e-12 * abs(o1[0])
    np.testing.assert_allclose(g[5], g1, rtol=1e-9, atol=1e-9 * np.abs(g1).max())


@pytest.mark.parametrize("n,D", [(257, 1), (1000, 3), (1438, 1), (2500, 2)])
def test_mid_size_gradient_from_one_augmented_factorisation(ctx, orc, n, D):
    """128 < n <= 3072: K^-1 and K^-1 y come out of ONE partial factorisation of the (2n + 1)-row matrix [[K], [y'], [I]]
    (trailing block = -[[z'z, a'], [a, K^-1]]) instead of two more launch chains (N = 1438, the reference's westbrook.R size:
    1146 -> 606 us): against the three-chain form (grad_aug_n = 0) and, at the smaller sizes, the oracle."""
    X, y = orc.synth(n, D, seed=n)
    a, r, s = 1.1, 0.35 * math.sqrt(D), 0.12
    out, g = ctx.logml_grad(X, y, a, [r], s)
    ctx.set_option("grad_aug_n", 0)
    try:
        out3, g3 = ctx.logml_grad(X, y, a, [r], s)
    finally:
        ctx.set_option("grad_aug_n", 3072)
    assert abs(out[0] - out3[0]) <= 1e-12 * abs(out3[0])
    np.testing.assert_allclose(g, g3, rtol=1e-8, atol=1e-8 * np.abs(g3).max())
    if n <= 1000:
        wo, wg, info = orc.logml_grad(X, y, a, r, s)
        assert info == 0
        np.testing.assert_allclose(g, wg, rtol=1e-8, atol=1e-8 * np.abs(wg).max())
    from gp_amd import NotPositiveDefinite
    if n == 257:
        with pytest.raises(NotPositiveDefinite):
            ctx.logml_grad(np.zeros((n, 1)), np.ones(n), 1.0, [0.5], 0.0)
