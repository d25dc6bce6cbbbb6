"""GPU: parity of the HIP path (through the C ABI) with the CPU oracle and the golden
fixtures.  fp64 floating point: tolerances are written per test; the headline bar is
|logml_gpu - logml_oracle| <= 1e-8 |logml_oracle| (BASELINE.json north_star)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ["QQ", "QR", "RQ", "RR", "QT", "TQ", "RT", "TR", "TT"]
LOGML_RTOL = 1e-8


def test_library_loaded_and_device_is_gfx950(ctx):
    import gp_amd
    assert gp_amd.device_count() >= 1


_PROBE_SCRIPT = r"""
import numpy as np, gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
# D = A B with ASYMMETRIC integer operands: catches swapped row/col maps exactly
rng = np.random.default_rng(0)
A = rng.integers(-8, 9, size=(16, 4)).astype(float)
B = rng.integers(-8, 9, size=(4, 16)).astype(float) + np.arange(16)[None, :] * 3
assert np.array_equal(ctx.probe_mfma(A, B), A @ B)
# the A/B kernel variants are re-orderings of the same factorisation
for n in (900, 2048):  # 2048: a multiple of the tile size, where the augmented row makes the tile grid a trapezoid
    X, y = synth(n, 3)
    base = ctx.logml(X, y, 1.0, [0.3], 0.1)[0]
    for opt, v, back in (("gemm_variant", 0, 3), ("gemm_variant", 1, 3), ("gemm_variant", 2, 3), ("lookahead", 1, 0),
                         ("syrk_order", 1, 2), ("syrk_order", 0, 2), ("diag_waves", 5, 4)):
        ctx.set_option(opt, v)
        got = ctx.logml(X, y, 1.0, [0.3], 0.1)[0]
        ctx.set_option(opt, back)
        assert abs(got - base) <= 1e-10 * abs(base), (n, opt, v, got, base)
print("probe build ok", base)
"""


def test_probe_build_fragment_layout_and_variants(ctx):
    # the probe library (tools/ only) in a process of its own: MFMA f64 fragment layout and the A/B kernels
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPMI_USE_PROBES="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _PROBE_SCRIPT], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0 and "probe build ok" in r.stdout, r.stdout + r.stderr
    # ... and the product library has neither the entry points nor the option names
    import gp_amd
    with pytest.raises(gp_amd.GpmiError):
        ctx.probe_mfma(np.zeros((16, 4)), np.zeros((4, 16)))
    for name in ("gemm_variant", "lookahead", "syrk_order", "diag_waves", "lane_lookahead"):
        with pytest.raises(gp_amd.GpmiError):
            ctx.set_option(name, 1)


@pytest.mark.parametrize("n,m,D", [(1, 1, 1), (21, 21, 1), (64, 64, 3), (65, 130, 3), (200, 77, 2), (257, 300, 5), (130, 129, 8), (70, 33, 9), (40, 50, 20),
                                   (100, 70, 64), (300, 200, 33), (129, 193, 17), (64, 64, 16), (65, 1, 48)])
def test_se_cov_rect(ctx, orc, n, m, D):
    rng = np.random.default_rng(n * 1000 + m)
    X = rng.random((n, D)) * 3; Y = rng.random((m, D)) * 3
    for ell in ([0.7], list(0.3 + 0.2 * np.arange(D))):
        K = ctx.se_cov(X, Y, 1.3, ell)
        want = orc.QQard(X, Y, 1.3, ell)
        # <= 4 ulp of alpha^2 absolute: inv-length-scale multiply vs R's divide
        assert np.max(np.abs(K - want)) <= 4 * np.finfo(float).eps * 1.3 ** 2


def test_se_cov_symmetric_lower_and_diag(ctx, orc):
    rng = np.random.default_rng(5)
    X = rng.random((150, 3))
    K = ctx.se_cov(X, None, 0.9, [0.4], diag_add=0.01)
    want = orc.cov_exp_quad(X, 0.9, 0.4) + 0.01 * np.eye(150)
    assert np.max(np.abs(K - want)) <= 4 * np.finfo(float).eps
    assert np.all(np.diag(K) == 0.9 * 0.9 + 0.01)  # Stan: exact diagonal
    assert np.array_equal(K, K.T)
    from gp_amd._lib import LOWER
    Kl = ctx.se_cov(X, None, 0.9, [0.4], diag_add=0.01, flags=LOWER)
    assert np.array_equal(np.tril(Kl), np.tril(K)) and np.all(np.triu(Kl, 1) == 0)


def test_se_cov_empty_and_bad_args(ctx):
    import gp_amd
    assert ctx.se_cov(np.zeros((0, 2)), np.zeros((0, 2)), 1.0, [1.0]).shape == (0, 0)
    with pytest.raises(gp_amd.GpmiError):
        ctx.se_cov(np.zeros((3, 2)), np.zeros((3, 2)), 1.0, [1.0, 2.0, 3.0])
    with pytest.raises(gp_amd.GpmiError):
        ctx.se_cov(np.zeros((3, 2)), None, 1.0, [-1.0])
    with pytest.raises(gp_amd.GpmiError):
        ctx.se_cov(np.zeros((3, 65)), None, 1.0, [1.0])  # D > 64 not supported (documented)


def test_deriv_kernels_elementwise_golden(ctx, golden):
    # the reference's own gp_derivs.py outputs (a^2 folded in)
    from gp_amd import derivative_kernels as dk
    for c in golden["gp_derivs"]["kernel_cases"]:
        for k in KINDS:
            got = c["a"] ** 2 * float(getattr(dk, k)(c["tj"], c["tk"], c["l"], ctx=ctx))
            assert got == pytest.approx(c["out"][k], rel=1e-14, abs=1e-300), (k, c)


@pytest.mark.parametrize("kind", KINDS)
def test_deriv_cov_vs_oracle(ctx, orc, kind):
    x = np.linspace(-2, 3, 131); y = np.linspace(-1, 4, 70)
    K = ctx.deriv_cov(kind, x, y, 1.2, 0.6)
    want = orc.deriv_cov(kind, x, y, 1.2, 0.6)
    scale = np.max(np.abs(want))
    assert np.max(np.abs(K - want)) <= 1e-14 * scale
    # vectorised elementwise API with broadcasting == outer()
    E = ctx.deriv_elem(kind, x[:, None], y[None, :], 0.6)
    assert np.max(np.abs(1.2 ** 2 * E - want)) <= 1e-14 * scale


def test_kernels_R_matrix_api(ctx, orc):
    from gp_amd import kernels
    x = np.linspace(-2, 2, 21); y = np.linspace(0, 1, 8)
    phi = [1.3, 0.8]
    np.testing.assert_allclose(kernels.QQ(x, y, phi, ctx=ctx), orc.QQ(x, y, 1.3, 0.8), rtol=1e-14)
    np.testing.assert_allclose(kernels.QR(x, y, phi, ctx=ctx), orc.QR(x, y, 1.3, 0.8), rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(kernels.RR(x, y, phi, ctx=ctx), orc.RR(x, y, 1.3, 0.8), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(kernels.RR(x, y, phi, compat=True, ctx=ctx), orc.RR(x, y, 1.3, 0.8, compat=True), rtol=1e-12, atol=1e-15)
    X = np.random.default_rng(1).random((30, 3)); Y = np.random.default_rng(2).random((12, 3))
    np.testing.assert_allclose(kernels.QQard(X, Y, {"a": 1.1, "l": [0.5, 1.0, 2.0]}, ctx=ctx),
                               orc.QQard(X, Y, 1.1, [0.5, 1.0, 2.0]), rtol=1e-14, atol=1e-16)


@pytest.mark.parametrize("n", [1, 7, 64, 100, 129])
def test_joint_cov(ctx, orc, n):
    from gp_amd._lib import LOWER, COMPAT_RR
    t = np.linspace(0, 3, n) if n > 1 else np.array([0.5])
    K = ctx.joint_cov(t, 1.1, 0.7, 0.1, 1e-6)
    want = orc.joint_cov(t, 1.1, 0.7, 0.1, 1e-6)
    assert np.max(np.abs(K - want)) <= 1e-14 * np.max(np.abs(want))
    Kl = ctx.joint_cov(t, 1.1, 0.7, 0.1, 1e-6, flags=LOWER)
    assert np.array_equal(np.tril(Kl), np.tril(K)) and np.all(np.triu(Kl, 1) == 0)
    Kc = ctx.joint_cov(t, 1.1, 0.7, 0.1, 1e-6, flags=COMPAT_RR)
    assert np.max(np.abs(Kc - orc.joint_cov(t, 1.1, 0.7, 0.1, 1e-6, compat=True))) <= 1e-14 * np.max(np.abs(want))


@pytest.mark.parametrize("n", [1, 2, 5, 16, 17, 21, 100, 128, 129, 199, 256, 257, 300, 511, 640, 1000])
def test_potrf_vs_oracle(ctx, orc, n):
    rng = np.random.default_rng(n)
    X = rng.random((n, 2))
    A = orc.cov_exp_quad(X, 1.0, 0.3) + 0.05 * np.eye(n)
    L = ctx.potrf(A)
    want = orc.cholesky(A, blocked=n > 300)
    assert np.all(np.triu(L, 1) == 0)
    assert np.max(np.abs(L - want)) <= 1e-11
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13 * math.sqrt(n) + 1e-15


@pytest.mark.parametrize("nbo", [128, 256, 384, 512])
def test_potrf_outer_block_sizes(ctx, orc, nbo):
    rng = np.random.default_rng(99)
    n = 700
    X = rng.random((n, 3))
    A = orc.cov_exp_quad(X, 1.0, 0.4) + 0.1 * np.eye(n)
    ctx.set_option("nb_outer", nbo)
    try:
        L = ctx.potrf(A)
    finally:
        ctx.set_option("nb_outer", 0)
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13


def test_potrf_not_positive_definite(ctx):
    import gp_amd
    A = np.eye(300); A[140, 140] = -1.0
    with pytest.raises(gp_amd.NotPositiveDefinite) as e:
        ctx.potrf(A)
    assert e.value.order == 141  # LAPACK-style: leading minor of order 141
    B = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(gp_amd.NotPositiveDefinite) as e:
        ctx.potrf(B)
    assert e.value.order == 2
    # the context stays usable afterwards
    assert ctx.potrf(np.eye(3))[2, 2] == 1.0


def test_trmv_trsv(ctx, orc):
    rng = np.random.default_rng(3)
    for n in (1, 17, 200, 300):
        A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
        L = np.linalg.cholesky(A); b = rng.standard_normal(n)
        np.testing.assert_allclose(ctx.trmv_lower(L, b), orc.trmv_lower(L, b), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ctx.trsv_lower(L, b), orc.trsv_lower(L, b), rtol=1e-10, atol=1e-12)


def test_trsv_wavefront_many_blocks(ctx):
    """The one-launch forward substitution (k_trsv_wave: workgroup w waits for the unknowns of the block-rows
    above it) over 8 ... 32 block-rows, ragged last block included, against LAPACK's dtrtrs; a NaN in the
    right-hand side propagates (and must not be mistaken for the kernel's "not yet published" bit pattern)."""
    import scipy.linalg as sla
    rng = np.random.default_rng(11)
    for n in (1000, 2049, 4096):
        G = rng.standard_normal((n, n)) / np.sqrt(n)
        L = np.linalg.cholesky(G @ G.T + np.eye(n)); b = rng.standard_normal(n)
        want = sla.solve_triangular(L, b, lower=True)
        got = ctx.trsv_lower(L, b)
        err = np.max(np.abs(got - want)) / np.max(np.abs(want))
        print("trsv n=%d: max rel err %.2e" % (n, err))
        assert err <= 1e-12
    b[700] = np.nan
    got = ctx.trsv_lower(L, b)
    assert np.all(np.isfinite(got[:700])) and np.all(np.isnan(got[700:]))


def test_logml_known_answers(ctx, golden):
    for k in golden["kat"]["kats"]:
        x = np.array(k["x"]); y = np.array(k["y"])
        lm, sld, q = ctx.logml(x.reshape(len(y), -1), y, k["alpha"], [k["rho"]], k["sigma"])
        assert abs(lm - k["logml"]) <= LOGML_RTOL * abs(k["logml"]), k["name"]
        assert sld == pytest.approx(k["sum_log_diag"], rel=1e-10)
        assert q == pytest.approx(k["quad"], rel=1e-7)


@pytest.mark.parametrize("n,D", [(1, 1), (2, 1), (21, 1), (127, 2), (128, 3), (129, 3), (255, 1), (256, 1), (300, 3), (777, 3), (1500, 3)])
def test_logml_vs_oracle(ctx, orc, n, D):
    X, y = orc.synth(n, D, seed=n)
    got = ctx.logml(X, y, 1.0, [0.3], 0.1)
    want = orc.logml(X, y, 1.0, 0.3, 0.1)
    assert want[3] == 0
    assert abs(got[0] - want[0]) <= LOGML_RTOL * abs(want[0])
    assert abs(got[1] - want[1]) <= 1e-9 * max(1.0, abs(want[1]))
    assert abs(got[2] - want[2]) <= 1e-8 * abs(want[2])


def test_logml_jitter_only_regime_small_n(ctx, orc):
    # exact_gp.stan:21-style 1e-10 jitter, no noise: ill-conditioned, parity only at small N
    x = np.linspace(0, 10, 40); y = np.sin(x)
    got = ctx.logml(x.reshape(-1, 1), y, 1.0, [0.5], 0.0, jitter=1e-10)
    want = orc.logml(x, y, 1.0, 0.5, 0.0, jitter=1e-10)
    assert want[3] == 0 and abs(got[0] - want[0]) <= 1e-5 * abs(want[0])


def test_logml_grid_and_not_pd_point(ctx, orc):
    X, y = orc.synth(200, 3)
    rho = np.array([0.1, 0.3, 1.0, 50.0, 0.5]); sig = np.array([0.05, 0.1, 0.5, 1e-9, 0.2])
    out, info = ctx.logml_grid(X, y, 1.0, rho, sig)
    for g in range(5):
        w = orc.logml(X, y, 1.0, rho[g], sig[g])
        if w[3]:
            assert info[g] > 0 and np.all(np.isnan(out[g]))
        else:
            assert info[g] == 0 and abs(out[g, 0] - w[0]) <= LOGML_RTOL * abs(w[0])
    assert info[3] > 0  # the grid continued past the failing point
    # a grid evaluation is bit-identical to the single evaluation (stateless, deterministic)
    single = ctx.logml(X, y, 1.0, [rho[1]], sig[1])
    assert single[0] == out[1, 0]


def test_stan_models(ctx, orc):
    from gp_amd import stan_models as sm
    t = np.round(np.arange(-2.0, 2.0 + 1e-9, 0.2), 10); y = np.exp(t)  # R/tests.R:5 grid
    lp = sm.fit_hyperparameters_log_prob(t, y, rho=1.0, alpha=1.0, sigma=0.05, ctx=ctx)
    w = orc.logml(t, y, 1.0, 1.0, 0.05)
    assert lp == pytest.approx(orc.stan_lp(w[1], w[2], 1.0, 1.0, 0.05), rel=1e-9)
    assert sm.fit_hyperparameters_log_prob(t, y, rho=50.0, alpha=1.0, sigma=1e-9, ctx=ctx) == -math.inf
    G = sm.gp_log_marginal_grid(t, y, 1.0, [0.5, 1.0, 2.0], [0.05, 0.1], ctx=ctx)
    assert G.shape == (3, 2)
    for i, r in enumerate([0.5, 1.0, 2.0]):
        for j, s in enumerate([0.05, 0.1]):
            assert G[i, j] == pytest.approx(orc.logml(t, y, 1.0, r, s)[0], rel=1e-8)
    best = sm.get_ml_from_grid(G, 1.0, [0.5, 1.0, 2.0], [0.05, 0.1])
    i, j = np.unravel_index(np.argmax(G), G.shape)
    assert best["rho"] == [0.5, 1.0, 2.0][i] and best["sigma"] == [0.05, 0.1][j]
    # exact_gp.stan: f = L z
    x = np.linspace(0, 10, 30); z = np.random.default_rng(0).standard_normal(30)
    f = sm.exact_gp_f(x, 0.4, z, ctx=ctx)
    Lw = orc.cholesky(orc.cov_exp_quad(x, 1.0, 0.4) + 1e-10 * np.eye(30))
    np.testing.assert_allclose(f, Lw @ z, rtol=0, atol=1e-6)  # 1e-10 jitter: cond ~1e10


@pytest.mark.parametrize("n", [13, 150, 260])
def test_rbf_cov_chol(ctx, orc, n):
    from gp_amd.covariance import rbf_cov_chol
    x = np.linspace(0, 0.5 * n, n); l = 0.45  # well-conditioned spacing
    r = rbf_cov_chol(x, l, ctx=ctx)
    L, dL = orc.rbf_cov_chol(x, l)
    assert set(r) == {"L", "dLdl"}
    np.testing.assert_allclose(r["L"], L, rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["dLdl"], dL, rtol=0, atol=1e-10)
    assert np.all(np.triu(r["L"], 1) == 0) and np.all(np.triu(r["dLdl"], 1) == 0)


def test_gp_posteriors(ctx, orc, golden):
    from gp_amd import ode_gp, pendulum
    t = np.round(np.arange(-2.0, 2.0 + 1e-9, 0.2), 10); f = np.exp(t)
    phi = [1.1, 0.9]; s = 0.05
    p = ode_gp.p_Xn(t, f, phi, s, ctx=ctx)
    mn, Kn = orc.p_Xn(t, f, 1.1, 0.9, s)
    assert p["mn"].shape == (21, 1)
    np.testing.assert_allclose(p["mn"][:, 0], mn, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(p["Kn"], Kn, rtol=0, atol=1e-8)
    p = ode_gp.p_dotXn(t, f, phi, s, ctx=ctx)
    mn, Kn = orc.p_dotXn(t, f, 1.1, 0.9, s)
    np.testing.assert_allclose(p["condMean"], mn, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(p["condVar"], Kn, rtol=0, atol=1e-8)
    p = ode_gp.p_dotXn(t, f, phi, s, joint=True, ctx=ctx)
    cm, cv = orc.p_dotXn_joint(t, f, 1.1, 0.9, s, 1e-6)
    np.testing.assert_allclose(p["condMean"], cm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(p["condVar"], cv, rtol=0, atol=1e-8)
    p = ode_gp.p_dotXn(t, f, phi, s, compat=True, ctx=ctx)
    mn, Kn = orc.p_dotXn(t, f, 1.1, 0.9, s, compat=True)
    np.testing.assert_allclose(p["Kn"], Kn, rtol=0, atol=1e-8)
    # sample_derivs moments: reference python posterior (golden) and the oracle
    g = golden["gp_derivs"]["posterior"]
    ts = np.array(g["ts"]); y = np.array(g["y"])
    mu, cov = pendulum.sample_derivs_moments([g["l"], g["a"], g["s"]], y, ts, jitter=0.0, ctx=ctx)
    np.testing.assert_allclose(mu, np.array(g["mu_deriv"]), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(cov, np.array(g["cov_deriv"]), rtol=0, atol=1e-8)
    mu8, cov8 = pendulum.sample_derivs_moments([0.7, 1.2, 0.1], y, ts, ctx=ctx)
    wmu, wcov = orc.sample_derivs_moments(ts, y, 0.7, 1.2, 0.1)
    np.testing.assert_allclose(mu8, wmu, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(cov8, wcov, rtol=0, atol=1e-8)
    # separate prediction times (lorenz.Rmd:80-107) and a draw with a fixed z
    tis = np.linspace(0.1, 4.9, 11)
    mu_s, cov_s = pendulum.sample_derivs_moments([0.7, 1.2, 0.1], y, ts, tis=tis, jitter=1e-6, ctx=ctx)
    K = orc.deriv_cov("QQ", ts, ts, 1.2, 0.7) + 0.01 * np.eye(ts.size)
    KsK = orc.deriv_cov("RQ", tis, ts, 1.2, 0.7); KsKs = orc.deriv_cov("RR", tis, tis, 1.2, 0.7)
    np.testing.assert_allclose(mu_s, KsK @ np.linalg.solve(K, y), rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(cov_s, KsKs - KsK @ np.linalg.solve(K, KsK.T) + 1e-6 * np.eye(11), rtol=0, atol=1e-8)
    z = np.random.default_rng(4).standard_normal(11)
    d = pendulum.sample_derivs([0.7, 1.2, 0.1], y, ts, tis=tis, jitter=1e-6, z=z, ctx=ctx)
    np.testing.assert_allclose(d, mu_s + np.linalg.cholesky(cov_s) @ z, rtol=0, atol=1e-7)


def test_joint_logml(ctx, orc):
    for n in (21, 200):
        t = np.linspace(0, 10, n); yy = np.concatenate([np.sin(t), np.cos(t)])
        got = ctx.joint_logml(t, yy, 1.0, 0.5, 0.1, 1e-6)
        want = orc.joint_logml(t, yy, 1.0, 0.5, 0.1, 1e-6)
        assert want[3] == 0 and abs(got[0] - want[0]) <= LOGML_RTOL * abs(want[0])


def test_joint_logml_grid_on_lanes(ctx, orc):
    import torch
    n = 700
    t = np.linspace(0, 10, n); yy = np.concatenate([np.sin(t), np.cos(t)])
    dev = torch.device("cuda:0")
    dt = torch.from_numpy(t).to(dev); dyy = torch.from_numpy(yy).to(dev)
    G = 6
    ls = 0.5 * (1 + 0.05 * np.arange(G)); sg = 0.1 * np.ones(G)
    out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
    ctx.joint_logml_grid_dev(dt.data_ptr(), n, dyy.data_ptr(), np.ones(G), ls, sg, 1e-6, out.data_ptr(), info.data_ptr())
    ctx.sync()
    o = out.cpu().numpy()
    assert np.all(info.cpu().numpy() == 0)
    for g in (0, 3, 5):
        assert o[g, 0] == ctx.joint_logml(t, yy, 1.0, ls[g], sg[g], 1e-6)[0]      # lanes == single call, bit for bit
    want = orc.joint_logml(t, yy, 1.0, ls[4], sg[4], 1e-6)
    assert abs(o[4, 0] - want[0]) <= LOGML_RTOL * abs(want[0])


def test_device_pointer_api_with_torch(ctx, orc):
    import torch
    X, y = orc.synth(500, 3)
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(np.asfortranarray(X).T.copy()).to(dev)  # (D, n) row-major == n x D column-major
    dy = torch.from_numpy(y).to(dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            ctx.logml_dev(dX.data_ptr(), 500, 500, 3, dy.data_ptr(), 1.0, [0.3], 0.1, 0.0, out.data_ptr(), info.data_ptr())
        stream.synchronize()
    finally:
        ctx.set_stream(None)
    want = orc.logml(X, y, 1.0, 0.3, 0.1)
    assert int(info.item()) == 0 and abs(out[0].item() - want[0]) <= LOGML_RTOL * abs(want[0])
    # torch's DEFAULT stream has handle 0 (HIP's legacy null stream): a torch op issued right after a
    # _dev call must see its result without any synchronisation in between -- single evaluation and
    # the multi-lane grid (lanes fork from / join into the caller's stream)
    cur = torch.cuda.current_stream(dev)
    assert cur.cuda_stream == 0
    ctx.set_stream(cur.cuda_stream)
    try:
        n2 = 3000
        X2, y2 = orc.synth(n2, 3)
        dX2 = torch.from_numpy(np.asfortranarray(X2).T.copy()).to(dev); dy2 = torch.from_numpy(y2).to(dev)
        o1 = torch.zeros(3, dtype=torch.float64, device=dev)
        ctx.logml_dev(dX2.data_ptr(), n2, n2, 3, dy2.data_ptr(), 1.0, [0.3], 0.1, 0.0, o1.data_ptr(), info.data_ptr())
        seen1 = o1.clone()                       # enqueued on the null stream right behind the evaluation
        og = torch.zeros((6, 3), dtype=torch.float64, device=dev); ig = torch.zeros(6, dtype=torch.int32, device=dev)
        ctx.logml_grid_dev(dX2.data_ptr(), n2, n2, 3, dy2.data_ptr(), np.ones(6), np.full(6, 0.3), np.full(6, 0.1), 0.0,
                           og.data_ptr(), ig.data_ptr())
        seeng = og.clone()
        torch.cuda.synchronize(dev)
    finally:
        ctx.set_stream(None)
    ref2 = ctx.logml(X2, y2, 1.0, [0.3], 0.1)[0]
    assert seen1[0].item() == ref2 and bool((seeng[:, 0] == ref2).all().item())
    # potrf_dev in place on a torch matrix
    A = orc.cov_exp_quad(X, 1.0, 0.3) + 0.01 * np.eye(500)
    dA = torch.from_numpy(np.ascontiguousarray(A.T)).to(dev)  # symmetric: either order
    ctx.potrf_dev(dA.data_ptr(), 500, 500, info.data_ptr())
    ctx.sync()
    L = dA.cpu().numpy().T
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) < 1e-13 and np.all(np.triu(L, 1) == 0)
