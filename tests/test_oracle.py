"""CPU: pin the oracle (oracle/gp_oracle.c) against the golden fixtures and independent
implementations before anything trusts it."""
import math

import numpy as np
import pytest
import scipy.linalg as sla

KINDS = ["QQ", "QR", "RQ", "RR", "QT", "TQ", "RT", "TR", "TT"]


def test_deriv_kernels_match_reference_python(orc, golden):
    # golden/gp_derivs.json: outputs of the reference's gp_derivs.py:15-40 (a^2 folded in)
    for c in golden["gp_derivs"]["kernel_cases"]:
        for k in KINDS:
            got = c["a"] ** 2 * float(orc.deriv_elem(k, c["tj"], c["tk"], c["l"]))
            want = c["out"][k]
            assert got == pytest.approx(want, rel=2e-15, abs=1e-300), (k, c)


def test_posterior_matches_reference_python(orc, golden):
    p = golden["gp_derivs"]["posterior"]
    ts = np.array(p["ts"]); y = np.array(p["y"])
    K = orc.deriv_cov("QQ", ts, ts, p["a"], p["l"])
    np.testing.assert_allclose(K, np.array(p["K"]), rtol=1e-15, atol=0)
    np.testing.assert_allclose(orc.deriv_cov("TQ", ts, ts, p["a"], p["l"]), np.array(p["KsKi_TQ"]), rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(orc.deriv_cov("TT", ts, ts, p["a"], p["l"]), np.array(p["KsKsi_TT"]), rtol=1e-14, atol=1e-300)
    # derivative posterior == sample_derivs moments without jitter (gp_derivs.py:97-113 uses numpy LU solve)
    mu, cov = orc.sample_derivs_moments(ts, y, p["l"], p["a"], p["s"], jitter=0.0)
    np.testing.assert_allclose(mu, np.array(p["mu_deriv"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(cov, np.array(p["cov_deriv"]), rtol=0, atol=1e-9)
    mn, Kn = orc.p_Xn(ts, y, p["a"], p["l"], p["s"])
    np.testing.assert_allclose(mn, np.array(p["mu_value"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(Kn, np.array(p["cov_value"]), rtol=0, atol=1e-9)


def test_logml_known_answers(orc, golden):
    for k in golden["kat"]["kats"]:
        x = np.array(k["x"]); y = np.array(k["y"])
        lm, sld, q, info = orc.logml(x, y, k["alpha"], k["rho"], k["sigma"])
        assert info == 0
        assert lm == pytest.approx(k["logml"], rel=1e-11), k["name"]
        assert sld == pytest.approx(k["sum_log_diag"], rel=1e-11)
        assert q == pytest.approx(k["quad"], rel=1e-9)
        Kc = orc.cov_exp_quad(x.reshape(len(y), -1), k["alpha"], k["rho"]) + k["sigma"] ** 2 * np.eye(len(y))
        assert orc.cholesky(Kc)[1, 0] == pytest.approx(k["L10"], rel=1e-13)


def test_cholesky_vs_lapack_and_blocked(orc):
    rng = np.random.default_rng(0)
    for n in (1, 2, 17, 130, 300):
        A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
        L = orc.cholesky(A)
        np.testing.assert_allclose(L, sla.cholesky(A, lower=True), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(orc.cholesky(A, blocked=True), L, rtol=1e-12, atol=1e-13)
        assert np.all(np.triu(L, 1) == 0)
    with pytest.raises(ValueError) as e:
        orc.cholesky(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert e.value.args[0] == 2  # leading minor of order 2


def test_triangular_and_lu(orc):
    rng = np.random.default_rng(1)
    n = 40
    A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
    L = np.linalg.cholesky(A); b = rng.standard_normal(n)
    np.testing.assert_allclose(orc.trsv_lower(L, b), sla.solve_triangular(L, b, lower=True), rtol=1e-12)
    np.testing.assert_allclose(orc.trmv_lower(L, b), L @ b, rtol=1e-13)
    B = rng.standard_normal((n, 3)); M = rng.standard_normal((n, n))
    np.testing.assert_allclose(orc.lu_solve(M, B), np.linalg.solve(M, B), rtol=1e-9, atol=1e-11)


def test_kernel_identities(orc):
    x = np.linspace(-1.5, 2.0, 13); y = np.linspace(-0.7, 1.1, 7); l, a = 0.8, 1.3
    # RQ = t(QR) (R/ode_gp.R:25); QT = -RR; symmetry of QQ, RR, TT
    np.testing.assert_allclose(orc.deriv_cov("RQ", x, y, a, l), orc.deriv_cov("QR", y, x, a, l).T, rtol=1e-15)
    np.testing.assert_allclose(orc.deriv_cov("QT", x, y, a, l), -orc.deriv_cov("RR", x, y, a, l), rtol=1e-15)
    # finite differences: d/dtk QQ = QR, d/dtj QR = RR (design_notes.Rmd:10-23), d/dtk QR = QT ...
    h = 1e-5
    tj, tk, l = 0.3, 1.1, 0.7
    fd = (orc.deriv_elem("QQ", tj, tk + h, l) - orc.deriv_elem("QQ", tj, tk - h, l)) / (2 * h)
    assert fd == pytest.approx(float(orc.deriv_elem("QR", tj, tk, l)), rel=1e-8)
    fd = (orc.deriv_elem("QR", tj + h, tk, l) - orc.deriv_elem("QR", tj - h, tk, l)) / (2 * h)
    assert fd == pytest.approx(float(orc.deriv_elem("RR", tj, tk, l)), rel=1e-8)
    fd = (orc.deriv_elem("RR", tj, tk + h, l) - orc.deriv_elem("RR", tj, tk - h, l)) / (2 * h)
    assert fd == pytest.approx(float(orc.deriv_elem("RT", tj, tk, l)), rel=1e-7)
    fd = (orc.deriv_elem("RT", tj + h, tk, l) - orc.deriv_elem("RT", tj - h, tk, l)) / (2 * h)
    assert fd == pytest.approx(float(orc.deriv_elem("TT", tj, tk, l)), rel=1e-7)
    l = 0.8
    # matrix API (R/kernels.R) == a^2 * elementwise API; RR compat differs only when a != 1
    np.testing.assert_allclose(orc.QQ(x, y, a, l), orc.deriv_cov("QQ", x, y, a, l), rtol=1e-15)
    np.testing.assert_allclose(orc.QR(x, y, a, l), orc.deriv_cov("QR", x, y, a, l), rtol=4e-16, atol=1e-300)
    np.testing.assert_allclose(orc.RR(x, y, a, l), orc.deriv_cov("RR", x, y, a, l), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(orc.RR(x, y, 1.0, l, compat=True), orc.RR(x, y, 1.0, l), rtol=0, atol=0)
    assert np.max(np.abs(orc.RR(x, y, a, l, compat=True) - orc.RR(x, y, a, l))) > 1e-3
    # QQard == QQ for D = 1 up to the last ulp; ARD vs isotropic
    np.testing.assert_allclose(orc.QQard(x.reshape(-1, 1), y.reshape(-1, 1), a, [l]), orc.QQ(x, y, a, l), rtol=4e-15)
    X = np.random.default_rng(2).random((9, 3)); Y = np.random.default_rng(3).random((5, 3))
    np.testing.assert_allclose(orc.QQard(X, Y, a, [l]), orc.QQard(X, Y, a, [l, l, l]), rtol=0, atol=0)
    d2 = ((X[:, None, :] - Y[None, :, :]) ** 2 / np.array([0.5, 1.0, 2.0]) ** 2).sum(-1)
    np.testing.assert_allclose(orc.QQard(X, Y, a, [0.5, 1.0, 2.0]), a * a * np.exp(-0.5 * d2), rtol=1e-14)
    # Stan cov_exp_quad: exact alpha^2 diagonal, symmetric, equals QQard off the diagonal
    K = orc.cov_exp_quad(X, a, l)
    assert np.all(np.diag(K) == a * a) and np.all(K == K.T)
    np.testing.assert_allclose(K, orc.QQard(X, X, a, [l]), rtol=1e-14)


def test_joint_cov_and_posteriors(orc):
    t = np.linspace(-2, 2, 21); f = np.exp(t); a, l, s = 1.1, 0.9, 0.05
    K = orc.joint_cov(t, a, l, s, 1e-6)
    n = t.size
    np.testing.assert_allclose(K, K.T, rtol=0, atol=1e-15)
    np.testing.assert_allclose(K[:n, :n], orc.QQ(t, t, a, l) + (s * s + 1e-6) * np.eye(n), rtol=1e-15)
    np.testing.assert_allclose(K[:n, n:], orc.QR(t, t, a, l), rtol=1e-15)
    np.testing.assert_allclose(K[n:, n:], orc.RR(t, t, a, l) + 1e-6 * np.eye(n), rtol=1e-15)
    # direct (R/ode_gp.R:19-32) vs dense numpy
    mn, Kn = orc.p_dotXn(t, f, a, l, s)
    QQm, QRm, RRm = orc.QQ(t, t, a, l), orc.QR(t, t, a, l), orc.RR(t, t, a, l)
    A = QQm + s * s * np.eye(n)
    np.testing.assert_allclose(mn, QRm.T @ np.linalg.solve(A, f), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(Kn, RRm - QRm.T @ np.linalg.solve(A, QRm), rtol=0, atol=1e-8)
    # joint / condMVN form (R/ode_gp_library.R:23-33) == direct form with the 1e-6 jitters
    cm, cv = orc.p_dotXn_joint(t, f, a, l, s, 1e-6)
    A6 = A + 1e-6 * np.eye(n)
    np.testing.assert_allclose(cm, QRm.T @ np.linalg.solve(A6, f), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(cv, RRm + 1e-6 * np.eye(n) - QRm.T @ np.linalg.solve(A6, QRm), rtol=0, atol=1e-8)
    # joint log marginal == dense formula
    yy = np.concatenate([np.sin(t), np.cos(t)])
    lm, sld, q, info = orc.joint_logml(t, yy, a, l, s, 1e-6)
    Kj = orc.joint_cov(t, a, l, s, 1e-6)
    Lj = np.linalg.cholesky(Kj); z = sla.solve_triangular(Lj, yy, lower=True)
    assert info == 0 and lm == pytest.approx(-0.5 * z @ z - np.log(np.diag(Lj)).sum() - n * np.log(2 * np.pi), rel=1e-9)


def test_rbf_cov_chol_tangent(orc):
    # covariance.cpp:9-47: L of exp(-(xi-xj)^2/(2 l^2)) + 1e-10 I and dL/dl; check L L^T, and the
    # tangent against a central finite difference of the factor (well-conditioned spacing)
    x = np.linspace(0, 6, 13); l = 0.45
    L, dL = orc.rbf_cov_chol(x, l)
    S = np.exp(-(x[:, None] - x[None, :]) ** 2 / (2 * l * l)) + 1e-10 * np.eye(x.size)
    np.testing.assert_allclose(L @ L.T, S, rtol=0, atol=1e-14)
    h = 1e-6
    Lp, _ = orc.rbf_cov_chol(x, l + h); Lm, _ = orc.rbf_cov_chol(x, l - h)
    np.testing.assert_allclose(dL, (Lp - Lm) / (2 * h), rtol=0, atol=2e-8)
    # product rule: d(L L^T)/dl == dSigma/dl
    dS = S * (x[:, None] - x[None, :]) ** 2 / l ** 3
    np.testing.assert_allclose(dL @ L.T + L @ dL.T, dS - np.diag(np.diag(dS)), rtol=0, atol=1e-12)


def test_approx_L_hermite(orc):
    # covariance.cpp:49-96 / cubic_spline_test.R:13-18: cubic Hermite through (y, dy/dl) knots
    x = np.linspace(0, 5, 9); lp = np.array([0.5, 0.7, 0.9, 1.2])
    Ls, dLs = zip(*[orc.rbf_cov_chol(x, l) for l in lp])
    for k, l in enumerate(lp[:-1]):
        np.testing.assert_allclose(orc.approx_L(l + 1e-12, lp, Ls, dLs), Ls[k], rtol=0, atol=1e-9)
    mid = orc.approx_L(0.8, lp, Ls, dLs)
    exact, _ = orc.rbf_cov_chol(x, 0.8)
    assert np.max(np.abs(mid - exact)) < 5e-3 and np.all(np.triu(mid, 1) == 0)


def test_stan_lp_formula(orc):
    import math
    v = orc.stan_lp(-3.5, 7.25, 1.2, 0.8, 0.3)
    want = 3.5 - 0.5 * 7.25 + 3 * math.log(0.8) - 4 * 0.8 - 0.72 - 0.045 + math.log(0.8) + math.log(1.2) + math.log(0.3)
    assert v == pytest.approx(want, rel=1e-14)


def test_synth_is_deterministic(orc):
    X, y = orc.synth(50, 3); X2, y2 = orc.synth(50, 3)
    assert np.array_equal(X, X2) and np.array_equal(y, y2)
    assert X.min() >= 0 and X.max() < 1 and abs(np.mean(X) - 0.5) < 0.1


def test_approx_Lz_is_blend_times_z(orc):
    # models/cubic_interpolated_gp.hpp:38-73: f = v z with v the approx_L blend
    x = np.linspace(0, 3, 7); lp = np.array([0.4, 0.6, 1.0])
    Ls, dLs = zip(*[orc.rbf_cov_chol(x, l) for l in lp])
    z = np.cos(np.arange(7.0))
    for l in (0.45, 0.6, 0.93):
        np.testing.assert_allclose(orc.approx_Lz(l, lp, Ls, dLs, z), orc.approx_L(l, lp, Ls, dLs) @ z, rtol=1e-14, atol=1e-15)


def test_logml_grad_matches_finite_differences(orc):
    # no reference fixture exists for the gradient (Stan computes it by autodiff): the restated formula
    # 1/2 tr((aa' - K^-1) dK) is pinned by central differences of the (KAT-pinned) log marginal likelihood
    rng = np.random.default_rng(11)
    X = rng.random((40, 2)); y = np.sin(3 * X[:, 0]) + 0.1 * rng.standard_normal(40)
    a, r, s = 1.3, 0.45, 0.2
    out, g, info = orc.logml_grad(X, y, a, r, s)
    assert info == 0 and out[0] == pytest.approx(orc.logml(X, y, a, r, s)[0], rel=1e-13)
    h = 1e-5
    fd = [(orc.logml(X, y, a + h, r, s)[0] - orc.logml(X, y, a - h, r, s)[0]) / (2 * h),
          (orc.logml(X, y, a, r + h, s)[0] - orc.logml(X, y, a, r - h, s)[0]) / (2 * h),
          (orc.logml(X, y, a, r, s + h)[0] - orc.logml(X, y, a, r, s - h)[0]) / (2 * h)]
    np.testing.assert_allclose(g, fd, rtol=2e-7)


def _seq_case(orc, n=40, D=1, seed=3):
    rng = np.random.default_rng(seed)
    X = np.sort(rng.uniform(0, n * 0.9, size=(n, D)), axis=0)
    t = np.linspace(0, 6, n)
    mn, Kn = orc.p_dotXn(t, np.sin(t), 1.0, 0.9, 0.1)
    return X, mn, Kn


def test_seq_sampler_chain_equals_joint(orc):
    """Pin of the create_p_dotXnS restatement (R/ode_gp_library.R:43-93): the conditionals it
    returns call by call must be those of the final joint N(m, K) obtained independently by a
    Cholesky factor of K (mu_i = m_i + l.w, var_i = L_ii^2), and the closure's bookkeeping
    (i, Xs, dot_Xs) must follow the calls."""
    import scipy.linalg
    X, mn, Kn = _seq_case(orc)
    s = orc.create_p_dotXnS([X[:, 0]], mn, Kn, 1.3, 0.8)
    pts = [3.1, 10.2, 3.4, 25.0, 17.7, 10.2001]
    zs = [0.3, -1.1, 0.7, 0.2, -0.4, 1.5]
    got = [s([p], z) for p, z in zip(pts, zs)]
    assert s.i == len(pts) + 1 and s.Xs.shape == (len(pts), 1)
    m, K = s.joint()
    L = scipy.linalg.cholesky(K, lower=True)
    w = scipy.linalg.solve_triangular(L, s.dot_Xs - m, lower=True)
    for i, g in enumerate(got):
        mu = m[i] + L[i, :i] @ w[:i]
        assert abs(g["mu"] - mu) <= 1e-9 * max(1.0, abs(mu))
        assert abs(g["sigma"] - L[i, i] ** 2) <= 1e-9 * max(1e-3, L[i, i] ** 2)
        assert g["dot_xs"] == g["mu"] + np.sqrt(g["sigma"]) * zs[i]
    # a state visited twice: the second draw is pinned to the first within the 1e-6 jitters
    assert abs(got[5]["mu"] - got[1]["dot_xs"]) < 1e-2 and got[5]["sigma"] < 1e-4
    # :83 as written: the variance is used as the standard deviation
    c = orc.create_p_dotXnS([X[:, 0]], mn, Kn, 1.3, 0.8, compat_sd=True)
    r = c([3.1], 0.3)
    assert r["dot_xs"] == r["mu"] + r["sigma"] * 0.3


def test_oracle_vs_reference_ch2_cells(orc, golden):
    """tests/golden/ch2.json: outputs of the reference's own ch2.py GP cells (its makeK at N = 1000,
    numpy.linalg.solve posterior).  makeK is eta2 exp(-(x1-x2)^2 / l2): alpha^2 = eta2, rho^2 = l2 / 2."""
    for cell in golden["ch2"]["cells"]:
        N = cell["N"]; alpha = math.sqrt(cell["eta2"]); rho = math.sqrt(cell["l2"] / 2.0)
        xs = np.linspace(0.0, 1.0, N); xd = np.array(cell["xd"]); f = np.array(cell["f"])
        ri = np.array(cell["rows"]); ci = np.array(cell["cols"])
        tol = 64 * np.finfo(float).eps * cell["eta2"]   # (x1-x2)^2 / l2 up to 20: a few ulp of the argument
        Kss = orc.QQ(xs, xs, alpha, rho)
        assert np.max(np.abs(Kss[np.ix_(ri, ci)] - np.array(cell["Kss_sample"]))) <= tol
        Ksd = orc.QQ(xs, xd, alpha, rho)
        assert np.max(np.abs(Ksd[ri, :] - np.array(cell["Ksd"]))) <= tol
        Kdd = orc.QQ(xd, xd, alpha, rho)
        assert np.max(np.abs(Kdd + cell["sigma2"] * np.eye(xd.size) - np.array(cell["Kdd"]))) <= tol
        mn, Kn = orc.gp_condition(Kdd, Ksd, Kss, f, cell["sigma2"], 0.0)
        m_ref = np.array(cell["m"])
        assert np.max(np.abs(mn - m_ref)) <= 1e-11 * np.max(np.abs(m_ref))
        assert np.max(np.abs(Kn[np.ix_(ri, ci)] - np.array(cell["Kt_sample"]))) <= 1e-11 * cell["eta2"]
        assert np.max(np.abs(np.diag(Kn) - np.array(cell["Kt_diag"]))) <= 1e-11 * cell["eta2"]
        # the same numbers through the D-dimensional builder (QQard) and the Stan-style one
        assert np.max(np.abs(orc.QQard(xs[ri].reshape(-1, 1), xs[ci].reshape(-1, 1), alpha, [rho])
                             - np.array(cell["Kss_sample"]))) <= tol


def _ch2_factored_matrix(orc, cell):
    """Kt + 1e-10 I of a ch2.py cell, rebuilt from the cell's inputs with the oracle's kernels."""
    N = cell["N"]; alpha = math.sqrt(cell["eta2"]); rho = math.sqrt(cell["l2"] / 2.0)
    xs = np.linspace(0.0, 1.0, N); xd = np.array(cell["xd"]); f = np.array(cell["f"])
    _, Kn = orc.gp_condition(orc.QQ(xd, xd, alpha, rho), orc.QQ(xs, xd, alpha, rho), orc.QQ(xs, xs, alpha, rho), f,
                             cell["sigma2"], 0.0)
    return Kn + cell["L_jitter"] * np.eye(N)


def test_oracle_cholesky_vs_the_factor_the_reference_computes(orc, golden):
    """ch2.py:42 / :84, L = numpy.linalg.cholesky(Kt + 1e-10 I) at N = 1000: the only Cholesky factor the
    reference itself can execute here (LAPACK dpotrf).  cond(Kt + 1e-10 I) = 2.7e9 / 2.5e12 (stored in the
    fixture), so a backward-stable factorisation of a matrix rebuilt to a few ulp agrees with it to
    cond * eps relative -- the tolerance used, stated per quantity; achieved: ~2e-8 / ~1e-5 on the diagonal."""
    eps = np.finfo(float).eps
    for cell in golden["ch2"]["cells"]:
        A = _ch2_factored_matrix(orc, cell)
        tol = cell["L_cond"] * eps
        for blocked in (False, True):
            L = orc.cholesky(A, blocked=blocked)
            d = np.diag(L); dref = np.array(cell["L_diag"])
            e_diag = np.max(np.abs(d - dref) / dref)
            rows = np.array(cell["L_rows"]); idx = cell["L_rows_idx"]
            e_rows = np.max(np.abs(L[idx, :] - rows)) / np.max(np.abs(rows))
            e_sld = abs(np.log(d).sum() - cell["L_sum_log_diag"])
            print("ch2 N=%d cond %.1e (%s): diag rel %.2e, rows rel %.2e, sum log diag abs %.2e (tolerance %.1e)"
                  % (cell["N"], cell["L_cond"], "blocked" if blocked else "unblocked", e_diag, e_rows, e_sld, tol))
            assert e_diag <= tol and e_rows <= tol
            assert e_sld <= tol * math.sqrt(cell["N"])   # N relative errors of size <= tol, added in quadrature


def test_approx_Lz_grad_is_the_derivative_of_approx_Lz(orc):
    """models/cubic_interpolated_gp.hpp:6-32,62-72: the `var` overload gives output i the partial
    (dvdl z)(i) with respect to l.  No reference output exists (Stan is absent): the restated dvdl is
    pinned by central differences of the (already pinned) value orc_approx_Lz inside one interval, by the
    Hermite end conditions dv/dl = dLdl at the knots, and by the value being unchanged."""
    x = np.linspace(0, 10, 30)
    from scipy.stats import gamma
    lp = np.linspace(gamma.ppf(0.05, 4.0, scale=0.25), gamma.ppf(0.95, 4.0, scale=0.25), 10)   # test_interpolate.R:9
    Ls, dLs = zip(*[orc.rbf_cov_chol(x, l) for l in lp])
    z = np.cos(0.7 * np.arange(30.0))
    for l in (0.5 * (lp[0] + lp[1]), lp[3] + 0.3 * (lp[4] - lp[3]), lp[8] + 0.9 * (lp[9] - lp[8])):
        f, g = orc.approx_Lz_grad(l, lp, Ls, dLs, z)
        np.testing.assert_allclose(f, orc.approx_Lz(l, lp, Ls, dLs, z), rtol=1e-15, atol=1e-16)
        h = 1e-6
        fd = (orc.approx_Lz(l + h, lp, Ls, dLs, z) - orc.approx_Lz(l - h, lp, Ls, dLs, z)) / (2 * h)
        np.testing.assert_allclose(g, fd, rtol=0, atol=1e-8 * np.max(np.abs(g)))
    for k in (0, 4, 8):   # just right of a knot: dv/dl -> dLdl[k]
        _, g = orc.approx_Lz_grad(lp[k] + 1e-13, lp, Ls, dLs, z)
        np.testing.assert_allclose(g, dLs[k] @ z, rtol=0, atol=1e-9 * np.max(np.abs(g)))
    _, g = orc.approx_Lz_grad(lp[5] - 1e-13, lp, Ls, dLs, z)   # ... and just left of one
    np.testing.assert_allclose(g, dLs[5] @ z, rtol=0, atol=1e-9 * np.max(np.abs(g)))
