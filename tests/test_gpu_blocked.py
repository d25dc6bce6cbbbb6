"""GPU: parity at the sizes the BLOCKED code paths run at (several outer blocks, ragged last blocks,
fused diagonal blocks, the partial factorisation with a rectangular trailing block, the recursive
right-solve) -- against the oracle where it finishes in seconds and against LAPACK (scipy) beyond.
Every test prints the error it achieved (pytest -s / the captured log shows them); the asserted
tolerance is written next to it.  Reference paths: models/fit_hyperparameters.stan:18-32 (c4 grid),
R/ode_gp_library.R:23-33 + pendulum_fit.R:242-251 (posteriors, c5), covariance.cpp:9-47."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-8  # BASELINE.json north_star: log marginal likelihood within 1e-8 relative


def _lapack_logml(K, y):
    import scipy.linalg as sla
    n = len(y)
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    z = sla.solve_triangular(L, y, lower=True, check_finite=False)
    sld = float(np.log(np.diag(L)).sum())
    q = float(z @ z)
    return -0.5 * q - sld - 0.5 * n * math.log(2 * math.pi), sld, q


def test_c4_grid_corners_and_centre_n8192_vs_lapack(ctx, orc):
    """Config c4 (N = 8192, D = 3): the four corners and the centre of the 8 x 8 (rho, sigma) grid of
    SURVEY section 8(d), evaluated through the GRID entry point (lanes), against LAPACK dpotrf/dtrtrs
    on the oracle's covariance matrix.  rho = 1.0 with sigma = 0.05 is the ill-conditioned corner."""
    from gp_amd.synth import synth
    n = 8192
    X, y = synth(n, 3)
    rho_ax = np.geomspace(0.1, 1.0, 8); sig_ax = np.geomspace(0.05, 0.5, 8)
    pts = [(0, 0), (0, 7), (7, 0), (7, 7), (3, 4)]
    rho = np.array([rho_ax[i] for i, _ in pts]); sig = np.array([sig_ax[j] for _, j in pts])
    out, info = ctx.logml_grid(X, y, np.ones(len(pts)), rho, sig)
    assert np.all(info == 0)
    worst = 0.0
    for k, (r, s) in enumerate(zip(rho, sig)):
        K = orc.cov_exp_quad(X, 1.0, r)
        K[np.diag_indices(n)] += s * s
        lm, sld, q = _lapack_logml(K, y)
        err = abs(out[k, 0] - lm) / abs(lm)
        print("c4 corner rho=%.3f sigma=%.3f: logml %.10e rel err %.2e (sum log diag %.1e, quad %.1e)"
              % (r, s, lm, err, abs(out[k, 1] - sld) / abs(sld), abs(out[k, 2] - q) / abs(q)))
        worst = max(worst, err)
        assert out[k, 0] == ctx.logml(X, y, 1.0, [r], s)[0]   # lanes == single evaluation, bit for bit
    assert worst <= RTOL, worst


def _c5_case(n):
    t = np.linspace(0, 10, n)
    return t, np.concatenate([np.sin(t), np.cos(t)])


def test_c5_joint_order4096_and_8192_vs_lapack(ctx, orc):
    """Config c5, joint [y, y'] covariance [[QQ + s^2 I, QR], [RQ, RR]] + 1e-6 I (R/ode_gp_library.R:29-30)
    at matrix orders 4096 and 8192 against LAPACK and (order 4096) the oracle, at the north-star 1e-8.
    The derivative block carries only the reference's 1e-6 jitter (cond(K) = 8.4e8 at order 4096,
    printed below); measured on MI355X in round 2: GPU vs LAPACK 9e-12 / 9e-13 on logml, 7e-12 on the
    log-determinant half, 3e-11 on the quadratic form; oracle vs LAPACK 9e-11 -- the 1e-7 gate of
    round 1 was never needed."""
    for n in (2048, 4096):
        t, yy = _c5_case(n)
        got = ctx.joint_logml(t, yy, 1.0, 0.5, 0.1, 1e-6)
        K = orc.joint_cov(t, 1.0, 0.5, 0.1, 1e-6)
        if n == 2048:
            ev = np.linalg.eigvalsh(K)
            want = orc.joint_logml(t, yy, 1.0, 0.5, 0.1, 1e-6)
            assert want[3] == 0
        lm, sld, q = _lapack_logml(K, yy)
        e_lm = abs(got[0] - lm) / abs(lm); e_sld = abs(got[1] - sld) / abs(sld); e_q = abs(got[2] - q) / abs(q)
        if n == 2048:
            print("c5 order %d: cond(K) = %.2e; oracle vs LAPACK logml rel %.2e (quad rel %.2e)"
                  % (2 * n, ev[-1] / ev[0], abs(want[0] - lm) / abs(lm), abs(want[2] - q) / abs(q)))
            assert abs(got[0] - want[0]) <= RTOL * abs(want[0])
        print("c5 order %d: GPU vs LAPACK logml rel %.2e, sum log diag rel %.2e, quad rel %.2e" % (2 * n, e_lm, e_sld, e_q))
        assert e_lm <= RTOL and e_sld <= RTOL and e_q <= RTOL, (n, e_lm, e_sld, e_q)


# the last two force wide outer blocks, so that the in-block products run at K = 256 / 512 (recursive
# halving) resp. at the fixed 128 / 256 levels -- different summation orders of the same factorisation
OPTION_SETS = [{}, {"fuse_diag": 0}, {"ksplit": 0, "block_recursive": 0}, {"nb_outer": 256, "fuse_diag": 3},
               {"nb_outer": 1024}, {"nb_outer": 512, "block_recursive": 0}]


def _with_options(ctx, opts, fn):
    defaults = {"fuse_diag": 15, "ksplit": 1, "block_recursive": 1, "nb_outer": 0}
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        return fn()
    finally:
        for k in opts:
            ctx.set_option(k, defaults[k])


@pytest.mark.parametrize("n,m", [(1500, 1500), (4096, 300), (2049, 2049), (300, 5000), (1337, 257), (4096, 512)])
def test_gp_condition_blocked_sizes(ctx, orc, n, m):
    """gpmi_gp_condition (p_Xn / p_dotXn / sample_derivs, R/ode_gp.R:1-32, pendulum_fit.R:242-251) in the
    regime where launch_potrf_partial runs with nfac < ncol over several outer blocks: derivative
    posterior (QQ, RQ, RR) at separate prediction times against the reference's formula with LU solves
    (numpy.linalg.solve == dgesv == base-R solve(); the oracle's own LU where it finishes in seconds),
    with the fused / split / blocking options toggled."""
    rng = np.random.default_rng(n + m)
    t = np.sort(rng.uniform(0.0, n / 10.0, n))          # ~10 points per length-scale: cond(K + s2 I) ~ 1e4
    ts = np.sort(rng.uniform(0.0, n / 10.0, m))
    y = np.sin(t) + 0.05 * rng.standard_normal(n)
    a, l, s2, jit = 1.2, 1.0, 0.01, 1e-8
    K = orc.deriv_cov("QQ", t, t, a, l); Ks = orc.deriv_cov("RQ", ts, t, a, l); Kss = orc.deriv_cov("RR", ts, ts, a, l)
    Kn_dense = K + s2 * np.eye(n)
    mn_ref = Ks @ np.linalg.solve(Kn_dense, y)
    Kn_ref = Kss - Ks @ np.linalg.solve(Kn_dense, Ks.T) + jit * np.eye(m)
    if n * n * (n + 2 * m) <= 2.5e10:  # the oracle's single-threaded LU path where it takes seconds
        mo, Ko = orc.gp_condition(K, Ks, Kss, y, s2, jit)
        assert np.max(np.abs(mo - mn_ref)) <= 1e-9 * np.max(np.abs(mn_ref))
        assert np.max(np.abs(Ko - Kn_ref)) <= 1e-9 * np.max(np.abs(Kn_ref))
    smn, sK = np.max(np.abs(mn_ref)), np.max(np.abs(Kn_ref))
    for opts in OPTION_SETS:
        mn, Kn = _with_options(ctx, opts, lambda: ctx.gp_condition(t, ts, y, a, l, s2, jit, "QQ", "RQ", "RR"))
        e1, e2 = np.max(np.abs(mn - mn_ref)) / smn, np.max(np.abs(Kn - Kn_ref)) / sK
        print("gp_condition n=%d m=%d %s: mn rel %.2e, Kn rel %.2e" % (n, m, opts, e1, e2))
        assert e1 <= RTOL and e2 <= RTOL, (opts, e1, e2)
        assert np.array_equal(Kn, Kn.T)   # both triangles written, symmetric


def test_p_dotXn_n8192_vs_lapack_schur(ctx, orc):
    """One p_dotXn (R/ode_gp.R:19-32; the caller of c5's joint covariance) at N = 8192 -- workspace order
    16385, 16 outer blocks factored, the other half Schur-complemented -- against a LAPACK Cholesky
    Schur complement RR - (L^-1 QR)^T (L^-1 QR)."""
    import scipy.linalg as sla
    from gp_amd import ode_gp
    n = 8192
    t = np.linspace(0.0, 800.0, n) + 0.03 * np.sin(np.arange(n))
    x = np.sin(0.5 * t)
    phi = [1.1, 0.9]; s = 0.1
    p = ode_gp.p_dotXn(t, x, phi, s, ctx=ctx)
    K = orc.deriv_cov("QQ", t, t, 1.1, 0.9)
    K[np.diag_indices(n)] += s * s
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    del K
    QR = orc.deriv_cov("QR", t, t, 1.1, 0.9)            # n x n, Cov(f(t_i), f'(t_j))
    V = sla.solve_triangular(L, QR, lower=True, check_finite=False, overwrite_b=True)   # L^-1 QR
    z = sla.solve_triangular(L, x, lower=True, check_finite=False)
    mn_ref = V.T @ z
    Kn_ref = orc.deriv_cov("RR", t, t, 1.1, 0.9) - V.T @ V
    e1 = np.max(np.abs(p["condMean"] - mn_ref)) / np.max(np.abs(mn_ref))
    e2 = np.max(np.abs(p["condVar"] - Kn_ref)) / np.max(np.abs(Kn_ref))
    print("p_dotXn N=8192: mn rel %.2e, Kn rel %.2e" % (e1, e2))
    assert e1 <= RTOL and e2 <= RTOL


@pytest.mark.parametrize("n", [1024, 3000])
def test_rbf_cov_chol_blocked_sizes(ctx, orc, n):
    """rbf_cov_chol (covariance.cpp:9-47: Sigma + 1e-10 I, L, dL/dl) where the recursive right-solve
    (launch_trsm_right: n > 128, many rows) and several outer blocks are in play.  Spacing 1.5 l keeps
    Sigma well conditioned (cond ~ 5) although the reference's jitter is only 1e-10.  Reference values:
    the oracle (n = 1024) and L Phi(L^-1 Sigma' L^-T) with LAPACK triangular solves."""
    import scipy.linalg as sla
    l = 0.8
    x = 1.5 * l * np.arange(n) + 0.1 * np.sin(np.arange(n))
    L, dL = ctx.rbf_cov_chol(x, l)
    r = x[:, None] - x[None, :]
    S = np.exp(-r * r / (2 * l * l)); Sd = S * r * r / l ** 3
    S[np.diag_indices(n)] += 1e-10
    Lr = sla.cholesky(S, lower=True)
    A = sla.solve_triangular(Lr, Sd, lower=True)
    A = sla.solve_triangular(Lr, A.T, lower=True).T        # L^-1 Sd L^-T
    Phi = np.tril(A); Phi[np.diag_indices(n)] *= 0.5
    dLr = Lr @ Phi
    eL, edL = np.max(np.abs(L - Lr)), np.max(np.abs(dL - dLr)) / np.max(np.abs(dLr))
    print("rbf_cov_chol n=%d: max|L - L_lapack| %.2e, dLdl rel %.2e" % (n, eL, edL))
    assert eL <= 1e-12 and edL <= 1e-10
    assert np.all(np.triu(L, 1) == 0.0)
    if n <= 1024:
        Lo, dLo = orc.rbf_cov_chol(x, l)
        assert np.max(np.abs(L - Lo)) <= 1e-12 and np.max(np.abs(dL - dLo)) <= 1e-10 * np.max(np.abs(dLo))
    # product rule: d(L L^T)/dl = Sigma'
    rows = np.arange(0, n, max(1, n // 40))
    lhs = dL[rows, :] @ L.T + L[rows, :] @ dL.T
    assert np.max(np.abs(lhs - Sd[rows, :])) <= 1e-11 * np.max(np.abs(Sd))


def test_reference_ch2_cells_on_the_gpu(ctx, golden):
    """tests/golden/ch2.json (outputs of the reference's own ch2.py cells, N = 1000): the SE build through
    gpmi_se_cov / gpmi_deriv_cov and the posterior m = Ksd Kdd^-1 f, Kt = Kss - Ksd Kdd^-1 Kds through
    gpmi_gp_condition (10 resp. 4 data points, 1000 prediction points: nfac << ncol)."""
    for cell in golden["ch2"]["cells"]:
        N = cell["N"]; alpha = math.sqrt(cell["eta2"]); rho = math.sqrt(cell["l2"] / 2.0)
        xs = np.linspace(0.0, 1.0, N); xd = np.array(cell["xd"]); f = np.array(cell["f"])
        ri = np.array(cell["rows"]); ci = np.array(cell["cols"])
        tol = 64 * np.finfo(float).eps * cell["eta2"]
        Kss = ctx.se_cov(xs.reshape(-1, 1), None, alpha, [rho])
        assert np.max(np.abs(Kss[np.ix_(ri, ci)] - np.array(cell["Kss_sample"]))) <= tol
        Ksd = ctx.deriv_cov("QQ", xs, xd, alpha, rho)
        assert np.max(np.abs(Ksd[ri, :] - np.array(cell["Ksd"]))) <= tol
        mn, Kn = ctx.gp_condition(xd, xs, f, alpha, rho, cell["sigma2"], 0.0, "QQ", "QQ", "QQ")
        m_ref = np.array(cell["m"])
        assert np.max(np.abs(mn - m_ref)) <= 1e-11 * np.max(np.abs(m_ref))
        assert np.max(np.abs(Kn[np.ix_(ri, ci)] - np.array(cell["Kt_sample"]))) <= 1e-11 * cell["eta2"]
        assert np.max(np.abs(np.diag(Kn) - np.array(cell["Kt_diag"]))) <= 1e-11 * cell["eta2"]


def test_reference_second_derivative_posterior_on_the_gpu(ctx, golden):
    """The (QQ, TQ, TT) posterior of the reference's gp_derivs.py (:88-113: mu(K, KsKi, y), cov(K, KsKi,
    KsKsi) -- the second derivative of the process given noisy values), stored in gp_derivs.json."""
    g = golden["gp_derivs"]["posterior"]
    ts = np.array(g["ts"]); y = np.array(g["y"])
    np.testing.assert_allclose(ctx.deriv_cov("TQ", ts, ts, g["a"], g["l"]), np.array(g["KsKi_TQ"]), rtol=0, atol=1e-14)
    np.testing.assert_allclose(ctx.deriv_cov("TT", ts, ts, g["a"], g["l"]), np.array(g["KsKsi_TT"]), rtol=0, atol=1e-14)
    mn, Kn = ctx.gp_condition(ts, ts, y, g["a"], g["l"], g["s"] ** 2, 0.0, "QQ", "TQ", "TT")
    mu2 = np.array(g["mu_second"]); cov2 = np.array(g["cov_second"])
    assert np.max(np.abs(mn - mu2)) <= RTOL * np.max(np.abs(mu2))
    assert np.max(np.abs(Kn - cov2)) <= RTOL * np.max(np.abs(cov2))


def test_sample_derivs_fused_and_batched(ctx, orc, golden):
    """sample_derivs (pendulum_fit.R:227-255) as ONE device call -- moments of gp_condition, covariance factored in place,
    draw = mu + chol(cov) z -- against the oracle's moments (LU solves as in the reference) and numpy's Cholesky, at the
    reference's own N = 25 data set and at a blocked size; and the batched form (the mclapply loop of :261-268) against
    single draws bit for bit."""
    from gp_amd import pendulum
    g = golden["gp_derivs"]["posterior"]
    ts = np.array(g["ts"]); y = np.array(g["y"])
    z = np.random.default_rng(7).standard_normal(ts.size)
    d = pendulum.sample_derivs([g["l"], g["a"], g["s"]], y, ts, jitter=1e-8, z=z, ctx=ctx)
    cov = np.array(g["cov_deriv"]) + 1e-8 * np.eye(ts.size)
    want = np.array(g["mu_deriv"]) + np.linalg.cholesky(cov) @ z      # the reference python's own moments
    assert np.max(np.abs(d - want)) <= 1e-7 * np.max(np.abs(want))
    # blocked size, separate prediction times (lorenz.Rmd:80-107)
    rng = np.random.default_rng(3)
    n, m = 1800, 1300
    t = np.sort(rng.uniform(0, n / 10.0, n)); tis = np.sort(rng.uniform(0, n / 10.0, m))
    yy = np.sin(t) + 0.1 * rng.standard_normal(n)
    l, a, sy = 0.9, 1.3, 0.1
    z = rng.standard_normal(m)
    draw, mu = ctx.sample_derivs(t, tis, yy, l, a, sy, 1e-6, z)
    K = orc.deriv_cov("QQ", t, t, a, l) + sy * sy * np.eye(n)
    Ks = orc.deriv_cov("RQ", tis, t, a, l); Kss = orc.deriv_cov("RR", tis, tis, a, l)
    mu_ref = Ks @ np.linalg.solve(K, yy)
    cov_ref = Kss - Ks @ np.linalg.solve(K, Ks.T) + 1e-6 * np.eye(m)
    cov_ref = 0.5 * (cov_ref + cov_ref.T)
    draw_ref = mu_ref + np.linalg.cholesky(cov_ref) @ z
    e_mu = np.max(np.abs(mu - mu_ref)) / np.max(np.abs(mu_ref)); e_d = np.max(np.abs(draw - draw_ref)) / np.max(np.abs(draw_ref))
    print("sample_derivs n=%d m=%d: mu rel %.2e, draw rel %.2e" % (n, m, e_mu, e_d))
    assert e_mu <= RTOL and e_d <= 1e-7   # the draw goes through chol(cov) with a 1e-6 jitter: cond ~ 1e6
    # batch == singles, bit for bit, for B not a multiple of the lane count; a failing draw does not stop the others
    B = 6
    P = np.array([[0.9 + 0.02 * b, 1.3, 0.1 + 0.01 * b] for b in range(B)])
    Y = np.column_stack([yy + 0.01 * b for b in range(B)]); Z = rng.standard_normal((m, B))
    draws, mus, info = ctx.sample_derivs_batch(t, tis, Y, P, 1e-6, Z)
    assert np.all(info == 0)
    for b in (0, 3, 5):
        db, mb = ctx.sample_derivs(t, tis, Y[:, b], P[b, 0], P[b, 1], P[b, 2], 1e-6, Z[:, b])
        assert np.array_equal(draws[:, b], db) and np.array_equal(mus[:, b], mb)
    many = pendulum.sample_derivs_many(list(P), list(Y.T), t, tis=tis, jitter=1e-6, Z=Z, ctx=ctx)
    assert np.array_equal(many, draws)
    Pbad = P.copy(); Pbad[2, 2] = 0.0; Pbad[2, 0] = 500.0     # sy = 0, huge length-scale: K is numerically singular
    _, _, info = ctx.sample_derivs_batch(t, tis, Y, Pbad, 0.0, Z)
    assert info[2] != 0 and np.all(np.delete(info, 2) != -1)


@pytest.mark.parametrize("n", [3583, 3585, 4609, 8193])
def test_adaptive_outer_blocks_at_their_thresholds_vs_lapack(ctx, orc, n):
    """The auto outer-block width follows the order of the matrix still to update (1024 / 512 / 256 / 128 columns,
    switching at 8192 / 4608 / 3584): ragged orders right at the switches, every width in one factorisation
    (n = 8193: 1024 -> 512 -> 256 -> 128), single-round trailing updates carrying the next diagonal block -- against
    LAPACK, 1e-10 relative on logml."""
    import scipy.linalg as sla
    X, y = orc.synth(n, 3, seed=n)
    K = orc.cov_exp_quad(X, 1.0, 0.3) + 0.01 * np.eye(n)
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    z = sla.solve_triangular(L, y, lower=True, check_finite=False)
    want = -0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * math.log(2 * math.pi)
    got = ctx.logml(X, y, 1.0, [0.3], 0.1)[0]
    print("n=%d: rel err vs LAPACK %.2e" % (n, abs(got - want) / abs(want)))
    assert abs(got - want) <= 1e-10 * abs(want)


def test_potrf_vs_the_factor_the_reference_computes(ctx, orc, golden):
    """gpmi_potrf against the reference's own executed Cholesky (ch2.py:42 / :84, numpy.linalg.cholesky of
    Kt + 1e-10 I at N = 1000; fixture tests/golden/ch2.json): 8 panels of the blocked code, cond 2.7e9 / 2.5e12.
    Tolerance cond * eps (forward error bound of a backward-stable Cholesky), as in the oracle's CPU test."""
    import math
    eps = np.finfo(float).eps
    for cell in golden["ch2"]["cells"]:
        N = cell["N"]; alpha = math.sqrt(cell["eta2"]); rho = math.sqrt(cell["l2"] / 2.0)
        xs = np.linspace(0.0, 1.0, N); xd = np.array(cell["xd"]); f = np.array(cell["f"])
        _, Kn = orc.gp_condition(orc.QQ(xd, xd, alpha, rho), orc.QQ(xs, xd, alpha, rho), orc.QQ(xs, xs, alpha, rho), f,
                                 cell["sigma2"], 0.0)
        L = ctx.potrf(Kn + cell["L_jitter"] * np.eye(N))
        tol = cell["L_cond"] * eps
        d = np.diag(L); dref = np.array(cell["L_diag"])
        rows = np.array(cell["L_rows"]); idx = cell["L_rows_idx"]
        e_diag = np.max(np.abs(d - dref) / dref); e_rows = np.max(np.abs(L[idx, :] - rows)) / np.max(np.abs(rows))
        e_sld = abs(np.log(d).sum() - cell["L_sum_log_diag"])
        print("ch2 N=%d cond %.1e on the GPU: diag rel %.2e, rows rel %.2e, sum log diag abs %.2e (tolerance %.1e)"
              % (N, cell["L_cond"], e_diag, e_rows, e_sld, tol))
        assert e_diag <= tol and e_rows <= tol and e_sld <= tol * math.sqrt(N)


@pytest.mark.parametrize("n,m", [(1801, 1300), (1337, 3701)])
def test_sample_derivs_posterior_factored_at_an_odd_offset(ctx, orc, n, m):
    """sample_derivs_core factors the Schur complement IN PLACE at W + n + n ld: with odd n the blocked factorisation
    (fused SYRK, quadrant tiles with 16-B loads, sub-tiled diagonal blocks) gets a base pointer that is 8- but not
    16-byte aligned; m = 3701 also reaches the adaptive outer-block widths (256 -> 128).  Against LU-solve moments and
    numpy's Cholesky, as in the aligned case."""
    rng = np.random.default_rng(n)
    t = np.sort(rng.uniform(0, n / 10.0, n)); tis = np.sort(rng.uniform(0, n / 10.0, m))
    yy = np.sin(t) + 0.1 * rng.standard_normal(n)
    l, a, sy = 0.9, 1.3, 0.1
    z = rng.standard_normal(m)
    draw, mu = ctx.sample_derivs(t, tis, yy, l, a, sy, 1e-6, z)
    K = orc.deriv_cov("QQ", t, t, a, l) + sy * sy * np.eye(n)
    Ks = orc.deriv_cov("RQ", tis, t, a, l); Kss = orc.deriv_cov("RR", tis, tis, a, l)
    mu_ref = Ks @ np.linalg.solve(K, yy)
    cov_ref = Kss - Ks @ np.linalg.solve(K, Ks.T) + 1e-6 * np.eye(m)
    draw_ref = mu_ref + np.linalg.cholesky(0.5 * (cov_ref + cov_ref.T)) @ z
    e_mu = np.max(np.abs(mu - mu_ref)) / np.max(np.abs(mu_ref)); e_d = np.max(np.abs(draw - draw_ref)) / np.max(np.abs(draw_ref))
    print("sample_derivs n=%d m=%d (odd offset): mu rel %.2e, draw rel %.2e" % (n, m, e_mu, e_d))
    assert e_mu <= RTOL and e_d <= 1e-7


@pytest.mark.parametrize("n", [4700, 6001, 8192, 12289])
def test_xcd_band_walk_of_the_trailing_updates_equals_row_major(ctx, n):
    """syrk_order = 2 (XCD-partitioned band walk of the multi-round trailing updates, tail split per XCD, thin last
    row as quadrants in its own workgroup) computes the same tiles with the same sums as the row-major walk: the
    log marginal likelihood agrees to 1e-13 (which tiles take the quadrant path differs, nothing else)."""
    from gp_amd.synth import synth
    X, y = synth(n, 3)
    try:
        ctx.set_option("syrk_order", 0)
        a = ctx.logml(X, y, 1.0, [0.3], 0.1)
        ctx.set_option("syrk_order", 2)
        b = ctx.logml(X, y, 1.0, [0.3], 0.1)
    finally:
        ctx.set_option("syrk_order", 2)
    print("n=%d: row-major %.12e, XCD bands %.12e, rel %.1e" % (n, a[0], b[0], abs(a[0] - b[0]) / abs(a[0])))
    assert abs(a[0] - b[0]) <= 1e-13 * abs(a[0]) and abs(a[2] - b[2]) <= 1e-12 * abs(a[2])
