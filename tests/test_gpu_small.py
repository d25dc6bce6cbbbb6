"""GPU: the one-workgroup small-N evaluation (k_logml_small / k_logml_small_batch / k_potrf_small) -- the sizes the
reference's own drivers run the path at: N = 21 (R/tests.R:5-19), 79 .. 199 (pendulum_fit*.R:206-214), 256
(BASELINE c1).  Parity against the oracle for EVERY n in 1 .. 256, agreement with the blocked multi-launch path
(small_n = 0) where both exist, the batched grid in one launch, non-PD points, ARD and D up to 8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-8


def _case(n, D, seed=0):
    rng = np.random.default_rng(1000 * D + n + seed)
    X = np.asfortranarray(rng.uniform(0.0, 1.0 + n / 40.0, (n, D)))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    return X, y


DEFAULTS = {"small_n": 256, "small_n1": 128, "small_m": 160}


@pytest.fixture
def one_wg(ctx):
    """Force the one-workgroup kernels wherever they exist (by default a SINGLE evaluation above n = 128 and a
    partial factorisation above 160 rows take the multi-CU launch chain, which is faster there)."""
    ctx.set_option("small_n1", 256)
    ctx.set_option("small_m", 640)
    yield ctx
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)


def test_small_logml_every_n_vs_oracle_and_blocked_path(one_wg, orc):
    ctx = one_wg
    worst_o = worst_b = 0.0
    for n in range(1, 257):
        X, y = _case(n, 1)
        got = ctx.logml(X, y, 1.1, [0.7], 0.2)
        want = orc.logml(X, y, 1.1, 0.7, 0.2)
        assert want[3] == 0
        e = max(abs(got[k] - want[k]) / max(abs(want[k]), 1e-300) for k in (1, 2))
        e0 = abs(got[0] - want[0]) / abs(want[0])
        assert e0 <= RTOL and e <= RTOL, (n, got, want)
        worst_o = max(worst_o, e0)
        if n % 5 == 1 or n > 250:
            try:
                ctx.set_option("small_n", 0)
                ctx.set_option("small_m", 0)
                blk = ctx.logml(X, y, 1.1, [0.7], 0.2)
            finally:
                ctx.set_option("small_n", 256)
                ctx.set_option("small_m", 640)
            eb = abs(got[0] - blk[0]) / abs(blk[0])
            assert eb <= 1e-12, (n, got, blk)
            worst_b = max(worst_b, eb)
    print("small-N logml, n = 1..256: worst rel err vs oracle %.2e, vs the blocked path %.2e" % (worst_o, worst_b))


@pytest.mark.parametrize("n,D", [(21, 1), (21, 3), (100, 2), (128, 3), (129, 1), (199, 1), (200, 5), (256, 8), (255, 4)])
def test_small_logml_dims_and_ard(one_wg, orc, n, D):
    ctx = one_wg
    X, y = _case(n, D, 7)
    ell = np.linspace(0.5, 1.5, D)
    got = ctx.logml(X, y, 0.9, ell, 0.15, 1e-10)
    want = orc.logml(X / ell, y, 0.9, 1.0, 0.15, 1e-10)   # ARD == isotropic rho = 1 on inputs scaled by 1 / ell
    assert want[3] == 0 and abs(got[0] - want[0]) <= RTOL * abs(want[0]), (got, want)
    # covariance arithmetic is shared with the big builder: the factor of the same matrix through gpmi_potrf
    K = ctx.se_cov(X, None, 0.9, ell, 0.15 ** 2 + 1e-10)
    L = np.linalg.cholesky(K)
    z = np.linalg.solve(L, y)
    ref = -0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * np.log(2 * np.pi)
    assert abs(got[0] - ref) <= RTOL * abs(ref)


def test_small_reference_known_answers(one_wg, golden):
    ctx = one_wg
    """K1 (the R/tests.R:5 grid, N = 21) and K2 (c1: N = 256) through the one-workgroup kernel."""
    for kat in golden["kat"]["kats"]:
        x = np.asarray(kat["x"], dtype=float)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        if x.shape[0] > 256 or x.shape[1] > 8:
            continue
        got = ctx.logml(x, np.asarray(kat["y"]), kat["alpha"], [kat["rho"]], kat["sigma"])
        assert abs(got[0] - kat["logml"]) <= RTOL * abs(kat["logml"]), (kat["name"], got[0], kat["logml"])


@pytest.mark.parametrize("n", [21, 128, 199, 256])
def test_small_grid_is_one_launch_per_128_points_and_equals_single_calls(one_wg, orc, n):
    ctx = one_wg
    X, y = _case(n, 2, 3)
    G = 150  # two launches: 128 + 22 points
    rng = np.random.default_rng(5)
    rho = rng.uniform(0.3, 2.0, G); sig = rng.uniform(0.05, 0.5, G); alpha = rng.uniform(0.5, 1.5, G)
    rho[17] = 60.0; sig[17] = 1e-9   # not positive definite in fp64: NaN + info, the others unaffected
    out, info = ctx.logml_grid(X, y, alpha, rho, sig)
    assert info[17] > 0 and np.all(np.isnan(out[17]))
    ok = np.arange(G) != 17
    assert np.all(info[ok] == 0) and np.all(np.isfinite(out[ok]))
    for g in (0, 1, 16, 18, 127, 128, 149):
        single = ctx.logml(X, y, alpha[g], [rho[g]], sig[g])
        assert tuple(out[g]) == tuple(single), (g, out[g], single)   # bit for bit
    for g in (3, 140):
        want = orc.logml(X, y, alpha[g], rho[g], sig[g])
        assert abs(out[g, 0] - want[0]) <= RTOL * abs(want[0])


def test_small_not_positive_definite_order(one_wg):
    ctx = one_wg
    """info = order of the first non-positive leading minor, as dpotrf / base-R chol() report it."""
    n = 150
    x = np.linspace(0, 1, n).reshape(-1, 1)
    from gp_amd import NotPositiveDefinite
    with pytest.raises(NotPositiveDefinite) as ei:
        ctx.logml(x, np.zeros(n), 1.0, [50.0], 0.0)   # rank-deficient in fp64, no noise, no jitter
    assert 1 <= ei.value.order <= n


@pytest.mark.parametrize("n,m", [(25, 25), (79, 40), (199, 150), (120, 260)])
def test_small_partial_factorisation_posteriors(one_wg, orc, n, m):
    ctx = one_wg
    """gpmi_gp_condition at pendulum sizes runs its augmented partial factorisation in ONE workgroup
    (k_potrf_small, forced up to 640 rows here): against the reference's LU formula and the blocked path."""
    rng = np.random.default_rng(n + m)
    t = np.sort(rng.uniform(0, n / 10.0, n)); ts = np.sort(rng.uniform(0, n / 10.0, m))
    y = np.sin(t) + 0.05 * rng.standard_normal(n)
    a, l, s2, jit = 1.2, 1.0, 0.01, 1e-8
    K = orc.deriv_cov("QQ", t, t, a, l); Ks = orc.deriv_cov("RQ", ts, t, a, l); Kss = orc.deriv_cov("RR", ts, ts, a, l)
    Kd = K + s2 * np.eye(n)
    mn_ref = Ks @ np.linalg.solve(Kd, y)
    Kn_ref = Kss - Ks @ np.linalg.solve(Kd, Ks.T) + jit * np.eye(m)
    ctx.set_option("small_gc", 0)     # the launch chain around the one-workgroup factorisation
    mn, Kn = ctx.gp_condition(t, ts, y, a, l, s2, jit, "QQ", "RQ", "RR")
    try:
        ctx.set_option("small_m", 0)
        mn_b, Kn_b = ctx.gp_condition(t, ts, y, a, l, s2, jit, "QQ", "RQ", "RR")
    finally:
        ctx.set_option("small_m", 640)
        ctx.set_option("small_gc", 180)
    if n + m + 1 <= 180:              # default: the whole call is ONE launch (k_gp_condition_small)
        mn_f, Kn_f = ctx.gp_condition(t, ts, y, a, l, s2, jit, "QQ", "RQ", "RR")
        assert np.max(np.abs(mn_f - mn)) <= 1e-12 * np.max(np.abs(mn)) and np.max(np.abs(Kn_f - Kn)) <= 1e-12 * np.max(np.abs(Kn))
        assert np.array_equal(Kn_f, Kn_f.T)
    e1 = np.max(np.abs(mn - mn_ref)) / np.max(np.abs(mn_ref)); e2 = np.max(np.abs(Kn - Kn_ref)) / np.max(np.abs(Kn_ref))
    print("gp_condition n=%d m=%d one workgroup: mn rel %.2e, Kn rel %.2e; vs blocked %.2e" %
          (n, m, e1, e2, np.max(np.abs(Kn - Kn_b)) / np.max(np.abs(Kn_ref))))
    assert e1 <= RTOL and e2 <= RTOL
    assert np.max(np.abs(mn - mn_b)) <= 1e-11 * np.max(np.abs(mn_ref)) and np.max(np.abs(Kn - Kn_b)) <= 1e-11 * np.max(np.abs(Kn_ref))


def test_default_thresholds_agree_across_the_switch_points(ctx, orc):
    """Defaults: one evaluation takes the one-workgroup kernel up to n = 128 and the launch chain beyond, a grid of
    >= 6 points the batched kernel up to n = 256: the results on either side of every switch agree to 1e-12."""
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    for n in (127, 128, 129, 159, 160, 161, 255, 256, 257):
        X, y = _case(n, 1, 11)
        single = ctx.logml(X, y, 1.0, [0.8], 0.1)
        out, info = ctx.logml_grid(X, y, np.ones(6), np.full(6, 0.8), np.full(6, 0.1))
        want = orc.logml(X, y, 1.0, 0.8, 0.1)
        assert np.all(info == 0) and abs(single[0] - want[0]) <= RTOL * abs(want[0])
        assert np.max(np.abs(out[:, 0] - single[0])) <= 1e-12 * abs(single[0]), (n, out[:, 0], single)


@pytest.mark.parametrize("n,D", [(60, 3), (256, 8), (700, 4), (300, 12)])
def test_ard_grid_equals_single_ard_evaluations(ctx, orc, n, D):
    """gpmi_logml_grid_ard: one length-scale per dimension and point (QQard's vector phi[[2]], R/kernels.R:11-19) --
    through the one-workgroup kernels (n <= 256, D <= 8: 32 points per launch with the parameters as kernel arguments; n = 700,
    40 points: parameters in device memory), the lanes and the LDS-tiled builder (D = 12): every point equals the single ARD evaluation bit for bit and the oracle to 1e-8."""
    X, y = _case(n, D, 21)
    G = 40
    rng = np.random.default_rng(n + D)
    ell = 0.5 + rng.random((G, D)) * 1.5; alpha = 0.8 + 0.4 * rng.random(G); sig = 0.1 + 0.2 * rng.random(G)
    alpha[7] = np.nan                 # a NaN amplitude: NaN covariance, "not positive definite", the others unaffected
    out, info = ctx.logml_grid_ard(X, y, alpha, ell, sig)
    assert info[7] > 0 and np.all(np.isnan(out[7])) and np.all(np.delete(info, 7) == 0)
    for g in (0, 6, 8, 31, 32, 39):
        single = ctx.logml(X, y, alpha[g], ell[g], sig[g])
        if 128 < n <= 1024 and D <= 8:   # the grid runs one workgroup per point, ONE evaluation of this size the launch chain: same numbers to rounding
            assert np.allclose(out[g], single, rtol=1e-11, atol=0), (g, out[g], single)
        else:
            assert tuple(out[g]) == tuple(single), (g, out[g], single)
    for g in (1, 33):
        want = orc.logml(X / ell[g], y, alpha[g], 1.0, sig[g])
        assert abs(out[g, 0] - want[0]) <= RTOL * abs(want[0])


@pytest.mark.parametrize("n,D,G", [(300, 1, 12), (512, 3, 40), (700, 2, 45), (1024, 3, 50), (1024, 8, 44)])
def test_mid_size_grid_one_workgroup_per_point_equals_the_blocked_path(ctx, orc, n, D, G):
    """Grids of at least 40 (n / 1024)^2 + 2 points at 256 < n <= 1024 run one workgroup per point with the parameters in
    device memory (k_logml_small_batch_dev): every point against the single evaluation through the launch chain (the
    same additions in the same order up to the trailing updates' tile shapes: 1e-11), two against the oracle, one
    point not positive definite."""
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    X, y = _case(n, D, 31)
    rng = np.random.default_rng(n + D + G)
    rho = rng.uniform(0.4, 2.0, G); sig = rng.uniform(0.05, 0.5, G); alpha = rng.uniform(0.5, 1.5, G)
    alpha[5] = np.nan
    out, info = ctx.logml_grid(X, y, alpha, rho, sig)
    assert info[5] > 0 and np.all(np.isnan(out[5])) and np.all(np.delete(info, 5) == 0)
    ctx.set_option("small_n2", 0)      # the same grid on the lanes of the blocked path
    try:
        ref, info_b = ctx.logml_grid(X, y, alpha, rho, sig)
    finally:
        ctx.set_option("small_n2", 1024)
    ok = np.arange(G) != 5
    assert np.all(info_b[ok] == 0)
    rel = np.max(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok]))
    assert rel <= 1e-11, rel
    for g in (0, G - 1):
        want = orc.logml(X, y, alpha[g], rho[g], sig[g])
        assert abs(out[g, 0] - want[0]) <= RTOL * abs(want[0]) and abs(out[g, 1] - want[1]) <= RTOL * abs(want[1])
    print("n=%d D=%d G=%d: one workgroup per point vs the blocked path %.1e" % (n, D, G, rel))


def test_mid_size_policy_and_large_point_counts(ctx, orc):
    """Below the break-even point count the lanes run (same numbers either way); a grid of more points than one launch
    holds workspace for (512) is cut into launches; an ARD grid of more than 32 points takes the device-parameter form."""
    X, y = _case(600, 2, 5)
    few, _ = ctx.logml_grid(X, y, np.ones(4), np.linspace(0.5, 1.0, 4), np.full(4, 0.2))        # 4 < 40 (600/1024)^2 + 2: lanes
    many, _ = ctx.logml_grid(X, y, np.ones(20), np.tile(np.linspace(0.5, 1.0, 4), 5), np.full(20, 0.2))
    assert np.max(np.abs(many[:4] - few) / np.abs(few)) <= 1e-11
    X, y = _case(40, 2, 6)
    G = 1100
    rng = np.random.default_rng(3)
    rho = rng.uniform(0.3, 2.0, G); sig = rng.uniform(0.05, 0.5, G)
    out, info = ctx.logml_grid(X, y, np.ones(G), rho, sig)
    assert np.all(info == 0)
    for g in (0, 511, 512, 1023, 1024, 1099):
        single = ctx.logml(X, y, 1.0, [rho[g]], sig[g])
        assert tuple(out[g]) == tuple(single), g
    X, y = _case(200, 3, 7)
    G = 70
    ell = 0.5 + rng.random((G, 3)); sig = 0.1 + 0.2 * rng.random(G)
    out, info = ctx.logml_grid_ard(X, y, np.ones(G), ell, sig)
    assert np.all(info == 0)
    for g in (0, 33, 69):
        want = orc.logml(X / ell[g], y, 1.0, 1.0, sig[g])
        assert abs(out[g, 0] - want[0]) <= RTOL * abs(want[0])


@pytest.mark.parametrize("n,m,B", [(25, 25, 3), (79, 40, 9), (199, 199, 12), (256, 300, 10)])
def test_sample_derivs_loop_one_workgroup_per_draw(ctx, orc, n, m, B):
    """The reference's derivative-imputation loop (pendulum_fit.R:261-268: 2 x 100 calls of sample_derivs, N = 199, each with
    its own (l, a, sy) and noisy series) as ONE launch, one workgroup per draw (k_sample_derivs_small_batch): moments against
    the reference's formulas with LU solves (pendulum_fit.R:242-251), the draw against numpy's Cholesky of that covariance
    (tolerance from its conditioning: jitter 1e-6), against the launch chains on the lanes, and a numerically singular
    draw that does not disturb the others."""
    rng = np.random.default_rng(n + m)
    t = np.sort(rng.uniform(0, n / 10.0, n)); tis = np.sort(rng.uniform(0, n / 10.0, m))
    P = np.column_stack([0.8 + 0.3 * rng.random(B), 1.0 + 0.4 * rng.random(B), 0.05 + 0.1 * rng.random(B)])
    Y = np.sin(t)[:, None] + 0.1 * rng.standard_normal((n, B)); Z = rng.standard_normal((m, B))
    jit = 1e-6
    ctx.set_option("small_sdb", 0)        # one workgroup per draw whatever the batch size
    try:
        draws, mus, info = ctx.sample_derivs_batch(t, tis, Y, P, jit, Z)
        Pbad = P.copy(); Pbad[1, 2] = 0.0; Pbad[1, 0] = 500.0     # sy = 0, huge length-scale: K is numerically singular
        dbad, _, ibad = ctx.sample_derivs_batch(t, tis, Y, Pbad, 0.0, Z)
    finally:
        ctx.set_option("small_sdb", 5)
    assert np.all(info == 0)
    assert ibad[1] != 0
    worst_mu = worst_d = 0.0
    for b in range(0, B, 3):
        l, a, sy = P[b]
        K = orc.deriv_cov("QQ", t, t, a, l) + sy * sy * np.eye(n)
        Ks = orc.deriv_cov("RQ", tis, t, a, l); Kss = orc.deriv_cov("RR", tis, tis, a, l)
        mu_ref = Ks @ np.linalg.solve(K, Y[:, b])
        cov_ref = Kss - Ks @ np.linalg.solve(K, Ks.T) + jit * np.eye(m)
        d_ref = mu_ref + np.linalg.cholesky(0.5 * (cov_ref + cov_ref.T)) @ Z[:, b]
        worst_mu = max(worst_mu, np.max(np.abs(mus[:, b] - mu_ref)) / np.max(np.abs(mu_ref)))
        worst_d = max(worst_d, np.max(np.abs(draws[:, b] - d_ref)) / np.max(np.abs(d_ref)))
    assert worst_mu <= RTOL and worst_d <= 1e-7, (worst_mu, worst_d)
    ctx.set_option("small_sd", 0)         # the same batch through the launch chains on the lanes
    try:
        dl, ml, il = ctx.sample_derivs_batch(t, tis, Y, P, jit, Z)
    finally:
        ctx.set_option("small_sd", 640)
    assert np.all(il == 0)
    assert np.max(np.abs(mus - ml)) <= 1e-11 * np.max(np.abs(ml)) and np.max(np.abs(draws - dl)) <= 1e-7 * np.max(np.abs(dl))
    print("sample_derivs n=%d m=%d B=%d, one workgroup per draw: mu rel %.1e, draw rel %.1e; vs the lanes mu %.1e draw %.1e"
          % (n, m, B, worst_mu, worst_d, np.max(np.abs(mus - ml)) / np.max(np.abs(ml)), np.max(np.abs(draws - dl)) / np.max(np.abs(dl))))


@pytest.mark.parametrize("kinds,compat", [(("QQ", "QQ", "QQ"), False), (("QQ", "RQ", "RR"), False), (("QQ", "RQ", "RR"), True),
                                          (("QQ", "TQ", "TT"), False)])
def test_gp_condition_in_one_launch(ctx, orc, kinds, compat):
    """gpmi_gp_condition at R/tests.R sizes is ONE launch (k_gp_condition_small: build, partial factorisation, mirrored
    Schur complement + jitter and mean written straight to mapped host memory): p_Xn (QQ, QQ, QQ), p_dotXn (QQ, RQ, RR; with
    R/kernels.R:31's amplitude slip as COMPAT_RR), second derivatives (TQ, TT) -- against the reference's formula with LU
    solves and against the launch chain bit for bit."""
    rng = np.random.default_rng(len(kinds[1]) + compat)
    n, m = 21, 30
    t = np.linspace(-2, 2, n); ts = np.sort(rng.uniform(-2, 2, m)); y = np.exp(t)
    a, l, s2, jit = 1.3, 0.8, 0.01, 1e-8
    from gp_amd._lib import COMPAT_RR
    flags = COMPAT_RR if compat else 0
    mn, Kn = ctx.gp_condition(t, ts, y, a, l, s2, jit, *kinds, flags=flags)
    ctx.set_option("small_gc", 0)
    try:
        mn_c, Kn_c = ctx.gp_condition(t, ts, y, a, l, s2, jit, *kinds, flags=flags)
    finally:
        ctx.set_option("small_gc", 180)
    assert np.array_equal(mn, mn_c) and np.array_equal(Kn, Kn_c)
    if not compat:
        K = orc.deriv_cov(kinds[0], t, t, a, l) + s2 * np.eye(n)
        Ks = orc.deriv_cov(kinds[1], ts, t, a, l); Kss = orc.deriv_cov(kinds[2], ts, ts, a, l)
        mn_ref = Ks @ np.linalg.solve(K, y); Kn_ref = Kss - Ks @ np.linalg.solve(K, Ks.T) + jit * np.eye(m)
        assert np.max(np.abs(mn - mn_ref)) <= RTOL * np.max(np.abs(mn_ref)) and np.max(np.abs(Kn - Kn_ref)) <= RTOL * np.max(np.abs(Kn_ref))


@pytest.mark.parametrize("n,D", [(1, 1), (30, 1), (100, 1), (256, 3), (257, 2), (700, 1)])
def test_exact_gp_transform_in_one_call(ctx, orc, n, D):
    """gpmi_exact_gp_f: f = cholesky_decompose(cov_exp_quad(x, alpha, rho) + jitter I) z (models/exact_gp.stan:17-25; the
    reference runs it at N = 100, test_interpolate.R:31-36) -- one launch of one workgroup up to n = 256, the device chain
    beyond -- against the oracle's Cholesky and against the composition of the separate entry points."""
    rng = np.random.default_rng(n)
    X = np.asfortranarray(rng.uniform(0, 1.0 + n / 8.0, (n, D))); z = rng.standard_normal(n)
    a, l, jit = 1.3, 0.6, 1e-6
    f = ctx.exact_gp_f(X, a, [l], z, jit)
    Lw = orc.cholesky(orc.cov_exp_quad(X, a, l) + jit * np.eye(n))
    want = Lw @ z
    assert np.max(np.abs(f - want)) <= 1e-8 * np.max(np.abs(want))     # cond(K + 1e-6 I) ~ 1e6
    K = ctx.se_cov(X, None, a, [l], diag_add=jit)
    comp = ctx.trmv_lower(ctx.potrf(K), z)
    assert np.max(np.abs(f - comp)) <= 1e-9 * np.max(np.abs(comp))
    if n == 30:
        from gp_amd import NotPositiveDefinite
        with pytest.raises(NotPositiveDefinite):
            ctx.exact_gp_f(np.zeros((n, 1)), 1.0, [1.0], z, 0.0)
