"""GPU: Cholesky-factor interpolation over the length-scale -- approx_L (covariance.cpp:49-96),
approx_Lz (models/cubic_interpolated_gp.hpp:38-73) and the table build of test_interpolate.R:9-19."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(orc, x, lp):
    Ls, dLs = zip(*[orc.rbf_cov_chol(x, l) for l in lp])
    return list(Ls), list(dLs)


def test_blend_of_a_loaded_table_is_the_oracles(ctx, orc):
    x = np.linspace(0, 5, 40); lp = np.array([0.5, 0.7, 0.9, 1.2])
    Ls, dLs = _table(orc, x, lp)
    ctx.interp_load(lp, Ls, dLs)
    z = np.sin(0.3 * np.arange(40.0))
    for l in (0.5, 0.55, 0.7, 0.81, 1.19, 1.2):
        got = ctx.approx_L(l)
        want = orc.approx_L(l, lp, Ls, dLs)
        assert np.max(np.abs(got - want)) <= 4e-16 * np.max(np.abs(want)), l  # same formula, same operation order
        assert np.all(np.triu(got, 1) == 0.0)
        np.testing.assert_allclose(ctx.approx_Lz(l, z), orc.approx_Lz(l, lp, Ls, dLs, z), rtol=1e-13, atol=1e-14)
    # outside the table the last / first interval is extrapolated (the reference reads past the end there)
    for l in (0.3, 1.5):
        np.testing.assert_allclose(ctx.approx_L(l), orc.approx_L(l, lp, Ls, dLs), rtol=1e-14, atol=1e-15)
    ctx.interp_free()


def test_table_built_on_the_device(ctx, orc):
    # gpmi_interp_build == P calls of rbf_cov_chol (test_interpolate.R:9-19).  With the reference's
    # 1e-10 jitter the factor of a densely sampled kernel is defined only to ~cond * eps, so the grid
    # here is sparse enough (cond ~ 1e4) for two fp64 factorisations to agree to 1e-10
    x = np.linspace(0, 8, 12); lp = np.linspace(0.6, 1.1, 5)
    ctx.interp_build(x, lp)
    Ls, dLs = _table(orc, x, lp)
    for l in (0.6, 0.72, 1.03):
        got = ctx.approx_L(l)
        want = orc.approx_L(l, lp, Ls, dLs)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    # at a knot the blend is the knot's own factor
    Lk, _ = ctx.rbf_cov_chol(x, lp[2])
    np.testing.assert_allclose(ctx.approx_L(lp[2]), Lk, rtol=0, atol=1e-12)
    ctx.interp_free()


def test_many_chunks_and_row_blocks(ctx, orc):
    # indexing at a size with several 256-row blocks and 128-column chunks, ragged edges included;
    # the table need not hold Cholesky factors for that
    rng = np.random.default_rng(5)
    n = 1111
    lp = np.array([1.0, 2.0, 2.5])
    Ls = [np.tril(rng.standard_normal((n, n))) for _ in lp]
    dLs = [np.tril(rng.standard_normal((n, n))) for _ in lp]
    ctx.interp_load(lp, Ls, dLs)
    z = rng.standard_normal(n)
    for l in (1.3, 2.2):
        want_L = orc.approx_L(l, lp, Ls, dLs)
        got_L = ctx.approx_L(l)
        assert np.max(np.abs(got_L - want_L)) <= 1e-15 * np.max(np.abs(want_L))
        got = ctx.approx_Lz(l, z)
        np.testing.assert_allclose(got, want_L @ z, rtol=0, atol=1e-12 * np.abs(want_L).sum(axis=1).max())
        assert np.array_equal(got, ctx.approx_Lz(l, z))  # fixed summation order: run-to-run identical
    ctx.interp_free()


def test_host_mirror_and_errors(ctx, orc):
    import gp_amd
    from gp_amd.covariance import approx_L, approx_Lz, FactorInterpolator
    x = np.linspace(0, 2, 12); lp = [0.5, 0.8, 1.3]
    Ls, dLs = _table(orc, x, lp)
    z = np.arange(12.0)
    np.testing.assert_allclose(approx_L(0.9, lp, Ls, dLs, ctx=ctx), orc.approx_L(0.9, lp, Ls, dLs), rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(approx_Lz(0.9, lp, Ls, dLs, z, ctx=ctx), orc.approx_Lz(0.9, lp, Ls, dLs, z), rtol=1e-13)
    fi = FactorInterpolator(x, lp, ctx=ctx)
    assert fi.L(0.7).shape == (12, 12) and fi.Lz(0.7, z).shape == (12,)
    ctx.interp_free()
    with pytest.raises(gp_amd.GpmiError):
        ctx.approx_L(0.7)  # no table
    with pytest.raises(gp_amd.GpmiError):
        ctx.interp_load([0.5], Ls[:1], dLs[:1])  # one knot is no interval
    with pytest.raises(gp_amd.GpmiError):
        ctx.interp_load([0.8, 0.5], Ls[:2], dLs[:2])  # not increasing
    with pytest.raises(gp_amd.GpmiError):
        ctx.interp_build(x, [0.5, -1.0 + 0.5, 0.9])  # not increasing / non-positive


def _reference_grid():
    # test_interpolate.R:5-9: x = seq(0, 10, length = N), lp = seq(qgamma(.05, 4, 4), qgamma(.95, 4, 4), length = 10)
    from scipy.stats import gamma
    return np.linspace(gamma.ppf(0.05, 4.0, scale=0.25), gamma.ppf(0.95, 4.0, scale=0.25), 10)


def test_approx_Lz_grad_vs_oracle_on_the_reference_grid(ctx, orc):
    """dfdl = (dv/dl) z, the partial the `var` overload of build_output attaches to approx_Lz
    (models/cubic_interpolated_gp.hpp:6-32, dvdl :67), on the P = 10 length-scale grid of
    test_interpolate.R:9 -- against the oracle's restatement (pinned by central differences in
    tests/test_oracle.py) and, directly, against central differences of the device's own value."""
    lp = _reference_grid()
    x = np.linspace(0.0, 10.0, 30)
    Ls, dLs = _table(orc, x, lp)
    ctx.interp_load(lp, Ls, dLs)
    z = np.cos(0.7 * np.arange(30.0))
    for l in (lp[0], 0.5 * (lp[0] + lp[1]), lp[3] + 0.3 * (lp[4] - lp[3]), lp[5], lp[8] + 0.9 * (lp[9] - lp[8]), lp[9], 0.2, 2.5):
        f, g = ctx.approx_Lz_grad(l, z)
        fo, go = orc.approx_Lz_grad(l, lp, Ls, dLs, z)
        np.testing.assert_allclose(f, fo, rtol=1e-13, atol=1e-14)
        assert np.max(np.abs(g - go)) <= 1e-12 * np.max(np.abs(go)), l
        assert np.array_equal(f, ctx.approx_Lz(l, z))        # the value is the value-only call's, bit for bit
    l = lp[6] + 0.4 * (lp[7] - lp[6]); h = 1e-6
    _, g = ctx.approx_Lz_grad(l, z)
    fd = (ctx.approx_Lz(l + h, z) - ctx.approx_Lz(l - h, z)) / (2 * h)
    assert np.max(np.abs(g - fd)) <= 1e-8 * np.max(np.abs(g))
    # host mirror
    from gp_amd.covariance import approx_Lz_grad
    f2, g2 = approx_Lz_grad(l, lp, Ls, dLs, z, ctx=ctx)
    assert np.array_equal(g2, g)
    ctx.interp_free()


def test_approx_Lz_grad_many_chunks(ctx, orc):
    rng = np.random.default_rng(11)
    n = 1111
    lp = np.array([1.0, 2.0, 2.5])
    Ls = [np.tril(rng.standard_normal((n, n))) for _ in lp]
    dLs = [np.tril(rng.standard_normal((n, n))) for _ in lp]
    ctx.interp_load(lp, Ls, dLs)
    z = rng.standard_normal(n)
    for l in (1.3, 2.2):
        f, g = ctx.approx_Lz_grad(l, z)
        fo, go = orc.approx_Lz_grad(l, lp, Ls, dLs, z)
        scale = np.abs(np.stack(Ls + dLs)).sum(axis=2).max()
        assert np.max(np.abs(f - fo)) <= 1e-12 * scale and np.max(np.abs(g - go)) <= 1e-11 * scale
        assert np.array_equal(g, ctx.approx_Lz_grad(l, z)[1])  # fixed summation order
    ctx.interp_free()


def test_table_built_on_lanes_equals_single_factorisations(ctx, orc):
    """gpmi_interp_build factors the P entries concurrently on the grid lanes (entry p on lane p mod 4):
    every entry must equal a stand-alone gpmi_rbf_cov_chol of the same length-scale bit for bit, at a
    size with several panels, for P not a multiple of the lane count, and with lanes = 1."""
    n = 700
    x = 1.2 * np.arange(n) + 0.2 * np.sin(np.arange(n))
    lp = np.linspace(0.7, 1.0, 7)
    singles = [ctx.rbf_cov_chol(x, l) for l in lp]
    for lanes in (0, 1, 3):
        ctx.set_option("grid_lanes", lanes)
        try:
            ctx.interp_build(x, lp)
        finally:
            ctx.set_option("grid_lanes", 0)
        for p, l in enumerate(lp):
            assert np.array_equal(ctx.approx_L(l), singles[p][0]), (lanes, p)   # t = 0 or 1: the knot itself
        z = np.sin(np.arange(n) * 0.1)
        _, g = ctx.approx_Lz_grad(lp[2] + 1e-13, z)   # just right of knot 2: dv/dl = dLdl[2]
        np.testing.assert_allclose(g, singles[2][1] @ z, rtol=0, atol=1e-8 * np.max(np.abs(g)))
    ctx.interp_free()
    # a failed factorisation (NaN input: no positive pivot) is reported through the lanes as well
    import gp_amd
    xb = np.arange(40.0); xb[7] = np.nan
    with pytest.raises(gp_amd.NotPositiveDefinite):
        ctx.interp_build(xb, [0.5, 0.8, 0.9])
    ctx.interp_free()


@pytest.mark.parametrize("n,P", [(5, 3), (50, 10), (100, 10), (128, 70)])
def test_small_table_build_in_one_launch(ctx, orc, n, P):
    """n <= 128 (test_interpolate.R:5-9 runs N = 100, P = 10): the interpolation table is built by ONE launch per 64 entries, one
    workgroup per length-scale (k_rbf_cov_chol_small: build, factor, tangent L Phi(L^-1 Sdot L^-T)), written straight into the
    table: every entry against the oracle's rbf_cov_chol, and a single call (one workgroup up to n = 64, the launch chain beyond)
    against the same."""
    x = np.linspace(0.0, 0.5 * n, n)       # about one point per length-scale: well conditioned with the 1e-10 jitter
    lp = np.linspace(0.3, 0.55, P)
    ctx.interp_build(x, lp)
    z = np.random.default_rng(n).standard_normal(n)
    for p in (0, P // 2, P - 1):
        L, dL = orc.rbf_cov_chol(x, lp[p])
        f = ctx.approx_Lz(lp[p], z)            # at a knot the blend is the tabulated factor itself
        assert np.max(np.abs(f - L @ z)) <= 1e-9 * max(1.0, np.max(np.abs(L @ z)))
        Lg, dLg = ctx.rbf_cov_chol(x, lp[p])
        np.testing.assert_allclose(Lg, L, rtol=0, atol=1e-11)
        np.testing.assert_allclose(dLg, dL, rtol=0, atol=1e-9 * max(1.0, np.abs(dL).max()))
        assert np.all(np.triu(Lg, 1) == 0) and np.all(np.triu(dLg, 1) == 0)
    if P >= 10:   # dv/dl at a knot is the tabulated tangent
        _, dfdl = ctx.approx_Lz_grad(lp[1], z)
        _, dL = orc.rbf_cov_chol(x, lp[1])
        assert np.max(np.abs(dfdl - dL @ z)) <= 1e-7 * max(1.0, np.max(np.abs(dL @ z)))
    ctx.interp_free()
