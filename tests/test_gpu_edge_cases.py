"""GPU: edge cases and error behaviour of the C ABI (empty / ragged inputs, bad arguments,
non-PD handling, fork detection, option handling)."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bad_arguments_are_status_codes(ctx):
    import gp_amd
    X = np.random.default_rng(0).random((10, 2)); y = np.zeros(10)
    with pytest.raises(gp_amd.GpmiError) as e:
        ctx.logml(X, np.zeros(9), 1.0, [0.3], 0.1)  # X and y disagree on N
    assert e.value.code == -1
    with pytest.raises(gp_amd.GpmiError):
        ctx.logml(X, y, 1.0, [0.3, 0.2, 0.1], 0.1)  # length-scale vector neither 1 nor D
    with pytest.raises(gp_amd.GpmiError):
        ctx.logml(X, y, 1.0, [0.0], 0.1)  # non-positive length-scale
    with pytest.raises(gp_amd.GpmiError):
        ctx.deriv_cov(11, [0.0], [0.0], 1.0, 1.0)  # unknown kernel kind
    with pytest.raises(gp_amd.GpmiError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(gp_amd.GpmiError):
        ctx.set_option("nb_outer", 100)  # not a multiple of 128
    with pytest.raises(gp_amd.GpmiError):
        ctx.joint_logml([0.0, 1.0], [0.0, 1.0, 2.0], 1.0, 1.0, 0.1)  # yy must have length 2N
    # the context is still healthy
    assert math.isfinite(ctx.logml(X, y + 1.0, 1.0, [0.3], 0.1)[0])


def test_empty_inputs(ctx):
    assert ctx.deriv_cov("QQ", [], [], 1.0, 1.0).shape == (0, 0)
    assert ctx.deriv_cov("RR", [0.0, 1.0], [], 1.0, 1.0).shape == (2, 0)
    assert ctx.deriv_elem("TT", [], [], 1.0).shape == (0,)
    assert ctx.joint_cov([], 1.0, 1.0, 0.1).shape == (0, 0)
    assert ctx.potrf(np.zeros((0, 0))).shape == (0, 0)
    out, info = ctx.logml_grid(np.zeros((5, 1)) + np.arange(5)[:, None], np.ones(5), [], [], [])
    assert out.shape == (0, 3) and info.shape == (0,)


def test_ard_length_scales_and_1d_inputs(ctx, orc):
    # ARD (length-D phi2) through the marginal likelihood; 1-D x given as a plain vector
    rng = np.random.default_rng(3)
    X = rng.random((90, 3)); y = rng.standard_normal(90)
    ell = [0.2, 0.5, 1.5]
    got = ctx.logml(X, y, 1.1, ell, 0.2)
    K = orc.QQard(X, X, 1.1, ell) + 0.04 * np.eye(90)
    L = np.linalg.cholesky(K); z = np.linalg.solve(L, y)
    want = -0.5 * z @ z - np.log(np.diag(L)).sum() - 45 * math.log(2 * math.pi)
    assert abs(got[0] - want) <= 1e-9 * abs(want)
    x = np.linspace(0, 4, 33)
    a = ctx.logml(x, np.sin(x), 1.0, [0.7], 0.1)
    b = ctx.logml(x.reshape(-1, 1), np.sin(x), 1.0, [0.7], 0.1)
    assert a == b


def test_every_pivot_position_reports_its_order(ctx):
    # the first non-positive pivot is reported 1-based, LAPACK style, wherever it sits in the
    # 16 x 16 / 128 x 128 / outer-block hierarchy
    import gp_amd
    for k in (1, 16, 17, 128, 129, 255, 256, 257, 400):
        n = k + 37
        A = np.eye(n) * 2.0
        A[k - 1, k - 1] = -1.0
        with pytest.raises(gp_amd.NotPositiveDefinite) as e:
            ctx.potrf(A)
        assert e.value.order == k, (k, e.value.order)
    # semi-definite (exactly singular) also fails like dpotrf
    B = np.ones((40, 40))
    with pytest.raises(gp_amd.NotPositiveDefinite) as e:
        ctx.potrf(B)
    assert e.value.order == 2


def test_nan_input_is_flagged_not_crashed(ctx):
    X = np.linspace(0, 1, 50).reshape(-1, 1); y = np.ones(50); y[7] = np.nan
    r = ctx.logml(X, y, 1.0, [0.3], 0.1)
    assert math.isnan(r[0]) or math.isnan(r[2])


def test_nan_coordinate_gives_nan_covariance_and_not_pd(ctx):
    # exp(NaN) is NaN in R (R/kernels.R:14) and Stan's cov_exp_quad rejects NaN inputs: a NaN point must not
    # silently become "infinitely far away" (covariance 0) through the clamped exp of the build kernel
    import gp_amd
    X = np.linspace(0, 1, 40).reshape(-1, 1).copy(); X[7, 0] = np.nan
    K = ctx.se_cov(X, None, 1.0, [0.3])
    off = ~np.eye(40, dtype=bool)
    assert np.all(np.isnan(K[7, off[7]])) and np.all(np.isnan(K[off[:, 7], 7]))
    assert np.all(np.isfinite(np.delete(np.delete(K, 7, 0), 7, 1)))
    with pytest.raises(gp_amd.NotPositiveDefinite):
        ctx.logml(X, np.ones(40), 1.0, [0.3], 0.1)
    Xb = np.random.default_rng(1).random((30, 12)); Xb[3, 11] = np.nan    # the D > 8 builder
    Kb = ctx.se_cov(Xb, None, 1.0, [0.5])
    assert np.isnan(Kb[3, 5]) and np.isnan(Kb[5, 3]) and np.isfinite(Kb[4, 5])


def test_context_is_rejected_in_forked_child(ctx):
    # HIP state does not survive fork(): the parent's context must answer GPMI_EFORK (-5) in a
    # child (the reference drivers fork with parallel::mclapply, pendulum_fit.R:268)
    from gp_amd import _lib
    h = _lib.load()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            import ctypes
            rc = h.gpmi_sync(ctx._h)
            # a NEW context cannot be made in the child either (HIP cannot be re-initialised there):
            # gpmi_create and gpmi_device_count must answer GPMI_EFORK without touching the runtime
            h2 = ctypes.c_void_p()
            rc2 = h.gpmi_create(ctypes.byref(h2), 0)
            cnt = ctypes.c_int(-1)
            rc3 = h.gpmi_device_count(ctypes.byref(cnt))
            os.write(w, ("%d %d %d %d %d" % (rc, rc2, rc3, cnt.value, int(bool(h2.value)))).encode())
        finally:
            os._exit(0)
    os.close(w)
    os.waitpid(pid, 0)
    got = [int(v) for v in (os.read(r, 64).decode() or "0 0 0 0 0").split()]
    os.close(r)
    assert got == [-5, -5, -5, 0, 0], got
    assert math.isfinite(ctx.logml(np.arange(5.0), np.ones(5), 1.0, [1.0], 0.5)[0])  # parent unaffected


def test_options_do_not_change_results_beyond_rounding(ctx, orc):
    X, y = orc.synth(900, 3)
    base = ctx.logml(X, y, 1.0, [0.3], 0.1)[0]
    vals = []
    for opt, v in (("nb_outer", 128), ("nb_outer", 512), ("ksplit", 0), ("stagger", 0), ("se_nt", 0), ("nb_adapt", 0),
                   ("fuse_diag", 0), ("fuse_diag", 3), ("block_recursive", 0), ("small_m", 1024)):
        ctx.set_option(opt, v)
        try:
            vals.append(ctx.logml(X, y, 1.0, [0.3], 0.1)[0])
        finally:
            ctx.set_option("nb_outer", 0); ctx.set_option("ksplit", 1); ctx.set_option("stagger", (2 << 16) | 4)
            ctx.set_option("se_nt", 1); ctx.set_option("nb_adapt", 1); ctx.set_option("fuse_diag", 15)
            ctx.set_option("block_recursive", 1); ctx.set_option("small_m", 160)
    assert all(abs(v - base) <= 1e-10 * abs(base) for v in vals), (base, vals)


def test_options_are_per_context(ctx, orc):
    # tuning lives in the context: a second context keeps its defaults while the first one is re-tuned
    import gp_amd
    X, y = orc.synth(700, 2)
    other = gp_amd.Context(0)
    try:
        ctx.set_option("nb_outer", 128); ctx.set_option("fuse_diag", 0); ctx.set_option("block_recursive", 0)
        a = ctx.logml(X, y, 1.0, [0.3], 0.1)[0]
        b = other.logml(X, y, 1.0, [0.3], 0.1)[0]
    finally:
        ctx.set_option("nb_outer", 0); ctx.set_option("fuse_diag", 15); ctx.set_option("block_recursive", 1)
    c = ctx.logml(X, y, 1.0, [0.3], 0.1)[0]
    other.close()
    assert b == c                       # `other` ran the default algorithm, bit for bit
    assert abs(a - c) <= 1e-10 * abs(c)  # the re-tuned one is a re-ordering of the same sums


def test_grid_lanes_match_single_evaluations(ctx, orc):
    X, y = orc.synth(700, 2)
    rho = np.geomspace(0.1, 1.0, 7); sig = np.geomspace(0.05, 0.5, 7)
    for lanes in (1, 2, 4, 8):
        ctx.set_option("grid_lanes", lanes)
        try:
            out, info = ctx.logml_grid(X, y, 1.0, rho, sig)
        finally:
            ctx.set_option("grid_lanes", 0)
        assert np.all(info == 0)
        for g in (0, 3, 6):
            assert out[g, 0] == ctx.logml(X, y, 1.0, [rho[g]], sig[g])[0]  # bit-identical


@pytest.mark.parametrize("n,nbo", [(2500, 1024), (1333, 512), (1153, 0)])
def test_fused_diagonal_modes_agree(ctx, orc, n, nbo):
    """The fused look-ahead variants (diagonal block factored inside the update that completes it;
    its tile cut into three sub-tiles on three CUs; option fuse_diag bits 0..3: in-block products, multi-round and single-round trailing updates) and the recursive /
    fixed-level in-block blockings are re-orderings of the same factorisation: every combination
    must give the LAPACK result (1e-10 relative on logml, far inside the 1e-8 bar), also with
    ragged last panels and a forced 1024-wide outer block (K = 512 sub-tiles)."""
    import scipy.linalg as sla
    X, y = orc.synth(n, 2, seed=n)
    K = orc.QQard(X, X, 1.0, [0.2]) + 0.01 * np.eye(n)
    L = sla.cholesky(K, lower=True); z = sla.solve_triangular(L, y, lower=True)
    want = -0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * math.log(2 * math.pi)
    try:
        ctx.set_option("nb_outer", nbo)
        for rec in (1, 0):
            ctx.set_option("block_recursive", rec)
            for mode in (0, 1, 2, 3, 7, 8, 15):
                ctx.set_option("fuse_diag", mode)
                got = ctx.logml(X, y, 1.0, [0.2], 0.1)
                assert abs(got[0] - want) <= 1e-10 * abs(want), (rec, mode, got[0], want)
        # a non-PD matrix is still reported at the right order through the fused paths
        ctx.set_option("fuse_diag", 15); ctx.set_option("block_recursive", 1)
        A = K.copy(); k = 1100 if n > 1200 else 300
        A[k, k] = -1.0
        import gp_amd
        with pytest.raises(gp_amd.NotPositiveDefinite) as e:
            ctx.potrf(A)
        assert e.value.order == k + 1
    finally:
        ctx.set_option("nb_outer", 0); ctx.set_option("fuse_diag", 15); ctx.set_option("block_recursive", 1)


def test_far_and_infinite_points_have_zero_covariance(ctx):
    """exp(-inf) = 0 in R and Stan: a point whose squared scaled distance overflows, or with an infinite coordinate, is
    infinitely far away (covariance exactly 0, its own variance alpha^2), not an error; only a NaN coordinate is NaN."""
    X = np.array([[0.0], [1.0], [1e200], [np.inf], [2.0]])
    K = ctx.se_cov(X, None, 1.5, [1e-3], diag_add=0.25)
    assert np.all(np.isfinite(K[:3, :3])) and np.all(K[2, [0, 1, 4]] == 0.0) and np.all(K[[0, 1, 4], 2] == 0.0)
    assert np.all(np.diag(K) == 1.5 ** 2 + 0.25)
    assert np.all(K[3, [0, 1, 2, 4]] == 0.0)                  # the infinite point: exp(-inf) = 0 against every finite one
    Kr = ctx.se_cov(X[:3], X[[4]], 1.0, [1.0])
    assert Kr[2, 0] == 0.0 and np.all(np.isfinite(Kr))
    # ... and the evaluation goes through: the far point decouples (its own 1 x 1 block)
    y = np.array([0.1, -0.2, 0.3])
    full = ctx.logml(X[:3], y, 1.0, [0.5], 0.3)[0]
    parts = ctx.logml(X[:2], y[:2], 1.0, [0.5], 0.3)[0] + ctx.logml(X[2:3], y[2:], 1.0, [0.5], 0.3)[0]
    assert abs(full - parts) <= 1e-13 * abs(full)
    Xn = X.copy(); Xn[1, 0] = np.nan
    assert np.isnan(ctx.se_cov(Xn, None, 1.0, [1.0])[1, 0])


def test_create_unwinds_on_every_failing_step_without_leaking(ctx):
    """gpmi_create acquires a stream, five events, six device buffers and four timing events; when step k fails
    (GPMI_FAIL_CREATE_AT = k: the test hook makes the k-th acquisition report an allocation failure) everything
    acquired before it is released again: the free device memory after 20 rounds of failing creates at every step is
    what it was before, and a normal create still works.  KTimer events of a context are released by gpmi_destroy."""
    import os
    import torch
    import gp_amd
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    try:
        for _ in range(20):
            for k in range(1, 18):
                os.environ["GPMI_FAIL_CREATE_AT"] = str(k)
                with pytest.raises(gp_amd.GpmiError) as e:
                    gp_amd.Context(0)
                assert e.value.code in (-3, -2)
    finally:
        os.environ.pop("GPMI_FAIL_CREATE_AT", None)
    other = gp_amd.Context(0)
    other.set_option("kernel_timing", 1)
    X = np.linspace(0, 1, 300).reshape(-1, 1)
    assert math.isfinite(other.logml(X, np.sin(X[:, 0]), 1.0, [0.3], 0.1)[0])
    other.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 <= 8 << 20, (free0, free1)   # nothing accumulates (allocator granularity aside)
