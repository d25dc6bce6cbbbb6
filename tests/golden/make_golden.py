"""Generate the committed golden fixtures under tests/golden/.

Run ONLY in the build container (needs /root/reference):

    MPLBACKEND=Agg python tests/golden/make_golden.py

Two sources, both data only (no reference source text is stored):

1. gp_derivs.json  -- outputs of the reference's own Python restatement of the
   derivative kernels and GP posterior, /root/reference/gp_derivs.py, executed
   with runpy (it is a script, not a module): the nine kernels QQ..TT
   (gp_derivs.py:15-40) on a small (tj, tk, l, a) grid, and K / KsK / KsKs /
   mu / cov for its N = 25 pendulum data set (gp_derivs.py:59-113).
2. ch2.json -- outputs of the two GP cells of /root/reference/ch2.py (:16-43 and :58-85), each cell
   executed in memory as the reference wrote it (Agg backend, seeded numpy.random): its own kernel
   `makeK` (eta2 exp(-(x1-x2)^2 / l2): alpha^2 = eta2, rho^2 = l2 / 2) at N = 1000, the data block
   Kdd, the posterior mean m = Ksd Kdd^-1 f (numpy.linalg.solve, i.e. LAPACK dgesv -- what base-R
   solve() is) and strided samples of Kss and of the posterior covariance Kt.  An SE build at 40x
   the gp_derivs.py size and a noise-free (sigma2 = 0) posterior.  Since round 3 also the factor the cell
   computes, L = numpy.linalg.cholesky(Kt + 1e-10 I) (:42, :84; the only Cholesky the reference can execute
   here): its diagonal, three rows, sum log diag and cond(Kt + 1e-10 I) = 2.7e9 / 2.5e12.
3. kat.json -- closed-form known-answer tests for the marginal-likelihood path
   (models/fit_hyperparameters.stan:18-32), computed with LAPACK (scipy
   dpotrf/dtrtrs) and confirmed with mpmath at 50 digits.  The reference has no
   executable Stan/R here, so these pin the oracle's restatement of that path
   independently of the oracle's own code.
"""
import json
import os
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gp_derivs.py"
REF_CH2 = "/root/reference/ch2.py"


def gen_gp_derivs():
    os.environ.setdefault("MPLBACKEND", "Agg")
    ns = runpy.run_path(REF)
    g = ns["QQ"].__globals__
    names = ["QQ", "QR", "RQ", "RR", "QT", "TQ", "RT", "TR", "TT"]
    pts = [(0.3, 1.1), (1.1, 0.3), (0.0, 0.0), (-2.0, 0.5), (4.75, 5.0), (2.5, -1.25)]
    cases = []
    for (l, a) in [(1.0, 1.0), (0.5, 1.0), (0.7, 1.3), (2.0, 0.4)]:
        g["l"], g["a"] = l, a
        for (tj, tk) in pts:
            cases.append({"l": l, "a": a, "tj": tj, "tk": tk,
                          "out": {n: float(ns[n](tj, tk)) for n in names}})
    # the script's own N = 25 posterior at its module defaults l = 1, a = 1, s = 0.1
    g["l"], g["a"], g["s"] = 1.0, 1.0, 0.1
    ts = np.asarray(ns["ts"]); y = np.asarray(ns["y"])
    cov_builder = None
    # 'cov' is rebound at gp_derivs.py:104 to the posterior-covariance function;
    # rebuild the kernel matrices with a local double loop over the script's kernels.
    def build(f, xs, ys):
        return np.array([[f(a_, b_) for b_ in ys] for a_ in xs])
    K = build(ns["QQ"], ts, ts)
    KsK_d = build(ns["RQ"], ts, ts)
    KsKs_d = build(ns["RR"], ts, ts)
    mu_q = ns["mu"](K, K, y)
    cov_q = ns["cov"](K, K, K)
    mu_d = ns["mu"](K, KsK_d, y)
    cov_d = ns["cov"](K, KsK_d, KsKs_d)
    # the script's own stored matrices (value posterior and second-derivative one)
    out = {
        "source": "runpy of /root/reference/gp_derivs.py (functions :15-40, mu/cov :97-113)",
        "kernel_cases": cases,
        "posterior": {
            "l": 1.0, "a": 1.0, "s": 0.1,
            "ts": ts.tolist(), "y": y.tolist(),
            "K": np.asarray(ns["K"]).tolist(),
            "KsKi_TQ": np.asarray(ns["KsKi"]).tolist(),
            "KsKsi_TT": np.asarray(ns["KsKsi"]).tolist(),
            "mu_value": np.asarray(mu_q).tolist(),
            "cov_value": np.asarray(cov_q).tolist(),
            "mu_deriv": np.asarray(mu_d).tolist(),
            "cov_deriv": np.asarray(cov_d).tolist(),
            "mu_second": np.asarray(ns["mu"](K, np.asarray(ns["KsKi"]), y)).tolist(),
            "cov_second": np.asarray(ns["cov"](K, np.asarray(ns["KsKi"]), np.asarray(ns["KsKsi"]))).tolist(),
        },
    }
    with open(os.path.join(HERE, "gp_derivs.json"), "w") as f:
        json.dump(out, f)
    print("wrote gp_derivs.json:", len(cases), "kernel cases; K[0,1] =", repr(out["posterior"]["K"][0][1]))


def gen_ch2():
    """The reference script is a sequence of `#%%` cells that re-bind the same names; each GP cell
    is executed on its own (after the import cell) so that ITS values are the ones captured."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    import matplotlib
    matplotlib.use("Agg")
    cells = open(REF_CH2).read().split("#%%")
    gp_cells = [c for c in cells if "makeK" in c]
    assert len(gp_cells) == 2, len(gp_cells)
    out = {"source": "cells of /root/reference/ch2.py executed in memory (:16-43, :58-85); data only", "cells": []}
    for cell in gp_cells:
        ns = {}
        exec(compile(cells[1], REF_CH2, "exec"), ns)      # the import cell
        ns["numpy"].random.seed(1234)
        exec(compile(cell, REF_CH2, "exec"), ns)
        N = int(ns["N"])
        ri = list(range(0, N, 37)); ci = list(range(0, N, 41))
        Kss = np.asarray(ns["Kss"]); Kt = np.asarray(ns["Kt"])
        out["cells"].append({
            "N": N, "eta2": float(ns["eta2"]), "l2": float(ns["l2"]), "sigma2": float(ns["sigma2"]),
            "xd": np.asarray(ns["xd"]).tolist(), "f": np.asarray(ns["f"]).tolist(),
            "Kdd": np.asarray(ns["Kdd"]).tolist(),
            "Ksd_rows": ri, "Ksd": np.asarray(ns["Ksd"])[ri, :].tolist(),
            "m": np.asarray(ns["m"]).tolist(),
            "rows": ri, "cols": ci,
            "Kss_sample": Kss[np.ix_(ri, ci)].tolist(),
            "Kt_sample": Kt[np.ix_(ri, ci)].tolist(),
            "Kt_diag": np.diag(Kt).tolist(),
            # the ONLY Cholesky factor the reference itself computes here: ch2.py:42 / :84,
            # L = numpy.linalg.cholesky(Kt + 1e-10 I) (LAPACK dpotrf) -- its diagonal, three rows, the
            # log-determinant half and the condition number of the factored matrix (the forward error
            # bound of any backward-stable Cholesky is cond * eps: the tests derive their tolerance from it)
            "L_jitter": 1e-10,
            "L_diag": np.diag(np.asarray(ns["L"])).tolist(),
            "L_sum_log_diag": float(np.log(np.diag(np.asarray(ns["L"]))).sum()),
            "L_rows_idx": [1, N // 2, N - 1],
            "L_rows": np.asarray(ns["L"])[[1, N // 2, N - 1], :].tolist(),
            "L_cond": float((lambda ev: ev[-1] / ev[0])(np.linalg.eigvalsh(Kt + 1e-10 * np.eye(N)))),
        })
        print("ch2 cell: N =", N, "eta2 =", ns["eta2"], "sigma2 =", ns["sigma2"], "m[500] =", repr(float(ns["m"][500])))
    with open(os.path.join(HERE, "ch2.json"), "w") as f:
        json.dump(out, f)


def _logml_lapack(x, y, alpha, rho, sigma, jitter=0.0):
    from scipy.linalg import cholesky, solve_triangular
    x = np.asarray(x, dtype=np.float64).reshape(len(y), -1)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    K = alpha ** 2 * np.exp(-0.5 * d2 / rho ** 2) + (sigma ** 2 + jitter) * np.eye(len(y))
    L = cholesky(K, lower=True)
    z = solve_triangular(L, y, lower=True)
    sld = np.log(np.diag(L)).sum()
    return float(-0.5 * z @ z - sld - 0.5 * len(y) * np.log(2 * np.pi)), float(sld), float(z @ z), float(L[1, 0])


def _logml_mpmath(x, y, alpha, rho, sigma):
    import mpmath as mp
    mp.mp.dps = 50
    n = len(y)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            r = mp.mpf(float(x[i])) - mp.mpf(float(x[j]))
            K[i, j] = mp.mpf(alpha) ** 2 * mp.exp(-r * r / (2 * mp.mpf(rho) ** 2))
        K[i, i] += mp.mpf(sigma) ** 2
    L = mp.cholesky(K)
    z = mp.lu_solve(L, mp.matrix([mp.mpf(float(v)) for v in y]))
    sld = sum(mp.log(L[i, i]) for i in range(n))
    q = sum(z[i] * z[i] for i in range(n))
    return float(-q / 2 - sld - mp.mpf(n) / 2 * mp.log(2 * mp.pi)), float(sld), float(q)


def gen_kat():
    kats = []
    # K1: the R/tests.R:5 grid t = -2,-1.8,...,2 (N = 21), y = exp(t) noise-free
    t = np.round(np.arange(-2.0, 2.0 + 1e-9, 0.2), 10)
    y = np.exp(t)
    lap = _logml_lapack(t, y, 1.0, 1.0, 0.05)
    mpv = _logml_mpmath(t, y, 1.0, 1.0, 0.05)
    kats.append({"name": "K1", "x": t.tolist(), "y": y.tolist(), "alpha": 1.0, "rho": 1.0, "sigma": 0.05,
                 "logml": mpv[0], "sum_log_diag": mpv[1], "quad": mpv[2], "L10": lap[3],
                 "logml_lapack": lap[0]})
    # K2: c1-shaped, x = linspace(0,10,256), y = sin(x)
    x = np.linspace(0, 10, 256); y2 = np.sin(x)
    lap2 = _logml_lapack(x, y2, 1.0, 1.0, 0.1)
    kats.append({"name": "K2", "x": x.tolist(), "y": y2.tolist(), "alpha": 1.0, "rho": 1.0, "sigma": 0.1,
                 "logml": lap2[0], "sum_log_diag": lap2[1], "quad": lap2[2], "L10": lap2[3],
                 "logml_lapack": lap2[0]})
    # K3: D = 3, N = 64 random (fixed seed), mpmath-confirmed via 1-D trick not available -> LAPACK
    rng = np.random.default_rng(12345)
    X3 = rng.random((64, 3)); y3 = np.sin(2 * np.pi * X3.sum(1)) + 0.1 * rng.standard_normal(64)
    lap3 = _logml_lapack(X3, y3, 1.2, 0.4, 0.15)
    kats.append({"name": "K3", "x": X3.tolist(), "y": y3.tolist(), "alpha": 1.2, "rho": 0.4, "sigma": 0.15,
                 "logml": lap3[0], "sum_log_diag": lap3[1], "quad": lap3[2], "L10": lap3[3],
                 "logml_lapack": lap3[0]})
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump({"source": "scipy LAPACK dpotrf/dtrtrs; K1 confirmed with mpmath at 50 digits", "kats": kats}, f)
    for k in kats:
        print(k["name"], repr(k["logml"]), repr(k["sum_log_diag"]), repr(k["L10"]))


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present; fixtures are committed, nothing to do")
    gen_gp_derivs()
    gen_ch2()
    gen_kat()
