"""Random-size sweep of the one-workgroup kernels against independent values: grids (n <= 1024) and value + gradient (n <= 256)
against the references numpy / LAPACK provide (Cholesky log marginal likelihood; gradient by the dense trace
formula), sample_derivs batches against LU-solve moments.  Exits non-zero above the tolerances."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
ctx = gp_amd.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {"grid": 0.0, "grad": 0.0, "sd_mu": 0.0, "sd_draw": 0.0}

def kern(X, a, ell):
    Z = X / ell
    d2 = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
    return a * a * np.exp(-0.5 * d2), d2

def ref_logml_grad(X, y, a, ell, s):
    n = len(y)
    K0, _ = kern(X, a, ell)
    K = K0 + s * s * np.eye(n)
    L = np.linalg.cholesky(K)
    z = np.linalg.solve(L, y)
    val = -0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * math.log(2 * math.pi)
    Ki = np.linalg.inv(K); al = Ki @ y
    Wm = np.outer(al, al) - Ki
    g = [0.5 * np.sum(Wm * (2.0 / a) * K0)]
    for d in range(X.shape[1]):
        r2 = (X[:, None, d] - X[None, :, d]) ** 2
        g.append(0.5 * np.sum(Wm * K0 * r2 / ell[d] ** 3))
    g.append(0.5 * np.sum(np.diag(Wm)) * 2.0 * s)
    return val, np.array(g)

for it in range(40):
    n = int(rng.integers(1, 1025)); D = int(rng.integers(1, 9)); G = int(rng.integers(1, 70))
    X = rng.random((n, D)) * (1.0 + n / 50.0) ** (1.0 / D); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    al = 0.7 + 0.6 * rng.random(G); rho = 0.3 + rng.random(G); sg = 0.08 + 0.3 * rng.random(G)
    out, info = ctx.logml_grid(X, y, al, rho, sg)
    assert np.all(info == 0), (n, D, G, info)
    for g in rng.choice(G, size=min(G, 3), replace=False):
        want, _ = ref_logml_grad(X, y, al[g], np.full(D, rho[g]), sg[g])
        worst["grid"] = max(worst["grid"], abs(out[g, 0] - want) / abs(want))
for it in range(40):
    n = int(rng.integers(1, 257)); D = int(rng.integers(1, 9)); G = int(rng.integers(1, 7))
    X = rng.random((n, D)) * (1.0 + n / 50.0) ** (1.0 / D); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    ell = 0.4 + rng.random(D); a = 0.7 + 0.6 * rng.random(); s = 0.1 + 0.3 * rng.random()
    ctx.set_option("small_ng1", 256)
    o, g = ctx.logml_grad(X, y, a, ell, s)            # ARD, one workgroup
    want, wg = ref_logml_grad(X, y, a, ell, s)
    worst["grad"] = max(worst["grad"], abs(o[0] - want) / abs(want), np.max(np.abs(g - wg)) / np.max(np.abs(wg)))
    al = 0.7 + 0.6 * rng.random(G); rho = 0.3 + rng.random(G); sg = 0.1 + 0.3 * rng.random(G)
    o, g, info = ctx.logml_grad_grid(X, y, al, rho, sg)   # chains
    assert np.all(info == 0)
    k = int(rng.integers(0, G))
    want, wg = ref_logml_grad(X, y, al[k], np.full(D, rho[k]), sg[k])
    wiso = np.array([wg[0], wg[1:1 + D].sum(), wg[-1]])
    worst["grad"] = max(worst["grad"], abs(o[k, 0] - want) / abs(want), np.max(np.abs(g[k] - wiso)) / np.max(np.abs(wiso)))
def dcov(kind, x, yv, a, l):   # derivative_kernels.R:39-53 times a^2: QQ, RQ(tj, tk) = QR(tk, tj), RR
    r = x[:, None] - yv[None, :]
    e = np.exp(-r * r / (2 * l * l))
    if kind == "QQ":
        return a * a * e
    if kind == "RQ":
        return a * a * (e * (-r)) / (l * l)
    return a * a * (e / (l * l) - (e * r * r) / l ** 4)

for it in range(12):
    n = int(rng.integers(2, 300)); m = int(rng.integers(1, 300)); B = int(rng.integers(1, 12))
    t = np.sort(rng.uniform(0, n / 10.0, n)); ts = np.sort(rng.uniform(0, n / 10.0, m))
    P = np.column_stack([0.8 + 0.3 * rng.random(B), 1.0 + 0.4 * rng.random(B), 0.05 + 0.1 * rng.random(B)])
    Y = np.sin(t)[:, None] + 0.1 * rng.standard_normal((n, B)); Z = rng.standard_normal((m, B))
    ctx.set_option("small_sdb", 0)
    d, mu, info = ctx.sample_derivs_batch(t, ts, Y, P, 1e-6, Z)
    assert np.all(info == 0), (n, m, B, info)
    b = int(rng.integers(0, B)); l, a, sy = P[b]
    K = dcov("QQ", t, t, a, l) + sy * sy * np.eye(n)
    Ks = dcov("RQ", ts, t, a, l); Kss = dcov("RR", ts, ts, a, l)
    mu_ref = Ks @ np.linalg.solve(K, Y[:, b]); cov = Kss - Ks @ np.linalg.solve(K, Ks.T) + 1e-6 * np.eye(m)
    d_ref = mu_ref + np.linalg.cholesky(0.5 * (cov + cov.T)) @ Z[:, b]
    worst["sd_mu"] = max(worst["sd_mu"], np.max(np.abs(mu[:, b] - mu_ref)) / np.max(np.abs(mu_ref)))
    worst["sd_draw"] = max(worst["sd_draw"], np.max(np.abs(d[:, b] - d_ref)) / np.max(np.abs(d_ref)))
print("worst relative errors:", {k: "%.1e" % v for k, v in worst.items()})
ok = worst["grid"] <= 1e-9 and worst["grad"] <= 1e-7 and worst["sd_mu"] <= 1e-8 and worst["sd_draw"] <= 1e-6
sys.exit(0 if ok else 1)
