"""ctypes binding of the CPU oracle (oracle/gp_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from gp_amd/ (the product).  See the header
of gp_oracle.c for what the oracle restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgporacle.so")

KINDS = ("QQ", "QR", "RQ", "RR", "QT", "TQ", "RT", "TR", "TT")


def build(force=False):
    src = os.path.join(_HERE, "gp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgporacle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_deriv_elem.restype = C.c_double
        _lib.orc_deriv_elem.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.orc_stan_lp.restype = C.c_double
        _lib.orc_stan_lp.argtypes = [C.c_double] * 5
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    """Column-major float64 copy."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _d(x):
    return C.c_double(float(x))


def QQard(X, Y, alpha, ell):
    X = _f(np.atleast_2d(X)); Y = _f(np.atleast_2d(Y))
    n, D = X.shape; m = Y.shape[0]
    ell = np.ascontiguousarray(np.atleast_1d(np.asarray(ell, dtype=np.float64)))
    K = np.empty((n, m), order="F")
    lib().orc_QQard(_p(X), n, n, _p(Y), m, m, D, _d(alpha), _p(ell), int(ell.size), _p(K), n)
    return K


def _k1d(fn, x, y, alpha, l, *extra):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    K = np.empty((x.size, y.size), order="F")
    getattr(lib(), fn)(_p(x), int(x.size), _p(y), int(y.size), _d(alpha), _d(l), *extra, _p(K), int(x.size))
    return K


def QQ(x, y, alpha, l):
    return _k1d("orc_QQ", x, y, alpha, l)


def QR(x, y, alpha, l):
    return _k1d("orc_QR", x, y, alpha, l)


def RR(x, y, alpha, l, compat=False):
    return _k1d("orc_RR", x, y, alpha, l, C.c_int(int(compat)))


def deriv_elem(kind, tj, tk, l):
    k = KINDS.index(kind) if isinstance(kind, str) else int(kind)
    tj, tk = np.broadcast_arrays(np.asarray(tj, dtype=np.float64), np.asarray(tk, dtype=np.float64))
    shape = tj.shape
    tj = np.ascontiguousarray(tj).ravel(); tk = np.ascontiguousarray(tk).ravel()
    out = np.empty(tj.size)
    lib().orc_deriv_vec(k, _p(tj), _p(tk), C.c_long(tj.size), _d(l), _p(out))
    return out.reshape(shape)


def deriv_cov(kind, x, y, alpha, l):
    k = KINDS.index(kind) if isinstance(kind, str) else int(kind)
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    K = np.empty((x.size, y.size), order="F")
    lib().orc_deriv_cov(k, _p(x), int(x.size), _p(y), int(y.size), _d(alpha), _d(l), _p(K), int(x.size))
    return K


def joint_cov(t, alpha, l, sigma, jitter=1e-6, compat=False):
    t = np.ascontiguousarray(t, dtype=np.float64); n = t.size
    K = np.empty((2 * n, 2 * n), order="F")
    lib().orc_joint_cov(_p(t), n, _d(alpha), _d(l), _d(sigma), _d(jitter), int(compat), _p(K), 2 * n)
    return K


def cov_exp_quad(X, alpha, rho):
    X = _f(np.asarray(X, dtype=np.float64).reshape(len(X), -1))
    n, D = X.shape
    K = np.empty((n, n), order="F")
    lib().orc_cov_exp_quad(_p(X), n, n, D, _d(alpha), _d(rho), _p(K), n)
    return K


def cholesky(A, blocked=False):
    """Lower factor (copy); raises ValueError(info) when not positive definite."""
    L = _f(A).copy(order="F"); n = L.shape[0]
    info = lib().orc_cholesky_blocked(_p(L), n, n, 64) if blocked else lib().orc_cholesky(_p(L), n, n)
    if info:
        raise ValueError(info)
    return L


def trsv_lower(L, b):
    L = _f(L); z = np.array(b, dtype=np.float64).copy()
    lib().orc_trsv_lower(_p(L), L.shape[0], L.shape[0], _p(z))
    return z


def trmv_lower(L, z):
    L = _f(L); z = np.ascontiguousarray(z, dtype=np.float64); f = np.empty_like(z)
    lib().orc_trmv_lower(_p(L), L.shape[0], L.shape[0], _p(z), _p(f))
    return f


def lu_solve(A, B):
    A = _f(A).copy(order="F"); B = _f(np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)).copy(order="F")
    info = lib().orc_lu_solve(_p(A), A.shape[0], A.shape[0], _p(B), B.shape[1], B.shape[0])
    if info:
        raise ValueError(info)
    return B


def logml(X, y, alpha, rho, sigma, jitter=0.0):
    """(logml, sum log L_ii, z'z, info) -- models/fit_hyperparameters.stan:18-32."""
    X = _f(np.asarray(X, dtype=np.float64).reshape(len(y), -1)); n, D = X.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(3)
    info = lib().orc_logml(_p(X), n, n, D, _p(y), _d(alpha), _d(rho), _d(sigma), _d(jitter), _p(out))
    return out[0], out[1], out[2], info


def logml_grad(X, y, alpha, rho, sigma, jitter=0.0):
    """((logml, sum log L_ii, z'z), (d/dalpha, d/drho, d/dsigma), info)."""
    X = _f(np.asarray(X, dtype=np.float64).reshape(len(y), -1)); n, D = X.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(3); g = np.empty(3)
    info = lib().orc_logml_grad(_p(X), n, n, D, _p(y), _d(alpha), _d(rho), _d(sigma), _d(jitter), _p(out), _p(g))
    return out, g, info


def stan_lp(sum_log_diag, quad, alpha, rho, sigma):
    return lib().orc_stan_lp(sum_log_diag, quad, alpha, rho, sigma)


def rbf_cov_chol(x, l):
    x = np.ascontiguousarray(x, dtype=np.float64); n = x.size
    L = np.empty((n, n), order="F"); dL = np.empty((n, n), order="F")
    info = lib().orc_rbf_cov_chol(_p(x), n, _d(l), _p(L), n, _p(dL), n)
    if info:
        raise ValueError(info)
    return L, dL


def approx_L(l, lp, Ls, dLs):
    lp = np.ascontiguousarray(lp, dtype=np.float64); P = lp.size
    n = Ls[0].shape[0]
    Ls = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in Ls]))
    dLs = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in dLs]))
    out = np.empty((n, n), order="F")
    lib().orc_approx_L(_d(l), _p(lp), P, _p(Ls), _p(dLs), n, _p(out), n)
    return out


def approx_Lz(l, lp, Ls, dLs, z):
    lp = np.ascontiguousarray(lp, dtype=np.float64); P = lp.size
    n = Ls[0].shape[0]
    Ls = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in Ls]))
    dLs = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in dLs]))
    z = np.ascontiguousarray(z, dtype=np.float64); f = np.empty(n)
    lib().orc_approx_Lz(_d(l), _p(lp), P, _p(Ls), _p(dLs), n, _p(z), _p(f))
    return f


def approx_Lz_grad(l, lp, Ls, dLs, z):
    """(f, dfdl): value v z and the reverse-mode partial dvdl z of models/cubic_interpolated_gp.hpp:6-32,62-72."""
    lp = np.ascontiguousarray(lp, dtype=np.float64); P = lp.size
    n = Ls[0].shape[0]
    Ls = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in Ls]))
    dLs = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in dLs]))
    z = np.ascontiguousarray(z, dtype=np.float64); f = np.empty(n); g = np.empty(n)
    lib().orc_approx_Lz_grad(_d(l), _p(lp), P, _p(Ls), _p(dLs), n, _p(z), _p(f), _p(g))
    return f, g


def gp_condition(K, Ks, Kss, y, s2, jitter=0.0):
    """(mn, Kn): mn = Ks solve(K + s2 I, y), Kn = Kss - Ks solve(K + s2 I, t(Ks)) + jitter I through the
    oracle's LU (base-R solve() == dgesv): the generic form of R/ode_gp.R:1-32, pendulum_fit.R:242-251."""
    K = _f(K); Ks = _f(Ks); Kss = _f(Kss); y = np.ascontiguousarray(y, dtype=np.float64)
    n = K.shape[0]; m = Ks.shape[0]
    mn = np.empty(m); Kn = np.empty((m, m), order="F")
    info = lib().orc_gp_condition(_p(K), n, _p(Ks), m, _p(Kss), _p(y), _d(s2), _d(jitter), _p(mn), _p(Kn))
    if info:
        raise ValueError(info)
    return mn, Kn


def p_Xn(tn, Xn, alpha, l, sigma):
    tn = np.ascontiguousarray(tn, dtype=np.float64); Xn = np.ascontiguousarray(Xn, dtype=np.float64)
    n = tn.size; mn = np.empty(n); Kn = np.empty((n, n), order="F")
    info = lib().orc_p_Xn(_p(tn), _p(Xn), n, _d(alpha), _d(l), _d(sigma), _p(mn), _p(Kn))
    if info:
        raise ValueError(info)
    return mn, Kn


def p_dotXn(tn, Xn, alpha, l, sigma, compat=False):
    tn = np.ascontiguousarray(tn, dtype=np.float64); Xn = np.ascontiguousarray(Xn, dtype=np.float64)
    n = tn.size; mn = np.empty(n); Kn = np.empty((n, n), order="F")
    info = lib().orc_p_dotXn(_p(tn), _p(Xn), n, _d(alpha), _d(l), _d(sigma), int(compat), _p(mn), _p(Kn))
    if info:
        raise ValueError(info)
    return mn, Kn


def p_dotXn_joint(tn, Xn, alpha, l, sigma, jitter=1e-6, compat=False):
    tn = np.ascontiguousarray(tn, dtype=np.float64); Xn = np.ascontiguousarray(Xn, dtype=np.float64)
    n = tn.size; mn = np.empty(n); Kn = np.empty((n, n), order="F")
    info = lib().orc_p_dotXn_joint(_p(tn), _p(Xn), n, _d(alpha), _d(l), _d(sigma), _d(jitter), int(compat), _p(mn), _p(Kn))
    if info:
        raise ValueError(info)
    return mn, Kn


def sample_derivs_moments(ti, ynoise, l, a, sy, jitter=1e-8):
    ti = np.ascontiguousarray(ti, dtype=np.float64); y = np.ascontiguousarray(ynoise, dtype=np.float64)
    n = ti.size; mu = np.empty(n); cov = np.empty((n, n), order="F")
    info = lib().orc_sample_derivs_moments(_p(ti), _p(y), n, _d(l), _d(a), _d(sy), _d(jitter), _p(mu), _p(cov))
    if info:
        raise ValueError(info)
    return mu, cov


def joint_logml(t, yy, alpha, l, sigma, jitter=1e-6):
    t = np.ascontiguousarray(t, dtype=np.float64); yy = np.ascontiguousarray(yy, dtype=np.float64)
    out = np.empty(3)
    info = lib().orc_joint_logml(_p(t), int(t.size), _p(yy), _d(alpha), _d(l), _d(sigma), _d(jitter), _p(out))
    return out[0], out[1], out[2], info


def synth(n, D, seed=20240601):
    """Deterministic synthetic inputs of SURVEY section 8(d): X (n x D, F-order), y."""
    X = np.empty((n, D), order="F"); y = np.empty(n)
    lib().orc_synth(n, D, C.c_ulonglong(seed), _p(X), _p(y))
    return X, y


class create_p_dotXnS:
    """create_p_dotXnS, R/ode_gp_library.R:43-93, restated statement by statement with numpy
    (test infrastructure, like everything in this module).  The reference factors
    K_XX + 1e-6 I with qr() and solves with it (:55-57, :76); every call rebuilds the joint mean
    and covariance of all star points (:71-77) and conditions the newest on the earlier draws
    with condMVNorm::condMVN (:80-81; third-party, unpinned: its published formula
    cMu = mu_d + C D^-1 (x_g - mu_g), cVar = B - C D^-1 C' is restated in _condMVN).
    rnorm's variate is supplied by the caller (z) -- R's RNG stream cannot be reproduced here;
    compat_sd=True passes condVar as the standard deviation, as :83 is written.
    Parity unpinned by the reference (no recorded outputs); pinned by
    tests/test_oracle.py::test_seq_sampler_chain_equals_joint."""

    def __init__(self, Xn_list, mn, Kn, alpha, ell, compat_sd=False):
        self.X = np.column_stack([np.asarray(x, dtype=np.float64).ravel() for x in Xn_list])  # :45
        self.N, self.D = self.X.shape
        self.alpha, self.ell = float(alpha), np.atleast_1d(np.asarray(ell, dtype=np.float64))
        self.i = 1
        K_XX = QQard(self.X, self.X, self.alpha, self.ell)                                   # :50
        self.K_XsX = np.zeros((0, self.N))
        self.K_XsXs = np.zeros((0, 0))
        self.Q, self.R = np.linalg.qr(K_XX + 1e-6 * np.eye(self.N))                          # :55
        self.K_XX_1_mn = self._solve(np.asarray(mn, dtype=np.float64).reshape(-1, 1))       # :56
        self.K_XX_1_Kn = self._solve(np.asarray(Kn, dtype=np.float64))                      # :57
        self.Xs = np.zeros((0, self.D))
        self.dot_Xs = np.zeros(0)
        self.compat_sd = compat_sd

    def _solve(self, B):
        import scipy.linalg
        return scipy.linalg.solve_triangular(self.R, self.Q.T @ B, lower=False)

    @staticmethod
    def _condMVN(mean, sigma, dep, given, x_given):
        if len(given) == 0:
            return mean[dep], sigma[np.ix_(dep, dep)]
        B = sigma[np.ix_(dep, dep)]; Cm = sigma[np.ix_(dep, given)]; Dm = sigma[np.ix_(given, given)]
        CDinv = Cm @ np.linalg.inv(Dm)   # condMVN: C %*% solve(D)
        return mean[dep] + CDinv @ (x_given - mean[given]), B - CDinv @ Cm.T

    def joint(self):
        """(m, K) of all star points so far, :74-77."""
        m = (self.K_XsX @ self.K_XX_1_mn).ravel()
        S = self._solve(self.K_XsX.T)
        K = self.K_XsXs - self.K_XsX @ S + self.K_XsX @ self.K_XX_1_Kn @ S
        K = (K + K.T) / 2 + 1e-6 * np.eye(K.shape[0])
        return m, K

    def __call__(self, xs_vec, z):
        xs = np.asarray(xs_vec, dtype=np.float64).reshape(1, -1)                             # :68
        self.K_XsX = np.vstack([self.K_XsX, QQard(xs, self.X, self.alpha, self.ell)])       # :71
        kq = QQard(self.Xs, xs, self.alpha, self.ell) if self.Xs.shape[0] else np.zeros((0, 1))
        self.K_XsXs = np.block([[self.K_XsXs, kq], [kq.T, QQard(xs, xs, self.alpha, self.ell)]])  # :72-73
        m, K = self.joint()
        i = self.i
        cm, cv = self._condMVN(m, K, [i - 1], list(range(i - 1)), self.dot_Xs)              # :80-81
        mu, var = float(np.ravel(cm)[0]), float(np.ravel(cv)[0])
        dot_xs = mu + (var if self.compat_sd else np.sqrt(var)) * float(z)                   # :83
        self.i += 1                                                                          # :86
        self.Xs = np.vstack([self.Xs, xs])
        self.dot_Xs = np.append(self.dot_Xs, dot_xs)
        return {"mu": mu, "sigma": var, "dot_xs": dot_xs}                                    # :92
