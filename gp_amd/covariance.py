"""Mirror of the Rcpp exports in covariance.cpp (rbf_cov_chol :9-47, approx_L :49-96) and of the Stan
external function approx_Lz (models/cubic_interpolated_gp.hpp:38-73)."""
import numpy as np

from ._lib import default_context


def rbf_cov_chol(x1, l_, ctx=None):
    """list(L=, dLdl=): Sigma_ij = exp(-(xi-xj)^2/(2 l^2)) + 1e-10 I, L = lower Cholesky,
    dLdl = exact dL/dl (the reference gets it by forward-mode AD, fvar<double>)."""
    L, dL = (ctx or default_context()).rbf_cov_chol(x1, float(l_))
    return {"L": L, "dLdl": dL}


def _neighbours(l, lp):
    """Interval rule of covariance.cpp:57-61 (first p with lp[p+1] >= l), clamped to the last interval."""
    lp = np.asarray(lp, dtype=np.float64)
    k = 0
    while k < lp.size - 1 and not lp[k + 1] >= l:
        k += 1
    return min(k, lp.size - 2)


def approx_L(l, lp, Ls, dLdls, ctx=None):
    """approx_L(l, lp, Ls, dLdls): cubic Hermite blend of the tabulated Cholesky factors at length-scale l
    (covariance.cpp:49-96).  Only the two neighbouring table entries travel to the device."""
    c = ctx or default_context()
    k = _neighbours(l, lp)
    c.interp_load([lp[k], lp[k + 1]], [Ls[k], Ls[k + 1]], [dLdls[k], dLdls[k + 1]])
    return c.approx_L(l)


def approx_Lz(l, lp, Ls, dLdls, z, ctx=None):
    """approx_Lz(l, lp, Ls, dLdls, z) = approx_L(l, ...) %*% z (models/cubic_interpolated_gp.hpp:38-73)."""
    c = ctx or default_context()
    k = _neighbours(l, lp)
    c.interp_load([lp[k], lp[k + 1]], [Ls[k], Ls[k + 1]], [dLdls[k], dLdls[k + 1]])
    return c.approx_Lz(l, z)


def approx_Lz_grad(l, lp, Ls, dLdls, z, ctx=None):
    """(f, dfdl) of approx_Lz: the value and the partial with respect to l that the `var` overload of
    build_output attaches under Stan's reverse mode (models/cubic_interpolated_gp.hpp:6-32, dvdl :67)."""
    c = ctx or default_context()
    k = _neighbours(l, lp)
    c.interp_load([lp[k], lp[k + 1]], [Ls[k], Ls[k + 1]], [dLdls[k], dLdls[k + 1]])
    return c.approx_Lz_grad(l, z)


class FactorInterpolator:
    """Device-resident table for repeated queries: what test_interpolate.R:9-19 builds with P calls of
    rbf_cov_chol and cubic_interpolated_gp.stan consumes once per leapfrog step."""

    def __init__(self, x, lp, ctx=None):
        self.ctx = ctx or default_context()
        self.lp = np.asarray(lp, dtype=np.float64)
        self.ctx.interp_build(x, self.lp)

    def L(self, l):
        return self.ctx.approx_L(l)

    def Lz(self, l, z):
        return self.ctx.approx_Lz(l, z)

    def Lz_grad(self, l, z):
        return self.ctx.approx_Lz_grad(l, z)
