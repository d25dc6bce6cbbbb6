"""GP posterior of a state and of its time derivative: mirror of R/ode_gp.R:1-32 and
R/ode_gp_library.R:4-33.

Both name pairs of the reference are returned: `mn`/`Kn` (R/ode_gp.R:13,31; `mn` is an
N x 1 matrix as `%*%` yields in R) and `condMean`/`condVar` (condMVNorm::condMVN through
R/ode_gp_library.R:17,32; R/tests.R uses both).  One Cholesky of K + sigma^2 I on the GPU
replaces the reference's two LU solves (mathematically identical).
"""
import numpy as np

from . import kernels as _k
from ._lib import COMPAT_RR, FULL, default_context


def _ret(mn, Kn):
    return {"mn": mn.reshape(-1, 1), "Kn": Kn, "condMean": mn.copy(), "condVar": Kn}


def p_Xn(tn, Xn, phi_n, sigma_n, joint=False, ctx=None):
    """mn = K (K + s^2 I)^-1 Xn, Kn = K - K (K + s^2 I)^-1 K, K = QQ(tn, tn, phi_n).
    joint=True adds the 1e-6 I_{2N} of the library variant (R/ode_gp_library.R:14-15)."""
    a, l = _k._phi(phi_n)
    jit = 1e-6 if joint else 0.0
    mn, Kn = (ctx or default_context()).gp_condition(tn, tn, Xn, a, l[0], float(sigma_n) ** 2 + jit, jit,
                                                    "QQ", "QQ", "QQ")
    return _ret(mn, Kn)


def p_dotXn(tn, Xn, phi_n, sigma_n, joint=False, compat=None, ctx=None):
    """mn = RQ (QQ + s^2 I)^-1 Xn, Kn = RR - RQ (QQ + s^2 I)^-1 QR  (R/ode_gp.R:19-32);
    joint=True is the [[UU + s^2 I, UD], [t(UD), DD]] + 1e-6 I form conditioned with
    condMVN (R/ode_gp_library.R:23-33)."""
    a, l = _k._phi(phi_n)
    jit = 1e-6 if joint else 0.0
    compat = _k.COMPAT_R_RR if compat is None else compat
    mn, Kn = (ctx or default_context()).gp_condition(tn, tn, Xn, a, l[0], float(sigma_n) ** 2 + jit, jit,
                                                    "QQ", "RQ", "RR", COMPAT_RR if compat else FULL)
    return _ret(mn, Kn)
