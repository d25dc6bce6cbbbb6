"""GP posterior of a state and of its time derivative: mirror of R/ode_gp.R:1-32 and
R/ode_gp_library.R:4-33, and the sequential derivative sampler create_p_dotXnS
(R/ode_gp_library.R:43-93).

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


# R/ode_gp_library.R:83 draws rnorm(1, condMean, condVar): the conditional VARIANCE is passed
# where rnorm expects a standard deviation.  False = the intended draw (sd = sqrt(condVar));
# True reproduces the file as written.  The returned `sigma` is condVar either way, as in the
# reference (:92).
COMPAT_R_RNORM_SD = False


def create_p_dotXnS(Xn_list, mn, Kn, theta, rng=None, compat_sd=None, max_steps=256, ctx=None):
    """Sampler of the derivative at new states, one at a time, each draw conditioned on the
    earlier ones: mirror of create_p_dotXnS, R/ode_gp_library.R:43-93 (used at R/tests.R:78-91).

    Xn_list: the state time series (one N-vector per dimension; X = cbind of them, :45),
    mn, Kn: derivative posterior at the data (p_dotXn), theta = (alpha, length-scale(s)) of QQard.
    Returns the closure p_dotXnS(xs_vec) -> dict(mu, sigma, dot_xs) (:92); the standard-normal
    variate of the draw comes from `rng` (numpy Generator) or the call's `z`.  The O(N^3) algebra
    (:55-57) runs once, on the GPU, here; a call reads one N x N matrix (gpmi_seq_*)."""
    X = np.column_stack([np.asarray(x, dtype=np.float64).ravel() for x in Xn_list])
    a, l = _k._phi(theta)
    smp = (ctx or default_context()).seq_sampler(X, np.asarray(mn, float).ravel(), Kn, a, l, 1e-6, max_steps)
    gen = rng or np.random.default_rng()
    compat = COMPAT_R_RNORM_SD if compat_sd is None else compat_sd

    def p_dotXnS(xs_vec, z=None):
        mu, var = smp.step(np.atleast_1d(np.asarray(xs_vec, dtype=np.float64)))
        sd = var if compat else np.sqrt(var)
        dot_xs = mu + sd * (gen.standard_normal() if z is None else float(z))
        smp.commit(dot_xs)
        return {"mu": mu, "sigma": var, "dot_xs": dot_xs}

    p_dotXnS.sampler = smp
    return p_dotXnS
