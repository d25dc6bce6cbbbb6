"""Elementwise SE derivative kernels: mirror of derivative_kernels.R:39-73.

QQ..TT(tj, tk, l): vectorised over tj, tk (numpy broadcasting stands in for R's
recycling), unit amplitude; Q = value, R = first, T = second derivative of the
process, first letter for tj.  Used as a^2 * outer(ti, ti, FUN) at
pendulum_fit.R:237-240; `outer(kind, x, y, l, a)` is that matrix-level fast path.

NB: these names collide with the matrix API of gp_amd.kernels exactly as the two R
files collide in the reference: import one module or the other.
"""
import numpy as np

from ._lib import default_context


def _mk(kind):
    def f(tj, tk, l, ctx=None):
        return (ctx or default_context()).deriv_elem(kind, tj, tk, l)
    f.__name__ = kind
    f.__doc__ = "%s(tj, tk, l) -- derivative_kernels.R" % kind
    return f


QQ = _mk("QQ")
QR = _mk("QR")
RQ = _mk("RQ")
RR = _mk("RR")
QT = _mk("QT")
TQ = _mk("TQ")
RT = _mk("RT")
TR = _mk("TR")
TT = _mk("TT")


def outer(kind, x, y, l, a=1.0, ctx=None):
    """a^2 * outer(x, y, FUN = function(tj, tk) kind(tj, tk, l)) in one kernel launch."""
    return (ctx or default_context()).deriv_cov(kind, np.asarray(x, float), np.asarray(y, float), a, l)
