"""Matrix-API squared-exponential kernels: mirror of R/kernels.R.

`phi` follows the reference: phi[[1]] = amplitude alpha, phi[[2]] = length-scale
(scalar; for QQard scalar or length-D vector).  A list, tuple, numpy vector or a
dict with those two entries in order is accepted, like R's `[[ ]]` on a list or
numeric vector (R/tests.R:36 passes hyperpar[1:2], a list).
"""
import numpy as np

from ._lib import COMPAT_RR, FULL, default_context

# R/kernels.R:31 applies phi1^2 to the first term of RR only (operator precedence).
# False = the mathematically intended kernel (== a^2 * derivative_kernels.R RR, as used
# at pendulum_fit.R:240); True reproduces the reference file as written.
COMPAT_R_RR = False


def _phi(phi):
    if isinstance(phi, dict):
        vals = list(phi.values())
    else:
        vals = list(phi)
    if len(vals) < 2:
        raise ValueError("phi must hold (alpha, length-scale)")
    return float(np.asarray(vals[0]).ravel()[0]), np.atleast_1d(np.asarray(vals[1], dtype=np.float64))


def QQ(x, y, phi, ctx=None):
    """phi1^2*exp(-((x - y)^2/(2 * phi2^2))) over all pairs -- R/kernels.R:22-24."""
    a, l = _phi(phi)
    return (ctx or default_context()).deriv_cov("QQ", x, y, a, l[0])


def QR(x, y, phi, ctx=None):
    """Cov(f(x_i), f'(y_j)) -- R/kernels.R:26-28."""
    a, l = _phi(phi)
    return (ctx or default_context()).deriv_cov("QR", x, y, a, l[0])


def RR(x, y, phi, compat=None, ctx=None):
    """Cov(f'(x_i), f'(y_j)) -- R/kernels.R:30-32 (see COMPAT_R_RR)."""
    a, l = _phi(phi)
    compat = COMPAT_R_RR if compat is None else compat
    return (ctx or default_context()).deriv_cov("RR", x, y, a, l[0], COMPAT_RR if compat else FULL)


def QQard(X, Y, phi, ctx=None):
    """phi1^2*exp(-(1/2)*sum(((x-y)/phi2)^2)) for all row pairs of X (n x D), Y (m x D)
    -- create_kernel_function / obs_list_outer, R/kernels.R:2-19."""
    a, l = _phi(phi)
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(1, -1) if Y.ndim == 2 and Y.shape[1] == X.size and X.size > 1 else X.reshape(-1, 1)
    if Y.ndim == 1:
        Y = Y.reshape(1, -1) if Y.size == X.shape[1] and X.shape[1] > 1 else Y.reshape(-1, 1)
    return (ctx or default_context()).se_cov(X, Y, a, l)
