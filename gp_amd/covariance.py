"""Mirror of the Rcpp export in covariance.cpp:9-47."""
from ._lib import default_context


def rbf_cov_chol(x1, l_, ctx=None):
    """list(L=, dLdl=): Sigma_ij = exp(-(xi-xj)^2/(2 l^2)) + 1e-10 I, L = lower Cholesky,
    dLdl = exact dL/dl (the reference gets it by forward-mode AD, fvar<double>)."""
    L, dL = (ctx or default_context()).rbf_cov_chol(x1, float(l_))
    return {"L": L, "dLdl": dL}
