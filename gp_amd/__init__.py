"""gp_amd -- MI355X-native exact-GP marginal-likelihood hot path (libgpmi + host mirror).

Host-side mirror of the reference's R / Rcpp / Stan interfaces for this path:

  kernels              R/kernels.R            QQ, QR, RR, QQard (matrix API)
  derivative_kernels   derivative_kernels.R   QQ..TT(tj, tk, l) (elementwise API)
  ode_gp               R/ode_gp.R, R/ode_gp_library.R   p_Xn, p_dotXn, create_p_dotXnS
  covariance           covariance.cpp         rbf_cov_chol
  stan_models          models/fit_hyperparameters.stan, models/exact_gp.stan
  pendulum             pendulum_fit.R:227-255 sample_derivs
  grid                 hyper-parameter grid sharded over GPUs (torch.distributed / RCCL)

All arithmetic runs in libgpmi.so (HIP, gfx950) through the C ABI of include/gpmi.h;
there is no CPU fallback.
"""
from ._lib import Context, GpmiError, NotPositiveDefinite, default_context, device_count  # noqa: F401

__all__ = ["Context", "GpmiError", "NotPositiveDefinite", "default_context", "device_count"]
