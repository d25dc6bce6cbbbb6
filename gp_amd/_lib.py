"""ctypes binding of libgpmi.so (include/gpmi.h) -- the only way this package computes.

There is no CPU fallback: if the shared library is missing or no gfx950 device is
visible, construction of a Context raises GpmiError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgpmi.so")
PROBES_LIB_PATH = os.path.join(_HERE, "csrc", "libgpmi_probes.so")

KINDS = ("QQ", "QR", "RQ", "RR", "QT", "TQ", "RT", "TR", "TT")
FULL, LOWER, COMPAT_RR = 0, 1, 2

# every symbol include/gpmi.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "gpmi_version", "gpmi_last_error", "gpmi_device_count", "gpmi_create", "gpmi_destroy",
    "gpmi_set_stream", "gpmi_reset_stream", "gpmi_sync", "gpmi_reserve", "gpmi_set_option",
    "gpmi_se_cov", "gpmi_se_cov_dev", "gpmi_deriv_cov", "gpmi_deriv_cov_dev", "gpmi_deriv_elem",
    "gpmi_joint_cov", "gpmi_potrf", "gpmi_potrf_dev", "gpmi_trmv_lower", "gpmi_trsv_lower", "gpmi_exact_gp_f",
    "gpmi_logml", "gpmi_logml_dev", "gpmi_logml_grid", "gpmi_logml_grid_dev", "gpmi_logml_grid_ard", "gpmi_logml_grid_ard_dev",
    "gpmi_joint_logml", "gpmi_joint_logml_dev", "gpmi_joint_logml_grid_dev", "gpmi_rbf_cov_chol", "gpmi_gp_condition", "gpmi_sample_derivs", "gpmi_sample_derivs_batch",
    "gpmi_interp_build", "gpmi_interp_load", "gpmi_approx_L", "gpmi_approx_Lz", "gpmi_approx_Lz_dev", "gpmi_approx_Lz_grad", "gpmi_approx_Lz_grad_dev",
    "gpmi_interp_free", "gpmi_logml_grad", "gpmi_logml_grad_grid",
    "gpmi_seq_create", "gpmi_seq_step", "gpmi_seq_commit", "gpmi_seq_count", "gpmi_seq_destroy",
    "gpmi_last_timing", "gpmi_kernel_timing", "gpmi_kernel_timing_ex",
)
# additionally exported by the probe build (libgpmi_probes.so, -DGPMI_PROBES; tools/ only)
PROBE_SYMBOLS = ("gpmi_probe_syrk", "gpmi_probe_mfma", "gpmi_probe_mfma_peak", "gpmi_probe_clock", "gpmi_probe_fused", "gpmi_probe_small", "gpmi_probe_body")


class GpmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgpmi error %d: %s" % (code, msg))
        self.code = code


class NotPositiveDefinite(GpmiError):
    """info = k > 0: the leading minor of order k is not positive definite
    (base-R chol() error / Stan cholesky_decompose domain_error)."""

    def __init__(self, k):
        RuntimeError.__init__(self, "the leading minor of order %d is not positive definite" % k)
        self.code = k
        self.order = k


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as ROCm's).  If libgpmi pulls in
    ROCm's copy first and torch is imported later, the process ends up with two HIP/HSA runtimes and
    torch finds no GPU.  When torch is installed but not yet loaded, map ITS runtime first so that
    both libraries share one; without torch (e.g. under R) libgpmi uses ROCm's as linked."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libgpmi.so; loud failure when it has not been built.  GPMI_USE_PROBES=1 (tools/ only)
    loads the probe build instead: same library plus the A/B kernels and gpmi_probe_* entry points."""
    global _lib
    if _lib is not None:
        return _lib
    probes = os.environ.get("GPMI_USE_PROBES", "") == "1"
    path = PROBES_LIB_PATH if probes else LIB_PATH
    if not os.path.exists(path):
        raise GpmiError(-4, "%s not found: build it with `python -m gp_amd._build%s` "
                        "(hipcc --offload-arch=gfx950); gp_amd has no CPU fallback" % (path, " --probes" if probes else ""))
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    lib.gpmi_last_error.restype = C.c_char_p
    for name in SYMBOLS + (PROBE_SYMBOLS if probes else ()):
        fn = getattr(lib, name)
        if name != "gpmi_last_error":
            fn.restype = C.c_int
    _lib = lib
    return lib


def _chk(rc, allow_info=False):
    if rc == 0:
        return 0
    if rc > 0:
        if allow_info:
            return rc
        raise NotPositiveDefinite(rc)
    raise GpmiError(rc, load().gpmi_last_error().decode())


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _d(x):
    return C.c_double(float(x))


def _vec(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())


def _mat(X):
    """2-D column-major float64 view/copy (R's native layout)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    return np.asfortranarray(X)


def _kind(k):
    return KINDS.index(k) if isinstance(k, str) else int(k)


def device_count():
    n = C.c_int(0)
    rc = load().gpmi_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context:
    """One gpmi_ctx: bound to one GPU and to this process."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._lib = load()
        _chk(self._lib.gpmi_create(C.byref(self._h), int(device)))
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gpmi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing -----------------------------------------------------------
    def set_stream(self, stream_handle):
        """Run on the caller's hipStream_t (handle 0 = the legacy null stream, torch's default
        stream); None goes back to the context's own stream."""
        if stream_handle is None:
            _chk(self._lib.gpmi_reset_stream(self._h))
        else:
            _chk(self._lib.gpmi_set_stream(self._h, C.c_void_p(int(stream_handle))))

    def sync(self):
        _chk(self._lib.gpmi_sync(self._h))

    def reserve(self, n_max):
        _chk(self._lib.gpmi_reserve(self._h, int(n_max)))

    def set_option(self, name, value):
        _chk(self._lib.gpmi_set_option(self._h, name.encode(), int(value)))

    def last_timing(self):
        out = np.zeros(3)
        _chk(self._lib.gpmi_last_timing(self._h, _p(out)))
        return out

    def kernel_timing(self, reset=True):
        """{category: (launches, total_ms, total_work)}: 'build' (bytes), 'syrk' and 'panel' (flops), 'syrk_multi_round'
        (the trailing-update launches of more than one round of tiles, a subset of 'syrk')."""
        out = np.zeros(12)
        _chk(self._lib.gpmi_kernel_timing_ex(self._h, int(bool(reset)), _p(out)))
        return {"build": tuple(out[0:3]), "syrk": tuple(out[3:6]), "panel": tuple(out[6:9]), "syrk_multi_round": tuple(out[9:12])}

    # ---- covariance builders (host buffers) ----------------------------------
    def se_cov(self, X, Y, alpha, ell, diag_add=0.0, flags=FULL):
        X = _mat(X)
        n, D = X.shape
        ell = _vec(ell)
        if Y is None:
            K = np.empty((n, n), order="F")
            _chk(self._lib.gpmi_se_cov(self._h, _p(X), n, max(n, 1), None, n, max(n, 1), D, _d(alpha), _p(ell),
                                       int(ell.size), _d(diag_add), int(flags), _p(K), max(n, 1)))
            return K
        Y = _mat(Y)
        if Y.shape[1] != D:
            raise GpmiError(-1, "X and Y must have the same number of columns")
        m = Y.shape[0]
        K = np.empty((n, m), order="F")
        _chk(self._lib.gpmi_se_cov(self._h, _p(X), n, max(n, 1), _p(Y), m, max(m, 1), D, _d(alpha), _p(ell),
                                   int(ell.size), _d(diag_add), int(flags), _p(K), max(n, 1)))
        return K

    def deriv_cov(self, kind, x, y, alpha, l, flags=FULL):
        x = _vec(x); y = _vec(y)
        K = np.empty((x.size, y.size), order="F")
        _chk(self._lib.gpmi_deriv_cov(self._h, _kind(kind), _p(x), int(x.size), _p(y), int(y.size), _d(alpha),
                                      _d(l), int(flags), _p(K), max(int(x.size), 1)))
        return K

    def deriv_elem(self, kind, tj, tk, l):
        tj, tk = np.broadcast_arrays(np.asarray(tj, dtype=np.float64), np.asarray(tk, dtype=np.float64))
        shape = tj.shape
        a = _vec(tj); b = _vec(tk)
        out = np.empty(a.size)
        _chk(self._lib.gpmi_deriv_elem(self._h, _kind(kind), _p(a), _p(b), C.c_size_t(a.size), _d(l), _p(out)))
        return out.reshape(shape)

    def joint_cov(self, t, alpha, l, sigma, jitter=1e-6, flags=FULL):
        t = _vec(t); n = t.size
        K = np.empty((2 * n, 2 * n), order="F")
        _chk(self._lib.gpmi_joint_cov(self._h, _p(t), n, _d(alpha), _d(l), _d(sigma), _d(jitter), int(flags),
                                      _p(K), max(2 * n, 1)))
        return K

    # ---- factorisation --------------------------------------------------------
    def potrf(self, A):
        """Lower Cholesky factor of A (copy); raises NotPositiveDefinite."""
        L = np.array(A, dtype=np.float64, order="F", copy=True)
        n = L.shape[0]
        if L.ndim != 2 or L.shape[1] != n:
            raise GpmiError(-1, "matrix must be square")
        _chk(self._lib.gpmi_potrf(self._h, _p(L), n, max(n, 1)))
        return L

    def trmv_lower(self, L, z):
        L = _mat(L); z = _vec(z); f = np.empty_like(z)
        _chk(self._lib.gpmi_trmv_lower(self._h, _p(L), L.shape[0], max(L.shape[0], 1), _p(z), _p(f)))
        return f

    def trsv_lower(self, L, b):
        L = _mat(L); b = _vec(b); z = np.empty_like(b)
        _chk(self._lib.gpmi_trsv_lower(self._h, _p(L), L.shape[0], max(L.shape[0], 1), _p(b), _p(z)))
        return z

    def exact_gp_f(self, X, alpha, ell, z, jitter=1e-10):
        """f = chol(cov_exp_quad(X, alpha, ell) + jitter I) z (models/exact_gp.stan:17-25), fused on the device; raises
        NotPositiveDefinite."""
        X = _mat(X); z = _vec(z); ell = _vec(ell)
        n, D = X.shape
        if z.size != n:
            raise GpmiError(-1, "X and z disagree on N")
        f = np.empty(n)
        _chk(self._lib.gpmi_exact_gp_f(self._h, _p(X), n, n, D, _d(alpha), _p(ell), int(ell.size), _d(jitter), _p(z), _p(f)))
        return f

    # ---- marginal likelihood -------------------------------------------------
    def logml(self, X, y, alpha, ell, sigma, jitter=0.0):
        """(logml, sum log L_ii, z'z); raises NotPositiveDefinite."""
        X = _mat(X); y = _vec(y); ell = _vec(ell)
        n, D = X.shape
        if y.size != n:
            raise GpmiError(-1, "X and y disagree on N")
        out = np.empty(3)
        _chk(self._lib.gpmi_logml(self._h, _p(X), n, n, D, _p(y), _d(alpha), _p(ell), int(ell.size), _d(sigma),
                                  _d(jitter), _p(out)))
        return out[0], out[1], out[2]

    def logml_grid(self, X, y, alpha, rho, sigma, jitter=0.0):
        """Arrays (G,3) of (logml, sum log L_ii, z'z) and info (G,) for G hyper-parameter points."""
        X = _mat(X); y = _vec(y)
        n, D = X.shape
        alpha, rho, sigma = np.broadcast_arrays(np.asarray(alpha, float), np.asarray(rho, float), np.asarray(sigma, float))
        a = _vec(alpha); r = _vec(rho); s = _vec(sigma)
        G = a.size
        out = np.empty((G, 3)); info = np.zeros(G, dtype=np.int32)
        _chk(self._lib.gpmi_logml_grid(self._h, _p(X), n, n, D, _p(y), _p(a), _p(r), _p(s), G, _d(jitter), _p(out),
                                       _p(info)))
        return out, info

    def logml_grid_ard(self, X, y, alpha, ell, sigma, jitter=0.0):
        """ARD grid: ell is (G, D), one length-scale per dimension and point; returns (G, 3), info (G,)."""
        X = _mat(X); y = _vec(y)
        n, D = X.shape
        E = np.ascontiguousarray(np.asarray(ell, dtype=np.float64).reshape(-1, D))
        G = E.shape[0]
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, float), (G,)))
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, float), (G,)))
        out = np.empty((G, 3)); info = np.zeros(G, dtype=np.int32)
        _chk(self._lib.gpmi_logml_grid_ard(self._h, _p(X), n, n, D, _p(y), _p(a), _p(E), _p(s), G, _d(jitter), _p(out),
                                           _p(info)))
        return out, info

    def joint_logml(self, t, yy, alpha, l, sigma, jitter=1e-6):
        t = _vec(t); yy = _vec(yy)
        if yy.size != 2 * t.size:
            raise GpmiError(-1, "yy must stack [y; y'] (length 2N)")
        out = np.empty(3)
        _chk(self._lib.gpmi_joint_logml(self._h, _p(t), int(t.size), _p(yy), _d(alpha), _d(l), _d(sigma),
                                        _d(jitter), _p(out)))
        return out[0], out[1], out[2]

    def rbf_cov_chol(self, x, l):
        x = _vec(x); n = x.size
        L = np.empty((n, n), order="F"); dL = np.empty((n, n), order="F")
        _chk(self._lib.gpmi_rbf_cov_chol(self._h, _p(x), n, _d(l), _p(L), max(n, 1), _p(dL), max(n, 1)))
        return L, dL

    def logml_grad(self, X, y, alpha, ell, sigma, jitter=0.0):
        """((logml, sum log L_ii, z'z), grad) with grad = (d/dalpha, d/dell..., d/dsigma)."""
        X = _mat(X); y = _vec(y); ell = _vec(ell)
        n, D = X.shape
        if y.size != n:
            raise GpmiError(-1, "X and y disagree on N")
        out = np.empty(3); g = np.empty(2 + ell.size)
        _chk(self._lib.gpmi_logml_grad(self._h, _p(X), n, max(n, 1), D, _p(y), _d(alpha), _p(ell), ell.size, _d(sigma),
                                       _d(jitter), _p(out), _p(g)))
        return out, g

    def logml_grad_grid(self, X, y, alpha, rho, sigma, jitter=0.0):
        """(out (G, 3), grad (G, 3), info (G,)): value and (d/dalpha, d/drho, d/dsigma) at G points, on the lanes."""
        X = _mat(X); y = _vec(y)
        n, D = X.shape
        alpha, rho, sigma = np.broadcast_arrays(np.asarray(alpha, float), np.asarray(rho, float), np.asarray(sigma, float))
        a = _vec(alpha); r = _vec(rho); s = _vec(sigma)
        G = a.size
        out = np.empty((G, 3)); g = np.empty((G, 3)); info = np.zeros(G, dtype=np.int32)
        _chk(self._lib.gpmi_logml_grad_grid(self._h, _p(X), n, max(n, 1), D, _p(y), _p(a), _p(r), _p(s), G, _d(jitter),
                                            _p(out), _p(g), _p(info)))
        return out, g, info

    # ---- Cholesky-factor interpolation over the length-scale ------------------
    def interp_build(self, x, lp):
        """Table of L(lp[p]), dL/dl(lp[p]) built and kept on the device (test_interpolate.R:9-19)."""
        x = _vec(x); lp = _vec(lp)
        _chk(self._lib.gpmi_interp_build(self._h, _p(x), x.size, _p(lp), lp.size))
        self._itp_n = x.size

    def interp_load(self, lp, Ls, dLdls):
        """Upload a caller-supplied table: sequences of P n x n matrices."""
        lp = _vec(lp)
        n = np.asarray(Ls[0]).shape[0]
        A = np.ascontiguousarray(np.stack([np.asfortranarray(np.asarray(a, dtype=np.float64)).ravel(order="F") for a in Ls]))
        B = np.ascontiguousarray(np.stack([np.asfortranarray(np.asarray(a, dtype=np.float64)).ravel(order="F") for a in dLdls]))
        if A.shape != (lp.size, n * n) or B.shape != A.shape:
            raise GpmiError(-1, "lp, Ls and dLdls disagree on the table shape")
        _chk(self._lib.gpmi_interp_load(self._h, _p(lp), lp.size, _p(A), _p(B), n, n))
        self._itp_n = n

    def approx_L(self, l):
        n = getattr(self, "_itp_n", 0)
        out = np.empty((n, n), order="F")
        _chk(self._lib.gpmi_approx_L(self._h, _d(l), _p(out), max(n, 1)))
        return out

    def approx_Lz(self, l, z):
        z = _vec(z)
        if z.size != getattr(self, "_itp_n", -1):
            raise GpmiError(-1, "z must have the table's order")
        f = np.empty(z.size)
        _chk(self._lib.gpmi_approx_Lz(self._h, _d(l), _p(z), _p(f)))
        return f

    def approx_Lz_grad(self, l, z):
        """(f, dfdl): approx_L(l) z and its partial in l (the `var` overload of build_output,
        models/cubic_interpolated_gp.hpp:6-32)."""
        z = _vec(z)
        if z.size != getattr(self, "_itp_n", -1):
            raise GpmiError(-1, "z must have the table's order")
        f = np.empty(z.size); g = np.empty(z.size)
        _chk(self._lib.gpmi_approx_Lz_grad(self._h, _d(l), _p(z), _p(f), _p(g)))
        return f, g

    def approx_Lz_grad_dev(self, l, dz_ptr, df_ptr, dg_ptr):
        _chk(self._lib.gpmi_approx_Lz_grad_dev(self._h, _d(l), C.c_void_p(dz_ptr), C.c_void_p(df_ptr), C.c_void_p(dg_ptr)))

    def approx_Lz_dev(self, l, dz_ptr, df_ptr):
        _chk(self._lib.gpmi_approx_Lz_dev(self._h, _d(l), C.c_void_p(dz_ptr), C.c_void_p(df_ptr)))

    def interp_free(self):
        _chk(self._lib.gpmi_interp_free(self._h))
        self._itp_n = 0

    def gp_condition(self, t, ts, y, alpha, l, s2, jitter, kindK, kindS, kindSS, flags=FULL):
        t = _vec(t); ts = _vec(ts); y = _vec(y)
        n, m = t.size, ts.size
        mn = np.empty(m); Kn = np.empty((m, m), order="F")
        _chk(self._lib.gpmi_gp_condition(self._h, _p(t), n, _p(ts), m, _p(y), _d(alpha), _d(l), _d(s2), _d(jitter),
                                         _kind(kindK), _kind(kindS), _kind(kindSS), int(flags), _p(mn), _p(Kn),
                                         max(m, 1)))
        return mn, Kn

    def sample_derivs(self, t, ts, y, l, a, sy, jitter, z):
        """(draw, mu): mu + chol(cov) z of sample_derivs (pendulum_fit.R:227-255), fused on the device."""
        t = _vec(t); ts = _vec(ts); y = _vec(y); z = _vec(z)
        if y.size != t.size or z.size != ts.size:
            raise GpmiError(-1, "y must match t and z must match ts")
        draw = np.empty(ts.size); mu = np.empty(ts.size)
        _chk(self._lib.gpmi_sample_derivs(self._h, _p(t), int(t.size), _p(ts), int(ts.size), _p(y), _d(l), _d(a), _d(sy),
                                          _d(jitter), _p(z), _p(draw), _p(mu)))
        return draw, mu

    def sample_derivs_batch(self, t, ts, Y, params, jitter, Z):
        """(draws (m, B), mus (m, B), info (B,)): B independent draws on the lanes; Y (n, B), params (B, 3) rows
        (l, a, sy), Z (m, B)."""
        t = _vec(t); ts = _vec(ts)
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64).reshape(t.size, -1))
        B = Y.shape[1]
        Z = np.asfortranarray(np.asarray(Z, dtype=np.float64).reshape(ts.size, -1))
        P = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1, 3))
        if Z.shape[1] != B or P.shape[0] != B:
            raise GpmiError(-1, "Y, params and Z disagree on the batch size")
        draws = np.empty((ts.size, B), order="F"); mus = np.empty((ts.size, B), order="F"); info = np.zeros(B, dtype=np.int32)
        _chk(self._lib.gpmi_sample_derivs_batch(self._h, _p(t), int(t.size), _p(ts), int(ts.size), _p(Y), max(int(t.size), 1),
                                                _p(P), B, _d(jitter), _p(Z), max(int(ts.size), 1), _p(draws), max(int(ts.size), 1),
                                                _p(mus), max(int(ts.size), 1), _p(info)))
        return draws, mus, info

    def seq_sampler(self, X, mn, Kn, alpha, ell, jitter=1e-6, max_steps=256):
        """Sequential conditional sampler (create_p_dotXnS, R/ode_gp_library.R:43-93)."""
        return SeqSampler(self, X, mn, Kn, alpha, ell, jitter, max_steps)

    # ---- device-pointer API (torch tensors own the memory; plumbing only) -----
    def logml_dev(self, dX_ptr, n, ldx, D, dy_ptr, alpha, ell, sigma, jitter, dout_ptr, dinfo_ptr):
        ell = _vec(ell)
        _chk(self._lib.gpmi_logml_dev(self._h, C.c_void_p(dX_ptr), int(n), int(ldx), int(D), C.c_void_p(dy_ptr),
                                      _d(alpha), _p(ell), int(ell.size), _d(sigma), _d(jitter),
                                      C.c_void_p(dout_ptr), C.c_void_p(dinfo_ptr)))

    def logml_grid_dev(self, dX_ptr, n, ldx, D, dy_ptr, alpha, rho, sigma, jitter, dout_ptr, dinfo_ptr):
        a = _vec(alpha); r = _vec(rho); s = _vec(sigma)
        _chk(self._lib.gpmi_logml_grid_dev(self._h, C.c_void_p(dX_ptr), int(n), int(ldx), int(D),
                                           C.c_void_p(dy_ptr), _p(a), _p(r), _p(s), int(a.size), _d(jitter),
                                           C.c_void_p(dout_ptr), C.c_void_p(dinfo_ptr)))

    def logml_grid_ard_dev(self, dX_ptr, n, ldx, D, dy_ptr, alpha, ell, sigma, jitter, dout_ptr, dinfo_ptr):
        E = np.ascontiguousarray(np.asarray(ell, dtype=np.float64).reshape(-1, int(D)))
        G = E.shape[0]
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, float), (G,)))
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, float), (G,)))
        _chk(self._lib.gpmi_logml_grid_ard_dev(self._h, C.c_void_p(dX_ptr), int(n), int(ldx), int(D), C.c_void_p(dy_ptr),
                                               _p(a), _p(E), _p(s), G, _d(jitter), C.c_void_p(dout_ptr), C.c_void_p(dinfo_ptr)))

    def joint_logml_dev(self, dt_ptr, n, dyy_ptr, alpha, l, sigma, jitter, dout_ptr, dinfo_ptr):
        _chk(self._lib.gpmi_joint_logml_dev(self._h, C.c_void_p(dt_ptr), int(n), C.c_void_p(dyy_ptr), _d(alpha),
                                            _d(l), _d(sigma), _d(jitter), C.c_void_p(dout_ptr), C.c_void_p(dinfo_ptr)))

    def joint_logml_grid_dev(self, dt_ptr, n, dyy_ptr, alpha, l, sigma, jitter, dout_ptr, dinfo_ptr):
        a = _vec(alpha); r = _vec(l); s = _vec(sigma)
        _chk(self._lib.gpmi_joint_logml_grid_dev(self._h, C.c_void_p(dt_ptr), int(n), C.c_void_p(dyy_ptr), _p(a), _p(r), _p(s),
                                                 int(a.size), _d(jitter), C.c_void_p(dout_ptr), C.c_void_p(dinfo_ptr)))

    def se_cov_dev(self, dX_ptr, n, ldx, dY_ptr, m, ldy, D, alpha, ell, diag_add, flags, dK_ptr, ldk):
        ell = _vec(ell)
        _chk(self._lib.gpmi_se_cov_dev(self._h, C.c_void_p(dX_ptr), int(n), int(ldx),
                                       C.c_void_p(dY_ptr) if dY_ptr else None, int(m), int(ldy), int(D), _d(alpha),
                                       _p(ell), int(ell.size), _d(diag_add), int(flags), C.c_void_p(dK_ptr), int(ldk)))

    def potrf_dev(self, dA_ptr, n, lda, dinfo_ptr):
        _chk(self._lib.gpmi_potrf_dev(self._h, C.c_void_p(dA_ptr), int(n), int(lda), C.c_void_p(dinfo_ptr)))

    # ---- probes (libgpmi_probes.so only: GPMI_USE_PROBES=1, tools/) -------------
    def _probe(self, name):
        fn = getattr(self._lib, name, None)
        if fn is None:
            raise GpmiError(-1, "%s exists in the probe build only (GPMI_USE_PROBES=1, python -m gp_amd._build --probes)" % name)
        return fn

    def probe_mfma(self, A, B):
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        out = np.empty((16, 16))
        _chk(self._probe("gpmi_probe_mfma")(self._h, _p(A), _p(B), _p(out)))
        return out

    def probe_syrk(self, m, k, reps=5):
        """(avg ms per launch, TFLOP/s at m(m+1)k algorithmic flops)."""
        ms = C.c_double(0.0)
        _chk(self._probe("gpmi_probe_syrk")(self._h, int(m), int(k), int(reps), C.byref(ms)))
        return ms.value, m * (m + 1.0) * k / (ms.value * 1e-3) / 1e12

    def probe_clock(self, reset=True):
        """(avg shader clock MHz, avg cycles per SYRK workgroup, workgroups) since the last reset."""
        out = np.zeros(3)
        _chk(self._probe("gpmi_probe_clock")(self._h, int(bool(reset)), _p(out)))
        if out[2] == 0:
            return 0.0, 0.0, 0
        return out[0] / max(out[1], 1.0) * 100.0, out[0] / out[2], int(out[2])

    def probe_fused(self):
        out = np.zeros(8)
        _chk(self._probe("gpmi_probe_fused")(self._h, _p(out)))
        return out

    def probe_body(self):
        """(loads, loop, stores, factor16, factor-wave wait, bodies): cycles of the diagonal-block bodies since the last call."""
        out = np.zeros(6)
        _chk(self._probe("gpmi_probe_body")(self._h, _p(out)))
        return out

    def probe_small(self):
        """(build, diagonal blocks, rows below, launches, trailing tiles, finalize): cycles since the last call."""
        out = np.zeros(6)
        _chk(self._probe("gpmi_probe_small")(self._h, _p(out)))
        return out

    def probe_mfma_peak(self, iters=20000):
        t = C.c_double(0.0); mhz = C.c_double(0.0)
        _chk(self._probe("gpmi_probe_mfma_peak")(self._h, int(iters), C.byref(t), C.byref(mhz)))
        return t.value, mhz.value


class SeqSampler:
    """One gpmi_seq: step(xs) -> (condMean, condVar) of the new point given the committed draws,
    commit(dot_xs) appends the draw."""

    def __init__(self, ctx, X, mn, Kn, alpha, ell, jitter=1e-6, max_steps=256):
        X = _mat(X); mn = _vec(mn); Kn = _mat(Kn); ell = _vec(ell)
        n, D = X.shape
        if mn.size != n or Kn.shape != (n, n):
            raise GpmiError(-1, "X, mn and Kn disagree on N")
        self._ctx = ctx  # keeps the context alive
        self._lib = ctx._lib
        self._h = C.c_void_p()
        self.D = D
        _chk(self._lib.gpmi_seq_create(ctx._h, C.byref(self._h), _p(X), n, max(n, 1), D, _p(mn), _p(Kn), max(n, 1),
                                       _d(alpha), _p(ell), int(ell.size), _d(jitter), int(max_steps)))

    def step(self, xs):
        xs = _vec(xs)
        if xs.size != self.D:
            raise GpmiError(-1, "xs must have D = %d entries" % self.D)
        out = np.empty(2)
        _chk(self._lib.gpmi_seq_step(self._h, _p(xs), _p(out)))
        return out[0], out[1]

    def commit(self, dot_xs):
        _chk(self._lib.gpmi_seq_commit(self._h, _d(dot_xs)))

    @property
    def count(self):
        return self._lib.gpmi_seq_count(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gpmi_seq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device=None):
    """Process-wide lazily created context (one per device).  In a forked child of a process that
    already used the GPU the library answers GPMI_EFORK (HIP cannot be re-initialised there): the
    parent's context is never handed out and none is created behind the caller's back."""
    if device is None:
        device = int(os.environ.get("GPMI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (os.getpid(), device)
    if key not in _default:
        _default[key] = Context(device)  # raises GpmiError(-5) in a forked child of a GPU process
    return _default[key]
