"""ctypes binding of ``libgpx_hip.so`` (the C ABI declared in ``include/gpx.h``).

There is NO CPU fallback: if the shared library is missing, or no gfx950 GPU is
visible, every compute entry point raises.  (The NumPy oracle under ``oracle/`` is
test infrastructure and is never imported from here.)
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

RBF, MATERN52, PERIODIC = 0, 1, 2

STAGE_GRAM, STAGE_GRAM_CROSS, STAGE_POTRF_DIAG, STAGE_TRSM_PANEL, STAGE_TRAILING, STAGE_EPILOGUE, STAGE_FACTOR_TOTAL, STAGE_LOOKAHEAD = range(8)
STAGE_NAMES = ["gram", "gram_cross", "potrf_diag", "trsm_panel", "trailing_update", "epilogue", "factor_total", "lookahead_update"]

ERR_NAMES = {
    -1: "GPX_ERR_ARG",
    -2: "GPX_ERR_HIP",
    -3: "GPX_ERR_STATE",
    -4: "GPX_ERR_NOMEM",
    -5: "GPX_ERR_COMM",
    -6: "GPX_ERR_UNSUPPORTED",
    -7: "GPX_ERR_RETRY",
    -8: "GPX_ERR_INTERNAL",
}
ERR_RETRY = -7

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgpx_hip.so")

_dp = C.POINTER(C.c_double)


class KernelParams(C.Structure):
    """``gpx_kernel_params`` of include/gpx.h."""

    _fields_ = [
        ("kind", C.c_int32),
        ("n_length", C.c_int32),
        ("n_period", C.c_int32),
        ("n_noise", C.c_int32),
        ("apply_noise", C.c_int32),
        ("reserved", C.c_int32),
        ("k_scale", C.c_double),
        ("k_jitter", C.c_double),
        ("k_length", _dp),
        ("k_period", _dp),
        ("k_noise", _dp),
    ]


class GpxError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol include/gpx.h declares
SIGNATURES = {
    "gpx_abi_version": (C.c_int, []),
    "gpx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gpx_destroy": (C.c_int, [C.c_void_p]),
    "gpx_last_error": (C.c_char_p, [C.c_void_p]),
    "gpx_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gpx_synchronize": (C.c_int, [C.c_void_p]),
    "gpx_gram": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), _dp, C.c_int64, _dp, C.c_int64, C.c_int32, _dp]),
    "gpx_set_train": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32]),
    "gpx_set_test": (C.c_int, [C.c_void_p, _dp, C.c_int64]),
    "gpx_fit_factor": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), _dp, _dp]),
    "gpx_log_marginal": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "gpx_lml_grad": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), _dp, _dp, _dp, _dp]),
    "gpx_lml_grad_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(KernelParams), _dp, _dp, _dp, _dp, C.POINTER(C.c_int32)]),
    "gpx_posterior": (C.c_int, [C.c_void_p, _dp, C.c_double, _dp, _dp, _dp]),
    "gpx_prior_cov": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), C.c_double, _dp]),
    "gpx_mvn_sample": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, _dp, C.c_int64, _dp]),
    "gpx_fit_predict": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), _dp, _dp, C.c_double, _dp, _dp, _dp]),
    "gpx_set_targets": (C.c_int, [C.c_void_p, _dp, _dp]),
    "gpx_fit_predict_async": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), C.c_double]),
    "gpx_fetch_results": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "gpx_fit_predict_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(KernelParams), _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int32)]),
    "gpx_sample_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(KernelParams), _dp, _dp, _dp, _dp, C.c_int64, C.c_int32,
                                   _dp, _dp, _dp, C.POINTER(C.c_int32)]),
    "gpx_runtime_info": (C.c_int, [C.c_char_p, C.c_int]),
    "gpx_acquisition": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.POINTER(C.c_int64), _dp]),
    "gpx_profile_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "gpx_profile_reset": (C.c_int, [C.c_void_p]),
    "gpx_profile_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gpx_profile_get": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64), _dp, _dp]),
    # include/gpx_dist.h
    "gpx_dist_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int64, C.c_int32, _dp, C.c_int64]),
    "gpx_dist_set_targets": (C.c_int, [C.c_void_p, _dp]),
    "gpx_dist_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "gpx_dist_unique_id": (C.c_int, [C.c_char_p]),
    "gpx_dist_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "gpx_dist_comm_destroy": (C.c_int, [C.c_void_p]),
    "gpx_dist_fit_predict": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), _dp, C.c_double, _dp, _dp, _dp]),
    "gpx_dist_build": (C.c_int, [C.c_void_p, C.POINTER(KernelParams), C.c_double]),
    "gpx_dist_step_diag": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_step_panel": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_step_update": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_step_gather": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_step_update_cols": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "gpx_dist_step_gather_group": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "gpx_dist_step_update_group": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "gpx_dist_group_size": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "gpx_dist_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "gpx_dist_step_diag_early": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_step_update_next_col": (C.c_int, [C.c_void_p, C.c_int64]),
    "gpx_dist_extract_z": (C.c_int, [C.c_void_p]),
    "gpx_dist_local_results": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.POINTER(C.c_int64), _dp, _dp, C.POINTER(C.c_int)]),
    "gpx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "gpx_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
}


def load():
    """Load the shared library (once).  Raises GpxError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpxError(
            f"{LIB_PATH} not found: the HIP library is not built.  Build it with "
            "`make -C gpax_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "gpax_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    compat = bool(os.environ.get("GPX_LIB_COMPAT"))   # tools only: A/B an older build of the library (tools/bits_r1.py, tools/ab_lib.py)
    for name, (res, args) in SIGNATURES.items():
        if compat and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gpx_abi_version() != 2 and not compat:
        raise GpxError("libgpx_hip.so ABI version mismatch")
    _lib = lib
    return lib


def runtime_info():
    """``(ok, text)``: which HIP runtime the library resolved to in this process (``gpx_runtime_info``).  ``ok`` is False
    when it is not the major.minor the library was compiled against -- the case when PyTorch, with its bundled
    ``libamdhip64.so`` of another ROCm release, was imported first."""
    buf = C.create_string_buffer(1024)
    rc = load().gpx_runtime_info(buf, 1024)
    return rc == 0, buf.value.decode()


def make_params_batch(param_list):
    """An array of ``gpx_kernel_params`` for the batched entry points from a list of ``make_params`` results."""
    arr = (KernelParams * len(param_list))()
    keep = []
    for i, (p, k) in enumerate(param_list):
        arr[i] = p
        keep.append(k)
    return arr, keep


def _arr(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def make_params(kind, k_scale, k_length, k_period=None, k_noise=None, k_jitter=1e-6, apply_noise=True):
    """Pack gpax's keyword set (gpax/kernels.py:112-121,152-162) into the C struct.
    Returns ``(struct, keepalive)``; hyper-parameters arrive as python floats or
    arrays of shape ``(1,)`` / ``(d,)`` / ``(n,)`` exactly as gpax passes them."""
    keep = []
    p = KernelParams()
    p.kind = int(kind)
    p.k_scale = float(np.asarray(k_scale, dtype=np.float64).reshape(-1)[0])
    p.k_jitter = float(np.asarray(k_jitter, dtype=np.float64).reshape(-1)[0])
    p.apply_noise = 1 if apply_noise else 0
    ln = _arr(k_length).reshape(-1)
    keep.append(ln)
    p.k_length, p.n_length = _ptr(ln), ln.size
    if k_period is not None:
        pr = _arr(k_period).reshape(-1)
        keep.append(pr)
        p.k_period, p.n_period = _ptr(pr), pr.size
    else:
        p.k_period, p.n_period = None, 0
    if k_noise is None:
        p.k_noise, p.n_noise = None, 0
    else:
        nz = _arr(k_noise).reshape(-1)
        keep.append(nz)
        p.k_noise, p.n_noise = _ptr(nz), nz.size
    return p, keep


class Core:
    """One handle = one GPU.  Thin, allocation-free-on-the-hot-path wrapper."""

    def __init__(self, device_id=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.gpx_create(int(device_id), C.byref(h))
        if rc != 0:
            msg = self._lib.gpx_last_error(None).decode()
            raise GpxError(f"gpx_create({device_id}) failed ({ERR_NAMES.get(rc, rc)}): {msg}")
        self._h = h
        self.n = self.m = self.d = 0
        self._gp_owner = self._gp_test = self._gp_factor = None  # bookkeeping of gpax_amd.gp
        self._async = None   # (params, new_noise) of the step in flight: re-issued when a hand-over timed out

    # -- plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        """rc < 0 raises; rc > 0 is the LAPACK-style info (returned)."""
        if rc < 0:
            msg = self._lib.gpx_last_error(self._h).decode()
            raise GpxError(f"{what} failed ({ERR_NAMES.get(rc, rc)}): {msg}")
        return rc

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self._lib.gpx_device_name(self._h, buf, 256), "gpx_device_name")
        return buf.value.decode()

    def synchronize(self):
        self._check(self._lib.gpx_synchronize(self._h), "gpx_synchronize")

    def set_option(self, name, value):
        self._check(self._lib.gpx_set_option(self._h, name.encode(), int(value)), "gpx_set_option")

    def get_option(self, name):
        v = C.c_int64()
        self._check(self._lib.gpx_get_option(self._h, name.encode(), C.byref(v)), "gpx_get_option")
        return v.value

    # -- the path
    def gram(self, params, x1, x2):
        p, _keep = params
        x1 = _arr(x1)
        x2 = _arr(x2)
        if x1.ndim != 2 or x2.ndim != 2 or x1.shape[1] != x2.shape[1]:
            raise ValueError("x1, x2 must be (n1,d), (n2,d)")
        out = np.empty((x1.shape[0], x2.shape[0]))
        self._check(
            self._lib.gpx_gram(self._h, C.byref(p), _ptr(x1), x1.shape[0], _ptr(x2), x2.shape[0], x1.shape[1], _ptr(out)),
            "gpx_gram",
        )
        return out

    def set_train(self, x):
        x = _arr(x)
        if x.ndim != 2:
            raise ValueError("x must be (n,d)")
        self._check(self._lib.gpx_set_train(self._h, _ptr(x), x.shape[0], x.shape[1]), "gpx_set_train")
        self.n, self.d = x.shape
        self.m = 0 if getattr(self, "_d_prev", x.shape[1]) != x.shape[1] else self.m
        self._d_prev = x.shape[1]

    def set_test(self, x_new):
        x_new = _arr(x_new)
        if x_new.ndim != 2 or (x_new.shape[0] and x_new.shape[1] != self.d):
            raise ValueError("x_new must be (m,d) with the training d")
        self._check(self._lib.gpx_set_test(self._h, _ptr(x_new), x_new.shape[0]), "gpx_set_test")
        self.m = x_new.shape[0]

    def fit_factor(self, params, want_k=False, want_l=False):
        p, _keep = params
        k = np.empty((self.n, self.n)) if want_k else None
        l = np.empty((self.n, self.n)) if want_l else None
        info = self._check(self._lib.gpx_fit_factor(self._h, C.byref(p), _ptr(k), _ptr(l)), "gpx_fit_factor")
        return info, k, l

    def log_marginal(self, y_minus_m, want_alpha=False):
        r = _arr(y_minus_m).reshape(-1)
        if r.size != self.n:
            raise ValueError("y_minus_m must have n entries")
        logp = C.c_double()
        alpha = np.empty(self.n) if want_alpha else None
        info = self._check(self._lib.gpx_log_marginal(self._h, _ptr(r), C.byref(logp), _ptr(alpha)), "gpx_log_marginal")
        return info, logp.value, alpha

    def lml_grad(self, params, y_minus_m, want_alpha=True):
        """Log-marginal likelihood and its gradient with respect to the kernel hyper-parameters
        (gpx_lml_grad).  Returns ``(info, logp, grads, alpha)``; ``grads`` is keyed like the kernel
        keywords (``k_scale`` (1,), ``k_length`` (n_length,), ``k_period`` (n_period,), ``k_noise`` (1,))."""
        p, _keep = params
        r = _arr(y_minus_m).reshape(-1)
        if r.size != self.n:
            raise ValueError("y_minus_m must have n entries")
        n_per = p.n_period if p.kind == PERIODIC else 0
        g = np.empty(2 + p.n_length + n_per)
        logp = C.c_double()
        alpha = np.empty(self.n) if want_alpha else None
        info = self._check(
            self._lib.gpx_lml_grad(self._h, C.byref(p), _ptr(r), C.byref(logp), _ptr(g), _ptr(alpha)), "gpx_lml_grad"
        )
        grads = {"k_scale": g[0:1], "k_length": g[1 : 1 + p.n_length]}
        if n_per:
            grads["k_period"] = g[1 + p.n_length : 1 + p.n_length + n_per]
        grads["k_noise"] = g[-1:]
        return info, logp.value, grads, alpha

    def lml_grad_batch(self, param_list, y_minus_m, want_alpha=True):
        """``gpx_lml_grad_batch``: B log-marginal likelihoods and gradients in one schedule.  ``param_list``: B
        ``make_params`` results of one form; ``y_minus_m`` (B, n).  Returns ``(info (B,), logp (B,), grads, alpha (B, n))``
        with ``grads`` a list of B dicts keyed like ``lml_grad``'s."""
        B = len(param_list)
        arr, _keep = make_params_batch(param_list)
        r = _arr(y_minus_m).reshape(B, -1)
        if r.shape[1] != self.n:
            raise ValueError("y_minus_m must be (B, n)")
        p0 = param_list[0][0]
        n_per = p0.n_period if p0.kind == PERIODIC else 0
        ng = 2 + p0.n_length + n_per
        g = np.empty((B, ng))
        logp = np.empty(B)
        alpha = np.empty((B, self.n)) if want_alpha else None
        info = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gpx_lml_grad_batch(self._h, B, arr, _ptr(r), _ptr(logp), _ptr(g), _ptr(alpha),
                                                 info.ctypes.data_as(C.POINTER(C.c_int32))), "gpx_lml_grad_batch")
        grads = []
        for b in range(B):
            gb = {"k_scale": g[b, 0:1], "k_length": g[b, 1 : 1 + p0.n_length]}
            if n_per:
                gb["k_period"] = g[b, 1 + p0.n_length : 1 + p0.n_length + n_per]
            gb["k_noise"] = g[b, -1:]
            grads.append(gb)
        return info, logp, grads, alpha

    def posterior(self, m_new=None, new_noise=0.0, want_mean=True, want_var=True, want_cov=False):
        mn = _arr(m_new).reshape(-1) if m_new is not None else None
        if mn is not None and mn.size != self.m:
            raise ValueError("m_new must have m entries")
        mean = np.empty(self.m) if want_mean else None
        var = np.empty(self.m) if want_var else None
        cov = np.empty((self.m, self.m)) if want_cov else None
        info = self._check(
            self._lib.gpx_posterior(self._h, _ptr(mn), float(new_noise), _ptr(mean), _ptr(var), _ptr(cov)), "gpx_posterior"
        )
        return info, mean, var, cov

    def prior_cov(self, params, new_noise=0.0):
        p, _keep = params
        cov = np.empty((self.m, self.m))
        self._check(self._lib.gpx_prior_cov(self._h, C.byref(p), float(new_noise), _ptr(cov)), "gpx_prior_cov")
        return cov

    def mvn_sample(self, mean, cov, eps):
        """``mean + eps @ chol(cov).T`` on the device; eps (nsamp, m) standard-normal draws."""
        cov = _arr(cov)
        eps = _arr(eps)
        m = cov.shape[0]
        if cov.shape != (m, m) or eps.ndim != 2 or eps.shape[1] != m:
            raise ValueError("cov must be (m,m), eps (nsamp,m)")
        mean = _arr(mean).reshape(-1) if mean is not None else None
        out = np.empty_like(eps)
        info = self._check(self._lib.gpx_mvn_sample(self._h, _ptr(mean), _ptr(cov), m, _ptr(eps), eps.shape[0], _ptr(out)), "gpx_mvn_sample")
        return info, out

    def fit_predict(self, params, y_minus_m, m_new=None, new_noise=0.0):
        p, _keep = params
        r = _arr(y_minus_m).reshape(-1)
        if r.size != self.n:
            raise ValueError("y_minus_m must have n entries")
        mn = _arr(m_new).reshape(-1) if m_new is not None else None
        logp = C.c_double()
        mean = np.empty(self.m)
        var = np.empty(self.m)
        info = self._check(
            self._lib.gpx_fit_predict(self._h, C.byref(p), _ptr(r), _ptr(mn), float(new_noise), C.byref(logp), _ptr(mean), _ptr(var)),
            "gpx_fit_predict",
        )
        return info, logp.value, mean, var

    # -- resident/asynchronous form used by bench.py
    def set_targets(self, y_minus_m, m_new=None):
        r = _arr(y_minus_m).reshape(-1)
        mn = _arr(m_new).reshape(-1) if m_new is not None else None
        self._check(self._lib.gpx_set_targets(self._h, _ptr(r), _ptr(mn)), "gpx_set_targets")

    def fit_predict_async(self, params, new_noise=0.0):
        p, _keep = params
        self._async = (params, float(new_noise))
        self._check(self._lib.gpx_fit_predict_async(self._h, C.byref(p), float(new_noise)), "gpx_fit_predict_async")

    def fetch_results(self):
        logp = C.c_double()
        mean = np.empty(self.m)
        var = np.empty(self.m)
        rc = self._lib.gpx_fetch_results(self._h, C.byref(logp), _ptr(mean), _ptr(var))
        if rc == ERR_RETRY and self._async is not None:
            # an in-kernel hand-over timed out (dispatches serialised or CUs contended): the handle now hands over at
            # kernel boundaries; the step is issued again, once, in this process
            params, new_noise = self._async
            self._check(self._lib.gpx_fit_predict_async(self._h, C.byref(params[0]), new_noise), "gpx_fit_predict_async (retry)")
            rc = self._lib.gpx_fetch_results(self._h, C.byref(logp), _ptr(mean), _ptr(var))
        info = self._check(rc, "gpx_fetch_results")
        return info, logp.value, mean, var

    def acquisition(self, kind, param, mu_scale=1.0, mu_shift=0.0, sd_scale=1.0, want_values=True):
        """``gpx_acquisition``: EI (kind 0, param = y_max) or UCB (kind 1, param = beta) over the posterior mean / variance
        the last ``posterior`` / ``fit_predict`` left on the device.  Returns ``(values or None, argmax, best)``."""
        vals = np.empty(self.m) if want_values else None
        idx, best = C.c_int64(), C.c_double()
        info = self._check(self._lib.gpx_acquisition(self._h, int(kind), float(param), float(mu_scale), float(mu_shift), float(sd_scale),
                                                     _ptr(vals), C.byref(idx), C.byref(best)), "gpx_acquisition")
        if info > 0:   # the factor behind the resident (mean, var) was not positive definite: NaNs, as the reference's jax path gives
            return (np.full(self.m, np.nan) if want_values else None), 0, float("nan")
        return vals, idx.value, best.value

    # -- batched hyper-parameter draws (gpax's vmap over hp samples, gpax/gp.py:354-364,384-390)
    def fit_predict_batch(self, param_list, y_minus_m, m_new=None, new_noise=None):
        """``gpx_fit_predict_batch``: B draws in one schedule.  ``param_list``: B ``make_params`` results;
        ``y_minus_m`` (B, n); ``m_new`` (B, m) or None; ``new_noise`` (B,) or None.
        Returns ``(info (B,), logp (B,), mean (B, m), var (B, m))``."""
        B = len(param_list)
        arr, _keep = make_params_batch(param_list)
        r = _arr(y_minus_m).reshape(B, -1)
        if r.shape[1] != self.n:
            raise ValueError("y_minus_m must be (B, n)")
        mn = _arr(m_new).reshape(B, self.m) if m_new is not None else None
        nn = _arr(new_noise).reshape(B) if new_noise is not None else None
        logp, mean, var = np.empty(B), np.empty((B, self.m)), np.empty((B, self.m))
        info = np.zeros(B, dtype=np.int32)
        self._check(
            self._lib.gpx_fit_predict_batch(self._h, B, arr, _ptr(r), _ptr(mn), _ptr(nn), _ptr(logp), _ptr(mean), _ptr(var),
                                            info.ctypes.data_as(C.POINTER(C.c_int32))),
            "gpx_fit_predict_batch",
        )
        return info, logp, mean, var

    def sample_batch(self, param_list, y_minus_m, m_new, new_noise, eps, conditioned=True):
        """``gpx_sample_batch``: per draw the posterior (or prior) mean, sd = sqrt(diag cov) and ``mean + eps chol(cov)^T``
        with the covariance kept in HBM.  ``eps`` (B, nsamp, m).  Returns ``(info (B,), mean (B, m), sd (B, m),
        samples (B, nsamp, m))``."""
        B = len(param_list)
        arr, _keep = make_params_batch(param_list)
        eps = _arr(eps)
        if eps.ndim != 3 or eps.shape[0] != B or eps.shape[2] != self.m:
            raise ValueError("eps must be (B, nsamp, m)")
        r = _arr(y_minus_m).reshape(B, -1) if y_minus_m is not None else None
        if conditioned and (r is None or r.shape[1] != self.n):
            raise ValueError("y_minus_m must be (B, n)")
        mn = _arr(m_new).reshape(B, self.m) if m_new is not None else None
        nn = _arr(new_noise).reshape(B) if new_noise is not None else None
        mean, sd, out = np.empty((B, self.m)), np.empty((B, self.m)), np.empty_like(eps)
        info = np.zeros(B, dtype=np.int32)
        self._check(
            self._lib.gpx_sample_batch(self._h, B, arr, _ptr(r), _ptr(mn), _ptr(nn), _ptr(eps), eps.shape[1], 1 if conditioned else 0,
                                       _ptr(mean), _ptr(sd), _ptr(out), info.ctypes.data_as(C.POINTER(C.c_int32))),
            "gpx_sample_batch",
        )
        return info, mean, sd, out

    # -- measurement
    def profile_mode(self, mode):
        self._check(self._lib.gpx_profile_mode(self._h, int(mode)), "gpx_profile_mode")

    def profile_dump(self, path):
        self._check(self._lib.gpx_profile_dump(self._h, str(path).encode()), "gpx_profile_dump")

    def profile_reset(self):
        self._check(self._lib.gpx_profile_reset(self._h), "gpx_profile_reset")

    def profile_get(self, stage):
        ms = C.c_double()
        n = C.c_int64()
        fl = C.c_double()
        by = C.c_double()
        self._check(self._lib.gpx_profile_get(self._h, int(stage), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "gpx_profile_get")
        return dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
