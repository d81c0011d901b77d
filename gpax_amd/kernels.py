"""GP kernels with gpax's interface (``gpax/kernels.py``); the gram matrices are built by the HIP
kernel ``gram_kernel`` through ``gpx_gram`` -- no NumPy arithmetic here.

As in the reference, learnable parameters start with the prefix ``"k_"`` and priors/parameters refer
to data AFTER the GP's transforms.
"""

from __future__ import annotations

import numpy as np

from gpax_amd import _lib, state
from gpax_amd.utils.prior_utils import HalfNormal, LogNormal, Parameter, Prior


def _add_jitter(x, jitter=1e-6):
    """gpax/kernels.py:33-36."""
    return x + jitter


def _squared_distance(x1, x2, lengthscale):
    """gpax/kernels.py:39-51: scaled squared distances.  Built from the HIP gram kernel (the RBF gram with
    unit scale is exp(-r2/2)), so it is exact only where exp/log round-trips: use for inspection, the
    factorisation path never materialises distances."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    params = _lib.make_params(_lib.RBF, 1.0, lengthscale, None, None, 0.0, False)
    k = state.get_core().gram(params, x1, x2)
    return -2.0 * np.log(k)


class Kernel(Prior):
    """gpax/kernels.py:54-94: base kernel with ``k_noise ~ HalfNormal(0.01)`` and the constant
    ``k_jitter = 1e-6``."""

    attribute_prefix = "k_"
    KIND = None

    def __init__(self, k_noise=None, k_jitter=None):
        self.k_noise = k_noise if k_noise is not None else Parameter(HalfNormal(0.01), 1)
        self.k_jitter = k_jitter if k_jitter is not None else Parameter(1.0e-6, 1)
        for p in (self.k_noise, self.k_jitter):
            if not isinstance(p, Parameter):
                raise TypeError("kernel parameters must be Parameter instances")  # attrs instance_of in the reference

    def sample_prior(self, x1, x2, rng=None):
        """gpax/kernels.py:83-87: the kernel at one draw of its priors."""
        rng = rng if rng is not None else state.get_rng_key()[1]
        return self.__call__(x1, x2, **self.sample_parameters(rng))

    def _gram(self, X1, X2, k_scale, k_length, k_period, k_noise, k_jitter, apply_noise):
        X1 = np.asarray(X1, dtype=np.float64)
        X2 = np.asarray(X2, dtype=np.float64)
        # the diagonal rule is keyed on shapes (gpax/kernels.py:124); the noise is irrelevant otherwise
        noise = k_noise if (apply_noise and X1.shape == X2.shape) else None
        params = _lib.make_params(self.KIND, k_scale, k_length, k_period, noise, k_jitter, apply_noise)
        return state.get_core().gram(params, X1, X2)

    def _kernel_function(self, x1, x2, **params):
        raise NotImplementedError

    def __call__(self, x1, x2, **params):
        return self._kernel_function(x1, x2, **params)


class ScaleKernel(Kernel):
    """gpax/kernels.py:97-107: adds ``k_scale``, ``k_length`` ~ LogNormal(0, 1)."""

    def __init__(self, k_scale=None, k_length=None, **kw):
        super().__init__(**kw)
        self.k_scale = k_scale if k_scale is not None else Parameter(LogNormal(0.0, 1.0), 1)
        self.k_length = k_length if k_length is not None else Parameter(LogNormal(0.0, 1.0), 1)


class RBFKernel(ScaleKernel):
    KIND = _lib.RBF

    def _kernel_function(self, X1, X2, k_scale, k_length, k_noise=0.0, k_jitter=1e-6, apply_noise=True):
        """gpax/kernels.py:112-126."""
        return self._gram(X1, X2, k_scale, k_length, None, k_noise, k_jitter, apply_noise)


class MaternKernel(ScaleKernel):
    KIND = _lib.MATERN52

    def _kernel_function(self, X, Z, k_scale, k_length, k_noise=0.0, k_jitter=1e-6, apply_noise=True):
        """gpax/kernels.py:131-147 (nu = 5/2)."""
        return self._gram(X, Z, k_scale, k_length, None, k_noise, k_jitter, apply_noise)


class PeriodicKernel(ScaleKernel):
    """gpax/kernels.py:150-168.  The reference class defines no ``k_period`` Parameter (so it cannot
    be sampled as shipped); here one may be passed to the constructor, and ``_kernel_function``
    requires the keyword exactly as the reference does."""

    KIND = _lib.PERIODIC

    def __init__(self, k_period=None, **kw):
        super().__init__(**kw)
        if k_period is not None:
            self.k_period = k_period

    def _kernel_function(self, X, Z, k_scale, k_length, k_period, k_noise=0.0, k_jitter=1e-6, apply_noise=True):
        return self._gram(X, Z, k_scale, k_length, k_period, k_noise, k_jitter, apply_noise)
