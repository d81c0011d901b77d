"""Acquisition functions (``gpax/acquisition.py``): analytic EI / UCB on the HIP posterior and the
Halton-sampling optimiser.  For a conditioned model with ``fast=True`` (the reference's default) the analytic
values are a device epilogue over the posterior mean / variance that stay in HBM (``gpx_acquisition``), and the
Halton optimiser evaluates ALL its candidates as the border rows of one factorisation -- the reference pushes them
through ``predict`` in batches of ``batch_threshold`` = 100, re-factorising k_train each time
(gpax/acquisition.py:203-210 -> gpax/gp.py:293-304).  The Monte-Carlo path (q > 1) is host NumPy over
``model.sample`` as in the reference."""

from __future__ import annotations

import math

import numpy as np
from scipy.special import ndtr
from scipy.stats import qmc

from gpax_amd import state

ACQ_EI, ACQ_UCB = 0, 1   # include/gpx.h GPX_ACQ_*


def split_array(x, s):
    """gpax/utils/utils.py:37-50: chunks of at most ``s`` rows."""
    return [x[i : i + s] for i in range(0, len(x), s)]


def x_shape_err_msg(x, desired_dims):
    """gpax/acquisition.py:31-40."""
    return f"x of shape {tuple(np.shape(x))} must have dimensions {tuple(desired_dims)}"


class AcquisitionFunction:
    """gpax/acquisition.py:42-223 (analytic q=1 path and the Halton optimiser)."""

    def __init__(self, bounds, q=1, penalty_function=None, penalty_strength=0.0, batch_threshold=100,
                 force_monte_carlo=False, fast=True):
        self.bounds = np.asarray(bounds, dtype=np.float64)
        assert self.bounds.ndim == 2 and self.bounds.shape[0] == 2
        if not isinstance(q, int) or q < 1:
            raise ValueError("q must be an int >= 1")
        self.q, self.penalty_strength, self.batch_threshold = q, penalty_strength, batch_threshold
        self.penalty_function = penalty_function if penalty_function is not None else (lambda _: 0.0)
        self.force_monte_carlo, self.fast = force_monte_carlo, fast

    @property
    def name(self):
        return f"{self.__class__.__name__}(q={self.q})"

    def _analytic(self, model, x):
        raise NotImplementedError(
            f"Acquisition function {self.__class__.__name__} does not have an analytic acquisition function defined"
        )

    def _integrand(self, model, x):
        raise NotImplementedError

    def _get_integrand_samples(self, model, x):
        """gpax/acquisition.py:70-82: (hp_samples x gp_samples, N, q)."""
        n, d = x.shape[0], x.shape[2]
        result = model.sample(x.reshape(-1, d), fast=self.fast)
        return result.y.reshape(-1, n, self.q)

    def _monte_carlo_evaluation(self, model, x):
        """gpax/acquisition.py:104-119: the integrand averaged over the GP samples."""
        if x.ndim == 2 and self.q == 1:
            x = x.reshape(x.shape[0], 1, x.shape[-1])
        if x.ndim != 3 or x.shape[1] != self.q:
            raise ValueError(x_shape_err_msg(x, ("N", "q", "d")))
        return self._integrand(model, x).mean(axis=0)

    def _analytic_evaluation(self, model, x):
        """gpax/acquisition.py:121-131."""
        if self.q > 1:
            raise ValueError("Cannot evaluate analytic expression for q!=1")
        if x.ndim == 3 and x.shape[1] == 1:
            x = x.reshape(x.shape[0], x.shape[-1])
        if x.ndim != 2:
            raise ValueError(x_shape_err_msg(x, ("N", "d")))
        return self._analytic(model, x)

    def __call__(self, model, x):
        """gpax/acquisition.py:133-164."""
        x = np.asarray(x, dtype=np.float64)
        if self.force_monte_carlo or self.q > 1:
            return self._monte_carlo_evaluation(model, x)
        return self._analytic_evaluation(model, x)

    def _optimize_halton(self, model, n_samples):
        """gpax/acquisition.py:166-216."""
        key, _ = state.get_rng_key()
        halton = qmc.Halton(d=self.bounds.shape[1] * self.q, seed=key)
        samples = halton.random(n=n_samples)
        l_bounds = np.tile(np.atleast_1d(self.bounds[0, :].squeeze()), self.q)
        u_bounds = np.tile(np.atleast_1d(self.bounds[1, :].squeeze()), self.q)
        samples = qmc.scale(samples, l_bounds, u_bounds).reshape(n_samples, self.q, -1)
        if self._device_path(model):
            # all candidates in ONE posterior call (they are independent rows: the values do not depend on the batching,
            # which the reference needs only to bound jax's memory); values, arg-max and maximum come from the device
            _, argmax, best = self._device_values(model, samples.reshape(n_samples, -1), want_values=False)
            return samples[argmax, ...], float(best)
        vals = [np.atleast_1d(self.__call__(model, xx)) for xx in split_array(samples, self.batch_threshold)]
        vals = np.concatenate(vals).flatten()
        argmax = int(vals.argmax())
        return samples[argmax, ...], float(vals[argmax])

    # -- device epilogue (subclasses with an analytic form set DEVICE_KIND and _device_param)
    DEVICE_KIND = None

    def _device_param(self, model):
        raise NotImplementedError

    def _device_path(self, model):
        return (self.DEVICE_KIND is not None and self.q == 1 and not self.force_monte_carlo and self.fast
                and model.x is not None and model.y is not None and model._output_reverse_is_affine())

    def _device_values(self, model, x, want_values=True):
        return model._acquisition_fast(x, self.DEVICE_KIND, self._device_param(model), want_values)

    def optimize(self, model, n, method="halton"):
        if method.lower() == "halton":
            return self._optimize_halton(model, n)
        raise ValueError(f"Unknown optimizer {method}")


class UpperConfidenceBound(AcquisitionFunction):
    """gpax/acquisition.py:247-274."""

    def __init__(self, *args, beta=10.0, **kw):
        super().__init__(*args, **kw)
        if beta < 0:
            raise ValueError("beta must be >= 0")
        self.beta = beta

    DEVICE_KIND = ACQ_UCB

    def _device_param(self, model):
        return self.beta

    def _analytic(self, model, x):
        if self._device_path(model):
            return self._device_values(model, x)[0]
        mu, sd = model.predict(x, fast=self.fast)
        if math.isinf(self.beta):
            return sd
        if self.beta == 0.0:
            return mu
        return mu + math.sqrt(self.beta + 1e-12) * sd

    def _integrand(self, model, x):
        samples = self._get_integrand_samples(model, x)
        mean = samples.mean(axis=0, keepdims=True)
        if math.isinf(self.beta):
            return np.abs(samples - mean).max(axis=-1)
        beta_prime = math.sqrt(self.beta * math.pi / 2.0)
        return (mean + beta_prime * np.abs(samples - mean)).max(axis=-1)


class ExpectedImprovement(AcquisitionFunction):
    """gpax/acquisition.py:277-305."""

    def _get_y_max(self, model):
        if model.y is not None:
            return model.y.max()
        acqf = UpperConfidenceBound(self.bounds, beta=0.0, q=1, fast=True)
        x_star, _ = acqf.optimize(model, n=1000, method="Halton")
        y_max, _ = model.predict(x_star, fast=True)
        return y_max

    DEVICE_KIND = ACQ_EI

    def _device_param(self, model):
        return float(self._get_y_max(model))

    def _analytic(self, model, x):
        if self._device_path(model):
            return self._device_values(model, x)[0]
        mu, sd = model.predict(x, fast=self.fast)
        y_max = self._get_y_max(model)
        u = (mu - y_max) / sd
        ucdf = ndtr(u)
        updf = np.exp(-0.5 * u * u - 0.5 * math.log(2 * math.pi))
        return sd * updf + (mu - y_max) * ucdf

    def _integrand(self, model, x):
        samples = self._get_integrand_samples(model, x)
        y_max = self._get_y_max(model)
        f_max_over_q = samples.max(axis=-1)
        return (f_max_over_q - y_max) * (f_max_over_q > 0)
