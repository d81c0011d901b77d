"""Prior plumbing: ``Parameter``, ``Prior`` and the few distributions gpax's kernels use.

Host-side mirror of ``gpax/utils/prior_utils.py:96-185`` without NumPyro: a parameter is
either a constant or a distribution object with ``sample``, ``log_prob``, ``median`` and a
bijection to unconstrained space (what NumPyro's NUTS works in).  No arithmetic of the hot
path lives here -- it only carries hyper-parameter dictionaries to the HIP library.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
from scipy.special import erfinv


class Distribution:
    """Minimal distribution interface (the subset of numpyro.distributions gpax touches)."""

    def sample(self, rng, shape=()):
        raise NotImplementedError

    def log_prob(self, x):
        raise NotImplementedError

    @property
    def median(self):
        raise NotImplementedError

    @property
    def mean(self):
        raise NotImplementedError

    # bijection unconstrained u -> constrained x, with log|dx/du|
    def from_unconstrained(self, u):
        return np.exp(u), u  # positive support by default

    def to_unconstrained(self, x):
        return np.log(x)

    def grad_log_prob(self, x):
        """d log_prob / dx (central differences unless a sub-class knows better)."""
        x = np.asarray(x, dtype=np.float64)
        h = 1e-6 * np.maximum(1.0, np.abs(x))
        return (self.log_prob(x + h) - self.log_prob(x - h)) / (2 * h)

    def grad_terms(self, u):
        """What a gradient-based sampler needs at the unconstrained point ``u``:
        ``(x, dx/du, d/du [log_prob(x(u)) + log|dx/du|])`` -- the role of jax.grad through NumPyro's
        biject_to transforms."""
        u = np.asarray(u, dtype=np.float64)
        x, _ = self.from_unconstrained(u)
        dx = self._dx_du(u, x)
        return x, dx, self.grad_log_prob(x) * dx + self._dlogjac_du(u, x)

    def _dx_du(self, u, x):
        return x  # exp

    def _dlogjac_du(self, u, x):
        return np.ones_like(u)


@dataclass
class HalfNormal(Distribution):
    """numpyro.distributions.HalfNormal(scale): prior of ``k_noise`` (gpax/kernels.py:69-72)."""

    scale: float = 1.0

    def sample(self, rng, shape=()):
        return np.abs(rng.standard_normal(shape)) * self.scale

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.where(x >= 0, 0.5 * math.log(2 / math.pi) - math.log(self.scale) - 0.5 * (x / self.scale) ** 2, -np.inf)

    def grad_log_prob(self, x):
        return -np.asarray(x, dtype=np.float64) / self.scale**2

    def fast_terms(self, u):
        """``grad_terms`` + log-density in one pass for the exp bijection: ``(x, dx/du, d/du[...], sum(log_prob(x) + u))``
        with one ``exp`` -- what a NUTS gradient call needs per parameter (the general path costs ~15 us per parameter)."""
        x = np.exp(u)
        q = (x / self.scale) ** 2
        lp = u.size * (0.5 * math.log(2 / math.pi) - math.log(self.scale)) + float(u.sum() - 0.5 * q.sum())
        return x, x, 1.0 - q, lp

    @property
    def median(self):
        return self.scale * math.sqrt(2.0) * float(erfinv(0.5))

    @property
    def mean(self):
        return self.scale * math.sqrt(2.0 / math.pi)


@dataclass
class LogNormal(Distribution):
    """numpyro.distributions.LogNormal(loc, scale): prior of ``k_scale`` / ``k_length``
    (gpax/kernels.py:101-107)."""

    loc: float = 0.0
    scale: float = 1.0

    def sample(self, rng, shape=()):
        return np.exp(self.loc + self.scale * rng.standard_normal(shape))

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        lx = np.log(x)
        return -lx - math.log(self.scale) - 0.5 * math.log(2 * math.pi) - 0.5 * ((lx - self.loc) / self.scale) ** 2

    def grad_log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -(1.0 + (np.log(x) - self.loc) / self.scale**2) / x

    def fast_terms(self, u):
        """See ``HalfNormal.fast_terms``: with x = exp(u), log_prob(x) + u = -log(scale) - log(2 pi)/2 - z^2/2, z = (u - loc)/scale."""
        x = np.exp(u)
        z = (u - self.loc) / self.scale
        lp = -u.size * (math.log(self.scale) + 0.5 * math.log(2 * math.pi)) - 0.5 * float((z * z).sum())
        return x, x, -z / self.scale, lp

    @property
    def median(self):
        return math.exp(self.loc)

    @property
    def mean(self):
        return math.exp(self.loc + 0.5 * self.scale**2)


@dataclass
class Uniform(Distribution):
    """numpyro.distributions.Uniform(low, high) (examples/04_Structured_GP.ipynb cell 10: ``m_t``)."""

    low: float = 0.0
    high: float = 1.0

    def sample(self, rng, shape=()):
        return rng.uniform(self.low, self.high, shape)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.where((x >= self.low) & (x <= self.high), -math.log(self.high - self.low), -np.inf)

    @property
    def median(self):
        return 0.5 * (self.low + self.high)

    mean = median

    def from_unconstrained(self, u):
        s = 1.0 / (1.0 + np.exp(-u))
        return self.low + (self.high - self.low) * s, np.log(self.high - self.low) + np.log(s) + np.log1p(-s)

    def to_unconstrained(self, x):
        s = (np.asarray(x, dtype=np.float64) - self.low) / (self.high - self.low)
        return np.log(s) - np.log1p(-s)

    def grad_log_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))

    def _dx_du(self, u, x):
        s = 1.0 / (1.0 + np.exp(-u))
        return (self.high - self.low) * s * (1.0 - s)

    def _dlogjac_du(self, u, x):
        return 1.0 - 2.0 / (1.0 + np.exp(-u))


@dataclass
class Normal(Distribution):
    loc: float = 0.0
    scale: float = 1.0

    def sample(self, rng, shape=()):
        return self.loc + self.scale * rng.standard_normal(shape)

    def log_prob(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -math.log(self.scale) - 0.5 * math.log(2 * math.pi) - 0.5 * ((x - self.loc) / self.scale) ** 2

    @property
    def median(self):
        return self.loc

    mean = median

    def from_unconstrained(self, u):
        return u, np.zeros_like(u)

    def to_unconstrained(self, x):
        return np.asarray(x, dtype=np.float64)

    def grad_log_prob(self, x):
        return -(np.asarray(x, dtype=np.float64) - self.loc) / self.scale**2

    def _dx_du(self, u, x):
        return np.ones_like(u)

    def _dlogjac_du(self, u, x):
        return np.zeros_like(u)


@dataclass
class Parameter:
    """gpax/utils/prior_utils.py:96-117: a distribution (learnable) or a float/int (constant),
    with ``plate`` independent copies."""

    value: object
    plate: int = 1

    def __post_init__(self):
        if not isinstance(self.plate, int) or self.plate < 1:
            raise ValueError("plate must be an int >= 1")
        if not isinstance(self.value, (Distribution, float, int)):
            raise ValueError(
                f"Provided distribution {self.value} was not of type float, int, or Distribution"
            )  # gpax/utils/prior_utils.py:60-65

    @property
    def is_constant(self):
        return not isinstance(self.value, Distribution)


class Prior:
    """gpax/utils/prior_utils.py:120-185.  Sub-classes set ``attribute_prefix`` ("k_" / "m_") and
    keep their Parameters as attributes with that prefix."""

    attribute_prefix = ""

    @property
    def _params(self):
        out = {}
        for key, obj in vars(self).items():
            if key.startswith(self.attribute_prefix) and isinstance(obj, Parameter):
                out[key] = obj
        return out

    @property
    def params(self):
        """name -> distribution or constant (gpax/utils/prior_utils.py:157-166)."""
        return {k: v.value for k, v in self._params.items()}

    def sample_parameters(self, rng, num_samples=None):
        """Draws from the priors: name -> array ``(plate,)`` or ``(num_samples, plate)``; constants
        are repeated (numpyro.deterministic, gpax/utils/prior_utils.py:57-58)."""
        out = {}
        for k, p in self._params.items():
            shape = (p.plate,) if num_samples is None else (num_samples, p.plate)
            if p.is_constant:
                out[k] = np.full(shape, float(p.value))
            else:
                out[k] = np.asarray(p.value.sample(rng, shape), dtype=np.float64)
        return out

    def get_sample_parameter_means(self):
        """gpax/utils/prior_utils.py:177-185."""
        return {k: (float(v.value) if v.is_constant else v.value.mean) for k, v in self._params.items()}
