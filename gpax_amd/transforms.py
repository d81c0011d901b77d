"""Data transforms (host NumPy; O(N d), outside the HIP path).  Mirror of ``gpax/transforms.py``
including its epsilon asymmetries between forward and reverse."""

from __future__ import annotations

import numpy as np

_EPSILON = 1e-8  # gpax/transforms.py:58
_TMIN, _TMAX = -1.0, 1.0  # gpax/transforms.py:222-223


def _ensure_2d(x):
    x = np.asarray(x, dtype=np.float64)
    return x.reshape(-1, 1) if x.ndim == 1 else x


class Transform:
    """gpax/transforms.py:67-197: ``fit`` once, then ``forward`` / ``reverse`` with
    ``transforms_as`` in {"mean", "std"}."""

    def __init__(self):
        self._is_fit = False

    def _fit(self, x):
        raise NotImplementedError

    def fit(self, x):
        if self._is_fit:
            raise RuntimeError("Cannot call fit on a Transform more than once")
        self._is_fit = True
        if x is None:
            return
        self._fit(_ensure_2d(x))

    def _apply(self, x, transforms_as, mean_fn, std_fn):
        if not self._is_fit:
            raise RuntimeError("Cannot execute transform before fit is called")
        if x is None:
            return None
        x = _ensure_2d(x)
        if transforms_as == "mean":
            return mean_fn(x)
        if transforms_as == "std":
            return std_fn(x)
        raise ValueError(f"arr has invalid transform type {transforms_as}")

    def forward(self, x, transforms_as="mean"):
        return self._apply(x, transforms_as, self._forward_mean, self._forward_std)

    def reverse(self, x, transforms_as="mean"):
        return self._apply(x, transforms_as, self._reverse_mean, self._reverse_std)


class IdentityTransform(Transform):
    """gpax/transforms.py:200-219."""

    def _fit(self, _):
        return

    def _forward_mean(self, x):
        return x

    _forward_std = _reverse_mean = _reverse_std = _forward_mean


class ScaleTransform(Transform):
    """gpax/transforms.py:227-274: min/max scaling to [-1, 1] per dimension."""

    def __init__(self):
        super().__init__()
        self._minima = None
        self._maxima = None

    def _fit(self, x):
        self._minima = x.min(axis=0, keepdims=True)
        self._maxima = x.max(axis=0, keepdims=True)

    def _forward_mean(self, x):
        delta_r = self._maxima - self._minima + _EPSILON
        return (x - self._minima) / delta_r * (_TMAX - _TMIN) + _TMIN

    def _forward_std(self, x):
        delta_r = self._maxima - self._minima + _EPSILON
        return x / delta_r * (_TMAX - _TMIN)

    def _reverse_mean(self, x):
        delta_r = self._maxima - self._minima + _EPSILON
        delta_t = _TMAX - _TMIN + _EPSILON
        return (x * delta_r + self._minima * delta_t - _TMIN * delta_r) / delta_t

    def _reverse_std(self, x):
        delta_r = self._maxima - self._minima + _EPSILON
        delta_t = _TMAX - _TMIN + _EPSILON
        return x * delta_r / delta_t


class NormalizeTransform(Transform):
    """gpax/transforms.py:277-317: standardisation (forward "mean" divides by std + eps, forward
    "std" by std alone)."""

    def __init__(self):
        super().__init__()
        self._mean = None
        self._std = None

    def _fit(self, x):
        self._mean = x.mean(axis=0, keepdims=True)
        self._std = x.std(axis=0, keepdims=True)

    def _forward_mean(self, x):
        return (x - self._mean) / (self._std + _EPSILON)

    def _forward_std(self, x):
        return x / self._std

    def _reverse_mean(self, x):
        return x * (self._std + _EPSILON) + self._mean

    def _reverse_std(self, x):
        return x * self._std
