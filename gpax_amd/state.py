"""Module-global state: verbosity, device, chained RNG key (mirror of ``gpax/state.py``)."""

import sys
from contextlib import contextmanager

import numpy as np

_this = sys.modules[__name__]
_KEY_MAX = 2**32 - 1
_VERBOSITY_LEVELS = {"debug": 2, "normal": 1, "silent": 0}

verbose = 1
device = 0          # HIP device ordinal (gpax/state.py:61-70 holds a jax device)
rng_key = 0
warnings = []
_cores = {}


def set_verbosity(mode="normal"):
    """gpax/state.py:18-27."""
    assert isinstance(mode, str)
    _this.verbose = _VERBOSITY_LEVELS[mode.lower()]


@contextmanager
def temporary_verbosity(mode):
    prev = _this.verbose
    set_verbosity(mode)
    try:
        yield
    finally:
        _this.verbose = prev


def silent_mode():
    """gpax/state.py:30-43."""
    return temporary_verbosity("silent")


def debug_mode():
    """gpax/state.py:45-58."""
    return temporary_verbosity("debug")


def set_device(dev):
    """gpax/state.py:61-70.  Accepts an int ordinal or "gpu"/"cuda:0"-style strings.  Unlike the
    reference there is no CPU fallback: the HIP library is the only backend."""
    if isinstance(dev, str):
        dev = int(dev.split(":")[1]) if ":" in dev else 0
    _this.device = int(dev)


def set_rng_key(key: int):
    """gpax/state.py:73-84."""
    assert isinstance(key, int) and 0 <= key < _KEY_MAX
    _this.rng_key = key


def get_rng_key():
    """gpax/state.py:87-126: every call advances the key by seeding numpy with the current key and
    drawing the next one.  Returns ``(new_key, numpy Generator)`` -- the Generator stands where the
    reference returns a jax PRNG key (the random streams differ: statistical parity only)."""
    np.random.seed(_this.rng_key)
    new_key = int(np.random.randint(low=0, high=_KEY_MAX))
    _this.rng_key = new_key
    return new_key, np.random.default_rng(new_key)


def get_core():
    """One ``gpax_amd.Core`` (HIP handle) per device, created on first use."""
    from gpax_amd._lib import Core

    if _this.device not in _this._cores:
        _this._cores[_this.device] = Core(_this.device)
    return _this._cores[_this.device]
