"""Shared pytest plumbing: the ``gpu`` marker, repo-root imports, golden-fixture loader."""

import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _unpack_tril(packed):
    n = int(round((np.sqrt(8 * packed.size + 1) - 1) / 2))
    out = np.zeros((n, n))
    out[np.tril_indices(n)] = packed
    return out


def load_golden(name):
    """Load ``tests/golden/<name>.npz`` into a dict; K (symmetric) and L (lower) are
    stored as packed lower triangles and unpacked here."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    L = _unpack_tril(g.pop("L_tril"))
    kl = _unpack_tril(g.pop("K_tril"))
    g["L"] = L
    g["K"] = kl + np.tril(kl, -1).T
    g["params"] = {k[3:]: g[k] for k in list(g) if k.startswith("hp_")}
    g["kind"] = int(g["kind"])
    g["observation_noise"] = bool(int(g["observation_noise"]))
    g["y_std"] = g.get("y_std")
    g["name"] = name
    return g


GOLDEN_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return load_golden(request.param)
