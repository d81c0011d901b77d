"""gpax_amd -- MI355X-native (gfx950) exact-GP core behind gpax's Python surface.

The arithmetic of the hot path (gram build, Cholesky, log-marginal likelihood,
predictive posterior, MVN draws) runs in hand-written HIP kernels reached through
the C ABI of ``include/gpx.h`` (``gpax_amd/_lib.py``).  There is no CPU fallback.
"""

from gpax_amd import state  # noqa: F401
from gpax_amd._lib import MATERN52, PERIODIC, RBF, Core, GpxError, make_params  # noqa: F401
from gpax_amd.gp import ExactGP, GPSample, VariationalInferenceGP  # noqa: F401
from gpax_amd.kernels import MaternKernel, PeriodicKernel, RBFKernel  # noqa: F401

__version__ = "0.1.0"
