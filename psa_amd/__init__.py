"""
psa_amd -- MI355X-native implementation of the PSA spectral-energy-density hot path.

Switching from the reference is an import change:

    from psa import Trajectory, SED, SEDCalculator          # h-walk/PSA (NumPy, CPU)
    from psa_amd import Trajectory, SED, SEDCalculator      # this package (HIP, gfx950)

Only what the SED path needs is here: the three core classes,
`parse_direction`, the ctypes binding of libpsa_hip.so, k-point sharding over RCCL and the
synthetic-trajectory generator used by the benchmark.  Loaders, plotting, CLI and GUI of the
reference are out of scope.
"""
from .core import SED, SEDCalculator, Trajectory
from .core.sed import fast_intensity
from .utils.helpers import parse_direction

__version__ = "0.2.0"
__all__ = ["Trajectory", "SED", "SEDCalculator", "parse_direction", "fast_intensity", "__version__"]
