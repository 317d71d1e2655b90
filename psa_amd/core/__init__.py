"""Core of the SED path (mirror of `psa.core`, reference src/psa/core/__init__.py:7-17)."""
from .sed import SED
from .sed_calculator import SEDCalculator
from .trajectory import Trajectory

__all__ = ["Trajectory", "SED", "SEDCalculator"]
