"""On-disk formats either side of the SED path (mirror of the `.npy` cache of `psa.io.loader`)."""
from .npy_cache import load_trajectory_npy, save_trajectory_npy

__all__ = ["load_trajectory_npy", "save_trajectory_npy"]
