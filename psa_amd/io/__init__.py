"""On-disk formats either side of the SED path: the `.npy` trajectory cache of `psa.io.loader`
and the LAMMPS dump `psa.io.writer.out_to_qdump` writes for iSED."""
from .npy_cache import load_trajectory_npy, save_trajectory_npy
from .writer import out_to_qdump

__all__ = ["load_trajectory_npy", "save_trajectory_npy", "out_to_qdump"]
