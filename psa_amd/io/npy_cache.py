"""
The trajectory `.npy` cache of the reference, without OVITO.

`psa.io.loader.TrajectoryLoader` keeps a parsed trajectory next to the input file as
`<stem>.positions.npy`, `.velocities.npy`, `.types.npy`, `.box_matrix.npy` (reference
src/psa/io/loader.py:48-79 reads them, :363-387 writes them together with
`.mean_positions.npy` and `.displacements.npy`).  These two functions read and write exactly
that layout, so a trajectory parsed once by PSA can be fed to `psa_amd` on a machine that has
no OVITO.  Large arrays are memory-mapped by default: `Engine.ensure_resident` then streams
them from the page cache to HBM in 256 MiB pieces without a second host copy.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Union

import numpy as np

from ..core.trajectory import Trajectory

logger = logging.getLogger(__name__)

_FIELDS = ("positions", "velocities", "types", "box_matrix")


def _stem(filepath: Union[str, Path]) -> Path:
    """`dump.lammpstrj` -> `dump` in the same directory (loader.py:48)."""
    filepath = Path(filepath)
    return filepath.parent / filepath.stem


def cache_files(filepath: Union[str, Path]) -> dict:
    stem = _stem(filepath)
    return {f: stem.with_suffix(f".{f}.npy") for f in _FIELDS}


def load_trajectory_npy(filepath: Union[str, Path], dt: float, mmap: bool = True) -> Trajectory:
    """Trajectory from the `.npy` cache that belongs to `filepath` (the original trajectory file
    name, or any name with the same stem).  `dt` is the timestep in ps, as for TrajectoryLoader.
    Raises FileNotFoundError when the cache is incomplete (the reference falls back to OVITO
    there; this package has no OVITO path)."""
    files = cache_files(filepath)
    missing = [str(p) for p in files.values() if not p.exists()]
    if missing:
        raise FileNotFoundError(f"No complete .npy cache for {Path(filepath).name}: missing {missing}")
    mode = "r" if mmap else None
    pos = np.load(files["positions"], mmap_mode=mode)
    vel = np.load(files["velocities"], mmap_mode=mode)
    types = np.load(files["types"])
    box = np.load(files["box_matrix"])
    if box.shape != (3, 3):
        raise ValueError(f"Cached box_matrix has shape {box.shape}, expected (3,3).")
    # lengths / tilts exactly as the reference derives them from the cached matrix (:66-67)
    lengths = np.array([box[0, 0], box[1, 1], box[2, 2]], dtype=np.float32)
    tilts = np.array([box[0, 1], box[0, 2], box[1, 2]], dtype=np.float32)
    steps = np.arange(pos.shape[0], dtype=np.float32) * dt
    return Trajectory(pos, vel, types, steps, box_matrix=box, box_lengths=lengths, box_tilts=tilts,
                      dt_ps=dt)


def save_trajectory_npy(traj: Trajectory, filepath: Union[str, Path], derived: bool = True) -> bool:
    """Write the cache for `filepath`; like the reference, an existing complete cache is left
    alone (returns False).  `derived` also writes `.mean_positions.npy` (float64 mean, as
    loader.py:383) and `.displacements.npy`."""
    files = cache_files(filepath)
    if all(p.exists() for p in files.values()):
        logger.info(".npy cache for %s exists; skipping save.", Path(filepath).name)
        return False
    stem = _stem(filepath)
    stem.parent.mkdir(parents=True, exist_ok=True)
    for field in _FIELDS:
        np.save(files[field], getattr(traj, field))
    if derived:
        mean = np.mean(traj.positions, axis=0)
        np.save(stem.with_suffix(".mean_positions.npy"), mean)
        np.save(stem.with_suffix(".displacements.npy"), traj.positions - mean[None, :, :])
    return True
