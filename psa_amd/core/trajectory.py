"""
Input container of the SED path.

Mirror of `psa.core.trajectory.Trajectory` (reference src/psa/core/trajectory.py:8-45):
same field order (it is constructed positionally), same validation messages, same
`n_frames` / `n_atoms` properties.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def _require(ok: bool, message: str) -> None:
    if not ok:
        raise ValueError(message)


@dataclass
class Trajectory:
    positions: np.ndarray     # (frames, atoms, 3)
    velocities: np.ndarray    # (frames, atoms, 3)
    types: np.ndarray         # (atoms,)
    timesteps: np.ndarray     # (frames,)
    box_matrix: np.ndarray    # (3, 3), rows are the cell vectors
    box_lengths: np.ndarray   # (3,)
    box_tilts: np.ndarray     # (3,)
    dt_ps: float              # timestep in picoseconds

    def __post_init__(self):
        for label, arr in (("Positions", self.positions), ("Velocities", self.velocities)):
            _require(arr.ndim == 3 and arr.shape[2] == 3,
                     f"{label} must be 3D (frames, atoms, xyz) and last dimension must be 3.")
        _require(self.types.ndim == 1, "Types must be 1D")
        _require(self.timesteps.ndim == 1, "Timesteps must be 1D")
        frames = {self.positions.shape[0], self.velocities.shape[0], len(self.timesteps)}
        _require(len(frames) == 1, "Frame count mismatch: positions, velocities, timesteps.")
        atoms = {self.positions.shape[1], self.velocities.shape[1], len(self.types)}
        _require(len(atoms) == 1, "Atom count mismatch: positions, velocities, types.")
        _require(self.box_matrix.shape == (3, 3),
                 f"Box matrix must be 3x3, got {self.box_matrix.shape}")
        _require(self.box_lengths.shape == (3,),
                 f"Box lengths must be a 3-element array, got {self.box_lengths.shape}")
        _require(self.box_tilts.shape == (3,),
                 f"Box tilts must be a 3-element array, got {self.box_tilts.shape}")

    @property
    def n_frames(self) -> int:
        return len(self.timesteps)

    @property
    def n_atoms(self) -> int:
        return len(self.types)
