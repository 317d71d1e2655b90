"""
LAMMPS text dump of a reconstructed motion (mirror of `psa.io.writer.out_to_qdump`,
reference src/psa/io/writer.py:139-228): the file `SEDCalculator.ised` produces.
"""
from __future__ import annotations

import logging
from pathlib import Path

import numpy as np

logger = logging.getLogger(__name__)


def _box_header(box_matrix: np.ndarray) -> str:
    """`ITEM: BOX BOUNDS` block.  Origin at 0, extents from the matrix diagonal, tilt factors
    xy, xz, yz = matrix[0,1], [0,2], [1,2]; an orthogonal box is written in the short form."""
    hi = [box_matrix[i, i] for i in range(3)]
    xy, xz, yz = box_matrix[0, 1], box_matrix[0, 2], box_matrix[1, 2]
    if all(np.isclose(t, 0.0) for t in (xy, xz, yz)):
        return "ITEM: BOX BOUNDS pp pp pp\n" + "".join(f"{0.0:.8f} {h:.8f}\n" for h in hi)
    # LAMMPS bounding box of a tilted cell
    x_lo, x_hi = min(0.0, xy, xz, xy + xz), hi[0] + max(0.0, xy, xz, xy + xz)
    y_lo, y_hi = min(0.0, yz), hi[1] + max(0.0, yz)
    return ("ITEM: BOX BOUNDS xy xz yz pp pp pp\n"
            f"{x_lo:.8f} {x_hi:.8f} {xy:.8f}\n"
            f"{y_lo:.8f} {y_hi:.8f} {xz:.8f}\n"
            f"{0.0:.8f} {hi[2]:.8f} {yz:.8f}\n")


def out_to_qdump(filename: str, positions_tf: np.ndarray, types_tf: np.ndarray, box_matrix: np.ndarray):
    """Write (frames, atoms, 3) positions as a LAMMPS dump: one TIMESTEP block per frame, atom ids
    1..N, integer types, coordinates with six decimals."""
    n_frames, n_atoms, _ = positions_tf.shape
    Path(filename).parent.mkdir(parents=True, exist_ok=True)
    box = _box_header(box_matrix)
    ids = np.arange(1, n_atoms + 1)
    kinds = np.asarray(types_tf).astype(int)
    with open(filename, "w") as fh:
        for frame in range(n_frames):
            fh.write(f"ITEM: TIMESTEP\n{frame}\nITEM: NUMBER OF ATOMS\n{n_atoms}\n{box}ITEM: ATOMS id type x y z\n")
            xyz = positions_tf[frame]
            fh.writelines(f"{i} {k} {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n" for i, k, p in zip(ids, kinds, xyz))
    logger.debug("Wrote iSED reconstruction to Qdump: %s", filename)
