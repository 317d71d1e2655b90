"""
SEDCalculator -- host side of the MI355X SED path.

Drop-in for `psa.core.sed_calculator.SEDCalculator` (reference
src/psa/core/sed_calculator.py): same constructor, same `get_k_path` / `get_k_grid` /
`calculate` / `calculate_chiral_phase` signatures, attributes (`a1..a3`, `b1..b3`,
`recip_vecs_prim`, `dt_ps`, `traj`, `use_displacements`) and ValueErrors.  What differs is
where the arithmetic runs: the reference's `_calculate_sed_for_group` (:58-84, NumPy
einsum + pocketfft) is replaced by libpsa_hip.so (phase table -> split-precision f16-MFMA projection, fp32-equivalent ->
batched rocFFT -> epilogue) through `psa_amd._hip.Engine`.  There is no CPU path here.

Residency: the first `calculate` uploads the velocity array (positions with
`use_displacements=True`) to HBM -- projecting the frames of each chunk as it lands -- and later
calls reuse it.  "The same array" is decided by object identity plus a hash of ~2000 sampled
elements, so in-place edits are normally noticed; after editing a trajectory array in place call
`calculator.invalidate()` to be certain (the reference re-reads the array on every call).

`calculate_kpath_sed` / `calculate_kgrid_sed` / `calculate_chiral_sed` are the composites
the reference's README names (README.md:100-140) and its GUI implements privately
(src/psa/gui/psa_gui.py:923-1017, :2099-2247): k generator -> `calculate` -> optional
chiral phase -> `SED`.
"""
from __future__ import annotations

import logging
import weakref
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .. import _hip
from ..io.writer import out_to_qdump
from ..utils.helpers import parse_direction
from .sed import SED
from .trajectory import Trajectory

logger = logging.getLogger(__name__)

_MODES = ("coherent", "incoherent")
# Cartesian components whose phase difference defines chirality about an axis
# (psa_gui.py:975-980)
_CHIRAL_PAIR = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}


class SEDCalculator:
    def __init__(self, traj: Trajectory, nx: int, ny: int, nz: int,
                 use_displacements: bool = False, dt_ps: Optional[float] = None):
        if not (nx > 0 and ny > 0 and nz > 0):
            raise ValueError("System dimensions (nx, ny, nz) must be positive.")
        self.traj = traj
        self.use_displacements = use_displacements

        # timestep: explicit argument (deprecated in the reference, :26-32) wins
        if dt_ps is not None:
            logger.warning("Explicitly providing dt_ps to SEDCalculator is deprecated. "
                           "The provided dt_ps will override the Trajectory's dt_ps.")
            self.dt_ps = dt_ps
        elif getattr(traj, "dt_ps", None) is not None:
            self.dt_ps = traj.dt_ps
        else:
            raise ValueError("Timestep dt_ps not found in Trajectory object and not provided to SEDCalculator.")
        if self.dt_ps <= 0:
            raise ValueError("Timestep dt_ps must be positive.")

        # primitive cell = rows of the box matrix divided by the replication counts (:40-41)
        self.a1, self.a2, self.a3 = (traj.box_matrix[i, :] / n for i, n in enumerate((nx, ny, nz)))
        if min(np.linalg.norm(v) for v in (self.a1, self.a2, self.a3)) < 1e-9:
            raise ValueError("One or more primitive vectors (a1,a2,a3) near zero. Check nx,ny,nz or box matrix.")
        volume = np.abs(np.dot(self.a1, np.cross(self.a2, self.a3)))
        if np.isclose(volume, 0):
            cell = np.vstack([self.a1, self.a2, self.a3])
            if np.linalg.matrix_rank(cell) < 3 or np.isclose(np.linalg.det(cell), 0):
                raise ValueError(f"Primitive cell vectors coplanar/collinear; volume zero ({volume:.2e}).")
            logger.warning("Primitive cell volume very small (%.2e).", volume)
        scale = 2 * np.pi / volume
        self.b1 = scale * np.cross(self.a2, self.a3)
        self.b2 = scale * np.cross(self.a3, self.a1)
        self.b3 = scale * np.cross(self.a1, self.a2)
        self.recip_vecs_prim = np.vstack([self.b1, self.b2, self.b3]).astype(np.float32)

        self._engine: Optional[_hip.Engine] = None
        self._shard = None                    # psa_amd.dist.KShardGroup when k-sharding
        self._mean_cache = None               # (weakref to positions, mean array)

    # ------------------------------------------------------------------ device plumbing
    @property
    def engine(self) -> "_hip.Engine":
        """The GPU context (created on first use; raises if libpsa_hip / a GPU is missing)."""
        if self._engine is None:
            self._engine = _hip.Engine()
        return self._engine

    def attach(self, engine=None, shard_group=None) -> "SEDCalculator":
        """Use an existing Engine and/or shard the k-points over a `dist.KShardGroup`."""
        if engine is not None:
            self._engine = engine
        if shard_group is not None:
            self._shard = shard_group
            self._engine = shard_group.engine
        return self

    def close(self):
        if self._engine is not None and self._shard is None:
            self._engine.close()
        self._engine = None

    def invalidate(self):
        """Forget everything cached about the trajectory arrays -- the copies resident in HBM and
        the mean positions.  Call it after modifying `traj.positions` / `traj.velocities` in
        place; the next `calculate` uploads afresh."""
        self._mean_cache = None
        if self._engine is not None:
            self._engine.invalidate()

    def _mean_positions(self) -> np.ndarray:
        """np.mean(positions, axis=0, dtype=float32) exactly as the reference (:205), cached per
        positions array because it is a full pass over (T,N,3).  In displacement mode the
        positions have to be in HBM anyway, so the pass runs there (mean_over_frames_kernel adds
        the frames in the same order into a float32 accumulator: bit-identical); otherwise it
        is the reference's own NumPy call on the host."""
        pos = self.traj.positions
        stamp = _hip.Engine._fingerprint(pos) if isinstance(pos, np.ndarray) and pos.size else None
        if self._mean_cache is not None and self._mean_cache[0]() is pos and self._mean_cache[2] == stamp:
            return self._mean_cache[1]
        frame_sharded = self._shard is not None and self._shard.nranks > 1 and self._shard.mode != "k"
        if self.use_displacements and not frame_sharded:
            self.engine.ensure_resident(_hip.SLOT_POSITIONS, pos)
            mean = self.engine.mean_positions(_hip.SLOT_POSITIONS)
        elif (isinstance(pos, np.ndarray) and pos.dtype == np.float32 and pos.flags.c_contiguous and pos.ndim == 3
              and pos.nbytes >= (64 << 20)):
            # velocity mode: the positions stay on the host; a large array is averaged by the library's
            # host threads -- the same sequential float32 chain per column as np.mean, bit for bit
            mean = _hip.host_mean_frames(pos)
        else:               # (a frame-sharded rank holds only its own frames in HBM)
            mean = np.mean(pos, axis=0, dtype=np.float32)
        try:
            self._mean_cache = (weakref.ref(pos), mean, stamp)
        except TypeError:
            self._mean_cache = None
        return mean

    def _data_slot(self):
        if self.use_displacements:
            return _hip.SLOT_POSITIONS, self.traj.positions, _hip.F_DISPLACEMENTS
        return _hip.SLOT_VELOCITIES, self.traj.velocities, 0

    def _device_groups(self, groups: Sequence[np.ndarray]):
        """None (= all atoms in order, the coalesced fast path) when that is what the single
        group is; the index lists otherwise."""
        n = self.traj.n_atoms
        if len(groups) == 1 and groups[0].size == n and np.array_equal(groups[0], np.arange(n)):
            return None
        return [np.asarray(g) for g in groups]

    def _run_device(self, k_vectors: np.ndarray, groups, intensity: bool, mean_pos_all,
                    fetch: bool = True):
        """(result, sum_c |result|^2 or None): the second array accompanies a complex result -- it is
        what `SED.intensity` returns (core/sed.py:22-24), produced on the device in the pass that
        writes the result."""
        slot, data, flags = self._data_slot()
        if intensity:
            flags |= _hip.F_INTENSITY
        want = not intensity
        eng = self.engine
        with eng.lock:                       # project + finalize must not interleave across threads
            K = len(k_vectors)
            T = self.traj.n_frames
            if self._shard is not None and self._shard.nranks > 1:
                out = self._shard.run(slot, data, mean_pos_all, k_vectors, groups, flags, T, fetch, with_intensity=want)
            elif not eng.is_resident(slot, data):
                # first call on this array: upload and project, overlapped
                eng.project_upload(slot, data, mean_pos_all, k_vectors, groups, flags)
                out = eng.finalize(T, K, intensity, fetch, with_intensity=want)
            elif fetch:                      # one library call; long complex results leave block by block
                out = eng.calculate(slot, mean_pos_all, k_vectors, groups, flags, with_intensity=want)
            else:
                eng.project(slot, mean_pos_all, k_vectors, groups, flags)
                out = eng.finalize(T, K, intensity, False, with_intensity=want)
        if want:
            return out if out is not None else (None, None)
        return out, None

    # ------------------------------------------------------------------ the seam
    def _calculate_sed_for_group(self, k_vectors_3d: np.ndarray, group_atom_indices: np.ndarray,
                                 mean_pos_all: np.ndarray) -> np.ndarray:
        """Complex SED (T,K,3) of one atom group -- the reference's :58-84, on the GPU."""
        n_t = self.traj.n_frames
        idx = np.asarray(group_atom_indices)
        if idx.size == 0 or len(k_vectors_3d) == 0:
            return np.zeros((n_t, len(k_vectors_3d), 3), dtype=np.complex64)
        return self._run_device(np.asarray(k_vectors_3d), self._device_groups([idx]), False,
                                mean_pos_all)[0]

    # ------------------------------------------------------------------ k generators
    def get_k_path(self, direction_spec: Union[str, int, float, List[float], Dict[str, float], np.ndarray],
                   bz_coverage: float, n_k: int, lat_param: Optional[float] = None
                   ) -> Tuple[np.ndarray, np.ndarray]:
        """(|k| (n_k,), k (n_k,3)) float32 along a direction (reference :86-125)."""
        k_hat = parse_direction(direction_spec)
        if lat_param is None or lat_param <= 1e-6:
            # extent of the reciprocal cell along k_hat: largest |k_hat . b_i|  (:94-104)
            proj = [float(abs(np.dot(k_hat, b))) for b in (self.b1, self.b2, self.b3)]
            extent = max(proj)
            if extent > 1e-6:
                logger.info("Using directional reciprocal lattice projection (%.3f 2pi/A) for k-path.", extent)
            else:
                len_a1 = np.linalg.norm(self.a1)
                if not len_a1 > 1e-6:
                    raise ValueError("Invalid/small lattice_param for k-path & reciprocal projections "
                                     "too small for auto-detection.")
                extent = 2 * np.pi / len_a1
                logger.warning("Reciprocal projections too small, using |a1| fallback.")
        else:
            extent = 2 * np.pi / lat_param
        k_max = bz_coverage * extent
        if n_k < 1:
            raise ValueError("n_k (k-points) must be >= 1.")
        if n_k > 1:
            k_mags = np.linspace(0, k_max, n_k, dtype=np.float32)
        else:
            k_mags = np.array([0.0 if np.isclose(k_max, 0) else k_max], dtype=np.float32)
        return k_mags, np.outer(k_mags, k_hat).astype(np.float32)

    def get_k_grid(self, plane: str, k_range_x: Tuple[float, float], k_range_y: Tuple[float, float],
                   n_kx: int, n_ky: int, k_fixed_val: float = 0.0
                   ) -> Tuple[np.ndarray, np.ndarray, Tuple[int, int]]:
        """(empty, k (n_kx*n_ky,3) float32, (n_kx,n_ky)); the first range is the slow index
        (reference :127-180)."""
        if n_kx <= 0 or n_ky <= 0:
            raise ValueError("Number of k-points (n_kx, n_ky) must be positive.")
        # column of the k-vector fed by (first range, second range, fixed value)
        columns = {"xy": (0, 1, 2), "yz": (1, 2, 0), "zx": (2, 0, 1)}.get(plane.lower())
        if columns is None:
            raise ValueError(f"Invalid plane specified: {plane}. Must be 'xy', 'yz', or 'zx'.")
        first = np.linspace(k_range_x[0], k_range_x[1], n_kx, dtype=np.float32)
        second = np.linspace(k_range_y[0], k_range_y[1], n_ky, dtype=np.float32)
        k_vecs = np.empty((n_kx, n_ky, 3), dtype=np.float32)
        k_vecs[:, :, columns[0]] = first[:, None]
        k_vecs[:, :, columns[1]] = second[None, :]
        k_vecs[:, :, columns[2]] = k_fixed_val
        return np.array([], dtype=np.float32), k_vecs.reshape(-1, 3), (n_kx, n_ky)

    # ------------------------------------------------------------------ atom groups
    def _resolve_groups(self, basis_atom_indices, basis_atom_types, summation_mode) -> List[np.ndarray]:
        """Atom-index arrays, one per group, with the reference's precedence and fallbacks
        (:208-266): types win over indices; flat type lists split per type only when
        incoherent; groups without atoms are dropped; nothing left -> all atoms."""
        n_atoms = self.traj.n_atoms
        groups: List[np.ndarray] = []

        def nested(seq, what):
            if all(isinstance(v, list) for v in seq):
                return True
            if all(isinstance(v, int) for v in seq):
                return False
            raise ValueError(f"{what} must be a list of ints or a list of lists of ints.")

        if basis_atom_types is not None:
            if basis_atom_indices is not None:
                logger.warning("Both basis_atom_types and basis_atom_indices provided. Using basis_atom_types.")
            type_sets: List[List[int]] = []
            if isinstance(basis_atom_types, list) and basis_atom_types:
                if nested(basis_atom_types, "basis_atom_types"):
                    type_sets = basis_atom_types
                elif summation_mode == "incoherent":
                    type_sets = [[t] for t in basis_atom_types]
                else:
                    type_sets = [list(basis_atom_types)]
            elif isinstance(basis_atom_types, int):
                type_sets = [[basis_atom_types]]
            for ts in type_sets:
                members = np.flatnonzero(np.isin(self.traj.types, ts))
                if members.size:
                    groups.append(members)
                else:
                    logger.warning("No atoms found for type group %s. Skipping.", ts)
        elif basis_atom_indices is not None:
            lists: List[np.ndarray] = []
            if isinstance(basis_atom_indices, list):
                if basis_atom_indices:
                    if nested(basis_atom_indices, "basis_atom_indices"):
                        lists = [np.asarray(sub, dtype=int) for sub in basis_atom_indices]
                    else:
                        lists = [np.asarray(basis_atom_indices, dtype=int)]
            elif isinstance(basis_atom_indices, np.ndarray):
                if basis_atom_indices.ndim == 1 and basis_atom_indices.size > 0:
                    lists = [basis_atom_indices.astype(int)]
                else:
                    logger.warning("Unsupported np.ndarray format for basis_atom_indices. "
                                   "Using all atoms if no other basis defined.")
            for members in lists:
                if members.size == 0:
                    continue
                if np.any(members >= n_atoms) or np.any(members < 0):
                    raise ValueError("Atom indices in basis out of bounds.")
                groups.append(members)

        if not groups:
            groups.append(np.arange(n_atoms))
            if summation_mode == "incoherent" and n_atoms > 0:
                logger.info("Using all atoms. Incoherent sum will effectively be a coherent sum of all atoms.")
        return groups

    # ------------------------------------------------------------------ calculate
    def calculate(self, k_points_mags: np.ndarray, k_vectors_3d: np.ndarray,
                  basis_atom_indices: Optional[Union[List[int], List[List[int]], np.ndarray]] = None,
                  basis_atom_types: Optional[Union[List[int], List[List[int]]]] = None,
                  summation_mode: str = 'coherent',
                  k_grid_shape: Optional[Tuple[int, int]] = None,
                  k_chunk_size: int = 500) -> SED:
        """SED of the trajectory at the given k-vectors (reference :182-336).

        coherent (or a single group): `sed` is (T,K,3) complex64; incoherent with several
        groups: (T,K) float32 = sum_g sum_c |S_g|^2.  `k_chunk_size` is accepted for
        compatibility; the GPU handles all k-points in one pass over the trajectory
        instead of re-gathering it per chunk (:287-290), which the reference itself only
        matches to ~8e-7.
        """
        if summation_mode not in _MODES:
            raise ValueError(f"summation_mode must be 'coherent' or 'incoherent', got {summation_mode}")
        n_t, n_atoms = self.traj.n_frames, self.traj.n_atoms
        if n_t == 0 or n_atoms == 0:
            logger.warning("Cannot calculate SED: 0 frames or 0 atoms.")
            return SED(np.array([], dtype=np.complex64).reshape(0, 0, 3), np.array([], dtype=np.float32),
                       k_points_mags, k_vectors_3d, k_grid_shape=k_grid_shape, is_complex=True, phase=None)

        mean_pos_all = self._mean_positions()
        freqs = np.fft.fftfreq(n_t, d=self.dt_ps)
        groups = self._resolve_groups(basis_atom_indices, basis_atom_types, summation_mode)
        is_complex = summation_mode == "coherent" or len(groups) <= 1
        n_k = len(k_vectors_3d)

        if n_k == 0:
            logger.warning("k_vectors_3d is empty. Returning SED object with empty SED data.")
            shape = (n_t, 0, 3) if is_complex else (n_t, 0)
            data, inten = np.zeros(shape, dtype=np.complex64 if is_complex else np.float32), None
        elif is_complex:
            # several coherent groups act as their sorted union (:297-298)
            members = np.unique(np.concatenate(groups)).astype(int) if len(groups) > 1 else groups[0]
            data, inten = self._run_device(np.asarray(k_vectors_3d), self._device_groups([members]), False,
                                           mean_pos_all)
        else:
            data, inten = self._run_device(np.asarray(k_vectors_3d), self._device_groups(groups), True,
                                           mean_pos_all)
        sed = SED(data, freqs, k_points_mags, k_vectors_3d, k_grid_shape=k_grid_shape,
                  is_complex=is_complex, phase=None)
        if inten is not None and isinstance(data, np.ndarray):
            sed._attach_intensity(inten)                 # `sed.intensity` is free: it came with the result
        from . import sed as _sed_module
        if (_sed_module._FAST_INTENSITY and is_complex and n_k and isinstance(data, np.ndarray) and data.ndim == 3
                and self._shard is None and hasattr(self._engine, "intensity_source")):
            # `sed.intensity` right after the calculation is served from the result still on the device
            sed._device_intensity = self._engine.intensity_source(data)
        return sed

    # ------------------------------------------------------------------ chiral phase
    def calculate_chiral_phase(self, Z1: np.ndarray, Z2: np.ndarray, angle_range_opt: str = "C") -> np.ndarray:
        """Phase relation of two complex component arrays as float32 (reference :338-371).

        "C": folded difference of arguments in [-pi/2, pi/2]; "A": angle between the two
        phasors in [0, pi]; "B": signed arcsin of their normalised cross product.  A and B
        are evaluated here as whole-array float32 operations (the reference walks them in
        a Python double loop); entries whose |Z|^2 < 1e-18 are 0.
        """
        if Z1.shape != Z2.shape:
            raise ValueError("Z1 and Z2 shapes must match for chiral phase.")
        if Z1.size == 0:
            return np.array([], dtype=np.float32).reshape(Z1.shape)
        if angle_range_opt == "C":
            diff = np.angle(Z1) - np.angle(Z2)
            diff = (diff + np.pi) % (2 * np.pi) - np.pi
            diff = np.where(diff > np.pi / 2, np.pi - diff, diff)
            diff = np.where(diff < -np.pi / 2, -np.pi - diff, diff)
            return diff.astype(np.float32)
        phase = np.zeros(Z1.shape, dtype=np.float32)
        if angle_range_opt not in ("A", "B"):
            logger.warning("Unknown angle_range_opt '%s'. Angle=0.", angle_range_opt)
            return phase
        re1, im1 = np.float32(1) * Z1.real, np.float32(1) * Z1.imag
        re2, im2 = np.float32(1) * Z2.real, np.float32(1) * Z2.imag
        sq1, sq2 = re1 * re1 + im1 * im1, re2 * re2 + im2 * im2
        usable = (sq1 >= 1e-18) & (sq2 >= 1e-18)
        with np.errstate(invalid="ignore", divide="ignore"):
            norm = np.sqrt(sq1) * np.sqrt(sq2)
            if angle_range_opt == "A":
                full = np.arccos(np.clip((re1 * re2 + im1 * im2) / norm, -1.0, 1.0))
            else:
                full = np.arcsin(np.clip((re1 * im2 - im1 * re2) / norm, -1.0, 1.0))
        phase[usable] = full[usable]
        return phase

    # ------------------------------------------------------------------ composites
    def _finish(self, sed: SED, chiral: bool, chiral_axis: str) -> SED:
        """Attach the option-"C" chiral phase of the component pair for `chiral_axis`
        (psa_gui.py:970-999); computed on the GPU from the result still resident there."""
        if not chiral or sed.sed is None or not sed.is_complex:
            return sed
        if sed.sed.ndim != 3 or sed.sed.shape[1] == 0:
            return sed
        c1, c2 = _CHIRAL_PAIR.get(chiral_axis, _CHIRAL_PAIR["z"])
        sed.phase = self.engine.result_chiral_phase(sed.sed.shape[0], sed.sed.shape[1], c1, c2)
        return sed

    def calculate_kpath_sed(self, direction, bz_coverage: float = 1.0, n_k: int = 100,
                            basis_atom_types=None, summation_mode: str = 'coherent',
                            basis_atom_indices=None, lat_param: Optional[float] = None,
                            chiral: bool = False, chiral_axis: str = 'z',
                            k_chunk_size: int = 500) -> SED:
        """k-path dispersion in one call (README.md:100-106; psa_gui.py:947-999)."""
        if chiral and summation_mode != 'coherent':
            logger.info("Chirality calculation selected, forcing coherent summation mode.")
            summation_mode = 'coherent'
        k_mags, k_vecs = self.get_k_path(direction, bz_coverage, n_k, lat_param=lat_param)
        # the phase is computed from the result still on the device: no other calculation on this
        # engine (another thread, another calculator attached to it) may come in between
        with self.engine.lock:
            sed = self.calculate(k_mags, k_vecs, basis_atom_indices=basis_atom_indices,
                                 basis_atom_types=basis_atom_types, summation_mode=summation_mode,
                                 k_chunk_size=k_chunk_size)
            return self._finish(sed, chiral, chiral_axis)

    def calculate_chiral_sed(self, direction, bz_coverage: float = 1.0, n_k: int = 100,
                             chiral_axis: str = 'z', **kwargs) -> SED:
        """README.md:117-122: a k-path SED with the chiral phase attached."""
        return self.calculate_kpath_sed(direction, bz_coverage, n_k, chiral=True,
                                        chiral_axis=chiral_axis, **kwargs)

    def calculate_kgrid_sed(self, plane: str = 'xy', k_ranges=(-1.0, 1.0, -1.0, 1.0),
                            n_kx: int = 20, n_ky: int = 20, k_fixed: float = 0.0,
                            basis_atom_types=None, summation_mode: str = 'coherent',
                            basis_atom_indices=None, chiral: bool = False, chiral_axis: str = 'z',
                            k_chunk_size: int = 500) -> SED:
        """2-D k-grid SED in one call (README.md:135-140; psa_gui.py:2135-2191).
        `k_ranges` = (first_min, first_max, second_min, second_max)."""
        if chiral and summation_mode != 'coherent':
            logger.info("Chirality calculation selected for K-Grid, forcing coherent summation mode.")
            summation_mode = 'coherent'
        k_mags, k_vecs, shape = self.get_k_grid(plane, (k_ranges[0], k_ranges[1]),
                                                (k_ranges[2], k_ranges[3]), n_kx, n_ky, k_fixed)
        with self.engine.lock:
            sed = self.calculate(k_mags, k_vecs, basis_atom_indices=basis_atom_indices,
                                 basis_atom_types=basis_atom_types, summation_mode=summation_mode,
                                 k_grid_shape=shape, k_chunk_size=k_chunk_size)
            return self._finish(sed, chiral, chiral_axis)

    # ------------------------------------------------------------------ iSED
    def _ised_groups(self, basis_atom_idx_ised, basis_atom_types_ised) -> List[np.ndarray]:
        """Atom groups of an iSED reconstruction (reference :389-433): indices win over types; a
        flat index list is one group, a flat type list is one group PER type."""
        n_atoms, kinds = self.traj.n_atoms, self.traj.types.astype(int)
        groups: List[np.ndarray] = []
        if basis_atom_idx_ised and len(basis_atom_idx_ised) > 0:
            nested = isinstance(basis_atom_idx_ised[0], list)
            for members in (basis_atom_idx_ised if nested else [basis_atom_idx_ised]):
                arr = np.asarray(members, dtype=int)
                if np.any(arr >= n_atoms) or np.any(arr < 0):
                    raise ValueError(f"Atom indices in group {members} out of bounds." if nested
                                     else "Atom indices out of bounds.")
                if arr.size:
                    groups.append(arr)
            if basis_atom_types_ised and len(basis_atom_types_ised) > 0:
                logger.warning("iSED: atom_indices and atom_types provided. Using atom_indices.")
        elif basis_atom_types_ised and len(basis_atom_types_ised) > 0:
            nested = isinstance(basis_atom_types_ised[0], list)
            for wanted in (basis_atom_types_ised if nested else [[t] for t in basis_atom_types_ised]):
                members = np.flatnonzero(np.isin(kinds, wanted))
                if members.size:
                    groups.append(members)
                else:
                    logger.warning("No atoms for type group %s in iSED.", wanted)
        else:
            groups.append(np.arange(n_atoms))
        return groups

    def _single_bin(self, k_vector: np.ndarray, members: np.ndarray, i_w: int, mean_pos_all: np.ndarray) -> np.ndarray:
        """S[i_w, k, :] (3 complex64) of one k-vector and one atom group: what `calculate(...).sed[i_w, i_k]`
        would hold (reference :58-84 for one k, one frequency)."""
        members = np.asarray(members)
        if np.any(members >= self.traj.n_atoms) or np.any(members < 0):
            raise ValueError("Atom indices in basis out of bounds.")
        if self._shard is not None and self._shard.nranks > 1:     # sharded engines hold partial data
            sed = self.calculate(np.zeros(1, np.float32), np.asarray(k_vector, np.float32)[None, :],
                                 basis_atom_indices=members)
            return sed.sed[i_w, 0, :]
        slot, data, flags = self._data_slot()
        eng = self.engine
        with eng.lock:
            eng.ensure_resident(slot, data)
            group = self._device_groups([members])
            return eng.single_bin(slot, mean_pos_all, k_vector, None if group is None else group[0], i_w, flags)

    def ised(self, k_dir_spec, k_target: float, w_target: float, char_len_k_path: float,
             nk_on_path: int = 100, bz_cov_ised: float = 1.0,
             basis_atom_idx_ised: Optional[List[int]] = None,
             basis_atom_types_ised: Optional[List[int]] = None,
             rescale_factor: Union[str, float] = 1.0, n_recon_frames: int = 100,
             dump_filepath: str = "iSED_reconstruction.dump",
             plot_dir_ised: Optional[Path] = None, plot_max_freq: Optional[float] = None,
             plot_theme: str = 'light') -> None:
        """Inverse SED (reference :373-588): reconstruct the real-space motion of the mode nearest
        to (k_target, w_target) along a k-path and write it as a LAMMPS dump.  Per atom group the
        complex SED comes from `calculate` -- on the GPU, against the trajectory already resident
        in HBM -- and the amplitude A of the selected (w, k) bin is replayed as
        Re[A exp(i tau - i k r.k_hat)] over one period.  Plotting the input spectrum (the
        reference's optional last step) is outside this package; `plot_dir_ised` is ignored."""
        mean_pos = self._mean_positions()              # np.mean(positions, axis=0, dtype=float32), cached per array
        kinds = self.traj.types.astype(int)
        n_atoms = self.traj.n_atoms
        k_hat = parse_direction(k_dir_spec)
        groups = self._ised_groups(basis_atom_idx_ised, basis_atom_types_ised)
        if not groups:
            logger.error("iSED: No atom groups for reconstruction. Aborting.")
            return
        k_mags, k_vecs = self.get_k_path(direction_spec=k_hat, bz_coverage=bz_cov_ised, n_k=nk_on_path,
                                         lat_param=char_len_k_path)
        i_k = int(np.argmin(np.abs(k_mags - k_target)))
        k_used = k_mags[i_k]
        tau = np.linspace(0, 2 * np.pi, n_recon_frames, endpoint=False)
        along_k = np.dot(mean_pos, k_hat)                 # r . k_hat per atom
        motion = np.zeros((n_recon_frames, n_atoms, 4), dtype=np.float32)   # x, y, z, type
        auto = isinstance(rescale_factor, str) and rescale_factor.lower() == "auto"
        peak, spread_sum, spread_atoms = 0.0, 0.0, 0

        # Of each group's path spectrum the reference consumes ONE bin, sed[i_w, i_k, :] (:483,
        # :494-499).  Only that bin is computed: one k-vector projected over the trajectory and one
        # DFT dot (psa_sed_single_bin) instead of nk_on_path projections and 3 nk_on_path FFTs.
        freqs = np.fft.fftfreq(self.traj.n_frames, d=self.dt_ps)
        i_w = int(np.argmin(np.abs(freqs - w_target)))
        for members in groups:
            amplitude = self._single_bin(k_vecs[i_k], members, i_w, mean_pos)
            carrier = np.exp(1j * tau[:, None] - 1j * k_used * along_k[members][None, :])
            for axis in range(3):
                motion[:, members, axis] += np.real(amplitude[axis] * carrier)
            if auto:
                peak = max(peak, float(np.amax(np.abs(motion[:, members, :3]))))
                thermal = self.traj.positions[:, members, :] - mean_pos[None, members, :]
                spread_sum += np.std(thermal) * len(members)
                spread_atoms += len(members)

        motion[0, :, 3] = kinds
        touched = np.unique(np.concatenate(groups))
        if auto:
            if peak > 1e-9:
                motion[:, touched, :3] /= peak
                typical = spread_sum / spread_atoms if spread_atoms > 0 else 0.0
                if typical > 1e-9:
                    motion[:, touched, :3] *= typical
            else:
                logger.warning("iSED: Max wiggle amp near zero. Auto-rescaling ineffective.")
        elif isinstance(rescale_factor, (int, float)):
            motion[:, touched, :3] *= rescale_factor
        out_to_qdump(dump_filepath, mean_pos[None, :, :] + motion[:, :, :3], motion[0, :, 3].astype(int),
                     self.traj.box_matrix)
        logger.info("iSED reconstruction saved: %s", dump_filepath)
        if plot_dir_ised:
            logger.warning("iSED: plotting the input spectrum is not part of psa_amd; plot_dir_ised ignored.")
