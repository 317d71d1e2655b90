"""
Synthetic diamond-silicon trajectories for benchmarks and parity tests.

The reference ships no trajectory data (its example .lammpstrj files are git-ignored,
SURVEY.md section 4), so BASELINE.json's configurations are realised synthetically:
diamond-cubic Si (a = 5.43 A, 8-atom basis, basis atoms 0-3 -> type 1, 4-7 -> type 2),
velocities = counter-hashed ~N(0,1) noise plus planted plane-wave modes so that the
dispersion has visible peaks.

The velocity field is defined so that the GPU (`psa_data_fill_synthetic`,
csrc/kernels_misc.hip) and NumPy (`velocities_block` below) produce the SAME BITS:
integer hash -> sum of four 16-bit uniforms -> one float32 multiply; modes from shared
host-built cos/sin tables with unfused float32 multiply/add.  A 25.8 GB configuration-3
trajectory can therefore be generated in place in HBM and still be checked frame-block by
frame-block against the CPU oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

A_SI = 5.43
_BASIS = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                   [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
_INV_SIGMA = np.float32(1.0) / np.float32(37837.0)     # 1/std of the 4x16-bit sum
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_KEYMUL = 0xD1B54A32D192ED03


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _GOLDEN
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


@dataclass
class Mode:
    """A planted plane wave  amp * cos(2 pi f t / T - k . r0)  on one Cartesian component."""
    amp: float
    freq_bin: int
    k_vec: Tuple[float, float, float]
    comp: int


@dataclass
class SyntheticSpec:
    cells: Tuple[int, int, int]
    n_frames: int
    dt_ps: float = 0.001
    seed: int = 0
    modes: List[Mode] = field(default_factory=list)

    @property
    def n_atoms(self) -> int:
        return 8 * self.cells[0] * self.cells[1] * self.cells[2]


def lattice(cells) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(r0 (N,3) float32, types (N,) int32, box_matrix (3,3) float32); atom order is
    cell-major, basis-minor."""
    cx, cy, cz = cells
    ii, jj, kk = np.meshgrid(np.arange(cx), np.arange(cy), np.arange(cz), indexing="ij")
    origin = np.stack([ii, jj, kk], axis=-1).reshape(-1, 1, 3).astype(np.float64)
    r0 = ((origin + _BASIS[None]) * A_SI).reshape(-1, 3).astype(np.float32)
    types = np.tile(np.array([1, 1, 1, 1, 2, 2, 2, 2], np.int32), cx * cy * cz)
    box = np.diag([cx * A_SI, cy * A_SI, cz * A_SI]).astype(np.float32)
    return r0, types, box


def mode_tables(spec: SyntheticSpec, r0: np.ndarray):
    """(amp, comp, ct, st, ca, sa) float32/int32 tables shared by both generators."""
    T, N, M = spec.n_frames, r0.shape[0], len(spec.modes)
    amp = np.array([m.amp for m in spec.modes], np.float32)
    comp = np.array([m.comp for m in spec.modes], np.int32)
    ct, st = np.zeros((M, T), np.float32), np.zeros((M, T), np.float32)
    ca, sa = np.zeros((M, N), np.float32), np.zeros((M, N), np.float32)
    t = np.arange(T, dtype=np.float64)
    for i, m in enumerate(spec.modes):
        wt = 2 * np.pi * m.freq_bin * t / T
        kr = r0.astype(np.float64) @ np.asarray(m.k_vec, np.float64)
        ct[i], st[i] = np.cos(wt), np.sin(wt)          # cos(wt - kr) = ct*ca + st*sa
        ca[i], sa[i] = np.cos(kr), np.sin(kr)
    return amp, comp, ct, st, ca, sa


def velocities_block(spec: SyntheticSpec, tables, t0: int, nt: int) -> np.ndarray:
    """Frames [t0, t0+nt) of the synthetic velocity array, (nt, N, 3) float32 -- the NumPy
    twin of fill_synthetic_kernel."""
    amp, comp, ct, st, ca, sa = tables
    N = spec.n_atoms
    key = np.uint64((spec.seed * _KEYMUL) & 0xFFFFFFFFFFFFFFFF)
    i = (np.arange(t0 * N * 3, (t0 + nt) * N * 3, dtype=np.uint64)).reshape(nt, N, 3)
    h = _splitmix64(i ^ key)
    m16 = np.uint64(0xFFFF)
    s = ((h & m16) + ((h >> np.uint64(16)) & m16) + ((h >> np.uint64(32)) & m16)
         + (h >> np.uint64(48))).astype(np.int64) - 131070
    v = s.astype(np.float32) * _INV_SIGMA
    for m in range(len(amp)):
        w = (ct[m, t0:t0 + nt, None] * ca[m][None, :]) + (st[m, t0:t0 + nt, None] * sa[m][None, :])
        v[:, :, comp[m]] = v[:, :, comp[m]] + amp[m] * w
    return v


def fill_device(engine, slot: int, spec: SyntheticSpec, tables, t_begin: int = 0, t_count: int = None) -> None:
    """Generate the same array directly in HBM -- or only its frames [t_begin, t_begin + t_count)
    (a frame-sharded rank holds just its slice)."""
    amp, comp, ct, st, ca, sa = tables
    if t_count is None:
        t_count = spec.n_frames - t_begin
    engine.alloc(slot, t_count, spec.n_atoms)
    engine.fill_synthetic(slot, spec.seed, amp, comp, ct[:, t_begin:t_begin + t_count], st[:, t_begin:t_begin + t_count],
                          ca, sa, t_offset=t_begin)


def reciprocal_step(cells) -> np.ndarray:
    """2 pi / (n_i a): the k-grid spacing commensurate with the supercell."""
    return 2 * np.pi / (np.asarray(cells, float) * A_SI)


# BASELINE.json configurations
def baseline_spec(name: str) -> Tuple[SyntheticSpec, dict]:
    """(spec, k-request) for C1..C5.  The k-request is what the config asks of
    `get_k_path` / `get_k_grid`."""
    kx = 2 * np.pi / A_SI
    if name == "C1":       # 512 atoms x 4096 steps x 32 k, [100]  (examples/Si_config.yaml shape)
        spec = SyntheticSpec((4, 4, 4), 4096, dt_ps=0.02)
        req = dict(kind="path", direction="100", bz_coverage=4.0, n_k=32)
    elif name == "C2":     # 8192 atoms x 16384 steps x 128 k, [100], coherent
        spec = SyntheticSpec((16, 8, 8), 16384)
        req = dict(kind="path", direction=[1, 0, 0], bz_coverage=1.0, n_k=128)
    elif name == "C3":     # 32768 atoms x 65536 steps x 256 k, [110], 2 basis types
        spec = SyntheticSpec((16, 16, 16), 65536)
        req = dict(kind="path", direction=[1, 1, 0], bz_coverage=1.0, n_k=256,
                   basis_atom_types=[1, 2])
    elif name == "C4":     # 50x50 k-grid x 16384 steps x 8192 atoms
        spec = SyntheticSpec((16, 8, 8), 16384)
        req = dict(kind="grid", plane="xy", k_ranges=(-3.5, 3.5, -3.5, 3.5), n_kx=50, n_ky=50)
    elif name == "C5":     # chiral: 16384 atoms x 32768 steps x 128 k, complex output
        spec = SyntheticSpec((16, 16, 8), 32768)
        req = dict(kind="path", direction=[1, 0, 0], bz_coverage=1.0, n_k=128, chiral=True)
    else:
        raise ValueError(f"unknown configuration {name}")
    T = spec.n_frames
    spec.modes = [Mode(3.0, T // 16, (0.25 * kx, 0.0, 0.0), 0),
                  Mode(2.0, T // 8, (0.5 * kx, 0.5 * kx, 0.0), 2),
                  Mode(1.5, T // 5, (0.125 * kx, 0.0, 0.0), 1)]
    return spec, req
