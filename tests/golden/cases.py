"""
Golden-case table shared by `make_golden.py` (runs the real reference in the build
container) and by the test-suite (replays the same inputs through the oracle and
through the HIP path).  Pure data + a deterministic input builder -- no reference
code lives here.
"""
from __future__ import annotations

import numpy as np

A_SI = 5.43
_BASIS = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                   [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])

# name -> how to build a small diamond-cubic trajectory
TRAJ = {
    "a": dict(cells=(2, 2, 2), T=128, seed=11, jitter=0.05, tri=False, dt=0.002, third_type=5),
    "b": dict(cells=(16, 1, 1), T=64, seed=12, jitter=0.05, tri=True, dt=0.001, third_type=0),
    "c": dict(cells=(1, 1, 3), T=100, seed=13, jitter=0.03, tri=False, dt=0.004, third_type=0),
}


def build_traj(name: str) -> dict:
    """Deterministic float32 trajectory arrays (positions, velocities, ...)."""
    spec = TRAJ[name]
    cx, cy, cz = spec["cells"]
    rng = np.random.default_rng(spec["seed"])
    cells = np.array([[i, j, k] for i in range(cx) for j in range(cy) for k in range(cz)], float)
    frac = (cells[:, None, :] + _BASIS[None, :, :]).reshape(-1, 3)
    box = np.diag([cx * A_SI, cy * A_SI, cz * A_SI]).astype(np.float64)
    if spec["tri"]:
        box[1, 0] = 0.3 * A_SI
        box[2, 0] = 0.1 * A_SI
        box[2, 1] = -0.2 * A_SI
    r0 = (frac / np.array([cx, cy, cz])) @ box           # rows of box are lattice vectors
    n = r0.shape[0]
    T = spec["T"]
    pos = (r0[None] + spec["jitter"] * rng.standard_normal((T, n, 3))).astype(np.float32)
    vel = rng.standard_normal((T, n, 3)).astype(np.float32)
    # one planted plane-wave mode so the spectrum has structure
    t = np.arange(T)[:, None]
    kx = 2 * np.pi / A_SI * 0.5
    vel[:, :, 0] += (1.5 * np.cos(2 * np.pi * 5 * t / T - kx * r0[None, :, 0])).astype(np.float32)
    types = np.tile(np.array([1, 1, 1, 1, 2, 2, 2, 2], np.int32), n // 8)
    if spec["third_type"]:
        types[-spec["third_type"]:] = 3
    return dict(
        positions=pos, velocities=vel, types=types,
        timesteps=np.arange(T, dtype=np.float32),
        box_matrix=box.astype(np.float32),
        box_lengths=np.array([box[0, 0], box[1, 1], box[2, 2]], np.float32),
        box_tilts=np.array([box[1, 0], box[2, 0], box[2, 1]], np.float32),
        dt_ps=spec["dt"], cells=spec["cells"],
    )


def nd(x):
    """marker: pass this basis as a numpy array rather than a list."""
    return ("ndarray", list(x))


def realise_kw(kw: dict) -> dict:
    out = {}
    for k, v in kw.items():
        if isinstance(v, tuple) and len(v) == 2 and v[0] == "ndarray":
            v = np.array(v[1])
        out[k] = v
    return out


_KP = ("path", "100", 1.0, 8, None)

# `calculate` cases: name, traj, k spec, ctor opts, calculate kwargs
CALC_CASES = [
    dict(name="coh_all", traj="a", k=_KP),
    dict(name="coh_types12", traj="a", k=_KP, kw=dict(basis_atom_types=[1, 2])),
    dict(name="inc_types12", traj="a", k=_KP,
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="inc_nested_types", traj="a", k=_KP,
         kw=dict(basis_atom_types=[[1, 2], [3]], summation_mode="incoherent")),
    dict(name="inc_single_type", traj="a", k=_KP,
         kw=dict(basis_atom_types=[1], summation_mode="incoherent")),
    dict(name="inc_all_atoms", traj="a", k=_KP, kw=dict(summation_mode="incoherent")),
    dict(name="types_fallback_all", traj="a", k=_KP, kw=dict(basis_atom_types=[9])),
    dict(name="types_partial_missing", traj="a", k=_KP,
         kw=dict(basis_atom_types=[2, 9], summation_mode="incoherent")),
    dict(name="type_scalar_int", traj="a", k=_KP, kw=dict(basis_atom_types=2)),
    dict(name="idx_flat", traj="a", k=_KP, kw=dict(basis_atom_indices=[0, 1, 2, 5, 9])),
    dict(name="idx_nested_coh", traj="a", k=_KP,
         kw=dict(basis_atom_indices=[[0, 1, 2], [2, 3]])),
    dict(name="idx_nested_inc", traj="a", k=_KP,
         kw=dict(basis_atom_indices=[[0, 1, 2], [2, 3]], summation_mode="incoherent")),
    dict(name="idx_ndarray_dup_unsorted", traj="a", k=_KP,
         kw=dict(basis_atom_indices=nd([7, 3, 3, 60]))),
    dict(name="idx_and_types", traj="a", k=_KP,
         kw=dict(basis_atom_indices=[0], basis_atom_types=[2])),
    dict(name="idx_empty_list", traj="a", k=_KP, kw=dict(basis_atom_indices=[])),
    dict(name="displacements", traj="a", k=_KP, ctor=dict(use_displacements=True)),
    dict(name="displacements_types_inc", traj="a", k=_KP, ctor=dict(use_displacements=True),
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="chunked3", traj="a", k=("path", [1, 1, 0], 2.0, 8, None), kw=dict(k_chunk_size=3)),
    dict(name="large_phase", traj="b", k=("path", [1, 1, 0], 4.0, 16, None)),
    dict(name="large_phase_latparam", traj="b", k=("path", "x", 4.0, 6, 2.491)),
    dict(name="nonpow2_T100", traj="c", k=("path", "z", 2.0, 5, None)),
    dict(name="single_k", traj="c", k=("path", "x", 1.0, 1, None)),
    dict(name="grid_xy", traj="a", k=("grid", "xy", (-1.5, 1.5), (-1.0, 1.0), 3, 4, 0.25)),
    dict(name="grid_zx_inc", traj="a", k=("grid", "zx", (-0.5, 1.5), (0.0, 1.0), 2, 3, -0.3),
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
]

# `calculate` cases with more than 16 k-vectors: these are the ones the product-default "2 x f16"
# projection kernel serves (api_project.hip make_geom: K > 16), in all its forms -- 64-row blocks (K <= 32)
# and 128-row blocks, row DMA (all atoms in order) and gather DMA (index lists / type groups),
# the materialised-displacement array, phases up to ~200 rad.  The reference output is stored as
# the full intensity plus every WIDE_SED_STRIDE-th frequency row of `sed` (keeps the fixture small).
WIDE_SED_STRIDE = 4
_K24 = ("path", "100", 1.0, 24, None)
_K40 = ("path", [1, 1, 0], 2.0, 40, None)
_K140 = ("path", "100", 3.0, 140, None)
CALC_WIDE_CASES = [
    dict(name="w_coh_all_k24", traj="a", k=_K24),
    dict(name="w_coh_all_k40", traj="a", k=_K40),
    dict(name="w_coh_all_k140", traj="a", k=_K140),
    dict(name="w_idx_dup_k40", traj="a", k=_K40,
         kw=dict(basis_atom_indices=nd([7, 3, 3, 60, 12, 12, 41, 0, 63, 5, 18]))),
    dict(name="w_idx_list_k140", traj="a", k=_K140,
         kw=dict(basis_atom_indices=list(range(1, 64, 2)) + [2, 2, 50])),
    dict(name="w_inc_types12_k24", traj="a", k=_K24,
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="w_inc_types_nested_k140", traj="a", k=_K140,
         kw=dict(basis_atom_types=[[1], [2, 3]], summation_mode="incoherent")),
    dict(name="w_coh_types12_k40", traj="a", k=_K40, kw=dict(basis_atom_types=[1, 2])),
    dict(name="w_displacements_k40", traj="a", k=_K40, ctor=dict(use_displacements=True)),
    dict(name="w_displacements_inc_k24", traj="a", k=_K24, ctor=dict(use_displacements=True),
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="w_large_phase_k40", traj="b", k=("path", [1, 1, 0], 4.0, 40, None)),
    dict(name="w_large_phase_k140", traj="b", k=("path", "x", 4.0, 140, 2.491)),
    dict(name="w_large_phase_inc_k40", traj="b", k=("path", [1, 1, 0], 4.0, 40, None),
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="w_nonpow2_T100_k24", traj="c", k=("path", "z", 2.0, 24, None)),
    dict(name="w_grid_xy_6x7", traj="a", k=("grid", "xy", (-1.5, 1.5), (-1.0, 1.0), 6, 7, 0.25)),
    dict(name="w_grid_zx_inc_5x8", traj="a", k=("grid", "zx", (-0.5, 1.5), (-1.0, 1.0), 5, 8, -0.3),
         kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="w_idx_nested_inc_k40", traj="a", k=_K40,
         kw=dict(basis_atom_indices=[list(range(0, 40)), list(range(30, 64)) + [1, 1]], summation_mode="incoherent")),
    dict(name="w_displacements_idx_k40", traj="b", k=("path", [1, 1, 0], 4.0, 40, None), ctor=dict(use_displacements=True),
         kw=dict(basis_atom_indices=nd(list(range(0, 128, 3)) + [5, 5, 127]))),
]


# `calculate` cases whose k-lists fill an EVEN number of 128-row blocks (100 k-vectors = 200 rows -> one 256-row block,
# partly filled; 250 = 500 rows -> two): the territory of the 256-row form of the planes kernel (k1_planes_wide.hip,
# round 3) -- whole trajectory, index list with duplicates, incoherent type groups, displacement mode and phases of
# ~200 rad on the triclinic box, a frame count that is no multiple of 64.  (140 k-vectors = three blocks stay on the
# 128-row form: the cases above.)  Stored like the wide cases.
_K100 = ("path", [1, 1, 0], 2.0, 100, None)
_K250 = ("path", "100", 3.0, 250, None)
CALC_W256_CASES = [
    dict(name="v_coh_all_k100", traj="a", k=_K100),
    dict(name="v_coh_all_k250", traj="a", k=_K250),
    dict(name="v_idx_dup_k100", traj="a", k=_K100,
         kw=dict(basis_atom_indices=nd([7, 3, 3, 60, 12, 12, 41, 0, 63, 5, 18] + list(range(20, 50))))),
    dict(name="v_inc_types_nested_k250", traj="a", k=_K250,
         kw=dict(basis_atom_types=[[1], [2, 3]], summation_mode="incoherent")),
    dict(name="v_large_phase_k100", traj="b", k=("path", [1, 1, 0], 4.0, 100, None)),
    dict(name="v_displacements_idx_k250", traj="b", k=("path", "x", 4.0, 250, 2.491), ctor=dict(use_displacements=True),
         kw=dict(basis_atom_indices=nd(list(range(0, 128, 3)) + [5, 5, 127]))),
    dict(name="v_nonpow2_T100_k100", traj="c", k=("path", "z", 2.0, 100, None)),
]


# `calculate` cases whose k-lists hold pairs (k, -k) and repeated vectors: the library projects one
# vector of each pair and writes the partner's columns from it (PSA_OPT_FOLD_PAIRS: S(-k)[w] =
# conj S(k)[(T-w) mod T]); the REFERENCE computed every vector on its own.  Grids symmetric about
# Gamma (examples/k_grid_heatmap_example.py:33-38) with an even and an odd point count (the odd one
# holds Gamma itself), a partly symmetric grid, a non-power-of-two T, displacement mode, an
# incoherent run, a list long enough for the block-by-block result path (400 k-vectors), and a
# hand-made list: a path, its mirror image in reverse order and two repeated vectors.  A grid that
# must NOT fold (k_fixed != 0) is `w_grid_xy_6x7` above.  Stored like the wide cases.
_G66 = ("grid", "xy", (-1.5, 1.5), (-1.0, 1.0), 6, 6, 0.0)
CALC_SYM_CASES = [
    dict(name="s_grid_xy_6x6_coh", traj="a", k=_G66),
    dict(name="s_grid_xy_6x6_inc", traj="a", k=_G66, kw=dict(basis_atom_types=[1, 2], summation_mode="incoherent")),
    dict(name="s_grid_xy_5x5_gamma", traj="a", k=("grid", "xy", (-1.0, 1.0), (-2.0, 2.0), 5, 5, 0.0)),
    dict(name="s_grid_xy_partial", traj="a", k=("grid", "xy", (-1.5, 1.0), (-1.0, 1.0), 6, 6, 0.0)),
    dict(name="s_grid_yz_T100", traj="c", k=("grid", "yz", (-2.0, 2.0), (-1.0, 1.0), 4, 6, 0.0)),
    dict(name="s_grid_zx_disp", traj="b", k=("grid", "zx", (-1.0, 1.0), (-3.0, 3.0), 4, 8, 0.0),
         ctor=dict(use_displacements=True)),
    dict(name="s_grid_xy_idx_inc", traj="a", k=_G66,
         kw=dict(basis_atom_indices=[list(range(0, 40)), list(range(30, 64)) + [1, 1]], summation_mode="incoherent")),
    dict(name="s_grid_xy_20x20", traj="a", k=("grid", "xy", (-3.5, 3.5), (-3.5, 3.5), 20, 20, 0.0)),
    dict(name="s_mirrored_path", traj="a", k=("mirrored_path", [1, 1, 0], 2.0, 21, None)),
]


def c1_inputs():
    """BASELINE configuration 1 (512 atoms x 4096 steps x 32 k-points, [100], bz 4.0, dt 0.02:
    examples/Si_config.yaml's shape) as arrays: the synthetic velocities of psa_amd/synth.py
    (bit-identical NumPy twin of the device generator) and jittered positions from a seeded
    NumPy generator.  Built the same way by make_golden.py (which feeds them to the real
    reference) and by the GPU test."""
    from psa_amd import synth
    spec, req = synth.baseline_spec("C1")
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    vel = np.concatenate([synth.velocities_block(spec, tables, t, 256) for t in range(0, spec.n_frames, 256)])
    rng = np.random.default_rng(101)
    pos = (r0[None] + 0.05 * rng.standard_normal(vel.shape, dtype=np.float32)).astype(np.float32)
    return spec, req, dict(positions=pos, velocities=vel, types=types, box_matrix=box,
                           timesteps=np.arange(spec.n_frames, dtype=np.float32),
                           box_lengths=np.diag(box).copy(), box_tilts=np.zeros(3, np.float32))


KPATH_CASES = [
    dict(traj="a", spec="100", cov=1.0, n_k=32, lat=None),
    dict(traj="a", spec=[1, 1, 0], cov=4.0, n_k=250, lat=None),
    dict(traj="a", spec="x", cov=1.0, n_k=1, lat=None),
    dict(traj="a", spec=[1, 1, 0], cov=4.0, n_k=17, lat=2.491),
    dict(traj="b", spec="100", cov=1.0, n_k=32, lat=None),
    dict(traj="b", spec=[1, 1, 0], cov=4.0, n_k=25, lat=None),
    dict(traj="b", spec={"h": 1, "k": 1, "l": 1}, cov=2.0, n_k=9, lat=None),
    dict(traj="b", spec=30.0, cov=0.5, n_k=7, lat=0.0),
]

KGRID_CASES = [
    dict(plane="xy", rx=(-3.5, 3.5), ry=(-3.5, 3.5), nx=5, ny=5, fixed=0.0),
    dict(plane="XY", rx=(-1.0, 2.0), ry=(0.0, 1.0), nx=3, ny=4, fixed=0.5),
    dict(plane="yz", rx=(-1.0, 2.0), ry=(0.0, 1.0), nx=4, ny=2, fixed=-0.25),
    dict(plane="zx", rx=(-1.0, 2.0), ry=(0.0, 1.0), nx=2, ny=3, fixed=1.25),
    dict(plane="xy", rx=(0.0, 0.0), ry=(1.0, 1.0), nx=1, ny=1, fixed=0.0),
]

DIRECTION_CASES = ["x", "y", "z", "xy", "yx", "xz", "zx", "yz", "zy", "xyz", "100", "010", "001",
                   "110", "111", "0,1,0", " 1 0 0 ", "180.0", "-30", 0, 90, 45, 33.3,
                   [1, 0, 0], (0, 5, 0), [1, 1, 1], [45], [2, -1, 0.5],
                   {"angle": 30}, {"h": 1, "k": 1, "l": 0}, {"h": 0, "k": 0, "l": 2}, {"k": 3}]


def k_from_spec(calc, spec):
    """Build (k_mags, k_vecs, grid_shape) by calling the calculator's own generators."""
    if spec[0] == "path":
        _, d, cov, n_k, lat = spec
        mags, vecs = calc.get_k_path(d, cov, n_k, lat_param=lat)
        return mags, vecs, None
    if spec[0] == "mirrored_path":           # the path, its negation in reverse order, two vectors once more
        _, d, cov, n_k, lat = spec
        mags, vecs = calc.get_k_path(d, cov, n_k, lat_param=lat)
        vecs = np.concatenate([-vecs[::-1], vecs[1:], vecs[3:5]]).astype(np.float32)
        mags = np.concatenate([-mags[::-1], mags[1:], mags[3:5]]).astype(np.float32)
        return mags, vecs, None
    _, plane, rx, ry, nkx, nky, fixed = spec
    return calc.get_k_grid(plane, rx, ry, nkx, nky, fixed)


# iSED reconstructions: (name, trajectory, keyword arguments of SEDCalculator.ised)
ISED_CASES = [
    ("all_atoms", "a", dict(k_dir_spec="x", k_target=0.58, w_target=19.5, char_len_k_path=A_SI,
                            nk_on_path=8, bz_cov_ised=1.0, n_recon_frames=6)),
    ("per_type_auto", "a", dict(k_dir_spec=[1, 1, 0], k_target=0.4, w_target=30.0, char_len_k_path=A_SI,
                                nk_on_path=5, bz_cov_ised=2.0, basis_atom_types_ised=[1, 2, 3],
                                rescale_factor="auto", n_recon_frames=4)),
    ("index_groups_triclinic", "b", dict(k_dir_spec="x", k_target=0.2, w_target=70.0, char_len_k_path=2.0,
                                         nk_on_path=6, basis_atom_idx_ised=[[0, 1, 2, 40], [5, 6]],
                                         rescale_factor=2.5, n_recon_frames=3)),
]


def qdump_tilted_inputs():
    """(positions (2,5,3), types, box matrix with upper-triangle tilts) for the dump writer alone."""
    rng = np.random.default_rng(5)
    pos = (10 * rng.standard_normal((2, 5, 3))).astype(np.float32)
    box = np.array([[12.0, 1.5, -0.75], [0.0, 9.0, -2.25], [0.0, 0.0, 7.5]], np.float32)
    return pos, np.array([1, 2, 2, 3, 1]), box
