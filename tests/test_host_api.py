"""CPU tests of the product's host side (no GPU): the mirrored reference classes, the k
generators and constructor against the golden vectors, group resolution against the
oracle, and `calculate`'s orchestration with the oracle-backed test double."""
import os

import numpy as np
import pytest

import cases as C
from conftest import make_calculator, make_trajectory, rel_max
from oracle import psa_oracle as O
from oracle_engine import OracleEngine
from psa_amd import SED, SEDCalculator, Trajectory, parse_direction


# ---------------------------------------------------------------- parse_direction
# (re-typed from the reference's tests/test_helpers.py:6-100: same inputs, same expectations)
@pytest.mark.parametrize("spec, want", [
    ("x", [1, 0, 0]), ("y", [0, 1, 0]), ("z", [0, 0, 1]), ("100", [1, 0, 0]),
    ("xy", [2 ** -.5, 2 ** -.5, 0]), ("110", [2 ** -.5, 2 ** -.5, 0]),
    ("xyz", [3 ** -.5] * 3), ("111", [3 ** -.5] * 3), ("0,1,0", [0, 1, 0]), (" 1 0 0 ", [1, 0, 0]),
    (0, [1, 0, 0]), (90, [0, 1, 0]), (45, [np.cos(np.pi / 4), np.sin(np.pi / 4), 0]),
    ("180.0", [-1, 0, 0]), ([1, 0, 0], [1, 0, 0]), ((0, 5, 0), [0, 1, 0]),
    (np.array([1, 1, 1]), [3 ** -.5] * 3), ([45], [np.cos(np.pi / 4), np.sin(np.pi / 4), 0]),
    (np.array(60.0), [0.5, np.sqrt(3) / 2, 0]), ({"angle": 30}, [np.sqrt(3) / 2, 0.5, 0]),
    ({"h": 1, "k": 0, "l": 0}, [1, 0, 0]), ({"h": 1, "k": 1, "l": 0}, [2 ** -.5, 2 ** -.5, 0]),
    ({"h": 0, "k": 0, "l": 2}, [0, 0, 1]),
])
def test_parse_direction_values(spec, want):
    np.testing.assert_allclose(parse_direction(spec), np.array(want, np.float32), atol=1e-6)


@pytest.mark.parametrize("bad", ["invalid_string", [1, 2], [1, 2, 3, 4],
                                 np.array([[1, 0, 0], [0, 1, 0]]), {"a": 1, "b": 2}, [0, 0, 0],
                                 np.array([1e-8, 1e-9, 1e-10], np.float32)])
def test_parse_direction_rejects(bad):
    with pytest.raises(ValueError):
        parse_direction(bad)


def test_parse_direction_type_error_and_small_norm():
    with pytest.raises(TypeError, match="Unsupported direction type: <class 'NoneType'>"):
        parse_direction(None)
    v = np.array([1e-7, 0, 0], np.float32)
    np.testing.assert_allclose(parse_direction(v), [1, 0, 0], atol=1e-6)


@pytest.mark.parametrize("i", range(len(C.DIRECTION_CASES)))
def test_parse_direction_golden(i, golden):
    got = parse_direction(C.DIRECTION_CASES[i])
    assert got.dtype == golden[f"dir{i}"].dtype
    np.testing.assert_allclose(got, golden[f"dir{i}"], rtol=3e-7, atol=1e-9)


# ---------------------------------------------------------------- Trajectory
# (re-typed from the reference's tests/test_trajectory.py)
def _traj_kwargs():
    rng = np.random.default_rng(0)
    return dict(positions=rng.random((2, 3, 3)).astype(np.float32),
                velocities=rng.random((2, 3, 3)).astype(np.float32),
                types=np.ones(3, np.int32), timesteps=np.arange(2, dtype=np.float32),
                box_matrix=np.eye(3, dtype=np.float32) * 10, box_lengths=np.full(3, 10, np.float32),
                box_tilts=np.zeros(3, np.float32), dt_ps=1.0)


def test_trajectory_ok():
    tr = Trajectory(**_traj_kwargs())
    assert (tr.n_frames, tr.n_atoms) == (2, 3)
    tr2 = Trajectory(*_traj_kwargs().values())          # positional order is part of the contract
    assert tr2.dt_ps == 1.0


@pytest.mark.parametrize("field, value, message", [
    ("positions", np.zeros((2, 3, 2)), "Positions must be 3D"),
    ("velocities", np.zeros((2, 3)), "Velocities must be 3D"),
    ("types", np.zeros((2, 3)), "Types must be 1D"),
    ("timesteps", np.zeros((2, 3)), "Timesteps must be 1D"),
    ("positions", np.zeros((3, 3, 3)), "Frame count mismatch"),
    ("types", np.ones(4), "Atom count mismatch"),
    ("box_matrix", np.eye(2), "Box matrix must be 3x3"),
    ("box_lengths", np.array([10, 10]), "Box lengths must be a 3-element array"),
    ("box_tilts", np.array([0, 0]), "Box tilts must be a 3-element array"),
])
def test_trajectory_validation(field, value, message):
    kw = _traj_kwargs()
    kw[field] = value
    with pytest.raises(ValueError, match=message):
        Trajectory(**kw)


# ---------------------------------------------------------------- SED
# (re-typed from the reference's tests/test_sed.py)
def _sed_kwargs():
    rng = np.random.default_rng(1)
    sed = (rng.random((10, 5, 3)) + 1j * rng.random((10, 5, 3))).astype(np.complex64)
    return dict(sed=sed, freqs=np.linspace(0, 10, 10, dtype=np.float32),
                k_points=np.linspace(0, 1, 5, dtype=np.float32),
                k_vectors=rng.random((5, 3)).astype(np.float32),
                phase=rng.random((10, 5)).astype(np.float32))


def test_sed_intensity_and_empty():
    kw = _sed_kwargs()
    s = SED(**kw)
    np.testing.assert_allclose(s.intensity, np.sum(np.abs(kw["sed"]) ** 2, axis=-1).astype(np.float32), atol=1e-6)
    assert s.intensity.dtype == np.float32 and s.is_complex and s.k_grid_shape is None
    empty = SED(sed=np.array([]).reshape(0, 0, 3), freqs=np.array([]), k_points=np.array([]),
                k_vectors=np.array([]).reshape(0, 3))
    assert empty.intensity.shape == (0, 0)
    flat = SED(sed=np.ones((4, 6), np.float32), freqs=np.zeros(4), k_points=np.zeros(6), k_vectors=np.zeros((6, 3)),
               is_complex=False)
    assert flat.intensity.shape == (4,)                   # reference quirk: sums over k (sed.py:24)


def test_sed_save_load_round_trip(tmp_path):
    kw = _sed_kwargs()
    base = tmp_path / "sub" / "run1"
    SED(k_grid_shape=(5, 1), **kw).save(base)
    for suffix in (".sed.npy", ".freqs.npy", ".k_points.npy", ".k_vectors.npy", ".phase.npy", ".k_grid_shape.npy"):
        assert base.with_suffix(suffix).exists()
    back = SED.load(base)
    for f in ("sed", "freqs", "k_points", "k_vectors", "phase"):
        np.testing.assert_array_equal(getattr(back, f), kw[f])
    assert back.k_grid_shape == (5, 1)
    kw["phase"] = None
    SED(**kw).save(tmp_path / "nophase")
    assert not (tmp_path / "nophase.phase.npy").exists()
    assert SED.load(tmp_path / "nophase").phase is None
    np.save((tmp_path / "partial").with_suffix(".sed.npy"), np.array([1]))
    with pytest.raises(FileNotFoundError):
        SED.load(tmp_path / "partial")


# ---------------------------------------------------------------- constructor, k generators
def test_sed_files_written_by_the_reference_load(tmp_path):
    """tests/golden/sed_saved/ holds `SED.save` output of the REFERENCE (make_golden.py): a k-grid
    result with phase and grid shape, and an incoherent k-path result.  `psa_amd.SED.load` reads
    them, and what `psa_amd.SED.save` writes back is byte-for-byte the reference's files."""
    from conftest import GOLDEN
    from psa_amd import SED
    src = GOLDEN / "sed_saved"
    grid = SED.load(src / "grid_xy_phase")
    assert grid.sed.shape == (128, 12, 3) and grid.sed.dtype == np.complex64
    assert grid.k_grid_shape == (3, 4) and grid.phase.shape == (128, 12) and grid.phase.dtype == np.float32
    assert grid.k_points.size == 0 and grid.k_vectors.shape == (12, 3) and grid.freqs.dtype == np.float64
    path = SED.load(src / "path_inc")
    assert path.sed.shape == (128, 8) and path.sed.dtype == np.float32
    assert path.k_grid_shape is None and path.phase is None
    for name, obj in (("grid_xy_phase", grid), ("path_inc", path)):
        obj.save(tmp_path / name)
        ours = sorted(f.name for f in tmp_path.glob(f"{name}.*"))
        theirs = sorted(f.name for f in src.glob(f"{name}.*"))
        assert ours == theirs
        for f in theirs:
            assert (tmp_path / f).read_bytes() == (src / f).read_bytes(), f


def test_constructor_attributes(golden, trajs):
    for t, d in trajs.items():
        calc = make_calculator(d)
        for nm in ("a1", "a2", "a3", "b1", "b2", "b3", "recip_vecs_prim"):
            np.testing.assert_allclose(getattr(calc, nm), golden[f"ctor_{t}/{nm}"], rtol=1e-6, atol=1e-9)
        assert calc.recip_vecs_prim.dtype == np.float32 and calc.dt_ps == float(golden[f"ctor_{t}/dt_ps"])


def test_constructor_errors(trajs):
    tr = make_trajectory(trajs["c"])
    with pytest.raises(ValueError, match="must be positive"):
        SEDCalculator(tr, 0, 1, 1)
    with pytest.raises(ValueError, match="dt_ps must be positive"):
        SEDCalculator(tr, 1, 1, 1, dt_ps=-1.0)
    assert SEDCalculator(tr, 1, 1, 1, dt_ps=0.5).dt_ps == 0.5        # explicit dt overrides
    flat = make_trajectory(trajs["c"])
    flat.box_matrix = np.array([[1, 0, 0], [2, 0, 0], [0, 0, 1]], np.float32)
    with pytest.raises(ValueError, match="coplanar|near zero"):
        SEDCalculator(flat, 1, 1, 1)


@pytest.mark.parametrize("i", range(len(C.KPATH_CASES)))
def test_get_k_path_golden(i, golden, trajs):
    kc = C.KPATH_CASES[i]
    mags, vecs = make_calculator(trajs[kc["traj"]]).get_k_path(kc["spec"], kc["cov"], kc["n_k"], lat_param=kc["lat"])
    assert mags.dtype == np.float32 and vecs.dtype == np.float32
    np.testing.assert_allclose(mags, golden[f"kpath{i}/mags"], rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(vecs, golden[f"kpath{i}/vecs"], rtol=3e-7, atol=1e-9)


@pytest.mark.parametrize("i", range(len(C.KGRID_CASES)))
def test_get_k_grid_golden(i, golden, trajs):
    g = C.KGRID_CASES[i]
    mags, vecs, shape = make_calculator(trajs["a"]).get_k_grid(g["plane"], g["rx"], g["ry"], g["nx"], g["ny"], g["fixed"])
    assert mags.size == 0 and mags.dtype == np.float32 and shape == tuple(golden[f"kgrid{i}/shape"])
    assert vecs.dtype == np.float32
    np.testing.assert_allclose(vecs, golden[f"kgrid{i}/vecs"], rtol=3e-7, atol=1e-9)


def test_k_generator_errors(trajs):
    calc = make_calculator(trajs["a"])
    with pytest.raises(ValueError, match="n_k"):
        calc.get_k_path("x", 1.0, 0)
    with pytest.raises(ValueError, match="must be positive"):
        calc.get_k_grid("xy", (0, 1), (0, 1), 0, 3)
    with pytest.raises(ValueError, match="Invalid plane"):
        calc.get_k_grid("xx", (0, 1), (0, 1), 2, 2)


# ---------------------------------------------------------------- group resolution
@pytest.mark.parametrize("kw", [
    {}, dict(basis_atom_types=[1, 2]), dict(basis_atom_types=[1, 2], summation_mode="incoherent"),
    dict(basis_atom_types=[[1, 2], [3]]), dict(basis_atom_types=[9]), dict(basis_atom_types=2),
    dict(basis_atom_types=[]), dict(basis_atom_types=[2, 9], summation_mode="incoherent"),
    dict(basis_atom_indices=[0, 1, 5]), dict(basis_atom_indices=[[0, 1], [], [1, 2]]),
    dict(basis_atom_indices=np.array([7, 3, 3])), dict(basis_atom_indices=np.zeros((2, 2), int)),
    dict(basis_atom_indices=[], summation_mode="incoherent"),
    dict(basis_atom_indices=[0], basis_atom_types=[2]),
])
def test_group_resolution_matches_oracle(kw, trajs):
    d = trajs["a"]
    calc = make_calculator(d)
    mode = kw.get("summation_mode", "coherent")
    got = calc._resolve_groups(kw.get("basis_atom_indices"), kw.get("basis_atom_types"), mode)
    want = O.resolve_groups(d["types"], len(d["types"]), kw.get("basis_atom_indices"),
                            kw.get("basis_atom_types"), mode)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("kw, message", [
    (dict(basis_atom_types=[1, [2]]), "basis_atom_types must be a list of ints"),
    (dict(basis_atom_indices=[1, [2]]), "basis_atom_indices must be a list of ints"),
    (dict(basis_atom_indices=[0, 64]), "out of bounds"), (dict(basis_atom_indices=[-1]), "out of bounds"),
])
def test_group_resolution_errors(kw, message, trajs):
    with pytest.raises(ValueError, match=message):
        make_calculator(trajs["a"])._resolve_groups(kw.get("basis_atom_indices"), kw.get("basis_atom_types"), "coherent")


# ---------------------------------------------------------------- calculate() orchestration
@pytest.mark.parametrize("case", C.CALC_CASES, ids=[c["name"] for c in C.CALC_CASES])
def test_calculate_orchestration(case, golden, trajs):
    """Host logic end to end with the oracle standing in for the GPU: what `calculate`
    asks of the engine must reproduce the reference's output for every golden case."""
    d = trajs[case["traj"]]
    name = case["name"]
    eng = OracleEngine()
    calc = make_calculator(d, **case.get("ctor", {})).attach(engine=eng)
    kw = C.realise_kw(case.get("kw", {}))
    shape = tuple(golden[f"{name}/grid_shape"]) or None
    sed = calc.calculate(golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"], k_grid_shape=shape, **kw)
    assert sed.sed.dtype == golden[f"{name}/sed"].dtype
    assert rel_max(sed.sed, golden[f"{name}/sed"]) <= 2e-6
    assert sed.is_complex == bool(golden[f"{name}/is_complex"]) and sed.phase is None
    np.testing.assert_array_equal(sed.freqs, golden[f"{name}/freqs"])
    assert sed.k_grid_shape == shape
    # displacement mode projects positions (slot 1), velocity mode slot 0
    assert eng.calls[0]["slot"] == (1 if case.get("ctor", {}).get("use_displacements") else 0)


def test_calculate_fast_path_and_edge_cases(trajs):
    d = trajs["a"]
    eng = OracleEngine()
    calc = make_calculator(d).attach(engine=eng)
    mags, vecs = calc.get_k_path("x", 1.0, 3)
    calc.calculate(mags, vecs, basis_atom_types=[1, 2, 3])
    assert eng.calls[-1]["groups"] is None                 # all atoms in order -> coalesced path
    calc.calculate(mags, vecs, basis_atom_types=[1])
    assert len(eng.calls[-1]["groups"]) == 1
    with pytest.raises(ValueError, match="summation_mode must be"):
        calc.calculate(mags, vecs, summation_mode="both")
    n = len(eng.calls)
    out = calc.calculate(mags[:0], vecs[:0], summation_mode="incoherent", basis_atom_types=[1, 2])
    assert out.sed.shape == (128, 0) and out.sed.dtype == np.float32 and len(eng.calls) == n
    tr0 = Trajectory(d["positions"][:0], d["velocities"][:0], d["types"], d["timesteps"][:0], d["box_matrix"],
                     d["box_lengths"], d["box_tilts"], d["dt_ps"])
    empty = SEDCalculator(tr0, 2, 2, 2).attach(engine=eng).calculate(mags, vecs)
    assert empty.sed.shape == (0, 0, 3) and empty.is_complex
    seam = calc._calculate_sed_for_group(vecs, np.array([], int), O.mean_positions(d["positions"]))
    assert seam.shape == (128, 3, 3) and not seam.any()


def test_mean_positions_cached_per_array(trajs):
    calc = make_calculator(trajs["a"])
    m1 = calc._mean_positions()
    assert calc._mean_positions() is m1
    np.testing.assert_array_equal(m1, O.mean_positions(trajs["a"]["positions"]))
    calc.traj.positions = calc.traj.positions.copy()
    assert calc._mean_positions() is not m1


@pytest.mark.parametrize("opt", ["C", "A", "B", "Q"])
def test_chiral_phase_golden(opt, golden, trajs):
    calc = make_calculator(trajs["c"])
    got = calc.calculate_chiral_phase(golden["z1"], golden["z2"], opt)
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, golden[f"phase_{opt}"], rtol=0, atol=1e-6)
    assert calc.calculate_chiral_phase(golden["z1"][:0], golden["z2"][:0]).shape == (0, 7)
    with pytest.raises(ValueError, match="shapes must match"):
        calc.calculate_chiral_phase(golden["z1"], golden["z2"][:3])


def test_composites(trajs):
    d = trajs["a"]
    calc = make_calculator(d).attach(engine=OracleEngine())
    sed = calc.calculate_kpath_sed([1, 1, 0], bz_coverage=2.0, n_k=5, basis_atom_types=[1, 2],
                                   summation_mode="incoherent", chiral=True, chiral_axis="y")
    mags, vecs = calc.get_k_path([1, 1, 0], 2.0, 5)
    ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs, basis_atom_types=[1, 2])
    assert sed.is_complex and rel_max(sed.sed, ref) <= 2e-6          # chirality forces coherent
    np.testing.assert_allclose(sed.phase, O.chiral_phase(ref[:, :, 0], ref[:, :, 2]), atol=1e-4)
    np.testing.assert_array_equal(sed.k_points, mags)
    plain = calc.calculate_kpath_sed("x", 1.0, 4)
    assert plain.phase is None and plain.sed.shape == (128, 4, 3)
    grid = calc.calculate_kgrid_sed("zx", (-0.5, 1.5, 0.0, 1.0), 2, 3, k_fixed=-0.3)
    _, gv, shape = calc.get_k_grid("zx", (-0.5, 1.5), (0.0, 1.0), 2, 3, -0.3)
    assert grid.k_grid_shape == shape == (2, 3) and grid.k_points.size == 0
    np.testing.assert_array_equal(grid.k_vectors, gv)
    assert calc.calculate_chiral_sed("x", 1.0, 3, chiral_axis="z").phase.shape == (128, 3)


# ---------------------------------------------------------------- .npy trajectory cache
def test_npy_cache_written_by_the_reference_loads(trajs):
    """tests/golden/npy_cache/ was written by the reference's TrajectoryLoader.save_trajectory_npy;
    npy_cache_loaded.npz holds what its own load() derived from it."""
    from conftest import GOLDEN
    from psa_amd.io import load_trajectory_npy
    d = trajs["c"]
    for mmap in (True, False):
        tr = load_trajectory_npy(GOLDEN / "npy_cache" / "run7.lammpstrj", dt=d["dt_ps"], mmap=mmap)
        np.testing.assert_array_equal(tr.positions, d["positions"])
        np.testing.assert_array_equal(tr.velocities, d["velocities"])
        np.testing.assert_array_equal(tr.types, d["types"])
        np.testing.assert_array_equal(tr.box_matrix, d["box_matrix"])
        with np.load(GOLDEN / "npy_cache_loaded.npz") as ref:
            np.testing.assert_array_equal(tr.timesteps, ref["timesteps"])
            np.testing.assert_array_equal(tr.box_lengths, ref["box_lengths"])
            np.testing.assert_array_equal(tr.box_tilts, ref["box_tilts"])
            assert tr.dt_ps == float(ref["dt_ps"])
    assert isinstance(load_trajectory_npy(GOLDEN / "npy_cache" / "run7", dt=1.0).positions, np.memmap)
    with pytest.raises(FileNotFoundError):
        load_trajectory_npy(GOLDEN / "npy_cache" / "other.lammpstrj", dt=1.0)


def test_npy_cache_writer_is_byte_compatible(trajs, tmp_path):
    from conftest import GOLDEN
    from psa_amd.io import load_trajectory_npy, save_trajectory_npy
    tr = make_trajectory(trajs["c"])
    target = tmp_path / "deep" / "run7.lammpstrj"
    assert save_trajectory_npy(tr, target) is True
    for ref_file in sorted((GOLDEN / "npy_cache").glob("run7.*.npy")):
        assert (target.parent / ref_file.name).read_bytes() == ref_file.read_bytes(), ref_file.name
    assert save_trajectory_npy(tr, target) is False                     # complete cache is left alone
    back = load_trajectory_npy(target, dt=tr.dt_ps)
    np.testing.assert_array_equal(back.positions, tr.positions)


def test_displacement_mode_takes_the_mean_from_the_device(trajs):
    d = trajs["a"]
    eng = OracleEngine()
    calc = make_calculator(d, use_displacements=True).attach(engine=eng)
    mean = calc._mean_positions()
    assert 1 in eng.slots                                             # positions were made resident
    np.testing.assert_array_equal(mean, O.mean_positions(d["positions"]))
    assert calc._mean_positions() is mean


def test_first_calculate_streams_the_upload_then_reuses_the_resident_array(trajs):
    """Host logic of residency: an array the engine does not hold goes through `project_upload`
    (upload and projection overlapped), later calls through `project`; `invalidate()` forgets."""
    from oracle_engine import OracleEngine
    eng = OracleEngine()
    calc = make_calculator(trajs["a"]).attach(engine=eng)
    mags, vecs = calc.get_k_path("100", 1.0, 8)
    first = calc.calculate(mags, vecs)
    assert eng.calls[-1].get("streamed") and eng.uploads == 1
    second = calc.calculate(mags, vecs, basis_atom_types=[1, 2], summation_mode="incoherent")
    assert not eng.calls[-1].get("streamed") and eng.uploads == 1
    calc.invalidate()
    assert calc._mean_cache is None
    third = calc.calculate(mags, vecs)
    assert eng.calls[-1].get("streamed") and eng.uploads == 2
    np.testing.assert_array_equal(first.sed, third.sed)
    assert second.sed.shape == (128, 8)


def test_ised_computes_one_bin_per_group(trajs, tmp_path):
    """iSED asks the engine for the single (k, omega) bin of each group, not for path spectra."""
    from golden import cases as C
    from oracle_engine import OracleEngine
    eng = OracleEngine()
    calc = make_calculator(trajs["a"]).attach(engine=eng)
    name, _, kw = C.ISED_CASES[1]                                   # three type groups
    calc.ised(dump_filepath=str(tmp_path / "x.dump"), **kw)
    bins = [c for c in eng.calls if c.get("single_bin")]
    assert len(bins) == 3 and len({c["i_w"] for c in bins}) == 1
    assert not any("K" in c for c in eng.calls)                     # no full projection at all


def test_rendezvous_wire_format_round_trips_and_rejects_garbage():
    from psa_amd import dist
    obj = [None, True, 3, 2.5, "text", b"\x00\x01raw", np.arange(12, dtype=np.complex64).reshape(3, 4),
           [np.float32(1.5), np.int64(7), (1, 2)], np.zeros((0, 3), np.float32)]
    back = dist._decode(dist._encode(obj))
    assert back[:5] == [None, True, 3, 2.5, "text"] and back[5] == b"\x00\x01raw"
    np.testing.assert_array_equal(back[6], obj[6])
    assert back[6].dtype == np.complex64 and back[7] == [1.5, 7, [1, 2]] and back[8].shape == (0, 3)
    assert dist._decode(dist._encode({"a": 1, "b": [2.0]})) == {"a": 1, "b": [2.0]}     # dicts with string keys do
    with pytest.raises(TypeError):
        dist._encode({1: "a"})                                      # only plain data crosses
    with pytest.raises(TypeError):
        dist._encode({"f": len})
    with pytest.raises(TypeError):
        dist._encode(np.array([object()]))
    import pickle
    with pytest.raises(Exception):
        dist._decode(pickle.dumps([1, 2, 3]))                       # never unpickled
    blob = dist._encode(np.arange(10))
    with pytest.raises(ConnectionError):
        dist._decode(blob[:-3])                                     # truncated payload


def test_tcp_rendezvous_ignores_strangers_and_duplicates():
    """Rank 0 drops connections that announce rank 0, a rank out of range, or one already taken."""
    import socket
    import struct
    import threading
    from psa_amd import dist
    port = 29000 + (os.getpid() % 2000)
    box = {}
    t = threading.Thread(target=lambda: box.setdefault("ex", dist.TcpExchange(0, 2, "127.0.0.1", port, timeout_s=20)))
    t.start()
    for bad in (0, 7):
        for _ in range(200):
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
                break
            except OSError:
                import time
                time.sleep(0.02)
        s.sendall(struct.pack("<I", bad))
        s.close()
    peer = dist.TcpExchange(1, 2, "127.0.0.1", port, timeout_s=20)
    t.join(20)
    assert "ex" in box
    res = {}
    t2 = threading.Thread(target=lambda: res.setdefault("r0", box["ex"].allgather("zero")))
    t2.start()
    assert peer.allgather(np.arange(3)) [0] == "zero"
    t2.join(20)
    np.testing.assert_array_equal(res["r0"][1], np.arange(3))
    peer.close()
    box["ex"].close()
