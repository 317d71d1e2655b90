"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and against
the golden vectors captured from the reference.  Tolerance: BASELINE.json's north_star asks
1e-5 relative on the SED intensity array; as SURVEY.md section 7-1 explains that has to be
a max-norm (near-empty bins have unbounded pointwise error), so every comparison below is
max|a-b| / max|b| <= 1e-5, and in practice lands near 1e-6."""
import numpy as np
import pytest

import cases as C
from conftest import make_calculator, rel_max
from oracle import psa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _calc(d, engine, **ctor):
    return make_calculator(d, **ctor).attach(engine=engine)


def test_device_is_mi355x(engine):
    info = engine.device_info()
    assert info["compute_units"] == 256, info
    assert info["hbm_bytes"] > 200e9, info


# ------------------------------------------------------------------ building blocks
def test_phase_table_matches_numpy(engine, trajs):
    d = trajs["b"]                                    # phases up to ~200 rad
    mean = O.mean_positions(d["positions"])
    calc = make_calculator(d)
    _, kv = calc.get_k_path([1, 1, 0], 4.0, 40)
    got = engine.debug_phase_table(mean, kv)
    ref = O.phase_table(kv, mean)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 5e-7           # sincos ulp-level; the argument is exact
    idx = np.array([5, 1, 1, 100, 127, 0], np.int32)
    got = engine.debug_phase_table(mean, kv, idx)
    assert np.max(np.abs(got - O.phase_table(kv, mean[idx]))) < 5e-7


def test_synthetic_generator_is_bit_exact(engine):
    from psa_amd import synth
    spec = synth.SyntheticSpec((2, 3, 2), 50, seed=7,
                               modes=[synth.Mode(2.0, 5, (0.3, 0.1, 0.0), 0),
                                      synth.Mode(1.0, 11, (0.0, 0.2, 0.4), 2),
                                      synth.Mode(0.5, 3, (0.1, 0.0, 0.0), 2)])
    r0, _, _ = synth.lattice(spec.cells)
    tabs = synth.mode_tables(spec, r0)
    synth.fill_device(engine, 0, spec, tabs)
    got = engine.download(0, 0, spec.n_frames)
    np.testing.assert_array_equal(got, synth.velocities_block(spec, tabs, 0, spec.n_frames))
    np.testing.assert_array_equal(engine.download(0, 17, 9), synth.velocities_block(spec, tabs, 17, 9))
    assert abs(float(got.std()) - 1.0) < 0.5


def test_mean_positions_is_bit_exact(engine, trajs):
    for name in ("a", "c"):
        pos = trajs[name]["positions"]
        engine.ensure_resident(1, pos)
        np.testing.assert_array_equal(engine.mean_positions(1), O.mean_positions(pos))
    rng = np.random.default_rng(3)
    pos = (50 + rng.standard_normal((3000, 40, 3))).astype(np.float32)   # long float32 accumulation
    engine.ensure_resident(1, pos)
    np.testing.assert_array_equal(engine.mean_positions(1), O.mean_positions(pos))


@pytest.mark.parametrize("k1", ["auto", "mfma32", "wave", "bf16x3"])
@pytest.mark.parametrize("idx", [None, [3, 9, 9, 60, 1, 17, 33]])
@pytest.mark.parametrize("disp", [False, True])
def test_projection_before_fft(engine, trajs, k1, idx, disp):
    from psa_amd import _hip
    d = trajs["a"]
    mean = O.mean_positions(d["positions"])
    _, kv = make_calculator(d).get_k_path([1, 1, 0], 2.0, 11)
    engine.set_k1({"auto": _hip.K1_AUTO, "mfma32": _hip.K1_MFMA32, "wave": _hip.K1_WAVE,
                   "bf16x3": _hip.K1_SPLIT_BF16}[k1])
    try:
        src = d["positions"] if disp else d["velocities"]
        engine.ensure_resident(1 if disp else 0, src)
        got = engine.debug_project_only(1 if disp else 0, mean, kv, idx,
                                        _hip.F_DISPLACEMENTS if disp else 0)
    finally:
        engine.set_k1(_hip.K1_AUTO)
    sel = np.arange(src.shape[1]) if idx is None else np.asarray(idx)
    data = src[:, sel, :] - mean[sel][None] if disp else src[:, sel, :]
    ref = O.project_group(data, O.phase_table(kv, mean[sel]))      # (T,K,3)
    assert rel_max(got.transpose(2, 0, 1), ref) < 2e-6


# ------------------------------------------------------------------ golden cases end to end
@pytest.mark.parametrize("case", C.CALC_CASES, ids=[c["name"] for c in C.CALC_CASES])
def test_calculate_matches_reference_golden(case, golden, trajs, engine):
    d = trajs[case["traj"]]
    name = case["name"]
    calc = _calc(d, engine, **case.get("ctor", {}))
    mags, vecs, shape = C.k_from_spec(calc, case["k"])
    # np.linspace rounds differently by 1 ulp on different host CPUs; the k-vectors are
    # INPUTS of the path, so feed the very ones the reference was run with
    np.testing.assert_allclose(vecs, golden[f"{name}/k_vecs"], rtol=3e-7, atol=1e-9)
    mags, vecs = golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"]
    kw = C.realise_kw(case.get("kw", {}))
    if shape is not None:
        kw["k_grid_shape"] = shape
    sed = calc.calculate(mags, vecs, **kw)
    ref = golden[f"{name}/sed"]
    assert sed.sed.dtype == ref.dtype and sed.sed.shape == ref.shape
    assert sed.is_complex == bool(golden[f"{name}/is_complex"])
    np.testing.assert_array_equal(sed.freqs, golden[f"{name}/freqs"])
    assert rel_max(sed.intensity, golden[f"{name}/intensity"]) <= TOL
    assert rel_max(sed.sed, ref) <= TOL
    assert (sed.k_grid_shape or ()) == tuple(golden[f"{name}/grid_shape"])


@pytest.mark.parametrize("case", C.CALC_WIDE_CASES, ids=[c["name"] for c in C.CALC_WIDE_CASES])
def test_calculate_wide_matches_reference_golden(case, golden, trajs, engine):
    """More than 16 k-vectors: the product-default "2 x f16" projection kernel (64- and 128-row
    blocks, whole trajectory and index lists / type groups, displacement mode, phases up to
    ~200 rad) held directly to REFERENCE output (tests/golden/calc_wide.npz: whole intensity,
    every WIDE_SED_STRIDE-th complex row)."""
    d = trajs[case["traj"]]
    name = case["name"]
    calc = _calc(d, engine, **case.get("ctor", {}))
    mags, vecs, shape = C.k_from_spec(calc, case["k"])
    np.testing.assert_allclose(vecs, golden[f"{name}/k_vecs"], rtol=3e-7, atol=1e-9)
    mags, vecs = golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"]
    kw = C.realise_kw(case.get("kw", {}))
    if shape is not None:
        kw["k_grid_shape"] = shape
    sed = calc.calculate(mags, vecs, **kw)
    assert sed.sed.shape == tuple(golden[f"{name}/sed_shape"])
    assert sed.is_complex == bool(golden[f"{name}/is_complex"])
    assert rel_max(sed.intensity if sed.is_complex else sed.sed, golden[f"{name}/intensity"]) <= TOL
    assert rel_max(sed.sed[::C.WIDE_SED_STRIDE], golden[f"{name}/sed_rows"]) <= TOL
    # a second call takes the cached split planes of the group (built on first use)
    again = calc.calculate(mags, vecs, **kw)
    assert rel_max(again.intensity if again.is_complex else again.sed, golden[f"{name}/intensity"]) <= TOL


@pytest.mark.parametrize("case", C.CALC_W256_CASES, ids=[c["name"] for c in C.CALC_W256_CASES])
def test_calculate_w256_matches_reference_golden(case, golden, trajs, engine):
    """k-lists that fill an even number of 128-row blocks (100 and 250 vectors): the 256-row form of the planes kernel
    (k1_planes_wide.hip) held directly to REFERENCE output (tests/golden/calc_w256.npz) -- and the 128-row forms on the
    same lists (PSA_OPT_K1_WIDE off)."""
    from psa_amd import _hip
    test_calculate_wide_matches_reference_golden(case, golden, trajs, engine)
    engine.set_option(_hip.OPT_K1_WIDE, 0)
    try:
        test_calculate_wide_matches_reference_golden(case, golden, trajs, engine)
    finally:
        engine.set_option(_hip.OPT_K1_WIDE, 1)


@pytest.mark.parametrize("case", C.CALC_SYM_CASES, ids=[c["name"] for c in C.CALC_SYM_CASES])
def test_calculate_sym_matches_reference_golden(case, golden, trajs, engine):
    """k-lists with (k, -k) pairs and repeated vectors (grids symmetric about Gamma, with and without
    Gamma itself, partly symmetric, T = 100, displacement mode, incoherent, 400 vectors through the
    block-by-block result path, a hand-mirrored path): the library projects one vector of each pair
    (PSA_OPT_FOLD_PAIRS) and must land on the REFERENCE's output, which computed every vector."""
    from psa_amd import _hip
    d = trajs[case["traj"]]
    name = case["name"]
    calc = _calc(d, engine, **case.get("ctor", {}))
    mags, vecs, shape = C.k_from_spec(calc, case["k"])
    np.testing.assert_allclose(vecs, golden[f"{name}/k_vecs"], rtol=3e-7, atol=1e-9)
    mags, vecs = golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"]
    kmap, unique = _hip.k_pairs(vecs)
    assert len(unique) < len(vecs)                                  # these lists do fold
    kw = C.realise_kw(case.get("kw", {}))
    if shape is not None:
        kw["k_grid_shape"] = shape
    for attempt in range(3):             # streamed first call; resident (pipelined for 400 vectors); cached planes
        sed = calc.calculate(mags, vecs, **kw)
        assert sed.sed.shape == tuple(golden[f"{name}/sed_shape"])
        assert sed.is_complex == bool(golden[f"{name}/is_complex"])
        assert rel_max(sed.sed[::C.WIDE_SED_STRIDE], golden[f"{name}/sed_rows"]) <= TOL
        assert rel_max(sed.intensity if sed.is_complex else sed.sed, golden[f"{name}/intensity"]) <= TOL
    # ... and the same with folding switched off projects every vector: same result to rounding
    engine.set_option(_hip.OPT_FOLD_PAIRS, 0)
    try:
        plain = calc.calculate(mags, vecs, **kw)
    finally:
        engine.set_option(_hip.OPT_FOLD_PAIRS, 1)
    assert rel_max(plain.sed, sed.sed) <= 2e-6


def test_folded_list_projected_in_two_shards_then_mapped(golden, trajs, engine):
    """What a sharded run does on each rank, on one GPU: the caller folds the list (`k_pairs`),
    projects the unique vectors as two blocks of slab rows, installs the k map (`psa_sed_set_kmap`) and
    finalizes -- complex and intensity results equal the reference's, which computed all 36 vectors."""
    from psa_amd import _hip
    d = trajs["a"]
    mean = O.mean_positions(d["positions"])
    engine.ensure_resident(_hip.SLOT_VELOCITIES, d["velocities"])
    T = d["velocities"].shape[0]
    for name, groups, flags in (("s_grid_xy_6x6_coh", None, 0),
                                ("s_grid_xy_6x6_inc", [np.flatnonzero(d["types"] == 1), np.flatnonzero(d["types"] == 2)],
                                 _hip.F_INTENSITY)):
        vecs = golden[f"{name}/k_vecs"]
        kmap, unique = _hip.k_pairs(vecs)
        assert len(unique) == 18
        u = vecs[unique]
        engine.project(_hip.SLOT_VELOCITIES, mean, u[:11], groups, flags, K_total=18, k_offset=0)
        engine.project(_hip.SLOT_VELOCITIES, mean, u[11:], groups, flags, K_total=18, k_offset=11)
        with pytest.raises(_hip.PsaHipError):
            engine.finalize(T, 36, bool(flags))                   # 18 rows until the map is installed
        with pytest.raises(_hip.PsaHipError):
            engine.set_kmap(np.full(36, 18, np.uint32))           # row 18 does not exist
        engine.set_kmap(kmap)
        if flags:
            got = engine.finalize(T, 36, True)
            assert rel_max(got, golden[f"{name}/intensity"]) <= TOL
        else:
            got, inten = engine.finalize(T, 36, False, with_intensity=True)
            assert rel_max(got[::C.WIDE_SED_STRIDE], golden[f"{name}/sed_rows"]) <= TOL
            assert rel_max(inten, golden[f"{name}/intensity"]) <= TOL
            np.testing.assert_allclose(inten, np.sum(np.abs(got) ** 2, axis=-1), rtol=5e-6)
            phase = engine.result_chiral_phase(T, 36, 0, 1)        # the other result_* calls see 36 columns too
            assert phase.shape == (T, 36)


@pytest.mark.parametrize("name", ["w_coh_all_k40", "w_coh_all_k140", "w_idx_list_k140", "w_displacements_k40",
                                  "w_large_phase_k140", "w_inc_types_nested_k140", "s_grid_xy_20x20"])
def test_loader_wavefront_form_of_the_planes_kernel(name, golden, trajs, engine):
    """PSA_OPT_K1_LOADER_WAVES: 128-row M blocks projected by k1_planes_lw_kernel (12 wavefronts per
    workgroup, 4 of them issue all LDS-DMA, component-major MFMAs, 4-slot ring) -- held to the same
    REFERENCE outputs as the default form, and to the default form itself (same products, same
    8-stage folds: the sums differ only in the order of the three terms' accumulation)."""
    from psa_amd import _hip
    case = next(c for c in C.CALC_WIDE_CASES + C.CALC_SYM_CASES if c["name"] == name)
    d = trajs[case["traj"]]
    calc = _calc(d, engine, **case.get("ctor", {}))
    mags, vecs = golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"]
    kw = C.realise_kw(case.get("kw", {}))
    engine.set_option(_hip.OPT_PLANES_EAGER, 1)                      # index lists get their planes at once
    try:
        engine.set_option(_hip.OPT_K1_LOADER_WAVES, 0)               # never: the eight-wavefront form
        plain = calc.calculate(mags, vecs, **kw)
        engine.timings()
        engine.set_option(_hip.OPT_K1_LOADER_WAVES, 1)               # the default: the loader-wavefront form
        got = calc.calculate(mags, vecs, **kw)
        assert engine.timings()["project"] > 0
    finally:
        engine.set_option(_hip.OPT_K1_LOADER_WAVES, 1)
        engine.set_option(_hip.OPT_PLANES_EAGER, 0)
    assert got.sed.shape == tuple(golden[f"{name}/sed_shape"])
    assert rel_max(got.sed[::C.WIDE_SED_STRIDE], golden[f"{name}/sed_rows"]) <= TOL
    assert rel_max(got.intensity if got.is_complex else got.sed, golden[f"{name}/intensity"]) <= TOL
    assert rel_max(got.sed, plain.sed) <= 1e-6


def test_lists_without_pairs_are_projected_whole(golden, engine):
    """k_fixed != 0: no vector's negation is in the grid -- nothing folds (the shortcut must not be
    taken); same for a k-path from Gamma outwards."""
    from psa_amd import _hip
    for name in ("w_grid_xy_6x7", "w_coh_all_k140"):
        vecs = golden[f"{name}/k_vecs"]
        kmap, unique = _hip.k_pairs(vecs)
        assert len(unique) == len(vecs) and np.array_equal(kmap, np.arange(len(vecs)))


def test_config1_matches_reference(engine):
    """BASELINE configuration 1 at full size -- 512 atoms x 4096 steps x 32 k-points, [100] path,
    bz 4.0 -- through the public API against the REAL reference's output (c1_reference.npz)."""
    from conftest import GOLDEN
    from psa_amd import SEDCalculator, Trajectory
    spec, req, d = C.c1_inputs()
    with np.load(GOLDEN / "c1_reference.npz") as z:
        ref = {k: z[k] for k in z.files}
    tr = Trajectory(d["positions"], d["velocities"], d["types"], d["timesteps"], d["box_matrix"],
                    d["box_lengths"], d["box_tilts"], spec.dt_ps)
    calc = SEDCalculator(tr, *spec.cells).attach(engine=engine)
    mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
    np.testing.assert_allclose(vecs, ref["k_vecs"], rtol=3e-7, atol=1e-9)
    sed = calc.calculate(ref["k_mags"], ref["k_vecs"])
    assert sed.sed.shape == (4096, 32, 3)
    np.testing.assert_array_equal(sed.freqs, ref["freqs"])
    assert rel_max(sed.intensity, ref["intensity"]) <= TOL
    assert rel_max(sed.sed[ref["rows"]], ref["sed_rows"]) <= TOL
    inc = calc.calculate(ref["k_mags"], ref["k_vecs"], basis_atom_types=[1, 2], summation_mode="incoherent")
    assert not inc.is_complex and rel_max(inc.sed, ref["intensity_incoherent_types12"]) <= TOL
    # the composite the north star names
    via = calc.calculate_kpath_sed(req["direction"], req["bz_coverage"], req["n_k"])
    assert rel_max(via.intensity, ref["intensity"]) <= TOL


def test_integration_md_seam_patch_runs_verbatim(golden, trajs):
    """INTEGRATION.md, option B: the ctypes stub a PSA maintainer would add (`HipSeam`, bound to
    psa_sed_calculate) is executed exactly as printed there -- only the library path is made
    absolute -- and must reproduce the reference's `_calculate_sed_for_group` output."""
    import re
    import types as pytypes
    from conftest import ROOT
    from psa_amd import _hip
    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class HipSeam" in b)
    assert 'C.CDLL("libpsa_hip.so")' in stub
    stub = stub.replace('C.CDLL("libpsa_hip.so")', f'C.CDLL("{_hip.LIB_PATH}")')
    mod = pytypes.ModuleType("_psa_hip_from_integration_md")
    exec(compile(stub, "INTEGRATION.md", "exec"), mod.__dict__)
    d = trajs["a"]
    traj = pytypes.SimpleNamespace(positions=d["positions"], velocities=d["velocities"])
    seam = mod.HipSeam(traj, use_displacements=False)
    got = seam(golden["seam/k_vecs"], golden["seam/idx"], golden["seam/mean_pos"])
    assert got.dtype == np.complex64 and rel_max(got, golden["seam/sed"]) <= TOL
    # a k-list long enough for the default kernel, all atoms, both data modes
    for disp in (False, True):
        seam = mod.HipSeam(traj, use_displacements=disp)
        case = "w_displacements_k40" if disp else "w_coh_all_k40"
        got = seam(golden[f"{case}/k_vecs"], np.arange(d["positions"].shape[1]), O.mean_positions(d["positions"]))
        assert rel_max(got[::C.WIDE_SED_STRIDE], golden[f"{case}/sed_rows"]) <= TOL


def test_seam_matches_reference(golden, trajs, engine):
    calc = _calc(trajs["a"], engine)
    got = calc._calculate_sed_for_group(golden["seam/k_vecs"], golden["seam/idx"], golden["seam/mean_pos"])
    assert got.dtype == np.complex64 and rel_max(got, golden["seam/sed"]) <= TOL
    empty = calc._calculate_sed_for_group(golden["seam/k_vecs"], np.array([], int), golden["seam/mean_pos"])
    np.testing.assert_array_equal(empty, golden["seam_empty/sed"])


def test_known_answer_single_atom(engine):
    from psa_amd import SEDCalculator, Trajectory
    T = 32
    pos = np.zeros((T, 1, 3), np.float32)
    pos[:, 0, 0] = 1
    vel = np.zeros((T, 1, 3), np.float32)
    vel[:, 0, 0] = np.cos(2 * np.pi * 4 * np.arange(T) / T)
    tr = Trajectory(pos, vel, np.ones(1, np.int32), np.arange(T, dtype=np.float32),
                    np.eye(3, dtype=np.float32) * 10, np.full(3, 10, np.float32), np.zeros(3, np.float32), 0.01)
    sed = SEDCalculator(tr, 1, 1, 1).attach(engine=engine).calculate(
        np.array([0.5], np.float32), np.array([[0.5, 0, 0]], np.float32))
    want = 0.5 * np.exp(0.5j)
    assert abs(sed.sed[4, 0, 0] - want) < 1e-6 and abs(sed.sed[28, 0, 0] - want) < 1e-6
    mask = np.ones(T, bool)
    mask[[4, 28]] = False
    assert np.max(np.abs(sed.sed[mask])) < 1e-6


# ------------------------------------------------------------------ every tile variant, ragged shapes
def _random_traj(n_atoms, n_frames, seed, spread=40.0):
    from psa_amd import Trajectory
    rng = np.random.default_rng(seed)
    r0 = (rng.random((n_atoms, 3)) * spread).astype(np.float32)
    pos = (r0[None] + 0.05 * rng.standard_normal((n_frames, n_atoms, 3))).astype(np.float32)
    vel = rng.standard_normal((n_frames, n_atoms, 3)).astype(np.float32)
    types = (1 + (np.arange(n_atoms) % 3)).astype(np.int32)
    box = np.eye(3, dtype=np.float32) * spread
    return Trajectory(pos, vel, types, np.arange(n_frames, dtype=np.float32), box,
                      np.full(3, spread, np.float32), np.zeros(3, np.float32), 0.002)


@pytest.mark.parametrize("n_atoms,n_frames,n_k", [
    (96, 200, 3),      # M block 32, ragged T
    (130, 257, 24),    # M block 64, N % 4 != 0 -> per-atom loader
    (256, 512, 50),    # M block 128
    (200, 300, 140),   # M block 256, two M blocks, ragged everything
    (33, 64, 1),       # single k, one atom over a stage boundary
    (1000, 1024, 128),
])
def test_shapes_against_oracle(engine, n_atoms, n_frames, n_k):
    from psa_amd import SEDCalculator
    tr = _random_traj(n_atoms, n_frames, seed=n_atoms + n_k)
    calc = SEDCalculator(tr, 2, 2, 2).attach(engine=engine)
    mags, vecs = calc.get_k_path([1, 0.3, 0.1], 3.0, n_k)
    got = calc.calculate(mags, vecs)
    ref, _, _ = O.calculate(tr.positions, tr.velocities, tr.types, tr.dt_ps, vecs)
    assert rel_max(got.intensity, O.intensity(ref)) <= TOL
    assert rel_max(got.sed, ref) <= TOL
    got = calc.calculate(mags, vecs, basis_atom_types=[1, 2, 3], summation_mode="incoherent")
    ref, _, cx = O.calculate(tr.positions, tr.velocities, tr.types, tr.dt_ps, vecs,
                             basis_atom_types=[1, 2, 3], summation_mode="incoherent")
    assert not cx and not got.is_complex and rel_max(got.sed, ref) <= TOL


@pytest.mark.parametrize("n_atoms,n_frames,n_k,idx", [
    (128, 257, 24, None),                          # f16 kernel, 64-row M block, row DMA
    (130, 257, 24, None),                          # ... per-atom gather DMA without an index list (N % 4 != 0)
    (128, 257, 24, [5, 3, 3, 100, 77, 2] * 9),     # ... gather through an index list with duplicates
    (132, 64, 17, None),                           # shortest k-list the f16 kernel takes
    (130, 300, 50, None),                          # 128-row M block, gather without an index list
    (256, 300, 50, list(range(0, 256, 3))),        # 128-row M block, index list
    (192, 320, 200, None),                         # two 128-row M blocks, row DMA, several chains per group
    (4096, 96, 40, None),                          # 128 stages: 16 fold periods
])
def test_f16_kernel_variants(engine, n_atoms, n_frames, n_k, idx):
    """Every instantiation of k1_pair_kernel (row tiles per wavefront x row / gather DMA) against the
    oracle, complex output."""
    from psa_amd import SEDCalculator
    tr = _random_traj(n_atoms, n_frames, seed=n_atoms + n_k)
    calc = SEDCalculator(tr, 2, 2, 2).attach(engine=engine)
    mags, vecs = calc.get_k_path([1, 0.3, 0.1], 3.0, n_k)
    kw = {} if idx is None else {"basis_atom_indices": idx}
    got = calc.calculate(mags, vecs, **kw)
    ref, _, _ = O.calculate(tr.positions, tr.velocities, tr.types, tr.dt_ps, vecs, **kw)
    assert rel_max(got.sed, ref) <= TOL
    assert rel_max(got.intensity, O.intensity(ref)) <= TOL


def test_f16_kernel_is_exactly_scale_invariant(engine):
    """The f16 kernel rescales the data by a power of two taken from its largest magnitude and
    undoes it exactly: multiplying the trajectory by 2^+-40 multiplies the result by exactly that."""
    from psa_amd import SEDCalculator
    base = _random_traj(256, 192, seed=5)
    calc = SEDCalculator(base, 2, 2, 2).attach(engine=engine)
    mags, vecs = calc.get_k_path("xyz", 2.0, 40)
    ref = calc.calculate(mags, vecs).sed
    for e in (-40, 40):
        tr = _random_traj(256, 192, seed=5)
        tr.velocities = (tr.velocities * np.float32(2.0 ** e)).astype(np.float32)
        got = SEDCalculator(tr, 2, 2, 2).attach(engine=engine).calculate(mags, vecs).sed
        assert np.array_equal(got, ref * np.complex64(2.0 ** e))


def test_f16_kernel_wide_dynamic_range(engine):
    """One atom 1e5 times faster than the rest, a third of the atoms 1e-4 times slower: every class
    still contributes at float32 accuracy (the scale is set by the outlier)."""
    from psa_amd import SEDCalculator
    tr = _random_traj(384, 160, seed=8)
    v = tr.velocities.copy()
    v[:, 7, :] *= 1e5
    v[:, 100:228, :] *= 1e-4
    tr.velocities = v
    calc = SEDCalculator(tr, 2, 2, 2).attach(engine=engine)
    mags, vecs = calc.get_k_path([1, 0.3, 0.1], 3.0, 48)
    for idx in (None, list(range(100, 228)), [i for i in range(384) if i != 7]):
        kw = {} if idx is None else {"basis_atom_indices": idx}
        got = calc.calculate(mags, vecs, **kw)
        ref, _, _ = O.calculate(tr.positions, tr.velocities, tr.types, tr.dt_ps, vecs, **kw)
        assert rel_max(got.sed, ref) <= TOL


def test_f16_kernel_linearity_and_atom_order(engine):
    """Properties that need no reference: the complex SED is linear in the velocities, and summing
    the atoms in another order (an index list that is a permutation -> the gather form of the kernel)
    gives the row-DMA result to rounding."""
    from psa_amd import SEDCalculator
    tr1, tr2 = _random_traj(512, 256, seed=21), _random_traj(512, 256, seed=22)
    tr2.positions = tr1.positions
    mags, vecs = SEDCalculator(tr1, 2, 2, 2).get_k_path([1, 1, 0], 2.0, 64)

    def run(vel, **kw):
        tr = _random_traj(512, 256, seed=21)
        tr.velocities = vel
        return SEDCalculator(tr, 2, 2, 2).attach(engine=engine).calculate(mags, vecs, **kw).sed.astype(np.complex128)

    s1, s2 = run(tr1.velocities), run(tr2.velocities)
    both = run((np.float32(0.75) * tr1.velocities + tr2.velocities).astype(np.float32))
    scale = np.abs(both).max()
    assert np.abs(both - (0.75 * s1 + s2)).max() / scale < 2e-6
    perm = np.random.default_rng(3).permutation(512)
    assert np.abs(run(tr1.velocities, basis_atom_indices=perm) - s1).max() / np.abs(s1).max() < 2e-6


def test_non_finite_data_takes_the_bf16_kernel(engine):
    """NaN / Inf in the array: no scale exists, the kernel that needs none runs, and the non-finite
    values propagate to every output that depends on them, as in the reference."""
    from psa_amd import SEDCalculator
    tr = _random_traj(128, 96, seed=3)
    v = tr.velocities.copy()
    v[10, 5, 1] = np.inf
    v[20, 6, 2] = np.nan
    tr.velocities = v
    calc = SEDCalculator(tr, 2, 2, 2).attach(engine=engine)
    mags, vecs = calc.get_k_path("x", 1.0, 24)
    got = calc.calculate(mags, vecs).sed
    with np.errstate(all="ignore"):
        ref, _, _ = O.calculate(tr.positions, tr.velocities, tr.types, tr.dt_ps, vecs)
    assert not np.isfinite(got[:, :, 1]).any() and not np.isfinite(got[:, :, 2]).any()      # FFT spreads them
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    assert rel_max(got[:, :, 0], ref[:, :, 0]) <= TOL


@pytest.mark.parametrize("n_atoms, n_k", [(300, 37), (1024, 100), (512, 9)])
def test_kernels_agree_with_each_other(engine, n_atoms, n_k):
    """Split-precision (3 x bf16) tile kernel vs exact-fp32 MFMA kernel vs the shuffle kernel:
    three different arithmetic schedules, same q to float32 rounding level."""
    from psa_amd import SEDCalculator, _hip
    tr = _random_traj(n_atoms, 400, seed=9)
    calc = SEDCalculator(tr, 1, 1, 1).attach(engine=engine)
    mags, vecs = calc.get_k_path("xyz", 2.0, n_k)
    out = {}
    try:
        for name, sel in (("auto", _hip.K1_AUTO), ("mfma32", _hip.K1_MFMA32), ("wave", _hip.K1_WAVE),
                          ("bf16x3", _hip.K1_SPLIT_BF16)):
            engine.set_k1(sel)
            out[name] = calc.calculate(mags, vecs).sed
    finally:
        engine.set_k1(_hip.K1_AUTO)
    assert rel_max(out["mfma32"], out["wave"]) < 2e-6
    assert rel_max(out["auto"], out["mfma32"]) < 2e-6


def test_device_intensity_and_chiral_phase(engine, trajs):
    calc = _calc(trajs["a"], engine)
    sed = calc.calculate_kpath_sed([1, 1, 0], 2.0, 9, chiral=True, chiral_axis="x",
                                   summation_mode="incoherent")
    assert sed.is_complex and sed.phase is not None and sed.phase.dtype == np.float32
    T, K = sed.sed.shape[:2]
    np.testing.assert_allclose(engine.result_intensity(T, K), sed.intensity, rtol=2e-6, atol=1e-12)
    ref = O.chiral_phase(sed.sed[:, :, 1], sed.sed[:, :, 2], "C")
    # the folded phase is continuous, but atan2 of tiny amplitudes is ill-conditioned:
    # compare where both components carry signal
    strong = (np.abs(sed.sed[:, :, 1]) > 1e-3) & (np.abs(sed.sed[:, :, 2]) > 1e-3)
    assert strong.mean() > 0.5
    assert np.max(np.abs(sed.phase - ref)[strong]) < 1e-4
    grid = calc.calculate_kgrid_sed("xy", (-1, 1, -0.5, 0.5), 3, 2, k_fixed=0.1)
    assert grid.k_grid_shape == (3, 2) and grid.sed.shape == (T, 6, 3)


def test_errors_surface_as_python_exceptions(engine, trajs):
    from psa_amd import _hip
    calc = _calc(trajs["a"], engine)
    mags, vecs = calc.get_k_path("x", 1.0, 4)
    with pytest.raises(ValueError, match="out of bounds"):
        calc.calculate(mags, vecs, basis_atom_indices=[0, 64])
    with pytest.raises(ValueError, match="summation_mode"):
        calc.calculate(mags, vecs, summation_mode="x")
    with pytest.raises(ValueError, match="out of bounds"):       # the ABI checks too
        engine.project(0, np.zeros((64, 3), np.float32), vecs, [np.array([99])])
    with pytest.raises(_hip.PsaHipError):
        engine.project(0, np.zeros((64, 3), np.float32), vecs, [np.array([1]), np.array([2])], 0)
    empty = calc.calculate(mags[:0], vecs[:0])
    assert empty.sed.shape == (trajs["a"]["positions"].shape[0], 0, 3)


def test_single_rank_communicator_and_gather(engine, trajs):
    """RCCL bring-up on one GPU: unique id, ncclCommInitRank(nranks=1), the gather entry (a
    no-op exchange with itself) and the barrier all-reduce must work and leave results intact."""
    from psa_amd import _hip, dist
    d = trajs["a"]
    mean = O.mean_positions(d["positions"])
    _, kv = make_calculator(d).get_k_path("x", 1.0, 6)
    engine.ensure_resident(0, d["velocities"])
    uid = engine.new_unique_id()
    assert len(uid) == _hip.UNIQUE_ID_BYTES
    engine.comm_init(uid, 0, 1)
    try:
        off, cnt = dist.shard_ranges(6, 1)
        engine.project(0, mean, kv, None, 0, K_total=6, k_offset=0)
        engine.gather(0, off, cnt)
        engine.gather(-1, off, cnt)
        engine.barrier()
        got = engine.finalize(d["velocities"].shape[0], 6, False)
    finally:
        engine.comm_destroy()
    ref = O.sed_for_group(d["positions"], d["velocities"], kv, np.arange(64), mean)
    assert rel_max(got, ref) <= TOL


def test_k_offset_slab_rows(engine, trajs):
    """Two projections into one slab (rows 0-3 and 4-6), as two ranks would, equal one pass."""
    d = trajs["a"]
    mean = O.mean_positions(d["positions"])
    _, kv = make_calculator(d).get_k_path([1, 1, 0], 2.0, 7)
    engine.ensure_resident(0, d["velocities"])
    T = d["velocities"].shape[0]
    for flags, groups in ((0, None), (2, [np.arange(0, 64, 2), np.arange(1, 64, 2)])):
        engine.project(0, mean, kv[:4], groups, flags, K_total=7, k_offset=0)
        engine.project(0, mean, kv[4:], groups, flags, K_total=7, k_offset=4)
        parts = engine.finalize(T, 7, bool(flags))
        engine.project(0, mean, kv, groups, flags)
        whole = engine.finalize(T, 7, bool(flags))
        assert rel_max(parts, whole) < 1e-6


def test_memory_mapped_npy_cache_trajectory(engine, trajs):
    """A trajectory read from the reference's .npy cache (np.memmap arrays) runs as is, in
    velocity mode and in displacement mode (mean computed on the device, bit-exact)."""
    from conftest import GOLDEN
    from psa_amd import SEDCalculator
    from psa_amd.io import load_trajectory_npy
    d = trajs["c"]
    tr = load_trajectory_npy(GOLDEN / "npy_cache" / "run7.lammpstrj", dt=d["dt_ps"])
    assert isinstance(tr.velocities, np.memmap)
    for disp in (False, True):
        calc = SEDCalculator(tr, *d["cells"], use_displacements=disp).attach(engine=engine)
        mags, vecs = calc.get_k_path("z", 2.0, 5)
        got = calc.calculate(mags, vecs)
        ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs,
                                use_displacements=disp)
        assert rel_max(got.sed, ref) <= TOL
        np.testing.assert_array_equal(calc._mean_positions(), O.mean_positions(d["positions"]))


def test_slab_rows_round_trip(engine, trajs):
    """psa_slab_read / psa_slab_write: the host transport that stands in for the RCCL gather."""
    d = trajs["a"]
    mean = O.mean_positions(d["positions"])
    _, kv = make_calculator(d).get_k_path([1, 1, 0], 2.0, 7)
    engine.ensure_resident(0, d["velocities"])
    T = d["velocities"].shape[0]
    engine.project(0, mean, kv, None, 0)
    whole = engine.finalize(T, 7, False)
    rows = engine.slab_read(2, 3, T, False)
    assert rows.shape == (3, 3, T) and rows.dtype == np.complex64
    # a second "rank" computes only rows 0-1 and 5-6, the missing rows arrive through the host
    engine.project(0, mean, kv[:2], None, 0, K_total=7, k_offset=0)
    engine.project(0, mean, kv[5:], None, 0, K_total=7, k_offset=5)
    engine.slab_write(2, rows)
    np.testing.assert_allclose(engine.finalize(T, 7, False), whole, rtol=0, atol=1e-6 * np.abs(whole).max())
    from psa_amd import _hip
    with pytest.raises(_hip.PsaHipError):
        engine.slab_read(5, 3, T, False)


@pytest.mark.parametrize("n_atoms, n_frames, n_k", [(1, 1, 1), (3, 2, 2), (5, 1, 40)])
def test_degenerate_shapes(engine, n_atoms, n_frames, n_k):
    from psa_amd import SEDCalculator
    tr = _random_traj(n_atoms, n_frames, seed=1)
    calc = SEDCalculator(tr, 1, 1, 1).attach(engine=engine)
    mags, vecs = calc.get_k_path("y", 1.0, n_k)
    got = calc.calculate(mags, vecs)
    ref, _, _ = O.calculate(tr.positions, tr.velocities, tr.types, tr.dt_ps, vecs)
    assert got.sed.shape == (n_frames, n_k, 3) and rel_max(got.sed, ref) <= TOL


def test_calculate_from_worker_threads(engine, trajs):
    """The reference GUI calculates on daemon threads (psa_gui.py:1015, :2246): concurrent
    calculations on one context must serialise and both be right."""
    import threading
    d = trajs["a"]
    calcs = [_calc(d, engine) for _ in range(2)]
    ks = [calcs[0].get_k_path("x", 1.0, 5), calcs[1].get_k_path([1, 1, 0], 2.0, 9)]
    out, errors = [None, None], []

    def work(i):
        try:
            for _ in range(5):
                out[i] = calcs[i].calculate(*ks[i]).sed
        except Exception as e:                       # surfaced below
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    for i in range(2):
        ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], ks[i][1])
        assert out[i].shape == ref.shape and rel_max(out[i], ref) <= TOL


def test_results_come_in_recycled_page_locked_memory(engine, trajs):
    """Results of 1 MiB and more are ordinary writable ndarrays on page-locked host memory; the block
    returns to the pool when its last view dies and is handed out again."""
    import gc
    from psa_amd import _hip
    calc = _calc(trajs["a"], engine)
    mags, vecs = calc.get_k_path("x", 1.0, 400)                     # (128, 400, 3) complex64 = 1.2 MB
    sed = calc.calculate(mags, vecs).sed
    assert sed.flags.writeable and sed.flags.c_contiguous and sed.dtype == np.complex64
    ref, _, _ = O.calculate(trajs["a"]["positions"], trajs["a"]["velocities"], trajs["a"]["types"],
                            trajs["a"]["dt_ps"], vecs)
    assert rel_max(sed, ref) <= TOL
    address, view = sed.ctypes.data, sed[3:5]
    del sed
    gc.collect()
    assert address not in sum(_hip._pinned_pool._idle.values(), [])   # a view still holds the block
    del view
    gc.collect()
    assert address in sum(_hip._pinned_pool._idle.values(), [])
    again = calc.calculate(mags, vecs).sed
    assert again.ctypes.data == address and rel_max(again, ref) <= TOL
    small = calc.calculate(mags[:4], vecs[:4]).sed                   # 12 KB: ordinary memory
    assert small.ctypes.data not in sum(_hip._pinned_pool._idle.values(), [])


@pytest.mark.parametrize("idx", [None, [4, 9, 9, 60, 1, 17, 33, 2]])
def test_displacement_mode_on_the_fast_kernels(engine, trajs, idx):
    """use_displacements=True: positions - mean is materialised once on the device (the reference's
    temporary, sed_calculator.py:70-72) and projected by the split-precision kernels; the array is
    reused while positions and mean stay the same and rebuilt when they change."""
    from psa_amd import _hip
    d = trajs["a"]
    calc = _calc(d, engine, use_displacements=True)
    mags, vecs = calc.get_k_path([1, 1, 0], 2.0, 40)                 # f16 kernel
    kw = {} if idx is None else {"basis_atom_indices": idx}
    got = calc.calculate(mags, vecs, **kw)
    ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs,
                            use_displacements=True, **kw)
    assert rel_max(got.sed, ref) <= TOL
    again = calc.calculate(mags[:9], vecs[:9], **kw)                 # bf16 kernel, cached displacements
    assert rel_max(again.sed, ref[:, :9]) <= TOL
    try:                                                             # the float32 loader agrees
        engine.set_k1(_hip.K1_MFMA32)
        exact = calc.calculate(mags, vecs, **kw)
    finally:
        engine.set_k1(_hip.K1_AUTO)
    assert rel_max(got.sed, exact.sed) <= 2e-6
