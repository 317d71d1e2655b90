"""The CPU oracle against the golden vectors captured from the real reference
(tests/golden/make_golden.py).  This is what pins the oracle; everything GPU is then
checked against the oracle and against the same fixtures."""
import numpy as np
import pytest

import cases as C
from conftest import rel_max
from oracle import psa_oracle as O


def _k_for(case, d):
    spec = case["k"]
    cx, cy, cz = d["cells"]
    if spec[0] == "path":
        _, direction, cov, n_k, lat = spec
        mags, vecs = O.k_path(d["box_matrix"], cx, cy, cz, direction, cov, n_k, lat)
        return mags, vecs
    if spec[0] == "mirrored_path":
        _, direction, cov, n_k, lat = spec
        mags, vecs = O.k_path(d["box_matrix"], cx, cy, cz, direction, cov, n_k, lat)
        return (np.concatenate([-mags[::-1], mags[1:], mags[3:5]]).astype(np.float32),
                np.concatenate([-vecs[::-1], vecs[1:], vecs[3:5]]).astype(np.float32))
    _, plane, rx, ry, nkx, nky, fixed = spec
    mags, vecs, _ = O.k_grid(plane, rx, ry, nkx, nky, fixed)
    return mags, vecs


@pytest.mark.parametrize("case", C.CALC_CASES, ids=[c["name"] for c in C.CALC_CASES])
def test_calculate_matches_reference(case, golden, trajs):
    d = trajs[case["traj"]]
    name = case["name"]
    mags, vecs = _k_for(case, d)
    np.testing.assert_allclose(vecs, golden[f"{name}/k_vecs"], rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(mags, golden[f"{name}/k_mags"], rtol=3e-7, atol=1e-9)
    mags, vecs = golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"]   # inputs as captured
    kw = C.realise_kw(case.get("kw", {}))
    sed, freqs, is_complex = O.calculate(
        d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs,
        use_displacements=case.get("ctor", {}).get("use_displacements", False), **kw)
    ref = golden[f"{name}/sed"]
    assert sed.dtype == ref.dtype and sed.shape == ref.shape
    assert bool(golden[f"{name}/is_complex"]) == is_complex
    np.testing.assert_array_equal(freqs, golden[f"{name}/freqs"])
    # same operations, same dtypes, same BLAS: agreement is at rounding level
    assert rel_max(sed, ref) <= 2e-6
    assert rel_max(O.intensity(sed), golden[f"{name}/intensity"]) <= 2e-6


@pytest.mark.parametrize("case", C.CALC_WIDE_CASES, ids=[c["name"] for c in C.CALC_WIDE_CASES])
def test_calculate_wide_matches_reference(case, golden, trajs):
    """More than 16 k-vectors (the default GPU kernel's territory): the oracle against the
    reference's intensity (whole) and complex rows (every WIDE_SED_STRIDE-th)."""
    d = trajs[case["traj"]]
    name = case["name"]
    mags, vecs = _k_for(case, d)
    np.testing.assert_allclose(vecs, golden[f"{name}/k_vecs"], rtol=3e-7, atol=1e-9)
    vecs = golden[f"{name}/k_vecs"]
    kw = C.realise_kw(case.get("kw", {}))
    sed, _, is_complex = O.calculate(
        d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs,
        use_displacements=case.get("ctor", {}).get("use_displacements", False), **kw)
    assert tuple(golden[f"{name}/sed_shape"]) == sed.shape
    assert bool(golden[f"{name}/is_complex"]) == is_complex
    assert rel_max(sed[::C.WIDE_SED_STRIDE], golden[f"{name}/sed_rows"]) <= 2e-6
    assert rel_max(O.intensity(sed) if is_complex else sed, golden[f"{name}/intensity"]) <= 2e-6


@pytest.mark.parametrize("case", C.CALC_W256_CASES, ids=[c["name"] for c in C.CALC_W256_CASES])
def test_calculate_w256_matches_reference(case, golden, trajs):
    """100 and 250 k-vectors (the 256-row GPU kernel's territory): the oracle against the reference's output."""
    test_calculate_wide_matches_reference(case, golden, trajs)


@pytest.mark.parametrize("case", C.CALC_SYM_CASES, ids=[c["name"] for c in C.CALC_SYM_CASES])
def test_calculate_sym_matches_reference(case, golden, trajs):
    """k-lists with (k, -k) pairs and repeated vectors: the oracle (which, like the reference,
    computes every vector on its own) against the reference's output."""
    d = trajs[case["traj"]]
    name = case["name"]
    mags, vecs = _k_for(case, d)
    np.testing.assert_allclose(vecs, golden[f"{name}/k_vecs"], rtol=3e-7, atol=1e-9)
    vecs = golden[f"{name}/k_vecs"]
    kw = C.realise_kw(case.get("kw", {}))
    sed, _, is_complex = O.calculate(
        d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs,
        use_displacements=case.get("ctor", {}).get("use_displacements", False), **kw)
    assert tuple(golden[f"{name}/sed_shape"]) == sed.shape
    assert bool(golden[f"{name}/is_complex"]) == is_complex
    assert rel_max(sed[::C.WIDE_SED_STRIDE], golden[f"{name}/sed_rows"]) <= 2e-6
    assert rel_max(O.intensity(sed) if is_complex else sed, golden[f"{name}/intensity"]) <= 2e-6


@pytest.mark.parametrize("case", C.CALC_SYM_CASES, ids=[c["name"] for c in C.CALC_SYM_CASES])
def test_reference_output_has_the_minus_k_symmetry(case, golden):
    """The premise of PSA_OPT_FOLD_PAIRS, checked on the REFERENCE's own numbers: wherever -k is in the
    list too, S(-k)[w] == conj S(k)[(T-w) mod T] and I(-k)[w] == I(k)[(T-w) mod T] -- to rounding
    (float32 FFT of conj(q) vs conj of the FFT: <= 1e-6 of the peak), far inside the 1e-5 bar."""
    name = case["name"]
    vecs, inten = golden[f"{name}/k_vecs"], golden[f"{name}/intensity"]
    rows = golden[f"{name}/sed_rows"]
    T = int(golden[f"{name}/sed_shape"][0])
    complex_out = bool(golden[f"{name}/is_complex"])
    assert inten.shape == (T, len(vecs))
    back = (-np.arange(T)) % T
    stored = np.arange(T)[::C.WIDE_SED_STRIDE]                     # frequencies whose complex row is kept
    pairs = 0
    for i, k in enumerate(vecs):
        for j in range(i):
            if np.array_equal(-k, vecs[j]):
                pairs += 1
                assert rel_max(inten[:, i], inten[back, j]) <= 1e-6
                if complex_out:
                    both = [(a, b) for a, w in enumerate(stored) for b, v in enumerate(stored) if v == back[w]]
                    a, b = np.array(both).T
                    assert rel_max(rows[a, i], np.conj(rows[b, j])) <= 1e-6
                break
    assert pairs >= 6


def test_config1_matches_reference():
    """BASELINE configuration 1 at full size (512 x 4096 x 32): the oracle against the real
    reference's output captured by make_golden.py."""
    from conftest import GOLDEN
    spec, req, d = C.c1_inputs()
    with np.load(GOLDEN / "c1_reference.npz") as z:
        ref = {k: z[k] for k in z.files}
    mags, vecs = O.k_path(d["box_matrix"], *spec.cells, req["direction"], req["bz_coverage"], req["n_k"])
    np.testing.assert_allclose(vecs, ref["k_vecs"], rtol=3e-7, atol=1e-9)
    sed, freqs, _ = O.calculate(d["positions"], d["velocities"], d["types"], spec.dt_ps, ref["k_vecs"])
    np.testing.assert_array_equal(freqs, ref["freqs"])
    assert rel_max(O.intensity(sed), ref["intensity"]) <= 2e-6
    assert rel_max(sed[ref["rows"]], ref["sed_rows"]) <= 2e-6
    inc, _, cx = O.calculate(d["positions"], d["velocities"], d["types"], spec.dt_ps, ref["k_vecs"],
                             basis_atom_types=[1, 2], summation_mode="incoherent")
    assert not cx and rel_max(inc, ref["intensity_incoherent_types12"]) <= 2e-6


def test_seam_matches_reference(golden, trajs):
    d = trajs["a"]
    got = O.sed_for_group(d["positions"], d["velocities"], golden["seam/k_vecs"],
                          golden["seam/idx"], golden["seam/mean_pos"])
    assert rel_max(got, golden["seam/sed"]) <= 2e-6
    empty = O.sed_for_group(d["positions"], d["velocities"], golden["seam/k_vecs"],
                            np.array([], int), golden["seam/mean_pos"])
    np.testing.assert_array_equal(empty, golden["seam_empty/sed"])
    np.testing.assert_allclose(O.mean_positions(d["positions"]), golden["seam/mean_pos"], rtol=3e-7)


@pytest.mark.parametrize("i", range(len(C.KPATH_CASES)))
def test_k_path(i, golden, trajs):
    kc = C.KPATH_CASES[i]
    d = trajs[kc["traj"]]
    mags, vecs = O.k_path(d["box_matrix"], *d["cells"], kc["spec"], kc["cov"], kc["n_k"], kc["lat"])
    # np.linspace may differ by 1 ulp between host CPUs
    np.testing.assert_allclose(mags, golden[f"kpath{i}/mags"], rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(vecs, golden[f"kpath{i}/vecs"], rtol=3e-7, atol=1e-9)
    assert mags.dtype == np.float32 and vecs.dtype == np.float32


@pytest.mark.parametrize("i", range(len(C.KGRID_CASES)))
def test_k_grid(i, golden):
    g = C.KGRID_CASES[i]
    mags, vecs, shape = O.k_grid(g["plane"], g["rx"], g["ry"], g["nx"], g["ny"], g["fixed"])
    np.testing.assert_allclose(vecs, golden[f"kgrid{i}/vecs"], rtol=3e-7, atol=1e-9)
    assert mags.size == 0 and tuple(golden[f"kgrid{i}/shape"]) == shape


def test_reciprocal_lattice(golden, trajs):
    for t, d in trajs.items():
        a, b, recip = O.reciprocal_lattice(d["box_matrix"], *d["cells"])
        for i in range(3):
            np.testing.assert_allclose(a[i], golden[f"ctor_{t}/a{i+1}"], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(b[i], golden[f"ctor_{t}/b{i+1}"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(recip, golden[f"ctor_{t}/recip_vecs_prim"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("i", range(len(C.DIRECTION_CASES)))
def test_unit_direction(i, golden):
    got = O.unit_direction(C.DIRECTION_CASES[i])
    np.testing.assert_allclose(got, golden[f"dir{i}"], rtol=3e-7, atol=1e-9)
    assert got.dtype == golden[f"dir{i}"].dtype


@pytest.mark.parametrize("opt", ["C", "A", "B", "Q"])
def test_chiral_phase(opt, golden):
    got = O.chiral_phase(golden["z1"], golden["z2"], opt)
    ref = golden[f"phase_{opt}"]
    assert got.dtype == np.float32 and got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    assert O.chiral_phase(golden["z1"][:0], golden["z2"][:0]).shape == golden["phase_empty"].shape


def test_phase_argument_is_fma_chain():
    """np.dot(k (K,3) f32, r.T) == fma(kz,rz, fma(ky,ry, kx*rx)) bit for bit on this host's
    BLAS -- the premise of the device phase-table kernel (csrc/kernels_misc.hip)."""
    rng = np.random.default_rng(5)
    k = (rng.standard_normal((64, 3)) * 3).astype(np.float32)
    r = (rng.random((1000, 3)) * 90).astype(np.float32)
    dot = np.dot(k, r.T)

    def fma(a, b, c):       # exact product in float64, one rounding to float32
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    kx, ky, kz = (k[:, i:i + 1] for i in range(3))
    rx, ry, rz = (r[:, i][None, :] for i in range(3))
    chain = fma(kz, rz, fma(ky, ry, (kx * rx).astype(np.float32)))
    # double rounding through float64 can differ in ~1e-9 of the cases; allow a handful
    assert np.mean(chain != dot) < 1e-5


def test_known_answer_single_atom():
    """One atom at r=(1,0,0), v_x = cos(2 pi 4 t/32), k=(0.5,0,0): bins 4 and 28 carry
    0.5*exp(+0.5i) (verified against the reference)."""
    T = 32
    pos = np.zeros((T, 1, 3), np.float32)
    pos[:, 0, 0] = 1
    vel = np.zeros((T, 1, 3), np.float32)
    vel[:, 0, 0] = np.cos(2 * np.pi * 4 * np.arange(T) / T)
    sed, freqs, _ = O.calculate(pos, vel, np.ones(1, np.int32), 0.01,
                                np.array([[0.5, 0, 0]], np.float32))
    want = 0.5 * np.exp(0.5j)
    assert abs(sed[4, 0, 0] - want) < 1e-6 and abs(sed[28, 0, 0] - want) < 1e-6
    mask = np.ones(T, bool)
    mask[[4, 28]] = False
    assert np.max(np.abs(sed[mask])) < 1e-6
    assert freqs[4] == pytest.approx(4 / (T * 0.01))


def test_error_behaviour():
    pos = np.zeros((4, 2, 3), np.float32)
    k = np.zeros((1, 3), np.float32)
    with pytest.raises(ValueError, match="summation_mode"):
        O.calculate(pos, pos, np.ones(2, int), 1.0, k, summation_mode="both")
    with pytest.raises(ValueError, match="out of bounds"):
        O.calculate(pos, pos, np.ones(2, int), 1.0, k, basis_atom_indices=[0, 2])
    with pytest.raises(ValueError, match="list of ints"):
        O.calculate(pos, pos, np.ones(2, int), 1.0, k, basis_atom_types=[1, [2]])
    sed, freqs, cx = O.calculate(pos[:0], pos[:0], np.ones(2, int), 1.0, k)
    assert sed.shape == (0, 0, 3) and cx
