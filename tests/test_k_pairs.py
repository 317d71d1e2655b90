"""Pair folding of k-lists (PSA_OPT_FOLD_PAIRS): the host-side detection (psa_k_pairs -- no GPU) and
the host logic around it, driven through the CPU test double."""
import numpy as np

import cases as C
from conftest import make_calculator, rel_max
from oracle import psa_oracle as O
from oracle_engine import OracleEngine
from psa_amd import _hip

MIRROR = 0x80000000


def _expand(kmap, unique, vecs):
    rows = vecs[unique][kmap & 0x7FFFFFFF]
    sign = np.where(kmap >> 31, -1.0, 1.0)[:, None].astype(np.float32)
    return rows * sign


def test_symmetric_grid_folds_to_half():
    calc = make_calculator(C.build_traj("a"))
    _, vecs, _ = calc.get_k_grid("xy", (-3.5, 3.5), (-3.5, 3.5), 50, 50, 0.0)      # BASELINE configuration 4
    kmap, unique = _hip.k_pairs(vecs)
    assert len(unique) == 1250 and np.array_equal(unique, np.arange(1250))
    # partner of point i is point K-1-i, mirrored
    assert np.array_equal(kmap[1250:], (np.arange(1249, -1, -1) | MIRROR).astype(np.uint32))
    np.testing.assert_array_equal(_expand(kmap, unique, vecs), vecs)


def test_gamma_duplicates_signed_zero_and_nan():
    vecs = np.array([[0, 0, 0], [1, 2, 3], [-1, -2, -3], [1, 2, 3], [-0.0, 0.0, -0.0], [1, 2, -3],
                     [np.nan, 0, 0], [np.nan, 0, 0], [-1, -2, 3], [0.5, 0, 0]], np.float32)
    kmap, unique = _hip.k_pairs(vecs)
    assert unique.tolist() == [0, 1, 5, 6, 7, 9]                     # NaN never matches, not even itself
    assert kmap.tolist() == [0, 1, 1 | MIRROR, 1, 0, 2, 3, 4, 2 | MIRROR, 5]
    empty_map, empty_unique = _hip.k_pairs(np.zeros((0, 3), np.float32))
    assert len(empty_map) == 0 and len(empty_unique) == 0


def test_path_and_offset_grid_do_not_fold():
    calc = make_calculator(C.build_traj("a"))
    _, path = calc.get_k_path([1, 1, 0], 2.0, 40)
    _, grid, _ = calc.get_k_grid("xy", (-1.5, 1.5), (-1.0, 1.0), 6, 7, 0.25)
    for vecs in (path, grid):
        kmap, unique = _hip.k_pairs(vecs)
        assert len(unique) == len(vecs) and np.array_equal(kmap, np.arange(len(vecs)))


def test_mirror_rule_on_the_oracle(trajs):
    """S(-k)[w] = conj S(k)[(T-w) mod T], the rule the epilogue applies, against the oracle computing
    both vectors (T = 100: not a power of two)."""
    d = trajs["c"]
    calc = make_calculator(d)
    _, vecs, _ = calc.get_k_grid("yz", (-2.0, 2.0), (-1.0, 1.0), 4, 6, 0.0)
    full, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs)
    kmap, unique = _hip.k_pairs(vecs)
    assert len(unique) == 12
    part, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs[unique])
    T = full.shape[0]
    back = (-np.arange(T)) % T
    rebuilt = part[:, kmap & 0x7FFFFFFF, :]
    flip = (kmap >> 31).astype(bool)
    rebuilt[:, flip, :] = np.conj(rebuilt[back][:, flip, :])
    assert rel_max(rebuilt, full) <= 1e-6


def test_intensity_comes_with_the_result_once(trajs):
    """`SED.intensity` of a fresh complex result is the array the engine delivered with it -- handed
    out once, and only while `sed.sed` is the untouched array that was returned."""
    d = trajs["a"]
    calc = make_calculator(d).attach(engine=OracleEngine())
    mags, vecs = calc.get_k_path("100", 1.0, 8)
    sed = calc.calculate(mags, vecs)
    snap = sed._intensity_snapshot[0]
    ref = np.sum(np.abs(sed.sed) ** 2, axis=-1).astype(np.float32)
    first = sed.intensity
    assert first is snap and rel_max(first, ref) <= 1e-6
    second = sed.intensity                                        # NumPy expression, a fresh array
    assert second is not first and np.array_equal(second, ref)
    # an edited or replaced result is never served from the snapshot
    sed = calc.calculate(mags, vecs)
    sed.sed *= 2
    assert rel_max(sed.intensity, 4 * ref) <= 1e-6
    sed = calc.calculate(mags, vecs)
    sed.sed = sed.sed[:, :4].copy()
    assert sed.intensity.shape == (sed.sed.shape[0], 4)
    # an incoherent result has no companion array; its `.intensity` keeps the reference's 2-D quirk
    inc = calc.calculate(mags, vecs, basis_atom_types=[1, 2], summation_mode="incoherent")
    assert not hasattr(inc, "_intensity_snapshot")
    assert inc.intensity.shape == (inc.sed.shape[0],)
    import copy
    import pickle
    sed = calc.calculate(mags, vecs)
    assert not hasattr(pickle.loads(pickle.dumps(sed)), "_intensity_snapshot")
    assert not hasattr(copy.deepcopy(sed), "_intensity_snapshot")


def test_host_mean_is_numpy_mean_bit_for_bit(tmp_path):
    """psa_host_mean_frames (the mean of a positions array that stays on the host, velocity mode) against
    np.mean(axis=0, dtype=float32) -- reference sed_calculator.py:205 -- for every split of the columns
    over threads, odd shapes, a memory-mapped file; and the calculator takes it for large arrays."""
    rng = np.random.default_rng(5)
    for shape in ((1, 5, 3), (7, 1, 3), (3000, 40, 3), (257, 1001, 3)):
        x = (50 + 3 * rng.standard_normal(shape)).astype(np.float32)
        for threads in (0, 1, 3, 16):
            np.testing.assert_array_equal(_hip.host_mean_frames(x, threads), np.mean(x, axis=0, dtype=np.float32))
    big = (20 + rng.standard_normal((1400, 4096, 3))).astype(np.float32)            # 69 MB: above the calculator's threshold
    path = tmp_path / "pos.npy"
    np.save(path, big)
    mapped = np.load(path, mmap_mode="r")
    want = np.mean(big, axis=0, dtype=np.float32)
    np.testing.assert_array_equal(_hip.host_mean_frames(mapped), want)
    with np.testing.assert_raises(ValueError):
        _hip.host_mean_frames(big.astype(np.float64))
    with np.testing.assert_raises(ValueError):
        _hip.host_mean_frames(big[:, ::2])
    from psa_amd import SEDCalculator, Trajectory
    n = big.shape[1]
    box = np.diag([10.0, 10.0, 10.0]).astype(np.float32)
    tr = Trajectory(mapped, big, np.ones(n, np.int32),
                    np.arange(big.shape[0], dtype=np.float32), box, np.diag(box).copy(), np.zeros(3, np.float32), 0.001)
    calc = SEDCalculator(tr, 1, 1, 1)
    calls = []
    real = _hip.host_mean_frames
    try:
        _hip.host_mean_frames = lambda x, threads=0: calls.append(x.shape) or real(x, threads)
        np.testing.assert_array_equal(calc._mean_positions(), want)
    finally:
        _hip.host_mean_frames = real
    assert calls == [big.shape]
