"""GPU tests of the round-2 machinery around the projection: cached split planes, frame-range
launches, the overlapped upload, the single (k, omega) bin, the frame-sharding entry points and
the residency rules.  Everything goes through the C ABI (psa_amd._hip) and is compared with the
CPU oracle or with the reference's golden output."""
import threading
import time

import numpy as np
import pytest

import cases as C
from conftest import make_calculator, rel_max
from oracle import psa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _random_traj(n_atoms, n_frames, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    pos = (rng.uniform(0, 20, (1, n_atoms, 3)) + 0.05 * rng.standard_normal((n_frames, n_atoms, 3))).astype(np.float32)
    vel = (scale * rng.standard_normal((n_frames, n_atoms, 3))).astype(np.float32)
    return pos, vel


def _kvecs(n_k, seed=1):
    rng = np.random.default_rng(seed)
    return rng.uniform(-2, 2, (n_k, 3)).astype(np.float32)


@pytest.fixture()
def fresh_engine():
    from psa_amd import _hip
    eng = _hip.Engine(0)
    yield eng
    eng.close()


# ------------------------------------------------------------------ split planes
@pytest.mark.parametrize("n_atoms, n_frames, n_k", [(64, 128, 24), (200, 100, 40), (333, 77, 140), (1000, 50, 17),
                                                    (96, 16, 300), (5, 3, 33)])
def test_planes_kernel_shapes_against_oracle(fresh_engine, n_atoms, n_frames, n_k):
    """Ragged atoms / frames / k-lists through split_planes_kernel + k1_planes_kernel (all three
    block heights), all atoms and an index list with duplicates (compacted by the split)."""
    from psa_amd import _hip
    eng = fresh_engine
    eng.set_option(_hip.OPT_PLANES_EAGER, 1)
    pos, vel = _random_traj(n_atoms, n_frames, seed=n_atoms)
    mean = O.mean_positions(pos)
    kv = _kvecs(n_k)
    eng.ensure_resident(0, vel)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, n_atoms, max(1, n_atoms // 2)).astype(np.int32)
    for sel in (None, idx):
        got = eng.debug_project_only(0, mean, kv, sel)
        atoms = np.arange(n_atoms) if sel is None else sel
        ref = O.project_group(vel[:, atoms, :], O.phase_table(kv, mean[atoms]))
        assert rel_max(got.transpose(2, 0, 1), ref) < 2e-6
    sets, nbytes = eng.plane_cache()
    assert sets == 2 and nbytes > 0


def test_plane_cache_policy(fresh_engine, trajs):
    """all atoms: planes on first use; an index list: on its second use (or the first with
    PSA_OPT_PLANES_EAGER); k-lists under PSA_OPT_PLANES_MIN_K never; a new upload drops the sets;
    PSA_OPT_PLANES = 0 empties the cache; a budget too small for a set falls back to the on-the-fly
    kernels -- and every path gives the same numbers."""
    from psa_amd import _hip
    eng = fresh_engine
    d = trajs["a"]
    vel, mean = d["velocities"], O.mean_positions(d["positions"])
    kv = _kvecs(40)
    idx = np.array([7, 3, 3, 60, 12, 12, 41, 0, 63, 5, 18], np.int32)
    ref_all = O.project_group(vel, O.phase_table(kv, mean))
    ref_idx = O.project_group(vel[:, idx, :], O.phase_table(kv, mean[idx]))
    eng.ensure_resident(0, vel)
    assert eng.plane_cache()[0] == 0
    s1 = eng.debug_project_only(0, mean, kv[:8])                 # short k-list: bf16 kernel, no planes built
    assert eng.plane_cache()[0] == 0
    a1 = eng.debug_project_only(0, mean, kv)
    assert eng.plane_cache()[0] == 1
    s2 = eng.debug_project_only(0, mean, kv[:8])                 # ... but used once they exist (32-row variant)
    assert eng.plane_cache()[0] == 1 and rel_max(s2, s1) < 2e-6 and not np.array_equal(s1, s2)
    i1 = eng.debug_project_only(0, mean, kv, idx)                # first sight of the list: gather kernel
    assert eng.plane_cache()[0] == 1
    i2 = eng.debug_project_only(0, mean, kv, idx)                # second: compacted planes
    assert eng.plane_cache()[0] == 2
    i3 = eng.debug_project_only(0, mean, kv, idx[::-1].copy())   # another list (same atoms, other order)
    assert eng.plane_cache()[0] == 2
    for got, ref in ((a1, ref_all), (i1, ref_idx), (i2, ref_idx), (i3, ref_idx[:, :, :])):
        assert rel_max(got.transpose(2, 0, 1), ref) < 2e-6
    np.testing.assert_allclose(i1, i2, rtol=0, atol=2e-6 * np.abs(i1).max())
    eng.ensure_resident(0, np.array(vel))                        # a new array in the slot
    assert eng.plane_cache()[0] == 0
    eng.set_option(_hip.OPT_PLANES_BUDGET, 4096)                 # nothing fits
    b1 = eng.debug_project_only(0, mean, kv)
    assert eng.plane_cache()[0] == 0 and rel_max(b1.transpose(2, 0, 1), ref_all) < 2e-6
    eng.set_option(_hip.OPT_PLANES_BUDGET, 0)
    eng.debug_project_only(0, mean, kv)
    assert eng.plane_cache()[0] == 1
    eng.set_option(_hip.OPT_PLANES, 0)
    assert eng.plane_cache()[0] == 0
    b2 = eng.debug_project_only(0, mean, kv)
    assert eng.plane_cache()[0] == 0 and rel_max(b2.transpose(2, 0, 1), ref_all) < 2e-6
    eng.set_option(_hip.OPT_PLANES, 1)
    eng.set_option(_hip.OPT_PLANES_MIN_K, 4)
    b3 = eng.debug_project_only(0, mean, kv[:8])                 # now the 32-row planes variant
    assert eng.plane_cache()[0] == 1
    assert rel_max(b3.transpose(2, 0, 1), O.project_group(vel, O.phase_table(kv[:8], mean))) < 2e-6
    with pytest.raises(_hip.PsaHipError):
        eng.set_option(99, 1)


def test_plane_cache_evicts_least_recently_used(fresh_engine):
    from psa_amd import _hip
    eng = fresh_engine
    eng.set_option(_hip.OPT_PLANES_EAGER, 1)
    pos, vel = _random_traj(256, 64, seed=3)
    mean = O.mean_positions(pos)
    kv = _kvecs(24)
    eng.ensure_resident(0, vel)
    eng.debug_project_only(0, mean, kv)
    one = eng.plane_cache()[1]                                   # bytes of the all-atoms set
    lists = [np.arange(0, 256, 2, dtype=np.int32), np.arange(1, 256, 2, dtype=np.int32), np.arange(64, dtype=np.int32)]
    eng.set_option(_hip.OPT_PLANES_BUDGET, int(2.2 * one))       # room for the big set + two half-size ones
    for sel in lists:
        got = eng.debug_project_only(0, mean, kv, sel)
        assert rel_max(got.transpose(2, 0, 1), O.project_group(vel[:, sel, :], O.phase_table(kv, mean[sel]))) < 2e-6
        assert eng.plane_cache()[1] <= 2.2 * one
    assert 1 <= eng.plane_cache()[0] <= 3


# ------------------------------------------------------------------ frame ranges
@pytest.mark.parametrize("n_k, idx", [(40, None), (8, None), (40, [5, 9, 9, 77, 3]), (24, None)])
def test_two_half_trajectory_projections_equal_the_whole(fresh_engine, n_k, idx):
    """Frame sharding / the streaming upload rest on this: projecting frames [0, h) and [h, T) into
    the columns of one slab is the projection of the whole."""
    from psa_amd import _hip
    eng = fresh_engine
    eng.set_option(_hip.OPT_PLANES_EAGER, 1)                     # the same kernel serves every call below
    pos, vel = _random_traj(160, 192, seed=8)
    mean, kv = O.mean_positions(pos), _kvecs(n_k)
    eng.ensure_resident(0, vel)
    sel = None if idx is None else np.asarray(idx, np.int32)
    whole = eng.debug_project_only(0, mean, kv, sel)
    for h in (64, 80, 16):
        lo = eng.debug_project_only(0, mean, kv, sel, frames=(0, h))
        hi = eng.debug_project_only(0, mean, kv, sel, frames=(h, 192 - h))
        assert not lo[:, :, h:].any() and not hi[:, :, :h].any()
        np.testing.assert_array_equal(lo + hi, whole)            # bit-identical: same per-frame arithmetic
    assert not eng.debug_project_only(0, mean, kv, sel, frames=(7, 0)).any()


# ------------------------------------------------------------------ overlapped upload
def test_first_call_streams_the_upload_and_matches_reference(fresh_engine, golden, trajs):
    """`calculate` on an array that is not resident goes through psa_sed_project_upload (chunks
    projected behind their copies); the second call runs on the resident array (planes).  Both
    reproduce the reference's golden output."""
    eng = fresh_engine
    for name in ("w_coh_all_k40", "w_inc_types12_k24", "w_idx_dup_k40", "coh_all", "inc_nested_types",
                 "w_displacements_k40"):
        case = next(c for c in C.CALC_CASES + C.CALC_WIDE_CASES if c["name"] == name)
        calc = make_calculator(trajs[case["traj"]], **case.get("ctor", {})).attach(engine=eng)
        kw = C.realise_kw(case.get("kw", {}))
        eng.invalidate()
        for _ in range(2):
            sed = calc.calculate(golden[f"{name}/k_mags"], golden[f"{name}/k_vecs"], **kw)
            want = golden[f"{name}/intensity"]
            if not name.startswith("w_") and not sed.is_complex:
                want = golden[f"{name}/sed"]
            assert rel_max(sed.intensity if sed.is_complex else sed.sed, want) <= TOL


def test_streamed_upload_many_chunks(fresh_engine, monkeypatch):
    """Chunks of 64 frames (PSA_UPLOAD_CHUNK_MIB floor): 9 chunks + a ragged tail, two groups."""
    from psa_amd import SEDCalculator, Trajectory, _hip
    monkeypatch.setenv("PSA_UPLOAD_CHUNK_MIB", "1")
    eng = fresh_engine
    n_atoms, n_frames = 2000, 601                                 # 24 KB per frame: 64-frame chunks = 1.5 MiB
    pos, vel = _random_traj(n_atoms, n_frames, seed=4)
    types = np.where(np.arange(n_atoms) % 3 == 0, 1, 2).astype(np.int32)
    box = np.diag([20.0, 20.0, 20.0]).astype(np.float32)
    tr = Trajectory(pos, vel, types, np.arange(n_frames, dtype=np.float32), box, np.diag(box).copy(),
                    np.zeros(3, np.float32), 0.002)
    calc = SEDCalculator(tr, 2, 2, 2).attach(engine=eng)
    kv = _kvecs(20)
    for kw in (dict(), dict(basis_atom_types=[1, 2], summation_mode="incoherent")):
        eng.invalidate()
        sed = calc.calculate(np.zeros(20, np.float32), kv, **kw)
        ref, _, _ = O.calculate(pos, vel, types, 0.002, kv, **kw)
        assert rel_max(sed.sed, ref) <= TOL
        assert eng.timings()["h2d"] > 0
    np.testing.assert_array_equal(eng.download(0, 590, 11), vel[590:])     # every frame landed


def test_first_call_on_a_memory_mapped_cache(fresh_engine, tmp_path):
    """Configuration-2-sized .npy cache (8192 atoms x 16384 frames = 1.6 GB), memory-mapped like the
    reference's loader leaves it (io/loader.py:48-79): the first calculate() -- upload with each
    chunk's frames projected behind its copy, FFT plan, FFT, epilogue, D2H -- gives the result of the
    resident path, and every frame lands.  (How long it takes against the upload alone is a
    measurement, not a gate: `first_call` in bench.py's line.)"""
    from psa_amd import SEDCalculator, Trajectory, _hip, synth
    spec, req = synth.baseline_spec("C2")
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    path = tmp_path / "run.velocities.npy"
    out = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(spec.n_frames, spec.n_atoms, 3))
    for t in range(0, spec.n_frames, 512):
        out[t:t + 512] = synth.velocities_block(spec, tables, t, 512)
    out.flush()
    del out
    vel = np.load(path, mmap_mode="r")
    pos = np.broadcast_to(r0, vel.shape)
    eng = fresh_engine
    tr = Trajectory(pos, vel, types, np.arange(spec.n_frames, dtype=np.float32), box, np.diag(box).copy(),
                    np.zeros(3, np.float32), spec.dt_ps)
    calc = SEDCalculator(tr, *spec.cells).attach(engine=eng)
    mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
    streamed = []
    for _ in range(2):
        eng.invalidate()
        sed = calc.calculate(mags, vecs)                          # not resident: upload + projection, overlapped
        assert eng.timings()["h2d"] > 0
        streamed.append((sed.sed[::1024].copy(), sed.intensity[::1024].copy()))
    again = calc.calculate(mags, vecs)                            # resident now: planes kernel
    assert eng.timings()["h2d"] < 1.0
    for rows, inten in streamed:
        assert rel_max(again.sed[::1024], rows) <= TOL
        assert rel_max(again.intensity[::1024], inten) <= TOL
    block = synth.velocities_block(spec, tables, 0, 256)
    np.testing.assert_array_equal(eng.download(0, 0, 256), block)
    np.testing.assert_array_equal(eng.download(0, spec.n_frames - 256, 256), synth.velocities_block(spec, tables, spec.n_frames - 256, 256))


@pytest.mark.parametrize("n_atoms, n_frames, n_k, idx", [
    (77, 150, 37, None), (33, 64, 33, None), (200, 65, 97, "list"), (64, 1000, 129, None), (1000, 96, 300, "list"),
    (77, 150, 128, None), (700, 130, 250, "list"), (2000, 64, 65, None)])
def test_ragged_shapes_through_every_form_of_the_planes_kernel(fresh_engine, n_atoms, n_frames, n_k, idx):
    """Atom counts that are no multiple of 32 (stage counts that are no multiple of the wide kernel's 20-stage
    period), frame counts that are no multiple of 64 (or of 16), k-lists that fill their last M block partly,
    index lists with duplicates: the 256-row form (default where the list fills an even number of 128-row
    blocks), the loader-wavefront form and the eight-wavefront form of the planes kernel against the oracle,
    before the FFT."""
    from psa_amd import _hip
    eng = fresh_engine
    pos, vel = _random_traj(n_atoms, n_frames, seed=n_atoms + n_k)
    mean, kv = O.mean_positions(pos), _kvecs(n_k, seed=n_k)
    members = None
    if idx:
        rng = np.random.default_rng(n_k)
        members = np.concatenate([rng.permutation(n_atoms)[: n_atoms // 2], [3, 3, n_atoms - 1]]).astype(np.int32)
    eng.set_option(_hip.OPT_PLANES_EAGER, 1)
    eng.ensure_resident(0, vel)
    sel = np.arange(n_atoms) if members is None else members
    ref = O.project_group(vel[:, sel, :], O.phase_table(kv, mean[sel]))            # (T, K, 3)
    got = {}
    for form in ((1, 1), (0, 1), (0, 0)):
        eng.set_option(_hip.OPT_K1_WIDE, form[0])
        eng.set_option(_hip.OPT_K1_LOADER_WAVES, form[1])
        eng.debug_project_only(0, mean, kv, members)                               # (builds the planes on first use)
        got[form] = eng.debug_project_only(0, mean, kv, members).transpose(2, 0, 1)
        assert rel_max(got[form], ref) < 2e-6, form
    assert eng.plane_cache()[0] == 1
    assert rel_max(got[0, 1], got[0, 0]) < 1e-6 and rel_max(got[1, 1], got[0, 1]) < 1e-6


# ------------------------------------------------------------------ residency rules
def test_in_place_edits_are_noticed_and_invalidate_is_explicit(fresh_engine, trajs):
    eng = fresh_engine
    d = dict(trajs["a"])
    d["velocities"] = np.array(d["velocities"])
    d["positions"] = np.array(d["positions"])
    calc = make_calculator(d).attach(engine=eng)
    mags, vecs = calc.get_k_path("100", 1.0, 24)
    first = calc.calculate(mags, vecs).intensity
    d["velocities"] *= 2.0                                        # in place: same object, same buffer
    doubled = calc.calculate(mags, vecs).intensity
    np.testing.assert_allclose(doubled, 4.0 * first, rtol=1e-5)
    d["velocities"][5, 7, 1] += 100.0                             # one element: the sample may miss it ...
    calc.invalidate()                                             # ... the explicit way never does
    ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs)
    assert rel_max(calc.calculate(mags, vecs).intensity, O.intensity(ref)) <= TOL
    # a float64, Fortran-ordered trajectory is converted once and stays resident
    d64 = dict(d, velocities=np.asfortranarray(d["velocities"].astype(np.float64)))
    calc64 = make_calculator(d64).attach(engine=eng)
    assert rel_max(calc64.calculate(mags, vecs).intensity, O.intensity(ref)) <= TOL
    assert eng.is_resident(0, d64["velocities"])
    eng.timings()
    calc64.calculate(mags, vecs)
    assert eng.timings()["h2d"] < 0.5                             # no second upload of the array


def test_result_buffers_are_checked_by_the_library(fresh_engine, trajs):
    """A finalize / result_* call whose buffer does not match the result resident on the device is
    refused (PSA_EINVAL) instead of overrunning it."""
    from psa_amd import _hip
    eng = fresh_engine
    d = trajs["a"]
    eng.ensure_resident(0, d["velocities"])
    mean = O.mean_positions(d["positions"])
    eng.project(0, mean, _kvecs(12))
    with pytest.raises(_hip.PsaHipError, match="the caller's buffer"):
        eng.finalize(128, 11, False)
    with pytest.raises(_hip.PsaHipError, match="the caller's buffer"):
        eng.finalize(128, 12, True)
    out = eng.finalize(128, 12, False)
    assert out.shape == (128, 12, 3)
    with pytest.raises(_hip.PsaHipError, match="the caller's buffer"):
        eng.result_intensity(128, 13)
    with pytest.raises(_hip.PsaHipError, match="the caller's buffer"):
        eng.result_chiral_phase(64, 12, 0, 1)
    np.testing.assert_allclose(eng.result_intensity(128, 12), np.sum(np.abs(out) ** 2, axis=-1), rtol=3e-6)


def test_composites_hold_the_engine_across_calculate_and_phase(fresh_engine, trajs):
    """Two calculators with different (T, K) on ONE engine, chiral composites from two threads:
    the phase of each result is computed from that result."""
    eng = fresh_engine
    ca = make_calculator(trajs["a"]).attach(engine=eng)
    cb = make_calculator(trajs["c"]).attach(engine=eng)
    want_a = ca.calculate_kpath_sed("100", 1.0, 24, chiral=True)
    want_b = cb.calculate_kpath_sed("z", 2.0, 9, chiral=True)
    errors = []

    def work(calc, args, want):
        try:
            for _ in range(6):
                got = calc.calculate_kpath_sed(*args, chiral=True)
                assert got.phase.shape == want.phase.shape
                np.testing.assert_allclose(got.phase, want.phase, atol=1e-5)
        except Exception as e:                                    # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(ca, ("100", 1.0, 24), want_a)),
               threading.Thread(target=work, args=(cb, ("z", 2.0, 9), want_b))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# ------------------------------------------------------------------ single (k, omega) bin
@pytest.mark.parametrize("disp", [False, True])
def test_single_bin_is_the_bin_of_the_full_spectrum(fresh_engine, trajs, disp):
    from psa_amd import _hip
    eng = fresh_engine
    d = trajs["b"]
    calc = make_calculator(d, use_displacements=disp).attach(engine=eng)
    mags, vecs = calc.get_k_path([1, 1, 0], 2.0, 9)
    mean = calc._mean_positions()
    for members in (np.arange(d["positions"].shape[1]), np.array([0, 1, 2, 40, 40, 7])):
        full = calc.calculate(mags, vecs, basis_atom_indices=members).sed
        ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs,
                                basis_atom_indices=members, use_displacements=disp)
        for i_k, i_w in ((4, 3), (8, 63), (0, 0), (2, 17)):
            got = calc._single_bin(vecs[i_k], members, i_w, mean)
            assert got.shape == (3,) and got.dtype == np.complex64
            scale = np.abs(ref[:, i_k, :]).max()
            assert np.abs(got - ref[i_w, i_k, :]).max() <= 2e-6 * scale
            assert np.abs(got - full[i_w, i_k, :]).max() <= 2e-6 * scale
    with pytest.raises((ValueError, _hip.PsaHipError)):
        eng.single_bin(0, mean, vecs[0], None, 10 ** 6)


# ------------------------------------------------------------------ frame sharding entry points
@pytest.mark.parametrize("intensity", [False, True])
def test_frame_sharded_steps_on_one_gpu(intensity, trajs):
    """Two contexts on one GPU stand for two ranks: each holds half of the frames, projects all k
    on them (psa_sed_fs_project), the blocks are traded through the host (psa_sed_fs_read/_write,
    what KShardGroup does when RCCL cannot be formed), each finishes its own k rows
    (psa_sed_fs_finish), and the slab rows put together equal the unsharded calculation."""
    from psa_amd import _hip, dist
    d = trajs["a"]
    vel, mean = d["velocities"], O.mean_positions(d["positions"])
    T, K = vel.shape[0], 40
    kv = _kvecs(K)
    groups = [np.flatnonzero(d["types"] == 1), np.flatnonzero(d["types"] != 1)] if intensity else [None]
    flags = _hip.F_INTENSITY if intensity else 0
    t_off, t_cnt = dist.frame_ranges(T, 2)
    k_off, k_cnt = dist.shard_ranges(K, 2)
    engines = [_hip.Engine(0), _hip.Engine(0)]
    try:
        for r, eng in enumerate(engines):
            eng.ensure_resident(0, np.ascontiguousarray(vel[t_off[r]:t_off[r] + t_cnt[r]]))
        for gi, members in enumerate(groups):
            for r, eng in enumerate(engines):
                eng.fs_project(0, mean, kv, members, flags, T, int(k_off[r]), int(k_cnt[r]))
            parts = [eng.fs_read(0, K, int(t_cnt[r])) for r, eng in enumerate(engines)]
            for r, eng in enumerate(engines):
                for src in range(2):
                    eng.fs_write(int(t_off[src]), parts[src][k_off[r]:k_off[r] + k_cnt[r]])
                eng.fs_finish(gi == 0)
        rows = [eng.slab_read(int(k_off[r]), int(k_cnt[r]), T, intensity) for r, eng in enumerate(engines)]
        slab = np.concatenate(rows)
    finally:
        for eng in engines:
            eng.close()
    if intensity:
        ref, _, _ = O.calculate(d["positions"], vel, d["types"], d["dt_ps"], kv, basis_atom_types=[[1], [2, 3]],
                                summation_mode="incoherent")
        assert rel_max(slab.T, ref) <= TOL
    else:
        ref, _, _ = O.calculate(d["positions"], vel, d["types"], d["dt_ps"], kv)
        # the slab holds FFT(q) (the 1/T is applied by the finalize transpose)
        assert rel_max(slab.transpose(2, 0, 1) / T, ref) <= TOL


def test_frame_sharded_single_rank_group(fresh_engine, trajs):
    """KShardGroup(mode="frames") with one rank degenerates to the plain path; with a one-rank RCCL
    communicator psa_sed_fs_exchange places the rank's own block."""
    from psa_amd import _hip, dist
    eng = fresh_engine
    d = trajs["a"]
    vel, mean = d["velocities"], O.mean_positions(d["positions"])
    kv = _kvecs(33)
    eng.comm_init(eng.new_unique_id(), 0, 1)
    try:
        eng.comm_selftest()
        eng.ensure_resident(0, vel)
        eng.fs_project(0, mean, kv, None, 0, vel.shape[0], 0, 33)
        eng.fs_exchange([0], [vel.shape[0]], [0], [33])
        eng.fs_finish(True)
        got = eng.finalize(vel.shape[0], 33, False)
    finally:
        eng.comm_destroy()
    ref, _, _ = O.calculate(d["positions"], vel, d["types"], d["dt_ps"], kv)
    assert rel_max(got, ref) <= TOL
    group = dist.KShardGroup(eng, dist.Exchange(), mode="frames")
    calc = make_calculator(d).attach(shard_group=group)
    assert rel_max(calc.calculate(np.zeros(33, np.float32), kv).sed, ref) <= TOL


# ------------------------------------------------------------------ result leaving block by block
@pytest.mark.parametrize("n_k, idx, disp", [(200, None, False), (333, None, False), (192, [3, 9, 9, 60, 1, 17, 33, 2], False),
                                            (260, None, True)])
def test_long_complex_results_are_pipelined_and_equal_the_plain_path(fresh_engine, trajs, n_k, idx, disp):
    """psa_sed_calculate produces a complex result of >= 192 k-vectors in blocks (project, FFT,
    transpose, 2-D D2H on a copy stream while the next block is projected): same numbers as
    project + finalize, and as the oracle."""
    from psa_amd import _hip
    eng = fresh_engine
    d = trajs["a"]
    src = d["positions"] if disp else d["velocities"]
    slot, flags = (1, _hip.F_DISPLACEMENTS) if disp else (0, 0)
    mean = O.mean_positions(d["positions"])
    kv = _kvecs(n_k, seed=n_k)
    groups = None if idx is None else [np.asarray(idx)]
    eng.ensure_resident(slot, src)
    for _ in range(2):                                            # on-the-fly kernels first, planes second
        piped = eng.calculate(slot, mean, kv, groups, flags)
        eng.project(slot, mean, kv, groups, flags)
        plain = eng.finalize(src.shape[0], n_k, False)
        assert piped.shape == plain.shape == (src.shape[0], n_k, 3)
        assert rel_max(piped, plain) < 2e-6
        # the result also stays on the device, whole
        np.testing.assert_allclose(eng.result_intensity(src.shape[0], n_k), np.sum(np.abs(plain) ** 2, axis=-1), rtol=3e-6)
    ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], kv,
                            basis_atom_indices=None if idx is None else list(idx), use_displacements=disp)
    assert rel_max(piped, ref) <= TOL


def test_intensity_comes_with_the_result_and_the_device_reread_is_opt_in(fresh_engine, trajs, monkeypatch):
    """`SED.intensity` of a fresh complex result is the (T,K) array the device produced in the pass that
    wrote the result: no host reduction runs (np.sum is made to fail while it is read).  It is handed
    out once; later accesses are the reference's NumPy expression -- or, with
    psa_amd.fast_intensity(True), a re-read of the intensity still resident on the device, guarded
    against other calculations and in-place edits."""
    import copy
    import pickle
    import psa_amd
    eng = fresh_engine
    calc = make_calculator(trajs["a"]).attach(engine=eng)
    mags, vecs = calc.get_k_path("100", 1.0, 24)
    sed = calc.calculate(mags, vecs)
    want = np.sum(np.abs(sed.sed) ** 2, axis=-1).astype(np.float32)
    companion = sed._intensity_snapshot[0]
    with monkeypatch.context() as m:
        m.setattr(np, "sum", lambda *a, **k: (_ for _ in ()).throw(AssertionError("host reduction ran")))
        first = sed.intensity
    assert first is companion and first.dtype == np.float32 and first.shape == want.shape
    np.testing.assert_allclose(first, want, rtol=5e-6)
    second = sed.intensity                                                   # handed out once: NumPy now, bit for bit
    assert second is not first
    np.testing.assert_array_equal(second, want)
    assert not hasattr(sed, "_device_intensity")
    # the streamed first call (array not resident yet) and the pipelined call deliver it as well
    eng.invalidate()
    for _ in range(2):
        sed = calc.calculate(mags, vecs)
        np.testing.assert_allclose(sed._intensity_snapshot[0], np.sum(np.abs(sed.sed) ** 2, axis=-1), rtol=5e-6)
    # an edited or replaced result never gets the companion
    sed.sed[:] *= 2
    np.testing.assert_array_equal(sed.intensity, np.sum(np.abs(sed.sed) ** 2, axis=-1).astype(np.float32))
    psa_amd.fast_intensity(True)
    try:
        sed = calc.calculate(mags, vecs)                                      # (the switch is read at calculation time)
        want = np.sum(np.abs(sed.sed) ** 2, axis=-1).astype(np.float32)
        first = sed.intensity
        again = sed.intensity                                                 # re-read from the device
        assert again is not first
        np.testing.assert_array_equal(again, first)
        np.testing.assert_allclose(again, want, rtol=5e-6)
        assert pickle.loads(pickle.dumps(sed)).intensity.shape == want.shape and copy.deepcopy(sed) is not None
        other = calc.calculate(mags, vecs[:20])                               # another result on the engine
        np.testing.assert_array_equal(sed.intensity, want)                    # stale hook -> NumPy
        np.testing.assert_allclose(other.intensity, np.sum(np.abs(other.sed) ** 2, axis=-1), rtol=5e-6)
        other.sed[:] *= 2                                                     # edited in place -> NumPy again
        np.testing.assert_array_equal(other.intensity, np.sum(np.abs(other.sed) ** 2, axis=-1).astype(np.float32))
    finally:
        psa_amd.fast_intensity(False)


def test_very_long_k_lists_are_projected_in_blocks(fresh_engine, monkeypatch):
    """The phase table (8 bytes per k-vector and atom) is bounded: a k-list whose table would exceed
    the limit (2 GiB; 1 MiB here) is projected in blocks of k-vectors -- same result."""
    from psa_amd import _hip
    eng = fresh_engine
    pos, vel = _random_traj(1000, 96, seed=6)
    mean, kv = O.mean_positions(pos), _kvecs(300, seed=2)
    types = np.where(np.arange(1000) < 400, 1, 2)
    eng.ensure_resident(0, vel)
    eng.project(0, mean, kv)
    whole = eng.finalize(96, 300, False)
    groups = [np.flatnonzero(types == 1), np.flatnonzero(types == 2)]
    eng.project(0, mean, kv, groups, _hip.F_INTENSITY)
    whole_inc = eng.finalize(96, 300, True)
    monkeypatch.setenv("PSA_PHASE_TABLE_MIB", "1")               # 1024 atoms x 8 B -> blocks of 128 k-vectors
    eng.project(0, mean, kv)
    # (300 k-vectors at once run in 128-row blocks, blocks of 128 in one 256-row block: the float32 folds differ)
    assert rel_max(eng.finalize(96, 300, False), whole) <= 2e-6
    eng.project(0, mean, kv, groups, _hip.F_INTENSITY)
    np.testing.assert_allclose(eng.finalize(96, 300, True), whole_inc, rtol=1e-6)
    ref, _, _ = O.calculate(pos, vel, types, 0.002, kv)
    assert rel_max(whole, ref) <= TOL


def test_upload_without_page_locked_staging(fresh_engine, trajs, monkeypatch):
    """If no page-locked memory can be had the upload falls back to plain copies; the streamed
    first call still projects chunk by chunk."""
    monkeypatch.setenv("PSA_UPLOAD_NO_STAGING", "1")
    monkeypatch.setenv("PSA_UPLOAD_CHUNK_MIB", "1")
    eng = fresh_engine
    d = trajs["b"]
    calc = make_calculator(d).attach(engine=eng)
    mags, vecs = calc.get_k_path([1, 1, 0], 2.0, 20)
    ref, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], d["dt_ps"], vecs)
    assert rel_max(calc.calculate(mags, vecs).sed, ref) <= TOL          # psa_sed_project_upload
    eng.invalidate()
    eng.ensure_resident(0, d["velocities"])                               # psa_data_upload
    np.testing.assert_array_equal(eng.download(0, 0, 64), d["velocities"])
    assert rel_max(calc.calculate(mags, vecs).sed, ref) <= TOL


def test_new_entry_points_reject_misuse(fresh_engine, trajs):
    """Every round-2 ABI function validates its arguments and the call sequence (PSA_EINVAL /
    PSA_ESTATE -> PsaHipError, index errors -> ValueError like the reference) and leaves the context usable."""
    from psa_amd import _hip
    eng = fresh_engine
    d = trajs["a"]
    vel, mean = d["velocities"], O.mean_positions(d["positions"])
    kv = _kvecs(20)
    with pytest.raises(_hip.PsaHipError, match="no frame-sharded projection"):
        eng.fs_finish(True)
    with pytest.raises(_hip.PsaHipError, match="holds no array"):
        eng.fs_project(0, mean, kv, None, 0, 128, 0, 20)
    eng.ensure_resident(0, vel)
    with pytest.raises(_hip.PsaHipError, match="outside"):
        eng.fs_project(0, mean, kv, None, 0, 128, 15, 10)                 # k rows 15..25 of 20
    with pytest.raises(_hip.PsaHipError):
        eng.fs_project(0, mean, kv, None, 0, 64, 0, 20)                   # T_total < the slot's frames
    with pytest.raises(ValueError, match="out of bounds"):
        eng.fs_project(0, mean, kv, np.array([0, 64], np.int32), 0, 128, 0, 20)
    eng.fs_project(0, mean, kv, None, 0, 128, 0, 20)
    with pytest.raises(_hip.PsaHipError, match="tile the trajectory"):
        eng.fs_exchange([0], [64], [0], [20])
    with pytest.raises(_hip.PsaHipError, match="row range"):
        eng.fs_exchange([0], [128], [0], [19])
    with pytest.raises(_hip.PsaHipError):
        eng.fs_read(10, 11, 128)
    with pytest.raises(_hip.PsaHipError):
        eng.fs_write(100, np.zeros((20, 3, 64), np.complex64))
    eng.fs_exchange([0], [128], [0], [20])
    eng.fs_finish(True)
    ref, _, _ = O.calculate(d["positions"], vel, d["types"], d["dt_ps"], kv)
    assert rel_max(eng.finalize(128, 20, False), ref) <= TOL               # ... and still works
    with pytest.raises(ValueError, match="out of bounds"):
        eng.single_bin(0, mean, kv[0], np.array([1, 2, 99], np.int32), 3)
    with pytest.raises(_hip.PsaHipError, match="frame range"):
        eng.debug_project_only(0, mean, kv, None, frames=(100, 64))
    with pytest.raises(_hip.PsaHipError):
        eng.set_option(_hip.OPT_PLANES_MIN_K, 0)
    with pytest.raises(_hip.PsaHipError):
        eng.set_option(_hip.OPT_PLANES_BUDGET, -5)
    with pytest.raises(ValueError, match="out of bounds"):
        eng.project_upload(0, np.array(vel), mean, kv, [np.array([0, 1, 64])], 0)
    eng.project(0, mean, kv)                                                # rejected before the slot was touched:
    assert rel_max(eng.finalize(128, 20, False), ref) <= TOL               # the old array is still there, whole
    eng.project_upload(0, vel, mean, kv, None, 0)
    assert rel_max(eng.finalize(128, 20, False), ref) <= TOL
