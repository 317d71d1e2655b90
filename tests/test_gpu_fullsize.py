"""GPU tests at BASELINE.json's sizes.

C2 (8192 x 16384 x 128) is small enough for the CPU oracle, so it is compared directly.
C3 (32768 x 65536 x 256; 25.8 GB of velocities generated in HBM) is checked through
size-independent properties plus exact spot checks: the projection of sampled frames against
the oracle (the generator has a bit-identical NumPy twin, so any frame block can be rebuilt on
the host), the FFT of the device's own q against numpy, Parseval, the planted plane-wave peaks
(known answer) and k-shard invariance -- coherent, and incoherent with its two 16384-atom type
groups (index lists: gather kernel on first use, compacted split planes afterwards).  C4 / C5
run through the public API against the oracle at reduced T, and at FULL size through the same
kind of properties (sampled frames, FFT, Parseval, k-subset / shard invariance, the k -> -k
symmetry of a real trajectory, the planted modes' frequency, chiral phase)."""
import numpy as np
import pytest

from conftest import rel_max
from oracle import psa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _host_velocities(spec, tables, n_frames=None):
    from psa_amd import synth
    n = n_frames or spec.n_frames
    return np.concatenate([synth.velocities_block(spec, tables, t, min(256, n - t)) for t in range(0, n, 256)])


def _calculator(spec, types, box, pos, vel, engine):
    from psa_amd import SEDCalculator, Trajectory
    T = vel.shape[0]
    tr = Trajectory(pos, vel, types, np.arange(T, dtype=np.float32), box, np.diag(box).copy(),
                    np.zeros(3, np.float32), spec.dt_ps)
    return SEDCalculator(tr, *spec.cells).attach(engine=engine)


def test_config2_full_size_against_oracle(engine):
    """8192 atoms x 16384 steps x 128 k-points, [100], coherent: the whole array vs the oracle."""
    from psa_amd import synth
    spec, req = synth.baseline_spec("C2")
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    vel = _host_velocities(spec, tables)
    pos = np.broadcast_to(r0, vel.shape)
    calc = _calculator(spec, types, box, pos, vel, engine)
    sed = calc.calculate_kpath_sed(req["direction"], req["bz_coverage"], req["n_k"])
    ref, freqs, _ = O.calculate(pos, vel, types, spec.dt_ps, sed.k_vectors)
    assert sed.sed.shape == (16384, 128, 3)
    assert rel_max(sed.intensity, O.intensity(ref)) <= TOL
    assert rel_max(sed.sed, ref) <= TOL
    np.testing.assert_array_equal(sed.freqs, freqs)
    # device-side intensity == host property
    np.testing.assert_allclose(engine.result_intensity(16384, 128), sed.intensity, rtol=3e-6, atol=0)


@pytest.fixture(scope="module")
def config3(engine):
    from psa_amd import SEDCalculator, Trajectory, _hip, synth
    spec, req = synth.baseline_spec("C3")
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables)
    stub = np.zeros((1, spec.n_atoms, 3), np.float32)
    calc = SEDCalculator(Trajectory(stub, stub, types, np.zeros(1, np.float32), box, np.diag(box).copy(),
                                    np.zeros(3, np.float32), spec.dt_ps), *spec.cells)
    _, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
    yield dict(spec=spec, r0=r0, types=types, tables=tables, vecs=vecs)
    engine.release(_hip.SLOT_VELOCITIES)


def test_config3_generated_trajectory_is_the_numpy_twin(engine, config3):
    from psa_amd import synth
    spec, tables = config3["spec"], config3["tables"]
    for t0 in (0, 31337, spec.n_frames - 8):
        np.testing.assert_array_equal(engine.download(0, t0, 8), synth.velocities_block(spec, tables, t0, 8))


def test_config3_projection_fft_and_parseval(engine, config3):
    """Full N = 32768, T = 65536: q of 6 k-points, sampled frames vs the oracle; then the FFT."""
    from psa_amd import synth
    spec, r0, tables = config3["spec"], config3["r0"], config3["tables"]
    T = spec.n_frames
    pick = [0, 1, 17, 128, 200, 255]
    vecs = config3["vecs"][pick]
    q = engine.debug_project_only(0, r0, vecs)                          # (6, 3, T) before the FFT
    phase = O.phase_table(vecs, r0)
    for t0 in (0, 40000, T - 32):
        block = synth.velocities_block(spec, tables, t0, 32)
        ref = O.project_group(block, phase)                             # (32, 6, 3), float32 BLAS
        got = q[:, :, t0:t0 + 32].transpose(2, 0, 1)
        # 32768-term float32 sums carry ~3e-6 of rounding noise on either side (it averages out
        # in the FFT); judge both against the same float64 evaluation
        exact = np.einsum("tac,ka->tkc", block.astype(np.float64), phase.astype(np.complex128))
        assert rel_max(got, exact) < 6e-6 and rel_max(ref, exact) < 6e-6
        assert rel_max(got, ref) < TOL
    engine.project(0, r0, vecs, None, 0)
    sed = engine.finalize(T, len(pick), False)                          # (T, 6, 3)
    want = (np.fft.fft(q, axis=2) / T).astype(np.complex64).transpose(2, 0, 1)
    assert rel_max(sed, want) < 2e-6
    # Parseval: sum_w |S|^2 = (1/T) sum_t |q|^2 for every (k, c)
    lhs = np.sum(np.abs(sed.astype(np.complex128)) ** 2, axis=0)
    rhs = np.sum(np.abs(q.astype(np.complex128)) ** 2, axis=2) / T
    np.testing.assert_allclose(lhs, rhs, rtol=2e-5)


def test_config3_f16_kernel_before_the_fft(engine, config3):
    """Full N = 32768: the "2 x f16" kernel's q (24 k-points -> 64-row blocks; 40 -> 128-row blocks),
    sampled frames against float64.  This is where chain length shows: one MFMA chain over all 1024
    stages was 1.4e-5 off at the coherent peak; folded every 8 stages it is as good as float32 BLAS."""
    from psa_amd import synth
    spec, r0, tables = config3["spec"], config3["r0"], config3["tables"]
    T = spec.n_frames
    for n_k in (24, 40):
        pick = np.linspace(0, 255, n_k).round().astype(int)
        vecs = config3["vecs"][pick]
        q = engine.debug_project_only(0, r0, vecs)
        phase = O.phase_table(vecs, r0)
        for t0 in (0, 40000, T - 32):
            block = synth.velocities_block(spec, tables, t0, 32)
            ref = O.project_group(block, phase)
            got = q[:, :, t0:t0 + 32].transpose(2, 0, 1)
            exact = np.einsum("tac,ka->tkc", block.astype(np.float64), phase.astype(np.complex128))
            assert rel_max(got, exact) < 3e-6 and rel_max(ref, exact) < 6e-6
            assert rel_max(got, ref) < TOL


def test_config3_planted_modes_and_shard_invariance(engine, config3):
    """All 256 k-points: the planted plane wave on the [110] path shows up as the known peak,
    and projecting the k-list in two shards (as two ranks would) changes nothing."""
    spec, r0, vecs = config3["spec"], config3["r0"], config3["vecs"]
    T, K, N = spec.n_frames, len(vecs), spec.n_atoms
    engine.project(0, r0, vecs, None, 0)
    engine.finalize(T, K, False, fetch=False)
    inten = engine.result_intensity(T, K)
    mode = spec.modes[1]                                               # (0.5,0.5,0) 2pi/a, z-polarised
    k_star = int(np.argmin(np.linalg.norm(vecs - np.asarray(mode.k_vec, np.float32), axis=1)))
    assert k_star == K - 1
    w_star = int(np.argmax(inten[:, k_star]))
    assert w_star == mode.freq_bin
    expect = (N * mode.amp / 2) ** 2                                   # |S|^2 of a coherent cos mode
    assert abs(inten[w_star, k_star] / expect - 1) < 0.02
    assert inten[w_star, k_star] > 100 * np.median(inten[:, k_star])
    half = K // 2
    engine.project(0, r0, vecs[:half], None, 0, K_total=K, k_offset=0)
    engine.project(0, r0, vecs[half:], None, 0, K_total=K, k_offset=half)
    engine.finalize(T, K, False, fetch=False)
    np.testing.assert_allclose(engine.result_intensity(T, K), inten, rtol=1e-6, atol=0)


def test_config3_public_calculate_at_full_size(engine, config3):
    """The PUBLIC call at configuration 3 -- SEDCalculator.calculate(k_mags, k_vecs, basis_atom_types=[1, 2])
    on the 25.8 GB trajectory generated in HBM: 256 k-vectors leave block by block (psa_sed_calculate's
    pipelined path), the intensity comes with them.  Held to (a) the engine-level project + finalize
    path over the whole (T, K, 3) array, (b) the oracle: the inverse FFT of 16 result columns gives
    q(t), compared on 64 blocks of 32 frames spread over the trajectory with the oracle's projection
    of the very same frames (rebuilt on the host by the generator's NumPy twin), (c) its own
    companion intensity."""
    import weakref
    from psa_amd import SEDCalculator, Trajectory, _hip, synth
    spec, r0, types, tables, vecs = (config3[k] for k in ("spec", "r0", "types", "tables", "vecs"))
    T, N, K = spec.n_frames, spec.n_atoms, len(vecs)
    box = synth.lattice(spec.cells)[2]
    stand_in = np.broadcast_to(np.float32(0), (T, N, 3))          # the trajectory exists only in HBM
    pos = np.broadcast_to(r0, (T, N, 3))
    calc = SEDCalculator(Trajectory(pos, stand_in, types, np.broadcast_to(np.float32(0), (T,)), box, np.diag(box).copy(),
                                    np.zeros(3, np.float32), spec.dt_ps), *spec.cells).attach(engine=engine)
    engine.adopt(_hip.SLOT_VELOCITIES, stand_in)
    calc._mean_cache = (weakref.ref(pos), r0, _hip.Engine._fingerprint(pos))
    sed = calc.calculate(np.zeros(K, np.float32), vecs, basis_atom_types=[1, 2])
    assert sed.sed.shape == (T, K, 3) and sed.sed.dtype == np.complex64 and sed.is_complex
    inten = sed.intensity
    assert inten.shape == (T, K) and inten.dtype == np.float32
    rows = np.arange(0, T, 257)
    np.testing.assert_allclose(inten[rows], np.sum(np.abs(sed.sed[rows]) ** 2, axis=-1), rtol=5e-6)
    # (a) the engine-level path, whole array
    engine.project(_hip.SLOT_VELOCITIES, r0, vecs, None, 0)
    plain = engine.finalize(T, K, False)
    assert rel_max(sed.sed, plain) <= 2e-6
    del plain
    # (b) the oracle on sampled frames, through the inverse transform of 16 columns
    pick = np.linspace(0, K - 1, 16).round().astype(int)
    q = np.fft.ifft(sed.sed[:, pick, :].astype(np.complex128), axis=0) * T          # (T, 16, 3)
    phase = O.phase_table(vecs[pick], r0)
    scale = float(np.max(np.abs(q)))
    worst = 0.0
    for t0 in np.linspace(0, T - 32, 64).astype(int) // 32 * 32:
        block = synth.velocities_block(spec, tables, int(t0), 32)
        ref = O.project_group(block, phase)                                         # (32, 16, 3)
        worst = max(worst, float(np.max(np.abs(q[t0:t0 + 32] - ref))) / scale)
    assert worst <= TOL, worst


# ---------------------------------------------------------------------------- full-size helpers
def _device_config(engine, name):
    """The configuration's trajectory generated in HBM + the host objects that describe it."""
    from psa_amd import SEDCalculator, Trajectory, _hip, synth
    spec, req = synth.baseline_spec(name)
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables)
    stub = np.zeros((1, spec.n_atoms, 3), np.float32)
    calc = SEDCalculator(Trajectory(stub, stub, types, np.zeros(1, np.float32), box, np.diag(box).copy(),
                                    np.zeros(3, np.float32), spec.dt_ps), *spec.cells)
    return spec, req, r0, types, tables, calc


def _check_sampled_frames(engine, spec, tables, r0, vecs, idx, frames, tol_exact=6e-6):
    """q of the k-vectors `vecs` for the atom list `idx` (None: all), sampled frame blocks against
    the oracle's float32 BLAS and against float64; returns q (K,3,T)."""
    from psa_amd import synth
    q = engine.debug_project_only(0, r0, vecs, idx)
    atoms = np.arange(spec.n_atoms) if idx is None else np.asarray(idx)
    phase = O.phase_table(vecs, r0[atoms])
    for t0 in frames:
        block = synth.velocities_block(spec, tables, t0, 32)[:, atoms, :]
        ref = O.project_group(block, phase)
        got = q[:, :, t0:t0 + 32].transpose(2, 0, 1)
        exact = np.einsum("tac,ka->tkc", block.astype(np.float64), phase.astype(np.complex128))
        assert rel_max(got, exact) < tol_exact and rel_max(ref, exact) < tol_exact
        assert rel_max(got, ref) < TOL
    return q


def _check_fft_and_parseval(engine, q, idx_groups, r0, vecs, T):
    """The device's FFT of its own q against numpy, and Parseval per (k, c)."""
    engine.project(0, r0, vecs, idx_groups, 0)
    sed = engine.finalize(T, len(vecs), False)
    want = (np.fft.fft(q, axis=2) / T).astype(np.complex64).transpose(2, 0, 1)
    assert rel_max(sed, want) < 2e-6
    lhs = np.sum(np.abs(sed.astype(np.complex128)) ** 2, axis=0)
    rhs = np.sum(np.abs(q.astype(np.complex128)) ** 2, axis=2) / T
    np.testing.assert_allclose(lhs, rhs, rtol=2e-5)
    return sed


def test_config3_incoherent_type_groups_at_full_size(engine, config3):
    """BASELINE's literal "2 basis types" workload: two groups of 16384 atoms (index lists) x 65536
    frames.  First use of a list -> gather-DMA kernel, second use -> its compacted split planes; both
    are held to the oracle on sampled frames, then the whole incoherent step (all 256 k) is checked
    for k-subset invariance, Parseval through the group sum, and against the coherent result's
    cross term on a sample."""
    from psa_amd import _hip
    spec, r0, types, tables, vecs = (config3[k] for k in ("spec", "r0", "types", "tables", "vecs"))
    T, K = spec.n_frames, len(vecs)
    groups = [np.flatnonzero(types == 1).astype(np.int32), np.flatnonzero(types == 2).astype(np.int32)]
    assert [len(g) for g in groups] == [16384, 16384]
    pick = [0, 3, 64, 129, 254, 255] + list(range(100, 118))            # 24 k-vectors: the f16 kernels
    sub = vecs[pick]
    qs = []
    for g in groups:
        n0 = engine.plane_cache()[0]
        q1 = _check_sampled_frames(engine, spec, tables, r0, sub, g, (0, 31337, T - 32))     # gather kernel
        assert engine.plane_cache()[0] == n0                                                   # (first sight)
        q2 = _check_sampled_frames(engine, spec, tables, r0, sub, g, (64, 40000, T - 64))    # compacted planes
        assert engine.plane_cache()[0] == n0 + 1
        assert rel_max(q1, q2) < 3e-6
        qs.append(q2)
    # the incoherent step on the subset: sum over groups of |FFT(q_g)/T|^2
    engine.project(0, r0, sub, groups, _hip.F_INTENSITY)
    inc_sub = engine.finalize(T, len(pick), True)                                             # (T, 24)
    want = sum(np.sum(np.abs(np.fft.fft(q, axis=2) / T) ** 2, axis=1) for q in qs).T
    assert rel_max(inc_sub, want.astype(np.float32)) < 2e-6
    # all 256 k-vectors: the subset's columns do not depend on the company they are computed in
    engine.project(0, r0, vecs, groups, _hip.F_INTENSITY)
    inc = engine.finalize(T, K, True)
    assert inc.shape == (T, K) and inc.dtype == np.float32
    np.testing.assert_allclose(inc[:, pick], inc_sub, rtol=0, atol=2e-6 * inc_sub.max())
    # Parseval through the whole step: sum_w I[w,k] = (1/T) sum_g sum_c sum_t |q_g|^2
    lhs = inc[:, pick].astype(np.float64).sum(axis=0)
    rhs = sum(np.sum(np.abs(q.astype(np.complex128)) ** 2, axis=(1, 2)) for q in qs) / T
    np.testing.assert_allclose(lhs, rhs, rtol=2e-5)
    # the planted [110] mode lives on all atoms: each type group carries half of its amplitude
    mode = spec.modes[1]
    expect = 2 * (len(groups[0]) * mode.amp / 2) ** 2
    assert abs(inc[mode.freq_bin, K - 1] / expect - 1) < 0.03


def test_config4_kgrid_reduced_T(engine):
    """50 x 50 k-grid (2500 k-points) x 8192 atoms, T = 2048, two basis types incoherent."""
    from psa_amd import synth
    spec, req = synth.baseline_spec("C4")
    spec.n_frames = 2048
    spec.modes = [m for m in spec.modes if m.freq_bin < 1024]
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    vel = _host_velocities(spec, tables)
    pos = np.broadcast_to(r0, vel.shape)
    calc = _calculator(spec, types, box, pos, vel, engine)
    sed = calc.calculate_kgrid_sed(req["plane"], req["k_ranges"], req["n_kx"], req["n_ky"],
                                   basis_atom_types=[1, 2], summation_mode="incoherent")
    assert sed.k_grid_shape == (50, 50) and sed.sed.shape == (2048, 2500) and not sed.is_complex
    ref, _, cx = O.calculate(pos, vel, types, spec.dt_ps, sed.k_vectors, basis_atom_types=[1, 2],
                             summation_mode="incoherent", k_chunk_size=10000)
    assert not cx and rel_max(sed.sed, ref) <= TOL
    heat = sed.sed.reshape(2048, 50, 50)                                # the xy heat-map layout
    assert heat[:, 3, 7].tolist() == sed.sed[:, 3 * 50 + 7].tolist()


def test_config5_chiral_reduced_T(engine):
    """16384 atoms x 128 k-points, complex output retained + chiral phase (T = 2048)."""
    from psa_amd import synth
    spec, req = synth.baseline_spec("C5")
    spec.n_frames = 2048
    spec.modes = [synth.Mode(3.0, 128, (0.25 * 2 * np.pi / synth.A_SI, 0, 0), 0),
                  synth.Mode(3.0, 128, (0.25 * 2 * np.pi / synth.A_SI, 0, 0), 1)]
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    vel = _host_velocities(spec, tables)
    rng = np.random.default_rng(2)
    pos = (r0[None] + 0.02 * rng.standard_normal((8, spec.n_atoms, 3))).astype(np.float32)
    pos = np.concatenate([pos] * (2048 // 8))                           # cheap non-static positions
    calc = _calculator(spec, types, box, pos, vel, engine)
    sed = calc.calculate_kpath_sed(req["direction"], req["bz_coverage"], req["n_k"], chiral=True,
                                   chiral_axis="z")
    ref, _, _ = O.calculate(pos, vel, types, spec.dt_ps, sed.k_vectors)
    assert sed.is_complex and rel_max(sed.sed, ref) <= TOL and rel_max(np.abs(sed.sed), np.abs(ref)) <= TOL
    phase_ref = O.chiral_phase(ref[:, :, 0], ref[:, :, 1], "C")
    strong = (np.abs(ref[:, :, 0]) > 1e-2 * np.abs(ref).max()) & (np.abs(ref[:, :, 1]) > 1e-2 * np.abs(ref).max())
    assert strong.sum() > 0
    assert np.max(np.abs(sed.phase - phase_ref)[strong]) < 1e-3


def _minus_k_symmetry(inten, n_kx, n_ky):
    """I[w, k] = I[-w, -k] for a real trajectory (q(-k) = conj q(k)); on a grid symmetric about 0 the
    partner of (i, j) is (n_kx-1-i, n_ky-1-j)."""
    T = inten.shape[0]
    grid = inten.reshape(T, n_kx, n_ky)
    partner = np.roll(grid[::-1], 1, axis=0)[:, ::-1, ::-1]              # w -> (T - w) % T
    return float(np.max(np.abs(grid - partner)) / np.max(np.abs(grid)))


def test_config4_kgrid_full_size_properties(engine):
    """50 x 50 k-grid x 16384 frames x 8192 atoms at full size: sampled frames of a 40-point subset
    against the oracle, FFT and Parseval on it, subset-vs-whole and two-shard invariance of the 2500-point
    result, and the k -> -k symmetry of the whole (T, 50, 50) heat map -- coherent and two-type incoherent."""
    from psa_amd import _hip
    spec, req, r0, types, tables, calc = _device_config(engine, "C4")
    T = spec.n_frames
    r = req["k_ranges"]
    _, vecs, shape = calc.get_k_grid(req["plane"], (r[0], r[1]), (r[2], r[3]), req["n_kx"], req["n_ky"], 0.0)
    K = len(vecs)
    assert (K, shape) == (2500, (50, 50))
    pick = np.linspace(0, K - 1, 40).round().astype(int)
    q = _check_sampled_frames(engine, spec, tables, r0, vecs[pick], None, (0, 9000, T - 32))
    sed_sub = _check_fft_and_parseval(engine, q, None, r0, vecs[pick], T)
    sub_int = np.sum(np.abs(sed_sub) ** 2, axis=-1)
    # the whole grid, coherent: intensity on the device, subset columns, shards, symmetry
    engine.project(0, r0, vecs, None, 0)
    engine.finalize(T, K, False, fetch=False)
    inten = engine.result_intensity(T, K)
    np.testing.assert_allclose(inten[:, pick], sub_int, rtol=0, atol=2e-6 * sub_int.max())
    half = 1250
    engine.project(0, r0, vecs[:half], None, 0, K_total=K, k_offset=0)
    engine.project(0, r0, vecs[half:], None, 0, K_total=K, k_offset=half)
    engine.finalize(T, K, False, fetch=False)
    np.testing.assert_allclose(engine.result_intensity(T, K), inten, rtol=0, atol=1e-6 * inten.max())
    assert _minus_k_symmetry(inten, 50, 50) < 5e-6
    # two basis types, incoherent (the heat-map use case): same checks on the (T, 2500) float32 output
    groups = [np.flatnonzero(types == t).astype(np.int32) for t in (1, 2)]
    for _ in range(2):                                                    # gather kernel, then compacted planes
        engine.project(0, r0, vecs, groups, _hip.F_INTENSITY)
        inc = engine.finalize(T, K, True)
        assert _minus_k_symmetry(inc, 50, 50) < 5e-6
        qs = [engine.debug_project_only(0, r0, vecs[pick[:8]], g) for g in groups]
        want = sum(np.sum(np.abs(np.fft.fft(x, axis=2) / T) ** 2, axis=1) for x in qs).T
        np.testing.assert_allclose(inc[:, pick[:8]], want, rtol=0, atol=3e-6 * want.max())
    heat = inc.reshape(T, 50, 50)
    assert heat[:, 3, 7].tolist() == inc[:, 3 * 50 + 7].tolist()
    engine.release(_hip.SLOT_VELOCITIES)


def test_config5_chiral_full_size_properties(engine):
    """16384 atoms x 32768 frames x 128 k-points, complex output kept + chiral phase, full size:
    sampled frames against the oracle, FFT / Parseval, the planted modes' frequency, shard
    invariance, and the device's chiral phase against the reference formula on the host."""
    from psa_amd import _hip
    spec, req, r0, types, tables, calc = _device_config(engine, "C5")
    T = spec.n_frames
    _, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
    K = len(vecs)
    q = _check_sampled_frames(engine, spec, tables, r0, vecs, None, (0, 20000, T - 32))
    sed = _check_fft_and_parseval(engine, q, None, r0, vecs, T)           # (T, 128, 3) on the host
    # planted x-polarised mode at k = (0.25, 0, 0) 2pi/a, frequency bin T/16: on the [100] path the
    # k-point nearest to it peaks at exactly that frequency in component x
    mode = spec.modes[0]
    k_star = int(np.argmin(np.linalg.norm(vecs - np.asarray(mode.k_vec, np.float32), axis=1)))
    assert int(np.argmax(np.abs(sed[:T // 2, k_star, 0]))) == mode.freq_bin
    # chiral phase "C" of (x, y) on the device == reference formula on the host result
    phase = engine.result_chiral_phase(T, K, 0, 1)
    want = O.chiral_phase(sed[:, :, 0], sed[:, :, 1], "C")
    # (same complex numbers on both sides -- `sed` is the device's own result -- so only atan2's
    # rounding differs, and the fold keeps the phase continuous across the +-pi wrap)
    assert np.max(np.abs(phase - want)) < 1e-4
    assert phase.dtype == np.float32 and np.all(np.abs(phase) <= np.pi / 2 + 1e-6)
    # four shards (the configuration's 4 GPUs) leave the result unchanged
    inten = engine.result_intensity(T, K)
    for lo in range(0, K, 32):
        engine.project(0, r0, vecs[lo:lo + 32], None, 0, K_total=K, k_offset=lo)
    engine.finalize(T, K, False, fetch=False)
    np.testing.assert_allclose(engine.result_intensity(T, K), inten, rtol=0, atol=2e-6 * inten.max())
    engine.release(_hip.SLOT_VELOCITIES)
