"""GPU tests at BASELINE.json's sizes.

C2 (8192 x 16384 x 128) is small enough for the CPU oracle, so it is compared directly.
C3 (32768 x 65536 x 256; 25.8 GB of velocities generated in HBM) is checked through
size-independent properties plus exact spot checks: the projection of sampled frames against
the oracle (the generator has a bit-identical NumPy twin, so any frame block can be rebuilt on
the host), the FFT of the device's own q against numpy, Parseval, the planted plane-wave peaks
(known answer) and k-shard invariance.  C4 / C5 run through the public API at reduced T."""
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
