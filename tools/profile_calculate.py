import cProfile, pstats, sys, time, weakref
sys.path.insert(0, '/root/repo')
import numpy as np
from psa_amd import _hip, synth, SEDCalculator, Trajectory
spec, req = synth.baseline_spec("C3")
r0, types, box = synth.lattice(spec.cells); tables = synth.mode_tables(spec, r0)
T, N = spec.n_frames, spec.n_atoms
eng = _hip.Engine(0)
synth.fill_device(eng, 0, spec, tables)
stand = np.broadcast_to(np.float32(0), (T, N, 3)); pos = np.broadcast_to(r0, (T, N, 3))
traj = Trajectory(pos, stand, types, np.broadcast_to(np.float32(0), (T,)), box, np.diag(box).copy(), np.zeros(3, np.float32), spec.dt_ps)
calc = SEDCalculator(traj, *spec.cells).attach(engine=eng)
eng.adopt(0, stand)
calc._mean_cache = (weakref.ref(pos), r0, _hip.Engine._fingerprint(pos))
mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
for _ in range(3): calc.calculate(mags, vecs, basis_atom_types=[1, 2])
pr = cProfile.Profile(); pr.enable()
for _ in range(5): s = calc.calculate(mags, vecs, basis_atom_types=[1, 2]); del s
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
