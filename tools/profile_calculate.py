import cProfile, pstats, sys, time, weakref
sys.path.insert(0, '/root/repo')
import numpy as np
from psa_amd import _hip, synth, SEDCalculator, Trajectory
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
spec, req = synth.baseline_spec(cfg)
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
kw = dict(basis_atom_types=req["basis_atom_types"]) if req.get("basis_atom_types") else {}
for _ in range(3): calc.calculate(mags, vecs, **kw)
n = 5 if cfg == "C3" else 200
t0 = time.perf_counter()
for _ in range(n): s = calc.calculate(mags, vecs, **kw); i = s.intensity; del s, i
print(f"{cfg}: calculate + intensity {1e3 * (time.perf_counter() - t0) / n:.3f} ms per call (no profiler)")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): s = calc.calculate(mags, vecs, **kw); i = s.intensity; del s, i
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
