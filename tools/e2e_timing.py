#!/usr/bin/env python3
"""End-to-end wall time of SEDCalculator.calculate() from host arrays (GPU box): first call
(uploads the trajectory), later calls (trajectory resident), with the library's stage timings."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import SEDCalculator, Trajectory, synth     # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
spec, req = synth.baseline_spec(cfg)
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
t0 = time.perf_counter()
vel = np.concatenate([synth.velocities_block(spec, tables, t, min(256, spec.n_frames - t))
                      for t in range(0, spec.n_frames, 256)])
print(f"{cfg}: host generation of V ({vel.nbytes/1e9:.2f} GB): {time.perf_counter()-t0:.1f} s")
pos = np.broadcast_to(r0, vel.shape)
tr = Trajectory(pos, vel, types, np.arange(spec.n_frames, dtype=np.float32), box, np.diag(box).copy(),
                np.zeros(3, np.float32), spec.dt_ps)
calc = SEDCalculator(tr, *spec.cells)
mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
units = spec.n_atoms * spec.n_frames * len(vecs)
for i in range(3):
    t0 = time.perf_counter()
    sed = calc.calculate(mags, vecs)
    dt = time.perf_counter() - t0
    tm = calc.engine.timings()
    print(f"  call {i}: {dt*1e3:8.1f} ms  ({units/dt:.3e} units/s)  stages(ms): " +
          ", ".join(f"{k}={v:.2f}" for k, v in tm.items() if v > 0.005))
t0 = time.perf_counter(); inten = sed.intensity; print(f"  SED.intensity on host: {(time.perf_counter()-t0)*1e3:.1f} ms")
