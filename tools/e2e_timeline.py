#!/usr/bin/env python3
"""Where the drop-in call spends its time (GPU box): SEDCalculator.calculate() + .intensity on the
resident configuration-3 trajectory with the library's per-block timeline (PSA_TIMELINE=1: device
events of both streams, ms since entry) and the host-side cost around the library call.
    python tools/e2e_timeline.py [C3] [blocks, e.g. 192,64]"""
import os
import sys
import time
import weakref
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
if len(sys.argv) > 2:
    os.environ["PSA_PIPELINE_BLOCKS"] = sys.argv[2]
import numpy as np                                             # noqa: E402
from psa_amd import SEDCalculator, Trajectory, _hip, synth     # noqa: E402

spec, req = synth.baseline_spec(cfg)
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
T, N = spec.n_frames, spec.n_atoms
eng = _hip.Engine(0)
synth.fill_device(eng, 0, spec, tables)
stand = np.broadcast_to(np.float32(0), (T, N, 3))
pos = np.broadcast_to(r0, (T, N, 3))
traj = Trajectory(pos, stand, types, np.broadcast_to(np.float32(0), (T,)), box, np.diag(box).copy(), np.zeros(3, np.float32),
                  spec.dt_ps)
calc = SEDCalculator(traj, *spec.cells).attach(engine=eng)
eng.adopt(0, stand)
calc._mean_cache = (weakref.ref(pos), r0, _hip.Engine._fingerprint(pos))
if req["kind"] == "path":
    mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
    shape = None
else:
    r = req["k_ranges"]
    mags, vecs, shape = calc.get_k_grid(req["plane"], (r[0], r[1]), (r[2], r[3]), req["n_kx"], req["n_ky"], 0.0)
kw = dict(basis_atom_types=req["basis_atom_types"]) if req.get("basis_atom_types") else {}
for _ in range(3):
    s = calc.calculate(mags, vecs, k_grid_shape=shape, **kw)
    del s
walls, lib = [], []
inner = eng._lib.psa_sed_calculate


def timed_inner(*a):
    t0 = time.perf_counter()
    rc = inner(*a)
    lib.append(time.perf_counter() - t0)
    return rc


eng._lib.psa_sed_calculate = timed_inner
for i in range(6):
    if i == 5:
        os.environ["PSA_TIMELINE"] = "1"
    t0 = time.perf_counter()
    s = calc.calculate(mags, vecs, k_grid_shape=shape, **kw)
    t1 = time.perf_counter()
    inten = s.intensity
    t2 = time.perf_counter()
    walls.append((t1 - t0, t2 - t1))
    del s, inten
os.environ.pop("PSA_TIMELINE", None)
for (c, i), l in zip(walls, lib):
    print(f"calculate {c * 1e3:7.3f} ms  (library call {l * 1e3:7.3f}, host around it {(c - l) * 1e3:6.3f})   .intensity {i * 1e3:6.3f} ms",
          flush=True)
print("blocks:", os.environ.get("PSA_PIPELINE_BLOCKS", "default"), " stages:", {k: round(v, 3) for k, v in eng.timings().items()})
print("page-locked result pool:", _hip._pinned_pool.stats)
