#!/usr/bin/env python3
"""The projection kernel on an all-zero trajectory of configuration-3 size (GPU box): same instruction
stream, no switching in the datapaths -- the gap to the time on random data is what the chip's
power management takes (MI355X_MICROARCH.md, DVFS give-back item 1)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import _hip, synth     # noqa: E402

spec, req = synth.baseline_spec("C3")
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
eng = _hip.Engine(0)
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)
vecs = (np.linspace(0, kmax, 256, dtype=np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
for name in ("random", "zeros", "random"):
    if name == "random":
        synth.fill_device(eng, 0, spec, tables)
    else:
        eng.ensure_resident(0, np.zeros((spec.n_frames, spec.n_atoms, 3), np.float32))
    for _ in range(3):
        eng.project(0, r0, vecs)
    eng.synchronize(); eng.k1_stats()
    for _ in range(8):
        eng.project(0, r0, vecs)
    eng.synchronize()
    n, ms = eng.k1_stats()
    print(f"{name:7s}: K1 {ms / n:.3f} ms", flush=True)
