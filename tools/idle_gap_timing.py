#!/usr/bin/env python3
"""Does the projection kernel run slower right after the GPU sat idle? (GPU box)  Configuration 3,
K = 256 / 192: launches back to back, then separated by host sleeps, then by a D2H of the result."""
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import _hip, synth     # noqa: E402

spec, req = synth.baseline_spec("C3")
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
eng = _hip.Engine(0)
synth.fill_device(eng, 0, spec, tables)
T = spec.n_frames
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)


def vecs(K):
    return (np.linspace(0, kmax, K, dtype=np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)


def one(kv):
    eng.project(0, r0, kv)
    eng.synchronize()
    n, ms = eng.k1_stats()
    return ms / max(1, n)


for K in (256, 192):
    kv = vecs(K)
    for _ in range(3):
        eng.project(0, r0, kv)
    eng.synchronize(); eng.k1_stats()
    for _ in range(8):
        eng.project(0, r0, kv)
    eng.synchronize()
    n, ms = eng.k1_stats()
    print(f"K={K}: back to back              {ms / n:7.3f} ms", flush=True)
    for gap in (0.0, 0.001, 0.004, 0.02, 0.2):
        ts = []
        for _ in range(6):
            time.sleep(gap)
            ts.append(one(kv))
        print(f"K={K}: sync + sleep {gap * 1e3:5.1f} ms each  " + " ".join(f"{t:7.3f}" for t in ts), flush=True)
    ts = []
    for _ in range(5):
        ts.append(one(kv))
        out = eng.finalize(T, K, False)          # transpose + 403 MB D2H
        del out
    print(f"K={K}: D2H of the result between " + " ".join(f"{t:7.3f}" for t in ts), flush=True)
