#!/usr/bin/env python3
"""Does the projection kernel's time depend on the trajectory's row stride (12 N bytes)?  Frame
rows a power-of-two-ish stride apart put every workgroup's current atom columns on the same
memory channels.  Times K1 for 256 k-points x 65536 frames at several atom counts (GPU box)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import _hip, synth     # noqa: E402

eng = _hip.Engine(0)
T, K = 65536, 256
counts = [int(a) for a in sys.argv[1:]] or [32768, 32832, 33024, 30720, 24576, 24640]
for n in counts:
    eng.alloc(_hip.SLOT_VELOCITIES, T, n)
    z = np.zeros((0,), np.float32)
    eng.fill_synthetic(_hip.SLOT_VELOCITIES, 7, z, np.zeros((0,), np.int32), z, z, z, z)
    rng = np.random.default_rng(1)
    mean = (rng.random((n, 3)) * 80).astype(np.float32)
    vecs = (np.linspace(0, 0.8, K, dtype=np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
    for rep in range(3):
        eng.k1_stats()
        eng.project(_hip.SLOT_VELOCITIES, mean, vecs, None, 0)
        eng.synchronize()
        cnt, ms = eng.k1_stats()
    print(f"N = {n:6d} (row stride {12*n:7d} B = {12*n/256:9.2f} x 256 B): K1 {ms/max(cnt,1):7.3f} ms, "
          f"{ms/max(cnt,1)/n*32768:7.3f} ms per 32768 atoms", flush=True)
