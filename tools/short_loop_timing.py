#!/usr/bin/env python3
"""Per-stage time of the projection kernel for short atom loops (GPU box): 8192 atoms (256 stages per
workgroup) with T = 16384 (2 rounds of workgroups per CU at 128 k-points) and T = 131072 (16 rounds);
32768 atoms for comparison."""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import _hip, synth     # noqa: E402

REPS = int(os.environ.get("PSA_DEBUG_REPEAT_K1", "1"))      # > 1: the library launches K1 that often back to back
eng = _hip.Engine(0)
for cells, T, K in (((16, 8, 8), 16384, 128), ((16, 8, 8), 131072, 128), ((16, 8, 8), 131072, 256), ((16, 16, 16), 16384, 128),
                    ((16, 16, 16), 65536, 128), ((16, 16, 8), 32768, 128), ((8, 8, 8), 262144, 128)):
    spec = synth.SyntheticSpec(cells, T)
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    synth.fill_device(eng, 0, spec, tables)
    kmax = 2 * np.pi / synth.A_SI
    vecs = (np.linspace(0, kmax, K, dtype=np.float32)[:, None] * np.array([1, 0, 0], np.float32)).astype(np.float32)
    for _ in range(3):
        eng.project(0, r0, vecs)
    eng.synchronize(); eng.k1_stats()
    for _ in range(10):
        eng.project(0, r0, vecs)
    eng.synchronize()
    n, ms = eng.k1_stats()
    n_stage = spec.n_atoms // 32
    wgs = (2 * K // 128) * (T // 64)
    rounds = wgs / 256
    print(f"[K1 x{REPS} per project] atoms {spec.n_atoms:6d} T {T:7d} K {K}: K1 {ms / n:8.3f} ms  workgroups {wgs} ({rounds:.0f} rounds x {n_stage} stages)  "
          f"{1e3 * ms / n / (rounds * n_stage):.3f} us per stage  frac {12.0 * spec.n_atoms * T * K / (ms / n * 1e-3) / 833.3e12:.3f}", flush=True)
