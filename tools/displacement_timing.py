#!/usr/bin/env python3
"""Displacement mode (use_displacements=True) at configuration-3 size on the GPU box: the synthetic
array stands in for positions; first call (builds the split planes of positions - mean in HBM), later calls, and the
float32 kernel that subtracts while staging."""
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import _hip, synth     # noqa: E402

spec, req = synth.baseline_spec("C3")
r0, types, box = synth.lattice(spec.cells)
eng = _hip.Engine(0)
synth.fill_device(eng, _hip.SLOT_POSITIONS, spec, synth.mode_tables(spec, r0))
for call in range(2):
    t0 = time.perf_counter()
    mean = eng.mean_positions(_hip.SLOT_POSITIONS)
    print(f"mean over frames (bit-exact sequential float32 sum): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)
vecs = (np.linspace(0, kmax, 256, dtype=np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
for name, sel in (("2xf16 kernel on planes built from positions - mean", _hip.K1_AUTO), ("float32 kernel, subtract while staging", _hip.K1_MFMA32)):
    eng.set_k1(sel)
    for call in range(3):
        eng.k1_stats()
        t0 = time.perf_counter()
        eng.project(_hip.SLOT_POSITIONS, mean, vecs, None, _hip.F_DISPLACEMENTS)
        eng.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        n, ms = eng.k1_stats()
        print(f"{name}: call {call}: wall {wall:7.2f} ms, K1 {ms / max(n, 1):7.2f} ms", flush=True)
