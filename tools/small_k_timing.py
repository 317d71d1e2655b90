import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from psa_amd import _hip, synth
spec, req = synth.baseline_spec("C3")
r0, types, box = synth.lattice(spec.cells); tables = synth.mode_tables(spec, r0)
eng = _hip.Engine(0)
synth.fill_device(eng, 0, spec, tables)
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)
for K in (16, 8, 4, 1):
    vecs = (np.linspace(0, kmax, max(K,2), dtype=np.float32)[:K, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
    for mink in (17, 1):
        eng.set_option(_hip.OPT_PLANES_MIN_K, mink)
        for _ in range(3): eng.project(0, r0, vecs)
        eng.synchronize(); eng.k1_stats()
        for _ in range(8): eng.project(0, r0, vecs)
        eng.synchronize(); n, ms = eng.k1_stats()
        print(f"K={K:3d} min_k={mink:2d}: K1 {ms/n:.3f} ms  ({'planes 32-row' if mink==1 else 'bf16x3'})  {25.77/(ms/n)/8*100/1e0:.1f}% of 8 TB/s", flush=True)
