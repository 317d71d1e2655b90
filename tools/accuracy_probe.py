#!/usr/bin/env python3
"""Accuracy probe (GPU box): split-precision kernel vs exact-fp32 kernel vs the float32 oracle,
all judged against a float64 evaluation of the same math, on a C3-shaped sample."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import psa_oracle as O            # noqa: E402
from psa_amd import _hip, synth                # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec, req = synth.baseline_spec("C3")
spec.n_frames = T
spec.modes = [m for m in synth.baseline_spec("C3")[0].modes]
for m in spec.modes:
    m.freq_bin = max(1, m.freq_bin * T // 65536)
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
vel = np.concatenate([synth.velocities_block(spec, tables, t, 128) for t in range(0, T, 128)])
pos = np.broadcast_to(r0, vel.shape)
mean = O.mean_positions(pos)
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)
vecs = (np.linspace(0, kmax, 256, dtype=np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
pick = np.array([0, 3, 64, 128, 201, 255])
kv = vecs[pick]

# float64 reference: same float32 phase ARGUMENT as the reference, everything after in float64
arg = np.dot(kv, mean.T).astype(np.float64)
P64 = np.exp(1j * arg)
q64 = np.einsum("tac,ka->tkc", vel.astype(np.float64), P64)
S64 = np.fft.fft(q64, axis=0) / T
I64 = np.sum(np.abs(S64) ** 2, axis=-1)

ref, _, _ = O.calculate(pos, vel, types, spec.dt_ps, kv)
eng = _hip.Engine(0)
eng.ensure_resident(0, vel)
res = {"oracle(f32)": ref}
for name, sel in (("f16x2", _hip.K1_AUTO), ("bf16x3", _hip.K1_SPLIT_BF16), ("mfma32", _hip.K1_MFMA32)):
    eng.set_k1(sel)
    res[name] = eng.calculate(0, mean, kv)
den = np.abs(I64).max()
print(f"T={T} N={vel.shape[1]} K={len(pick)}   max-norm relative error of intensity vs float64")
for name, s in res.items():
    I = np.sum(np.abs(s.astype(np.complex128)) ** 2, axis=-1)
    print(f"  {name:12s} {np.abs(I - I64).max() / den:.3e}    sed: {np.abs(s - S64).max() / np.abs(S64).max():.3e}")
for a, b in (("split", "oracle(f32)"), ("mfma32", "oracle(f32)"), ("split", "mfma32")):
    Ia = np.sum(np.abs(res[a]) ** 2, axis=-1).astype(np.float32)
    Ib = np.sum(np.abs(res[b]) ** 2, axis=-1).astype(np.float32)
    print(f"  {a} vs {b}: {np.abs(Ia - Ib).max() / np.abs(Ib).max():.3e}")
