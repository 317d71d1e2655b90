#!/usr/bin/env python3
"""Ablation of the six split terms: GPU (PSA_TERMS mask) vs float64 emulation of the same terms."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import psa_oracle as O            # noqa: E402
from psa_amd import _hip, synth                # noqa: E402

def bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split3(x):
    x = x.astype(np.float32); x1 = bf16(x); r = (x - x1).astype(np.float32); x2 = bf16(r); r2 = (r - x2).astype(np.float32)
    return x1, x2, bf16(r2)

T = int(os.environ.get("PROBE_T", "2048"))
mask = int(os.environ.get("PSA_TERMS", "63"))
spec, req = synth.baseline_spec("C3")
spec.n_frames = T
for m in spec.modes:
    m.freq_bin = max(1, m.freq_bin * T // 65536)
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
vel = np.concatenate([synth.velocities_block(spec, tables, t, 128) for t in range(0, T, 128)])
mean = O.mean_positions(np.broadcast_to(r0, vel.shape))
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)
kv = (np.array([kmax], np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
arg = np.dot(kv, mean.T)
P = np.concatenate([np.cos(arg), np.sin(arg)]).astype(np.float32)        # (2, N) rows cos, sin  (float32 like the kernel's sincosf ~)
p = split3(P); v = split3(vel[:, :, 2])                                   # z component only
terms = {1: (2, 0), 2: (0, 2), 4: (1, 1), 8: (1, 0), 16: (0, 1), 32: (0, 0)}
q = np.zeros((2, T))
for bit, (i, j) in terms.items():
    if mask & bit:
        q += p[i].astype(np.float64) @ v[j].T.astype(np.float64)
S_emul = np.fft.fft(q[0] + 1j * q[1]) / T
eng = _hip.Engine(0)
eng.ensure_resident(0, vel)
S = eng.calculate(0, mean, kv)[:, 0, 2].astype(np.complex128)
w = int(np.argmax(np.abs(S_emul)))
print(f"terms mask {mask:2d}: peak bin {w}  S_emul = {S_emul[w]:.6e}   (S_gpu - S_emul)/S = {(S[w]-S_emul[w])/S_emul[w]:.3e}   rms rel err {np.sqrt(np.mean(np.abs(S-S_emul)**2))/np.abs(S_emul[w]):.2e}")
