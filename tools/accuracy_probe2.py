#!/usr/bin/env python3
"""Where do the projection kernels differ from float64?  Pre-FFT q per frame (max, rms, signed
bias) and, after the FFT, the location of the largest intensity error and dS/S there -- the probe
that exposed the coherent bias of the single-accumulator split kernel."""
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
for m in spec.modes:
    m.freq_bin = max(1, m.freq_bin * T // 65536)
r0, types, box = synth.lattice(spec.cells)
tables = synth.mode_tables(spec, r0)
vel = np.concatenate([synth.velocities_block(spec, tables, t, 128) for t in range(0, T, 128)])
mean = O.mean_positions(np.broadcast_to(r0, vel.shape))
kmax = 2 * np.pi / synth.A_SI / np.sqrt(2)
vecs = (np.linspace(0, kmax, 256, dtype=np.float32)[:, None] * np.array([1, 1, 0], np.float32) / np.sqrt(2)).astype(np.float32)
kv = vecs[[0, 3, 64, 128, 201, 255]]
arg = np.dot(kv, mean.T).astype(np.float64)
q64 = np.einsum("tac,ka->kct", vel.astype(np.float64), np.exp(1j * arg))
eng = _hip.Engine(0)
eng.ensure_resident(0, vel)
for name, sel in (("f16x2", _hip.K1_AUTO), ("bf16x3", _hip.K1_SPLIT_BF16), ("mfma32", _hip.K1_MFMA32)):
    eng.set_k1(sel)
    q = eng.debug_project_only(0, mean, kv)
    d = np.abs(q - q64)
    scale = np.abs(q64).max()
    print(f"{name}: max|dq|/max|q| = {d.max()/scale:.3e}  rms = {np.sqrt((d**2).mean())/scale:.3e}")
    per_t = d.max(axis=(0, 1))
    worst = np.argsort(per_t)[-5:]
    print("   worst frames:", worst, per_t[worst] / scale)
    per_k = d.max(axis=(1, 2)) / scale
    print("   per k:", per_k)
    # mean signed relative error (bias) of the dominant component
    rel = ((q - q64) / scale)
    print("   mean signed err re/im:", rel.real.mean(), rel.imag.mean(), " mean over t of err at k=255,c=2:", rel[5, 2].mean())

print("---- after the FFT ----")
S64 = (np.fft.fft(q64, axis=2) / T).transpose(2, 0, 1)          # (T,K,3)
I64 = np.sum(np.abs(S64) ** 2, axis=-1)
for name, sel in (("f16x2", _hip.K1_AUTO), ("bf16x3", _hip.K1_SPLIT_BF16), ("mfma32", _hip.K1_MFMA32)):
    eng.set_k1(sel)
    S = eng.calculate(0, mean, kv).astype(np.complex128)
    I = np.sum(np.abs(S) ** 2, axis=-1)
    dI = np.abs(I - I64)
    w, k = np.unravel_index(np.argmax(dI), dI.shape)
    print(f"{name}: max dI/maxI = {dI.max()/I64.max():.3e} at w={w} k={k}; I64 there = {I64[w,k]:.4e}, maxI = {I64.max():.4e} at {np.unravel_index(np.argmax(I64), I64.shape)}")
    for c in range(3):
        print(f"    c={c}: S64 = {S64[w,k,c]:.6e}   dS/S = {(S[w,k,c]-S64[w,k,c])/S64[w,k,c]:.3e}")
    # second: exclude that bin
    dI2 = dI.copy(); dI2[w, k] = 0
    w2, k2 = np.unravel_index(np.argmax(dI2), dI2.shape)
    print(f"    next: dI/maxI = {dI2.max()/I64.max():.3e} at w={w2} k={k2}, I64 = {I64[w2,k2]:.4e}")
