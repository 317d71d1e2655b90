#!/usr/bin/env python3
"""The 256-row form of the planes kernel (PSA_OPT_K1_WIDE) and the 128-row forms against the oracle, before the FFT,
on a few ragged shapes (GPU box; PSA_HIP_LIBRARY selects a side build of tools/k1_experiments.sh).  The oracle is the
checker here, as in tests/."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from psa_amd import _hip
import oracle.psa_oracle as O
eng = _hip.Engine(0)
eng.set_option(_hip.OPT_PLANES_EAGER, 1)
for (n_atoms, n_frames, n_k) in [(77, 150, 70), (640, 200, 128), (1000, 96, 300), (2000, 130, 129)]:
    rng = np.random.default_rng(n_atoms)
    r0 = rng.uniform(0, 30, (n_atoms, 3)).astype(np.float32)
    vel = rng.standard_normal((n_frames, n_atoms, 3)).astype(np.float32)
    kv = rng.uniform(-2, 2, (n_k, 3)).astype(np.float32)
    eng.ensure_resident(0, vel)
    ref = O.project_group(vel, O.phase_table(kv, r0))
    out = {}
    for wide in (0, 1):
        eng.set_option(_hip.OPT_K1_WIDE, wide)
        eng.debug_project_only(0, r0, kv, None)
        got = eng.debug_project_only(0, r0, kv, None).transpose(2, 0, 1)
        out[wide] = got
        print(n_atoms, n_frames, n_k, "wide" if wide else "narrow", float(np.abs(got - ref).max() / np.abs(ref).max()), flush=True)
    print("   wide vs narrow", float(np.abs(out[1] - out[0]).max() / np.abs(ref).max()))
