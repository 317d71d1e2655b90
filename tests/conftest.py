import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The suite needs libpsa_hip.so (git-ignored build product): build it if it is not there
    (hipcc cross-compiles gfx950 without a GPU)."""
    lib = ROOT / "psa_amd" / "csrc" / "libpsa_hip.so"
    import shutil
    import subprocess
    if shutil.which("make") and (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
        # unconditional: a no-op when the library is newer than its sources, and a stale
        # library never survives a source edit
        subprocess.run(["make", "-C", str(lib.parent), "-j", "8"], check=True, stdout=subprocess.DEVNULL)
    elif not lib.exists():
        raise RuntimeError(f"{lib} is missing and there is no hipcc to build it")
    yield


def rel_max(a, b):
    """max-norm relative error  max|a-b| / max|b|  (SURVEY.md section 8d parity metric)."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    den = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / (den if den > 0 else 1.0)


@pytest.fixture(scope="session")
def golden():
    """All captured reference outputs, keyed like 'coh_all/sed'."""
    out = {}
    for f in ("calc_cases.npz", "calc_wide.npz", "calc_sym.npz", "calc_w256.npz", "kgen_cases.npz", "chiral_cases.npz"):
        with np.load(GOLDEN / f) as z:
            out.update({k: z[k] for k in z.files})
    return out


@pytest.fixture(scope="session")
def trajs():
    out = {}
    for name in ("a", "b", "c"):
        with np.load(GOLDEN / f"traj_{name}.npz") as z:
            d = {k: z[k] for k in z.files}
        d["dt_ps"] = float(d["dt_ps"])
        d["cells"] = tuple(int(v) for v in d["cells"])
        out[name] = d
    return out


def make_trajectory(d):
    from psa_amd import Trajectory
    return Trajectory(d["positions"], d["velocities"], d["types"], d["timesteps"],
                      d["box_matrix"], d["box_lengths"], d["box_tilts"], d["dt_ps"])


def make_calculator(d, **ctor):
    from psa_amd import SEDCalculator
    cx, cy, cz = d["cells"]
    return SEDCalculator(make_trajectory(d), cx, cy, cz, **ctor)


@pytest.fixture(scope="session")
def engine():
    """One GPU context for the whole session (GPU tests only)."""
    from psa_amd import _hip
    eng = _hip.Engine(0)
    yield eng
    eng.close()
