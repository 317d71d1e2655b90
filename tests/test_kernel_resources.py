"""Register budgets the planes kernels rely on, checked at build time (hipcc cross-compiles gfx950 here): they wait
for their LDS-DMA with a COUNTED `s_waitcnt vmcnt(N)`, and scratch loads / stores are vector-memory operations that
count too -- a kernel that spills a single register inside its loop would read LDS before the data has landed.  The
256-row kernel sits at 250 of the 256 VGPRs two wavefronts per SIMD may have, so a compiler update can tip it."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SRC = ROOT / "psa_amd" / "csrc"


def _flags():
    line = next(ln for ln in (SRC / "Makefile").read_text().splitlines() if ln.startswith("CXXFLAGS"))
    cont = (SRC / "Makefile").read_text().split(line)[1].splitlines()[1]
    raw = (line.split(":=")[1].rstrip("\\") + " " + cont).split()
    return [f.replace("$(ARCH)", "gfx950").replace("$(ROOT)", str(ROOT)) for f in raw if not f.startswith("-W")]


@pytest.mark.parametrize("source, max_vgprs", [("k1_planes_wide.hip", 256), ("k1_planes_lw.hip", 168)])
def test_counted_vmcnt_kernels_use_no_scratch(source, max_vgprs, tmp_path):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    res = subprocess.run([HIPCC, *_flags(), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", str(SRC / source),
                          "-o", str(tmp_path / "k.s")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", res.stderr)]
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", res.stderr)]
    vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", res.stderr)]
    assert scratch and all(s == 0 for s in scratch), res.stderr[-1500:]
    assert all(s == 0 for s in spills) and all(v <= max_vgprs for v in vgprs), (spills, vgprs)
    assert "scratch_" not in (tmp_path / "k.s").read_text()
