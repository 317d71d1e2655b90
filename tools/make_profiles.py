#!/usr/bin/env python3
"""Assemble the round's committed profile summaries under profiles/ from what the GPU runs left in
gpurun_out/ (scratch):
    tools/make_profiles.py r2
expects  gpurun_out/prof_<tag>/**/_kernel_stats.csv + prof_<tag>_bench.json   (rocprofv3 --kernel-trace --stats)
         gpurun_out/pmc_<tag>.json, gpurun_out/pmc_<tag>k32.json              (tools/pmc_k1.sh)
"""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
out, scratch = ROOT / "profiles", ROOT / "gpurun_out"

stats = sorted((scratch / f"prof_{tag}").rglob("*kernel_stats.csv"), key=lambda p: p.stat().st_mtime)
if stats:
    shutil.copy(stats[-1], out / f"{tag}_C3_kernel_stats.csv")
    shutil.copy(scratch / f"prof_{tag}_bench.json", out / f"{tag}_C3_bench.json")

NOTE = ("gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by exactly 2x (MI355X_MICROARCH.md, HBM section): "
        "K1 streams its operands with 16-byte LDS-DMA pieces, so its read traffic is 2 x FETCH_SIZE; WRITE_SIZE is exact.  "
        "Both counters are in KiB per dispatch.  Counts L2->fabric requests (Infinity-Cache hits included).")


def k1(pmc):
    name = next(k for k in pmc if "k1_" in k)
    return name, pmc[name]


def fetch_write(src, dst, workload, n_atoms, n_frames, n_k):
    name, c = k1(json.loads(src.read_text()))
    raw = c["FETCH_SIZE"] * 1024.0
    algo = 12 * n_atoms * n_frames + 8 * n_k * n_atoms + 24 * n_frames * n_k
    dst.write_text(json.dumps({
        "command": f"tools/pmc_k1.sh {src.stem[4:]}  (rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 bench.py "
                   f"--steps 2 --warmup 1 --no-cpu-baseline --no-extras; FETCH_SIZE and WRITE_SIZE in passes of their own; {workload})",
        "kernel": name, "unit": "bytes per dispatch", "note": NOTE,
        "k1_summary": {"fetch_bytes_raw": raw, "fetch_bytes_corrected_x2": 2 * raw, "write_bytes": c["WRITE_SIZE"] * 1024.0,
                       "algorithmic_bytes": algo, "traffic_over_algorithmic": (2 * raw + c["WRITE_SIZE"] * 1024.0) / algo,
                       "kernel_ms_under_profiler": c.get("kernel_ms")}}, indent=1) + "\n")
    return name, c


src = scratch / f"pmc_{tag}.json"
if src.exists():
    name, c = fetch_write(src, out / f"{tag}_C3_pmc_fetch_write.json", "configuration 3, 256 k-points", 32768, 65536, 256)
    keep = {k: v for k, v in c.items() if k.startswith(("SQ_", "GRBM_")) or k in ("kernel_ms", "effective_clock_GHz")}
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["kernel_ms"] * 1e-3 * c["effective_clock_GHz"] * 1e9 * 1024)
    (out / f"{tag}_C3_pmc_sq.json").write_text(json.dumps({
        "command": f"tools/pmc_k1.sh {tag}  (one rocprofv3 --pmc pass per counter group; configuration 3)",
        "kernel": f"{name}, 32768 atoms x 65536 frames x 256 k-points",
        "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; WAIT_ANY = parked in "
                "s_waitcnt / s_barrier, WAIT_INST_ANY = stalled at issue, ACTIVE_INST_ANY = issuing (they partition WAVE_CYCLES).  "
                "Per 32-atom stage and wavefront: divide by 32768 wavefronts x 1024 stages.",
        "counters": keep,
        "mfma_pipe_busy_fraction": busy,
        "mfma_pipe_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x effective clock x 1024 SIMDs); effective clock = "
                               "GRBM_GUI_ACTIVE / 8 / kernel time (under the profiler, which clocks lower than a plain run)"},
        indent=1) + "\n")
for n_k, share in ((32, 8), (64, 4)):
    src = scratch / f"pmc_{tag}k{n_k}.json"
    if src.exists():
        fetch_write(src, out / f"{tag}_K{n_k}_pmc_fetch_write.json",
                    f"configuration 3's trajectory, {n_k} k-points = one rank's k-shard of {share}", 32768, 65536, n_k)
stats = sorted((scratch / f"prof_{tag}_default").rglob("*kernel_stats.csv"), key=lambda p: p.stat().st_mtime)
if stats:
    shutil.copy(stats[-1], out / f"{tag}_C3_default_run_kernel_stats.csv")
    shutil.copy(scratch / f"prof_{tag}_default_bench.json", out / f"{tag}_C3_default_run_bench.json")
print("\n".join(sorted(p.name for p in out.glob(f"{tag}_*"))))
