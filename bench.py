#!/usr/bin/env python3
"""
SED hot-path benchmark (contract: see the task's bench.py section).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--summation coherent]

One "step" = one pass of the hot path over one synthetic trajectory already resident in
HBM: phase table -> fp32-MFMA k-projection -> batched rocFFT over time -> scale/|.|^2
epilogue -> (N>1: RCCL gather of the k-shards to rank 0) -> transpose to the reference's
(T,K,3) layout + SED.intensity, result left on the device.  Default workload is the
configuration BASELINE.json's target is quoted on (C3: 32768 atoms x 65536 steps x 256
k-points, [110] path, basis types [1,2]); it fits one MI355X.  With N GPUs the 256
k-points are sharded over the ranks (total work fixed -> "strong" scaling), each rank holding
its own copy of the trajectory.

Launch for N>1 (one process per GPU):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
torch.distributed (gloo) is used only as the host rendezvous: unique-id broadcast, barriers,
max-over-ranks of the wall time.  The data path is libpsa_hip.so + RCCL.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL needs it on this pool

from psa_amd import _hip, dist, synth                      # noqa: E402
from psa_amd.core.sed_calculator import SEDCalculator      # noqa: E402
from psa_amd.core.trajectory import Trajectory             # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:41-42 (dense, v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # ibid. :43 (dense bf16 MFMA)
SPLIT_MFMA_FLOP_PER_UNIT = {"auto": 36, "bf16x3": 72}   # 16-bit products issued per fp32 product: 3 (2xf16) / 6 (3xbf16)
PEAK_HBM_GBS = 8000.0             # ibid. :36 (spec; 6.29 TB/s measured copy)
FLOP_PER_UNIT = 12                # 3 components x (re, im) x FMA per (k, t, atom)  
# k-split model for N > 1 (measured on one MI355X; the link rate is an assumption)
K1_UNITS_PER_S = 3.4e13           # slope of the f16 projection kernel's time over the k-count
HBM_BOUND_K1_BPS = 5.3e12         # trajectory bytes per second of its HBM-bound small-K variant
XGMI_LINK_BPS = 64e9              # one direction of one xGMI link


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--summation", default="coherent", choices=["coherent", "incoherent"])
    ap.add_argument("--k-points", type=int, default=0,
                    help="override the config's k-point count (diagnostics, e.g. 32 = one rank's shard of C3 on 8 GPUs)")
    ap.add_argument("--k1", default="auto", choices=["auto", "bf16x3", "mfma32"],
                    help="projection kernel: auto = split-precision 2xf16 MFMA (product default), "
                         "bf16x3 = split-precision 3xbf16 MFMA, mfma32 = exact-fp32 MFMA")
    ap.add_argument("--even-split", action="store_true",
                    help="N > 1: give every rank the same number of k-points instead of the root-heavy split")
    ap.add_argument("--check", action="store_true",
                    help="after timing, rank 0 recomputes every k-point on its own GPU and compares the "
                         "gathered result with it (multi-rank plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def k_request(calc, req):
    if req["kind"] == "path":
        mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
        return mags, vecs, None
    r = req["k_ranges"]
    return calc.get_k_grid(req["plane"], (r[0], r[1]), (r[2], r[3]), req["n_kx"], req["n_ky"], 0.0)


def cpu_baseline(spec, tables, r0, types, vecs, kw, n_frames):
    """The oracle (NumPy restatement of the reference path, same operations and dtypes) timed
    on this box's host cores over a bounded sample: the first `n_frames` frames, all atoms,
    all k-points.  Throughput is per (k, t, atom) unit, so the sample rate is the full-size
    rate up to the FFT's log factor (< 3 % of the reference's time)."""
    from oracle import psa_oracle as O
    vel = np.concatenate([synth.velocities_block(spec, tables, t, min(128, n_frames - t))
                          for t in range(0, n_frames, 128)])
    pos = np.broadcast_to(r0, (n_frames,) + r0.shape)     # static lattice: no second big array
    O.calculate(pos[:8], vel[:8], types, spec.dt_ps, vecs[:2], **kw)       # warm BLAS/threads
    best = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        sed, _, is_complex = O.calculate(pos, vel, types, spec.dt_ps, vecs, **kw)
        inten = O.intensity(sed) if is_complex else sed
        best = min(best, time.perf_counter() - t0)
    n_units = sum_group_atoms(types, kw) * n_frames * len(vecs)
    # the mean the oracle used: a float32 running sum over frames drifts off r0 itself
    # (sed_calculator.py:205); the parity leg must feed the GPU the same numbers
    return n_units / best, best, vel, inten, O.mean_positions(pos)


def sum_group_atoms(types, kw):
    """Atoms the projection sums over (incoherent: the sum over groups)."""
    t = kw.get("basis_atom_types")
    if not t:
        return len(types)
    return int(np.isin(types, t).sum())


def main():
    args = parse_args()
    # The contract is ONE JSON line on stdout.  Gloo and RCCL print banners to fd 1, so the real
    # stdout is set aside and everything else is routed to stderr.
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # libpsa_hip (and with it /opt/rocm's HIP runtime, rocFFT, RCCL) is loaded BEFORE torch so that
    # single- and multi-process runs execute the very same libraries; torch is only used for its
    # gloo rendezvous below.
    _hip.load_library()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                     f"python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py")
        args.gpus = world

    exchange = dist.Exchange()
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("gloo")
        exchange = dist.TorchExchange()

    spec, req = synth.baseline_spec(args.config)
    if args.k_points and req["kind"] == "path":
        req["n_k"] = args.k_points
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    T, N = spec.n_frames, spec.n_atoms

    engine = _hip.Engine(local_rank % max(1, _hip.device_count()))
    info = engine.device_info()
    engine.set_k1({"auto": _hip.K1_AUTO, "bf16x3": _hip.K1_SPLIT_BF16, "mfma32": _hip.K1_MFMA32}[args.k1])
    synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables)     # V generated in HBM
    group = dist.KShardGroup(engine, exchange, gather="root", root=0)

    # host objects only for the k generators / group resolution (no big arrays on the host)
    stub = np.zeros((1, N, 3), np.float32)
    traj = Trajectory(stub, stub, types, np.zeros(1, np.float32), box, np.diag(box).copy(),
                      np.zeros(3, np.float32), spec.dt_ps)
    calc = SEDCalculator(traj, *spec.cells)
    _, vecs, grid_shape = k_request(calc, req)
    K = len(vecs)
    kw = {}
    if req.get("basis_atom_types"):
        kw["basis_atom_types"] = req["basis_atom_types"]
    kw["summation_mode"] = args.summation
    groups = calc._resolve_groups(None, kw.get("basis_atom_types"), args.summation)
    intensity_out = args.summation == "incoherent" and len(groups) > 1
    if not intensity_out and len(groups) > 1:
        groups = [np.unique(np.concatenate(groups))]
    dev_groups = calc._device_groups(groups)
    flags = _hip.F_INTENSITY if intensity_out else 0
    mean_pos = r0                         # mean of a static lattice; positions never leave the host
    n_sum_atoms = sum(len(g) for g in groups)
    if world > 1 and not args.even_split:
        # only rank 0 receives the result: it takes more k-vectors than the ranks that have to ship
        # their rows to it (dist.root_heavy_counts).  Cost model from the one-GPU measurements in
        # K1 time = max(one trajectory pass at the HBM-bound rate,
        # 0.45 of that + n_k * atoms * frames / 3.4e13 units/s); one xGMI link per sender.
        floor_s = 12.0 * n_sum_atoms * T / HBM_BOUND_K1_BPS
        group.balance = dict(per_k_s=n_sum_atoms * T / K1_UNITS_PER_S, base_s=0.45 * floor_s, floor_s=floor_s,
                             per_k_bytes=(4.0 if intensity_out else 24.0) * T, link_bytes_per_s=XGMI_LINK_BPS,
                             block_k=64)          # 128-row M blocks of the projection kernel

    def step():
        group.project(_hip.SLOT_VELOCITIES, mean_pos, vecs, dev_groups, flags)
        if group.has_result:
            engine.finalize(T, K, intensity_out, fetch=False)
            if not intensity_out:
                engine._lib.psa_result_intensity(engine._h, None, 0)

    for _ in range(args.warmup):
        step()
    engine.synchronize()
    engine.timings()
    engine.k1_stats()
    exchange.barrier()
    engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    engine.synchronize()
    exchange.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max(exchange.allgather(elapsed))
    stages = engine.timings()
    k1_n, k1_ms = engine.k1_stats()

    if rank == 0:
        units = float(n_sum_atoms) * T * K
        ms_per_step = 1e3 * elapsed / args.steps
        k_local = group.my_range(K)[1]
        # dominant kernel = the projection (K1); algorithmic work of ONE launch on this rank
        per_launch_units = (float(n_sum_atoms) / len(groups)) * T * k_local
        k1_avg_ms = k1_ms / max(1, k1_n)
        flops = FLOP_PER_UNIT * per_launch_units
        algo_bytes = 12.0 * (n_sum_atoms / len(groups)) * T + 8.0 * k_local * (n_sum_atoms / len(groups)) \
            + 24.0 * T * k_local
        split = args.k1 != "mfma32"          # every velocity-mode group runs a split kernel
        # the library's own rule (api.hip make_geom): "2 x f16" for velocity-mode groups with more
        # than 16 k-vectors on this rank, "3 x bf16" for short k-lists
        f16 = args.k1 == "auto" and k_local > 16
        split_name = "2xf16" if f16 else "3xbf16" if split else ""
        # matrix-core ceiling for the ALGORITHMIC flop: the fp32 MFMA peak for the exact kernel; for a
        # split kernel the dense 16-bit peak over the MFMA products one fp32 product costs (3 or 6)
        products = SPLIT_MFMA_FLOP_PER_UNIT["auto" if f16 else "bf16x3"] // FLOP_PER_UNIT if split else 1
        peak_mfma = PEAK_BF16_MFMA_TFLOPS / products if split else PEAK_FP32_MFMA_TFLOPS
        t_mfma = flops / (peak_mfma * 1e12)
        t_hbm = algo_bytes / (PEAK_HBM_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        if bound == "mfma":
            achieved, peak, unit = flops / (k1_avg_ms * 1e-3) / 1e12, peak_mfma, "TFLOP/s"
        else:
            achieved, peak, unit = algo_bytes / (k1_avg_ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        kernel_name = ("k1_pair_kernel (k-projection, 2xf16 split-precision MFMA, fp32-equivalent)" if f16
                       else f"k1_split_kernel (k-projection, 3xbf16 split-precision MFMA, fp32-equivalent)" if split
                       else "k1_mfma_kernel (k-projection, exact-fp32 MFMA)")
        roof_note = ("achieved = algorithmic 12 flop/unit (or algorithmic bytes) over the kernel time; mfma peak = "
                     + (f"2500 TFLOP/s dense 16-bit MFMA / {products} products per fp32-equivalent product"
                        if split else "157.3 TFLOP/s dense fp32 MFMA")
                     + "; the side whose time at peak is longer is reported as the bound")
        executed = None
        if split:
            issued = SPLIT_MFMA_FLOP_PER_UNIT["auto" if split_name == "2xf16" else "bf16x3"]
            ex_rate = issued * per_launch_units / (k1_avg_ms * 1e-3) / 1e12
            executed = {"what": f"16-bit MFMA flop actually issued ({issued // 12} "
                                f"products per fp32 product)",
                        "rate": ex_rate, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ex_rate / PEAK_BF16_MFMA_TFLOPS}
        out = {
            "metric": "SED throughput (k-points*timesteps*atoms/s)",
            "value": units * args.steps / elapsed,
            "unit": "k-points*timesteps*atoms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {N} atoms x {T} steps x {K} k-points, "
                                   f"{req.get('direction', req.get('plane'))} "
                                   f"{'k-path' if req['kind'] == 'path' else 'k-grid'}, {args.summation}"
                                   f"{', basis types ' + str(req['basis_atom_types']) if req.get('basis_atom_types') else ''}",
                       "atoms": N, "timesteps": T, "k_points": K, "atom_groups": len(groups),
                       "output": "(T,K) float32 intensity" if intensity_out else "(T,K,3) complex64 + intensity",
                       "parallelism": (f"k-shard x{world} ({'RCCL' if group.transport == 'rccl' else 'HOST-STAGED (RCCL unavailable)'}"
                                       f" gather to rank 0; k-points per rank {group.ranges(K)[1].tolist()})") if world > 1 else "single GPU",
                       "device": info["name"]},
            "roofline": {"kernel": kernel_name, "bound": bound,
                         "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                         "traffic": None, "avg_launch_ms": k1_avg_ms, "launches": k1_n,
                         "algorithmic_flop_per_launch": flops, "algorithmic_bytes_per_launch": algo_bytes,
                         "hbm_frac_if_bytes_bound": (algo_bytes / (k1_avg_ms * 1e-3) / 1e9) / PEAK_HBM_GBS,
                         "vs_fp32_mfma_peak": flops / (k1_avg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "note": roof_note},
            "stages_ms_per_step": {k: v / args.steps for k, v in stages.items()},
        }
        if executed:
            out["roofline"]["executed_mfma"] = executed
        out["dtype"] = f"f32 ({split_name} split MFMA, fp32 accumulate)" if split else "f32"
        # HBM-side traffic of K1 from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs; FETCH_SIZE doubled for 16-byte/lane streaming reads as
        # MI355X_MICROARCH.md prescribes; counts L2->fabric requests, Infinity-Cache hits included)
        # -- only when this run is the profiled workload
        pmc = ROOT / "profiles" / ("r1g_C3_pmc_fetch_write.json" if f16 else "r1d_C3_pmc_fetch_write.json" if split
                                   else "r1b_C3_pmc_fetch_write.json")
        if args.config == "C3" and K == 256 and world == 1 and not intensity_out and pmc.exists():
            summ = json.loads(pmc.read_text())["k1_summary"]
            out["roofline"]["traffic"] = summ["fetch_bytes_corrected_x2"] + summ["write_bytes"]
            out["roofline"]["traffic_source"] = f"profiles/{pmc.name} (PMC, per launch)"
        if world == 1 and not args.no_cpu_baseline:
            cores = len(os.sched_getaffinity(0))
            try:                                   # threads the BLAS under NumPy actually runs
                from threadpoolctl import threadpool_info
                blas = [i["num_threads"] for i in threadpool_info() if i.get("user_api") == "blas"]
                if blas:
                    cores = min(cores, max(blas))
            except Exception:
                pass
            n_frames = args.cpu_frames or int(min(T, max(64, 2 ** int(np.log2(1.0e11 / (n_sum_atoms * K))))))
            rate, secs, vel, ref_int, mean_sample = cpu_baseline(spec, tables, r0, types, vecs, kw, n_frames)
            # parity of the HIP path on the very same sample
            engine.ensure_resident(_hip.SLOT_VELOCITIES, vel)
            engine.project(_hip.SLOT_VELOCITIES, mean_sample, vecs, dev_groups, flags)
            got = engine.finalize(n_frames, K, intensity_out)
            got_int = got if intensity_out else np.sum(np.abs(got) ** 2, axis=-1).astype(np.float32)
            err = float(np.max(np.abs(got_int - ref_int)) / np.max(np.abs(ref_int)))
            out["cpu_baseline"] = {
                "value": rate, "unit": "k-points*timesteps*atoms/s", "cores": cores, "kind": "port",
                "sample": f"first {n_frames} of {T} frames, all {N} atoms, all {K} k-points "
                          f"(oracle/psa_oracle.py: NumPy einsum+pocketfft restatement, best of 2, {secs:.1f} s)",
                "speedup_vs_cpu": units * args.steps / elapsed / rate,
                "parity_max_rel_intensity_on_sample": err}
        if args.check and group.has_result:
            gathered = engine.result_intensity(T, K) if not intensity_out else engine.finalize(T, K, True)
            engine.project(_hip.SLOT_VELOCITIES, mean_pos, vecs, dev_groups, flags)
            alone = engine.finalize(T, K, intensity_out)
            alone = alone if intensity_out else np.sum(np.abs(alone) ** 2, axis=-1).astype(np.float32)
            out["shard_check_max_rel"] = float(np.max(np.abs(gathered - alone)) / np.max(np.abs(alone)))
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    group.close()
    engine.close()
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
