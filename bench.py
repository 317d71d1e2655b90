#!/usr/bin/env python3
"""
SED hot-path benchmark (contract: see the task's bench.py section).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--summation coherent]
                    [--shard auto|k|frames]

One "step" = one pass of the hot path over one synthetic trajectory already resident in HBM:
phase table -> k-projection (split-precision f16 MFMA from the group's cached split planes,
fp32-equivalent) -> batched rocFFT over time -> scale / |.|^2 epilogue -> (N > 1: exchange, see
below) -> transpose to the reference's (T,K,3) layout + SED.intensity, result left on the device.
Default workload is the configuration BASELINE.json's target is quoted on (C3: 32768 atoms x 65536
steps x 256 k-points, [110] path, basis types [1,2]); it fits one MI355X.  The same JSON line
carries `end_to_end` (the public `SEDCalculator.calculate` on the resident trajectory, result as a
host ndarray) and `variants` (the other summation mode: the two basis types as separate groups).

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (children of
this process, before anything touches HIP); under a launcher that already exported WORLD_SIZE
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) it is one of the ranks.
Host rendezvous (RCCL unique id, barriers, max of the wall times) is a small TCP exchange on
MASTER_ADDR:MASTER_PORT+17; the data path is libpsa_hip.so + RCCL over xGMI.  Sharding
(psa_amd/dist.py): "k" = every rank holds the whole trajectory and projects its block of
k-points, gather to rank 0; "frames" = every rank holds 1/N of the frames, projects all k-points
on them, all-to-all before the FFT, gather to rank 0; "auto" = frames when a rank's k-block would be
<= 64 k-points.  Total work is fixed as N grows -> "scaling": "strong".
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL needs it on this pool


def _one_socket_cores():
    """Logical CPUs of ONE socket among those this process may run on, one per physical core:
    the CPU baseline is a 1-socket NumPy run (BASELINE.json north_star)."""
    allowed = sorted(os.sched_getaffinity(0))
    by_socket = {}
    for cpu in allowed:
        base = Path(f"/sys/devices/system/cpu/cpu{cpu}/topology")
        try:
            pkg = int((base / "physical_package_id").read_text())
            core = int((base / "core_id").read_text())
        except (OSError, ValueError):
            pkg, core = 0, cpu
        by_socket.setdefault(pkg, {}).setdefault(core, cpu)
    first = by_socket[min(by_socket)]
    return sorted(first.values()), len(by_socket)


# BLAS threads = physical cores of one socket; must be in the environment before NumPy loads
_SOCKET_CPUS, _N_SOCKETS = _one_socket_cores()
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = str(len(_SOCKET_CPUS))

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:41-42 (dense, v_mfma_f32_32x32x2_f32)
PEAK_16BIT_MFMA_TFLOPS = 2500.0   # ibid. :43 (dense f16 / bf16 MFMA)
PEAK_HBM_GBS = 8000.0             # ibid. :36 (spec; 6.29 TB/s measured copy)
FLOP_PER_UNIT = 12                # 3 components x (re, im) x FMA per (k, t, atom)  
# k-split model for N > 1, mode "k" (one-GPU measurements; the link rate is an assumption)
K1_UNITS_PER_S = 4.3e13           # slope of the planes kernel's time over the k-count
HBM_BOUND_K1_BPS = 6.2e12         # trajectory bytes per second of its HBM-bound small-K variant
XGMI_LINK_BPS = 64e9              # one direction of one xGMI link


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--summation", default="coherent", choices=["coherent", "incoherent"])
    ap.add_argument("--shard", default="auto", choices=["auto", "k", "frames"],
                    help="N > 1: how the calculation is split over the ranks (psa_amd/dist.py)")
    ap.add_argument("--k-points", type=int, default=0,
                    help="override the config's k-point count (diagnostics, e.g. 32 = one rank's shard of C3 on 8 GPUs)")
    ap.add_argument("--k1", default="auto", choices=["auto", "wide", "narrow", "loaderwaves", "eightwaves", "onthefly", "bf16x3", "mfma32"],
                    help="projection kernel: auto = 2xf16 split-precision MFMA from cached split planes (product "
                         "default), onthefly = the same arithmetic splitting in the kernel (no plane cache), "
                         "loaderwaves / eightwaves = auto with the loader-wavefront form of the planes kernel "
                         "(k1_planes_lw.hip: the default for 128-row M blocks) forced on / off, wide / narrow = auto with the 256-row "
                         "workgroup tile (k1_planes_wide.hip, lists of more than 64 k-vectors) forced on / off, "
                         "bf16x3 = 3xbf16 split-precision MFMA, mfma32 = exact-fp32 MFMA")
    ap.add_argument("--even-split", action="store_true",
                    help="N > 1, mode k: give every rank the same number of k-points instead of the root-heavy split")
    ap.add_argument("--no-check", action="store_true",
                    help="N > 1: skip the comparison of the gathered result with a one-GPU recomputation on rank 0")
    ap.add_argument("--check", action="store_true", help="N = 1: no effect (kept for old command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end, variants, shards, first_call (kernel experiments)")
    ap.add_argument("--no-first-call", action="store_true", help="skip the first_call probe (writes a 1.6 GB file)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST ONLY: run the orchestration (spawn, rendezvous, sharding, host exchange, JSON) with an "
                         "engine that computes nothing; the line is marked and carries no measurement")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- self-launch
def _free_port():
    """A port p with p + 17 free as well (the ranks' TCP rendezvous listens on MASTER_PORT + 17)."""
    import socket
    for _ in range(64):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            p = s.getsockname()[1]
        if p + 17 > 65535:
            continue
        try:
            with socket.socket() as s2:
                s2.bind(("127.0.0.1", p + 17))
            return p
        except OSError:
            continue
    return 29500


def spawn_ranks(n: int, argv) -> int:
    """Start the n ranks as children of this process (which has not loaded libpsa_hip, HIP or
    NumPy-side GPU state), relay rank 0's JSON line, return non-zero if any rank fails."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PSA_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import signal

    def _stop(signum, _frame):                  # the launcher was told to stop: no rank outlives it
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, _stop)
    signal.signal(signal.SIGINT, _stop)
    # watch all ranks: the first one that fails takes the others down with it (a rank waiting in the
    # rendezvous for a dead peer would otherwise sit there until its timeout)
    deadline = time.time() + float(os.environ.get("PSA_BENCH_TIMEOUT_S", "3000"))
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.1)
    out = procs[0].stdout.read()
    codes = [p.wait() for p in procs]
    if failed or any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return 1
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    if len(lines) != 1:
        sys.stderr.write(f"bench.py: rank 0 printed {len(lines)} lines instead of one\n")
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


# --------------------------------------------------------------------------- helpers
def k_request(calc, req):
    if req["kind"] == "path":
        mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
        return mags, vecs, None
    r = req["k_ranges"]
    return calc.get_k_grid(req["plane"], (r[0], r[1]), (r[2], r[3]), req["n_kx"], req["n_ky"], 0.0)


def sum_group_atoms(types, kw):
    """Atoms the projection sums over (incoherent: the sum over groups)."""
    import numpy as np
    t = kw.get("basis_atom_types")
    if not t:
        return len(types)
    return int(np.isin(types, t).sum())


def cpu_model():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(spec, tables, r0, types, vecs, kw, n_frames, reps=3):
    """The oracle (NumPy restatement of the reference path, same operations and dtypes) timed on one
    socket of this box over a bounded sample: the first `n_frames` frames, all atoms, all k-points.
    Throughput is per (k, t, atom) unit, so the sample rate is the full-size rate up to the FFT's
    log factor (< 3 % of the reference's time)."""
    import numpy as np
    from oracle import psa_oracle as O
    from psa_amd import synth
    before = os.sched_getaffinity(0)
    os.sched_setaffinity(0, set(_SOCKET_CPUS))
    try:
        vel = np.concatenate([synth.velocities_block(spec, tables, t, min(128, n_frames - t))
                              for t in range(0, n_frames, 128)])
        pos = np.broadcast_to(r0, (n_frames,) + r0.shape)     # static lattice: no second big array
        O.calculate(pos[:8], vel[:8], types, spec.dt_ps, vecs[:2], **kw)       # warm BLAS/threads
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            sed, _, is_complex = O.calculate(pos, vel, types, spec.dt_ps, vecs, **kw)
            inten = O.intensity(sed) if is_complex else sed
            times.append(time.perf_counter() - t0)
        mean = O.mean_positions(pos)
    finally:
        os.sched_setaffinity(0, before)
    n_units = float(sum_group_atoms(types, kw)) * n_frames * len(vecs)
    # the mean the oracle used: a float32 running sum over frames drifts off r0 itself
    # (sed_calculator.py:205); the parity leg must feed the GPU the same numbers
    return n_units, times, vel, inten, mean


class _StubEngine:
    """TEST ONLY (--stub-engine): the Engine interface with no GPU behind it, so that the
    multi-process orchestration can run in a container without one.  Computes nothing."""

    def __init__(self, rank=0):
        import numpy as np
        import threading
        self.np, self.rank, self.nranks, self.slot = np, rank, 1, {}
        self._slab = self._fs = None
        self.lock = threading.RLock()

    def device_info(self):
        return {"name": "stub (no GPU)", "compute_units": 0, "hbm_bytes": 0}

    def adopt(self, slot, array): pass
    def is_resident(self, slot, array): return True

    def calculate(self, slot, mean, kv, groups=None, flags=0, with_intensity=False):
        self.project(slot, mean, kv, groups, flags)
        return self.finalize(self.slot[slot][0], len(kv), bool(flags & 2), True, with_intensity)

    def ensure_resident(self, slot, array): pass

    def set_k1(self, s): pass
    def set_option(self, o, v): pass
    def alloc(self, slot, T, N): self.slot[slot] = (T, N)
    def fill_synthetic(self, *a, **k): pass
    def shape(self, slot): return self.slot[slot]
    def synchronize(self): pass
    def timings(self): return dict.fromkeys(("h2d", "phase", "project", "fft", "epilogue", "gather", "transpose", "d2h"), 0.0)
    def k1_stats(self): return 0, 0.0
    def oneoff_stats(self): return dict.fromkeys(("rocfft_plan", "absmax", "split_planes", "upload"), 0.0)
    def new_unique_id(self): return b"\0" * 128
    def comm_destroy(self): self.nranks = 1
    def close(self): pass

    def comm_init(self, uid, rank, nranks):
        from psa_amd import _hip
        self.rank, self.nranks = rank, nranks
        raise _hip.PsaHipError("stub engine: no RCCL")          # -> the host transport is exercised

    def _ensure(self, T, K, intensity):
        shape = (K, T) if intensity else (K, 3, T)
        if self._slab is None or self._slab.shape != shape:
            self._slab = self.np.zeros(shape, self.np.float32 if intensity else self.np.complex64)

    def project(self, slot, mean, kv, groups=None, flags=0, K_total=None, k_offset=0):
        T = self.slot[slot][0]
        self._ensure(T, len(kv) if K_total is None else K_total, bool(flags & 2))
        self._slab[k_offset:k_offset + len(kv)] = self.rank + 1

    def fs_project(self, slot, mean, kv, idx, flags, T_total, k_offset, k_count):
        self._ensure(T_total, len(kv), bool(flags & 2))
        self._fs = (self.np.full((len(kv), 3, self.slot[slot][0]), self.rank + 1, self.np.complex64), k_offset, k_count, T_total)

    def fs_read(self, k0, nk, T_local): return self._fs[0][k0:k0 + nk]
    def fs_write(self, t0, block): pass
    def fs_finish(self, first): self._slab[self._fs[1]:self._fs[1] + self._fs[2]] = self.rank + 1
    def slab_read(self, row0, nrows, T, intensity): return self._slab[row0:row0 + nrows]
    def slab_write(self, row0, rows): self._slab[row0:row0 + len(rows)] = rows
    def finalize(self, T, K, intensity, fetch=True, with_intensity=False):
        np = self.np                      # (zero-stride stand-ins of the result's shape: nothing is computed)
        out = (np.broadcast_to(np.float32(0), (T, K)) if intensity else np.broadcast_to(np.complex64(0), (T, K, 3))) if fetch else None
        inten = np.broadcast_to(np.float32(0), (T, K)) if fetch and not intensity else None
        return (out, inten) if with_intensity else out


# --------------------------------------------------------------------------- one rank
def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))          # before libpsa_hip / HIP are touched
    import numpy as np
    from psa_amd import _hip, dist, synth
    from psa_amd.core.sed_calculator import SEDCalculator
    from psa_amd.core.trajectory import Trajectory

    # The contract is ONE JSON line on stdout.  RCCL prints banners to fd 1, so the real
    # stdout is set aside and everything else is routed to stderr.
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    exchange = dist.TcpExchange.from_env(timeout_s=600.0) if world > 1 else dist.Exchange()

    spec, req = synth.baseline_spec(args.config)
    if args.k_points and req["kind"] == "path":
        req["n_k"] = args.k_points
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    T, N = spec.n_frames, spec.n_atoms

    if args.stub_engine:
        engine = _StubEngine(rank)
    else:
        _hip.load_library()
        engine = _hip.Engine(local_rank % max(1, _hip.device_count()))
    info = engine.device_info()
    engine.set_k1({"auto": _hip.K1_AUTO, "wide": _hip.K1_AUTO, "narrow": _hip.K1_AUTO, "loaderwaves": _hip.K1_AUTO, "eightwaves": _hip.K1_AUTO, "onthefly": _hip.K1_AUTO,
                   "bf16x3": _hip.K1_SPLIT_BF16, "mfma32": _hip.K1_MFMA32}[args.k1])
    if args.k1 == "onthefly":
        engine.set_option(_hip.OPT_PLANES, 0)
    if args.k1 in ("loaderwaves", "eightwaves"):
        engine.set_option(_hip.OPT_K1_LOADER_WAVES, 1 if args.k1 == "loaderwaves" else 0)
    if args.k1 in ("wide", "narrow"):
        engine.set_option(_hip.OPT_K1_WIDE, 1 if args.k1 == "wide" else 0)
    group = dist.KShardGroup(engine, exchange, gather="root", root=0, mode=args.shard)

    # host objects only for the k generators / group resolution (no big arrays on the host):
    # zero-stride stand-ins of the trajectory's shape
    stand_in = np.broadcast_to(np.float32(0), (T, N, 3))
    pos_stand_in = np.broadcast_to(r0, (T, N, 3))
    traj = Trajectory(pos_stand_in, stand_in, types, np.broadcast_to(np.float32(0), (T,)), box, np.diag(box).copy(),
                      np.zeros(3, np.float32), spec.dt_ps)
    calc = SEDCalculator(traj, *spec.cells)
    _, vecs, grid_shape = k_request(calc, req)
    K = len(vecs)
    kw = {}
    if req.get("basis_atom_types"):
        kw["basis_atom_types"] = req["basis_atom_types"]

    def workload(summation):
        groups = calc._resolve_groups(None, kw.get("basis_atom_types"), summation)
        intensity = summation == "incoherent" and len(groups) > 1
        if not intensity and len(groups) > 1:
            groups = [np.unique(np.concatenate(groups))]
        return groups, calc._device_groups(groups), intensity

    groups, dev_groups, intensity_out = workload(args.summation)
    flags = _hip.F_INTENSITY if intensity_out else 0
    mean_pos = r0                         # mean of a static lattice; positions never leave the host
    n_sum_atoms = sum(len(g) for g in groups)
    mode = group.mode_for(K, T) if world > 1 else "single"
    if mode == "k" and world > 1 and not args.even_split:
        # only rank 0 receives the result: it takes more k-vectors than the ranks that have to ship
        # their rows to it (dist.root_heavy_counts).  Cost model from the one-GPU measurements in
        # K1 time = max(one trajectory pass at the HBM-bound rate,
        # 0.45 of that + n_k * atoms * frames / K1_UNITS_PER_S); one xGMI link per sender.
        floor_s = 12.0 * n_sum_atoms * T / HBM_BOUND_K1_BPS
        group.balance = dict(per_k_s=n_sum_atoms * T / K1_UNITS_PER_S, base_s=0.45 * floor_s, floor_s=floor_s,
                             per_k_bytes=(4.0 if intensity_out else 24.0) * T, link_bytes_per_s=XGMI_LINK_BPS,
                             block_k=64)          # 128-row M blocks of the projection kernel
    # the trajectory, generated in HBM: all of it, or this rank's frames
    t_begin, t_count = (group.my_frames(T) if mode == "frames" else (0, T))
    synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables, t_begin, t_count)

    def step(dev_groups=dev_groups, flags=flags, intensity_out=intensity_out):
        group.project(_hip.SLOT_VELOCITIES, mean_pos, vecs, dev_groups, flags, n_frames=T)
        if group.has_result:
            # transpose to the reference's layout; a complex result leaves its intensity beside it
            engine.finalize(T, K, intensity_out, fetch=False)

    def timed(n_steps, **kwargs):
        exchange.barrier()
        engine.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(**kwargs)
        engine.synchronize()
        exchange.barrier()
        return max(exchange.allgather(time.perf_counter() - t0))

    def measure(n_steps, n_warm, **kwargs):
        """Collective: warm up, time n_steps (contract: barrier + synchronize on both sides, max over
        ranks), return (seconds, per-rank stage ms per step, this rank's K1 launch count and ms)."""
        for _ in range(n_warm):
            step(**kwargs)
        engine.synchronize()
        engine.timings()
        engine.k1_stats()
        seconds = timed(n_steps, **kwargs)
        mine = {k: v / n_steps for k, v in engine.timings().items()}
        n_k1, ms_k1 = engine.k1_stats()
        mine["k1_avg_launch_ms"] = ms_k1 / max(1, n_k1)
        by_rank = exchange.allgather(mine)
        return seconds, by_rank, (n_k1, ms_k1)

    def stage_summary(by_rank):
        return ({k: max(r[k] for r in by_rank) for k in by_rank[0] if k != "k1_avg_launch_ms"},
                [{k: round(v, 4) for k, v in r.items()} for r in by_rank])

    engine.oneoff_stats()
    t0 = time.perf_counter()
    step()
    engine.synchronize()
    first_step_s = time.perf_counter() - t0
    oneoff = engine.oneoff_stats()                               # plan build, magnitude pass, plane build
    elapsed, stages_by_rank, (k1_n, k1_ms) = measure(args.steps, max(0, args.warmup - 1))
    _, k_cnt = group.ranges(K, T)
    real = not args.stub_engine

    # N > 1: the gathered result against a one-GPU recomputation on rank 0 (default on)
    shard_check = None
    if world > 1 and not args.no_check and real:
        gathered = None
        if group.has_result:
            gathered = engine.finalize(T, K, True) if intensity_out else engine.result_intensity(T, K)
        exchange.barrier()
        if rank == 0:
            if mode == "frames":
                synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables)        # now the whole trajectory
            engine.project(_hip.SLOT_VELOCITIES, mean_pos, vecs, dev_groups, flags)
            alone = engine.finalize(T, K, intensity_out)
            alone = alone if intensity_out else np.sum(np.abs(alone) ** 2, axis=-1).astype(np.float32)
            shard_check = float(np.max(np.abs(gathered - alone)) / np.max(np.abs(alone)))
            if mode == "frames":
                synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables, t_begin, t_count)
        exchange.barrier()

    # ------------------------------------------------------------------ variants (collective)
    variants = {}
    extras = not args.no_extras
    v_steps = max(3, args.steps // 2)

    def variant(n_units, seconds, by_rank, k1, what, **more):
        smax, _ = stage_summary(by_rank)
        return dict(what=what, ms_per_step=1e3 * seconds / v_steps, value=0.0 if not real else n_units * v_steps / seconds,
                    k1_launches_per_step=k1[0] / v_steps, k1_avg_launch_ms=k1[1] / max(1, k1[0]),
                    stages_ms_per_step=smax, **more)

    if extras and world == 1:
        # the other summation mode of the same workload
        other = "incoherent" if args.summation == "coherent" else "coherent"
        o_groups, o_dev, o_int = workload(other)
        if (o_int, len(o_groups)) != (intensity_out, len(groups)):
            o_kw = dict(dev_groups=o_dev, flags=_hip.F_INTENSITY if o_int else 0, intensity_out=o_int)
            sec, by_rank, k1 = measure(v_steps, 3, **o_kw)              # (index-list planes are built on the 2nd use)
            variants[other] = variant(float(sum(len(g) for g in o_groups)) * T * K, sec, by_rank, k1,
                                      f"summation_mode='{other}' on the same trajectory", atom_groups=len(o_groups),
                                      output="(T,K) float32 intensity" if o_int else "(T,K,3) complex64 + intensity")
    if extras and world > 1:
        if not intensity_out:
            # the same coherent sum delivered as intensity only: (K,T) float32 rows travel, 6x less than complex
            i_kw = dict(dev_groups=dev_groups, flags=_hip.F_INTENSITY, intensity_out=True)
            sec, by_rank, k1 = measure(v_steps, 2, **i_kw)
            variants["intensity_only"] = variant(float(n_sum_atoms) * T * K, sec, by_rank, k1,
                                                 "the same calculation with the result gathered as (T,K) float32 intensity "
                                                 "(sum_c |S|^2 taken on each rank before the gather: 4 bytes per (k,t) "
                                                 "over the links instead of 24)", shard_mode=mode)
        if mode == "frames":
            # the north star's partitioning (k-points over ranks, every rank holds the whole trajectory) beside
            # the frame sharding that `--shard auto` picked
            synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables)
            group.mode = "k"
            if not args.even_split:
                floor_s = 12.0 * n_sum_atoms * T / HBM_BOUND_K1_BPS
                group.balance = dict(per_k_s=n_sum_atoms * T / K1_UNITS_PER_S, base_s=0.45 * floor_s, floor_s=floor_s,
                                     per_k_bytes=(4.0 if intensity_out else 24.0) * T, link_bytes_per_s=XGMI_LINK_BPS, block_k=64)
            sec, by_rank, k1 = measure(v_steps, 2)
            _, kk_cnt = group.ranges(K, T)
            variants["shard_k"] = variant(float(n_sum_atoms) * T * K, sec, by_rank, k1,
                                          "mode 'k': every rank holds all frames and projects its block of k-points, "
                                          "gather to rank 0 (BASELINE.json north_star's partitioning)",
                                          shard_mode="k", k_points_per_rank=kk_cnt.tolist())
            group.mode, group.balance = args.shard, None
            synth.fill_device(engine, _hip.SLOT_VELOCITIES, spec, tables, t_begin, t_count)

    # ------------------------------------------------------------------ the drop-in call (collective)
    end_to_end = None
    if extras:
        import weakref
        # residency bookkeeping for a trajectory that exists only in HBM: the stand-in array IS the slot
        if mode == "frames":
            view = stand_in[t_begin:t_begin + t_count]
            group._slice = (weakref.ref(stand_in), (_hip.SLOT_VELOCITIES, t_begin, t_count), view)
            engine.adopt(_hip.SLOT_VELOCITIES, view)
        else:
            engine.adopt(_hip.SLOT_VELOCITIES, stand_in)
        if world > 1:
            calc.attach(shard_group=group)
        else:
            calc.attach(engine=engine)
        calc._mean_cache = (weakref.ref(pos_stand_in), mean_pos, _hip.Engine._fingerprint(pos_stand_in))
        call_kw = dict(kw, summation_mode=args.summation)
        mags = np.zeros(K, np.float32)

        def timed_call(n):
            only, both = [], []
            for _ in range(n):
                exchange.barrier()
                t0 = time.perf_counter()
                sed = calc.calculate(mags, vecs, k_grid_shape=grid_shape, **call_kw)
                t1 = time.perf_counter()
                if sed.sed is not None:                  # (N > 1: only the root receives the result)
                    inten = sed.intensity if sed.is_complex else sed.sed      # what cpu_baseline's timed call computes too
                    assert isinstance(sed.sed, np.ndarray) and sed.sed.shape[:2] == (T, K) and inten.shape == (T, K)
                t2 = time.perf_counter()
                only.append(max(exchange.allgather(t1 - t0)))
                both.append(max(exchange.allgather(t2 - t0)))
                sed = inten = None
            return only, both

        timed_call(1)                                          # page-locks the result buffers
        only, both = timed_call(5)
        e2e = float(np.median(both))
        end_to_end = {
            "what": "psa_amd.SEDCalculator.calculate(k_mags, k_vecs, ...) + SED.intensity on the trajectory resident in "
                    "HBM: .sed and the intensity as host ndarrays (D2H into recycled page-locked memory included) -- "
                    "the call BASELINE.md section 2 times on the reference, the one cpu_baseline times"
                    + ("; N > 1: every rank calls it, rank 0 receives the result, slowest rank's wall time" if world > 1 else ""),
            "ms": 1e3 * e2e, "min_ms": 1e3 * min(both), "value": 0.0 if not real else float(n_sum_atoms) * T * K / e2e,
            "unit": "k-points*timesteps*atoms/s",
            "calculate_only_ms": 1e3 * float(np.median(only)),
            "intensity_access_ms": 1e3 * (e2e - float(np.median(only))),
            "result_bytes": (4 if intensity_out else 28) * T * K,
            "note": "the (T,K) float32 intensity is summed on the device in the pass that writes the complex result "
                    "and copied beside it; mean positions of the static lattice are cached per positions array "
                    "(first call: one np.mean pass, as in the reference on every call); first-call costs: first_step "
                    "and first_call in this line"}

    if rank == 0:
        units = float(n_sum_atoms) * T * K
        ms_per_step = 1e3 * elapsed / args.steps
        k_proj = int(group.last_projected_k) if world > 1 else (int(len(_hip.k_pairs(vecs)[1])) if real else K)
        # k-vectors one launch on rank 0 projects (pairs (k, -k) are projected once: PSA_OPT_FOLD_PAIRS)
        k_local = k_proj if (mode == "frames" or world == 1) else int(k_cnt[0])
        t_local = t_count
        # dominant kernel = the projection (K1); algorithmic work of ONE launch on this rank
        n_launch_atoms = float(n_sum_atoms) / len(groups)
        k1_avg_ms = k1_ms / max(1, k1_n)

        def roofline_of(k_launch, t_launch, avg_ms):
            """Roofline of one projection launch over k_launch k-vectors x t_launch frames x n_launch_atoms."""
            per_launch_units = n_launch_atoms * t_launch * k_launch
            flops = FLOP_PER_UNIT * per_launch_units
            algo_bytes = 12.0 * n_launch_atoms * t_launch + 8.0 * k_launch * n_launch_atoms + 24.0 * t_launch * k_launch
            # the library's rule (api_project.hip get_planes / make_geom): "2 x f16" for groups with more than 16
            # k-vectors -- from cached planes ("auto") or splitting in the kernel -- "3 x bf16" below
            if args.k1 == "mfma32":
                products, kernel_name, dtype = 1, "k1_mfma_kernel (k-projection, exact-fp32 MFMA)", "f32"
            elif args.k1 == "bf16x3" or k_launch <= 16:
                products, kernel_name = 6, "k1_split_kernel (k-projection, 3xbf16 split-precision MFMA, fp32-equivalent)"
                dtype = "f32 (3xbf16 split MFMA, fp32 accumulate)"
            else:
                products = 3
                # the library's rules (k1_planes_block_rows, launch_projection): 256-row M blocks where the list fills
                # an even number of 128-row blocks, else 128-row blocks in the loader-wavefront form
                wide = args.k1 in ("auto", "wide") and k_launch > 32 and (-(-2 * k_launch // 128)) % 2 == 0
                lw = not wide and args.k1 in ("auto", "narrow", "wide", "loaderwaves") and k_launch > 32
                kernel_name = ("k1_pair_kernel (k-projection, 2xf16 split-precision MFMA, fp32-equivalent)" if args.k1 == "onthefly"
                               else ("k1_planes_wide_kernel" if wide else "k1_planes_lw_kernel" if lw else "k1_planes_kernel") +
                               " (k-projection from cached split planes, 2xf16 split-precision MFMA, fp32-equivalent"
                               + ("; 256-row x 64-frame workgroup tile, 8 wavefronts)" if wide else
                                  "; 4 loader + 8 compute wavefronts per workgroup)" if lw else ")"))
                dtype = "f32 (2xf16 split MFMA, fp32 accumulate)"
            # matrix-core ceiling for the ALGORITHMIC flop: the fp32 MFMA peak for the exact kernel; for a
            # split kernel the dense 16-bit peak over the MFMA products one fp32 product costs (3 or 6)
            peak_mfma = PEAK_FP32_MFMA_TFLOPS if products == 1 else PEAK_16BIT_MFMA_TFLOPS / products
            t_mfma = flops / (peak_mfma * 1e12)
            t_hbm = algo_bytes / (PEAK_HBM_GBS * 1e9)
            bound = "mfma" if t_mfma >= t_hbm else "hbm"
            k1_s = max(avg_ms, 1e-9) * 1e-3
            if bound == "mfma":
                achieved, peak, unit = flops / k1_s / 1e12, peak_mfma, "TFLOP/s"
            else:
                achieved, peak, unit = algo_bytes / k1_s / 1e9, PEAK_HBM_GBS, "GB/s"
            r = {"kernel": kernel_name, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                 "frac": achieved / peak, "traffic": None, "avg_launch_ms": avg_ms,
                 "algorithmic_flop_per_launch": flops, "algorithmic_bytes_per_launch": algo_bytes,
                 "hbm_frac_if_bytes_bound": (algo_bytes / k1_s / 1e9) / PEAK_HBM_GBS,
                 "vs_fp32_mfma_peak": flops / k1_s / 1e12 / PEAK_FP32_MFMA_TFLOPS}
            if products > 1:
                ex_rate = products * flops / k1_s / 1e12
                r["executed_mfma"] = {"what": f"16-bit MFMA flop actually issued ({products} products per fp32 product)",
                                      "rate": ex_rate, "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s",
                                      "frac": ex_rate / PEAK_16BIT_MFMA_TFLOPS}
            return r, products, dtype

        roof, products, dtype = roofline_of(k_local, t_local, k1_avg_ms)
        roof["launches"] = k1_n
        roof["note"] = ("achieved = algorithmic 12 flop/unit (or algorithmic bytes) of ONE launch on rank 0 over the "
                        "kernel's mean duration (HIP events on the library's stream); mfma peak = "
                        + (f"2500 TFLOP/s dense 16-bit MFMA / {products} MFMA products per fp32-equivalent product"
                           if products > 1 else "157.3 TFLOP/s dense fp32 MFMA")
                        + "; the side whose time at peak is longer is reported as the bound")
        stage_max, stage_list = stage_summary(stages_by_rank)
        out = {
            "metric": "SED throughput (k-points*timesteps*atoms/s)",
            "value": 0.0 if args.stub_engine else units * args.steps / elapsed,
            "unit": "k-points*timesteps*atoms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype, "data": "STUB ENGINE: orchestration test, nothing measured" if args.stub_engine else "synthetic",
            "config": {"workload": f"{args.config}: {N} atoms x {T} steps x {K} k-points, "
                                   f"{req.get('direction', req.get('plane'))} "
                                   f"{'k-path' if req['kind'] == 'path' else 'k-grid'}, {args.summation}"
                                   f"{', basis types ' + str(req['basis_atom_types']) if req.get('basis_atom_types') else ''}",
                       "atoms": N, "timesteps": T, "k_points": K, "atom_groups": len(groups),
                       "k_points_projected": k_proj,
                       "output": ("(T,K) float32 intensity" if intensity_out else "(T,K,3) complex64 + intensity")
                                 + ", left in HBM (end_to_end: on the host)",
                       "parallelism": "single GPU" if world == 1 else
                                      {"k": f"k-shard x{world}: every rank holds all {T} frames, k-points per rank "
                                            f"{k_cnt.tolist()}, gather to rank 0",
                                       "frames": f"frame-shard x{world}: {t_count} frames per rank, all {K} k-points projected "
                                                 f"on them, all-to-all before the FFT (k rows per rank {k_cnt.tolist()}), "
                                                 f"gather to rank 0"}[mode],
                       "shard_mode": mode, "transport": "single" if world == 1 else group.transport,
                       "k_points_per_rank": k_cnt.tolist(), "device": info["name"]},
            "roofline": roof,
            "stages_ms_per_step": stage_max,
            "first_step": {"wall_ms": 1e3 * first_step_s, "rocfft_plan_build_ms": oneoff["rocfft_plan"],
                           "magnitude_pass_ms": oneoff["absmax"], "split_planes_build_ms": oneoff["split_planes"],
                           "note": "first step after the trajectory is in HBM: rocFFT plan (run-time compiled kernels, "
                                   "kept in a per-user cache file for later processes), one largest-magnitude pass and "
                                   "the split-plane build, all cached afterwards"},
        }
        if world > 1:
            out["stages_ms_per_step_by_rank"] = stage_list
            out["stages_note"] = ("stages_ms_per_step = per stage, the slowest rank's HIP-event time per step; "
                                  "stages_ms_per_step_by_rank = every rank's own times (gather = the RCCL exchange "
                                  "as seen by that rank's stream, i.e. including its wait for the peers)")
            if group.transport == "host":
                out["config"]["transport_error"] = getattr(group, "transport_error", None)
                out["config"]["transport_note"] = ("RCCL communicator unavailable: slab rows travelled D2H -> TCP rendezvous "
                                                   "on rank 0 -> H2D; the timings are NOT xGMI numbers")
        if shard_check is not None:
            out["shard_check_max_rel"] = shard_check
        # HBM-side traffic of K1 from the committed PMC passes of this round (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs; FETCH_SIZE doubled for 16-byte/lane streaming reads as
        # MI355X_MICROARCH.md prescribes; counts L2->fabric requests, Infinity-Cache hits included)
        # -- only when this run is the profiled workload
        pmcs = sorted((ROOT / "profiles").glob("r*_C3_pmc_fetch_write.json"))
        if (args.config == "C3" and K == 256 and world == 1 and not intensity_out and args.k1 == "auto" and pmcs):
            summ = json.loads(pmcs[-1].read_text())["k1_summary"]
            out["roofline"]["traffic"] = summ["fetch_bytes_corrected_x2"] + summ["write_bytes"]
            out["roofline"]["traffic_source"] = (f"profiles/{pmcs[-1].name} (PMC passes of the same command, per launch; "
                                                 "a committed measurement, not taken in this run)")
        if variants:
            out["variants"] = variants
        if end_to_end:
            out["end_to_end"] = end_to_end

        solo = world == 1 and real and extras
        if solo and req["kind"] == "path" and K >= 64 and args.k1 in ("auto", "loaderwaves", "eightwaves"):
            # ---- HBM-bound shapes: one rank's share of this k-path under k-sharding on 4 / 8 GPUs ----------
            shards = {}
            for k_dev in (64, 32):
                sub = vecs[:k_dev]
                for _ in range(2):
                    engine.project(_hip.SLOT_VELOCITIES, mean_pos, sub, dev_groups, flags)
                engine.synchronize()
                engine.k1_stats()
                for _ in range(5):
                    engine.project(_hip.SLOT_VELOCITIES, mean_pos, sub, dev_groups, flags)
                engine.synchronize()
                n_l, ms_l = engine.k1_stats()
                r, _, _ = roofline_of(k_dev, T, ms_l / max(1, n_l))
                shards[f"k_dev_{k_dev}"] = {"k_points": k_dev, "avg_launch_ms": r["avg_launch_ms"], "bound": r["bound"],
                                            "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                                            "hbm_frac_if_bytes_bound": r["hbm_frac_if_bytes_bound"], "launches": n_l}
            out["shards"] = dict(shards, what=f"the projection kernel on the first 64 / 32 k-points of this path against the "
                                              f"same resident trajectory = one rank's launch under k-sharding on "
                                              f"{K // 64} / {K // 32} GPUs; algorithmic bytes (trajectory once + phase table "
                                              f"+ q) over the mean launch time vs 8 TB/s")
        if solo and not args.no_first_call:
            try:                                        # (needs 1.6 GB of scratch disk: never at the price of the line)
                out["first_call"] = first_call_probe(engine)
            except Exception as err:                    # noqa: BLE001
                out["first_call"] = {"error": f"{type(err).__name__}: {err}"}
        if not args.no_cpu_baseline and real:
            n_frames = args.cpu_frames or int(min(T, max(64, 2 ** int(np.log2(1.0e11 / (n_sum_atoms * K))))))
            call_kw = dict(kw, summation_mode=args.summation)
            n_units, times, vel, ref_int, mean_sample = cpu_baseline(spec, tables, r0, types, vecs, call_kw, n_frames)
            # parity of the HIP path on the very same sample (rank 0's GPU alone)
            engine.ensure_resident(_hip.SLOT_VELOCITIES, vel)
            engine.project(_hip.SLOT_VELOCITIES, mean_sample, vecs, dev_groups, flags)
            got = engine.finalize(n_frames, K, intensity_out)
            got_int = got if intensity_out else np.sum(np.abs(got) ** 2, axis=-1).astype(np.float32)
            err = float(np.max(np.abs(got_int - ref_int)) / np.max(np.abs(ref_int)))
            rate = n_units / min(times)
            gpu_rate = out.get("end_to_end", {}).get("value", out["value"])
            out["cpu_baseline"] = {
                "value": rate, "unit": "k-points*timesteps*atoms/s", "cores": len(_SOCKET_CPUS), "kind": "port",
                "median_value": n_units / float(np.median(times)), "reps": len(times),
                "seconds": {"min": min(times), "median": float(np.median(times))},
                "cpu_model": cpu_model(), "sockets_visible": _N_SOCKETS,
                "pinning": f"os.sched_setaffinity to the {len(_SOCKET_CPUS)} physical cores of one socket visible to this "
                           f"process; OPENBLAS/OMP/MKL_NUM_THREADS={len(_SOCKET_CPUS)} set before NumPy loaded"
                           + ("; timed on rank 0 while the other ranks wait" if world > 1 else ""),
                "sample": f"first {n_frames} of {T} frames, all {N} atoms, all {K} k-points "
                          f"(oracle/psa_oracle.py: NumPy einsum+pocketfft restatement of the reference path), "
                          f"calculate + intensity",
                "speedup_vs_cpu": gpu_rate / rate,
                "speedup_basis": "end_to_end.value (calculate + intensity, results on the host, like the CPU path's timed call)"
                                 if "end_to_end" in out else "value (result left in HBM)",
                "parity_max_rel_intensity_on_sample": err}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    exchange.barrier()
    group.close()
    engine.close()


def first_call_probe(engine):
    """First `calculate` on a trajectory that is NOT in HBM yet: a configuration-2-sized .npy cache
    (8192 atoms x 16384 frames = 1.6 GB), memory-mapped like the reference's loader leaves it
    (io/loader.py:48-79).  Upload alone vs upload + projection + FFT + epilogue + D2H (psa_sed_project_upload
    projects each chunk behind its copy)."""
    import tempfile
    import numpy as np
    from psa_amd import SEDCalculator, Trajectory, _hip, synth
    spec, req = synth.baseline_spec("C2")
    r0, types, box = synth.lattice(spec.cells)
    tables = synth.mode_tables(spec, r0)
    with tempfile.TemporaryDirectory(dir=os.environ.get("PSA_BENCH_TMP")) as tmp:
        path = Path(tmp) / "run.velocities.npy"
        arr = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(spec.n_frames, spec.n_atoms, 3))
        for t in range(0, spec.n_frames, 512):
            arr[t:t + 512] = synth.velocities_block(spec, tables, t, 512)
        arr.flush()
        del arr
        vel = np.load(path, mmap_mode="r")
        pos = np.broadcast_to(r0, vel.shape)
        tr = Trajectory(pos, vel, types, np.arange(spec.n_frames, dtype=np.float32), box, np.diag(box).copy(),
                        np.zeros(3, np.float32), spec.dt_ps)
        calc = SEDCalculator(tr, *spec.cells).attach(engine=engine)
        mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
        calc._mean_positions()
        engine.invalidate()
        engine.ensure_resident(_hip.SLOT_VELOCITIES, vel)           # page cache + staging buffers warm
        plain, first = [], []
        for _ in range(3):
            engine.invalidate()
            t0 = time.perf_counter()
            engine.ensure_resident(_hip.SLOT_VELOCITIES, vel)
            plain.append(time.perf_counter() - t0)
        for _ in range(3):
            engine.invalidate()
            t0 = time.perf_counter()
            sed = calc.calculate(mags, vecs)
            first.append(time.perf_counter() - t0)
            del sed
        t0 = time.perf_counter()
        sed = calc.calculate(mags, vecs)
        resident = time.perf_counter() - t0
        del sed, vel
        engine.invalidate()
    return {"what": f"SEDCalculator.calculate on a memory-mapped .npy cache of {spec.n_atoms} atoms x {spec.n_frames} frames "
                    f"({12 * spec.n_atoms * spec.n_frames / 1e9:.2f} GB) x {len(vecs)} k-points that is not in HBM: upload "
                    "through the page-locked staging pipeline with each chunk's frames projected behind its copy",
            "upload_alone_ms": 1e3 * min(plain), "upload_GBps": 12e-9 * spec.n_atoms * spec.n_frames / min(plain),
            "first_calculate_ms": 1e3 * first[0], "later_first_calculates_ms": 1e3 * min(first[1:]),
            "ratio_to_upload": min(first[1:]) / min(plain), "resident_calculate_ms": 1e3 * resident,
            "note": "first_calculate_ms also builds the rocFFT plan of this length beside the upload"}


if __name__ == "__main__":
    main()
