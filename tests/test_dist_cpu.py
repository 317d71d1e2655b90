"""Multi-process tests of the k-shard path on CPU: world_size-2 `gloo` process groups (and
the dependency-free TCP rendezvous) drive the product's KShardGroup / SEDCalculator host
logic; the GPU engine is replaced by the oracle-backed test double whose "RCCL gather" moves
slab rows through the same exchange."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = Path(__file__).resolve().parent
for p in (str(HERE.parent), str(HERE), str(HERE / "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from psa_amd import dist            # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_k, nranks", [(256, 8), (2500, 8), (5, 8), (7, 2), (0, 3), (1, 1), (128, 4)])
def test_shard_ranges_partition(n_k, nranks):
    off, cnt = dist.shard_ranges(n_k, nranks)
    assert off.dtype == np.int64 and cnt.dtype == np.int64 and len(off) == nranks
    assert cnt.sum() == n_k and cnt.max() - cnt.min() <= 1
    assert off[0] == 0 and np.array_equal(off[1:], np.cumsum(cnt)[:-1])      # contiguous, ordered
    with pytest.raises(ValueError):
        dist.shard_ranges(4, 0)


MODEL = dict(per_k_s=1.0, base_s=0.5, floor_s=2.0, per_k_bytes=4.0, link_bytes_per_s=1.0)   # sending costs 4 per row


@pytest.mark.parametrize("n_k, nranks, root", [(256, 2, 0), (256, 4, 0), (256, 8, 3), (7, 2, 0), (5, 8, 0), (0, 2, 1), (9, 1, 0)])
def test_root_heavy_counts(n_k, nranks, root):
    cnt = dist.root_heavy_counts(n_k, nranks, root, **MODEL)
    off, cnt2 = dist.shard_ranges(n_k, nranks, cnt)
    assert cnt.sum() == n_k and np.array_equal(cnt, cnt2) and cnt.min() >= 0
    others = np.delete(cnt, root)
    assert (others == others[0]).all() if nranks > 1 else True
    if nranks > 1:
        assert cnt[root] >= others[0]                     # the root never gets less than the others
        even = n_k // nranks

        def finish(c_root, c_other):
            comp = lambda n: max(MODEL["floor_s"], MODEL["base_s"] + MODEL["per_k_s"] * n) if n else 0.0
            return max(comp(c_root), comp(c_other) + 4.0 * c_other)
        assert finish(cnt[root], others[0]) <= finish(n_k - even * (nranks - 1), even) + 1e-12
    with pytest.raises(ValueError):
        dist.shard_ranges(10, 2, [5, 4])


def test_root_heavy_counts_with_free_links_is_the_even_split():
    cnt = dist.root_heavy_counts(256, 4, 0, per_k_s=1.0, base_s=0.0, floor_s=0.0, per_k_bytes=0.0, link_bytes_per_s=1.0)
    assert list(cnt) == [64, 64, 64, 64]


def _sharded_worker(rank, world, port, backend, gather, results, no_rccl=False, balance=None, mode="k"):
    """One rank of a 2-process sharded `calculate` (oracle-backed engine)."""
    import numpy as np
    import conftest
    from oracle_engine import OracleEngine
    from psa_amd import dist as D

    if backend == "gloo":
        import torch.distributed as td
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                          WORLD_SIZE=str(world))
        td.init_process_group("gloo", rank=rank, world_size=world)
        ex = D.TorchExchange()
    else:
        ex = D.TcpExchange(rank, world, "127.0.0.1", port)

    class ExchangeEngine(OracleEngine):
        """slab rows travel through the host exchange instead of RCCL"""
        def comm_init(self, uid, r, n):
            if no_rccl:                         # what RCCL does for two ranks on one GPU
                from psa_amd import _hip
                raise _hip.PsaHipError("psa_comm_init failed (rc=-4): invalid usage")
            super().comm_init(uid, r, n)

        def gather(self, root, k_offsets, k_counts):
            assert not no_rccl, "RCCL gather used although the communicator could not be formed"
            lo, n = int(k_offsets[self.rank]), int(k_counts[self.rank])
            parts = ex.allgather(self._slab[lo:lo + n])
            if root < 0 or root == self.rank:
                for r, rows in enumerate(parts):
                    self._slab[int(k_offsets[r]):int(k_offsets[r] + k_counts[r])] = rows

        def fs_exchange(self, t_off, t_cnt, k_off, k_cnt):       # the "RCCL" all-to-all of the double
            assert not no_rccl
            parts = ex.allgather(self._fs["q"])
            lo, n = int(k_off[self.rank]), int(k_cnt[self.rank])
            for r, q in enumerate(parts):
                self.fs_write(int(t_off[r]), np.asarray(q)[lo:lo + n])

    with np.load(conftest.GOLDEN / "traj_a.npz") as z:
        d = {k: z[k] for k in z.files}
    d["dt_ps"], d["cells"] = float(d["dt_ps"]), tuple(int(v) for v in d["cells"])
    eng = ExchangeEngine(rank=rank)
    group = D.KShardGroup(eng, ex, gather=gather, root=0, balance=balance, mode=mode.split("+")[0])
    calc = conftest.make_calculator(d).attach(shard_group=group)
    mags, vecs = calc.get_k_path([1, 1, 0], 2.0, 7)               # 7 k-points over 2 ranks: 4 + 3
    if mode.endswith("+grid"):                                    # 4 x 6 grid symmetric about Gamma: 12 pairs
        mags, vecs, _ = calc.get_k_grid("xy", (-2.0, 2.0), (-1.0, 1.0), 4, 6, 0.0)
    if mode.endswith("+tiny"):                                    # 2 x 3 grid: 6 vectors, 3 distinct +-k
        mags, vecs, _ = calc.get_k_grid("xy", (-1.0, 1.0), (-0.5, 0.5), 2, 3, 0.0)
    out = {}
    for name, kw in (("coh", {}), ("inc", dict(basis_atom_types=[1, 2], summation_mode="incoherent"))):
        sed = calc.calculate(mags, vecs, **kw)
        out[name] = None if sed.sed is None else np.array(sed.sed)
        out[name + "_range"] = (eng.calls[-1]["k_offset"], eng.calls[-1]["K"], eng.calls[-1]["K_total"])
        out[name + "_frames"] = eng.slots[0].shape[0]
    out["transport"] = group.transport
    out["mode"] = group.last_mode
    ex.barrier()
    results[rank] = out
    group.close()
    if backend == "gloo":
        import torch.distributed as td
        td.destroy_process_group()


@pytest.mark.parametrize("backend, gather, no_rccl, balance, mode", [
    ("gloo", "all", False, None, "k"), ("gloo", "root", False, None, "k"), ("tcp", "all", False, None, "k"),
    ("gloo", "root", True, None, "k"), ("gloo", "root", False, MODEL, "k"), ("tcp", "all", False, MODEL, "k"),
    ("gloo", "all", False, None, "frames"), ("tcp", "root", False, None, "frames"),
    ("tcp", "all", True, None, "frames"), ("gloo", "root", False, MODEL, "auto")])
def test_two_rank_sharded_calculate_equals_unsharded(backend, gather, no_rccl, balance, mode):
    import conftest
    from oracle import psa_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        procs = [ctx.Process(target=_sharded_worker,
                             args=(r, world, port, backend, gather, results, no_rccl, balance, mode))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0, f"rank exited with {p.exitcode}"
        res = {r: dict(results[r]) for r in range(world)}

    with np.load(conftest.GOLDEN / "traj_a.npz") as z:
        d = {k: z[k] for k in z.files}
    calc = conftest.make_calculator(dict(d, dt_ps=float(d["dt_ps"]), cells=tuple(int(v) for v in d["cells"])))
    _, vecs = calc.get_k_path([1, 1, 0], 2.0, 7)
    ref_c, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], float(d["dt_ps"]), vecs)
    ref_i, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], float(d["dt_ps"]), vecs,
                              basis_atom_types=[1, 2], summation_mode="incoherent")
    frames = mode in ("frames", "auto")                   # 7 k-vectors over 2 ranks: "auto" picks frames
    assert res[0]["mode"] == res[1]["mode"] == ("frames" if frames else "k")
    if balance and gather == "root" and not frames:       # 7 rows, sending costs 4 per row: the root takes 6
        assert res[0]["coh_range"] == (0, 6, 7) and res[1]["coh_range"] == (6, 1, 7)
    else:                                                 # (gather="all" and frame sharding ignore the model)
        assert res[0]["coh_range"] == (0, 4, 7) and res[1]["coh_range"] == (4, 3, 7)
    # frame sharding keeps half of the 128 frames on each rank, k sharding all of them
    assert res[0]["coh_frames"] == res[1]["inc_frames"] == (64 if frames else 128)
    assert res[0]["transport"] == res[1]["transport"] == ("host" if no_rccl else "rccl")
    for rank in range(world):
        if gather == "root" and rank != 0:
            assert res[rank]["coh"] is None and res[rank]["inc"] is None
            continue
        assert conftest.rel_max(res[rank]["coh"], ref_c) <= 2e-6
        assert conftest.rel_max(res[rank]["inc"], ref_i) <= 2e-6


@pytest.mark.parametrize("mode, gather", [("k+grid", "all"), ("frames+grid", "root")])
def test_two_rank_sharded_grid_projects_each_pair_once(mode, gather):
    """A k-grid symmetric about Gamma over two ranks: the 12 unique k-vectors are sharded 6 + 6, the
    ranks that finalize install the k map, and the 24-point result equals the unsharded oracle."""
    import conftest
    from oracle import psa_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, "tcp", gather, results, False, None, mode))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0, f"rank exited with {p.exitcode}"
        res = {r: dict(results[r]) for r in range(world)}
    with np.load(conftest.GOLDEN / "traj_a.npz") as z:
        d = {k: z[k] for k in z.files}
    calc = conftest.make_calculator(dict(d, dt_ps=float(d["dt_ps"]), cells=tuple(int(v) for v in d["cells"])))
    _, vecs, _ = calc.get_k_grid("xy", (-2.0, 2.0), (-1.0, 1.0), 4, 6, 0.0)
    ref_c, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], float(d["dt_ps"]), vecs)
    ref_i, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], float(d["dt_ps"]), vecs,
                              basis_atom_types=[1, 2], summation_mode="incoherent")
    assert res[0]["coh_range"] == (0, 6, 12) and res[1]["inc_range"] == (6, 6, 12)      # 12 of 24 projected
    for rank in range(world):
        if gather == "root" and rank != 0:
            assert res[rank]["coh"] is None
            continue
        assert res[rank]["coh"].shape == ref_c.shape and res[rank]["inc"].shape == ref_i.shape
        assert conftest.rel_max(res[rank]["coh"], ref_c) <= 2e-6
        assert conftest.rel_max(res[rank]["inc"], ref_i) <= 2e-6


def test_five_ranks_share_a_folded_grid_with_fewer_vectors_than_ranks():
    """2 x 3 grid centred on Gamma: 6 k-vectors, 3 distinct +-k, over 5 ranks (two of them get nothing to
    project) -- k sharding, root gather; the root's result equals the unsharded oracle."""
    import conftest
    from oracle import psa_oracle as O
    world, port = 5, _free_port()
    ctx = mp.get_context("spawn")

    with ctx.Manager() as mgr:
        results = mgr.dict()
        procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, "tcp", "root", results, True, None, "k+tiny"))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0, f"rank exited with {p.exitcode}"
        res = {r: dict(results[r]) for r in range(world)}
    with np.load(conftest.GOLDEN / "traj_a.npz") as z:
        d = {k: z[k] for k in z.files}
    calc = conftest.make_calculator(dict(d, dt_ps=float(d["dt_ps"]), cells=tuple(int(v) for v in d["cells"])))
    _, vecs, _ = calc.get_k_grid("xy", (-1.0, 1.0), (-0.5, 0.5), 2, 3, 0.0)
    ref_c, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], float(d["dt_ps"]), vecs)
    ref_i, _, _ = O.calculate(d["positions"], d["velocities"], d["types"], float(d["dt_ps"]), vecs,
                              basis_atom_types=[1, 2], summation_mode="incoherent")
    assert [res[r]["coh_range"][1] for r in range(world)] == [1, 1, 1, 0, 0] and res[0]["coh_range"][2] == 3
    assert conftest.rel_max(res[0]["coh"], ref_c) <= 2e-6 and conftest.rel_max(res[0]["inc"], ref_i) <= 2e-6
    assert all(res[r]["coh"] is None for r in range(1, world))


def test_rendezvous_wire_format_round_trips_dicts():
    obj = {"a": 1.5, "b": [1, None, "x", {"c": np.arange(3, dtype=np.int32)}], "raw": b"\x00\x01"}
    back = dist._decode(dist._encode(obj))
    assert back["a"] == 1.5 and back["b"][:3] == [1, None, "x"] and back["raw"] == b"\x00\x01"
    assert np.array_equal(back["b"][3]["c"], np.arange(3))
    with pytest.raises(TypeError):
        dist._encode({1: 2})


def test_exchange_defaults_single_process():
    ex = dist.Exchange()
    assert ex.allgather(3) == [3] and ex.broadcast("x") == "x"
    ex.barrier()
    tcp = dist.TcpExchange(0, 1)
    assert tcp.allgather({"a": 1}) == [{"a": 1}]
    tcp.close()
