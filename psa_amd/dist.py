"""
k-point sharding over the GPUs of one node: one process per GPU.

The k-points of a path or grid are independent through projection, FFT and |.|^2 (the
reference already loops over independent k-chunks, src/psa/core/sed_calculator.py:287-311),
so rank r computes the contiguous block of k-vectors `shard_ranges(K, nranks)[r]` against
its own resident copy of the trajectory and writes it into rows [offset, offset+count) of a
k-major slab.  The only exchange step is the final gather of those rows over RCCL/xGMI
(`psa_sed_gather`, grouped ncclSend/ncclRecv -- direct peer links, no ring), after which the
receiving rank(s) transpose to the reference's (T,K,3) / (T,K) layout.

Host-side rendezvous (shipping the 128-byte RCCL unique id, barriers, timing reductions)
goes through a small `Exchange` object.  Two are provided: `TorchExchange` rides an
already-initialised `torch.distributed` process group (gloo is enough -- no tensors touch it
on the data path), `TcpExchange` needs nothing but the MASTER_ADDR/MASTER_PORT the launcher
exports.
"""
from __future__ import annotations

import logging
import os
import pickle
import socket
import struct
import time
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import _hip

logger = logging.getLogger(__name__)


def shard_ranges(n_k: int, nranks: int, counts: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Contiguous split of n_k rows: (offsets, counts).  Default: balanced, the first
    n_k % nranks ranks take one extra row, ranks beyond n_k get empty ranges.  `counts` gives the
    rows per rank explicitly (see `root_heavy_counts`)."""
    if nranks < 1:
        raise ValueError("nranks must be >= 1")
    if counts is None:
        base, extra = divmod(int(n_k), nranks)
        counts = [base + (1 if r < extra else 0) for r in range(nranks)]
    counts = np.asarray(counts, dtype=np.int64)
    if counts.shape != (nranks,) or counts.min() < 0 or counts.sum() != n_k:
        raise ValueError("counts must be one non-negative row count per rank adding up to n_k")
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    return offsets, counts


def root_heavy_counts(n_k: int, nranks: int, root: int, per_k_s: float, base_s: float, floor_s: float,
                      per_k_bytes: float, link_bytes_per_s: float, block_k: int = 1) -> np.ndarray:
    """Rows per rank when only `root` receives the result: the root sends nothing, every other rank
    ships its rows over its own link after computing them, so the root takes more rows.  Model:
    a rank with n rows computes for max(floor_s, base_s + per_k_s * n') (floor_s = one pass over
    the trajectory at the HBM rate; n' = n rounded up to the projection kernel's block of block_k
    k-vectors, half a block for the short-list variant) and then sends for
    n * per_k_bytes / link_bytes_per_s; the
    latest finishing time is minimised over the rows given to each non-root rank (never more than
    the even share, which also wins ties)."""
    if nranks == 1:
        return np.array([n_k], dtype=np.int64)
    send = per_k_bytes / link_bytes_per_s

    def compute(n: int) -> float:
        if n <= 0:
            return 0.0
        padded = n if block_k <= 1 else (block_k // 2 if 2 * n <= block_k else -(-n // block_k) * block_k)
        return max(floor_s, base_s + per_k_s * padded)

    best, best_t = n_k // nranks, np.inf
    for other in range(n_k // nranks, (0 if n_k < nranks else 1) - 1, -1):     # even share first: it wins ties
        t = max(compute(n_k - other * (nranks - 1)), compute(other) + send * other)
        if t < best_t * (1 - 1e-9):
            best, best_t = other, t
    counts = np.full(nranks, best, dtype=np.int64)
    counts[root] = n_k - best * (nranks - 1)
    return counts


# --------------------------------------------------------------------------- exchanges
class Exchange:
    rank: int = 0
    nranks: int = 1

    def broadcast(self, obj: Any, root: int = 0) -> Any:
        return obj

    def allgather(self, obj: Any) -> List[Any]:
        return [obj]

    def barrier(self) -> None:
        self.allgather(None)

    def close(self) -> None:
        pass


class TorchExchange(Exchange):
    """Host-object collectives over an initialised torch.distributed group (gloo or nccl)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dist, self._group = dist, group
        self.rank, self.nranks = dist.get_rank(group), dist.get_world_size(group)

    def broadcast(self, obj, root=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=root, group=self._group)
        return box[0]

    def allgather(self, obj):
        out = [None] * self.nranks
        self._dist.all_gather_object(out, obj, group=self._group)
        return out


def _send_msg(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n: int) -> bytes:
    chunks = []
    while n:
        part = sock.recv(min(n, 1 << 20))
        if not part:
            raise ConnectionError("peer closed during rendezvous")
        chunks.append(part)
        n -= len(part)
    return b"".join(chunks)


def _recv_msg(sock) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class TcpExchange(Exchange):
    """Star rendezvous on MASTER_ADDR:port -- rank 0 listens, the others connect."""

    def __init__(self, rank: int, nranks: int, addr: str = "127.0.0.1", port: int = 29555,
                 timeout_s: float = 120.0):
        self.rank, self.nranks = rank, nranks
        self._peers: List[socket.socket] = []
        self._up: Optional[socket.socket] = None
        if nranks == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(nranks)
            srv.settimeout(timeout_s)
            by_rank = {}
            while len(by_rank) < nranks - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                by_rank[r] = conn
            srv.close()
            self._peers = [by_rank[r] for r in range(1, nranks)]
        else:
            deadline = time.time() + timeout_s
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<I", rank))
            self._up = s

    @classmethod
    def from_env(cls, port_offset: int = 17) -> "TcpExchange":
        rank = int(os.environ.get("RANK", "0"))
        n = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
        return cls(rank, n, addr, port)

    def allgather(self, obj):
        if self.nranks == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [pickle.loads(_recv_msg(p)) for p in self._peers]
            blob = pickle.dumps(items)
            for p in self._peers:
                _send_msg(p, blob)
            return items
        _send_msg(self._up, pickle.dumps(obj))
        return pickle.loads(_recv_msg(self._up))

    def broadcast(self, obj, root=0):
        return self.allgather(obj if self.rank == root else None)[root]

    def close(self):
        for s in self._peers + ([self._up] if self._up else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._up = [], None


# --------------------------------------------------------------------------- the group
class KShardGroup:
    """A rank's view of a k-sharded SED calculation.

    gather="all"  : every rank ends up with the full result (drop-in semantics: each
                    process's `SEDCalculator.calculate` returns what the reference returns);
    gather="root" : only rank `root` receives and returns the result, the others return None.
    """

    def __init__(self, engine: "_hip.Engine", exchange: Exchange, gather: str = "all", root: int = 0,
                 balance: Optional[dict] = None):
        """balance (gather="root" only): keyword arguments of `root_heavy_counts` other than n_k,
        nranks and root -- gives the root a larger block of k-vectors to offset the result rows
        the other ranks have to ship to it."""
        if gather not in ("all", "root"):
            raise ValueError("gather must be 'all' or 'root'")
        self.engine, self.exchange = engine, exchange
        self.rank, self.nranks = exchange.rank, exchange.nranks
        self.gather_mode, self.root = gather, root
        self.balance = balance if gather == "root" else None
        self.has_result = False
        self.transport = "rccl"
        if self.nranks > 1:
            uid = engine.new_unique_id() if self.rank == 0 else None
            uid = exchange.broadcast(uid, 0)
            err = None
            try:
                engine.comm_init(uid, self.rank, self.nranks)
            except _hip.PsaHipError as e:          # e.g. two ranks on one GPU
                err = str(e)
            # every rank must take the same path
            errors = [e for e in exchange.allgather(err) if e]
            if errors:
                self.transport = "host"
                logger.warning("RCCL communicator could not be formed (%s); slab rows will be exchanged "
                               "through the host rendezvous instead -- correct, but not the xGMI path",
                               errors[0].splitlines()[-1])

    def ranges(self, n_k: int) -> Tuple[np.ndarray, np.ndarray]:
        counts = None
        if self.balance and self.nranks > 1:
            counts = root_heavy_counts(n_k, self.nranks, self.root, **self.balance)
        return shard_ranges(n_k, self.nranks, counts)

    def my_range(self, n_k: int) -> Tuple[int, int]:
        off, cnt = self.ranges(n_k)
        return int(off[self.rank]), int(cnt[self.rank])

    def project(self, slot, mean_pos_all, k_vectors, groups, flags):
        """Project this rank's block of k-vectors and exchange slab rows.  Asynchronous on
        the engine's stream; no host sync."""
        n_k = len(k_vectors)
        off, cnt = self.ranges(n_k)
        lo, n = int(off[self.rank]), int(cnt[self.rank])
        self.engine.project(slot, mean_pos_all, np.asarray(k_vectors)[lo:lo + n], groups, flags,
                            K_total=n_k, k_offset=lo)
        if self.nranks > 1:
            root = -1 if self.gather_mode == "all" else self.root
            if self.transport == "rccl":
                self.engine.gather(root, off, cnt)
            else:
                self._host_gather(root, off, cnt, self.engine.shape(slot)[0], bool(flags & _hip.F_INTENSITY))
        self.has_result = self.gather_mode == "all" or self.rank == self.root

    def _host_gather(self, root, off, cnt, T, intensity):
        """Stand-in for psa_sed_gather when RCCL is unavailable: D2H of this rank's rows, exchange
        over the host rendezvous, H2D of the others' rows on the receiving rank(s)."""
        mine = self.engine.slab_read(int(off[self.rank]), int(cnt[self.rank]), T, intensity)
        parts = self.exchange.allgather(mine)
        if root < 0 or root == self.rank:
            for r, rows in enumerate(parts):
                if r != self.rank and cnt[r] > 0:
                    self.engine.slab_write(int(off[r]), rows)

    def run(self, slot, mean_pos_all, k_vectors, groups, flags, T: int, fetch: bool = True):
        self.project(slot, mean_pos_all, k_vectors, groups, flags)
        if not self.has_result:
            self.engine.synchronize()
            return None
        return self.engine.finalize(T, len(k_vectors), bool(flags & _hip.F_INTENSITY), fetch)

    def close(self):
        if self.nranks > 1:
            self.engine.comm_destroy()
        self.exchange.close()
