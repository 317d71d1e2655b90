"""
Sharding one SED calculation over the GPUs of one node: one process per GPU.

mode "k" -- the k-points of a path or grid are independent through projection, FFT and |.|^2
(the reference already loops over independent k-chunks, src/psa/core/sed_calculator.py:287-311),
so rank r computes the contiguous block of k-vectors `shard_ranges(K, nranks)[r]` against
its own resident copy of the WHOLE trajectory and writes it into rows [offset, offset+count) of a
k-major slab.  The only exchange step is the final gather of those rows over RCCL/xGMI
(`psa_sed_gather`, grouped ncclSend/ncclRecv -- direct peer links, no ring), after which the
receiving rank(s) transpose to the reference's (T,K,3) / (T,K) layout.  Every rank streams the
whole trajectory, so this stops scaling once a rank's k-block makes its projection HBM-bound
(about 32-64 k-vectors).

mode "frames" -- the projection is also independent per frame (:80-81; the FFT runs along frames,
:83).  Rank r keeps only frames `frame_ranges(T, nranks)[r]` in HBM (1/n of the array, 1/n of the
upload), projects ALL k-vectors on them, and an all-to-all over RCCL (`psa_sed_fs_exchange`: every
pair of ranks trades one block over its own link) hands each rank the missing frames of its own
block of k rows; FFT, epilogue and the final gather then proceed as in mode "k".  The projection
time falls as 1/n whatever K is.

mode "auto" -- "frames" when a rank's k-block would be 64 k-vectors or fewer, "k" otherwise.

Host-side rendezvous (shipping the 128-byte RCCL unique id, barriers, timing reductions)
goes through a small `Exchange` object.  Two are provided: `TorchExchange` rides an
already-initialised `torch.distributed` process group (gloo is enough -- no tensors touch it
on the data path), `TcpExchange` needs nothing but the MASTER_ADDR/MASTER_PORT the launcher
exports.
"""
from __future__ import annotations

import json
import logging
import os
import socket
import struct
import time
import weakref
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


def frame_ranges(n_frames: int, nranks: int, granule: int = 64) -> Tuple[np.ndarray, np.ndarray]:
    """Contiguous split of the frames: (offsets, counts), every block but the last a multiple of
    `granule` frames (the projection kernels' frame tile; the split planes are cut in groups of 16)."""
    if nranks < 1:
        raise ValueError("nranks must be >= 1")
    per = -(-int(n_frames) // nranks)
    per = -(-per // granule) * granule
    offsets = np.minimum(np.arange(nranks, dtype=np.int64) * per, n_frames)
    ends = np.minimum(offsets + per, n_frames)
    return offsets, (ends - offsets).astype(np.int64)


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


def _recv_msg(sock, limit: int = 1 << 36) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError(f"rendezvous message of {n} bytes refused")
    return _recv_exact(sock, n)


# Wire format of the host rendezvous: a JSON header describing a tree of None / bool / int / float /
# str / bytes / ndarray / list / dict with string keys, followed by the raw buffers of the bytes and ndarray leaves.  Nothing
# received from the network is ever executed or unpickled.
def _encode(obj) -> bytes:
    blobs: List[bytes] = []

    def walk(o):
        if o is None or isinstance(o, (bool, int, float, str)):
            return {"v": o}
        if isinstance(o, (np.integer, np.floating)):
            return {"v": o.item()}
        if isinstance(o, (bytes, bytearray)):
            blobs.append(bytes(o))
            return {"b": len(o)}
        if isinstance(o, np.ndarray):
            if o.dtype.hasobject:
                raise TypeError("object arrays cannot cross the rendezvous")
            a = np.ascontiguousarray(o)
            blobs.append(a.tobytes())
            return {"a": [a.dtype.str, list(a.shape)]}
        if isinstance(o, (list, tuple)):
            return {"l": [walk(x) for x in o]}
        if isinstance(o, dict) and all(isinstance(k, str) for k in o):
            return {"d": [[k, walk(v)] for k, v in o.items()]}
        raise TypeError(f"{type(o).__name__} cannot cross the rendezvous")

    head = json.dumps(walk(obj)).encode()
    return struct.pack("<I", len(head)) + head + b"".join(blobs)


def _decode(buf: bytes):
    (n,) = struct.unpack_from("<I", buf, 0)
    tree = json.loads(buf[4:4 + n].decode())
    pos = [4 + n]

    def take(k):
        if k < 0 or pos[0] + k > len(buf):
            raise ConnectionError("truncated rendezvous message")
        out = buf[pos[0]:pos[0] + k]
        pos[0] += k
        return out

    def walk(t):
        if "v" in t:
            return t["v"]
        if "b" in t:
            return take(int(t["b"]))
        if "a" in t:
            dt = np.dtype(t["a"][0])
            if dt.hasobject:
                raise ConnectionError("object dtype refused")
            shape = tuple(int(x) for x in t["a"][1])
            count = int(np.prod(shape, dtype=np.int64))
            return np.frombuffer(take(count * dt.itemsize), dtype=dt, count=count).reshape(shape).copy()
        if "d" in t:
            return {str(k): walk(v) for k, v in t["d"]}
        return [walk(x) for x in t["l"]]

    return walk(tree)


class TcpExchange(Exchange):
    """Star rendezvous on MASTER_ADDR:port -- rank 0 listens, the others connect.  Loopback unless
    an address is given; every socket operation times out; ranks are validated on connect."""

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
            try:
                while len(by_rank) < nranks - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout_s)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    try:
                        (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                    except (OSError, ConnectionError):
                        conn.close()
                        continue
                    if not 1 <= r < nranks or r in by_rank:      # not one of ours, or a duplicate
                        conn.close()
                        continue
                    by_rank[r] = conn
            finally:
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
            s.settimeout(timeout_s)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<I", rank))
            self._up = s

    @classmethod
    def from_env(cls, port_offset: int = 17, timeout_s: float = 120.0) -> "TcpExchange":
        rank = int(os.environ.get("RANK", "0"))
        n = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
        return cls(rank, n, addr, port, timeout_s)

    def allgather(self, obj):
        if self.nranks == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [_decode(_recv_msg(p)) for p in self._peers]
            blob = _encode(items)
            for p in self._peers:
                _send_msg(p, blob)
            return items
        _send_msg(self._up, _encode(obj))
        return _decode(_recv_msg(self._up))

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
    """A rank's view of a sharded SED calculation (modes: module docstring).

    gather="all"  : every rank ends up with the full result (drop-in semantics: each
                    process's `SEDCalculator.calculate` returns what the reference returns);
    gather="root" : only rank `root` receives and returns the result, the others return None.
    """

    def __init__(self, engine: "_hip.Engine", exchange: Exchange, gather: str = "all", root: int = 0,
                 balance: Optional[dict] = None, mode: str = "k"):
        """balance (gather="root", mode "k" only): keyword arguments of `root_heavy_counts` other than
        n_k, nranks and root -- gives the root a larger block of k-vectors to offset the result rows
        the other ranks have to ship to it."""
        if gather not in ("all", "root"):
            raise ValueError("gather must be 'all' or 'root'")
        if mode not in ("k", "frames", "auto"):
            raise ValueError("mode must be 'k', 'frames' or 'auto'")
        self.engine, self.exchange = engine, exchange
        self.rank, self.nranks = exchange.rank, exchange.nranks
        self.gather_mode, self.root = gather, root
        self.mode = mode
        self.balance = balance if gather == "root" else None
        self.has_result = False
        self.transport = "rccl"
        self.last_mode = "k"                 # what the last run actually did
        self.last_projected_k = 0            # k-vectors projected by the last run (after pair folding)
        self.fold_pairs = True               # PSA_OPT_FOLD_PAIRS across ranks
        self._slice = None                   # (weakref to the whole array, frame range, the slice view)
        self.transport_error = None          # why RCCL could not be used (transport == "host")
        if self.nranks > 1 and mode != "k" and hasattr(engine, "set_option"):
            # a frame-sharded slot holds T/n frames while the FFT runs over all T: the plan primed from
            # the slot's own length (PSA_OPT_FFT_PRIME) would be compiled for nothing
            engine.set_option(_hip.OPT_FFT_PRIME, 0)
        if self.nranks > 1:
            # the host driver of this pool supports dmabuf IPC only: without this RCCL's peer-memory
            # exchange fails with "hipIpcGetMemHandle: invalid argument".  `_hip.load_library` sets it
            # before HIP initialises (where the HSA runtime reads it); repeated here for engines that
            # bind the library some other way and for RCCL's own child helpers.  A launcher's setting wins.
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            uid = engine.new_unique_id() if self.rank == 0 else None
            uid = exchange.broadcast(uid, 0)
            err = None
            try:
                engine.comm_init(uid, self.rank, self.nranks)
            except _hip.PsaHipError as e:          # e.g. two ranks on one GPU
                err = str(e)
            # every rank must take the same path
            errors = [e for e in exchange.allgather(err) if e]
            if not errors and hasattr(engine, "comm_selftest"):
                # a communicator that formed must also move data: one small round in the data path's
                # own pattern (every pair of ranks trades a stamped block), before anything depends on it
                try:
                    engine.comm_selftest()
                except _hip.PsaHipError as e:
                    err = str(e)
                errors = [e for e in exchange.allgather(err) if e]
                if errors:
                    engine.comm_destroy()
            if errors:
                self.transport = "host"
                self.transport_error = errors[0]
                logger.warning("RCCL communicator could not be formed (%s); slab rows will be exchanged "
                               "through the host rendezvous instead -- correct, but not the xGMI path",
                               errors[0].splitlines()[-1])

    # -- geometry ----------------------------------------------------------------------
    def mode_for(self, n_k: int, n_frames: Optional[int] = None) -> str:
        """The stated rule: frame sharding when a rank's block would be <= 64 k-vectors (there the
        k-sharded projection is bound by every rank streaming the whole trajectory).  A trajectory too
        short to give every rank a frame tile is k-sharded whatever was asked."""
        if self.mode == "k" or (n_frames is not None and int(frame_ranges(n_frames, self.nranks)[1].min()) == 0):
            return "k"
        if self.mode == "frames":
            return "frames"
        return "frames" if -(-n_k // self.nranks) <= 64 else "k"

    def ranges(self, n_k: int, n_frames: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        counts = None
        if self.balance and self.nranks > 1 and self.mode_for(n_k, n_frames) == "k":
            counts = root_heavy_counts(n_k, self.nranks, self.root, **self.balance)
        return shard_ranges(n_k, self.nranks, counts)

    def my_range(self, n_k: int) -> Tuple[int, int]:
        off, cnt = self.ranges(n_k)
        return int(off[self.rank]), int(cnt[self.rank])

    def my_frames(self, n_frames: int) -> Tuple[int, int]:
        off, cnt = frame_ranges(n_frames, self.nranks)
        return int(off[self.rank]), int(cnt[self.rank])

    # -- residency -----------------------------------------------------------------------
    def ensure_resident(self, slot: int, data: np.ndarray, n_k: int):
        """The whole array (mode "k") or this rank's frames of it (mode "frames") into the slot."""
        if self.nranks == 1 or self.mode_for(n_k, data.shape[0]) == "k":
            self.engine.ensure_resident(slot, data)
            return
        t0, nt = self.my_frames(data.shape[0])
        held = self._slice
        if held is None or held[0]() is not data or held[1] != (slot, t0, nt):
            # one view object per (array, range): residency is keyed on object identity
            view = data[t0:t0 + nt]
            try:
                held = self._slice = (weakref.ref(data), (slot, t0, nt), view)
            except TypeError:
                held = (None, (slot, t0, nt), view)
        self.engine.ensure_resident(slot, held[2])

    # -- the calculation -------------------------------------------------------------------
    def project(self, slot, mean_pos_all, k_vectors, groups, flags, n_frames: Optional[int] = None):
        """Project this rank's share and exchange slab rows; afterwards `has_result` tells whether
        this rank holds the whole k-major slab.  Asynchronous on the engine's stream unless a host
        transport stands in for RCCL.  n_frames: frames of the whole trajectory (mode "frames";
        default: the slot's frames x nothing -- required there)."""
        k_vectors = np.asarray(k_vectors)
        kmap = None
        if self.nranks > 1 and self.fold_pairs and hasattr(self.engine, "set_kmap") and len(k_vectors) > 1:
            # k-vectors whose negation (or twin) is in the list are not projected by anyone: the unique
            # vectors are sharded, the rank(s) that finalize install the map (psa_sed_set_kmap)
            kmap, unique = _hip.k_pairs(k_vectors)
            if len(unique) < len(k_vectors):
                k_vectors = k_vectors[unique]
            else:
                kmap = None
        n_k = len(k_vectors)
        self.last_projected_k = n_k
        mode = self.mode_for(n_k, n_frames) if self.nranks > 1 else "k"
        self.last_mode = mode
        off, cnt = self.ranges(n_k, n_frames)
        lo, n = int(off[self.rank]), int(cnt[self.rank])
        intensity = bool(flags & _hip.F_INTENSITY)
        if mode == "k":
            self.engine.project(slot, mean_pos_all, np.asarray(k_vectors)[lo:lo + n], groups, flags,
                                K_total=n_k, k_offset=lo)
            T = self.engine.shape(slot)[0]
        else:
            if n_frames is None:
                raise ValueError("frame sharding needs the trajectory's total frame count")
            T = int(n_frames)
            t_off, t_cnt = frame_ranges(T, self.nranks)
            T_local = self.engine.shape(slot)[0]
            if T_local != int(t_cnt[self.rank]):
                raise RuntimeError(f"rank {self.rank} holds {T_local} frames, its share is {int(t_cnt[self.rank])}")
            todo = [None] if groups is None else [np.asarray(g) for g in groups if len(g)]
            if not intensity and len(todo) != 1:
                raise ValueError("complex output needs exactly one atom group")
            for gi, members in enumerate(todo):
                self.engine.fs_project(slot, mean_pos_all, k_vectors, members, flags, T, lo, n)
                if self.transport == "rccl":
                    self.engine.fs_exchange(t_off, t_cnt, off, cnt)
                else:
                    self._host_all_to_all(t_off, t_cnt, off, cnt, T_local)
                self.engine.fs_finish(gi == 0)
        if self.nranks > 1:
            root = -1 if self.gather_mode == "all" else self.root
            if self.transport == "rccl":
                self.engine.gather(root, off, cnt)
            else:
                self._host_gather(root, off, cnt, T, intensity)
        self.has_result = self.gather_mode == "all" or self.rank == self.root
        if kmap is not None and self.has_result:
            self.engine.set_kmap(kmap)

    def _host_gather(self, root, off, cnt, T, intensity):
        """Stand-in for psa_sed_gather when RCCL is unavailable: D2H of this rank's rows, exchange
        over the host rendezvous, H2D of the others' rows on the receiving rank(s)."""
        mine = self.engine.slab_read(int(off[self.rank]), int(cnt[self.rank]), T, intensity)
        parts = self.exchange.allgather(mine)
        if root < 0 or root == self.rank:
            for r, rows in enumerate(parts):
                if r != self.rank and cnt[r] > 0:
                    self.engine.slab_write(int(off[r]), rows)

    def _host_all_to_all(self, t_off, t_cnt, k_off, k_cnt, T_local):
        """Stand-in for psa_sed_fs_exchange: every rank publishes its q_local (all k rows, its own
        frames) over the host rendezvous and picks its block of rows out of everyone's."""
        parts = self.exchange.allgather(self.engine.fs_read(0, int(sum(k_cnt)), T_local))
        lo, n = int(k_off[self.rank]), int(k_cnt[self.rank])
        for r, q in enumerate(parts):
            if t_cnt[r] > 0 and n > 0:
                self.engine.fs_write(int(t_off[r]), np.asarray(q)[lo:lo + n])

    def run(self, slot, data, mean_pos_all, k_vectors, groups, flags, T: int, fetch: bool = True,
            with_intensity: bool = False):
        """Residency + projection + exchange + (on the ranks that receive it) the result; with_intensity:
        (result, sum_c |result|^2) for a complex result."""
        self.ensure_resident(slot, data, len(k_vectors))
        self.project(slot, mean_pos_all, k_vectors, groups, flags, n_frames=T)
        if not self.has_result:
            self.engine.synchronize()
            return (None, None) if with_intensity else None
        return self.engine.finalize(T, len(k_vectors), bool(flags & _hip.F_INTENSITY), fetch, with_intensity=with_intensity)

    def close(self):
        if self.nranks > 1:
            self.engine.comm_destroy()
        self.exchange.close()
