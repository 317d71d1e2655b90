"""Test double for `psa_amd._hip.Engine`, backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY: it lets the `-m "not gpu"` suite exercise the product's host
logic (group resolution, flags, k-sharding, result assembly) in a container without a GPU.
It is injected with `calc.attach(engine=OracleEngine())`; nothing under psa_amd/ knows it
exists and the product never falls back to it."""
import numpy as np

from oracle import psa_oracle as O
from psa_amd import _hip


class OracleEngine:
    def __init__(self, peers=None, rank=0):
        self.slots, self.calls = {}, []
        self.rank, self.nranks = rank, 1
        self._slab = self._meta = self._out = None
        self._peers = peers                      # shared dict rank -> engine (fake "RCCL")
        import threading
        self.lock = threading.RLock()

    # residency
    def ensure_resident(self, slot, array):
        self.slots[slot] = np.asarray(array, np.float32)
        self.uploads = getattr(self, "uploads", 0) + 1
        self._held = getattr(self, "_held", {})
        self._held[slot] = array

    def is_resident(self, slot, array):
        return getattr(self, "_held", {}).get(slot) is array

    def invalidate(self, slot=None):
        self._held = {}

    def project_upload(self, slot, array, mean_pos_all, k_vectors, groups=None, flags=0):
        self.ensure_resident(slot, array)
        self.project(slot, mean_pos_all, k_vectors, groups, flags)
        self.calls[-1]["streamed"] = True

    def single_bin(self, slot, mean_pos_all, k_vector, idx, i_w, flags=0):
        data = self.slots[slot]
        g = np.arange(data.shape[1]) if idx is None else np.asarray(idx)
        s = O.sed_for_group(data, data, np.asarray(k_vector, np.float32).reshape(1, 3), g,
                            np.asarray(mean_pos_all, np.float32),
                            use_displacements=bool(flags & _hip.F_DISPLACEMENTS))
        self.calls.append(dict(single_bin=True, i_w=i_w, n=len(g)))
        return s[i_w, 0, :]

    def shape(self, slot):
        return self.slots[slot].shape[:2]

    def mean_positions(self, slot):
        return O.mean_positions(self.slots[slot])

    # hot path
    def project(self, slot, mean_pos_all, k_vectors, groups=None, flags=0, K_total=None, k_offset=0):
        self.calls.append(dict(slot=slot, groups=groups, flags=flags, K=len(k_vectors),
                               K_total=K_total, k_offset=k_offset))
        data = self.slots[slot]
        T, N = data.shape[:2]
        K = len(k_vectors)
        K_total = K if K_total is None else K_total
        intensity = bool(flags & _hip.F_INTENSITY)
        disp = bool(flags & _hip.F_DISPLACEMENTS)
        if groups is None:
            groups = [np.arange(N)]
        if not intensity:
            assert len(groups) == 1
        if self._slab is None or self._meta != (T, K_total, intensity):
            self._slab = np.zeros((K_total, T) if intensity else (K_total, 3, T),
                                  np.float32 if intensity else np.complex64)
            self._meta = (T, K_total, intensity)
        self._kmap = None
        rows = slice(k_offset, k_offset + K)
        acc = np.zeros((T, K), np.float32)
        for g in groups:
            g = np.asarray(g)
            if g.size == 0:
                continue
            if np.any(g < 0) or np.any(g >= N):
                raise ValueError("Atom indices in basis out of bounds.")
            s = O.sed_for_group(data, data, np.asarray(k_vectors, np.float32), g,
                                np.asarray(mean_pos_all, np.float32), use_displacements=disp)
            if intensity:
                acc += np.sum(np.abs(s) ** 2, axis=-1)
            else:
                self._slab[rows] = s.transpose(1, 2, 0)
        if intensity:
            self._slab[rows] = acc.T

    def calculate(self, slot, mean_pos_all, k_vectors, groups=None, flags=0, with_intensity=False):
        self.project(slot, mean_pos_all, k_vectors, groups, flags)
        T = self.slots[slot].shape[0]
        return self.finalize(T, len(k_vectors), bool(flags & _hip.F_INTENSITY), with_intensity=with_intensity)

    def set_kmap(self, kmap):
        self._kmap = np.asarray(kmap, np.uint32)

    def gather(self, root, k_offsets, k_counts):
        for r, eng in self._peers.items():
            if r == self.rank or k_counts[r] == 0:
                continue
            if root < 0 or root == self.rank:
                rows = slice(int(k_offsets[r]), int(k_offsets[r] + k_counts[r]))
                self._slab[rows] = eng._slab[rows]

    def slab_read(self, row0, nrows, T, intensity):
        return self._slab[row0:row0 + nrows].copy()

    def slab_write(self, row0, rows):
        self._slab[row0:row0 + rows.shape[0]] = rows

    def finalize(self, T, K, intensity, fetch=True, with_intensity=False):
        slab, kmap = self._slab, getattr(self, "_kmap", None)
        if kmap is not None:                     # folded pairs: column k from row kmap[k], mirrored partners
            self._kmap = None                    # (the map belongs to the result that was just projected)
            back = (-np.arange(T)) % T           # S(-k)[w] = conj S(k)[(T-w) mod T]
            rows = slab[kmap & 0x7FFFFFFF]
            flip = (kmap >> 31).astype(bool)
            rows[flip] = rows[flip][..., back] if intensity else np.conj(rows[flip][..., back])
            slab = rows
        assert slab.shape[0] == K, (slab.shape, K)
        self._out = slab.T.copy() if intensity else slab.transpose(2, 0, 1).copy()
        if not with_intensity:
            return self._out if fetch else None
        inten = None if intensity else np.sum(np.abs(self._out) ** 2, axis=-1).astype(np.float32)
        return (self._out, inten) if fetch else (None, None)

    def result_chiral_phase(self, T, K, c1, c2):
        return O.chiral_phase(self._out[:, :, c1], self._out[:, :, c2], "C")

    def synchronize(self):
        pass

    def new_unique_id(self):
        return b"\0" * 128

    def comm_init(self, uid, rank, nranks):
        self.rank, self.nranks = rank, nranks

    def comm_destroy(self):
        self.nranks = 1

    def close(self):
        pass

    # frame sharding: the slot holds this rank's frames only
    def fs_project(self, slot, mean_pos_all, k_vectors, idx, flags, T_total, k_offset, k_count):
        data = self.slots[slot]
        T_local, N = data.shape[:2]
        g = np.arange(N) if idx is None else np.asarray(idx)
        mean = np.asarray(mean_pos_all, np.float32)
        intensity = bool(flags & _hip.F_INTENSITY)
        K = len(k_vectors)
        if self._slab is None or self._meta != (T_total, K, intensity):
            self._slab = np.zeros((K, T_total) if intensity else (K, 3, T_total), np.float32 if intensity else np.complex64)
            self._meta = (T_total, K, intensity)
        sel = data[:, g, :] - mean[g][None] if flags & _hip.F_DISPLACEMENTS else data[:, g, :]
        q = O.project_group(sel, O.phase_table(np.asarray(k_vectors, np.float32), mean[g]))    # (T_local, K, 3)
        self._fs = dict(q=np.ascontiguousarray(q.transpose(1, 2, 0)), T=T_total, k0=k_offset, nk=k_count,
                        intensity=intensity, rows=np.zeros((k_count, 3, T_total), np.complex64))
        self.calls.append(dict(fs=True, K_total=K, k_offset=k_offset, K=k_count, T_local=T_local))

    def fs_read(self, k0, nk, T_local):
        return self._fs["q"][k0:k0 + nk].copy()

    def fs_write(self, t0, block):
        self._fs["rows"][:, :, t0:t0 + block.shape[2]] = block

    def fs_exchange(self, t_off, t_cnt, k_off, k_cnt):
        raise AssertionError("the test double has no RCCL: the exchange must go through the host transport")

    def fs_finish(self, first_group):
        f = self._fs
        spec = (np.fft.fft(f["rows"], axis=2) / f["T"]).astype(np.complex64)       # sed_calculator.py:83-84
        rows = slice(f["k0"], f["k0"] + f["nk"])
        if f["intensity"]:
            inten = np.sum(np.abs(spec) ** 2, axis=1).astype(np.float32)
            self._slab[rows] = inten if first_group else self._slab[rows] + inten
        else:
            self._slab[rows] = spec
