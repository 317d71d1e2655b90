"""
ctypes binding of libpsa_hip.so (C ABI: include/psa_hip.h).

This is the only door between the Python host code and the GPU.  There is no CPU
implementation behind it: if the library is not built, or no MI355X is visible, every
entry raises -- the caller is never silently routed somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libpsa_hip.so"

SLOT_VELOCITIES, SLOT_POSITIONS = 0, 1
F_DISPLACEMENTS, F_INTENSITY = 0x1, 0x2
K1_AUTO, K1_WAVE, K1_MFMA32, K1_SPLIT_BF16 = 0, 1, 2, 3
OPT_PLANES, OPT_PLANES_BUDGET, OPT_PLANES_EAGER, OPT_PLANES_MIN_K, OPT_FOLD_PAIRS, OPT_FFT_PRIME = 0, 1, 2, 3, 4, 5
OPT_K1_LOADER_WAVES = 6
OPT_K1_WIDE = 7
KMAP_MIRROR = 0x80000000
ABI_VERSION = 3
UNIQUE_ID_BYTES = 128
TIMING_NAMES = ("h2d", "phase", "project", "fft", "epilogue", "gather", "transpose", "d2h")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_ctx = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/psa_hip.h declares
SIGNATURES = {
    "psa_abi_version": (C.c_int, []),
    "psa_last_error": (C.c_char_p, []),
    "psa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "psa_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "psa_host_free": (C.c_int, [C.c_void_p]),
    "psa_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "psa_destroy": (C.c_int, [_ctx]),
    "psa_synchronize": (C.c_int, [_ctx]),
    "psa_set_k1": (C.c_int, [_ctx, C.c_int]),
    "psa_set_option": (C.c_int, [_ctx, C.c_int, C.c_int64]),
    "psa_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_int, C.POINTER(C.c_int), _i64p]),
    "psa_data_upload": (C.c_int, [_ctx, C.c_int, _f32p, C.c_int64, C.c_int64]),
    "psa_data_alloc": (C.c_int, [_ctx, C.c_int, C.c_int64, C.c_int64]),
    "psa_data_download": (C.c_int, [_ctx, C.c_int, _f32p, C.c_int64, C.c_int64]),
    "psa_data_release": (C.c_int, [_ctx, C.c_int]),
    "psa_data_shape": (C.c_int, [_ctx, C.c_int, _i64p, _i64p]),
    "psa_data_fill_synthetic": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_int64, C.c_int, _f32p, _i32p,
                                          _f32p, _f32p, _f32p, _f32p]),
    "psa_mean_positions": (C.c_int, [_ctx, C.c_int, _f32p]),
    "psa_host_mean_frames": (C.c_int, [_f32p, C.c_int64, C.c_int64, _f32p, C.c_int]),
    "psa_sed_project": (C.c_int, [_ctx, C.c_int, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64,
                                  _i32p, _i64p, C.c_int32, C.c_int32]),
    "psa_sed_project_upload": (C.c_int, [_ctx, C.c_int, _f32p, C.c_int64, C.c_int64, _f32p, _f32p, C.c_int64,
                                         _i32p, _i64p, C.c_int32, C.c_int32]),
    "psa_sed_finalize": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, _f32p, C.c_size_t]),
    "psa_sed_calculate": (C.c_int, [_ctx, C.c_int, _f32p, _f32p, C.c_int64, _i32p, _i64p,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, _f32p, C.c_size_t]),
    "psa_k_pairs": (C.c_int, [_f32p, C.c_int64, _i32p, _i32p, _i64p]),
    "psa_sed_set_kmap": (C.c_int, [_ctx, _i32p, C.c_int64]),
    "psa_sed_single_bin": (C.c_int, [_ctx, C.c_int, _f32p, _f32p, _i32p, C.c_int64, C.c_int32, C.c_int64, _f32p]),
    "psa_slab_read": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_void_p]),
    "psa_slab_write": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_void_p]),
    "psa_result_intensity": (C.c_int, [_ctx, _f32p, C.c_size_t]),
    "psa_result_chiral_phase": (C.c_int, [_ctx, C.c_int, C.c_int, _f32p, C.c_size_t]),
    "psa_last_timings": (C.c_int, [_ctx, C.POINTER(C.c_double)]),
    "psa_k1_stats": (C.c_int, [_ctx, _i64p, C.POINTER(C.c_double)]),
    "psa_oneoff_stats": (C.c_int, [_ctx, C.POINTER(C.c_double)]),
    "psa_debug_phase_table": (C.c_int, [_ctx, _f32p, _f32p, C.c_int64, _i32p, C.c_int64,
                                        C.c_int64, C.c_void_p]),
    "psa_debug_project_only": (C.c_int, [_ctx, C.c_int, _f32p, _f32p, C.c_int64, _i32p,
                                         C.c_int64, C.c_int32, C.c_void_p]),
    "psa_debug_project_frames": (C.c_int, [_ctx, C.c_int, _f32p, _f32p, C.c_int64, _i32p,
                                           C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "psa_debug_plane_cache": (C.c_int, [_ctx, _i64p, _i64p]),
    "psa_comm_unique_id": (C.c_int, [C.c_void_p]),
    "psa_comm_init": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "psa_comm_destroy": (C.c_int, [_ctx]),
    "psa_comm_selftest": (C.c_int, [_ctx]),
    "psa_sed_gather": (C.c_int, [_ctx, C.c_int, _i64p, _i64p]),
    "psa_comm_barrier": (C.c_int, [_ctx]),
    "psa_sed_fs_project": (C.c_int, [_ctx, C.c_int, _f32p, _f32p, C.c_int64, _i32p, C.c_int64, C.c_int32,
                                     C.c_int64, C.c_int64, C.c_int64]),
    "psa_sed_fs_exchange": (C.c_int, [_ctx, _i64p, _i64p, _i64p, _i64p]),
    "psa_sed_fs_read": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_void_p]),
    "psa_sed_fs_write": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_void_p]),
    "psa_sed_fs_finish": (C.c_int, [_ctx, C.c_int32]),
}


class PsaHipError(RuntimeError):
    """A libpsa_hip entry point returned a non-zero code."""


_lib = None
_lib_lock = threading.Lock()


def load_library() -> C.CDLL:
    """dlopen libpsa_hip.so and bind every declared symbol.  Raises if it is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = Path(os.environ.get("PSA_HIP_LIBRARY", LIB_PATH))
        if not path.exists():
            raise PsaHipError(
                f"{path} not found: build it with `make -C psa_amd/csrc` (needs hipcc, gfx950). "
                "psa_amd has no CPU fallback.")
        # multi-process runs (psa_amd/dist.py): the host driver of this pool supports dmabuf IPC only, and the
        # HSA runtime reads this when the first HIP call initialises it -- so it is set before the library
        # (and with it HIP) is loaded; a launcher's own setting wins
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().psa_last_error().decode("utf-8", "replace")
        if rc == -1 and "out of bounds" in msg:      # same exception type as the reference
            raise ValueError(msg)
        raise PsaHipError(f"{what} failed (rc={rc}): {msg}")


def _f32(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _as_f32(a, shape_tail=None) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if shape_tail is not None and arr.shape[1:] != shape_tail:
        raise ValueError(f"expected (*,{shape_tail}) array, got {arr.shape}")
    return arr


def device_count() -> int:
    n = C.c_int(0)
    _check(load_library().psa_device_count(C.byref(n)), "psa_device_count")
    return n.value


def k_pairs(k_vectors):
    """(kmap uint32 (K,), unique_idx int32 (U,)): pairs (k, -k) and repeated vectors of a k-list
    (psa_k_pairs; bit 31 of kmap = the vector is the exact negation of row kmap & 0x7fffffff)."""
    kv = _as_f32(k_vectors, (3,)) if len(k_vectors) else np.zeros((0, 3), np.float32)
    K = kv.shape[0]
    kmap, uidx, n = np.zeros(K, np.int32), np.zeros(K, np.int32), C.c_int64(0)
    _check(load_library().psa_k_pairs(_f32(kv), K, kmap.ctypes.data_as(_i32p), uidx.ctypes.data_as(_i32p), C.byref(n)),
           "psa_k_pairs")
    return kmap.view(np.uint32), uidx[:n.value].copy()


def host_mean_frames(x: np.ndarray, threads: int = 0) -> np.ndarray:
    """np.mean(x, axis=0, dtype=np.float32) of a C-contiguous float32 (T, ...) array, bit for bit, on
    several host threads (psa_host_mean_frames)."""
    if x.dtype != np.float32 or not x.flags.c_contiguous or x.ndim < 2 or x.shape[0] < 1 or x.size == 0:
        raise ValueError("host_mean_frames needs a non-empty C-contiguous float32 array")
    out = np.empty(x.shape[1:], np.float32)
    cols = int(np.prod(x.shape[1:], dtype=np.int64))
    _check(load_library().psa_host_mean_frames(_f32(x), x.shape[0], cols, _f32(out), int(threads)), "psa_host_mean_frames")
    return out


def pack_groups(groups: Optional[Sequence[np.ndarray]]):
    """list of index arrays -> (idx int32, off int64, G) for the ABI; None -> all atoms."""
    if groups is None:
        return None, None, 1
    off = np.zeros(len(groups) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(g) for g in groups])
    idx = (np.concatenate([np.asarray(g).ravel() for g in groups]) if off[-1] > 0
           else np.zeros(0, dtype=np.int64))
    if idx.size and (idx.min() < 0 or idx.max() >= 2 ** 31):
        raise ValueError("Atom indices in basis out of bounds.")
    return np.ascontiguousarray(idx, dtype=np.int32), off, len(groups)


class _PinnedPool:
    """Page-locked host buffers for result arrays, recycled by size.  A buffer goes back to the pool
    when the last NumPy view of it is garbage-collected; at most `keep_bytes` idle bytes are kept."""

    granule = 2 << 20
    min_bytes = 1 << 20                     # smaller results: ordinary memory
    keep_bytes = 4 << 30                    # idle buffers kept for reuse
    max_live_bytes = 16 << 30               # page-locked bytes handed out and not yet collected: beyond
                                            # this, results go to ordinary memory

    def __init__(self):
        self._idle = {}                     # size -> [address, ...]
        self._idle_bytes = 0
        self._live_bytes = 0
        self._lock = threading.Lock()
        self.stats = {"recycled": 0, "allocated": 0, "pageable": 0}    # results of >= min_bytes, by how they were served

    def _give_back(self, address: int, size: int):
        with self._lock:
            self._live_bytes -= size
            if self._idle_bytes + size <= self.keep_bytes:
                self._idle.setdefault(size, []).append(address)
                self._idle_bytes += size
                return
        load_library().psa_host_free(C.c_void_p(address))

    def empty(self, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.min_bytes:
            return np.empty(shape, dtype)
        size = -(-nbytes // self.granule) * self.granule
        with self._lock:
            if self._live_bytes + size > self.max_live_bytes:
                self.stats["pageable"] += 1
                return np.empty(shape, dtype)
            self._live_bytes += size
            stack = self._idle.get(size)
            address = stack.pop() if stack else None
            if address is not None:
                self._idle_bytes -= size
        if address is None:
            p = C.c_void_p()
            try:
                _check(load_library().psa_host_alloc(size, C.byref(p)), "psa_host_alloc")
            except PsaHipError:             # e.g. the locked-memory limit: an ordinary array will do
                with self._lock:
                    self._live_bytes -= size
                    self.stats["pageable"] += 1
                return np.empty(shape, dtype)
            address = p.value
            self.stats["allocated"] += 1
        else:
            self.stats["recycled"] += 1
        block = (C.c_char * size).from_address(address)
        weakref.finalize(block, self._give_back, address, size)     # block dies with its last view
        return np.frombuffer(block, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)

    def drain(self):
        with self._lock:
            idle, self._idle, self._idle_bytes = self._idle, {}, 0
        for stack in idle.values():
            for address in stack:
                load_library().psa_host_free(C.c_void_p(address))


_pinned_pool = _PinnedPool()


def pinned_empty(shape, dtype) -> np.ndarray:
    """np.empty in page-locked memory (results of at least 1 MiB); an ordinary writable ndarray."""
    return _pinned_pool.empty(shape, dtype)


_SAMPLE_CACHE = {}


def _sample_positions(shape):
    """(flat positions, the same as per-axis index arrays) of the ~2k elements `_fingerprint` reads of an
    array of this shape; built once per shape."""
    hit = _SAMPLE_CACHE.get(shape)
    if hit is None:
        size = int(np.prod(shape, dtype=np.int64))
        lin = np.arange(0, size, max(1, size // 1021), dtype=np.int64)[:1021]
        if size > 2048:
            lin = np.concatenate([lin, np.sort(np.random.default_rng(size).integers(0, size, 1021)), [size - 1]])
        if len(_SAMPLE_CACHE) > 64:
            _SAMPLE_CACHE.clear()
        hit = _SAMPLE_CACHE[shape] = (lin, np.unravel_index(lin, shape))
    return hit


class Engine:
    """One HIP context (one GPU, one stream) with trajectory arrays resident in HBM."""

    def __init__(self, device: Optional[int] = None):
        self._lib = load_library()
        if device is None:
            device = int(os.environ.get("PSA_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = device_count()
            if n == 0:
                raise PsaHipError("no HIP device visible; psa_amd has no CPU fallback")
            device %= n
        h = _ctx()
        _check(self._lib.psa_create(int(device), C.byref(h)), "psa_create")
        self._h = h
        self.device = int(device)
        self._resident = {}          # slot -> (weakref to the array, data pointer, shape, fingerprint)
        self._converted = {}         # slot -> (weakref to a non-float32 source, fingerprint, float32 copy)
        # One calculation = upload + project (+ gather) + finalize on ONE context; callers that may
        # race (the reference GUI computes on worker threads) hold this around the sequence.
        self.lock = threading.RLock()
        self.rank, self.nranks = 0, 1
        self.result_serial = 0       # bumped by every call that replaces the result resident on the device

    # -- lifecycle -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.psa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._lib.psa_synchronize(self._h), "psa_synchronize")

    def set_k1(self, selector: int):
        _check(self._lib.psa_set_k1(self._h, selector), "psa_set_k1")

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(0), C.c_int64(0)
        _check(self._lib.psa_device_info(self._h, name, 256, C.byref(cu), C.byref(mem)),
               "psa_device_info")
        return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value}

    # -- trajectory residency --------------------------------------------------------
    @staticmethod
    def _fingerprint(a: np.ndarray) -> int:
        """Hash of ~2k elements spread over the array: catches in-place edits of a resident
        trajectory (scaling, overwriting, loading new frames into the same buffer) without
        reading it.  Not a proof of equality -- `invalidate()` is the explicit way."""
        if a.size == 0:
            return 0
        # a memory-mapped file (the reference's .npy cache, io/loader.py:48-79) is not sampled --
        # scattered reads of a file that may not sit in the page cache cost a disk seek each --
        # but identified by the file's name, size and modification time
        base = a
        while base is not None and not isinstance(base, np.memmap):
            base = getattr(base, "base", None)
        if base is not None and getattr(base, "filename", None):
            try:
                st = os.stat(base.filename)
                return hash((str(base.filename), st.st_size, st.st_mtime_ns, a.shape, a.strides))
            except OSError:
                pass
        # an even sweep plus scattered positions (an even stride alone can sit on one column of a
        # (T, N, 3) or (T, K, 3) array for ever); positions depend on the shape only and are kept
        lin, idx = _sample_positions(a.shape)
        if a.flags.c_contiguous:
            return hash(a.reshape(-1)[lin].tobytes())                # a gather: never copies the array
        return hash(a[idx].tobytes())

    def _as_device_layout(self, slot: int, array: np.ndarray) -> np.ndarray:
        """The array as C-contiguous float32.  A converted copy is kept (per slot) while the
        source object lives, so that float64 or strided trajectories are not re-converted and
        re-uploaded on every call."""
        if array.ndim != 3 or array.shape[2] != 3:
            raise ValueError("trajectory array must be (frames, atoms, 3)")
        if array.dtype == np.float32 and array.flags.c_contiguous:
            return array
        held = self._converted.get(slot)
        fp = self._fingerprint(array)
        if held is not None and held[0]() is array and held[1] == fp:
            return held[2]
        conv = np.ascontiguousarray(array, dtype=np.float32)
        try:
            self._converted[slot] = (weakref.ref(array), fp, conv)
        except TypeError:
            self._converted.pop(slot, None)
        return conv

    def is_resident(self, slot: int, array: np.ndarray) -> bool:
        """True if this very array (same object, same buffer, same sampled contents) is what the
        slot holds."""
        held = self._resident.get(slot)
        if held is None or held[0] is None:
            return False
        a = array if held[0]() is array else self._as_device_layout(slot, array)
        return (held[0]() is a and held[1:3] == (a.ctypes.data, a.shape) and held[3] == self._fingerprint(a))

    def _note_resident(self, slot: int, a: np.ndarray):
        # a weak reference, not id(): a freed array's id and buffer can be reused
        try:
            ref = weakref.ref(a)
        except TypeError:
            ref = None
        self._resident[slot] = (ref, a.ctypes.data, a.shape, self._fingerprint(a))

    def ensure_resident(self, slot: int, array: np.ndarray):
        """Upload a (T,N,3) array unless this very array is already in the slot."""
        if self.is_resident(slot, array):
            return
        a = self._as_device_layout(slot, array)
        T, N = a.shape[0], a.shape[1]
        _check(self._lib.psa_data_upload(self._h, slot, _f32(a), T, N), "psa_data_upload")
        self._note_resident(slot, a)

    def invalidate(self, slot: Optional[int] = None):
        """Forget what is resident (call after modifying a trajectory array in place)."""
        if slot is None:
            self._resident.clear()
            self._converted.clear()
        else:
            self._resident.pop(slot, None)
            self._converted.pop(slot, None)

    def alloc(self, slot: int, T: int, N: int):
        _check(self._lib.psa_data_alloc(self._h, slot, T, N), "psa_data_alloc")
        self._resident[slot] = (None, None, (T, N, 3), None)

    def fill_synthetic(self, slot: int, seed: int, amp, comp, ct, st, ca, sa, t_offset: int = 0):
        """ct / st hold the rows of the slot's own frames [t_offset, t_offset + T)."""
        amp = np.ascontiguousarray(amp, np.float32)
        comp = np.ascontiguousarray(comp, np.int32)
        tabs = [np.ascontiguousarray(x, np.float32) for x in (ct, st, ca, sa)]
        _check(self._lib.psa_data_fill_synthetic(
            self._h, slot, C.c_uint64(seed), int(t_offset), len(amp), _f32(amp), comp.ctypes.data_as(_i32p),
            *[_f32(t) for t in tabs]), "psa_data_fill_synthetic")

    def set_option(self, option: int, value: int):
        _check(self._lib.psa_set_option(self._h, option, int(value)), "psa_set_option")

    def plane_cache(self):
        """(number of cached split-plane sets, their bytes)"""
        n, b = C.c_int64(0), C.c_int64(0)
        _check(self._lib.psa_debug_plane_cache(self._h, C.byref(n), C.byref(b)), "psa_debug_plane_cache")
        return n.value, b.value

    def download(self, slot: int, t0: int, nt: int) -> np.ndarray:
        T, N = self.shape(slot)
        out = np.empty((nt, N, 3), np.float32)
        _check(self._lib.psa_data_download(self._h, slot, _f32(out), t0, nt), "psa_data_download")
        return out

    def release(self, slot: int):
        _check(self._lib.psa_data_release(self._h, slot), "psa_data_release")
        self._resident.pop(slot, None)

    def shape(self, slot: int):
        T, N = C.c_int64(0), C.c_int64(0)
        _check(self._lib.psa_data_shape(self._h, slot, C.byref(T), C.byref(N)), "psa_data_shape")
        return T.value, N.value

    def mean_positions(self, slot: int) -> np.ndarray:
        _, N = self.shape(slot)
        out = np.empty((N, 3), np.float32)
        _check(self._lib.psa_mean_positions(self._h, slot, _f32(out)), "psa_mean_positions")
        return out

    # -- the hot path ----------------------------------------------------------------
    def project(self, slot, mean_pos_all, k_vectors, groups=None, flags=0,
                K_total=None, k_offset=0):
        mean = _as_f32(mean_pos_all, (3,))
        kv = _as_f32(k_vectors, (3,)) if len(k_vectors) else np.zeros((0, 3), np.float32)
        idx, off, G = pack_groups(groups)
        K_local = kv.shape[0]
        self.result_serial += 1
        _check(self._lib.psa_sed_project(
            self._h, slot, _f32(mean), _f32(kv), K_local,
            K_local if K_total is None else K_total, k_offset,
            idx.ctypes.data_as(_i32p) if idx is not None else None,
            off.ctypes.data_as(_i64p) if off is not None else None, G, flags),
            "psa_sed_project")

    def project_upload(self, slot, array, mean_pos_all, k_vectors, groups=None, flags=0):
        """`ensure_resident` + `project` for an array that is not in HBM yet, overlapped: frames are
        projected as their chunk lands (psa_sed_project_upload)."""
        a = self._as_device_layout(slot, array)
        mean = _as_f32(mean_pos_all, (3,))
        kv = _as_f32(k_vectors, (3,))
        idx, off, G = pack_groups(groups)
        self._resident.pop(slot, None)
        self.result_serial += 1
        _check(self._lib.psa_sed_project_upload(
            self._h, slot, _f32(a), a.shape[0], a.shape[1], _f32(mean), _f32(kv), kv.shape[0],
            idx.ctypes.data_as(_i32p) if idx is not None else None,
            off.ctypes.data_as(_i64p) if off is not None else None, G, flags), "psa_sed_project_upload")
        self._note_resident(slot, a)

    def finalize(self, T: int, K: int, intensity: bool, fetch: bool = True, with_intensity: bool = False):
        """(T,K) / (T,K,3) is what the caller expects: the library refuses (PSA_EINVAL) if the
        result resident on the device has another size.  with_intensity (complex results): returns
        (sed, sum_c |sed|^2) -- the (T,K) float32 intensity comes out of the same pass on the device."""
        if not fetch:
            _check(self._lib.psa_sed_finalize(self._h, None, 0, None, 0), "psa_sed_finalize")
            return (None, None) if with_intensity else None
        out = pinned_empty((T, K), np.float32) if intensity else pinned_empty((T, K, 3), np.complex64)
        inten = pinned_empty((T, K), np.float32) if with_intensity and not intensity else None
        _check(self._lib.psa_sed_finalize(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes,
                                          _f32(inten) if inten is not None else None,
                                          inten.nbytes if inten is not None else 0), "psa_sed_finalize")
        return (out, inten) if with_intensity else out

    def single_bin(self, slot, mean_pos_all, k_vector, idx, i_w: int, flags=0) -> np.ndarray:
        """S[i_w, k, :] of one k-vector and one atom group as (3,) complex64."""
        mean = _as_f32(mean_pos_all, (3,))
        kv = np.ascontiguousarray(k_vector, np.float32).reshape(3)
        ii = None if idx is None else np.ascontiguousarray(idx, np.int32)
        out = np.empty(3, np.complex64)
        _check(self._lib.psa_sed_single_bin(
            self._h, slot, _f32(mean), _f32(kv), ii.ctypes.data_as(_i32p) if ii is not None else None,
            0 if ii is None else len(ii), flags, int(i_w), out.ctypes.data_as(_f32p)), "psa_sed_single_bin")
        return out

    def calculate(self, slot, mean_pos_all, k_vectors, groups=None, flags=0, with_intensity: bool = False):
        """project + finalize in one library call (psa_sed_calculate): a complex result of a long
        k-list is produced block by block, each block's D2H copy overlapping the next projection.
        with_intensity (complex results): returns (sed, sum_c |sed|^2), the second array computed on the
        device in the pass that writes the first."""
        T, _ = self.shape(slot)
        mean = _as_f32(mean_pos_all, (3,))
        kv = _as_f32(k_vectors, (3,))
        idx, off, G = pack_groups(groups)
        K = kv.shape[0]
        intensity = bool(flags & F_INTENSITY)
        out = pinned_empty((T, K), np.float32) if intensity else pinned_empty((T, K, 3), np.complex64)
        inten = pinned_empty((T, K), np.float32) if with_intensity and not intensity else None
        self.result_serial += 1
        _check(self._lib.psa_sed_calculate(
            self._h, slot, _f32(mean), _f32(kv), K,
            idx.ctypes.data_as(_i32p) if idx is not None else None,
            off.ctypes.data_as(_i64p) if off is not None else None, G, flags,
            out.ctypes.data_as(C.c_void_p), out.nbytes,
            _f32(inten) if inten is not None else None, inten.nbytes if inten is not None else 0), "psa_sed_calculate")
        return (out, inten) if with_intensity else out

    def set_kmap(self, kmap: np.ndarray):
        """Install the k map of a result whose slab rows were projected from a folded list (`k_pairs`)."""
        m = np.ascontiguousarray(kmap, np.uint32).view(np.int32)
        _check(self._lib.psa_sed_set_kmap(self._h, m.ctypes.data_as(_i32p), len(m)), "psa_sed_set_kmap")

    def slab_read(self, row0: int, nrows: int, T: int, intensity: bool) -> np.ndarray:
        out = np.empty((nrows, T), np.float32) if intensity else np.empty((nrows, 3, T), np.complex64)
        _check(self._lib.psa_slab_read(self._h, row0, nrows, out.ctypes.data_as(C.c_void_p)), "psa_slab_read")
        return out

    def slab_write(self, row0: int, rows: np.ndarray):
        rows = np.ascontiguousarray(rows)
        self.result_serial += 1
        _check(self._lib.psa_slab_write(self._h, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p)),
               "psa_slab_write")

    def intensity_source(self, array: np.ndarray):
        """A callable for `SED._device_intensity`: given the array now in `SED.sed`, the intensity of
        the complex result resident on the device -- or None if that is no longer this array's
        result (another calculation came in between, the array was replaced or edited)."""
        serial, ref, stamp = self.result_serial, weakref.ref(array), self._fingerprint(array)
        T, K = array.shape[0], array.shape[1]

        def source(current):
            if current is not ref() or self._h is None:
                return None
            with self.lock:
                if self.result_serial != serial or self._fingerprint(current) != stamp:
                    return None
                try:
                    return self.result_intensity(T, K)
                except PsaHipError:
                    return None
        return source

    def result_intensity(self, T: int, K: int) -> np.ndarray:
        out = pinned_empty((T, K), np.float32)
        _check(self._lib.psa_result_intensity(self._h, _f32(out), out.nbytes), "psa_result_intensity")
        return out

    def result_chiral_phase(self, T: int, K: int, c1: int, c2: int) -> np.ndarray:
        out = pinned_empty((T, K), np.float32)
        _check(self._lib.psa_result_chiral_phase(self._h, c1, c2, _f32(out), out.nbytes),
               "psa_result_chiral_phase")
        return out

    def timings(self) -> dict:
        ms = (C.c_double * 8)()
        _check(self._lib.psa_last_timings(self._h, ms), "psa_last_timings")
        return dict(zip(TIMING_NAMES, list(ms)))

    def oneoff_stats(self) -> dict:
        """Host wall clock (ms) of once-per-array / once-per-shape work since the last call."""
        ms = (C.c_double * 4)()
        _check(self._lib.psa_oneoff_stats(self._h, ms), "psa_oneoff_stats")
        return dict(zip(("rocfft_plan", "absmax", "split_planes", "upload"), list(ms)))

    def adopt(self, slot: int, array: np.ndarray):
        """Declare that the slot's device-generated contents (alloc + fill_synthetic) ARE `array` as
        far as residency goes: `calculate` on a Trajectory holding `array` then finds it resident.
        For benchmarks whose trajectory exists only in HBM (`array` can be a zero-stride stand-in)."""
        T, N = self.shape(slot)
        if tuple(array.shape) != (T, N, 3) or array.dtype != np.float32 or not array.flags.c_contiguous and array.strides != (0, 0, 0):
            raise ValueError("stand-in must be a float32 (T, N, 3) array of the slot's shape")
        self._note_resident(slot, array)

    def k1_stats(self):
        n, ms = C.c_int64(0), C.c_double(0.0)
        _check(self._lib.psa_k1_stats(self._h, C.byref(n), C.byref(ms)), "psa_k1_stats")
        return n.value, ms.value

    # -- diagnostics -----------------------------------------------------------------
    def debug_phase_table(self, mean_pos_all, k_vectors, idx=None) -> np.ndarray:
        mean = _as_f32(mean_pos_all, (3,))
        kv = _as_f32(k_vectors, (3,))
        N = mean.shape[0]
        ii = None if idx is None else np.ascontiguousarray(idx, np.int32)
        n_g = N if ii is None else len(ii)
        out = np.empty((kv.shape[0], n_g), np.complex64)
        _check(self._lib.psa_debug_phase_table(
            self._h, _f32(mean), _f32(kv), kv.shape[0],
            ii.ctypes.data_as(_i32p) if ii is not None else None, n_g, N,
            out.ctypes.data_as(C.c_void_p)), "psa_debug_phase_table")
        return out

    def debug_project_only(self, slot, mean_pos_all, k_vectors, idx=None, flags=0, frames=None) -> np.ndarray:
        """q before the FFT, (K,3,T); `frames=(t_begin, t_count)` projects only those frames (the
        other columns are zero)."""
        T, N = self.shape(slot)
        mean = _as_f32(mean_pos_all, (3,))
        kv = _as_f32(k_vectors, (3,))
        ii = None if idx is None else np.ascontiguousarray(idx, np.int32)
        n_g = N if ii is None else len(ii)
        out = np.empty((kv.shape[0], 3, T), np.complex64)
        ip = ii.ctypes.data_as(_i32p) if ii is not None else None
        if frames is None:
            _check(self._lib.psa_debug_project_only(
                self._h, slot, _f32(mean), _f32(kv), kv.shape[0], ip, n_g, flags,
                out.ctypes.data_as(C.c_void_p)), "psa_debug_project_only")
        else:
            _check(self._lib.psa_debug_project_frames(
                self._h, slot, _f32(mean), _f32(kv), kv.shape[0], ip, n_g, flags, int(frames[0]), int(frames[1]),
                out.ctypes.data_as(C.c_void_p)), "psa_debug_project_frames")
        return out

    # -- k-point sharding ------------------------------------------------------------
    @staticmethod
    def new_unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        _check(load_library().psa_comm_unique_id(buf), "psa_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        _check(self._lib.psa_comm_init(self._h, buf, rank, nranks), "psa_comm_init")
        self.rank, self.nranks = rank, nranks

    def comm_selftest(self):
        _check(self._lib.psa_comm_selftest(self._h), "psa_comm_selftest")

    def comm_destroy(self):
        _check(self._lib.psa_comm_destroy(self._h), "psa_comm_destroy")
        self.rank, self.nranks = 0, 1

    def gather(self, root: int, k_offsets, k_counts):
        o = np.ascontiguousarray(k_offsets, np.int64)
        n = np.ascontiguousarray(k_counts, np.int64)
        _check(self._lib.psa_sed_gather(self._h, root, o.ctypes.data_as(_i64p),
                                        n.ctypes.data_as(_i64p)), "psa_sed_gather")

    def barrier(self):
        _check(self._lib.psa_comm_barrier(self._h), "psa_comm_barrier")

    # -- frame sharding --------------------------------------------------------------
    def fs_project(self, slot, mean_pos_all, k_vectors, idx, flags, T_total: int, k_offset: int, k_count: int):
        """All K k-vectors on the slot's own frames for one atom group -> q_local on the device."""
        mean = _as_f32(mean_pos_all, (3,))
        kv = _as_f32(k_vectors, (3,))
        ii = None if idx is None else np.ascontiguousarray(idx, np.int32)
        self.result_serial += 1
        _check(self._lib.psa_sed_fs_project(
            self._h, slot, _f32(mean), _f32(kv), kv.shape[0], ii.ctypes.data_as(_i32p) if ii is not None else None,
            0 if ii is None else len(ii), flags, int(T_total), int(k_offset), int(k_count)), "psa_sed_fs_project")

    def fs_exchange(self, t_offsets, t_counts, k_offsets, k_counts):
        arrs = [np.ascontiguousarray(x, np.int64) for x in (t_offsets, t_counts, k_offsets, k_counts)]
        _check(self._lib.psa_sed_fs_exchange(self._h, *[a.ctypes.data_as(_i64p) for a in arrs]), "psa_sed_fs_exchange")

    def fs_read(self, k0: int, nk: int, T_local: int) -> np.ndarray:
        out = np.empty((nk, 3, T_local), np.complex64)
        _check(self._lib.psa_sed_fs_read(self._h, k0, nk, out.ctypes.data_as(C.c_void_p)), "psa_sed_fs_read")
        return out

    def fs_write(self, t0: int, block: np.ndarray):
        block = np.ascontiguousarray(block, np.complex64)
        _check(self._lib.psa_sed_fs_write(self._h, t0, block.shape[2], block.ctypes.data_as(C.c_void_p)),
               "psa_sed_fs_write")

    def fs_finish(self, first_group: bool):
        _check(self._lib.psa_sed_fs_finish(self._h, 1 if first_group else 0), "psa_sed_fs_finish")
