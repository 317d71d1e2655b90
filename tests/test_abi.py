"""The C ABI without a GPU: libpsa_hip.so loads, exports exactly the symbols
include/psa_hip.h declares, the ctypes table covers them, and it fails loudly (no CPU
fallback) when there is no device."""
import ctypes
import re
from pathlib import Path

import pytest

from psa_amd import _hip

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "psa_hip.h").read_text()


def declared_symbols():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(psa_[a-z0-9_]+)\s*\(", body)))


def test_header_and_binding_agree():
    names = declared_symbols()
    assert len(names) >= 25
    assert sorted(_hip.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(_hip.LIB_PATH))
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in psa_hip.h but not exported"


def test_version_and_error_string():
    lib = _hip.load_library()
    m = re.search(r"#define\s+PSA_HIP_ABI_VERSION\s+(\d+)", HEADER)
    assert lib.psa_abi_version() == int(m.group(1))
    assert isinstance(lib.psa_last_error(), bytes)


def test_constants_match_header():
    def const(name):
        return int(re.search(rf"#define\s+{name}\s+(-?\w+)", HEADER).group(1), 0)
    assert (_hip.SLOT_VELOCITIES, _hip.SLOT_POSITIONS) == (const("PSA_SLOT_VELOCITIES"), const("PSA_SLOT_POSITIONS"))
    assert (_hip.F_DISPLACEMENTS, _hip.F_INTENSITY) == (const("PSA_F_DISPLACEMENTS"), const("PSA_F_INTENSITY"))
    assert (_hip.K1_AUTO, _hip.K1_WAVE, _hip.K1_MFMA32, _hip.K1_SPLIT_BF16) == (
        const("PSA_K1_AUTO"), const("PSA_K1_WAVE"), const("PSA_K1_MFMA32"), const("PSA_K1_SPLIT_BF16"))
    assert _hip.UNIQUE_ID_BYTES == const("PSA_UNIQUE_ID_BYTES")


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must raise, never compute."""
    try:
        n = _hip.device_count()
    except _hip.PsaHipError:
        n = 0
    if n:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_hip.PsaHipError):
        _hip.Engine()


def test_pack_groups():
    import numpy as np
    assert _hip.pack_groups(None) == (None, None, 1)
    idx, off, g = _hip.pack_groups([np.array([3, 1]), np.array([], int), np.array([2])])
    assert idx.dtype == np.int32 and off.dtype == np.int64 and g == 3
    assert idx.tolist() == [3, 1, 2] and off.tolist() == [0, 2, 2, 3]
    with pytest.raises(ValueError, match="out of bounds"):
        _hip.pack_groups([np.array([-1])])


def test_pinned_pool_degrades_to_ordinary_memory_without_a_gpu():
    """No device: page-locking fails inside the library, the pool hands out a plain ndarray and keeps
    no account of it."""
    import numpy as np
    from psa_amd import _hip
    try:
        if _hip.device_count() > 0:
            pytest.skip("a GPU is present")
    except _hip.PsaHipError:
        pass                                                            # no ROCm device at all
    a = _hip.pinned_empty((1 << 19,), np.float32)
    a[:] = 2.0
    assert a.flags.writeable and float(a.sum()) == float(1 << 20) and _hip._pinned_pool._live_bytes == 0
    assert _hip.pinned_empty((8,), np.complex64).shape == (8,)          # small: never page-locked
