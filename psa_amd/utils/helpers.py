"""
Direction parsing for k-paths.

Mirror of `psa.utils.helpers.parse_direction` (reference src/psa/utils/helpers.py:13-109):
same accepted inputs, same float32 results, same ValueError / TypeError behaviour.  The
reference's unrelated dict/dir helpers (`update_dict_recursively`, `ensure_directory`,
...) are outside the SED hot path and are not provided.
"""
from __future__ import annotations

import logging
import math
from typing import Dict, List, Tuple, Union

import numpy as np

logger = logging.getLogger(__name__)

DirectionSpec = Union[str, int, float, List[float], Tuple[float, ...], np.ndarray, Dict[str, float]]

_R2, _R3 = 1 / np.sqrt(2), 1 / np.sqrt(3)
# named axes and Miller strings understood by the reference (helpers.py:39-52)
_AXES = {}
for _names, _v in (
    (("x", "100"), (1, 0, 0)), (("y", "010"), (0, 1, 0)), (("z", "001"), (0, 0, 1)),
    (("xy", "yx", "110"), (_R2, _R2, 0)), (("xz", "zx"), (_R2, 0, _R2)),
    (("yz", "zy"), (0, _R2, _R2)), (("xyz", "111"), (_R3, _R3, _R3)),
):
    for _n in _names:
        _AXES[_n] = _v


def _in_plane(angle_deg) -> np.ndarray:
    """Unit vector at `angle_deg` degrees from +x in the xy plane (helpers.py:34-35)."""
    phi = np.deg2rad(angle_deg)
    return np.array([np.cos(phi), np.sin(phi), 0.0], dtype=np.float32)


def _from_text(text: str) -> np.ndarray:
    named = _AXES.get(text.lower())
    if named is not None:
        return np.array(named, dtype=np.float32)
    try:                                    # a bare number is an in-plane angle
        return _in_plane(float(text))
    except ValueError:
        pass
    fields = text.replace(",", " ").split()  # "h k l" / "x,y,z"
    if len(fields) == 3:
        try:
            return np.array([float(f) for f in fields], dtype=np.float32)
        except ValueError:
            pass
    raise ValueError(f"Unknown direction string: {text}.")


def _from_sequence(seq) -> np.ndarray:
    arr = np.asarray(seq, dtype=np.float32).squeeze()
    if arr.ndim == 0:
        return _in_plane(arr.item())
    if arr.ndim > 1:
        raise ValueError(f"Direction array has too many dims: {arr.ndim}, expected 0 or 1 (squeezed).")
    if arr.size == 1:
        return _in_plane(arr[0])
    if arr.size == 3:
        return arr
    raise ValueError(f"Direction array must have 1 (angle) or 3 (vector) components, got {arr.size}")


def _from_mapping(d: dict) -> np.ndarray:
    if "angle" in d:
        return _in_plane(float(d["angle"]))
    if any(key in d for key in "hkl"):
        return np.array([float(d.get(key, 0.0)) for key in "hkl"], dtype=np.float32)
    raise ValueError("Direction dict must contain 'angle' or Miller indices ('h','k','l').")


def parse_direction(direction_spec: DirectionSpec) -> np.ndarray:
    """Turn a direction specification into a normalised float32 3-vector.

    Accepted forms (reference helpers.py:13-109): axis / Miller strings ('x', 'xy', '110',
    ...), numeric strings and numbers (angle in degrees in the xy plane), 'x,y,z' strings,
    3-sequences, 1-sequences / 0-d arrays (angle), {'angle': deg}, {'h':..,'k':..,'l':..}.
    Raises ValueError for unknown / zero directions, TypeError for unsupported types.
    """
    if isinstance(direction_spec, (int, float)):
        vec = _in_plane(float(direction_spec))
    elif isinstance(direction_spec, str):
        vec = _from_text(direction_spec)
    elif isinstance(direction_spec, (list, tuple, np.ndarray)):
        vec = _from_sequence(direction_spec)
    elif isinstance(direction_spec, dict):
        vec = _from_mapping(direction_spec)
    else:
        raise TypeError(f"Unsupported direction type: {type(direction_spec)}")

    if np.allclose(vec, 0, atol=1e-8):
        raise ValueError("Direction vector is zero. For k-path, direction must be non-zero if n_k > 1.")
    length = np.linalg.norm(vec)
    if length < 1e-9:
        logger.warning("Direction vector norm (%.2e) is very small, returning unnormalized vector.", length)
        return vec
    return vec / length
