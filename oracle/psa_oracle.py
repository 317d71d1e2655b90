"""
CPU ORACLE for the PSA SED hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This module is a NumPy restatement of the reference algorithm (h-walk/PSA,
`src/psa/core/sed_calculator.py`, `src/psa/core/sed.py`,
`src/psa/utils/helpers.py`).  It exists only to CHECK the HIP path:

  * imported by `tests/`, by `__graft_entry__.smoke()` and by the
    `cpu_baseline` leg of `bench.py` -- nowhere else;
  * `psa_amd/` (the product) never imports it and has no CPU fallback.

Parity status: PINNED.  Every function below is checked in
`tests/test_oracle_golden.py` against golden vectors captured from the real
reference running in the build container (`tests/golden/make_golden.py`
imports `/root/reference/src/psa` unmodified, with only the matplotlib plotter
module stubbed because its line 345 needs Python >= 3.12).  The reference's own
unit tests hold no vectors for the projection/FFT, so the
captured fixtures are the pin.

Arithmetic follows the reference operation-for-operation and dtype-for-dtype
(float32 mean accumulator, float32 sgemm phase argument, complex64 exp,
complex64 einsum, complex64 pocketfft under NumPy >= 2) because the 1e-5
intensity tolerance is dominated by those float32 roundings.

All "ref:" citations are relative to /root/reference/src/psa/.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "unit_direction", "reciprocal_lattice", "k_path", "k_grid",
    "mean_positions", "phase_table", "project_group", "sed_for_group",
    "resolve_groups", "calculate", "intensity", "chiral_phase",
]


# --------------------------------------------------------------------------
# direction parsing            ref: utils/helpers.py:13-109
# --------------------------------------------------------------------------
_S2 = 1.0 / np.sqrt(2)
_S3 = 1.0 / np.sqrt(3)
_NAMED = {
    "x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1),
    "100": (1, 0, 0), "010": (0, 1, 0), "001": (0, 0, 1),
    "xy": (_S2, _S2, 0), "yx": (_S2, _S2, 0), "110": (_S2, _S2, 0),
    "xz": (_S2, 0, _S2), "zx": (_S2, 0, _S2),
    "yz": (0, _S2, _S2), "zy": (0, _S2, _S2),
    "xyz": (_S3, _S3, _S3), "111": (_S3, _S3, _S3),
}


def _from_angle(deg) -> np.ndarray:
    rad = np.deg2rad(deg)
    return np.array([np.cos(rad), np.sin(rad), 0.0], dtype=np.float32)


def unit_direction(spec) -> np.ndarray:
    """ref: utils/helpers.py:13-109 (`parse_direction`)."""
    if isinstance(spec, (int, float)):                      # helpers.py:33-35
        vec = _from_angle(float(spec))
    elif isinstance(spec, str):                             # helpers.py:37-70
        key = spec.lower()
        if key in _NAMED:
            vec = np.array(_NAMED[key], dtype=np.float32)
        else:
            try:
                vec = _from_angle(float(spec))
            except ValueError:
                parts = spec.replace(",", " ").split()
                try:
                    if len(parts) != 3:
                        raise ValueError
                    vec = np.array([float(p) for p in parts], dtype=np.float32)
                except ValueError:
                    raise ValueError(f"Unknown direction string: {spec}.")
    elif isinstance(spec, (list, tuple, np.ndarray)):       # helpers.py:72-86
        arr = np.asarray(spec, dtype=np.float32).squeeze()
        if arr.ndim == 0:
            vec = _from_angle(arr.item())
        elif arr.ndim == 1:
            if arr.size == 1:
                vec = _from_angle(arr[0])
            elif arr.size == 3:
                vec = arr
            else:
                raise ValueError(
                    f"Direction array must have 1 (angle) or 3 (vector) components, got {arr.size}")
        else:
            raise ValueError(
                f"Direction array has too many dims: {arr.ndim}, expected 0 or 1 (squeezed).")
    elif isinstance(spec, dict):                            # helpers.py:88-99
        if "angle" in spec:
            vec = _from_angle(float(spec["angle"]))
        elif any(k in spec for k in ("h", "k", "l")):
            vec = np.array([float(spec.get(k, 0.0)) for k in ("h", "k", "l")],
                           dtype=np.float32)
        else:
            raise ValueError("Direction dict must contain 'angle' or Miller indices ('h','k','l').")
    else:
        raise TypeError(f"Unsupported direction type: {type(spec)}")

    if np.allclose(vec, 0, atol=1e-8):                      # helpers.py:101-102
        raise ValueError("Direction vector is zero. For k-path, direction must be non-zero if n_k > 1.")
    nrm = np.linalg.norm(vec)
    if nrm < 1e-9:                                          # helpers.py:105-107
        return vec
    return vec / nrm


# --------------------------------------------------------------------------
# reciprocal lattice, k generators      ref: core/sed_calculator.py:40-56,86-180
# --------------------------------------------------------------------------
def reciprocal_lattice(box_matrix, nx, ny, nz):
    """ref: core/sed_calculator.py:40-56.  Returns (a(3,3), b(3,3), recip f32)."""
    a = [box_matrix[0, :] / nx, box_matrix[1, :] / ny, box_matrix[2, :] / nz]
    vol = np.abs(np.dot(a[0], np.cross(a[1], a[2])))
    b = [(2 * np.pi / vol) * np.cross(a[1], a[2]),
         (2 * np.pi / vol) * np.cross(a[2], a[0]),
         (2 * np.pi / vol) * np.cross(a[0], a[1])]
    return a, b, np.vstack(b).astype(np.float32)


def k_path(box_matrix, nx, ny, nz, direction_spec, bz_coverage, n_k, lat_param=None):
    """ref: core/sed_calculator.py:86-125 (`get_k_path`)."""
    khat = unit_direction(direction_spec)
    a, b, _ = reciprocal_lattice(box_matrix, nx, ny, nz)
    if lat_param is None or lat_param <= 1e-6:              # :91-114
        extent = max(abs(np.dot(khat, bi)) for bi in b)
        if not extent > 1e-6:
            na = np.linalg.norm(a[0])
            if not na > 1e-6:
                raise ValueError("Invalid/small lattice_param for k-path & reciprocal "
                                 "projections too small for auto-detection.")
            extent = 2 * np.pi / na
    else:                                                   # :115-118
        extent = 2 * np.pi / lat_param
    kmax = bz_coverage * extent                             # :120
    if n_k < 1:
        raise ValueError("n_k (k-points) must be >= 1.")
    if n_k > 1:                                             # :123
        mags = np.linspace(0, kmax, n_k, dtype=np.float32)
    else:
        mags = np.array([0.0 if np.isclose(kmax, 0) else kmax], dtype=np.float32)
    return mags, np.outer(mags, khat).astype(np.float32)    # :124


def k_grid(plane, k_range_x, k_range_y, n_kx, n_ky, k_fixed_val=0.0):
    """ref: core/sed_calculator.py:127-180 (`get_k_grid`).  First range = outer loop."""
    if n_kx <= 0 or n_ky <= 0:
        raise ValueError("Number of k-points (n_kx, n_ky) must be positive.")
    first = np.linspace(k_range_x[0], k_range_x[1], n_kx, dtype=np.float32)
    second = np.linspace(k_range_y[0], k_range_y[1], n_ky, dtype=np.float32)
    p = plane.lower()
    rows = []
    for u in first:
        for w in second:
            if p == "xy":
                rows.append([u, w, k_fixed_val])            # :159-162
            elif p == "yz":
                rows.append([k_fixed_val, u, w])            # :163-166
            elif p == "zx":
                rows.append([w, k_fixed_val, u])            # :167-170
            else:
                raise ValueError(f"Invalid plane specified: {plane}. Must be 'xy', 'yz', or 'zx'.")
    return (np.array([], dtype=np.float32),
            np.array(rows, dtype=np.float32),
            (n_kx, n_ky))


# --------------------------------------------------------------------------
# the numerical core                    ref: core/sed_calculator.py:58-84,205
# --------------------------------------------------------------------------
def mean_positions(positions) -> np.ndarray:
    """ref: core/sed_calculator.py:205 -- float32 accumulator, sequential in t."""
    return np.mean(positions, axis=0, dtype=np.float32)


def phase_table(k_vectors, mean_pos_group) -> np.ndarray:
    """ref: core/sed_calculator.py:78 -- (K, N_g) complex64 = exp(i k.r)."""
    return np.exp(1j * np.dot(k_vectors, mean_pos_group.T))


def project_group(data_group, phase) -> np.ndarray:
    """ref: core/sed_calculator.py:75,80-81 -- q[t,k,c] = sum_a d[t,a,c] P[k,a]."""
    n_t = data_group.shape[0]
    q = np.zeros((n_t, phase.shape[0], 3), dtype=np.complex64)
    for c in range(3):
        q[:, :, c] = np.einsum("ta,ak->tk", data_group[:, :, c], phase.T, optimize=True)
    return q


def sed_for_group(positions, velocities, k_vectors, idx, mean_pos_all,
                  use_displacements=False) -> np.ndarray:
    """ref: core/sed_calculator.py:58-84 (`_calculate_sed_for_group`)."""
    n_t = velocities.shape[0]
    idx = np.asarray(idx)
    if idx.size == 0:                                       # :64-65
        return np.zeros((n_t, len(k_vectors), 3), dtype=np.complex64)
    mp = mean_pos_all[idx]                                  # :67
    if use_displacements:                                   # :69-72
        data = positions[:, idx, :] - mp[None, :, :]
    else:
        data = velocities[:, idx, :]
    q = project_group(data, phase_table(k_vectors, mp))
    if n_t == 0:
        return np.array([], dtype=np.complex64).reshape(0, len(k_vectors), 3)
    return (np.fft.fft(q, axis=0) / n_t).astype(np.complex64)   # :83-84


def resolve_groups(types, n_atoms, basis_atom_indices, basis_atom_types, summation_mode):
    """ref: core/sed_calculator.py:208-266 -- list of index arrays (one per group)."""
    groups = []
    if basis_atom_types is not None:                        # :211-234
        tg = []
        if isinstance(basis_atom_types, list) and len(basis_atom_types) > 0:
            if all(isinstance(x, list) for x in basis_atom_types):
                tg = basis_atom_types
            elif all(isinstance(x, int) for x in basis_atom_types):
                tg = ([[t] for t in basis_atom_types] if summation_mode == "incoherent"
                      else [list(basis_atom_types)])
            else:
                raise ValueError("basis_atom_types must be a list of ints or a list of lists of ints.")
        elif isinstance(basis_atom_types, int):
            tg = [[basis_atom_types]]
        for g in tg:
            sel = np.where(np.isin(types, g))[0]
            if sel.size > 0:
                groups.append(sel)
    elif basis_atom_indices is not None:                    # :236-260
        cand = []
        if isinstance(basis_atom_indices, list):
            if len(basis_atom_indices) == 0:
                pass
            elif all(isinstance(x, list) for x in basis_atom_indices):
                cand = [np.asarray(s, dtype=int) for s in basis_atom_indices]
                cand = [c for c in cand if c.size > 0]
            elif all(isinstance(x, int) for x in basis_atom_indices):
                arr = np.asarray(basis_atom_indices, dtype=int)
                if arr.size > 0:
                    cand.append(arr)
            else:
                raise ValueError("basis_atom_indices must be a list of ints or a list of lists of ints.")
        elif isinstance(basis_atom_indices, np.ndarray):
            if basis_atom_indices.ndim == 1 and basis_atom_indices.size > 0:
                cand.append(basis_atom_indices.astype(int))
        for g in cand:
            if np.any(g >= n_atoms) or np.any(g < 0):
                raise ValueError("Atom indices in basis out of bounds.")
            if g.size > 0:
                groups.append(g)
    if not groups:                                          # :262-266
        groups.append(np.arange(n_atoms))
    return groups


def calculate(positions, velocities, types, dt_ps, k_vectors,
              basis_atom_indices=None, basis_atom_types=None,
              summation_mode="coherent", k_chunk_size=500, use_displacements=False):
    """ref: core/sed_calculator.py:182-336 (`calculate`).

    Returns (sed, freqs, is_complex): sed is (T,K,3) complex64 when coherent (or
    <= 1 group) else (T,K) float32 = sum_g sum_c |S_g|^2.
    """
    if summation_mode not in ("coherent", "incoherent"):    # :190-191
        raise ValueError(f"summation_mode must be 'coherent' or 'incoherent', got {summation_mode}")
    n_t, n_atoms = velocities.shape[0], velocities.shape[1]
    if n_t == 0 or n_atoms == 0:                            # :193-203
        return (np.array([], dtype=np.complex64).reshape(0, 0, 3),
                np.array([], dtype=np.float32), True)
    mean_all = mean_positions(positions)                    # :205
    freqs = np.fft.fftfreq(n_t, d=dt_ps)                    # :206
    groups = resolve_groups(types, n_atoms, basis_atom_indices, basis_atom_types, summation_mode)

    n_k = len(k_vectors)                                    # :269-272
    chunk = min(max(1, k_chunk_size), n_k) if n_k > 0 else 1
    n_chunks = (n_k + chunk - 1) // chunk if n_k > 0 else 0
    is_complex = summation_mode == "coherent" or len(groups) <= 1   # :276
    if is_complex:
        out = np.zeros((len(freqs), n_k, 3), dtype=np.complex64)
    else:
        out = np.zeros((len(freqs), n_k), dtype=np.float32)

    for ic in range(n_chunks):                              # :287-327
        lo, hi = ic * chunk, min((ic + 1) * chunk, n_k)
        kc = k_vectors[lo:hi]
        if kc.shape[0] == 0:
            continue
        if is_complex:
            idx = (np.unique(np.concatenate(groups)).astype(int)
                   if len(groups) > 1 else groups[0])       # :297-300
            if idx.size == 0:
                continue
            out[:, lo:hi, :] = sed_for_group(positions, velocities, kc, idx, mean_all,
                                             use_displacements)
        else:
            acc = np.zeros((len(freqs), kc.shape[0]), dtype=np.float32)
            for g in groups:
                if g.size == 0:
                    continue
                s = sed_for_group(positions, velocities, kc, g, mean_all, use_displacements)
                acc += np.sum(np.abs(s) ** 2, axis=-1)      # :325
            out[:, lo:hi] = acc
    return out, freqs, is_complex


def intensity(sed) -> np.ndarray:
    """ref: core/sed.py:22-24 (`SED.intensity`)."""
    return np.sum(np.abs(sed) ** 2, axis=-1).astype(np.float32)


# --------------------------------------------------------------------------
# chiral phase                          ref: core/sed_calculator.py:338-371
# --------------------------------------------------------------------------
def chiral_phase(z1, z2, opt="C") -> np.ndarray:
    """ref: core/sed_calculator.py:338-371.  Options A/B are evaluated elementwise in
    float64 then stored as float32, as the reference's Python-scalar loop does."""
    if z1.shape != z2.shape:
        raise ValueError("Z1 and Z2 shapes must match for chiral phase.")
    if z1.size == 0:
        return np.array([], dtype=np.float32).reshape(z1.shape)
    if opt == "C":                                          # :344-350
        d = np.angle(z1) - np.angle(z2)
        d = (d + np.pi) % (2 * np.pi) - np.pi
        hi = d > (np.pi / 2)
        d[hi] = np.pi - d[hi]
        lo = d < (-np.pi / 2)
        d[lo] = -np.pi - d[lo]
        return d.astype(np.float32)
    out = np.zeros(z1.shape, dtype=np.float32)              # :352-371
    if opt not in ("A", "B"):
        return out
    # The reference squares the float32 parts as NumPy float32 scalars (v1r**2 etc.
    # stay float32), takes float32 sqrt, and only the arccos/arcsin argument is
    # float32 too; reproduce those dtypes exactly.
    a, b = z1.real.astype(np.float32), z1.imag.astype(np.float32)
    c, d = z2.real.astype(np.float32), z2.imag.astype(np.float32)
    m1sq = a * a + b * b
    m2sq = c * c + d * d
    ok = ~((m1sq < 1e-18) | (m2sq < 1e-18))
    with np.errstate(invalid="ignore", divide="ignore"):
        den = np.sqrt(m1sq) * np.sqrt(m2sq)
        if opt == "A":
            val = np.arccos(np.clip((a * c + b * d) / den, -1.0, 1.0))
        else:
            val = np.arcsin(np.clip((a * d - b * c) / den, -1.0, 1.0))
    out[ok] = val[ok].astype(np.float32)
    return out
