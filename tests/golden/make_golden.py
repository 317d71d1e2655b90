#!/usr/bin/env python3
"""
Capture golden vectors from the REAL reference (h-walk/PSA at /root/reference).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported unmodified from /root/reference/src.  Its package
`__init__` pulls in `psa.visualization.sed_plotter`, whose line 345 is Python >= 3.12
syntax (an ordinary SyntaxError on this image's Python 3.10); that one matplotlib
module -- which the SED arithmetic never touches -- is pre-seeded in `sys.modules`
with an empty `SEDPlotter` so the import proceeds.  No reference source is copied:
only inputs and the reference's outputs are written, as .npz data.
"""
import json
import os
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
REF_SRC = os.environ.get("PSA_REFERENCE_SRC", "/root/reference/src")


def import_reference():
    sys.path.insert(0, REF_SRC)
    stub = types.ModuleType("psa.visualization.sed_plotter")

    class SEDPlotter:                                  # plotting is out of scope
        def __init__(self, *a, **k):
            pass

        def generate_plot(self, *a, **k):
            pass

    stub.SEDPlotter = SEDPlotter
    sys.modules["psa.visualization.sed_plotter"] = stub
    import psa                                          # noqa: F401
    from psa.core.sed import SED
    from psa.core.sed_calculator import SEDCalculator
    from psa.core.trajectory import Trajectory
    from psa.utils.helpers import parse_direction
    return SED, SEDCalculator, Trajectory, parse_direction


def main():
    import cases as C
    SED, SEDCalculator, Trajectory, parse_direction = import_reference()
    meta = {"numpy": np.__version__, "python": sys.version.split()[0],
            "reference": "h-walk/PSA @ /root/reference (2025-05-23 snapshot)"}

    trajs = {}
    for name in C.TRAJ:
        d = C.build_traj(name)
        np.savez_compressed(HERE / f"traj_{name}.npz",
                            **{k: v for k, v in d.items() if k != "cells"},
                            cells=np.array(d["cells"]))
        trajs[name] = d

    def make_calc(tname, **ctor):
        d = trajs[tname]
        tr = Trajectory(d["positions"], d["velocities"], d["types"], d["timesteps"],
                        d["box_matrix"], d["box_lengths"], d["box_tilts"], d["dt_ps"])
        cx, cy, cz = d["cells"]
        return SEDCalculator(tr, cx, cy, cz, **ctor)

    def wide_fixture(cases, fname):
        out = {}
        for case in cases:
            calc = make_calc(case["traj"], **case.get("ctor", {}))
            mags, vecs, shape = C.k_from_spec(calc, case["k"])
            kw = C.realise_kw(case.get("kw", {}))
            if shape is not None:
                kw["k_grid_shape"] = shape
            sed = calc.calculate(mags, vecs, **kw)
            n = case["name"]
            out[f"{n}/k_mags"], out[f"{n}/k_vecs"] = mags, vecs
            out[f"{n}/sed_rows"] = sed.sed[::C.WIDE_SED_STRIDE]
            out[f"{n}/sed_shape"] = np.array(sed.sed.shape)
            out[f"{n}/is_complex"] = np.array(sed.is_complex)
            out[f"{n}/intensity"] = sed.intensity if sed.is_complex else sed.sed
        np.savez_compressed(HERE / fname, **out)
        print(f"{fname} {(HERE / fname).stat().st_size / 1024:8.1f} KiB")

    # ---- calculate() on k-lists that fill an even number of 128-row blocks (the 256-row kernel) ------
    if "--only-w256" in sys.argv:                        # (added in round 3; the other fixtures are unchanged)
        wide_fixture(C.CALC_W256_CASES, "calc_w256.npz")
        return

    # ---- calculate() on k-lists with (k, -k) pairs and repeated vectors -------------------------
    out = {}
    for case in C.CALC_SYM_CASES:
        calc = make_calc(case["traj"], **case.get("ctor", {}))
        mags, vecs, shape = C.k_from_spec(calc, case["k"])
        kw = C.realise_kw(case.get("kw", {}))
        if shape is not None:
            kw["k_grid_shape"] = shape
        sed = calc.calculate(mags, vecs, **kw)
        n = case["name"]
        out[f"{n}/k_mags"], out[f"{n}/k_vecs"] = mags, vecs
        out[f"{n}/sed_rows"] = sed.sed[::C.WIDE_SED_STRIDE]
        out[f"{n}/sed_shape"] = np.array(sed.sed.shape)
        out[f"{n}/is_complex"] = np.array(sed.is_complex)
        out[f"{n}/intensity"] = sed.intensity if sed.is_complex else sed.sed
    np.savez_compressed(HERE / "calc_sym.npz", **out)
    if "--only-sym" in sys.argv:                         # (added in round 3; the other fixtures are unchanged)
        print(f"calc_sym.npz {(HERE / 'calc_sym.npz').stat().st_size / 1024:8.1f} KiB")
        return

    wide_fixture(C.CALC_W256_CASES, "calc_w256.npz")

    # ---- calculate() cases ------------------------------------------------
    out = {}
    for case in C.CALC_CASES:
        calc = make_calc(case["traj"], **case.get("ctor", {}))
        mags, vecs, shape = C.k_from_spec(calc, case["k"])
        kw = C.realise_kw(case.get("kw", {}))
        if shape is not None:
            kw["k_grid_shape"] = shape
        sed = calc.calculate(mags, vecs, **kw)
        n = case["name"]
        out[f"{n}/k_mags"], out[f"{n}/k_vecs"] = mags, vecs
        out[f"{n}/sed"], out[f"{n}/freqs"] = sed.sed, sed.freqs
        out[f"{n}/is_complex"] = np.array(sed.is_complex)
        out[f"{n}/intensity"] = sed.intensity          # incl. the 2-D quirk (sums over k)
        out[f"{n}/grid_shape"] = np.array(sed.k_grid_shape if sed.k_grid_shape else [], int)
    # the seam itself: `_calculate_sed_for_group` on an explicit index list
    calc = make_calc("a")
    mags, vecs = calc.get_k_path("110", 2.0, 6)
    mean = np.mean(calc.traj.positions, axis=0, dtype=np.float32)
    idx = np.array([5, 1, 1, 33, 62])
    out["seam/k_vecs"], out["seam/idx"], out["seam/mean_pos"] = vecs, idx, mean
    out["seam/sed"] = calc._calculate_sed_for_group(vecs, idx, mean)
    out["seam_empty/sed"] = calc._calculate_sed_for_group(vecs, np.array([], int), mean)
    np.savez_compressed(HERE / "calc_cases.npz", **out)

    # ---- calculate() with more than 16 k-vectors (the "2 x f16" kernel's territory) ----------
    out = {}
    for case in C.CALC_WIDE_CASES:
        calc = make_calc(case["traj"], **case.get("ctor", {}))
        mags, vecs, shape = C.k_from_spec(calc, case["k"])
        kw = C.realise_kw(case.get("kw", {}))
        if shape is not None:
            kw["k_grid_shape"] = shape
        sed = calc.calculate(mags, vecs, **kw)
        n = case["name"]
        out[f"{n}/k_mags"], out[f"{n}/k_vecs"] = mags, vecs
        out[f"{n}/sed_rows"] = sed.sed[::C.WIDE_SED_STRIDE]
        out[f"{n}/sed_shape"] = np.array(sed.sed.shape)
        out[f"{n}/is_complex"] = np.array(sed.is_complex)
        out[f"{n}/intensity"] = sed.intensity if sed.is_complex else sed.sed
    np.savez_compressed(HERE / "calc_wide.npz", **out)

    # ---- BASELINE configuration 1 through the real reference (full size) -----------------------
    sys.path.insert(0, str(HERE.parent.parent))
    spec, req, d = C.c1_inputs()
    tr = Trajectory(d["positions"], d["velocities"], d["types"], d["timesteps"],
                    d["box_matrix"], d["box_lengths"], d["box_tilts"], spec.dt_ps)
    calc = SEDCalculator(tr, *spec.cells)
    mags, vecs = calc.get_k_path(req["direction"], req["bz_coverage"], req["n_k"])
    sed = calc.calculate(mags, vecs)
    inc = calc.calculate(mags, vecs, basis_atom_types=[1, 2], summation_mode="incoherent")
    rows = np.array([0, 1, 255, 256, 819, 2048, 4095])
    np.savez_compressed(HERE / "c1_reference.npz", k_mags=mags, k_vecs=vecs, intensity=sed.intensity,
                        sed_rows=sed.sed[rows], rows=rows, freqs=sed.freqs,
                        intensity_incoherent_types12=inc.sed)

    # ---- SED.save written by the reference (six .npy files incl. grid shape and phase) ---------
    import shutil
    saved = HERE / "sed_saved"
    shutil.rmtree(saved, ignore_errors=True)
    saved.mkdir()
    calc = make_calc("a")
    mags, vecs, shape = calc.get_k_grid("xy", (-1.5, 1.5), (-1.0, 1.0), 3, 4, 0.25)
    sed = calc.calculate(mags, vecs, k_grid_shape=shape)
    sed.phase = calc.calculate_chiral_phase(sed.sed[:, :, 0], sed.sed[:, :, 1], "C")
    sed.save(saved / "grid_xy_phase")
    mags, vecs = calc.get_k_path("100", 1.0, 8)
    calc.calculate(mags, vecs, basis_atom_types=[1, 2], summation_mode="incoherent").save(saved / "path_inc")

    # ---- k generators, ctor attributes -----------------------------------
    out = {}
    for i, kc in enumerate(C.KPATH_CASES):
        calc = make_calc(kc["traj"])
        mags, vecs = calc.get_k_path(kc["spec"], kc["cov"], kc["n_k"], lat_param=kc["lat"])
        out[f"kpath{i}/mags"], out[f"kpath{i}/vecs"] = mags, vecs
    calc = make_calc("a")
    for i, g in enumerate(C.KGRID_CASES):
        mags, vecs, shape = calc.get_k_grid(g["plane"], g["rx"], g["ry"], g["nx"], g["ny"], g["fixed"])
        out[f"kgrid{i}/mags"], out[f"kgrid{i}/vecs"] = mags, vecs
        out[f"kgrid{i}/shape"] = np.array(shape)
    for t in C.TRAJ:
        calc = make_calc(t)
        for nm in ("a1", "a2", "a3", "b1", "b2", "b3", "recip_vecs_prim"):
            out[f"ctor_{t}/{nm}"] = np.asarray(getattr(calc, nm))
        out[f"ctor_{t}/dt_ps"] = np.array(calc.dt_ps)
    for i, spec in enumerate(C.DIRECTION_CASES):
        out[f"dir{i}"] = parse_direction(spec)
    np.savez_compressed(HERE / "kgen_cases.npz", **out)

    # ---- chiral phase (options C / A / B) ----------------------------------
    rng = np.random.default_rng(99)
    z1 = (rng.standard_normal((12, 7)) + 1j * rng.standard_normal((12, 7))).astype(np.complex64)
    z2 = (rng.standard_normal((12, 7)) + 1j * rng.standard_normal((12, 7))).astype(np.complex64)
    z1[0, 0] = 0
    z2[1, 1] = 1e-12
    z1[2, 2] = z2[2, 2]                                 # parallel -> acos(1)
    z1[3, 3] = -z2[3, 3]                                # antiparallel
    z1[4, 4] = 1j * z2[4, 4]                            # +90 deg
    out = {"z1": z1, "z2": z2}
    calc = make_calc("c")
    for opt in ("C", "A", "B", "Q"):
        out[f"phase_{opt}"] = calc.calculate_chiral_phase(z1, z2, opt)
    out["phase_empty"] = calc.calculate_chiral_phase(z1[:0], z2[:0], "C")
    np.savez_compressed(HERE / "chiral_cases.npz", **out)

    # ---- trajectory .npy cache written by the reference's own loader ------------------------
    from psa.io.loader import TrajectoryLoader
    cache = HERE / "npy_cache"
    shutil.rmtree(cache, ignore_errors=True)
    cache.mkdir()
    (cache / "run7.lammpstrj").write_text("")            # the loader insists the source file exists
    d = trajs["c"]
    tr = Trajectory(d["positions"], d["velocities"], d["types"], d["timesteps"],
                    d["box_matrix"], d["box_lengths"], d["box_tilts"], d["dt_ps"])
    TrajectoryLoader(str(cache / "run7.lammpstrj"), dt=d["dt_ps"]).save_trajectory_npy(tr)
    back = TrajectoryLoader(str(cache / "run7.lammpstrj"), dt=d["dt_ps"]).load()
    np.savez_compressed(HERE / "npy_cache_loaded.npz", timesteps=back.timesteps, box_lengths=back.box_lengths,
                        box_tilts=back.box_tilts, dt_ps=np.array(back.dt_ps))

    # ---- iSED: the reference's reconstruction dumps (text) for three parameter sets ---------------
    ised_dir = HERE / "ised"
    shutil.rmtree(ised_dir, ignore_errors=True)
    ised_dir.mkdir()
    for name, tname, kw in C.ISED_CASES:
        make_calc(tname).ised(dump_filepath=str(ised_dir / f"{name}.dump"), **kw)
    from psa.io.writer import out_to_qdump               # the writer alone, tilted cell
    out_to_qdump(str(ised_dir / "writer_tilted.dump"), *C.qdump_tilted_inputs())

    (HERE / "META.json").write_text(json.dumps(meta, indent=1) + "\n")
    for f in sorted(HERE.glob("*.npz")):
        print(f"{f.name:24s} {f.stat().st_size/1024:8.1f} KiB")


if __name__ == "__main__":
    main()
