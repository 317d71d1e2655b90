"""
iSED (`SEDCalculator.ised`, reference src/psa/core/sed_calculator.py:373-588) and the LAMMPS dump
it writes (`out_to_qdump`, reference src/psa/io/writer.py:139-228), against dumps the reference
itself produced for the same inputs (tests/golden/ised/*.dump, made by make_golden.py).

The CPU tests drive the host logic through the oracle-backed engine double; the `gpu` tests drive
the same calls through the HIP library.
"""
import logging
from pathlib import Path

import numpy as np
import pytest

from conftest import make_calculator
from golden import cases as C
from oracle_engine import OracleEngine
from psa_amd.io import out_to_qdump

GOLD = Path(__file__).parent / "golden" / "ised"


def parse_dump(path):
    """-> list of (header lines, (n_atoms, 5) array) per frame."""
    lines = Path(path).read_text().splitlines()
    frames, i = [], 0
    while i < len(lines):
        assert lines[i] == "ITEM: TIMESTEP"
        n_atoms = int(lines[i + 3])
        j = i + 4
        while not lines[j].startswith("ITEM: ATOMS"):
            j += 1
        header = lines[i:j + 1]
        body = np.array([ln.split() for ln in lines[j + 1:j + 1 + n_atoms]], dtype=float)
        frames.append((header, body))
        i = j + 1 + n_atoms
    return frames


def assert_same_dump(ours, theirs, atol):
    a, b = parse_dump(ours), parse_dump(theirs)
    assert len(a) == len(b)
    for (ha, xa), (hb, xb) in zip(a, b):
        assert ha == hb                                    # timestep, atom count, box bounds: verbatim
        assert np.array_equal(xa[:, :2], xb[:, :2])        # ids and types
        # coordinates are printed with 6 decimals: allow the SED's 1e-5 relative tolerance on the
        # wiggle plus one unit in the last printed place
        np.testing.assert_allclose(xa[:, 2:], xb[:, 2:], rtol=0, atol=atol)


def wiggle_scale(theirs, d):
    mean = np.mean(d["positions"], axis=0, dtype=np.float32)
    return max(np.abs(x[:, 2:] - mean).max() for _, x in parse_dump(theirs))


@pytest.mark.parametrize("name,tname,kw", C.ISED_CASES, ids=[c[0] for c in C.ISED_CASES])
def test_ised_matches_reference_dump_host(name, tname, kw, trajs, tmp_path):
    calc = make_calculator(trajs[tname]).attach(engine=OracleEngine())
    out = tmp_path / f"{name}.dump"
    calc.ised(dump_filepath=str(out), **kw)
    assert_same_dump(out, GOLD / f"{name}.dump", atol=1e-5 * wiggle_scale(GOLD / f"{name}.dump", trajs[tname]) + 1.01e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,tname,kw", C.ISED_CASES, ids=[c[0] for c in C.ISED_CASES])
def test_ised_matches_reference_dump_gpu(name, tname, kw, trajs, engine, tmp_path):
    calc = make_calculator(trajs[tname]).attach(engine=engine)
    out = tmp_path / f"{name}.dump"
    calc.ised(dump_filepath=str(out), **kw)
    assert_same_dump(out, GOLD / f"{name}.dump", atol=1e-5 * wiggle_scale(GOLD / f"{name}.dump", trajs[tname]) + 1.01e-6)


def test_ised_golden_has_motion(trajs):
    """The fixtures are not trivially 'mean positions only'."""
    for name, tname, _ in C.ISED_CASES:
        assert wiggle_scale(GOLD / f"{name}.dump", trajs[tname]) > 1e-3


def test_ised_group_resolution(trajs):
    calc = make_calculator(trajs["a"]).attach(engine=OracleEngine())
    n = calc.traj.n_atoms
    g = calc._ised_groups(None, None)
    assert len(g) == 1 and np.array_equal(g[0], np.arange(n))
    g = calc._ised_groups([3, 1, 2], [1])                   # indices win; flat list = one group
    assert len(g) == 1 and list(g[0]) == [3, 1, 2]
    g = calc._ised_groups([[0, 1], [], [5]], None)          # empty groups are dropped
    assert [list(x) for x in g] == [[0, 1], [5]]
    g = calc._ised_groups(None, [1, 2, 9])                  # flat type list = one group per type
    assert len(g) == 2
    assert np.array_equal(g[0], np.flatnonzero(calc.traj.types == 1))
    g = calc._ised_groups(None, [[1, 3], [2]])
    assert np.array_equal(g[0], np.flatnonzero(np.isin(calc.traj.types, [1, 3])))
    with pytest.raises(ValueError, match="out of bounds"):
        calc._ised_groups([0, n], None)
    with pytest.raises(ValueError, match=r"group \[-1\] out of bounds"):
        calc._ised_groups([[0], [-1]], None)


def test_ised_no_groups_writes_nothing(trajs, tmp_path, caplog):
    calc = make_calculator(trajs["a"]).attach(engine=OracleEngine())
    out = tmp_path / "none.dump"
    with caplog.at_level(logging.ERROR):
        calc.ised("x", 0.5, 10.0, C.A_SI, nk_on_path=4, basis_atom_types_ised=[9], dump_filepath=str(out))
    assert not out.exists() and "No atom groups" in caplog.text


def test_ised_numeric_rescale_is_linear(trajs, tmp_path):
    d = trajs["c"]
    mean = np.mean(d["positions"], axis=0, dtype=np.float32)
    kw = dict(k_dir_spec="z", k_target=0.3, w_target=20.0, char_len_k_path=C.A_SI, nk_on_path=4, n_recon_frames=3)
    outs = []
    for s in (1.0, 4.0):
        calc = make_calculator(d).attach(engine=OracleEngine())
        p = tmp_path / f"s{s}.dump"
        calc.ised(dump_filepath=str(p), rescale_factor=s, **kw)
        outs.append(np.stack([x[:, 2:] for _, x in parse_dump(p)]) - mean)
    np.testing.assert_allclose(outs[1], 4.0 * outs[0], atol=6e-6)


def test_qdump_layout(tmp_path):
    pos = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3) / 7
    box = np.diag([4.0, 5.0, 6.0]).astype(np.float32)
    p = tmp_path / "sub" / "o.dump"                          # parent directory is created
    out_to_qdump(str(p), pos, np.array([1, 2, 2]), box)
    text = p.read_text().splitlines()
    assert text[:9] == ["ITEM: TIMESTEP", "0", "ITEM: NUMBER OF ATOMS", "3", "ITEM: BOX BOUNDS pp pp pp",
                        "0.00000000 4.00000000", "0.00000000 5.00000000", "0.00000000 6.00000000",
                        "ITEM: ATOMS id type x y z"]
    assert text[9] == "1 1 0.000000 0.142857 0.285714"
    assert text[12:14] == ["ITEM: TIMESTEP", "1"]
    assert len(text) == 2 * (9 + 3)
    box[0, 1], box[0, 2], box[1, 2] = 0.5, -0.25, 1.0
    out_to_qdump(str(p), pos[:1], np.array([1, 2, 2]), box)
    text = p.read_text().splitlines()
    assert text[4:8] == ["ITEM: BOX BOUNDS xy xz yz pp pp pp", "-0.25000000 4.50000000 0.50000000",
                         "0.00000000 6.00000000 -0.25000000", "0.00000000 6.00000000 1.00000000"]


def test_qdump_tilted_cell_is_verbatim_reference_output(tmp_path):
    p = tmp_path / "t.dump"
    out_to_qdump(str(p), *C.qdump_tilted_inputs())
    assert p.read_text() == (GOLD / "writer_tilted.dump").read_text()
