"""The C ABI from plain C: examples/abi_client.c (C99, gcc -pedantic, nothing but include/psa_hip.h and
libpsa_hip.so) is what a host program with a C FFI -- not Python -- does to run the hot path.  Built
here without a GPU (the header is C, every symbol links); on the GPU box it runs one coherent
calculation, whole trajectory and an index list, and is compared with the oracle."""
import shutil
import struct
import subprocess

import numpy as np
import pytest

import oracle.psa_oracle as O
from conftest import ROOT, rel_max

TOL = 1e-5      # max|dI| / max|I|


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = tmp_path_factory.mktemp("abi_client") / "abi_client"
    lib_dir = ROOT / "psa_amd" / "csrc"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", f"-I{ROOT / 'include'}",
                    str(ROOT / "examples" / "abi_client.c"), f"-L{lib_dir}", "-lpsa_hip", f"-Wl,-rpath,{lib_dir}",
                    "-o", str(exe)], check=True)
    return exe


def test_c_client_builds_against_the_header_and_links(client):
    res = subprocess.run([str(client)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "usage" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n_atoms, n_frames, n_k, with_idx", [(96, 200, 40, False), (150, 128, 24, True), (64, 96, 7, False)])
def test_c_client_matches_the_oracle(client, tmp_path, n_atoms, n_frames, n_k, with_idx):
    rng = np.random.default_rng(n_atoms + n_k)
    r0 = rng.uniform(0, 30, (n_atoms, 3))
    pos = (r0[None] + 0.05 * rng.standard_normal((n_frames, n_atoms, 3))).astype(np.float32)
    vel = rng.standard_normal((n_frames, n_atoms, 3)).astype(np.float32)
    kv = rng.uniform(-2, 2, (n_k, 3)).astype(np.float32)
    idx = np.concatenate([rng.permutation(n_atoms)[: n_atoms // 2], [5, 5]]).astype(np.int32) if with_idx else np.zeros(0, np.int32)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(src, "wb") as f:
        f.write(struct.pack("<4q", n_frames, n_atoms, n_k, idx.size))
        for a in (pos, vel, kv, idx):
            f.write(np.ascontiguousarray(a).tobytes())
    res = subprocess.run([str(client), str(src), str(dst)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    raw = np.fromfile(dst, dtype=np.float32)
    n_sed = n_frames * n_k * 3 * 2
    assert raw.size == n_sed + n_frames * n_k
    sed = raw[:n_sed].view(np.complex64).reshape(n_frames, n_k, 3)
    inten = raw[n_sed:].reshape(n_frames, n_k)
    mean = O.mean_positions(pos)
    sel = idx if with_idx else np.arange(n_atoms)
    ref = O.sed_for_group(pos, vel, kv, sel, mean)
    assert rel_max(sed, ref) <= TOL
    assert rel_max(inten, O.intensity(ref)) <= TOL
