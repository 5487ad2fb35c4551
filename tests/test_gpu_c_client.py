"""The C ABI used from plain C: tests/c_client/c_abi_client.c is compiled with gcc against include/dptm.h
(no HIP, no Python in the client) and linked to libdptm.so; its probabilities must equal the oracle's."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from deepptmpred_amd import _native, spec
from oracle import ptm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_client(tmp_path):
    exe = str(tmp_path / "c_abi_client")
    libdir = os.path.dirname(_native.library_path())
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_client", "c_abi_client.c"), "-o", exe,
           "-L", libdir, "-ldptm", "-Wl,-rpath," + libdir]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_plain_c():
    """include/dptm.h compiles as C99 and as C++ without any HIP or torch header."""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "dptm.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)


@pytest.mark.gpu
def test_c_client_matches_oracle(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this machine")
    exe = build_client(tmp_path)
    w = spec.synthetic_weights(7)
    n = 300
    x = spec.synthetic_inputs(n, 321)
    blob = spec.pack_blob(w).astype("<f4")
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(np.array([blob.size, n], dtype="<i8").tobytes())
        f.write(blob.tobytes())
        for a, dt in zip(x, ("u1", "u1", "u1", "<f4", "<f4")):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    outp = tmp_path / "out.bin"
    r = subprocess.run([exe, str(inp), str(outp)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    p = np.fromfile(outp, dtype="<f4").reshape(n, 2)
    with np.errstate(over="ignore"):
        ref = O.forward(w, x, np.float64)
    assert np.abs(p - ref).max() < 1e-4
