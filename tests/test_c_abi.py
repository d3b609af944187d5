"""The boundary from plain C (what cgo / a Rust extern block sees): tests/c_abi_smoke.c is compiled with gcc against
include/mina_verify.h and linked to libminaverify.so.  CPU leg: it compiles and links.  GPU leg: it runs green."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c_abi_smoke.bin")


def build_exe():
    import mina_bridge_amd as m
    assert os.path.exists(m.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", EXE, "-L", os.path.dirname(m.LIB_PATH), "-lminaverify",
                           "-Wl,-rpath," + os.path.dirname(m.LIB_PATH)])
    return EXE


def test_header_is_plain_c_and_links():
    build_exe()                                   # -Werror: the header is valid C99, every used symbol resolves


@pytest.mark.gpu
def test_c_consumer_runs_on_gpu():
    exe = build_exe()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_smoke ok" in r.stdout
