"""The C++ host layer (include/winterfell_hip.hpp) — the compiled-language mirror of the reference's interfaces above the C
ABI — checked against the CPU oracle by a small C++ program (tests/cpp/host_parity.cpp) built here with g++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_parity.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_parity.bin")


def build():
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", SRC, "-o", BIN, "-L" + os.path.join(ROOT, "winterfell_amd"), "-lwinterfell_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "winterfell_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)


def test_cpp_host_layer_compiles():
    """CPU side: the header and the parity program compile and link against both libraries (no GPU needed to build)."""
    build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_host_layer_parity():
    build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(out.stdout[-4000:])
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
