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
           "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "winterfell_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread"]
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


EX_SRC = os.path.join(ROOT, "examples", "fib_small.cpp")
EX_BIN = os.path.join(ROOT, "examples", "fib_small.bin")


def build_example():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), EX_SRC, "-o", EX_BIN,
                           "-L" + os.path.join(ROOT, "winterfell_amd"), "-lwinterfell_hip", "-Wl,-rpath," + os.path.join(ROOT, "winterfell_amd")])


def test_cpp_example_compiles():
    build_example()
    assert os.path.exists(EX_BIN)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,hash_id,hname,D", [(10, 0, "Blake3_256", 2), (9, 1, "Rp64_256", 1), (11, 2, "Sha3_256", 3)])
def test_cpp_and_python_host_layers_produce_the_same_proof(log_n, hash_id, hname, D):
    """examples/fib_small.cpp (C++ host layer: its own coin, channel and step order) and winterfell_amd.prover.prove() (Python
    host layer) run the same fib_small instance: both transcripts must agree on every commitment, the proof-of-work nonce and
    the query positions — two independent drivers of the same device pipeline."""
    import json

    import numpy as np
    import winterfell_amd
    from winterfell_amd import air as wair, crypto, prover
    from winterfell_amd.math import fields
    build_example()
    out = subprocess.run([EX_BIN, str(log_n), str(hash_id), str(D), "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    cpp = json.loads(out.stdout.strip().splitlines()[-1])
    f, n = fields.f64, 1 << log_n
    a, b, c0, c1 = 1, 1, [], []
    for _ in range(n):
        c0.append(a)
        c1.append(b)
        a = (a + b) % f.M
        b = (a + b) % f.M
    trace = np.stack([f.pack([f.new(v) for v in c0]), f.pack([f.new(v) for v in c1])])
    ctx = winterfell_amd.default_context()
    hasher = getattr(crypto, hname)
    options = prover.ProofOptions(28, 8, 16, ext_degree=D, fri_folding_factor=8, fri_remainder_max_degree=127)
    proof = prover.prove(wair.FibSmall(n, f.new(c1[-1]), 8, f), prover.ColMatrix(trace, 1, ctx, f), options, hasher, [f.new(c1[-1])])
    assert cpp["trace_root"] == bytes(proof.trace_commitment).hex()
    assert cpp["constraint_root"] == bytes(proof.constraint_commitment).hex()
    assert cpp["last_fri_commitment"] == bytes(proof.commitments[-1]).hex()
    assert cpp["pow_nonce"] == proof.pow_nonce
    assert cpp["num_unique_queries"] == len(proof.query_positions) and cpp["first_position"] == proof.query_positions[0]
    assert cpp["fri_layers"] == proof.fri_proof.num_layers() and cpp["fri_remainder_len"] == proof.fri_proof.num_remainder_elements()


PIPE_SRC = os.path.join(ROOT, "tools", "host_pipeline_bench.cpp")
PIPE_BIN = os.path.join(ROOT, "tools", "host_pipeline_bench.bin")


def build_pipeline_bench():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), PIPE_SRC, "-o", PIPE_BIN,
                           "-L" + os.path.join(ROOT, "winterfell_amd"), "-lwinterfell_hip", "-Wl,-rpath," + os.path.join(ROOT, "winterfell_amd")])


def test_host_pipeline_bench_compiles():
    build_pipeline_bench()
    assert os.path.exists(PIPE_BIN)


@pytest.mark.gpu
@pytest.mark.parametrize("field,log_n,cols,parts,pin,group", [(0, 14, 20, 1, 1, 8), (0, 12, 5, 1, 0, 2), (1, 13, 16, 4, 1, 8)])
def test_pipelined_host_entry_equals_the_serial_one(field, log_n, cols, parts, pin, group):
    """wf::new_trace_lde_from_host (uploads, group-wise interpolation, polynomial downloads and LDE + commit overlapped on three
    contexts) against the serial upload -> wf_build_trace_commitment -> download sequence: the same polynomials word for word and the
    same root, page-locked and pageable host columns, ragged last group, f64 and f128 with partitions.  (The serial call itself is held
    against the oracle by tests/cpp/host_parity.cpp and the Python suites.)"""
    import json
    build_pipeline_bench()
    out = subprocess.run([PIPE_BIN, str(field), str(log_n), str(cols), str(parts), "2", str(pin), str(group)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["same_polys_and_root"] is True and res["pipelined_total_ms"] > 0 and res["h2d_trace_ms"] > 0
