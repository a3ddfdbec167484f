#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.json.

Two kinds of vectors, both committed so the GPU box (which has no /root/reference) can use them:

  * "reference": values parsed verbatim out of the reference's OWN test sources (the only fixed
    vectors the reference holds for this path — SURVEY.md section 8(c)).
  * "derived": values computed here by INDEPENDENT means — Python big-int arithmetic straight from the
    mathematical definition (DFT sum, Horner evaluation, dense MDS) and the upstream BLAKE3 C code that
    LLVM bundles (llvm_blake3_hasher_*).  Nothing here uses the oracle/ or the HIP code.

Run only in the build container:  python tests/golden/make_golden.py
"""
import ctypes, json, pathlib, re

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent / "reference_vectors.json"
P = 2**64 - 2**32 + 1


def ints(s):
    return [int(x) for x in re.findall(r"\d+", s)]


def parse_leaves(src, name):
    m = re.search(r"static %s: \[\[u8; 32\]; \d+\] = \[(.*?)\n\];" % name, src, re.S)
    v = ints(m.group(1))
    assert len(v) % 32 == 0
    return [v[i:i + 32] for i in range(0, len(v), 32)]


def new_vals(s):
    return [int(x) for x in re.findall(r"BaseElement::new\((\d+)\)", s)]


# ---- independent BLAKE3 (LLVM bundles the upstream C implementation) --------------------------------
_llvm = None
for cand in ("/usr/lib/x86_64-linux-gnu/libLLVM-15.so.1", "/opt/rocm/lib/llvm/lib/libclang-cpp.so"):
    try:
        _llvm = ctypes.CDLL(cand)
        _llvm.llvm_blake3_hasher_init
        break
    except (OSError, AttributeError):
        _llvm = None


def blake3(data: bytes) -> bytes:
    st = ctypes.create_string_buffer(4096)
    _llvm.llvm_blake3_hasher_init(st)
    _llvm.llvm_blake3_hasher_update(st, data, ctypes.c_size_t(len(data)))
    out = ctypes.create_string_buffer(32)
    _llvm.llvm_blake3_hasher_finalize(st, out, ctypes.c_size_t(32))
    return out.raw


def merkle_root(leaves):
    lvl = [bytes(l) for l in leaves]
    while len(lvl) > 1:
        lvl = [blake3(lvl[i] + lvl[i + 1]) for i in range(0, len(lvl), 2)]
    return lvl[0]


# ---- independent Rescue (dense MDS, big ints) -------------------------------------------------------
def rescue_tables():
    src = (REF / "crypto/src/hash/rescue/rp64_256/mod.rs").read_text()
    def table(name):
        m = re.search(r"const %s: \[\[BaseElement; STATE_WIDTH\]; \w+\] = \[(.*?)\n\];" % name, src, re.S)
        v = new_vals(m.group(1))
        return [v[i:i + 12] for i in range(0, len(v), 12)]
    return table("MDS"), table("ARK1"), table("ARK2")


def rescue_perm(state):
    mds, ark1, ark2 = rescue_tables()
    inv_alpha = 10540996611094048183
    s = list(state)
    for r in range(7):
        s = [pow(x, 7, P) for x in s]
        s = [sum(mds[i][j] * s[j] for j in range(12)) % P for i in range(12)]
        s = [(x + k) % P for x, k in zip(s, ark1[r])]
        s = [pow(x, inv_alpha, P) for x in s]
        s = [sum(mds[i][j] * s[j] for j in range(12)) % P for i in range(12)]
        s = [(x + k) % P for x, k in zip(s, ark2[r])]
    return s


def rescue_hash_elements(elems):
    st = [0] * 12
    st[0] = len(elems) % P
    i = 0
    for e in elems:
        st[4 + i] = (st[4 + i] + e) % P
        i += 1
        if i % 8 == 0:
            st = rescue_perm(st)
            i = 0
    if i > 0:
        st = rescue_perm(st)
    return st[4:8]


# ---- independent NTT / LDE by definition ------------------------------------------------------------
ROOT_2_32 = 7277203076849721926


def root(log_n):
    return pow(ROOT_2_32, 1 << (32 - log_n), P)


def dft(x):
    n = len(x)
    w = root(n.bit_length() - 1)
    return [sum(x[j] * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]


def lde(p, blowup, offset):
    n = len(p)
    N = n * blowup
    g = root(N.bit_length() - 1)
    return [sum(p[j] * pow(offset * pow(g, k, P) % P, j, P) for j in range(n)) % P for k in range(N)]


def main():
    vec = {"reference": {}, "derived": {}}
    msrc = (REF / "crypto/src/merkle/tests.rs").read_text()
    vec["reference"]["LEAVES4"] = parse_leaves(msrc, "LEAVES4")      # crypto/src/merkle/tests.rs:14-31
    vec["reference"]["LEAVES8"] = parse_leaves(msrc, "LEAVES8")      # :33-66

    rsrc = (REF / "crypto/src/hash/rescue/rp64_256/tests.rs").read_text()
    m = re.search(r"fn apply_permutation\(\).*?let expected = vec!\[(.*?)\];", rsrc, re.S)
    vec["reference"]["rp64_256_permutation_in"] = list(range(12))    # tests.rs:70-84
    vec["reference"]["rp64_256_permutation_out"] = new_vals(m.group(1))  # tests.rs:89-102
    assert len(vec["reference"]["rp64_256_permutation_out"]) == 12

    fsrc = (REF / "math/src/field/f64/tests.rs").read_text()
    m = re.search(r"fn quad_mul\(\)(.*?)\n}\n", fsrc, re.S)
    M = P
    vec["reference"]["f64_quad_mul"] = [                              # f64/tests.rs:228-246
        {"a": [3, 1], "b": [4, 2], "out": [8, 12]},
        {"a": [3, M - 1], "b": [M - 3, 5], "out": [1, 13]},
        {"a": [3, M - 1], "b": [10, M - 2], "out": [26, 18446744069414584307]},
    ]
    assert "18446744069414584307" in m.group(1)
    m = re.search(r"fn cube_mul\(\)(.*?)\n}\n", fsrc, re.S)
    v = new_vals(m.group(1))
    assert len(v) == 27
    vec["reference"]["f64_cube_mul"] = [                              # f64/tests.rs:294-346
        {"a": v[i:i + 3], "b": v[i + 3:i + 6], "out": v[i + 6:i + 9]} for i in range(0, 27, 9)]
    vec["reference"]["f64_edge"] = {                                   # f64/tests.rs:64-73 (mul edge cases)
        "m_minus_1_squared": 1, "m_minus_1_times_2": M - 2, "half_times_2": 1}
    # f62 cubic extension products (math/src/field/f62/tests.rs:128-187): within bounds and two cases "with overflow"
    f62src = (REF / "math/src/field/f62/tests.rs").read_text()
    m = re.search(r"fn cube_mul\(\)(.*?)\n}\n", f62src, re.S)
    v = new_vals(m.group(1))
    assert len(v) == 27 and v[:3] == [15, 22, 8] and v[6] == 4611624995532046021
    vec["reference"]["f62_cube_mul"] = [
        {"a": v[i:i + 3], "b": v[i + 3:i + 6], "out": v[i + 6:i + 9]} for i in range(0, 27, 9)]
    # trace-LDE fixture: prover/src/tests/mod.rs:19-31 build_fib_trace(16) and
    # prover/src/trace/trace_lde/default/tests.rs:22-106 expected polynomial evaluations
    vec["reference"]["fib_trace_col0"] = [1, 2, 5, 13, 34, 89, 233, 610]
    vec["reference"]["fib_trace_col1"] = [1, 3, 8, 21, 55, 144, 377, 987]
    tsrc = (REF / "prover/src/trace/trace_lde/default/tests.rs").read_text()
    assert "1u32, 2, 5, 13, 34, 89, 233, 610" in tsrc and "1u32, 3, 8, 21, 55, 144, 377, 987" in tsrc

    d = vec["derived"]
    # public BLAKE3 vectors (checked against the bundled upstream implementation)
    d["blake3"] = [{"in_hex": b.hex(), "out_hex": blake3(b).hex()} for b in
                   [b"", b"abc", b"\x00", bytes(i % 251 for i in range(1025)),
                    bytes(i % 251 for i in range(64)), bytes(i % 251 for i in range(65)),
                    bytes(i % 251 for i in range(1024)), bytes(i % 251 for i in range(2048)),
                    bytes(i % 251 for i in range(2049)), bytes(i % 251 for i in range(3073)),
                    bytes(i % 251 for i in range(4096)), bytes(i % 251 for i in range(7169))]]
    assert d["blake3"][0]["out_hex"] == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert d["blake3"][1]["out_hex"] == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
    d["blake3_root_LEAVES4"] = merkle_root(vec["reference"]["LEAVES4"]).hex()
    d["blake3_root_LEAVES8"] = merkle_root(vec["reference"]["LEAVES8"]).hex()
    d["blake3_f64_hash_elements_1_2"] = blake3((1).to_bytes(8, "little") + (2).to_bytes(8, "little")).hex()
    assert rescue_perm(list(range(12))) == vec["reference"]["rp64_256_permutation_out"]
    d["rp64_merge_zero"] = rescue_perm([8] + [0] * 11)[4:8]
    d["rp64_hash_elements_1_2_3_4"] = rescue_hash_elements([1, 2, 3, 4])
    d["rp64_hash_elements_0_to_18"] = rescue_hash_elements(list(range(19)))
    d["f64_ntt8_1_to_8"] = dft(list(range(1, 9)))
    d["f64_lde_1234_b2_o7"] = lde([1, 2, 3, 4], 2, 7)
    assert d["f64_ntt8_1_to_8"][0] == 36
    assert d["f64_lde_1234_b2_o7"][0] == 1534
    x16 = [(i * i * 0x9E3779B97F4A7C15 + 12345) % P for i in range(16)]
    d["f64_ntt16_in"] = x16
    d["f64_ntt16_out"] = dft(x16)
    d["f64_lde16_b8_o7"] = lde(x16, 8, 7)
    OUT.write_text(json.dumps(vec, indent=1))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
