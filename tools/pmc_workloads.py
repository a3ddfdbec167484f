#!/usr/bin/env python3
"""Run ON THE GPU BOX under `rocprofv3 --pmc <counter>` (one counter per run: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2):
the workloads whose HBM rooflines bench.py reports, each as  warm-up call | marker | K measured calls | marker,  so that
tools/summarize_workloads_pmc.py can attribute every dispatch between two markers to its workload and divide by K.
The marker is a launch of `twiddles_kernel` (fft::get_twiddles of a 4-point domain), which none of the workloads uses.
The order and K of the workloads are written to gpurun_out/pmc_workloads_manifest.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import winterfell_amd
from winterfell_amd import crypto, fri as wfri, prover
from winterfell_amd.math import fft, fields

K = int(os.environ.get("WL_CALLS", "2"))
SPIN_S = float(os.environ.get("WL_SPIN_S", "0"))
ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(1)
manifest = []


def marker():
    fft.get_twiddles(4)
    torch.cuda.synchronize()


def measure(name, fn):
    fn()                                   # warm-up: allocations, tables
    torch.cuda.synchronize()
    if SPIN_S > 0:                         # steady clocks before the measured calls, as bench.py's legs have them (kernel-trace runs only:
        import time                        # under --pmc every dispatch is serialised and a spin-up is minutes of counter collection)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < SPIN_S:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
    marker()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    marker()
    manifest.append({"name": name, "calls": K})


def lde_case(key, field, log_rows, cols, parts=1, hasher=None):
    rows = 1 << log_rows
    if field is fields.f64:
        tr = ctx.to_device(rng.integers(0, fields.M, (cols, rows), dtype=np.uint64))
    else:
        tr = torch.randint(0, 1 << 62, (cols, rows * 2), dtype=torch.int64, device=ctx.device)
    cm, dom, po = prover.ColMatrix(tr, field=field), prover.StarkDomain(rows, 8, field=field), prover.PartitionOptions(parts, 1)
    measure("lde_commit_" + key, lambda: prover.build_trace_commitment(hasher or crypto.Blake3_256, cm, dom, po))
    del tr, cm
    torch.cuda.empty_cache()


only = set(a for a in sys.argv[1:])
want = lambda k: not only or k in only
if want("lde_small"):
    lde_case("2^20x4_b8_f64_blake3", fields.f64, 20, 4)
if want("lde_wide"):
    lde_case("2^22x32_b8_f64_blake3", fields.f64, 22, 32)
if want("lde_long"):
    lde_case("2^24x4_b8_f64_blake3", fields.f64, 24, 4)
if want("config3"):
    lde_case("2^22x64_b8_f128_blake3_p8", fields.f128, 22, 64, parts=8)
if want("lde_bench_widths"):
    lde_case("2^19x64_b8_f64_blake3", fields.f64, 19, 64)
    lde_case("2^19x96_b8_f64_blake3", fields.f64, 19, 96)
if want("ntt_fields"):
    for key, field, D in (("2^20_f128", fields.f128, 1), ("2^20_f62", fields.f62, 1), ("2^20_f64_quad", fields.f64, 2), ("2^20_f64_cubic", fields.f64, 3)):
        if field is fields.f64:
            d_ = ctx.to_device(rng.integers(0, fields.M, (1 << 20) * D, dtype=np.uint64))
        elif field is fields.f128:
            d_ = ctx.to_device(rng.integers(0, 1 << 62, (1 << 20) * D * 2, dtype=np.uint64))
        else:
            d_ = ctx.to_device(rng.integers(0, 1 << 61, (1 << 20) * D, dtype=np.uint64))
        measure("ntt_" + key, lambda: fft.evaluate_poly(d_, ext_degree=D, field=field))
        del d_
if want("lde_rescue"):
    lde_case("2^20x4_b8_f64_rp64", fields.f64, 20, 4, hasher=crypto.Rp64_256)
if want("merkle"):
    lv = ctx.to_device(rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8))
    measure("merkle_blake3_2^23_leaves", lambda: crypto.MerkleTree.new(crypto.Blake3_256, lv))
    del lv
if want("fri"):
    ev = ctx.to_device(rng.integers(0, fields.M, (1 << 24) * 2, dtype=np.uint64))

    def fri_run():
        pr = wfri.FriProver(wfri.FriOptions(8, 4, 31), crypto.Blake3_256, ext_degree=2)
        pr.build_layers(wfri.DefaultProverChannel(1 << 24, 32, crypto.Blake3_256, ext_degree=2), ev)

    measure("fri_build_layers_2^24_quad_fold4_blake3", fri_run)
if want("ntt"):
    data = ctx.to_device(rng.integers(0, fields.M, 1 << 24, dtype=np.uint64))
    measure("ntt_2^24_f64_forward", lambda: fft.evaluate_poly(data))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "pmc_workloads_manifest.json"), "w") as f:
    json.dump(manifest, f)
