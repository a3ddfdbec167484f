"""Debug aid for the guard-page session (tools/guard_session.sh): which bytes of which digests differ for Blake3_192 when every
device buffer ends on the last mapped byte of its own range.   WF_DEBUG_GUARD=1 python tools/debug_guard192.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest

conftest._guarded_session()
import numpy as np
import torch

import oracle
import winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd.math import fields

oracle.build()
ctx = winterfell_amd.default_context()
print("guard mode", ctx.lib.wf_debug_guard_mode(), "align", os.environ.get("WF_DEBUG_GUARD_ALIGN", "right"))
rng = np.random.default_rng(1)
for rows, cols in ((129, 17), (64, 8), (1000, 24), (5, 3), (256, 4)):
    data = oracle.f64_from_int(rng.integers(0, fields.M, rows * cols, dtype=np.uint64)).reshape(rows, cols)
    for hname, hid in (("Blake3_256", 0), ("Blake3_192", 5), ("Sha3_256", 2)):
        h = getattr(crypto, hname)
        d = ctx.to_device(data)
        m = prover.RowMatrix(d, cols, cols, 1, ctx, fields.f64)
        out = m.hash_rows(h, prover.PartitionOptions(1, 1))
        got = ctx.to_host(out)
        want = oracle.hash_rows(hid, data, cols, D=1, num_partitions=1, hash_rate=1)
        bad = np.nonzero((got != want).any(axis=1))[0]
        print("%-11s %5d x %2d: %d bad rows%s  in=%#x out=%#x" % (hname, rows, cols, len(bad), (" first %d" % bad[0]) if len(bad) else "", d.data_ptr(), out.data_ptr()))
        if len(bad):
            r = bad[0]
            print("   got ", got[r].tobytes().hex())
            print("   want", want[r].tobytes().hex())
            # is it the hash of some other row / of shifted data?
            for s in (-2, -1, 1, 2):
                sh = np.roll(data.reshape(-1), s).reshape(rows, cols)
                w2 = oracle.hash_rows(hid, sh, cols, D=1, num_partitions=1, hash_rate=1)
                if np.array_equal(w2[r], got[r]):
                    print("   = hash of the data shifted by %d words" % s)
