"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per pass) into per-kernel averages, and derive the
HBM traffic of a 2^24-point f64 transform.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import sys
from collections import defaultdict

raw, rnd = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("%s/pmc_*/%s_counter_collection.csv" % (raw, rnd)):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "ntt_pass" not in name and "hash_rows" not in name and "merkle" not in name and "lde_transpose" not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {k: {c: {"launches": len(v), "avg": sum(v) / len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}


def traffic(sub, last):
    tot = 0.0
    for k, cs in kernels.items():
        is_last = ", true>" in k
        if "ntt_pass<F64" in k and is_last == last and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            tot = (2 * cs["FETCH_SIZE"]["avg"] + cs["WRITE_SIZE"]["avg"]) * 1024
    return tot


p, l = traffic("ntt_pass", False), traffic("ntt_pass", True)
out = {
    "note": "rocprofv3 --pmc, separate passes for FETCH_SIZE / WRITE_SIZE / SQ_*; bench.py --steps 3 --log-n 24; FETCH_SIZE doubled per the gfx950 correction",
    "ntt_2^24_f64": {"ntt_pass_bytes_per_launch": p, "ntt_pass_last_bytes_per_launch": l, "hbm_bytes_per_transform": 2 * p + l,
                     "algorithmic_bytes_per_transform": 268435456},
    "kernels": kernels,
}
print(json.dumps(out, indent=1))
