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


import re


def traffic(last, twtab):
    """(2 x FETCH_SIZE + WRITE_SIZE) bytes per launch of the radix-256 f64 pass kernel with these template flags
    (ntt_pass<F64, LOG_A, LOG_B, LAST, TWTAB, PF = false, RH = false>), and how many launches the counter pass saw."""
    tot, launches = 0.0, 0
    for k, cs in kernels.items():
        m = re.search(r"ntt_pass<F64, 4, 4, (true|false), (true|false), false(?:, false)*(?:, 0)?>", k)      # PF, RH (round 3) are false and the tile mode (round 6) is 0 for a plain transform
        if m and (m.group(1) == "true") == last and (m.group(2) == "true") == twtab and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            tot = (2 * cs["FETCH_SIZE"]["avg"] + cs["WRITE_SIZE"]["avg"]) * 1024
            launches = cs["FETCH_SIZE"]["launches"]
    return tot, launches


# a 2^24-point transform = two non-last passes (progression twiddles; a table pass only in -DNTT_F64_TW_TABLES builds) + the last
# pass: the launch counts of the counter run say how many of each there are per transform
(p0, n0), (p1, n1), (l, nl) = traffic(False, False), traffic(False, True), traffic(True, False)
per_transform = (p0 * n0 + p1 * n1 + l * nl) / nl if nl else 0.0
out = {
    "note": "rocprofv3 --pmc, separate passes for FETCH_SIZE / WRITE_SIZE / SQ_*; bench.py --steps 3 --log-n 24; FETCH_SIZE doubled per the gfx950 correction",
    "ntt_2^24_f64": {"ntt_pass_progression_bytes_per_launch": p0, "ntt_pass_table_bytes_per_launch": p1, "ntt_pass_last_bytes_per_launch": l,
                     "launches_per_transform": {"progression": n0 / nl if nl else 0, "table": n1 / nl if nl else 0, "last": 1},
                     "hbm_bytes_per_transform": per_transform,
                     "algorithmic_bytes_per_transform": 268435456},
    "kernels": kernels,
}
print(json.dumps(out, indent=1))
