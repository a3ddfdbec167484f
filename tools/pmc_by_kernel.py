"""Aggregate rocprofv3 --pmc counter_collection.csv files by kernel: python tools/pmc_by_kernel.py <dir> [...]  -> per kernel:
launches and the SUM of every counter over its dispatches (FETCH_SIZE / WRITE_SIZE in KiB as the tool reports them)."""
import csv
import glob
import json
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[(name, row["Counter_Name"])].add((path, row["Dispatch_Id"]))
out = {}
for name, c in agg.items():
    out[name] = {k: v for k, v in c.items()}
    out[name]["launches"] = max(len(launches[(name, k)]) for k in c)
print(json.dumps(out, indent=1, sort_keys=True))
