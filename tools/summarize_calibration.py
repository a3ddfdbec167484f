"""FETCH_SIZE / WRITE_SIZE (KiB, rocprofv3 --pmc) of tools/microbench_mall.bin's read-modify-write kernels against the bytes they
are known to move (1 GiB read + 1 GiB written per launch, ping-pong between two buffers): the factor to apply to the counters for
8-byte-per-lane and 16-byte-per-lane accesses."""
import csv
import glob
import json
import sys
from collections import defaultdict

raw, rnd = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("%s/cal_*/%s_counter_collection.csv" % (raw, rnd)):
    for row in csv.DictReader(open(path)):
        if "rmw<" in row["Kernel_Name"]:
            width = "16B_per_lane" if "uint4" in row["Kernel_Name"] or "HIP_vector_type<unsigned int, 4" in row["Kernel_Name"] else "8B_per_lane"
            acc[width][row["Counter_Name"]].append(float(row["Counter_Value"]))
known = 1024.0 * 1024 * 1024
out = {"known_bytes_read_per_launch": known, "known_bytes_written_per_launch": known}
for w, cs in acc.items():
    out[w] = {c: {"avg_KiB": sum(v) / len(v), "launches": len(v), "known_over_counted": known / (1024.0 * sum(v) / len(v))} for c, v in cs.items()}
print(json.dumps(out, indent=1))
