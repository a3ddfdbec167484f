"""Summarise the rocprofv3 --pmc runs of tools/pmc_workloads.py: HBM bytes per call of every workload = (2 x FETCH_SIZE +
WRITE_SIZE) KiB summed over the dispatches between the workload's two markers, divided by its number of calls.  FETCH_SIZE is
doubled per the gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section; calibrated for this code's 8- and 16-byte
lanes in pmc_calibration.json); the counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass).
usage: summarize_workloads_pmc.py <raw dir> <round> <manifest.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict

raw, rnd, manifest_path = sys.argv[1], sys.argv[2], sys.argv[3]
manifest = json.load(open(manifest_path))


def per_workload(counter):
    paths = glob.glob("%s/wl_%s/**/%s_counter_collection.csv" % (raw, counter.lower(), rnd), recursive=True)
    if not paths:
        return None
    rows = []
    with open(paths[0]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    out, seg, cur, per_kernel = [], -1, None, None
    marks = [i for i, r in enumerate(rows) if "twiddles_kernel" in r[1]]
    assert len(marks) == 2 * len(manifest), "expected %d markers, found %d" % (2 * len(manifest), len(marks))
    for w, item in enumerate(manifest):
        a, b = marks[2 * w], marks[2 * w + 1]
        total, kern = 0.0, defaultdict(lambda: [0, 0.0])
        for _, name, v in rows[a + 1:b]:
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            total += v
            kern[short][0] += 1
            kern[short][1] += v
        out.append((item["name"], item["calls"], total, {k: {"launches_per_call": c / item["calls"], "kib_per_call": v / item["calls"]} for k, (c, v) in kern.items()}))
    return out


def per_workload_sq():
    """SQ counters of the `wl_sq` pass (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES), summed over the dispatches
    of a call: bench.py prices a workload's VALU issue floor with them (SQ_ACTIVE_INST_VALU x 4 cycles over 1024 SIMDs)."""
    paths = glob.glob("%s/wl_sq/**/%s_counter_collection.csv" % (raw, rnd), recursive=True)
    if not paths:
        return None
    rows = defaultdict(dict)
    with open(paths[0]) as f:
        for row in csv.DictReader(f):
            d = rows[int(row["Dispatch_Id"])]
            d["name"] = row["Kernel_Name"]
            d[row["Counter_Name"]] = float(row["Counter_Value"])
    order = sorted(rows)
    marks = [i for i, k in enumerate(order) if "twiddles_kernel" in rows[k]["name"]]
    assert len(marks) == 2 * len(manifest), "expected %d markers, found %d" % (2 * len(manifest), len(marks))
    out = []
    for w, item in enumerate(manifest):
        tot, kern = defaultdict(float), defaultdict(lambda: defaultdict(float))
        for k in order[marks[2 * w] + 1:marks[2 * w + 1]]:
            short = rows[k]["name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            for c, v in rows[k].items():
                if c != "name":
                    tot[c] += v / item["calls"]
                    kern[short][c] += v / item["calls"]
        out.append((dict(tot), {k: dict(v) for k, v in kern.items()}))
    return out


fetch, write, sq = per_workload("FETCH_SIZE"), per_workload("WRITE_SIZE"), per_workload_sq()
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of tools/pmc_workloads.py; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
               "over all dispatches of a call (torch's own copy / fill kernels between library calls included)", "workloads": {}}
for i, item in enumerate(manifest):
    f_kib = fetch[i][2] / item["calls"] if fetch else None
    w_kib = write[i][2] / item["calls"] if write else None
    entry = {"calls": item["calls"], "fetch_size_kib_per_call": f_kib, "write_size_kib_per_call": w_kib}
    if f_kib is not None and w_kib is not None:
        entry["hbm_read_bytes_per_call"] = 2 * f_kib * 1024
        entry["hbm_write_bytes_per_call"] = w_kib * 1024
        entry["hbm_bytes_per_call"] = (2 * f_kib + w_kib) * 1024
    kern = {}
    for src, key in ((fetch, "fetch_kib_per_call"), (write, "write_kib_per_call")):
        if src:
            for k, v in src[i][3].items():
                kern.setdefault(k, {"launches_per_call": v["launches_per_call"]})[key] = v["kib_per_call"]
    if sq:
        entry["sq"] = sq[i][0]
        for k, v in sq[i][1].items():
            kern.setdefault(k, {})["sq"] = v
    entry["kernels"] = kern
    res["workloads"][item["name"]] = entry
print(json.dumps(res, indent=1))
