"""Every HBM-roofline fraction of the bench line, recomputed from profiler files instead of from the library's own HIP events
(VERDICT r5 item 3): tools/pmc_workloads.py under `rocprofv3 --kernel-trace` gives the duration of every dispatch; the dispatches
between a workload's two markers, divided by its calls, are one call; frac = algorithmic bytes / sum of kernel durations / 8 TB/s.
With the bench detail file the same fractions are printed beside the bench's (events on the launch stream) and their ratio.
With the counter summary (tools/summarize_workloads_pmc.py) every kernel of a call also gets its VALU wave-instructions, its issue time
at the slow class's rate and its HBM bytes: the per-kernel table VERDICT r5 item 6 asks for.

  python tools/roofline_check.py <kernel_trace.csv> <manifest.json> [bench_detail.json] [workloads_pmc_summary.json] > roofline_check.txt
Also writes <kernel_trace dir>/../workloads_kernel_table.json when given a 5th argument (output path)."""
import csv
import json
import sys
from collections import defaultdict

HBM = 8e12
trace, manifest = sys.argv[1], json.load(open(sys.argv[2]))
detail = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
pmc = json.load(open(sys.argv[4]))["workloads"] if len(sys.argv) > 4 and sys.argv[4] != "-" else {}
out_json = sys.argv[5] if len(sys.argv) > 5 else None


def alg_bytes(name):
    """trace LDE + commit n c s (2 + b) + 64 b n; transform 2 n e; Merkle 64 N; FRI per layer len e + len/4 e + 64 len/4"""
    import re
    m = re.match(r"lde_commit_2\^(\d+)x(\d+)_b(\d+)_(f64|f128)", name)
    if m:
        L, c, b, f = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
        n, s = 1 << L, 8 if f == "f64" else 16
        return n * c * s * (2 + b) + 64 * b * n
    m = re.match(r"ntt_2\^(\d+)_(f128|f62|f64_quad|f64_cubic|f64)", name)
    if m:
        e = {"f128": 16, "f62": 8, "f64": 8, "f64_quad": 16, "f64_cubic": 24}[m.group(2)]
        return 2 * (1 << int(m.group(1))) * e
    if name.startswith("merkle_blake3_2^23"):
        return 64 * (1 << 23)
    if name.startswith("fri_build_layers_2^24_quad"):
        tot, ln = 0, 1 << 24
        while ln > 256:
            tot += ln * 16 + (ln // 4) * 16 + 64 * (ln // 4)
            ln //= 4
        return tot
    return None


rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Dispatch_Id"]) if "Dispatch_Id" in r else int(r["Start_Timestamp"]), r["Kernel_Name"],
                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])))
rows.sort(key=lambda t: t[3])
marks = [i for i, r in enumerate(rows) if "twiddles_kernel" in r[1]]
assert len(marks) == 2 * len(manifest), "expected %d markers, found %d" % (2 * len(manifest), len(marks))
table = {}
print("# frac = algorithmic bytes / sum of kernel durations (rocprofv3 --kernel-trace, per call) / 8 TB/s; bench = the same from the library's HIP events")
print("%-44s %12s %10s %8s %10s %8s" % ("workload", "kernels us", "alg MB", "frac", "bench frac", "ratio"))
for w, item in enumerate(manifest):
    a, b = marks[2 * w], marks[2 * w + 1]
    kern = defaultdict(lambda: [0, 0.0])
    for _, name, us, _ in rows[a + 1:b]:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if short.startswith(("at::", "__amd_rocclr", "vectorized_elementwise")) or "elementwise_kernel" in short:
            continue                              # torch's own copies / fills between library calls (trace.clone()): not the library's kernels
        kern[short][0] += 1
        kern[short][1] += us
    calls = item["calls"]
    total_us = sum(v[1] for v in kern.values()) / calls
    name = item["name"].replace("_forward", "")
    ab = alg_bytes(name)
    frac = ab / (total_us * 1e-6) / HBM if ab else None
    bfrac = None
    if detail:
        if name == "ntt_2^24_f64":
            bfrac = detail.get("roofline", {}).get("frac")
        else:
            bfrac = detail.get("rooflines", {}).get(name, {}).get("frac")
    print("%-44s %12.1f %10.1f %8s %10s %8s" % (item["name"], total_us, (ab or 0) / 1e6, "%.4f" % frac if frac else "-", "%.4f" % bfrac if bfrac else "-",
                                                 "%.3f" % (frac / bfrac) if frac and bfrac else "-"))
    pk = pmc.get(item["name"], {}).get("kernels", {})
    ktab = {}
    for k, (cnt, us) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
        e = {"launches_per_call": cnt / calls, "us_per_call": us / calls}
        q = pk.get(k, {})
        if "sq" in q and "SQ_INSTS_VALU" in q["sq"]:
            e["valu_wave_insts_per_call"] = q["sq"]["SQ_INSTS_VALU"]
            e["issue_us_at_4p3_clk_2p25GHz"] = q["sq"]["SQ_INSTS_VALU"] * 4.3 / (1024 * 2.25e9) * 1e6
        if "fetch_kib_per_call" in q or "write_kib_per_call" in q:
            e["hbm_bytes_per_call"] = (2 * q.get("fetch_kib_per_call", 0.0) + q.get("write_kib_per_call", 0.0)) * 1024
            e["hbm_us_at_5TBps"] = e["hbm_bytes_per_call"] / 5e12 * 1e6
        ktab[k] = e
    table[item["name"]] = {"kernels_us_per_call": total_us, "algorithmic_bytes": ab, "frac": frac, "bench_frac": bfrac, "kernels": ktab}
if out_json:
    with open(out_json, "w") as f:
        json.dump(table, f, indent=1)
