#!/bin/bash
# Run ON THE GPU BOX:  tools/profile_kernel_sq.sh <tag> <kernel-name-substring> -- <command ...>
# SQ counters (two passes) of the kernels whose name contains the substring, averaged per launch.
set -u
TAG=$1; PAT=$2; shift 3
RAW=gpurun_out/prof_raw_$TAG
mkdir -p gpurun_out/r02 "$RAW"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$RAW/a" -o $TAG --output-format csv -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d "$RAW/b" -o $TAG --output-format csv -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$RAW/kt" -o $TAG --output-format csv -- "$@" > /dev/null 2>&1
python - "$RAW" "$TAG" "$PAT" <<'PY'
import csv, glob, sys
from collections import defaultdict
raw, tag, pat = sys.argv[1:4]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("%s/[ab]/%s_counter_collection.csv" % (raw, tag)):
    for row in csv.DictReader(open(path)):
        if pat in row["Kernel_Name"]:
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k[:100])
    for c, v in sorted(cs.items()):
        print("   %-24s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
for path in glob.glob("%s/kt/%s_kernel_stats.csv" % (raw, tag)):
    for row in csv.DictReader(open(path)):
        if pat in row["Name"]:
            print("   trace: calls %s avg %.1f us" % (row["Calls"], float(row["AverageNs"]) / 1e3))
PY
