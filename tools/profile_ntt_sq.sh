#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  tools/profile_ntt_sq.sh <tag>   -> gpurun_out/r02/<tag>_sq.json (per-kernel SQ counter averages of the 2^24 f64 NTT)
set -u
TAG=${1:-sq}
RAW=gpurun_out/prof_raw_$TAG
mkdir -p gpurun_out/r02 "$RAW"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$RAW/pmc_sq" -o $TAG --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d "$RAW/pmc_sq2" -o $TAG --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python tools/summarize_pmc.py "$RAW" $TAG > gpurun_out/r02/${TAG}_sq.json
python - <<PY
import json
d=json.load(open("gpurun_out/r02/${TAG}_sq.json"))
for k,v in d["kernels"].items():
    print(k[:60])
    for c,x in v.items(): print("   %-24s %14.0f" % (c, x["avg"]))
PY
