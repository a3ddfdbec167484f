#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/profile_round.sh r06
# Produces under gpurun_out/profiles/<round>/ (copy what is to be judged into profiles/<round>/):
#   workloads_pmc_summary.json      HBM bytes + SQ counters per call and per kernel of every workload behind bench.py's rooflines
#   bench_pmc_summary.json          counters of the 2^24 transform's kernels from the bench's own --pmc passes
#   bench_driver_protocol.json      the ONE line of `python bench.py --gpus 1 --steps 20 --warmup 5` (what the driver runs); bench_detail.json beside it
#   bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the same bench command; bench_kernel_trace_by_size.csv per grid size
#   roofline_check.txt              every fraction of the line recomputed from a kernel trace of tools/pmc_workloads.py; workloads_kernel_table.json
#   pmc_calibration.json, fri_breakdown.log, fri_kernel_trace_by_size.csv, microbench_*.txt
# Counter passes are separate runs with --pmc only (never combined with a trace option).
set -u
R=${1:-r06}
OUT=gpurun_out/profiles/$R
RAW=/tmp/prof_raw_$R      # raw rocprofv3 output stays on the box (gpurun_out/ is copied back only up to 64 MiB)
mkdir -p "$OUT" "$RAW" profiles/$R
export TMPDIR=/tmp
# 1. workloads: HBM bytes and SQ counters per call
rocprofv3 --pmc FETCH_SIZE -d "$RAW/wl_fetch_size" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/wl_write_size" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$RAW/wl_sq" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
python tools/summarize_workloads_pmc.py "$RAW" $R gpurun_out/pmc_workloads_manifest.json > "$OUT/workloads_pmc_summary.json"
# bench.py quotes these counters as `rooflines.*.traffic`: put this run's summary where it looks (profiles/<round>/ of THIS copy of the repo)
cp "$OUT/workloads_pmc_summary.json" profiles/$R/workloads_pmc_summary.json
# 2. the bench's own counter passes (the 2^24 transform)
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --detail -"
rocprofv3 --pmc FETCH_SIZE -d "$RAW/pmc_fetch" -o $R --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/pmc_write" -o $R --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$RAW/pmc_sq" -o $R --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d "$RAW/pmc_sq2" -o $R --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d "$RAW/pmc_sq3" -o $R --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d "$RAW/pmc_sq4" -o $R --output-format csv -- $B > /dev/null 2>&1
python tools/summarize_pmc.py "$RAW" $R > "$OUT/bench_pmc_summary.json"
cp "$OUT/bench_pmc_summary.json" profiles/$R/bench_pmc_summary.json
# 3. the driver's command, plain: the line and its detail file
python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_driver_protocol.json" 2> "$RAW/bench_n1.err"
# 4. the same command under the kernel trace
rocprofv3 --kernel-trace --stats -d "$RAW/kt" -o $R --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --detail - > "$OUT/bench_under_rocprof.log" 2>&1
cp "$RAW/kt/${R}_kernel_stats.csv" "$OUT/bench_kernel_stats.csv"
python tools/kernel_trace_table.py "$RAW/kt/${R}_kernel_trace.csv" > "$OUT/bench_kernel_trace_by_size.csv"
# 5. every fraction of the line from a kernel trace of the workloads (markers between them), beside the bench's own
WL_SPIN_S=0.6 WL_CALLS=5 rocprofv3 --kernel-trace -d "$RAW/kt_wl" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
python tools/roofline_check.py "$RAW/kt_wl/${R}_kernel_trace.csv" gpurun_out/pmc_workloads_manifest.json "$OUT/bench_detail.json" "$OUT/workloads_pmc_summary.json" \
    "$OUT/workloads_kernel_table.json" > "$OUT/roofline_check.txt" 2>&1
cat "$OUT/roofline_check.txt"
cut -c1-600 "$OUT/bench_driver_protocol.json"

# FETCH_SIZE / WRITE_SIZE calibration in this code's own access width (MI355X_MICROARCH.md: the x2 FETCH_SIZE correction is
# documented for 16-byte-per-lane loads only): a read-modify-write of a known byte count with 8-byte and 16-byte lanes
if [ -x tools/microbench_mall.bin ]; then
rocprofv3 --pmc FETCH_SIZE -d "$RAW/cal_fetch" -o $R --output-format csv -- tools/microbench_mall.bin calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/cal_write" -o $R --output-format csv -- tools/microbench_mall.bin calib > /dev/null 2>&1
python tools/summarize_calibration.py "$RAW" $R > "$OUT/pmc_calibration.json"
cat "$OUT/pmc_calibration.json"
fi

# the FRI commit phase alone, launch by launch, and the small-kernel microbenchmarks
rocprofv3 --kernel-trace -d "$RAW/kt_fri" -o $R --output-format csv -- env PYTHONPATH=$PWD python tools/fri_tail_breakdown.py > "$OUT/fri_breakdown.log" 2>&1
python tools/kernel_trace_table.py "$RAW/kt_fri/${R}_kernel_trace.csv" > "$OUT/fri_kernel_trace_by_size.csv"
for b in microbench_stage microbench_launch; do [ -x tools/$b.bin ] && tools/$b.bin > "$OUT/$b.txt" 2>&1; done
