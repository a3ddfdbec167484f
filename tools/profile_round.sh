#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/profile_round.sh r03
# Produces under gpurun_out/profiles/<round>/: the plain bench line, the rocprofv3 --kernel-trace --stats summary of the
# same bench command, and the PMC summary (separate --pmc passes, no tracing options combined with them).
set -u
R=${1:-r05}
OUT=gpurun_out/profiles/$R
RAW=/tmp/prof_raw_$R      # raw rocprofv3 output stays on the box (gpurun_out/ is copied back only up to 64 MiB)
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
# HBM bytes per call of the workloads behind bench.py's `rooflines` (LDE + commit shapes, Merkle, FRI), counters in separate runs
rocprofv3 --pmc FETCH_SIZE -d "$RAW/wl_fetch_size" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/wl_write_size" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$RAW/wl_sq" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
python tools/summarize_workloads_pmc.py "$RAW" $R gpurun_out/pmc_workloads_manifest.json > "$OUT/workloads_pmc_summary.json"
# bench.py quotes these counters as `rooflines.*.traffic`: put this run's summary where it looks (profiles/<round>/ of THIS copy of the repo)
mkdir -p profiles/$R && cp "$OUT/workloads_pmc_summary.json" profiles/$R/workloads_pmc_summary.json
python bench.py > "$OUT/bench_n1.json" 2> "$RAW/bench_n1.err"
rocprofv3 --kernel-trace --stats -d "$RAW/kt" -o $R --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
cp "$RAW/kt/${R}_kernel_stats.csv" "$OUT/bench_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE -d "$RAW/pmc_fetch" -o $R --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/pmc_write" -o $R --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$RAW/pmc_sq" -o $R --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
# where the waves' cycles go: LDS issue stalls, scalar instructions, vector-memory cycles, waves and busy time — three
# more passes, so that one counter this build of rocprofv3 does not know cannot cost the others
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d "$RAW/pmc_sq2" -o $R --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d "$RAW/pmc_sq3" -o $R --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d "$RAW/pmc_sq4" -o $R --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python tools/summarize_pmc.py "$RAW" $R > "$OUT/bench_pmc_summary.json"
# per-size kernel table of the traced bench run (the --stats summary averages a kernel over launches of very different sizes)
python tools/kernel_trace_table.py "$RAW/kt/${R}_kernel_trace.csv" > "$OUT/bench_kernel_trace_by_size.csv"
cut -c1-400 "$OUT/bench_n1.json"

# FETCH_SIZE / WRITE_SIZE calibration in this code's own access width (MI355X_MICROARCH.md: the x2 FETCH_SIZE correction is
# documented for 16-byte-per-lane loads only): a read-modify-write of a known byte count with 8-byte and 16-byte lanes
rocprofv3 --pmc FETCH_SIZE -d "$RAW/cal_fetch" -o $R --output-format csv -- tools/microbench_mall.bin calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/cal_write" -o $R --output-format csv -- tools/microbench_mall.bin calib > /dev/null 2>&1
python tools/summarize_calibration.py "$RAW" $R > "$OUT/pmc_calibration.json"
cat "$OUT/pmc_calibration.json"

# the FRI commit phase alone, launch by launch, and the small-kernel microbenchmarks
rocprofv3 --kernel-trace -d "$RAW/kt_fri" -o $R --output-format csv -- env PYTHONPATH=$PWD python tools/fri_tail_breakdown.py > "$OUT/fri_breakdown.log" 2>&1
python tools/kernel_trace_table.py "$RAW/kt_fri/${R}_kernel_trace.csv" > "$OUT/fri_kernel_trace_by_size.csv"
for b in microbench_stage microbench_launch microbench_field microbench_sqr; do [ -x tools/$b.bin ] && tools/$b.bin > "$OUT/$b.txt" 2>&1; done
