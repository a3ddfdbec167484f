set -u
export TMPDIR=/tmp
R=r06; OUT=gpurun_out/profiles/$R; RAW=/tmp/raw9; mkdir -p $OUT $RAW
python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_driver_protocol.json" 2> $RAW/bench.err
WL_SPIN_S=0.6 WL_CALLS=5 rocprofv3 --kernel-trace -d "$RAW/kt_wl" -o $R --output-format csv -- python tools/pmc_workloads.py > /dev/null 2>&1
python tools/roofline_check.py "$RAW/kt_wl/${R}_kernel_trace.csv" gpurun_out/pmc_workloads_manifest.json "$OUT/bench_detail.json" profiles/$R/workloads_pmc_summary.json "$OUT/workloads_kernel_table.json" > "$OUT/roofline_check.txt" 2>&1
cat "$OUT/roofline_check.txt"
timeout 900 python tools/plan_sweep.py $OUT/plan_sweep.csv 5 24 > gpurun_out/plan_sweep.log 2>&1; echo "sweep rc=$?"; tail -2 gpurun_out/plan_sweep.log
