#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE per kernel of one trace LDE + commit shape, default library settings against the
# round-4 tile order / row hash (separate --pmc passes, no tracing options).  usage: tools/pmc_lde_quick.sh [log_n=22] [cols=32]
L=${1:-22}; C=${2:-32}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lde_quick; mkdir -p $OUT
run() {  # name, counter, env...
  name=$1; ctr=$2; shift 2
  env "$@" timeout 240 rocprofv3 --pmc $ctr -d $OUT/raw_${name}_$ctr -o x --output-format csv -- python tools/wl_lde.py $L $C 1 > $OUT/${name}_$ctr.log 2>&1 || tail -2 $OUT/${name}_$ctr.log
}
run default FETCH_SIZE A=1
run default WRITE_SIZE A=1
run r04order FETCH_SIZE WF_NTT_COSET_ORDER=0 WF_ROWS_HASH_WIDE=0
python tools/pmc_by_kernel.py $OUT/raw_default_FETCH_SIZE $OUT/raw_default_WRITE_SIZE > $OUT/default_2p${L}x${C}.json
python tools/pmc_by_kernel.py $OUT/raw_r04order_FETCH_SIZE > $OUT/r04order_2p${L}x${C}.json
rm -rf $OUT/raw_*
