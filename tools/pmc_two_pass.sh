#!/bin/bash
# Run ON THE GPU BOX: counters of the 2^22 x 32 f64 LDE + commit under both plans (separate --pmc passes, no tracing options).
# usage: tools/pmc_two_pass.sh [log_n=22] [cols=32]       output: gpurun_out/pmc_two_pass/{three,two}_pass.json
L=${1:-22}; C=${2:-32}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_two_pass; RAW=/tmp/pmc_two_pass_raw
mkdir -p $OUT $RAW
for plan in 0 1; do
  name=$([ $plan = 0 ] && echo three_pass || echo two_pass)
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    WF_NTT_BIG=$plan rocprofv3 --pmc $set -d $RAW/$name/$tag -o x --output-format csv -- python tools/wl_lde.py $L $C 2 > $RAW/$name.$tag.log 2>&1 || tail -3 $RAW/$name.$tag.log
  done
  python tools/pmc_by_kernel.py $RAW/$name > $OUT/${name}_2p${L}x${C}.json
done
