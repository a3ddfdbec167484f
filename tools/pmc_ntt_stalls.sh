#!/bin/bash
# Run ON THE GPU BOX: where the cycles of the 2^24-point f64 transform's passes go — SQ wait / active / instruction
# counters in separate --pmc passes (no tracing options), summed per kernel over 4 transforms.   output: gpurun_out/pmc_ntt_stalls.json
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ntt_stalls; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set -d $OUT/raw_$i -o x --output-format csv -- python tools/wl_ntt.py ${1:-24} 3 > $OUT/pass_$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -2 $OUT/pass_$i.log; }
done
python tools/pmc_by_kernel.py $OUT/raw_* > gpurun_out/pmc_ntt_stalls.json
rm -rf $OUT/raw_*
