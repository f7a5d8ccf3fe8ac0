#!/bin/bash
# Where do the waves of the fp16 3x3 kernel and of the 1x1 streaming kernel wait?  SQ counter passes (each its own rocprofv3
# run, kernel trace only) on tools/conv_bench.py; folded by tools/pmc_summary.py.
set -u
OUT=gpurun_out/pmc_wait; mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P -d $OUT/f16_p$i --output-format csv -- python3 tools/conv_bench.py --precision fp16 --half-io --reps 3 128,128,3,1,4,544,960,5 > $OUT/f16_p$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $P -d $OUT/pw_p$i --output-format csv -- python3 tools/conv_bench.py --reps 3 128,128,1,1,1,544,960,6 > $OUT/pw_p$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT/wait_breakdown.json fp16_k3_128_128_x4=$OUT/f16_p1,$OUT/f16_p2,$OUT/f16_p3 fp32_k1_128_128=$OUT/pw_p1,$OUT/pw_p2,$OUT/pw_p3 > $OUT/summary.log 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
tail -5 $OUT/f16_p3.log
