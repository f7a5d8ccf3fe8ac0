#!/bin/bash
# Round-3 profile collection on the MI355X box (run from the repo root through gpurun): the fp16-path LDS-DMA convolution
# pipeline (csrc/conv_dma.h) next to the classic fp16 instances.  Counter passes are separate rocprofv3 runs with
# --kernel-trace only, the program itself after "--" (never a shell or env wrapper).
set -u
OUT=gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SHAPES="128,128,3,4,544,960,5 64,32,7,4,1088,1920,7 32,64,7,4,1088,1920,1 128,128,3,1,544,960,5"
# per-kernel time (20 launches per timing round, 2 rounds, classic and DMA interleaved)
rocprofv3 --kernel-trace --stats -d $OUT/named --output-format csv -- python3 tools/dma_check.py --reps 20 --rounds 2 $SHAPES > $OUT/named.log 2>&1
# counter passes, ONE shape per run (the persistent kernel has the same grid for every shape: they must not share a pass)
declare -A SH=( [k3_128_128_x4]="128,128,3,4,544,960,5" [k7_64_32_x4]="64,32,7,4,1088,1920,7" [k7_32_64_x4]="32,64,7,4,1088,1920,1" [k3_128_128_x1]="128,128,3,1,544,960,5" )
: > $OUT/summary.log
for name in "${!SH[@]}"; do
  # matrix-pipe utilisation
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/mfma_$name --output-format csv -- python3 tools/dma_check.py --reps 3 --rounds 1 ${SH[$name]} > $OUT/mfma_$name.log 2>&1
  # where the waves wait
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/wait_$name --output-format csv -- python3 tools/dma_check.py --reps 3 --rounds 1 ${SH[$name]} > $OUT/wait_$name.log 2>&1
  # HBM traffic (separate passes)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${ctr}_$name --output-format csv -- python3 tools/dma_check.py --reps 3 --rounds 1 ${SH[$name]} > $OUT/${ctr}_$name.log 2>&1
  done
  python3 tools/pmc_summary.py $OUT/pmc_$name.json mfma=$OUT/mfma_$name wait=$OUT/wait_$name fetch=$OUT/FETCH_SIZE_$name write=$OUT/WRITE_SIZE_$name >> $OUT/summary.log 2>&1
done
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
ls $OUT $OUT/named 2>/dev/null | head -40
