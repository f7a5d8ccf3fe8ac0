#!/bin/bash
# Round-3 profile of the streaming 1x1 kernel (VC_CFG_PWS, csrc/conv_pws.hip) on the MI355X box (run from the repo root through
# gpurun).  Counter passes are separate rocprofv3 runs with --kernel-trace only, the program itself after "--"; one tensor-type
# mode per pass (the persistent kernel has one grid size, passes must not mix shapes).
set -u
OUT=gpurun_out/prof_pws_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SHAPE=128,128,1,544,960
# per-kernel time: old streaming kernel (cfg 6) and the new one (cfg 9) interleaved, 20 launches per round
rocprofv3 --kernel-trace --stats -d $OUT/named --output-format csv -- python3 tools/pw_check.py --cfgs 6,9 --reps 20 --rounds 2 --modes 'f32,f32+res,h->h,h->f+res' $SHAPE > $OUT/named.log 2>&1
: > $OUT/summary.log
for mode in f32 f32+res 'h->h' 'h->f+res'; do
  tag=$(echo $mode | tr -d '>+-' )
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/mfma_$tag --output-format csv -- python3 tools/pw_check.py --cfgs 9 --reps 3 --rounds 1 --modes "$mode" $SHAPE > $OUT/mfma_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/wait_$tag --output-format csv -- python3 tools/pw_check.py --cfgs 9 --reps 3 --rounds 1 --modes "$mode" $SHAPE > $OUT/wait_$tag.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${ctr}_$tag --output-format csv -- python3 tools/pw_check.py --cfgs 9 --reps 3 --rounds 1 --modes "$mode" $SHAPE > $OUT/${ctr}_$tag.log 2>&1
  done
  python3 tools/pmc_summary.py $OUT/pmc_pws_$tag.json mfma=$OUT/mfma_$tag wait=$OUT/wait_$tag fetch=$OUT/FETCH_SIZE_$tag write=$OUT/WRITE_SIZE_$tag >> $OUT/summary.log 2>&1
done
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
ls $OUT $OUT/named 2>/dev/null | head -40
