#!/bin/bash
# Round-2 profile collection on the MI355X box (run from the repo root through gpurun).  Counter passes are separate
# rocprofv3 runs with --kernel-trace only, the program itself after "--" (never a shell or env wrapper).
set -u
OUT=gpurun_out/prof_r02
mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
declare -A SHAPES=( [k7_64_32]="64,32,7,1,4,1088,1920,7" [k7_32_64]="32,64,7,1,4,1088,1920,7" [k3_128_128]="128,128,3,1,1,544,960,5" [cal_k1_64_32]="64,32,1,1,4,1088,1920,2" )
for name in "${!SHAPES[@]}"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${name}_$ctr --output-format csv -- python3 tools/conv_bench.py --reps 3 ${SHAPES[$name]} > $OUT/${name}_$ctr.log 2>&1
  done
done
# per-kernel time of the two named kernels alone (22 launches each)
rocprofv3 --kernel-trace --stats -d $OUT/named --output-format csv -- python3 tools/conv_bench.py --reps 20 64,32,7,1,4,1088,1920,7 32,64,7,1,4,1088,1920,7 128,128,3,1,1,544,960,5 > $OUT/named.log 2>&1
# matrix-pipe utilisation of the fp16-path kernels (BASELINE configs[4])
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/fp16_mfma --output-format csv -- python3 tools/conv_bench.py --precision fp16 --half-io --reps 3 64,32,7,1,4,1088,1920,7 32,64,7,1,4,1088,1920,7 128,128,3,1,4,544,960,5 > $OUT/fp16_mfma.log 2>&1
# matrix-pipe utilisation of the fp32 headline kernels (16-row 7x7 tiles, 2x2-wave 3x3)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/fp32_mfma --output-format csv -- python3 tools/conv_bench.py --reps 3 64,32,7,1,4,1088,1920,7 32,64,7,1,4,1088,1920,7 128,128,3,1,4,544,960,5 > $OUT/fp32_mfma.log 2>&1
unset VC_AUTOTUNE
# whole bench under the kernel trace (1 warm-up + 1 timed step + the instrumented frame)
rocprofv3 --kernel-trace --stats -d $OUT/bench --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
python3 tools/pmc_traffic.py $OUT/traffic.json "conv k7 s1 64->32 @4x1088x1920"=$OUT/k7_64_32_FETCH_SIZE,$OUT/k7_64_32_WRITE_SIZE "conv k7 s1 32->64 @4x1088x1920"=$OUT/k7_32_64_FETCH_SIZE,$OUT/k7_32_64_WRITE_SIZE "conv k3 s1 128->128 @1x544x960"=$OUT/k3_128_128_FETCH_SIZE,$OUT/k3_128_128_WRITE_SIZE "calibration conv k1 s1 64->32 @4x1088x1920"=$OUT/cal_k1_64_32_FETCH_SIZE,$OUT/cal_k1_64_32_WRITE_SIZE > $OUT/traffic.log 2>&1
python3 tools/pmc_summary.py $OUT/fp16_mfma.json fp16=$OUT/fp16_mfma >> $OUT/traffic.log 2>&1
python3 tools/pmc_summary.py $OUT/fp32_mfma.json fp32=$OUT/fp32_mfma >> $OUT/traffic.log 2>&1
# keep only the small summaries (the merge back is capped at 64 MiB)
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
