#!/bin/bash
# HBM traffic of the fp32 headline kernels on this round's sources (round 4) (FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes) ->
# gpurun_out/prof_r04/traffic.json (copied to profiles/r04/traffic.json; bench.py quotes it as `roofline.traffic`).
set -u
OUT=gpurun_out/prof_r04
mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# (cfg 8 = the exact fp32 LDS-DMA instances the tuner settles on this round; cfg 7 / 5 = the classic instances next to them)
declare -A SHAPES=( [k7_64_32]="64,32,7,1,4,1088,1920,8" [k7_32_64]="32,64,7,1,4,1088,1920,8" [k3_128_128]="128,128,3,1,1,544,960,8" [k7_64_32_classic]="64,32,7,1,4,1088,1920,7" [k7_32_64_classic]="32,64,7,1,4,1088,1920,7" [k3_128_128_classic]="128,128,3,1,1,544,960,5" [cal_k1_64_32]="64,32,1,1,4,1088,1920,2" )
for name in "${!SHAPES[@]}"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${name}_$ctr --output-format csv -- python3 tools/conv_bench.py --reps 3 ${SHAPES[$name]} > $OUT/${name}_$ctr.log 2>&1
  done
done
python3 tools/pmc_traffic.py $OUT/traffic.json "conv k7 s1 64->32 @4x1088x1920"=$OUT/k7_64_32_FETCH_SIZE,$OUT/k7_64_32_WRITE_SIZE "conv k7 s1 32->64 @4x1088x1920"=$OUT/k7_32_64_FETCH_SIZE,$OUT/k7_32_64_WRITE_SIZE "conv k3 s1 128->128 @1x544x960"=$OUT/k3_128_128_FETCH_SIZE,$OUT/k3_128_128_WRITE_SIZE "classic instance: conv k7 s1 64->32 @4x1088x1920"=$OUT/k7_64_32_classic_FETCH_SIZE,$OUT/k7_64_32_classic_WRITE_SIZE "classic instance: conv k7 s1 32->64 @4x1088x1920"=$OUT/k7_32_64_classic_FETCH_SIZE,$OUT/k7_32_64_classic_WRITE_SIZE "classic instance: conv k3 s1 128->128 @1x544x960"=$OUT/k3_128_128_classic_FETCH_SIZE,$OUT/k3_128_128_classic_WRITE_SIZE "calibration conv k1 s1 64->32 @4x1088x1920"=$OUT/cal_k1_64_32_FETCH_SIZE,$OUT/cal_k1_64_32_WRITE_SIZE > $OUT/traffic.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
cat $OUT/traffic.json | head -50
