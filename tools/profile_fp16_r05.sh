#!/bin/bash
# Round 5: fresh counters of the fp16-path kernels this round touched: matrix-pipe busy and clock on the
# LDS-DMA 3x3 128->128 kernel (plain epilogue, and the new half-identity epilogue of the residual blocks), the 7x7 layers, and HBM
# traffic of the 3x3; separate rocprofv3 passes, program directly after `--`.
set -u
OUT=gpurun_out/prof_r05
mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/f16_${name}_sq --output-format csv -- python3 tools/conv_bench.py --precision fp16 --half-io --reps 5 "$@" > $OUT/f16_${name}_sq.log 2>&1; }
run k3_128_128_x4 128,128,3,1,4,544,960,8
run k3_128_128_x1 128,128,3,1,1,544,960,8
run k3_128_128_reshalf --residual-half 128,128,3,1,1,544,960
run k7_64_32 64,32,7,1,4,1088,1920,8
run k7_32_64 32,64,7,1,4,1088,1920,8
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $OUT/f16_k3_reshalf_$ctr --output-format csv -- python3 tools/conv_bench.py --precision fp16 --half-io --residual-half --reps 3 128,128,3,1,1,544,960 > $OUT/f16_k3_reshalf_$ctr.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_fp16.json k3_128_128_x4=$OUT/f16_k3_128_128_x4_sq k3_128_128_x1=$OUT/f16_k3_128_128_x1_sq k3_128_128_half_identity=$OUT/f16_k3_128_128_reshalf_sq k7_64_32=$OUT/f16_k7_64_32_sq k7_32_64=$OUT/f16_k7_32_64_sq > $OUT/pmc_fp16.log 2>&1
python3 tools/pmc_traffic.py $OUT/traffic_fp16.json "fp16 conv k3 s1 128->128 @1x544x960, half tensors + half identity"=$OUT/f16_k3_reshalf_FETCH_SIZE,$OUT/f16_k3_reshalf_WRITE_SIZE > $OUT/traffic_fp16.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
grep -h "TFLOP" $OUT/f16_*_sq.log
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/prof_r05/pmc_fp16.json"))["kernels"]
for k,v in d.items():
    print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="per_kernel"})
PY
