#!/bin/bash
# Round 6: counters of the split-operand kernels (csrc/conv_split.h) on the MI355X box.  Separate rocprofv3 passes: SQ / GRBM
# counters (matrix-pipe busy, clock), FETCH_SIZE, WRITE_SIZE; program directly after `--`.
set -u
OUT=gpurun_out/prof_r06
mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
declare -A SHAPES=( [k7_32_64]="32,64,7,1,4,1088,1920" [k7_64_32]="64,32,7,1,4,1088,1920" [k3_128_128]="128,128,3,1,1,544,960" [k3_128_128_x4]="128,128,3,1,4,544,960" [k7_32_16]="32,16,7,1,4,1088,1920" )
# the output format of each layer inside the models: split tensor where the consumer is a split layer (32->16 feeds the native 16->2 head)
declare -A OUTF=( [k7_32_64]="--split-out" [k7_64_32]="--split-out" [k3_128_128]="--split-out" [k3_128_128_x4]="--split-out" [k7_32_16]="" )
for name in "${!SHAPES[@]}"; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d $OUT/${name}_sq --output-format csv -- python3 tools/conv_bench.py --split ${OUTF[$name]} --reps 5 ${SHAPES[$name]} > $OUT/${name}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/${name}_lds --output-format csv -- python3 tools/conv_bench.py --split ${OUTF[$name]} --reps 5 ${SHAPES[$name]} > $OUT/${name}_lds.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${name}_$ctr --output-format csv -- python3 tools/conv_bench.py --split ${OUTF[$name]} --reps 3 ${SHAPES[$name]} > $OUT/${name}_$ctr.log 2>&1
  done
done
python3 tools/pmc_summary.py $OUT/pmc_split.json k7_32_64=$OUT/k7_32_64_sq,$OUT/k7_32_64_lds k7_64_32=$OUT/k7_64_32_sq,$OUT/k7_64_32_lds k3_128_128=$OUT/k3_128_128_sq,$OUT/k3_128_128_lds k3_128_128_x4=$OUT/k3_128_128_x4_sq,$OUT/k3_128_128_x4_lds k7_32_16=$OUT/k7_32_16_sq,$OUT/k7_32_16_lds > $OUT/pmc_split.log 2>&1
python3 tools/pmc_traffic.py $OUT/traffic.json "conv k7 s1 32->64 @4x1088x1920"=$OUT/k7_32_64_FETCH_SIZE,$OUT/k7_32_64_WRITE_SIZE "conv k7 s1 64->32 @4x1088x1920"=$OUT/k7_64_32_FETCH_SIZE,$OUT/k7_64_32_WRITE_SIZE "conv k3 s1 128->128 @1x544x960"=$OUT/k3_128_128_FETCH_SIZE,$OUT/k3_128_128_WRITE_SIZE "conv k7 s1 32->16 @4x1088x1920"=$OUT/k7_32_16_FETCH_SIZE,$OUT/k7_32_16_WRITE_SIZE > $OUT/traffic.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/prof_r06/pmc_split.json"))["kernels"]
for k,v in d.items():
    print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="per_kernel"})
print(open("gpurun_out/prof_r06/traffic.json").read()[:1500])
PY
