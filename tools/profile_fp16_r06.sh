#!/bin/bash
# Round 6 ("or close the chapter with counters"): is the operand stream the wall of the fp16-path kernels?
# Two rocprofv3 passes per shape on tools/conv_bench.py (VC_AUTOTUNE=0, program directly after `--`): matrix pipe / clock, and the LDS
# side (instructions, array cycles, bank conflicts, cycles waves wait on LDS).  tools/pmc_summary.py folds them; the derived figures
# (LDS-array busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256 CUs), LDS bytes per MFMA) are printed at the end.
set -u
OUT=gpurun_out/prof_r06
mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d $OUT/f16_${name}_sq --output-format csv -- python3 tools/conv_bench.py --precision fp16 --half-io --reps 5 "$@" > $OUT/f16_${name}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/f16_${name}_lds --output-format csv -- python3 tools/conv_bench.py --precision fp16 --half-io --reps 5 "$@" > $OUT/f16_${name}_lds.log 2>&1
}
run k3_128_128_x4 128,128,3,1,4,544,960,8
run k3_128_128_x1 128,128,3,1,1,544,960,8
run k3_128_128_reshalf --residual-half 128,128,3,1,1,544,960
run k3_128_128_2160 128,128,3,1,1,1088,1920,8
run k3_64_64_2160 64,64,3,1,1,1088,1920,8
run k7_64_32 64,32,7,1,4,1088,1920,8
run k5s2_320_128 320,128,5,2,1,1088,1920
python3 tools/pmc_summary.py $OUT/pmc_fp16.json k3_128_128_x4=$OUT/f16_k3_128_128_x4_sq,$OUT/f16_k3_128_128_x4_lds k3_128_128_x1=$OUT/f16_k3_128_128_x1_sq,$OUT/f16_k3_128_128_x1_lds k3_128_128_half_identity=$OUT/f16_k3_128_128_reshalf_sq,$OUT/f16_k3_128_128_reshalf_lds k3_128_128_2160p=$OUT/f16_k3_128_128_2160_sq,$OUT/f16_k3_128_128_2160_lds k3_64_64_2160p=$OUT/f16_k3_64_64_2160_sq,$OUT/f16_k3_64_64_2160_lds k7_64_32=$OUT/f16_k7_64_32_sq,$OUT/f16_k7_64_32_lds k5s2_320_128=$OUT/f16_k5s2_320_128_sq,$OUT/f16_k5s2_320_128_lds > $OUT/pmc_fp16.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
grep -h "TFLOP" $OUT/f16_*_sq.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/prof_r06/pmc_fp16.json"))["kernels"]
for k, v in d.items():
    cu_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * 256.0            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs
    out = {"mfma_busy": round(v.get("mfma_busy_fraction", 0.0), 3),
           "lds_array_busy": round(v["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 3),           # LDS-array cycles per CU cycle (one array per CU)
           "lds_conflict_share": round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1.0), 3),
           "lds_insts_per_mfma": round(v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"], 3),
           "valu_per_mfma": round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 3),
           "wave_cycles_waiting_on_lds": round(v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"], 3),
           "wave_cycles_waiting_any": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3)}
    print(k, out)
PY
