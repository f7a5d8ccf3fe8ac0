#!/bin/bash
# Round 6: every measurement of the round behind one entry point (run from the repo root on the MI355X box: `gpurun -- 'bash tools/r06.sh <what>'`).
#   headline             the headline run under rocprofv3 --kernel-trace --stats        -> gpurun_out/prof_r06/a_*
#   split-pmc            SQ / GRBM counters + FETCH_SIZE / WRITE_SIZE of the split kernels -> gpurun_out/prof_r06/pmc_split.json, traffic.json
#   fp16-pmc             fp16-path kernels, matrix pipe + LDS side                        -> gpurun_out/prof_r06/pmc_fp16.json
#   final                bench lines + kernel tables of every configuration               -> gpurun_out/final_r06/
#   logs                 printed reports of the parity tests, ew_bench, suite durations, default bench line -> gpurun_out/logs_r06/
#   all                  headline, split-pmc, final, then the default `python bench.py` line
#   li-diag              two processes x the diagnostic variants of the level-input kernel (make li_diag; VARIANTS / REPS / ROUNDS)
#   li-corun             variant 1 beside a torch tenant / tools/hammer.py / itself / alone (REPS)
#   li-corun2            variant 1 beside a tenant of tiny torch kernels / of this library's short element-wise kernels / itself (REPS)
#   li-mix               variant 1 beside a twin on the shipped form / on variant 1, alternating (REPS)
#   li-vashift           variant 1 beside a twin with a shifted virtual-address layout (REPS)
#   two-process-spynet   two processes, production library, tools/spynet_determinism.py (REPS / ROUNDS)
#   two-process-forward  two processes, production library, tools/forward_determinism.py (150 runs each)
#   stamps               shader-clock stamps per phase segment of the split period kernels (make split_diag)
#   stamps-epilogue      the same with / without a split residual
#   f16-stamps           the same stamps for the fp16 LDS-DMA kernel (make dma_diag)
#   epilogue-ab          headline with another library (OLD_LIB=path) against the shipped one, alternating
# Copy what is to be judged from gpurun_out/ into profiles/r06/ (profiles/README.md lists what went where).
what=${1:-}
case "$what" in
headline)
# Round-6 profile collection on the MI355X box (run from the repo root through gpurun): the HEADLINE run itself under
# rocprofv3 --kernel-trace --stats (the program directly after "--": no env / shell hop), folded per kernel name and per
# (kernel, grid); bench.py names the kernel instance its dominant launch ran (roofline.kernel_symbol).
set -u
OUT=gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/headline --output-format csv -- python3 bench.py --no-cpu-baseline --no-strong-block --skip-extras --steps 3 --warmup 1 > $OUT/a_headline_under_rocprofv3_line.json 2> $OUT/a_headline.err
python3 tools/trace_summary.py $OUT/headline $OUT/a_kernel_trace_by_grid.json
cp $(find $OUT/headline -name "*kernel_stats.csv" | head -1) $OUT/a_rocprofv3_kernel_stats.csv
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
rm -rf $OUT/headline
tail -c 900 $OUT/a_headline_under_rocprofv3_line.json
ls -la $OUT
;;
split-pmc)
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
;;
fp16-pmc)
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
;;
final)
# Round-6 bench lines on the MI355X box (run from the repo root through gpurun); one JSON line + kernel table per configuration.
OUT=gpurun_out/final_r06; mkdir -p $OUT
run() { name=$1; shift; python bench.py "$@" --kernel-table $OUT/kernel_table_$name.json > $OUT/bench_line_$name.json 2> $OUT/$name.err; tail -1 $OUT/bench_line_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), d['unit'], 'ms/step', round(d['ms_per_step'],1), d.get('conv_engine',{}).get('timed_region_tflops'), d.get('hbm_kernels_ms_per_frame'))" || tail -3 $OUT/$name.err; }
run lhbdc_fp32 --no-cpu-baseline
run flex_fp32 --model flex --no-cpu-baseline
run flex_fp32_native --model flex --fp32-mode native --no-cpu-baseline
run icip_fp32 --model icip2024 --no-cpu-baseline
run icip_fp32_native --model icip2024 --fp32-mode native --no-cpu-baseline
run lhbdc_fp16 --precision fp16 --no-cpu-baseline
run flex_fp16 --model flex --precision fp16 --no-cpu-baseline
run icip_fp16 --model icip2024 --precision fp16 --no-cpu-baseline
run icip_fp16_2160p --model icip2024 --precision fp16 --resolution 2160p --no-cpu-baseline
run lhbdc_fp16_2160p --precision fp16 --resolution 2160p --no-cpu-baseline
run lhbdc_fp32_2160p --resolution 2160p --no-cpu-baseline
;;
logs)
# Round 6: the printed reports of the parity tests + the memory-bound kernels' micro-benchmark + the suite's durations + the default bench line
O=gpurun_out/logs_r06; mkdir -p $O
python -m pytest tests/test_byte_equality_gpu.py -q -s 2>&1 | grep -v "alone on the oracle\|amdgpu.ids" | cut -c1-1500 > $O/e_byte_equality.log
python -m pytest tests/test_reference_1080p_gpu.py tests/test_refine_gpu.py tests/test_bitstream_gpu.py "tests/test_fullsize_gpu.py::test_flex_1080p_against_oracle" -q -s 2>&1 | grep -v "alone on the oracle\|amdgpu.ids" | cut -c1-1500 > $O/e_reference_1080p_and_flex.log
(echo "== shipped form (grid-stride over pixels)"; python tools/ew_bench.py --reps 30 2>&1 | grep -v amdgpu.ids; echo "== VC_LI_FORM=rows (opt-in)"; VC_LI_FORM=rows python tools/ew_bench.py --reps 30 2>&1 | grep level_input) > $O/g_ew_bench.log
python -m pytest tests/ -q -m gpu --durations=25 2>&1 | tail -40 > $O/f_gpu_suite_durations.log
python bench.py > $O/h_bench_line_final.json 2> $O/h_bench.err
tail -3 $O/e_byte_equality.log | cut -c1-300; grep -c "passed\|failed" $O/e_reference_1080p_and_flex.log; tail -2 $O/f_gpu_suite_durations.log; tail -c 600 $O/h_bench_line_final.json
;;
li-diag)
# Round 6, DESIGN section 5f: the 3-D-grid forms of the SPyNet level-input kernel (make -C video-compression_amd/csrc li_diag), two
# processes sharing the GPU, every run of either must equal its first.  VARIANTS / REPS / ROUNDS from the environment.
rm -rf gpurun_out/li; mkdir -p gpurun_out/li
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_lidiag.so
for v in ${VARIANTS:-1 0 5 3 4 6}; do
  for r in $(seq 1 ${ROUNDS:-1}); do
    export VC_LI_VARIANT=$v
    (VC_LI_DUMP=gpurun_out/li/dump_v${v}.pt timeout 400 python tools/spynet_determinism.py ${REPS:-40} $SPYARGS > gpurun_out/li/v${v}_r${r}_a.log 2>&1 &)
    timeout 400 python tools/spynet_determinism.py ${REPS:-40} $SPYARGS > gpurun_out/li/v${v}_r${r}_b.log 2>&1
    sleep 12
    echo "== variant $v round $r: $(grep -h 'runs differ' gpurun_out/li/v${v}_r${r}_a.log gpurun_out/li/v${v}_r${r}_b.log | tr '\n' ';')"
    grep -h "wrong pixels by\|wave lifetimes" gpurun_out/li/v${v}_r${r}_a.log gpurun_out/li/v${v}_r${r}_b.log | cut -c1-700 | head -16
  done
done
ls gpurun_out/li/*.pt 2>/dev/null | grep -v "dump_v${KEEP_DUMP:-1}.pt" | xargs -r rm -f
;;
two-process-spynet)
# SPyNet (tools/spynet_determinism.py) in two processes sharing the GPU, production library: every run of either must equal its first.
# REPS (default 40) runs per process, ROUNDS (default 1) pairs.
mkdir -p gpurun_out/period
for r in $(seq 1 ${ROUNDS:-1}); do
  (timeout 500 python tools/spynet_determinism.py ${REPS:-40} $SPYARGS > gpurun_out/period/spy_a$r.log 2>&1 &)
  timeout 500 python tools/spynet_determinism.py ${REPS:-40} $SPYARGS > gpurun_out/period/spy_b$r.log 2>&1
  sleep 15
  echo "round $r: $(grep -h 'runs differ' gpurun_out/period/spy_a$r.log gpurun_out/period/spy_b$r.log | tr '\n' ';')"
  grep -h "wrong pixels" gpurun_out/period/spy_a$r.log gpurun_out/period/spy_b$r.log | cut -c1-400 | head -6
done
;;
two-process-forward)
mkdir -p gpurun_out/period
(timeout 800 python tools/forward_determinism.py 150 > gpurun_out/period/fwd_a.log 2>&1 &)
timeout 800 python tools/forward_determinism.py 150 > gpurun_out/period/fwd_b.log 2>&1
sleep 20
grep -h "runs differ\|differs from" gpurun_out/period/fwd_a.log gpurun_out/period/fwd_b.log | cut -c1-300 | head
;;
stamps)
# Round 6: where a phase of the split period kernels goes (make -C video-compression_amd/csrc split_diag): shader-clock totals per segment
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_splitdiag.so VC_AUTOTUNE=0
for shape in 128,128,3,1,4,544,960 128,128,3,1,1,544,960 32,64,7,1,4,1088,1920; do
  echo "== $shape (instrumented, then plain)"
  VC_SPLIT_VARIANT=64 python tools/conv_bench.py --split --split-out --reps 5 $shape 2>&1 | grep -v amdgpu.ids
  python tools/conv_bench.py --split --split-out --reps 5 $shape 2>&1 | grep -v amdgpu.ids
done
;;
stamps-epilogue)
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_splitdiag.so VC_AUTOTUNE=0
for extra in "" "--residual-split"; do
for v in 64 0; do echo "== variant $v $extra"; VC_SPLIT_VARIANT=$v python tools/conv_bench.py --split --split-out $extra --reps 5 128,128,3,1,4,544,960 2>&1 | grep -v amdgpu.ids; done; done
;;
li-corun)
# co-runner experiment: the 3-D-grid form (variant 1) beside (a) a torch-only tenant, (b) the hammer (our persistent conv kernel), (c) itself
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_lidiag.so VC_LI_VARIANT=1
mkdir -p gpurun_out/li
cat > /tmp/torch_tenant.py <<'PY'
import time, torch
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
t = time.time() + 150
while time.time() < t:
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
PY
(timeout 200 python /tmp/torch_tenant.py > /dev/null 2>&1 &)
sleep 8
timeout 300 python tools/spynet_determinism.py ${REPS:-120} > gpurun_out/li/corun_torch.log 2>&1
echo "beside a torch matmul tenant: $(grep -h 'runs differ' gpurun_out/li/corun_torch.log)"
sleep 10
(VC_HIP_LIB= timeout 200 python tools/hammer.py 150 > /dev/null 2>&1 &)
sleep 8
timeout 300 python tools/spynet_determinism.py ${REPS:-120} > gpurun_out/li/corun_hammer.log 2>&1
echo "beside the persistent-convolution tenant: $(grep -h 'runs differ' gpurun_out/li/corun_hammer.log)"
sleep 10
(timeout 300 python tools/spynet_determinism.py ${REPS:-120} > gpurun_out/li/corun_self_a.log 2>&1 &)
timeout 300 python tools/spynet_determinism.py ${REPS:-120} > gpurun_out/li/corun_self_b.log 2>&1
sleep 10
echo "beside itself: $(grep -h 'runs differ' gpurun_out/li/corun_self_a.log gpurun_out/li/corun_self_b.log | tr '\n' ';')"
timeout 300 python tools/spynet_determinism.py ${REPS:-120} > gpurun_out/li/corun_alone.log 2>&1
echo "alone: $(grep -h 'runs differ' gpurun_out/li/corun_alone.log)"
grep -h "wave lifetimes" gpurun_out/li/corun_*.log | cut -c1-500 | head -6
;;
li-corun2)
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_lidiag.so VC_LI_VARIANT=1
mkdir -p gpurun_out/li
cat > /tmp/tiny_tenant.py <<'PY'
import time, torch
xs = [torch.randn(1 << 16, device="cuda") for _ in range(8)]
t = time.time() + 150
while time.time() < t:
    for _ in range(2000):
        for x in xs: x.mul_(1.0001)
    torch.cuda.synchronize()
PY
cat > /tmp/ew_tenant.py <<'PY'
import sys, time, torch
sys.path.insert(0, "video-compression_amd")
from vcamd import hip
dev = torch.device("cuda:0")
a = hip.T.empty(2, 136, 240, 64, dev); a.buf.normal_()
imgs = [hip.T.empty(2, 272, 480, 3, dev) for _ in range(2)]
for t_ in imgs: t_.buf.uniform_()
fl = hip.T.empty(2, 272, 480, 2, dev); fl.buf.normal_()
t = time.time() + 150
while time.time() < t:
    for _ in range(300):
        hip.upsample_bilinear(a, 2)
        hip.warp(hip.WARP_W1, imgs[0], fl)
        hip.avgpool_reflectpad(imgs[1], 2)
    torch.cuda.synchronize()
PY
(timeout 200 python /tmp/tiny_tenant.py > /dev/null 2>&1 &)
sleep 8
timeout 300 python tools/spynet_determinism.py ${REPS:-260} > gpurun_out/li/corun_tiny.log 2>&1
echo "beside a tenant of tiny torch kernels: $(grep -h 'runs differ' gpurun_out/li/corun_tiny.log)"
sleep 12
(VC_LI_VARIANT=0 timeout 200 python /tmp/ew_tenant.py > gpurun_out/li/ew_tenant.log 2>&1 &)
sleep 10
timeout 300 python tools/spynet_determinism.py ${REPS:-260} > gpurun_out/li/corun_ew.log 2>&1
echo "beside a tenant of this library's short element-wise kernels: $(grep -h 'runs differ' gpurun_out/li/corun_ew.log)"; tail -2 gpurun_out/li/ew_tenant.log | cut -c1-200
sleep 12
(timeout 300 python tools/spynet_determinism.py ${REPS:-260} > gpurun_out/li/corun2_self_a.log 2>&1 &)
timeout 300 python tools/spynet_determinism.py ${REPS:-260} > gpurun_out/li/corun2_self_b.log 2>&1
sleep 10
echo "control, beside itself: $(grep -h 'runs differ' gpurun_out/li/corun2_self_a.log gpurun_out/li/corun2_self_b.log | tr '\n' ';')"
;;
li-mix)
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_lidiag.so
mkdir -p gpurun_out/li
for pair in "1 0" "1 1" "1 0" "1 1"; do
  set -- $pair
  (VC_LI_VARIANT=$2 timeout 300 python tools/spynet_determinism.py ${REPS:-260} > gpurun_out/li/mix_b.log 2>&1 &)
  VC_LI_VARIANT=$1 timeout 300 python tools/spynet_determinism.py ${REPS:-260} > gpurun_out/li/mix_a.log 2>&1
  sleep 12
  echo "process A on variant $1 beside a twin on variant $2: A $(grep -h 'runs differ' gpurun_out/li/mix_a.log); twin $(grep -h 'runs differ' gpurun_out/li/mix_b.log)"
done
;;
li-vashift)
# the 3-D-grid form (variant 1), two identical processes -- one of them with a shifted virtual-address layout
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_lidiag.so VC_LI_VARIANT=1
mkdir -p gpurun_out/li
for shift in 3001 0 777 0; do
  (VC_VA_SHIFT_MB=$shift timeout 400 python tools/spynet_determinism.py ${REPS:-200} > gpurun_out/li/va_${shift}_a.log 2>&1 &)
  VC_VA_SHIFT_MB= timeout 400 python tools/spynet_determinism.py ${REPS:-200} > gpurun_out/li/va_${shift}_b.log 2>&1
  sleep 12
  echo "twin shifted by $shift MiB: $(grep -h 'runs differ' gpurun_out/li/va_${shift}_a.log gpurun_out/li/va_${shift}_b.log | tr '\n' ';') $(grep -h 'virtual-address' gpurun_out/li/va_${shift}_a.log | cut -c1-80)"
done
;;
f16-stamps)
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_dmadiag.so VC_AUTOTUNE=0 VC_DMA_VARIANT=64
for shape in 128,128,3,1,4,544,960,8 128,128,3,1,1,1088,1920,8 64,64,3,1,1,1088,1920,8 64,32,7,1,4,1088,1920,8; do
  python tools/conv_bench.py --precision fp16 --half-io --reps 5 $shape 2>&1 | grep -v amdgpu.ids
done
python tools/conv_bench.py --precision fp16 --half-io --residual-half --reps 5 128,128,3,1,1,544,960 2>&1 | grep -v amdgpu.ids
;;
epilogue-ab)
for i in 1 2; do
VC_HIP_LIB=$OLD_LIB python bench.py --no-cpu-baseline --no-strong-block --skip-extras --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('other library ', round(d['value'],2), round(d['roofline']['avg_launch_ms'],3), d['conv_engine']['timed_region_tflops'])"
python bench.py --no-cpu-baseline --no-strong-block --skip-extras --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shipped library', round(d['value'],2), round(d['roofline']['avg_launch_ms'],3), d['conv_engine']['timed_region_tflops'])"
done
;;
all)
bash tools/r06.sh headline > gpurun_out/prof_r06_headline.log 2>&1
bash tools/r06.sh split-pmc > gpurun_out/prof_r06_split.log 2>&1
bash tools/r06.sh final > gpurun_out/final_r06_summary.log 2>&1
python bench.py > gpurun_out/final_r06/h_bench_line_final.json 2> gpurun_out/final_r06/h_bench.err
tail -5 gpurun_out/prof_r06_headline.log | cut -c1-300; tail -12 gpurun_out/prof_r06_split.log | cut -c1-400; cat gpurun_out/final_r06_summary.log; tail -c 1500 gpurun_out/final_r06/h_bench_line_final.json
;;
*) echo "usage: bash tools/r06.sh headline|split-pmc|fp16-pmc|final|logs|all|li-diag|li-corun|li-corun2|li-mix|li-vashift|two-process-spynet|two-process-forward|stamps|stamps-epilogue|f16-stamps|epilogue-ab"; exit 2;;
esac
