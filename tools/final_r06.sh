#!/bin/bash
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
