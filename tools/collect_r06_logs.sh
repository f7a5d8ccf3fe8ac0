#!/bin/bash
# Round 6: the printed reports of the parity tests + the memory-bound kernels' micro-benchmark + the suite's durations + the default bench line
O=gpurun_out/logs_r06; mkdir -p $O
python -m pytest tests/test_byte_equality_gpu.py -q -s 2>&1 | grep -v "alone on the oracle\|amdgpu.ids" | cut -c1-1500 > $O/e_byte_equality.log
python -m pytest tests/test_reference_1080p_gpu.py tests/test_refine_gpu.py tests/test_bitstream_gpu.py "tests/test_fullsize_gpu.py::test_flex_1080p_against_oracle" -q -s 2>&1 | grep -v "alone on the oracle\|amdgpu.ids" | cut -c1-1500 > $O/e_reference_1080p_and_flex.log
(echo "== shipped form (grid-stride over pixels)"; python tools/ew_bench.py --reps 30 2>&1 | grep -v amdgpu.ids; echo "== VC_LI_FORM=rows (opt-in)"; VC_LI_FORM=rows python tools/ew_bench.py --reps 30 2>&1 | grep level_input) > $O/g_ew_bench.log
python -m pytest tests/ -q -m gpu --durations=25 2>&1 | tail -40 > $O/f_gpu_suite_durations.log
python bench.py > $O/h_bench_line_final.json 2> $O/h_bench.err
tail -3 $O/e_byte_equality.log | cut -c1-300; grep -c "passed\|failed" $O/e_reference_1080p_and_flex.log; tail -2 $O/f_gpu_suite_durations.log; tail -c 600 $O/h_bench_line_final.json
