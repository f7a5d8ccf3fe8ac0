#!/bin/bash
# run the same python tool in two processes that share the GPU:  bash tools/two_process.sh tools/level_input_repro.py [args]
mkdir -p gpurun_out/period
(timeout 600 python "$@" > gpurun_out/period/tp_a.log 2>&1 &)
timeout 600 python "$@" > gpurun_out/period/tp_b.log 2>&1
sleep 20
cat gpurun_out/period/tp_a.log gpurun_out/period/tp_b.log | grep -v amdgpu.ids | cut -c1-300
