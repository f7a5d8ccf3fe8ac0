#!/bin/bash
# Debugging aid: SPyNet (tools/spynet_determinism.py) in two processes sharing the GPU: every run of either must equal its first.
mkdir -p gpurun_out/period
(timeout 500 python tools/spynet_determinism.py ${REPS:-40} $SPYARGS > gpurun_out/period/spy_a.log 2>&1 &)
timeout 500 python tools/spynet_determinism.py ${REPS:-40} $SPYARGS > gpurun_out/period/spy_b.log 2>&1
sleep 30
cat gpurun_out/period/spy_a.log gpurun_out/period/spy_b.log | grep -v amdgpu.ids | cut -c1-300 | tail -8
