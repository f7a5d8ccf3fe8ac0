#!/bin/bash
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
