(timeout 800 python tools/forward_determinism.py 150 > gpurun_out/period/fwd_a.log 2>&1 &)
timeout 800 python tools/forward_determinism.py 150 > gpurun_out/period/fwd_b.log 2>&1
sleep 20
grep -h "runs differ\|differs from" gpurun_out/period/fwd_a.log gpurun_out/period/fwd_b.log | cut -c1-300 | head
