#!/bin/bash
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
    grep -h "wrong pixels by" gpurun_out/li/v${v}_r${r}_a.log gpurun_out/li/v${v}_r${r}_b.log | cut -c1-300 | head -12
  done
done
ls gpurun_out/li/*.pt 2>/dev/null | grep -v "dump_v${KEEP_DUMP:-1}.pt" | xargs -r rm -f
