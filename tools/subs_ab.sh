#!/bin/bash
# A/B of the fp32 3x3 loop order (two 16-channel passes per 32-channel chunk, round 4, against the single pass of rounds 1-3):
#   hipcc ... -DVC_TU_F16=0 -DVC_NO_F32_SUBS -c conv_k3.hip -> libvc_hip_nosubs.so
export VC_AUTOTUNE=0
SH="128,128,3,1,1,544,960,5 128,128,3,1,1,544,960,0 128,128,3,1,1,544,960,1 128,128,3,1,4,544,960,5 128,128,3,1,1,136,240,2 192,256,3,1,1,136,240,0 128,512,3,1,1,272,480,0 96,96,3,1,1,544,960,1"
echo "== two passes (this build)"; python tools/conv_bench.py --reps 20 $SH 2>&1 | grep conv
echo "== single pass (rounds 1-3)"; VC_HIP_LIB=video-compression_amd/libvc_hip_nosubs.so python tools/conv_bench.py --reps 20 $SH 2>&1 | grep conv
