#!/bin/bash
# Round 6: where a phase of the split period kernels goes (make -C video-compression_amd/csrc split_diag): shader-clock totals per segment
export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_splitdiag.so VC_AUTOTUNE=0
for shape in 128,128,3,1,4,544,960 128,128,3,1,1,544,960 32,64,7,1,4,1088,1920; do
  echo "== $shape (instrumented, then plain)"
  VC_SPLIT_VARIANT=64 python tools/conv_bench.py --split --split-out --reps 5 $shape 2>&1 | grep -v amdgpu.ids
  python tools/conv_bench.py --split --split-out --reps 5 $shape 2>&1 | grep -v amdgpu.ids
done
