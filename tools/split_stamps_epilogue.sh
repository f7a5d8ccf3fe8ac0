export VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_splitdiag.so VC_AUTOTUNE=0
for extra in "" "--residual-split"; do
for v in 64 0; do echo "== variant $v $extra"; VC_SPLIT_VARIANT=$v python tools/conv_bench.py --split --split-out $extra --reps 5 128,128,3,1,4,544,960 2>&1 | grep -v amdgpu.ids; done; done
