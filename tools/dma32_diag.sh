#!/bin/bash
# knock-out / stamp variants of the fp32 LDS-DMA 7x7 64 -> 32 instance (diagnostic library: make -C video-compression_amd/csrc dma_diag)
export VC_AUTOTUNE=0 VC_HIP_LIB=video-compression_amd/libvc_hip_dmadiag.so
for v in ${VARIANTS:-0 1 8 16 25 32 64 128}; do
  echo "== VC_DMA_VARIANT=$v"
  VC_DMA_VARIANT=$v python tools/conv_bench.py --reps 10 64,32,7,1,4,1088,1920,8 2>&1 | grep -v amdgpu.ids
done
