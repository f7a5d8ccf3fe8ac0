#!/bin/bash
# Knock-out / ring-depth variants of the LDS-DMA kernel (make -C video-compression_amd/csrc dma_diag), one process per variant.
# usage: tools/dma_diag.sh <out.log> <shape> [variants...]
out=$1; shape=$2; shift; shift
vars="$@"
[ -z "$vars" ] && vars="0 1 2 3 7 11 19 27 32 100 101 102"
mkdir -p $(dirname $out); : > $out
for v in $vars; do
  echo "== VC_DMA_VARIANT=$v (1 no epilogue, 2 no vmcnt waits, 4 no MFMA, 8 no LDS reads, 16 no DMA, 32 no stagger; 100 ring 9, 101 ring 9 no epilogue, 102 ring 4)" >> $out
  VC_HIP_LIB=video-compression_amd/libvc_hip_dmadiag.so VC_DMA_VARIANT=$v timeout 200 python tools/dma_check.py --rounds 2 $shape 2>&1 | grep "^conv\|Error\|error\|cycles" >> $out
done
cat $out
