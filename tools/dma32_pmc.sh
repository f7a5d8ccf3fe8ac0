#!/bin/bash
# Does the fp32 LDS-DMA kernel lose matrix time to the DMA ISSUE or to a lower clock under memory traffic?  One PMC pass per
# knock-out variant of the diagnostic library (VC_DMA_VARIANT: 0 full, 16 no DMA, 25 MFMAs + barriers only, 2048 no weight DMA):
# GRBM_GUI_ACTIVE / kernel time = clock; SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) = matrix pipe busy.
set -u
OUT=gpurun_out/prof_r04/dma32_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export VC_AUTOTUNE=0 VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_dmadiag.so
for v in 0 16 25 2048 1024; do
  export VC_DMA_VARIANT=$v
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/v$v --output-format csv -- python3 tools/conv_bench.py --reps 5 64,32,7,1,4,1088,1920,8 > $OUT/v$v.log 2>&1
  python3 - <<PY
import csv, glob
rows = [r for f in glob.glob("$OUT/v$v/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "conv_dma" in r["Kernel_Name"]]
trace = [r for f in glob.glob("$OUT/v$v/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "conv_dma" in r["Kernel_Name"]]
def mean(name):
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
    return sum(v) / max(1, len(v))
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace) / max(1, len(trace))
gui, mfma = mean("GRBM_GUI_ACTIVE"), mean("SQ_VALU_MFMA_BUSY_CYCLES")
print(f"variant $v: {len(trace)} launches, {dur / 1e6:.3f} ms under the profiler, GRBM_GUI_ACTIVE {gui:.4g} -> clock {gui / 8 / dur:.3f} GHz (XCD-summed / 8), "
      f"matrix pipe busy {mfma / (gui / 8 * 1024) * 100:.1f} %")
PY
done
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
