#!/bin/bash
# Debugging aid: the strong-scaling block's per-frame R-D records must be the same bits (a) run after run, (b) with one rank or two
# ranks sharing the GPU, (c) whatever stale bytes the allocator hands out (bench.py --poison).  Prints which frames differ.
# VC_HIP_LIB selects another build of the library.
mkdir -p gpurun_out/period
A="--scaling strong --sequences 2 --frames-per-sequence 17 --steps 1 --warmup 0 --gops-per-step 2 --no-cpu-baseline $EXTRA"
VC_BENCH_DUMP_RECORDS=gpurun_out/period/rec_one.json python bench.py $A > /dev/null 2>gpurun_out/period/one.err
for i in 1 2 3 4; do
  VC_BENCH_DUMP_RECORDS=gpurun_out/period/rec_two_$i.json VC_BENCH_SHARE_GPU=1 VC_BENCH_BACKEND=gloo python bench.py $A --gpus 2 > /dev/null 2>gpurun_out/period/two_$i.err
  python - $i <<'PY'
import json, sys
a = json.load(open("gpurun_out/period/rec_one.json")); b = json.load(open(f"gpurun_out/period/rec_two_{sys.argv[1]}.json"))
bad = [(float.fromhex(x[0]), float.fromhex(x[1]), float.fromhex(x[2]), float.fromhex(x[3]) - float.fromhex(y[3]), float.fromhex(x[4]) - float.fromhex(y[4])) for x, y in zip(a, b) if x != y]
print(f"two-rank run {sys.argv[1]}: {len(a)} / {len(b)} records, {len(bad)} differ (video, frame, level, dPSNR, dbits):", bad[:12])
PY
done
