cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof16; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/lhbdc16 --output-format csv -- python3 bench.py --precision fp16 --steps 1 --warmup 1 --no-cpu-baseline --kernel-table $O/kt_lhbdc16.json > $O/lhbdc16.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/icip16_4k --output-format csv -- python3 bench.py --model icip2024 --precision fp16 --resolution 2160p --steps 1 --warmup 1 --no-cpu-baseline --kernel-table $O/kt_icip16_4k.json > $O/icip16_4k.log 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
for d in lhbdc16 icip16_4k; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; head -25 "$f" | cut -c1-200; done
