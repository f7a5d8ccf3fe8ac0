#!/bin/bash
# Round-6 profile collection on the MI355X box (run from the repo root through gpurun): the HEADLINE run itself under
# rocprofv3 --kernel-trace --stats (the program directly after "--": no env / shell hop), folded per kernel name and per
# (kernel, grid); bench.py names the kernel instance its dominant launch ran (roofline.kernel_symbol).
set -u
OUT=gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/headline --output-format csv -- python3 bench.py --no-cpu-baseline --no-strong-block --skip-extras --steps 3 --warmup 1 > $OUT/a_headline_under_rocprofv3_line.json 2> $OUT/a_headline.err
python3 tools/trace_summary.py $OUT/headline $OUT/a_kernel_trace_by_grid.json
cp $(find $OUT/headline -name "*kernel_stats.csv" | head -1) $OUT/a_rocprofv3_kernel_stats.csv
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
rm -rf $OUT/headline
tail -c 900 $OUT/a_headline_under_rocprofv3_line.json
ls -la $OUT
