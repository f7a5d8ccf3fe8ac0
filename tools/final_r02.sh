#!/bin/bash
# Round-end evidence collection on the MI355X box: profiles (tools/profile_r02.sh), the whole -m gpu suite, smoke(), and one
# bench line per configuration.  Everything lands under gpurun_out/final/.
set -u
F=gpurun_out/final; mkdir -p $F
bash tools/profile_r02.sh > $F/profile.log 2>&1
cp gpurun_out/prof_r02/traffic.json profiles/r02/traffic.json   # so that the bench lines below quote THIS build's traffic (copy it back into the repo afterwards)
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $F/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $F/smoke.log 2>&1
python bench.py --steps 3 --warmup 1 > $F/bench_default.log 2>&1
for cfg in "--precision fp16" "--model flex" "--model flex --precision fp16" "--model icip2024" "--model icip2024 --precision fp16" "--resolution 2160p" "--resolution 2160p --precision fp16" "--model icip2024 --precision fp16 --resolution 2160p"; do
  name=$(echo $cfg | tr -d '-' | tr ' ' '_')
  python bench.py $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_$name.json
done
python bench.py --scaling strong --sequences 2 --frames-per-sequence 33 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_strong_small.json
tail -3 $F/pytest_gpu.log; cat $F/smoke.log | tail -1; tail -1 $F/bench_default.log | cut -c1-300
