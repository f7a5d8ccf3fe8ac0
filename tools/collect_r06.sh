bash tools/profile_r06.sh > gpurun_out/prof_r06_headline.log 2>&1
bash tools/profile_split_r06.sh > gpurun_out/prof_r06_split.log 2>&1
bash tools/final_r06.sh > gpurun_out/final_r06_summary.log 2>&1
python bench.py > gpurun_out/final_r06/h_bench_line_final.json 2> gpurun_out/final_r06/h_bench.err
tail -5 gpurun_out/prof_r06_headline.log | cut -c1-300; tail -12 gpurun_out/prof_r06_split.log | cut -c1-400; cat gpurun_out/final_r06_summary.log; tail -c 1500 gpurun_out/final_r06/h_bench_line_final.json
