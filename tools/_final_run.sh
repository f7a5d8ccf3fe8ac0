O=gpurun_out/final4_r06; mkdir -p $O
python -m pytest tests/ -q -m gpu --durations=25 2>&1 | grep -v amdgpu.ids | tail -40 > $O/f_gpu_suite_durations.log; tail -2 $O/f_gpu_suite_durations.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
