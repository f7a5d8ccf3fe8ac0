for i in 1 2; do
VC_HIP_LIB=$PWD/video-compression_amd/libvc_hip_r5epi.so python bench.py --no-cpu-baseline --no-strong-block --skip-extras --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('r5 epilogue ', round(d['value'],2), round(d['roofline']['avg_launch_ms'],3), d['conv_engine']['timed_region_tflops'])"
python bench.py --no-cpu-baseline --no-strong-block --skip-extras --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('record stores', round(d['value'],2), round(d['roofline']['avg_launch_ms'],3), d['conv_engine']['timed_region_tflops'])"
done
