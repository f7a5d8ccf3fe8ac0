# sweep of the 2-D tile order's band height (VC_TILE_BAND): time and FETCH_SIZE of the two dominant 7x7 kernels and the 3x3
set -u
OUT=gpurun_out/r2m; mkdir -p $OUT
export VC_AUTOTUNE=0
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for band in 1 2 4 8 16 34; do
  export VC_TILE_BAND=$band
  echo "== band $band"
  python tools/conv_bench.py --reps 10 64,32,7,1,4,1088,1920,7 32,64,7,1,4,1088,1920,7 128,128,3,1,4,544,960,5 2>&1 | grep conv
  python tools/conv_bench.py --precision fp16 --half-io --reps 10 64,32,7,1,4,1088,1920,7 128,128,3,1,4,544,960,5 2>&1 | grep conv
  for shape in 64,32,7,1,4,1088,1920,7 32,64,7,1,4,1088,1920,7 128,128,3,1,4,544,960,5; do
    rm -rf $OUT/f_$band; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f_$band --output-format csv -- python3 tools/conv_bench.py --reps 3 $shape > /dev/null 2>&1
    python3 - $OUT/f_$band $shape <<'PY'
import csv,glob,sys
vals=[float(r['Counter_Value']) for p in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(p)) if r['Counter_Name']=='FETCH_SIZE' and 'conv_' in r['Kernel_Name']]
print('   FETCH (doubled) GB', round(2*sum(vals)/len(vals)*1024/1e9,3), sys.argv[2])
PY
  done
done
