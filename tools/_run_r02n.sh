OUT=gpurun_out/r02n; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/$name.err ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})"
}
run head PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_head.so
run new X=1
run head2 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_head.so
run new2 X=1
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ts -o trace -- python tools/ts_timing.py 192 108 4 > $OUT/ts_small.json 2> $OUT/ts_small.err ); cat $OUT/ts_small.json
find $OUT/prof_ts -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_ts.csv \; ; head -14 $OUT/kernel_stats_ts.csv | cut -c1-150; rm -rf $OUT/prof_ts
( timeout 900 python tools/fullsize_parity.py 5 --out=$OUT/fullsize_parity_config5.json > $OUT/fullsize_5.log 2>&1 ); grep "^{" $OUT/fullsize_5.log | cut -c1-900; tail -2 $OUT/fullsize_5.log | cut -c1-200
