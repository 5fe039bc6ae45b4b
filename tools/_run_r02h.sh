OUT=gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_stats.so timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats.json 2> $OUT/stats.err; grep "k_trace<false> lanes" $OUT/stats.err
run() { # name env...
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items() if 'trace' in a})"
}
run base X=1
run triw4 PG_TRACE_TRIW=4
run triw12 PG_TRACE_TRIW=12
run triw16 PG_TRACE_TRIW=16
run triw24 PG_TRACE_TRIW=24
run refill4 PG_TRACE_REFILL=4
run refill16 PG_TRACE_REFILL=16
run refill24 PG_TRACE_REFILL=24
run seg64 PG_TRACE_SEG=64
run seg256 PG_TRACE_SEG=256
run depth8 PG_TRACE_DEPTH=8
run depth14 PG_TRACE_DEPTH=14
run grid1536 PG_TRACE_GRID=1536
run grid3072 PG_TRACE_GRID=3072
