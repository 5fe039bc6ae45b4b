#!/bin/bash
# one gpurun call: quad-cooperative record fetch in k_trace (experiment build -DPG_QUAD_FETCH) at 5 / 6 waves per SIMD
OUT=gpurun_out/exp13; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
B="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
{
run base $B
run q5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_q5.so $B
run q5_d15 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_q5.so PG_TRACE_DEPTH=15 $B
run q6 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_q6.so $B
run q6_d13 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_q6.so PG_TRACE_DEPTH=13 $B
run q5_5m PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_q5.so PG_TRACE_DEPTH=15 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid 1582 --spp 64
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_q5.so PBRT_SKIP_SLOW=1 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) | tee $OUT/pytest_q5.log
