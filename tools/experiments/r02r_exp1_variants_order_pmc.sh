#!/bin/bash
# one gpurun call: A/Bs of kernel variants + the ray-order experiment + the PMC diagnosis of what binds k_shade / k_trace
OUT=gpurun_out/exp1; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
  grep -h "ray-order experiment" $OUT/$name.err | tail -1
}
B="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
S=$PWD/gpurun_in_libpbrt_gpu_sort.so
{
run base $B
run s1 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_s1.so $B
run s1b PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_s1b.so $B
run ol PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_ol.so $B
run sort_off PBRT_GPU_LIB=$S $B
run sort_oct8 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=1 PG_SORT_BITS=8 $B
run sort_cell8 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=2 PG_SORT_BITS=8 $B
run sort_cell5 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=2 PG_SORT_BITS=5 $B
run sort_oct5 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=1 PG_SORT_BITS=5 $B
run sort_shadow_only PBRT_GPU_LIB=$S PG_SORT_RAYS=2 PG_SORT_MODE=1 PG_SORT_BITS=8 $B
run base_b $B
} | tee $OUT/ab.txt
# the sorted order must not change any result: parity tests through the experiment library
( PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=2 PBRT_SKIP_SLOW=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) | tee $OUT/pytest_sorted.log
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_s1.so PBRT_SKIP_SLOW=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 ) | tee $OUT/pytest_s1.log
timeout 1200 bash tools/pmc_diag.sh exp1/diag > $OUT/diag.log 2>&1
tail -150 $OUT/diag.log
