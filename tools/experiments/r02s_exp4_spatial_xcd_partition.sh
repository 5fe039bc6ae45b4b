#!/bin/bash
# one gpurun call: what would a SPATIAL partition of the rays across the 8 XCDs (each L2 sees one part of space) buy k_trace?
OUT=gpurun_out/exp4; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
  grep -h "ray-order experiment" $OUT/$name.err | tail -1
}
B="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
M="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid 1582 --spp 64"
S=$PWD/gpurun_in_libpbrt_gpu_sort.so
{
run sort_off PBRT_GPU_LIB=$S $B
run spatial8 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=3 PG_SORT_BITS=8 $B
run spatial4 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=3 PG_SORT_BITS=4 $B
run spatial2 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=3 PG_SORT_BITS=2 $B
run inregion8 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=2 PG_SORT_BITS=8 $B
run 5m_off PBRT_GPU_LIB=$S $M
run 5m_spatial8 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=3 PG_SORT_BITS=8 $M
run 5m_spatial4 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=3 PG_SORT_BITS=4 $M
run 5m_inregion8 PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=2 PG_SORT_BITS=8 $M
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$S PG_SORT_RAYS=7 PG_SORT_MODE=3 PBRT_SKIP_SLOW=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not sampler_" 2>&1 | tail -4 ) | tee $OUT/pytest_spatial.log
