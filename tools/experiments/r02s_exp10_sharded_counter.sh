#!/bin/bash
# one gpurun call: sharded light-test counters (+ batched division-free Halton loops, light-table row in one round trip)
OUT=gpurun_out/exp10; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
B="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
{
run new $B
run norow PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_norow.so $B
run new_b $B
run norow_b PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_norow.so $B
run new_5m timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid 1582 --spp 64
run new_vol timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32
} | tee $OUT/ab.txt
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) | tee $OUT/pytest.log
