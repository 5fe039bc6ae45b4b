#!/bin/bash
# one gpurun call: triangle records on their own 64-B lines (stride 4 float4, the new build) against packed 48-B records (t3); volpath with batched Halton draws
OUT=gpurun_out/exp12; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
B="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
M="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid 1582 --spp 64"
V="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32"
T=$PWD/gpurun_in_libpbrt_gpu_t3.so
{
run s4 $B
run s3 PBRT_GPU_LIB=$T $B
run s4_b $B
run s3_b PBRT_GPU_LIB=$T $B
run s4_5m $M
run s3_5m PBRT_GPU_LIB=$T $M
run s4_vol $V
} | tee $OUT/ab.txt
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) | tee $OUT/pytest.log
