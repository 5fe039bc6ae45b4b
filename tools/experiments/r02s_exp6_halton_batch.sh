#!/bin/bash
# one gpurun call: k_shade with the Halton dimensions of a vertex computed side by side (hb4), + the light table's row in one round trip (hb5)
OUT=gpurun_out/exp6; mkdir -p $OUT; export TMPDIR=/tmp
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
run base $B
run hb4 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb4.so $B
run hb5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb5.so $B
run prof_hb5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prof.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep -A12 "k_shade phases" $OUT/prof_hb5.err | tail -13
run vol_base timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32
run vol_hb5 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb5.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32
run base_b $B
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb5.so PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) | tee $OUT/pytest_hb5.log
