#!/bin/bash
# one gpurun call: k_shade with batched, division-free Halton digit loops: hb6 = with the light table row in one round trip, hb7 = without
OUT=gpurun_out/exp7; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 120 ./gpurun_in_ubench.so 2>&1 ) | tee $OUT/ubench.txt
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
run hb6 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb6.so $B
run hb7 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb7.so $B
run prof_hb7 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prof.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep -A12 "k_shade phases" $OUT/prof_hb7.err | tail -13
run vol_base timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32
run vol_hb7 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb7.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload synthetic-vol --spp 32
run base_b $B
run hb6_b PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb6.so $B
run hb7_b PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb7.so $B
run base_c $B
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_hb6.so PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) | tee $OUT/pytest_hb6.log
