#!/bin/bash
# one gpurun call: restructured k_trace (v6), k_shade over virtual blocks (persistent grid sizes), per-phase wave time of k_shade
OUT=gpurun_out/exp2; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
B="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
P=$PWD/gpurun_in_libpbrt_gpu_p.so
{
run base $B
run v6 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_v6.so $B
run p_grid0 PBRT_GPU_LIB=$P PG_SHADE_GRID=0 $B
run p_grid2048 PBRT_GPU_LIB=$P PG_SHADE_GRID=2048 $B
run p_grid4096 PBRT_GPU_LIB=$P PG_SHADE_GRID=4096 $B
run p_grid16384 PBRT_GPU_LIB=$P PG_SHADE_GRID=16384 $B
run all_grid2048 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_all.so PG_SHADE_GRID=2048 $B
run prof_grid0 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prof.so PG_SHADE_GRID=0 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline
grep -A12 "k_shade phases" $OUT/prof_grid0.err | tail -13
run prof_grid2048 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prof.so PG_SHADE_GRID=2048 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline
grep -A12 "k_shade phases" $OUT/prof_grid2048.err | tail -13
run base_b $B
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_all.so PG_SHADE_GRID=2048 PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) | tee $OUT/pytest_all_2048.log
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_all.so PG_SHADE_GRID=0 PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) | tee $OUT/pytest_all_0.log
