#!/bin/bash
# one gpurun call: record-fetch microbenchmark; staged record loads in k_trace / k_shade
OUT=gpurun_out/exp3; mkdir -p $OUT; export TMPDIR=/tmp
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
run n PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_n.so $B
run v8 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_v8.so $B
run st PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_st.so $B
run v8st PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_v8st.so $B
run v8st_5m PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_v8st.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid 1582 --spp 64
run base_5m timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --grid 1582 --spp 64
run base_b $B
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_v8st.so PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) | tee $OUT/pytest_v8st.log
