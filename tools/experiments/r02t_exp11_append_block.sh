#!/bin/bash
# one gpurun call: k_shade's append (one atomic instruction for the three queues) and block size, now that the counter no longer hides them
OUT=gpurun_out/exp11; mkdir -p $OUT; export TMPDIR=/tmp
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
for v in m m64 b64 b256 m256; do run $v PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_$v.so $B; done
run new_b $B
run m_b PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_m.so $B
} | tee $OUT/ab.txt
( timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>$OUT/bench_gather.err ) | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(j['roofline'].get('gather')))" | tee $OUT/gather.json
