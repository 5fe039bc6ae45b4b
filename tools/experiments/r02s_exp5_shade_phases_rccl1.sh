#!/bin/bash
# one gpurun call: k_shade per-phase wave time (sampled), single-rank RCCL pre-flight of bench.py's collective path, triW sweep, microbenchmark
OUT=gpurun_out/exp5; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 120 ./gpurun_in_ubench.so 2>&1 ) | tee $OUT/ubench.txt
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
run prof PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_prof.so $B
grep -A12 "k_shade phases" $OUT/prof.err | tail -13
run new $B
run triw10 PG_TRACE_TRIW=10 $B
run triw12 PG_TRACE_TRIW=12 $B
run triw6 PG_TRACE_TRIW=6 $B
run rccl1 PBRT_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --out $OUT/rccl1.pfm
tail -5 $OUT/rccl1.err
run plain_out timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --out $OUT/plain.pfm
python - <<'PY'
import numpy as np
def pfm(p):
    f=open(p,'rb'); f.readline(); w,h=map(int,f.readline().split()); f.readline(); return np.frombuffer(f.read(),dtype='<f4')
try:
    a,b=pfm('gpurun_out/exp5/rccl1.pfm'),pfm('gpurun_out/exp5/plain.pfm'); print('RCCL single-rank image == plain image:', bool((a==b).all()))
except Exception as e: print('image comparison failed', e)
PY
} | tee $OUT/ab.txt
rm -f $OUT/*.pfm
