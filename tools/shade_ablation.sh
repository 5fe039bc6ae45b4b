#!/bin/bash
# one gpurun call: ablation of k_shade -- the first shading launch (maxdepth 1: 133 M vertices shaded once) with one part left
# out at a time.  The images are WRONG by construction; only the kernel time is read.
#   0 complete | 1 no pending-term stores | 2 no output stores (rays, state, pdInfo) | 3 no append (no barriers, no atomics)
#   4 triangle records from 1024 cached entries | 5 no direct lighting | 6 vertex not shaded at all (streams only)
#   7 one light table for all voxels | 8 no BSDF-sampled half of MIS | 9 no light-sampled half (f, pdf, shadow ray)
OUT=gpurun_out/exp9; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
B="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
{
for a in 0 7 8 9 5 0; do run abl$a PBRT_BENCH_MAXDEPTH=1 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_abl$a.so $B; done
} | tee $OUT/ab.txt
