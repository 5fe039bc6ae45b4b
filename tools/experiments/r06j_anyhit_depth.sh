#!/bin/bash
# r06j: LDS stack depth of the any-hit launches on its own (PG_TRACE_DEPTH_ANY): their kernels have the registers for 8 resident blocks, 11 entries allow 7
OUT=gpurun_out/${1:-r06j}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:18s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C3="timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
M5="timeout 400 python bench.py --steps 2 --warmup 1 --grid 1582 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
for d in 11 10 9 8 6; do run c3_any$d PG_TRACE_DEPTH_ANY=$d $C3; done
for d in 11 10 8; do run m5_any$d PG_TRACE_DEPTH_ANY=$d $M5; done
for d in 11 10 9; do run div5m_any$d PG_TRACE_DEPTH_ANY=$d $DIV; done
for d in 10 9 8; do run div5m_x8_any$d PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_x8.so PG_TRACE_DEPTH_ANY=$d $DIV; done
run c3_any11_b PG_TRACE_DEPTH_ANY=11 $C3
} | tee $OUT/ab.txt
