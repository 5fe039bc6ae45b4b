#!/bin/bash
# r06b: (1) how much k_trace<., XP_INST | XP_ALPHA> depends on resident waves (PG_TRACE_LDS_PAD holds the baseline at 4 and 3 blocks per CU),
# (2) combinations of r06a's small winners, (3) alpha-step group sizes, (4) the step-policy knobs on this scene.  One gpurun call.
OUT=gpurun_out/${1:-r06b}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:14s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
run base $DIV
run pad4w PG_TRACE_LDS_PAD=10240 $DIV
run pad3w PG_TRACE_LDS_PAD=18432 $DIV
run depth8 PG_TRACE_DEPTH=8 $DIV
run depth14 PG_TRACE_DEPTH=14 $DIV
for v in c3 c4 c3w6 agrp2 agrp4; do run $v PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_$v.so $DIV; done
run c4w6_d10 PG_TRACE_DEPTH=10 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_c4w6.so $DIV
run c3w6_d8 PG_TRACE_DEPTH=8 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_c3w6.so $DIV
for t in 4 12 16 24; do run triw$t PG_TRACE_TRIW=$t $DIV; done
for t in 8 24 32; do run refill$t PG_TRACE_REFILL=$t $DIV; done
for t in 64 256; do run seg$t PG_TRACE_SEG=$t $DIV; done
run base_b $DIV
} | tee $OUT/ab.txt
