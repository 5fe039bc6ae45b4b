#!/bin/bash
# r06e: the new defaults (k_trace<., XP_INST ...> at 7 waves) -- GPU suite, the four workloads, and which of the diet's switches the triangle-only kernel wants
OUT=gpurun_out/${1:-r06e}; mkdir -p $OUT; export TMPDIR=/tmp
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
L() { echo PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_$1.so; }
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
VOL="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C3="timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
M5="timeout 400 python bench.py --steps 2 --warmup 1 --grid 1582 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C0="timeout 300 python bench.py --workload config0 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
run div5m $DIV
run div10m_vol $VOL
run c3 $C3
for v in 1 2 4 7; do run c3_fd$v $(L fd$v) $C3; done
run c3_b $C3
run m5 $M5
run m5_fd7 $(L fd7) $M5
run config0 $C0
} | tee $OUT/ab.txt
( PBRT_SKIP_SLOW=1 timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -8 ) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
