#!/bin/bash
# r06d: k_trace<., XP_INST | XP_ALPHA> after the register diet + the one-record, row-by-row instance entry: 6 / 7 resident waves (closest hit), 7 / 8 (any hit)
OUT=gpurun_out/${1:-r06d}; mkdir -p $OUT; export TMPDIR=/tmp
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
{
echo "# divergent 5 M (config-4 stand-in)"
run base $DIV
run e2w6_d13 $(L e2w6) PG_TRACE_DEPTH=13 $DIV
run e2w6_d11 $(L e2w6) $DIV
run e2w7_d11 $(L e2w7) $DIV
run e2w7_d10 $(L e2w7) PG_TRACE_DEPTH=10 $DIV
run e2w78_d10 $(L e2w78) PG_TRACE_DEPTH=10 $DIV
run e2w78_d9 $(L e2w78) PG_TRACE_DEPTH=9 $DIV
run e2w7_d11_t12 $(L e2w7) PG_TRACE_TRIW=12 $DIV
run e2w7_d11_t12_r8 $(L e2w7) PG_TRACE_TRIW=12 PG_TRACE_REFILL=8 $DIV
run e2w7_d11_t12_r12 $(L e2w7) PG_TRACE_TRIW=12 PG_TRACE_REFILL=12 $DIV
run e2w7_d11_t16_r8 $(L e2w7) PG_TRACE_TRIW=16 PG_TRACE_REFILL=8 $DIV
run e2w6_d13_t12_r8 $(L e2w6) PG_TRACE_DEPTH=13 PG_TRACE_TRIW=12 PG_TRACE_REFILL=8 $DIV
run e2w7_any_t12 $(L e2w7) PG_TRACE_TRIW_ANY=12 $DIV
run e2w7_any_t24 $(L e2w7) PG_TRACE_TRIW_ANY=24 $DIV
run e2w7_any_r16 $(L e2w7) PG_TRACE_REFILL_ANY=16 $DIV
run e2w7_any_r48 $(L e2w7) PG_TRACE_REFILL_ANY=48 $DIV
run base_b $DIV
echo "# divergent 10 M volpath (config-5 stand-in)"
run vol_base $VOL
run vol_e2w6_d13 $(L e2w6) PG_TRACE_DEPTH=13 $VOL
run vol_e2w7_d11 $(L e2w7) $VOL
run vol_e2w7_d11_t12_r8 $(L e2w7) PG_TRACE_TRIW=12 PG_TRACE_REFILL=8 $VOL
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_e2w7.so PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_anyhit_order.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5 ) > $OUT/pytest_e2w7.log
cat $OUT/pytest_e2w7.log
