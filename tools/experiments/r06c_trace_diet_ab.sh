#!/bin/bash
# r06c: the register diet of k_trace (wave-level statistics counters, mbcnt, hit record to memory at acceptance, Sz from the reciprocals, the pending
# instance in `cur`, the world ray parked in LDS) at 5 / 6 resident waves and the LDS stack depths that fit; the triangle-only kernel at 8.
OUT=gpurun_out/${1:-r06c}; mkdir -p $OUT; export TMPDIR=/tmp
ulimit -c 0
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json
  python - $name $OUT/$name.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print(f"{sys.argv[1]:16s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:8.1f} ms/frame  " + "  ".join(f"{a} {b:.1f}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
L() { echo PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_$1.so; }
DIV="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
VOL="timeout 400 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
C3="timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
M5="timeout 400 python bench.py --steps 2 --warmup 1 --grid 1582 --no-cpu-baseline --no-live-pmc --no-hbm-regime"
{
echo "# divergent 5 M (config-4 stand-in)"
run base $DIV
run d1_d11 $(L d1) $DIV
run d1_d13 $(L d1) PG_TRACE_DEPTH=13 $DIV
run d1_d12 $(L d1) PG_TRACE_DEPTH=12 $DIV
run d1w6_d13 $(L d1w6) PG_TRACE_DEPTH=13 $DIV
run d2_d12 $(L d2) PG_TRACE_DEPTH=12 $DIV
run d2w6_d10 $(L d2w6) PG_TRACE_DEPTH=10 $DIV
run d3_d11 $(L d3) $DIV
run d3w6_d8 $(L d3w6) PG_TRACE_DEPTH=8 $DIV
run d4_d10 $(L d4) PG_TRACE_DEPTH=10 $DIV
run d1_d13_t12 $(L d1) PG_TRACE_DEPTH=13 PG_TRACE_TRIW=12 $DIV
run d1_d13_t12_r8 $(L d1) PG_TRACE_DEPTH=13 PG_TRACE_TRIW=12 PG_TRACE_REFILL=8 $DIV
run base_b $DIV
echo "# divergent 10 M volpath (config-5 stand-in)"
run vol_base $VOL
run vol_d1_d13 $(L d1) PG_TRACE_DEPTH=13 $VOL
run vol_d2w6_d10 $(L d2w6) PG_TRACE_DEPTH=10 $VOL
echo "# config 3"
run c3_base $C3
run c3_d1_d11 $(L d1) $C3
run c3_d1_d10 $(L d1) PG_TRACE_DEPTH=10 $C3
run c3_d1f8_d10 $(L d1f8) PG_TRACE_DEPTH=10 $C3
run c3_d1f8_d11 $(L d1f8) $C3
echo "# 5 M-triangle heightfield"
run m5_base $M5
run m5_d1_d11 $(L d1) $M5
run m5_d1f8_d10 $(L d1f8) PG_TRACE_DEPTH=10 $M5
} | tee $OUT/ab.txt
( PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_d2w6.so PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_anyhit_order.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 ) > $OUT/pytest_d2w6.log
tail -3 $OUT/pytest_d2w6.log
