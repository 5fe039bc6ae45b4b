#!/bin/bash
# r03z2: refill threshold / leaf weighting of the any-hit launches alone (closest hit stays at 16 / 8)
OUT=gpurun_out/r03z; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime $ARGS 2> $OUT/$name.err ) > $OUT/$name.json
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[1]))
    ks = {k["kernel"].split(" ")[0]: round(k["avg_launch_ms"], 2) for k in b["roofline_kernels"]}
    print(sys.argv[2], round(b["value"], 1), "Mrays/s", round(b["ms_per_step"], 1), "ms", ks)
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
ARGS=""
run any_16_8 PG_TRACE_REFILL_ANY=16 PG_TRACE_TRIW_ANY=8
run any_32_16 X=1
run any_32_8 PG_TRACE_REFILL_ANY=32 PG_TRACE_TRIW_ANY=8
run any_32_24 PG_TRACE_REFILL_ANY=32 PG_TRACE_TRIW_ANY=24
run any_40_16 PG_TRACE_REFILL_ANY=40 PG_TRACE_TRIW_ANY=16
run any_48_24 PG_TRACE_REFILL_ANY=48 PG_TRACE_TRIW_ANY=24
run any_48_32 PG_TRACE_REFILL_ANY=48 PG_TRACE_TRIW_ANY=32
ARGS="--workload divergent --tris 5000000 --spp 64"
run div_any_16_8 PG_TRACE_REFILL_ANY=16 PG_TRACE_TRIW_ANY=8
run div_any_32_16 X=1
run div_any_48_24 PG_TRACE_REFILL_ANY=48 PG_TRACE_TRIW_ANY=24
