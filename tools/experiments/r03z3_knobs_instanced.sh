#!/bin/bash
# r03z3: the closest-hit knobs on the instanced stand-in (k_trace<0, INST|ALPHA>: 5 waves per SIMD instead of 7)
OUT=gpurun_out/r03z; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --workload divergent --tris 5000000 --spp 64 2> $OUT/$name.err ) > $OUT/$name.json
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[1]))
    ks = {k["kernel"].split(" ")[0]: round(k["avg_launch_ms"], 2) for k in b["roofline_kernels"]}
    print(sys.argv[2], round(b["value"], 1), "Mrays/s", round(b["ms_per_step"], 1), "ms", ks)
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run div_default X=1
for r in 8 24 32; do run div_refill$r PG_TRACE_REFILL=$r; done
for w in 4 12 16; do run div_triw$w PG_TRACE_TRIW=$w; done
run div_depth8 PG_TRACE_DEPTH=8
run div_depth14 PG_TRACE_DEPTH=14
