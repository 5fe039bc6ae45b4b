#!/bin/bash
# r03m: volpath shading with the medium sample drawn ahead (k_shade_order<true>) and medium / surface vertices in separate waves.
OUT=gpurun_out/r03m; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime $ARGS 2> $OUT/$name.err ) > $OUT/$name.json
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[1]))
    ks = {k["kernel"].split(" ")[0]: round(k["avg_launch_ms"], 2) for k in b["roofline_kernels"]}
    print(sys.argv[2], round(b["value"], 1), "Mrays/s", round(b["ms_per_step"], 1), "ms", ks)
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
ARGS="--workload synthetic-vol --grid 2237 --spp 128"
run syn10mvol_queue_order PG_SHADE_ORDER=0
run syn10mvol_ordered PG_SHADE_ORDER=1
ARGS="--workload divergent-vol --tris 10000000 --spp 32"
run div10mvol_queue_order PG_SHADE_ORDER=0
run div10mvol_ordered PG_SHADE_ORDER=1
