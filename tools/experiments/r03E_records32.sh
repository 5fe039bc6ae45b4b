#!/bin/bash
# r03E: closest hit through 32-B records (six stored bounds + selectors, the node's box in registers and on a 32-B-entry stack) against the
# 64-B child-pair records.  Same results (tests on the emulated device); config 3 and the 5 M-triangle scene.
OUT=gpurun_out/r03E; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime $ARGS 2> $OUT/$name.err ) > $OUT/$name.json
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
run cfg3_records64 PG_RECORDS32=0
for d in 1 2 3 4; do run cfg3_records32_lds$d PG_RECORDS32=1 PG_TRACE_DEPTH32=$d; done
ARGS="--grid 1582 --spp 128"
run 5m_records64 PG_RECORDS32=0
run 5m_records32_lds2 PG_RECORDS32=1 PG_TRACE_DEPTH32=2
run 5m_records32_lds3 PG_RECORDS32=1 PG_TRACE_DEPTH32=3
