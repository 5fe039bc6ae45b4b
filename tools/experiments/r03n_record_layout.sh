#!/bin/bash
# r03n: which record shares a 128-B line with which (an L2 miss fills the whole line).  0 = depth-first (the product so far),
# 1 = a node with its larger child, 2 = the two children of a node.  Host-side renumbering only; results are identical.
OUT=gpurun_out/r03n; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime $ARGS 2> $OUT/$name.err ) > $OUT/$name.json
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
for L in 0 1 2; do run cfg3_layout$L PG_RECORD_LAYOUT=$L; done
ARGS="--grid 1582 --spp 128"
for L in 0 1 2; do run 5m_layout$L PG_RECORD_LAYOUT=$L; done
