#!/bin/bash
# r03D: k_generate gives XCD r the r-th eighth of the slots (a band of tiles) instead of every eighth block: what each L2 has to hold
OUT=gpurun_out/r03D; mkdir -p $OUT
run() { local name=$1; shift
  ( env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime $ARGS 2> $OUT/$name.err ) > $OUT/$name.json
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[1]))
    ks = {k["kernel"].split(" ")[0]: round(k["avg_launch_ms"], 2) for k in b["roofline_kernels"]}
    print(sys.argv[2], round(b["value"], 1), "Mrays/s", round(b["ms_per_step"], 1), "ms", ks, b["kernel_ms_per_step"].get("generate"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
ARGS=""; run cfg3_round_robin PG_GEN_BANDS=0; run cfg3_bands PG_GEN_BANDS=1
ARGS="--grid 1582 --spp 128"; run 5m_round_robin PG_GEN_BANDS=0; run 5m_bands PG_GEN_BANDS=1
ARGS="--workload divergent --tris 5000000 --spp 64"; run div5m_round_robin PG_GEN_BANDS=0; run div5m_bands PG_GEN_BANDS=1
