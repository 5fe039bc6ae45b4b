#!/bin/bash
# r03s: DScene::self (no per-lane scratch copy of the scene structure in the kernels that call the texture evaluators)
OUT=gpurun_out/r03s; mkdir -p $OUT
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
ARGS="--workload divergent --tris 5000000 --spp 64"; run div5m X=1
ARGS="--workload divergent-vol --tris 10000000 --spp 32"; run div10mvol X=1
PMC_PASSES=4 bash tools/pmc_pass.sh r03s/pmc_div5m --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --workload divergent --tris 5000000 --spp 16 > /dev/null 2>&1
grep -A22 "k_shade<2" gpurun_out/r03s/pmc_div5m/summary.txt | grep -E "k_shade|FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_WAIT_ANY "
