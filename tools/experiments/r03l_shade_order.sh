#!/bin/bash
# r03l: shading grouped by material class inside windows of the main queue (k_shade_order) -- A/B on the divergent stand-ins.
# PG_SHADE_ORDER=0 = queue order (round 3's kernels before this), default = windows of 4096, variants 1024 / 16384.
OUT=gpurun_out/r03l; mkdir -p $OUT
run() { # name env... -- bench args
  local name=$1; shift
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
ARGS="--workload divergent --tris 5000000 --spp 64"
run div5m_queue_order PG_SHADE_ORDER=0
run div5m_w4096 PG_SHADE_ORDER=1
run div5m_w1024 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w1024.so
run div5m_w16384 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_w16384.so
ARGS="--workload divergent-vol --tris 10000000 --spp 32"
run div10mvol_queue_order PG_SHADE_ORDER=0
run div10mvol_w4096 PG_SHADE_ORDER=1
