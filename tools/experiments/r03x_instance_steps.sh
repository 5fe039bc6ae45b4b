#!/bin/bash
# r03x: instance entry / exit as wave steps of their own (k_trace<., XP_INST...>): group size 4 / 8 / 12 (default) / 20 / 32
OUT=gpurun_out/r03x; mkdir -p $OUT
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
ARGS="--workload divergent --tris 5000000 --spp 64"
run div5m_g12 X=1
for g in 4 8 20 32; do run div5m_g$g PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_ig$g.so; done
ARGS="--workload divergent-vol --tris 10000000 --spp 32"
run div10mvol_g12 X=1
run div10mvol_g32 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_ig32.so
