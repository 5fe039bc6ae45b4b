#!/bin/bash
# r03u: occupancy of k_shade<2, .> again, now that its write flood is gone (DScene::self): launch bounds for 1 / 2 (default) / 3 waves per SIMD
OUT=gpurun_out/r03u; mkdir -p $OUT
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
for W in div5m div10mvol; do
  [ $W = div5m ] && ARGS="--workload divergent --tris 5000000 --spp 64" || ARGS="--workload divergent-vol --tris 10000000 --spp 32"
  run ${W}_default X=1
  run ${W}_w2 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_s2w2.so
  run ${W}_w3 PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_s2w3.so
done
