#!/bin/bash
# A/B of traversal configurations on a short bench run: each line = "ENV... | Mrays/s closest_ms/launch achieved GB/s shadow GB/s"
# usage: bash tools/ab.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...   (bench args via BENCH_ARGS)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS=${BENCH_ARGS:---steps 1 --warmup 1 --spp 8 --no-cpu-baseline}
i=0
for CFG in "$@"; do
  i=$((i+1))
  env $CFG timeout 300 python bench.py $ARGS > $OUT/ab$i.json 2> $OUT/ab$i.err
  python - "$CFG" $OUT/ab$i.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f"{sys.argv[1]:50s} | {j['value']:8.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  closest {r['avg_launch_ms']:7.3f} ms/launch {r['achieved']:7.1f} GB/s")
except Exception as e:
    print(f"{sys.argv[1]:50s} | FAILED {e}")
PY
done | tee $OUT/ab.txt
