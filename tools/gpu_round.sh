#!/bin/bash
# One gpurun call: GPU parity tests, the headline bench, and the rocprofv3 kernel-trace summary.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log
( timeout 600 python bench.py --steps 2 --warmup 1 2> $OUT/bench.err ) > $OUT/bench.json
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err )
find $OUT/prof -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace*' -size +8M -delete
tail -5 $OUT/pytest_gpu.log; cat $OUT/bench.json; head -12 $OUT/kernel_stats.csv
bash tools/pmc_traffic.sh $TAG/traffic > $OUT/traffic.log 2>&1; cp $OUT/traffic/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
