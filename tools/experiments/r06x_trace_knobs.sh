#!/bin/bash
# r06x: the traversal's launch-time knobs on the round's closing kernels (no rebuild): chunk size, refill threshold, triangle-step weight, grid size, LDS stack depth.
# config 3 (flat kernels) and the config-4 stand-in (instanced kernels); full sample counts; ms per frame and per closest-hit launch.
OUT=gpurun_out/r06x; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name'.ljust(28), round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()})
except Exception as e: print('$name FAILED', e)"
}
C3="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc"
D5="timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc --workload divergent --tris 5000000 --spp 64"
for W in c3 d5; do
  [ $W = c3 ] && B="$C3" || B="$D5"
  run ${W}_base $B
  run ${W}_seg64 PG_TRACE_SEG=64 $B
  run ${W}_seg256 PG_TRACE_SEG=256 $B
  run ${W}_refill8 PG_TRACE_REFILL=8 $B
  run ${W}_refill12 PG_TRACE_REFILL=12 $B
  run ${W}_refill24 PG_TRACE_REFILL=24 $B
  run ${W}_triw4 PG_TRACE_TRIW=4 $B
  run ${W}_triw12 PG_TRACE_TRIW=12 $B
  run ${W}_triw24 PG_TRACE_TRIW=24 $B
  run ${W}_grid1024 PG_TRACE_GRID=1024 $B
  run ${W}_grid4096 PG_TRACE_GRID=4096 $B
  run ${W}_depth9 PG_TRACE_DEPTH=9 $B
  run ${W}_depth10 PG_TRACE_DEPTH=10 $B
  run ${W}_base2 $B
done 2>&1 | tee $OUT/knobs.txt
