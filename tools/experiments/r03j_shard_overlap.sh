#!/bin/bash
# Round 3: what one rank of an N-GPU run does, with and without the any-hit launch on a second stream beside the closest-hit launch
# (the shards' launches are small: their tails weigh more), and the headline line on the final tree (PMC replay from r03i).
OUT=gpurun_out/r03j; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python tools/shard_timing.py > $OUT/shard_timing.json 2> $OUT/shard_timing.err ); python -c "
import json; j=json.load(open('$OUT/shard_timing.json')); print({n:(s['render_ms'], s['speedup_bound'], s['kernels_ms']) for n,s in j['shards'].items()})"
( PG_OVERLAP_SHADOW=1 timeout 600 python tools/shard_timing.py > $OUT/shard_timing_overlap.json 2> $OUT/shard_timing_overlap.err ); python -c "
import json; j=json.load(open('$OUT/shard_timing_overlap.json')); print('overlap', {n:(s['render_ms'], s['speedup_bound']) for n,s in j['shards'].items()})"
( timeout 900 python bench.py --steps 5 --warmup 2 2> $OUT/bench.err ) > $OUT/bench.json; cut -c1-500 $OUT/bench.json
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-regime > $OUT/bench_prof.json 2> $OUT/prof.err )
find $OUT/prof -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_cfg3.csv \; ; rm -rf $OUT/prof; head -5 $OUT/kernel_stats_cfg3.csv | cut -c1-120
