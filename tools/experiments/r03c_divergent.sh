#!/bin/bash
# Round 3, GPU call 3: first timings of the kernels real scenes select: the divergent stand-ins of configs 4 / 5
# (scenes/gen_divergent.py), per-kernel times + rocprofv3 kernel stats; the headline again on the r02-equivalent traversal kernel;
# full-size parity of configs 4 / 5 windows against the reference binary.
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( env "$@" 2>$OUT/$name.err ) > $OUT/$name.json; python -c "
import json,sys
try:
    j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
    print('$name', round(j['value'],1), 'Mrays/s', round(j['ms_per_step'],1), 'ms', {a:round(b,1) for a,b in k.items()}, j['config']['workload'][:120])
except Exception as e: print('$name FAILED', e)"; tail -2 $OUT/$name.err
}
run cfg3 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run div5m timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent --tris 5000000 --spp 64
run div10m_vol timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload divergent-vol --tris 10000000 --spp 32
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_div5m -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload divergent --tris 5000000 --spp 64 > $OUT/bench_prof_div5m.json 2> $OUT/prof_div5m.err )
find $OUT/prof_div5m -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_div5m.csv \; ; rm -rf $OUT/prof_div5m; head -12 $OUT/kernel_stats_div5m.csv | cut -c1-200
( timeout 1500 python tools/fullsize_parity.py 4 5 --out=$OUT/fullsize_parity_config4_5.json > $OUT/fullsize_4_5.log 2>&1 ); tail -3 $OUT/fullsize_4_5.log | cut -c1-1500
