#!/bin/bash
# Last refresh of the headline line with the final library: bench.py as the driver runs it (timed frames, hbm_regime, PMC child passes, reference thread
# sweep), then the kernel-trace summary of the same workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export TMPDIR=/tmp
OUT=gpurun_out/r05r; mkdir -p $OUT
( timeout 260 python bench.py --steps 3 --warmup 1 2> $OUT/bench.err ) > $OUT/bench.json; cut -c1-220 $OUT/bench.json
( timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg3 -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --no-live-pmc --no-overlap > $OUT/bench_prof_cfg3.json 2> $OUT/prof_cfg3.err )
find $OUT/prof_cfg3 -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_cfg3.csv \;
rm -rf $OUT/prof_cfg3; head -5 $OUT/kernel_stats_cfg3.csv
