#!/bin/bash
# Round-5 last refresh after the packed lobe records went in: suite, headline bench + kernel stats, the two divergent stand-ins (bench with live
# PMC + kernel stats), configs 4 / 5 over the whole frame against the reference fingerprints.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export TMPDIR=/tmp
OUT=gpurun_out/r05n; mkdir -p $OUT
prof() { local name=$1; shift
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --no-live-pmc --no-overlap "$@" > $OUT/bench_prof_$name.json 2> $OUT/prof_$name.err )
  find $OUT/prof_$name -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_$name.csv \;
  rm -rf $OUT/prof_$name; }
( PBRT_SKIP_SLOW=1 timeout 900 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -45 ) > $OUT/pytest_gpu.log; grep -E 'passed|failed|error' $OUT/pytest_gpu.log
( timeout 600 python bench.py --steps 3 --warmup 1 2> $OUT/bench.err ) > $OUT/bench.json; cut -c1-200 $OUT/bench.json
prof cfg3
( timeout 600 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline 2> $OUT/bench_div5m.err ) > $OUT/bench_div5m.json; cut -c1-200 $OUT/bench_div5m.json
prof div5m --workload divergent --tris 5000000 --spp 64
( timeout 600 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline 2> $OUT/bench_div10m_vol.err ) > $OUT/bench_div10m_vol.json; cut -c1-200 $OUT/bench_div10m_vol.json
prof div10m_vol --workload divergent-vol --tris 10000000 --spp 32
( timeout 600 python tools/fullsize_parity.py 41 51 "--fingerprint-in=tests/golden_large/fullframe_reference_fingerprint_config{config}.json" --out=$OUT/fullframe_parity_config4_5.json > $OUT/fullframe_parity.log 2>&1 ); cut -c1-400 $OUT/fullframe_parity.log | tail -2
find $OUT -name '*counter_collection.csv' -delete
