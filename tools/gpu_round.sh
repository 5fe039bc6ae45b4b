#!/bin/bash
# One gpurun call: GPU parity tests, the headline bench, the rocprofv3 kernel-trace summary and the PMC traffic passes.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh TAG [quick|std|full]
#   quick: tests (without the slow full-size reference comparison) + bench + kernel stats
#   std  : + PMC traffic passes for config 3, the 5 M-triangle stand-in (bench + kernel stats) and the 10 M-triangle volpath stand-in (bench)
#   full : + PMC passes for the 5 M stand-in, the slow full-size comparisons with the reference binary, tile-serial and shard timing
#   final: quick + bench and kernel stats of the two divergent stand-ins (the bench run measures its PMC passes itself)
TAG=${1:-r02}; MODE=${2:-quick}
OUT=gpurun_out/$TAG
ulimit -c 0  # a GPU core dump fills the box
mkdir -p $OUT
export TMPDIR=/tmp
prof() {  # prof NAME bench-args...: kernel-trace stats of one bench run
  local name=$1; shift
  # --no-overlap: every kernel alone on the chip, as in the serialised frame bench.py takes its per-kernel times from (the timed frames of
  # a default run overlap any-hit with closest-hit launches: their durations under the profiler would not be the kernels' own)
  ( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --no-live-pmc --no-overlap "$@" > $OUT/bench_prof_$name.json 2> $OUT/prof_$name.err )
  find $OUT/prof_$name -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_$name.csv \;
  rm -rf $OUT/prof_$name
}
[ "$SKIP_TESTS" = 1 ] && echo 'tests skipped (SKIP_TESTS=1)' > $OUT/pytest_gpu.log || ( PBRT_SKIP_SLOW=1 timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -45 ) > $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 3 --warmup 1 2> $OUT/bench.err ) > $OUT/bench.json
prof cfg3
tail -5 $OUT/pytest_gpu.log; cat $OUT/bench.json; head -8 $OUT/kernel_stats_cfg3.csv
# BASELINE config 0 (the reference's own killeroo-simple.pbrt: a sphere light -> the quadric instantiation of k_trace), at its 400x400 @ 8 spp and at 64 spp
( timeout 300 python bench.py --workload config0 --steps 5 --warmup 2 2> $OUT/bench_config0.err ) > $OUT/bench_config0.json; cut -c1-400 $OUT/bench_config0.json
( timeout 300 python bench.py --workload config0 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench_config0_64spp.err ) > $OUT/bench_config0_64spp.json; cut -c1-300 $OUT/bench_config0_64spp.json
prof config0 --workload config0 --spp 64
head -6 $OUT/kernel_stats_config0.csv
if [ "$MODE" != quick ]; then
  if [ "$MODE" != final ]; then
  bash tools/pmc_traffic.sh $TAG/traffic_cfg3 > $OUT/traffic_cfg3.log 2>&1
  cp $OUT/traffic_cfg3/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
  ( timeout 900 python bench.py --steps 2 --warmup 1 --grid 1582 --spp 256 --no-cpu-baseline 2> $OUT/bench_5m.err ) > $OUT/bench_5m.json
  prof 5m --grid 1582 --spp 256
  cat $OUT/bench_5m.json; head -6 $OUT/kernel_stats_5m.csv
  ( timeout 600 python bench.py --steps 2 --warmup 1 --workload synthetic-vol --grid 2237 --spp 128 --no-cpu-baseline 2> $OUT/bench_10m_vol.err ) > $OUT/bench_10m_vol.json; cut -c1-400 $OUT/bench_10m_vol.json
  fi
  # the divergent stand-ins of configs 4 / 5 (instanced PLY meshes, textures, alpha masks, 8 materials): bench + kernel stats
  ( timeout 900 python bench.py --steps 2 --warmup 1 --workload divergent --tris 5000000 --spp 64 --no-cpu-baseline 2> $OUT/bench_div5m.err ) > $OUT/bench_div5m.json; cut -c1-300 $OUT/bench_div5m.json
  prof div5m --workload divergent --tris 5000000 --spp 64
  ( timeout 900 python bench.py --steps 2 --warmup 1 --workload divergent-vol --tris 10000000 --spp 32 --no-cpu-baseline 2> $OUT/bench_div10m_vol.err ) > $OUT/bench_div10m_vol.json; cut -c1-300 $OUT/bench_div10m_vol.json
  prof div10m_vol --workload divergent-vol --tris 10000000 --spp 32
  find $OUT -name '*counter_collection.csv' -size +4M -delete
fi
if [ "$MODE" = final ] || [ "$MODE" = full ]; then
  # configs 4 / 5 over the WHOLE frame at their own spp against the reference binary's image (its fingerprint, rendered beforehand where host
  # time is free: tools/fullsize_parity.py --reference-only), and what a 1/8 shard costs on one GPU
  ( timeout 900 python tools/fullsize_parity.py 41 51 "--fingerprint-in=tests/golden_large/fullframe_reference_fingerprint_config{config}.json" --out=$OUT/fullframe_parity_config4_5.json > $OUT/fullframe_parity.log 2>&1 ); cut -c1-400 $OUT/fullframe_parity.log | tail -2
  ( timeout 300 python tools/shard_timing.py > $OUT/shard_timing.json 2> $OUT/shard_timing.err ); cut -c1-600 $OUT/shard_timing.json
  ( timeout 600 python bench.py --steps 2 --warmup 1 --grid 1582 --no-cpu-baseline --no-hbm-regime 2> $OUT/bench_5m.err ) > $OUT/bench_5m.json; cut -c1-300 $OUT/bench_5m.json
  prof 5m --grid 1582
fi
if [ "$MODE" = full ]; then
  bash tools/pmc_traffic.sh $TAG/traffic_5m --steps 1 --warmup 0 --no-cpu-baseline --grid 1582 --spp 256 > $OUT/traffic_5m.log 2>&1
  cp $OUT/traffic_5m/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
  ( timeout 1500 python -m pytest tests/test_gpu_fullsize_reference.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_slow.log
  cp gpurun_out/fullsize_parity_config*.json $OUT/ 2>/dev/null
  tail -5 $OUT/pytest_slow.log
  find $OUT -name '*counter_collection.csv' -size +4M -delete
  ( timeout 600 python tools/ts_timing.py 960 540 16 > $OUT/ts_timing.json 2> $OUT/ts_timing.err ); cat $OUT/ts_timing.json
  # more random scenes than the suite holds, through the kernels (film, counters, rays against the oracle): ~2.5 GPU-minutes
  ( timeout 600 python -u tools/fuzz_emulated_device.py 120 220 2>&1 | grep -v '^Warning' > $OUT/fuzz_gpu.log ); tail -3 $OUT/fuzz_gpu.log
  ( timeout 300 python tools/shard_timing.py > $OUT/shard_timing.json 2> $OUT/shard_timing.err ); cat $OUT/shard_timing.json
fi
