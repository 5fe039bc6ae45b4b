OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
# 1. config 0 with the quadric-only instantiation (this build) -- the r04a numbers of the general instantiation are in profiles/r04a_*config0*
( timeout 300 python bench.py --workload config0 --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench_config0.err ) > $OUT/bench_config0.json; cut -c1-260 $OUT/bench_config0.json
( timeout 300 python bench.py --workload config0 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench_config0_64spp.err ) > $OUT/bench_config0_64spp.json; cut -c1-260 $OUT/bench_config0_64spp.json
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c0 -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --workload config0 --spp 64 > $OUT/bench_prof_config0.json 2> $OUT/prof_config0.err ); find $OUT/prof_c0 -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_config0.csv \; ; rm -rf $OUT/prof_c0; head -5 $OUT/kernel_stats_config0.csv | cut -c1-200
# 2. the new GPU tests: config 0 in both orders with the new instantiation, sphere / quadric rays, native sharded render incl. the one-rank RCCL gather
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "config0 or native_sharded or sphere or quadric or malformed" 2>&1 | tail -5 ) > $OUT/pytest_new.log; cat $OUT/pytest_new.log
# 3. any-hit launches on a second stream at N = 1
for ov in 0 1; do ( PG_OVERLAP_SHADOW=$ov timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime 2> $OUT/bench_overlap$ov.err ) > $OUT/bench_overlap$ov.json; echo overlap $ov; cut -c1-120 $OUT/bench_overlap$ov.json; done
# 4. one-rank RCCL pre-flight of bench.py's gather on the final tree
( PBRT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime 2> $OUT/bench_rccl_one_rank.err ) > $OUT/bench_rccl_one_rank.json; cut -c1-200 $OUT/bench_rccl_one_rank.json; tail -2 $OUT/bench_rccl_one_rank.err
# 5. PMC passes of the two divergent workloads
bash tools/pmc_traffic.sh r04b/traffic_div5m --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --workload divergent --tris 5000000 --spp 16 > $OUT/traffic_div5m.log 2>&1
cp $OUT/traffic_div5m/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
bash tools/pmc_traffic.sh r04b/traffic_div10m --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-regime --workload divergent-vol --tris 10000000 --spp 8 > $OUT/traffic_div10m.log 2>&1
cp $OUT/traffic_div10m/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
find $OUT -name '*counter_collection.csv' -size +2M -delete
tail -3 $OUT/traffic_div5m.log $OUT/traffic_div10m.log
( timeout 300 python tools/shard_timing.py > $OUT/shard_timing.json 2> $OUT/shard_timing.err ); cut -c1-300 $OUT/shard_timing.json
