#!/bin/bash
# The round's closing GPU call: tools/gpu_round.sh TAG final (benches, kernel stats, whole-frame parity of the config-4 / 5 stand-ins, shard timing),
# the N > 1 path's two pre-flights that one GPU allows (one rank over RCCL; two ranks sharing the device through gloo), and the slow full-size
# comparisons with the reference binary.   usage (repo root, GPU box): bash tools/gpu_final.sh TAG
TAG=${1:-final}; OUT=gpurun_out/$TAG
SKIP_TESTS=${SKIP_TESTS:-1} bash tools/gpu_round.sh $TAG final
( PBRT_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc 2> $OUT/bench_rccl_one_rank.err ) > $OUT/bench_rccl_one_rank.json; cut -c1-400 $OUT/bench_rccl_one_rank.json
( PBRT_BENCH_OVERSUBSCRIBE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-regime --no-live-pmc 2> $OUT/bench_2ranks_one_gpu_preflight.err ) > $OUT/bench_2ranks_one_gpu_preflight.json; cut -c1-400 $OUT/bench_2ranks_one_gpu_preflight.json
( timeout 1800 python -m pytest tests/test_gpu_fullsize_reference.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu_slow.log
cp gpurun_out/fullsize_parity_config*.json $OUT/ 2>/dev/null
tail -4 $OUT/pytest_gpu_slow.log
