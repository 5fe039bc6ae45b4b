OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "config0 or native_sharded or sphere or quadric or malformed" > $OUT/pytest_new.log 2>&1 ); grep -E "passed|failed|error" $OUT/pytest_new.log | tail -3
( PBRT_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime 2> $OUT/bench_rccl_one_rank.err ) > $OUT/bench_rccl_one_rank.json; cut -c1-240 $OUT/bench_rccl_one_rank.json; tail -2 $OUT/bench_rccl_one_rank.err
( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-regime 2> /dev/null ) > $OUT/bench_plain.json; cut -c1-130 $OUT/bench_plain.json
( timeout 600 python tools/substream_timing.py 64 2 > $OUT/substream_timing_k2.json 2> $OUT/substream_timing_k2.err ); cat $OUT/substream_timing_k2.json; tail -3 $OUT/substream_timing_k2.err
( timeout 600 python tools/substream_timing.py 64 3 > $OUT/substream_timing_k3.json 2> $OUT/substream_timing_k3.err ); cat $OUT/substream_timing_k3.json
