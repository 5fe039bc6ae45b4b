ulimit -c 0
mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
( PBRT_SKIP_SLOW=1 timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 ) > $O/pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 2> $O/bench.err ) > $O/bench.json
( timeout 600 python bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-hbm-regime --no-live-pmc 2> $O/bench_no_overlap.err ) > $O/bench_no_overlap.json
tail -4 $O/pytest_gpu.log; cut -c1-700 $O/bench.json; echo; cut -c1-300 $O/bench_no_overlap.json; tail -3 $O/bench.err
