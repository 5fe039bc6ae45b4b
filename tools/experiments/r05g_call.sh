# r05g: the driver's own bench command, timed; the four slow full-size comparisons with the reference binary
ulimit -c 0
mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
/usr/bin/time -v -o $O/bench_time.txt timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
grep -E "Elapsed|Maximum resident" $O/bench_time.txt; cut -c1-200 $O/bench_driver_cmd.json
( timeout 1500 python -m pytest tests/test_gpu_fullsize_reference.py -m gpu -q --durations=4 2>&1 | tail -8 ) > $O/pytest_slow.log; cat $O/pytest_slow.log
cp gpurun_out/fullsize_parity_config*.json $O/ 2>/dev/null; ls $O
